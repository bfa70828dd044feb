#!/usr/bin/env python
"""Headline benchmark: NRMS-shape two-tower TRAINING step (news encoder = user encoder = one
multi-head self-attention layer + additive pooling, T=32, L=50, K=4, 300-d, 12 heads) on
synthetic MIND-large-shaped data, fp32, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or — when
WORLD_SIZE is not set — bench.py starts its N ranks ITSELF, as fresh child processes created before the parent has
made any GPU call (the reference's entry does the same with mp.spawn, main/twotower.py:55-61).  One process per GPU,
torch.distributed "nccl" = RCCL over xGMI.

One "step" = forward + backward + Adam over one batch of B impressions per GPU (B = 32, the
reference's -bs default).  Rank 0 prints ONE JSON line: train impressions/s over all ranks
(max-over-ranks time, barrier + synchronize on both sides), plus
  roofline     — the dominant kernel family (f32-MFMA GEMM): algorithmic flops / HIP-event time
  cpu_baseline — the CPU oracle (oracle/, torch fp32) timed on this host's cores, N=1 only
  eval         — cached evaluation: news-cache build (news/s) and impression scoring (rows/s)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "news-recommendation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

V, E, D, H, T, L, K = 30522, 300, 300, 12, 32, 50, 4
NEWS_TRAIN = 101527 + 1        # MINDlarge_train rows (+1 pad row), reference manager.py:175
NEWS_DEV = 72023 + 1           # MINDlarge_dev
PEAK_F32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0


def synth_table(g, Nn):
    """MIND-shaped token table: row 0 all PAD; [CLS]=101 w... [SEP]=102 then PAD"""
    lens = torch.randint(5, T + 1, (Nn,), generator=g)
    lens[0] = 0
    pos = torch.arange(T)[None, :]
    msk = (pos < lens[:, None]).long()
    ids = torch.randint(1000, V, (Nn, T), generator=g) * msk
    ids[:, 0] = torch.where(lens > 0, torch.tensor(101), torch.tensor(0))
    ids[torch.arange(Nn), (lens - 1).clamp(min=0)] = torch.where(lens > 0, torch.tensor(102), torch.tensor(0))
    return ids, msk


def synth_train_batch(g, B, Nn, ids, msk):
    cdd_idx = torch.randint(1, Nn, (B, K + 1), generator=g)
    nneg = torch.randint(1, 41, (B,), generator=g)                       # negatives pool 1..40, zero-padded below K
    for b in torch.nonzero(nneg < K).flatten().tolist():
        cdd_idx[b, 1 + int(nneg[b]):] = 0
    hl = torch.randint(0, 71, (B,), generator=g).clamp(max=L)            # history 0..70 truncated to L
    his_idx = torch.randint(1, Nn, (B, L), generator=g)
    live = torch.arange(L)[None, :] < hl[:, None]
    his_idx = his_idx * live
    his_mask = live.long()
    his_mask[hl == 0, 0] = 1
    return dict(cdd_idx=cdd_idx, his_idx=his_idx, his_mask=his_mask, cdd_token_id=ids[cdd_idx], cdd_attn_mask=msk[cdd_idx],
                his_token_id=ids[his_idx], his_attn_mask=msk[his_idx], label=torch.zeros(B, dtype=torch.int64))


WORKLOADS = {   # name: (news encoder, user encoder, embedding dim, hidden dim, description)
    "nrms": ("tfm", "tfm", 300, 300, "configs[1]: NRMS-style MHSA news+user encoder (TwoTower-Tfm-Tfm), T=32, L=50, K=4, 300-d, 12 heads"),
    "cnn_rnn": ("cnn", "rnn", 768, 768, "configs[0] shape on the GPU: main.twotower defaults (TwoTower-Cnn-Rnn), 768-d"),
    "bert": ("bert", "attn", 768, 768, "configs[4] shape on one GPU: BERT-base news encoder (12 layers, 768-d) + attention user encoder, fp32"),
}


def build_model(device, workload="nrms"):
    import types
    from models.TwoTower import TwoTowerModel
    from models.modules import encoder as Enc
    ne, ue, E, D, _ = WORKLOADS[workload]
    mgr = types.SimpleNamespace(
        plm_dim=E, hidden_dim=D, head_num=H, his_size=L, sequence_length=T, negative_num=K, vocab_size=V, device=device,
        learning_rate=1e-5, rank=0, world_size=1, verbose=None, enable_fields=["title"], dropout_p=0.1, synthetic_plm=True,
        plm_dir="", distributed=False, debug=False, impr_size=20, batch_size_eval=200, scheduler="none", epochs=1, warmup=0.1)
    torch.manual_seed(3407)
    news = {"tfm": Enc.TfmNewsEncoder, "cnn": Enc.CnnNewsEncoder, "bert": Enc.AllBertNewsEncoder}[ne](mgr)
    user = {"tfm": Enc.TfmUserEncoder, "rnn": Enc.RnnUserEncoder, "attn": Enc.AttnUserEncoder}[ue](mgr)
    return TwoTowerModel(mgr, news, user), mgr


def train_flops_per_impression():
    per_seq = lambda S: 26 * S * D * D + 4 * S * S * D                   # SURVEY §8d (fwd, incl. pooling projection)
    return 3 * ((K + 1 + L) * per_seq(T) + per_seq(L))


def pmc_traffic(family="gemm_family"):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command
    (tools/profile.sh -> tools/summarize_profile.py -> profiles/*_counters.json; counters cannot be read
    from inside the process).  (2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")))
    for f in reversed(files):
        try:
            fam = json.load(open(f)).get(family)
            if fam and "hbm_bytes_per_launch" in fam:
                return {"traffic": fam["hbm_bytes_per_launch"], "traffic_unit": "B per launch (PMC)",
                        "traffic_source": "profiles/" + os.path.basename(f), "mfma_busy_frac_pmc": fam.get("mfma_busy_frac")}
        except (OSError, ValueError):
            continue
    return {"traffic": None}


def hbm_kernel_leg(model, B, batch=None):
    """each HBM-bound kernel of the path alone, timed with events on the launch stream; GB/s = algorithmic bytes
     / time.  `*_executed` = the shape the packed train step really launches (the rows / titles of `batch`
    that survive packing and dead-history elimination: ~22 560 rows of 56 320 at B = 32); the unsuffixed entries are the
    padded shapes of round 1 (B*(1+K+L) titles x T tokens), kept for continuity.  The eval_* entries read the 86 MB
    news-vector cache, which fits the 256 MB Infinity Cache (as it does in production): their GB/s is cache-fed
    bandwidth priced against the 8 TB/s HBM figure, not an HBM measurement."""
    from newsrec_hip import ops
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    M, R = B * (K + 1 + L), B * (K + 1 + L) * T

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = {}

    def add(name, nbytes, fn, note):
        t = timed(fn)
        out[name] = {"GB/s": round(nbytes / t / 1e9, 1), "frac_of_8TBs": round(nbytes / t / 1e9 / PEAK_HBM_GBS, 4),
                     "us": round(t * 1e6, 1), "MB": round(nbytes / 1e6, 1), "shape": note}

    table = model.newsEncoder.embedding.weight.detach()
    ids = torch.randint(1, V, (M, T), generator=g).to(dev)
    add("token_gather", 2.0 * R * E * 4 + 8.0 * R, lambda: ops.gather_rows(table, ids), f"{R} token rows x {E} f32 from the {V}x{E} table")
    x = torch.randn(M, T, D, generator=g).to(dev)
    key = torch.tanh(torch.randn(M, T, D, generator=g)).to(dev)
    q = torch.randn(1, D, generator=g).to(dev)
    msk = torch.ones(M, T, dtype=torch.int64, device=dev)
    add("news_pooling_fwd", M * (2.0 * T * D * 4 + D * 4), lambda: ops.query_pool(key, x, q, msk), f"{M} titles x {T} tokens x {D}")
    h, xx = torch.randn(R, D, generator=g).to(dev), torch.randn(R, D, generator=g).to(dev)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y = torch.empty(R, D, device=dev)
    L_ = ops.lib()
    st = torch.cuda.current_stream().cuda_stream
    add("add_layernorm_fwd", 3.0 * R * D * 4, lambda: ops.check(L_.nr_add_layernorm_fwd_f32(
        h.data_ptr(), xx.data_ptr(), 0, gam.data_ptr(), bet.data_ptr(), R, D, 1e-12, 0.0, 0, 0, y.data_ptr(), None, None, None, st), "ln"),
        f"{R} rows x {D} (read h, x; write y)")
    if batch is not None:              # the shapes the packed step executes
        blk, C_, ids_b, masks_b, _hm, keep_b, pend = model._train_inputs(batch)
        packed = ops.pack_titles(ids_b.reshape(-1, T), masks_b.reshape(-1, T), keep_b, pend)
        if packed is not None:
            ids_p, plan = packed
            Rx, Nt = plan.rows, plan.N
            live = int((plan.offs[1:] > plan.offs[:-1]).sum())
            add("token_gather_executed", 2.0 * Rx * E * 4 + 8.0 * Rx, lambda: ops.gather_rows(table, ids_p),
                f"{Rx} packed token rows x {E} f32 from the {V}x{E} table")
            hx, xxx, yx = h[:Rx], xx[:Rx], y[:Rx]
            add("add_layernorm_fwd_executed", 3.0 * Rx * D * 4, lambda: ops.check(L_.nr_add_layernorm_fwd_f32(
                hx.data_ptr(), xxx.data_ptr(), 0, gam.data_ptr(), bet.data_ptr(), Rx, D, 1e-12, 0.0, 0, 0, yx.data_ptr(), None, None, None, st), "ln"),
                f"{Rx} rows x {D} (read h, x; write y)")
            keyp, valp = key.reshape(-1, D)[:Rx].contiguous(), x.reshape(-1, D)[:Rx].contiguous()
            mk = masks_b.reshape(-1, T)
            outp, probp = torch.empty(Nt, D, device=dev), torch.empty(Nt, T, device=dev)
            add("news_pooling_fwd_executed", 2.0 * Rx * D * 4 + live * D * 4.0, lambda: ops.check(L_.nr_query_pool_packed_fwd_f32(
                keyp.data_ptr(), valp.data_ptr(), plan.offs.data_ptr(), Rx, q.data_ptr(), mk.data_ptr(), Nt, T, D, outp.data_ptr(), probp.data_ptr(), st), "pool"),
                f"{live} live titles of {Nt} ({Rx} packed rows) x {D}")
    # cached evaluation kernels: Re rows of (Ce candidates, L history ids) against the [Nn, D] cache
    Re, Ce = 32768, 20
    cache = torch.randn(NEWS_DEV, D, generator=g).to(dev)
    cdd_idx = torch.randint(0, NEWS_DEV, (Re, Ce), generator=g).to(dev)
    his_idx = torch.randint(0, NEWS_DEV, (Re, L), generator=g).to(dev)
    user = torch.randn(Re, D, generator=g).to(dev)
    add("eval_gather_dot_scoring", Re * ((Ce + 1.0) * D * 4 + Ce * 8 + Ce * 4), lambda: ops.score_from_table(cache, cdd_idx, user),
        f"{Re} rows x {Ce} candidates gathered from the {NEWS_DEV}x{D} cache (86 MB: Infinity-Cache resident)")
    add("eval_history_attn_pool", Re * (2.0 * L * D * 4 + L * 8 + D * 4), lambda: ops.pool_from_table(cache, his_idx, q),
        f"{Re} rows x {L} history vectors gathered (key and value passes)")
    add("eval_history_sum", Re * (1.0 * L * D * 4 + L * 8 + D * 4), lambda: ops.seq_reduce_from_table(cache, his_idx),
        f"{Re} rows x {L} history vectors gathered")
    return out


def cpu_baseline(B, steps=16):
    """the oracle's train step (same shapes, same synthetic batch) on this host's cores"""
    from oracle import twotower_oracle as O
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    model, _ = build_model("cpu")
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    ids, msk = synth_table(g, 20000)
    X = synth_train_batch(g, B, 20000, ids, msk)
    params = list(P.values())
    opt = torch.optim.Adam(params, lr=1e-5)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, _ = O.train_forward(P, "tfm", "tfm", X, head_num=H)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"value": round(B / best, 3), "unit": "train impressions/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed oracle (torch-fp32 CPU restatement) train steps of B={B} (~10-15 s of CPU work) after 1 warm-up, min taken; "
                      "eval-mode dropout"}


_REPORT_FD = None


def quiet_stdout():
    """From here on everything written to fd 1 — Python prints and C-level ones (RCCL's version banner and gloo's
    connection lines go to stdout) — lands on stderr; the original stdout is kept for report(), so that this command's
    stdout is the ONE JSON line and nothing else."""
    global _REPORT_FD
    if _REPORT_FD is None:
        sys.stdout.flush()
        _REPORT_FD = os.dup(1)
        os.dup2(2, 1)


def report(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REPORT_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
        return
    sys.stdout.flush()
    os.write(_REPORT_FD, line)


def launch_ranks(n, argv, extra_env=None, timeout=None):
    """start `n` fresh processes of this file, one per GPU, with the torch.distributed environment (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT = a free port) and wait for them.  The caller must not have touched
    the GPU (nothing here does: no torch.cuda call is made in the parent).  Rank 0 inherits stdout, so its one JSON
    line is this command's output.  The first rank to fail ends the others (exact PIDs).  Returns the exit status."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL needs it)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.time()
    status = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is not None:
                live.remove(p)
                if rc != 0 and status == 0:
                    status = rc
        if status != 0 or (timeout is not None and time.time() - t0 > timeout):
            for p in live:
                p.kill()
            for p in live:
                p.wait()
            return status or 124
        time.sleep(0.05)
    return status


def timed_steps(step, fence, steps, warmup, reduce, reduce_max):
    """the contract's timed region: `warmup` untimed steps, then EXACTLY `steps` steps between two fences (barrier +
    device synchronize on both sides); the time is the MAX over ranks.  Returns (seconds, last step's result)."""
    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = step(i)
    fence()
    dt = time.perf_counter() - t0
    if reduce:
        dt = reduce_max(dt)
    return dt, last


def stub_worker(args):
    """CPU rehearsal of the multi-rank plumbing (tests/test_distributed_cpu.py; NR_BENCH_STUB=1): gloo instead of RCCL
    and a tiny torch model instead of the HIP path, through the same launcher, FlatGradSync step, fences, max-over-ranks
    timing and one-line report.  Never a measurement."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from utils.dp import FlatGradSync, all_reduce_
    torch.manual_seed(rank)
    model = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    net = FlatGradSync(model, world)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = torch.randn(args.batch_size, 16)

    def step(i):
        net.zero_grad()
        loss = net(x).square().mean()
        loss.backward()
        net.sync()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()

    def reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64)
        all_reduce_(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    dt, loss = timed_steps(step, fence, args.steps, args.warmup, world > 1, reduce_max)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    same = True
    if world > 1:
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        same = all(torch.equal(both[0], b) for b in both)
    if rank == 0:
        report({"metric": "stub", "stub": True, "value": world * args.batch_size * args.steps / dt, "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
                "replicas_identical": same, "scaling": "weak"})
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=32, help="impressions per GPU per step (reference -bs default)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="nrms",
                    help="nrms = the headline workload (BASELINE configs[1]); the others are extra measurements "
                         "(no roofline/cpu_baseline/eval legs)")
    ap.add_argument("--gemm", choices=["fp32", "fp32-mfma", "bf16"], default="fp32",
                    help="matrix-core operand precision (nr_set_gemm_compute): bf16 is the 1e-2-parity build of SURVEY §8 a7, "
                         "an extra measurement next to the fp32 headline")
    ap.add_argument("--repeats", type=int, default=4, help="extra timed regions of --steps steps after the headline one (min/median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the B = 64 / 128 information leg (profiling runs)")
    args = ap.parse_args()

    # N > 1 and not already one of N ranks: start the ranks (fresh processes; this parent never touches the GPU)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    quiet_stdout()
    if os.environ.get("NR_BENCH_STUB") == "1":
        return stub_worker(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # rehearsal on a one-GPU box only (never set by the driver): every rank on device 0, gloo instead of RCCL, so the
    # multi-rank logic (DDP wrap, sharded cache + all-reduce, max-over-ranks timing) can be exercised without 2 GPUs
    rehearse = os.environ.get("NR_BENCH_REHEARSE_ONE_GPU") == "1"
    if os.environ.get("NR_BENCH_HANG_DUMP"):    # debugging aid: python stacks of every rank after N seconds
        import faulthandler
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        faulthandler.dump_traceback_later(int(os.environ["NR_BENCH_HANG_DUMP"]), exit=True,
                                          file=open(os.path.join(ROOT, "gpurun_out", f"hang_rank{rank}.txt"), "w"))
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    # NR_BENCH_FORCE_DIST=1 (tests/test_gpu_rccl_world1.py): take the N > 1 code path — RCCL process group, FlatGradSync,
    # barriers, all-gathered cache — with a single rank, which is all a one-GPU box can run of it on RCCL itself
    multi = world > 1 or os.environ.get("NR_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL needs dmabuf IPC on this pool (torchrun passes the env through; this covers a bare launch)
        if world == 1:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1])); sk.close()
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from newsrec_hip import ops
    from utils.dp import all_reduce_
    ops.set_gemm_compute(args.gemm)

    B = args.batch_size
    extra = args.workload != "nrms"
    if extra:
        args.no_eval = args.no_cpu_baseline = True
    model, mgr = build_model(local, args.workload)
    model = model.cuda().train()
    net = model
    if multi:
        from utils.dp import FlatGradSync
        net = FlatGradSync(model, world, always_reduce=True)
    opt, _ = model.get_optimizer(mgr, 1000)

    g = torch.Generator().manual_seed(1000 + rank)
    ids, msk = synth_table(g, NEWS_TRAIN)
    batches = []
    for _ in range(8):                                                    # pre-staged in HBM: the path, not the loader
        X = synth_train_batch(g, B, NEWS_TRAIN, ids, msk)
        batches.append({k: v.cuda() for k, v in X.items()})

    ar_events = []                              # (start, end) events around the gradient all-reduce, roofline leg only

    def step(i, time_sync=False):
        if multi:
            net.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        loss = net(batches[i % len(batches)])
        model.plan_ahead(batches[(i + 1) % len(batches)])     # as Manager._train does: the next batch's pack plan, one step early
        loss.backward()
        if multi:
            if time_sync:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            net.sync()                       # one all-reduce (average) over the flat gradient buffer
            if time_sync:
                e1.record()                  # the current stream waits for the collective, so the pair brackets it
                ar_events.append((e0, e1))
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(v):
        tt = torch.tensor([v] if not isinstance(v, (list, tuple)) else list(v), device="cuda", dtype=torch.float64)
        all_reduce_(tt, op=dist.ReduceOp.MAX)
        out = [float(x) for x in tt.tolist()]
        return out if isinstance(v, (list, tuple)) else out[0]

    # as Manager._train does before its loop: a generation-2 garbage collection over the long-lived objects (model,
    # tables, batches) costs 65-80 ms and would land in a timed region every few hundred steps
    import gc
    gc.collect()
    gc.freeze()
    dt, loss = timed_steps(step, fence, args.steps, args.warmup, multi, reduce_max)
    final_loss = float(loss.detach())
    ms_per_step = 1000.0 * dt / args.steps
    value = world * B * args.steps / dt
    # the same region again, --repeats times (reported next to the headline, never instead of it: VERDICT r1 item 8)
    rep = []
    for _ in range(args.repeats):
        d, _l = timed_steps(step, fence, args.steps, 0, multi, reduce_max)
        rep.append(1000.0 * d / args.steps)

    # ---- roofline leg: same steps with HIP events around every launch of each kernel family.  The timed region above runs
    # the weight-gradient products on a second stream; a kernel's duration only means something when it has the chip to
    # itself, so this leg (and tools/profile.sh, whose rocprof durations it must agree with) runs them one at a time.
    ops.set_side_stream(False)
    ops.prof_enable(True)
    for i in range(args.steps):
        step(i, time_sync=True)
    torch.cuda.synchronize()
    ops.set_side_stream(True)
    allreduce = None
    if multi:
        ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1)
        nbytes = net.flat.numel() * 4
        # each of the two collectives alone (what sync() issues: the dense parameters, started during the backward at the
        # token table's scatter-add, and the table gradient), 10 repetitions between events on rank 0
        pieces = {}
        for name, buf in (("dense_params", net.flat_early), ("token_table", net.flat_late)):
            if buf.numel() == 0:
                continue
            scratch = torch.zeros_like(buf)
            for _ in range(2):
                all_reduce_(scratch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                all_reduce_(scratch)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            pieces[name] = {"bytes": buf.numel() * 4, "ms": round(ms, 4), "algbw_GB/s": round(buf.numel() * 4 / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
        allreduce = {"ms_per_step": round(ar_ms, 4), "frac_of_step": round(ar_ms / ms_per_step, 4), "bytes": nbytes,
                     "algbw_GB/s": round(nbytes / (ar_ms * 1e-3) / 1e9, 1) if ar_ms > 0 else None,
                     "pieces_alone": pieces,
                     "backend": dist.get_backend(),
                     "note": "ms_per_step = events around FlatGradSync.sync on rank 0 in the one-kernel-at-a-time leg: the exposed part "
                             "(table all-reduce + wait for the dense one, which was started at the scatter-add + 1/N scale); "
                             "pieces_alone = each collective by itself"}
    fam = {}
    for name, idx in (("gemm", 0), ("gather", 1), ("attention", 2), ("layernorm", 3), ("pool", 4), ("score", 5), ("misc", 6), ("scatter_add", 7),
                      ("gemm_split", 8), ("hbm_small", 9)):
        ms, n, work = ops.prof_read(idx)
        fam[name] = dict(ms_per_step=round(ms / args.steps, 4), launches_per_step=n / args.steps, work_per_step=work / args.steps)
    ops.prof_enable(False)
    tfl = lambda f: f["work_per_step"] / (f["ms_per_step"] * 1e-3) / 1e12 if f["ms_per_step"] > 0 else 0.0
    gm, gs = fam["gemm"], fam["gemm_split"]
    all_ms = gm["ms_per_step"] + gs["ms_per_step"]
    all_flops = gm["work_per_step"] + gs["work_per_step"]
    all_tflops = all_flops / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0
    fam_line = lambda f: {"achieved": round(tfl(f), 2), "ms_per_step": f["ms_per_step"], "launches_per_step": f["launches_per_step"],
                          "flops_per_step": f["work_per_step"]}
    if gs["ms_per_step"] > gm["ms_per_step"]:
        # default mode: the split-operand kernel carries the step.  It runs on the bf16 matrix pipe and issues six
        # bf16 MFMAs per algorithmic multiply-add, so it is priced against dense bf16 peak / 6 (achieved = ALGORITHMIC
        # flops of the launched products over their measured time; executed = 6x that against the 2500 peak: same frac)
        peak = round(PEAK_BF16_MATRIX_TFLOPS / 6.0, 1)
        roofline = {"kernel": "gemm_x3_kernel (fp32 operands split into 3 bf16 pieces in LDS, 6 x v_mfma_f32_32x32x16_bf16 per k-step, f32 accumulate): "
                              "every launch of it in one train step",
                    "bound": "mfma", "achieved": round(tfl(gs), 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tfl(gs) / peak, 4),
                    "peak_note": "2500 TFLOP/s dense bf16 / 6 piece products per fp32 product; executed bf16 rate = 6 x achieved",
                    "executed_bf16_tflops": round(6 * tfl(gs), 1), "vs_f32_mfma_peak_157.3": round(tfl(gs) / PEAK_F32_MATRIX_TFLOPS, 4),
                    **pmc_traffic("gemm_split_family"),
                    "launches_per_step": gs["launches_per_step"], "avg_launch_ms": round(gs["ms_per_step"] / max(gs["launches_per_step"], 1), 5),
                    "flops_per_step": gs["work_per_step"],
                    "timing_note": "kernel durations measured with the kernels running one at a time (nr_set_side_stream(0)); the timed region of "
                                   "`value` overlaps the weight-gradient products with the rest of the backward on a second stream",
                    "other_gemm_launches": {**fam_line(gm), "kernel": "gemm_x3s_kernel (the same arithmetic on 64x64 tiles, eight waves splitting the k-tile): products with too few "
                                                                      "tiles for gemm_x3_kernel (the user tower's 1 600 rows); latency-bound 125-475-workgroup grids",
                                            "peak": peak, "frac": round(tfl(gm) / peak, 4)},
                    "all_gemm_launches": {"achieved": round(all_tflops, 2), "ms_per_step": round(all_ms, 4), "flops_per_step": all_flops}}
    else:
        bf16 = args.gemm == "bf16"          # mixed launches then: large products on the bf16 pipe (dense peak 2500 TFLOP/s), the rest fp32
        peak = PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS
        roofline = {"kernel": ("gemm_bf16_kernel / gemm_h_kernel (v_mfma_f32_32x32x16_bf16) + gemm_f32_kernel for small/conv products" if bf16 else
                               "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)") + ", all GEMM launches of one train step",
                    "bound": "mfma", "achieved": round(tfl(gm), 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(tfl(gm) / peak, 4), **({"traffic": None} if bf16 else pmc_traffic()),
                    "launches_per_step": gm["launches_per_step"], "avg_launch_ms": round(gm["ms_per_step"] / max(gm["launches_per_step"], 1), 5),
                    "flops_per_step": gm["work_per_step"]}
    # rows of the batch that are padding / unused history and are never computed (packed title tokens)
    try:
        tot = ex = 0
        for x in batches:
            _blk, C_, ids_, masks_, _hm, _keep, pend = model._train_inputs(x)
            tot += ids_.numel()
            if pend is None:
                ex += ids_.numel()
            else:
                pend.event.synchronize()
                ex += int(pend.host[0])
        rows = {"token_rows_per_step": tot / len(batches), "executed_rows_per_step": ex / len(batches),
                "padded_rows_frac": round(1.0 - ex / tot, 4)}
    except Exception as e:                     # noqa: BLE001
        rows = {"error": repr(e)[:200]}
    hbm_step = {}
    for name in ("gather", "layernorm", "pool", "score", "scatter_add"):
        f = fam[name]
        if f["ms_per_step"] > 0:
            gbs = f["work_per_step"] / (f["ms_per_step"] * 1e-3) / 1e9
            hbm_step[name] = {"GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK_HBM_GBS, 4), "ms_per_step": f["ms_per_step"]}
    f = fam["hbm_small"]
    hbm_step["small_launches"] = {"ms_per_step": f["ms_per_step"], "launches_per_step": f["launches_per_step"], "MB_per_step": round(f["work_per_step"] / 1e6, 2),
                                  "note": "launches of the gather / layernorm / pool / score families that move under 16 MB (the 32-sequence user-tower calls, "
                                          "the 32-row scoring): launch-bound, booked apart from the families above"}
    try:
        with torch.no_grad():
            hbm = hbm_kernel_leg(model, B, batches[0]) if not extra else {}
    except Exception as e:                     # noqa: BLE001 — secondary leg, reported in the JSON line
        hbm = {"error": repr(e)[:300]}

    out = {
        "metric": "train impressions/sec (node), NRMS-shape MIND-large; eval impressions/sec",
        "value": round(value, 2), "unit": "impressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32 (large dense products: f32 operands split exactly into 3 bf16 pieces, 6 piece products accumulated in f32 on the bf16 matrix pipe; f32 accuracy, 1e-5 parity)",
                  "fp32-mfma": "f32", "bf16": "bf16 operands, f32 accumulate (tensors f32)"}[args.gemm], "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload][4] + ", train step = fwd+bwd+Adam with dropout 0.1",
                   "per_gpu_batch": B, "global_batch": B * world, "vocab": V, "parallelism": f"dp{world}"},
        "loss": round(final_loss, 5),
        "repeats_ms_per_step": {"n": len(rep), "min": round(min(rep), 4), "median": round(sorted(rep)[len(rep) // 2], 4),
                                "max": round(max(rep), 4)} if rep else None,
        "allreduce": allreduce,
        "executed_tflops": round((all_flops + fam["attention"]["work_per_step"]) / (ms_per_step * 1e-3) / 1e12, 2),
        "padded_model_tflops": round(value * train_flops_per_impression() / 1e12, 2) if not extra else None,
        "rows": rows,
        "roofline": roofline, "hbm_kernels": hbm, "hbm_families_in_step": hbm_step, "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in fam.items()},
    }

    # ---- larger per-GPU batches, for information only (the headline stays at the reference's default -bs 32): the same step
    # at B = 64 / 128 shows how much of the B = 32 step is fixed cost (the 1 600-row user tower, launch-bound small kernels)
    if not extra and not multi and not args.no_sweep:
        try:
            sweep = {}
            for Bx in (64, 128):
                gx = torch.Generator().manual_seed(77 + Bx)
                bx = [{k: v.cuda() for k, v in synth_train_batch(gx, Bx, NEWS_TRAIN, ids, msk).items()} for _ in range(4)]

                def stepx(i):
                    opt.zero_grad(set_to_none=True)
                    loss = net(bx[i % 4])
                    model.plan_ahead(bx[(i + 1) % 4])
                    loss.backward()
                    opt.step()
                    return loss
                d, _l = timed_steps(stepx, fence, 20, 3, False, reduce_max)
                sweep[str(Bx)] = {"impressions_per_s": round(Bx * 20 / d, 1), "ms_per_step": round(1000.0 * d / 20, 4)}
                del bx
            out["batch_sweep"] = sweep
        except Exception as e:                 # noqa: BLE001 — secondary leg
            out["batch_sweep"] = {"error": repr(e)[:200]}

    # ---- cached evaluation leg (BASELINE config 4 shape, per GPU): cache build + impression scoring
    if not args.no_eval:
        try:                               # a secondary leg: its failure must not take the headline line with it
            model.eval()
            import numpy as np
            import types
            gd = torch.Generator().manual_seed(7)
            dids, dmsk = synth_table(gd, NEWS_DEV)
            # the product's own cache build: TwoTowerBaseModel.encode_news shards the corpus rows over the ranks, encodes
            # its slice and all-gathers the [Nn, D] table (dist.all_gather_into_tensor on RCCL); the token table is
            # attached in HBM first (inputs resident before the timed region)
            news_ds = types.SimpleNamespace(token_ids=dids.numpy(), attn_masks=dmsk.numpy())
            loader_news = types.SimpleNamespace(dataset=type("NewsRows", (), {"__len__": lambda self: NEWS_DEV, "token_ids": news_ds.token_ids,
                                                                             "attn_masks": news_ds.attn_masks})(), batch_size=200)
            mgr.distributed, mgr.world_size, mgr.rank = multi, world, rank
            model.world_size, model.rank = world, rank
            model.attach_news_corpus(dids, dmsk)
            with torch.no_grad():
                model.encode_news(mgr, loader_news)
                fence()
                t0 = time.perf_counter()
                model.encode_news(mgr, loader_news)
                fence()
                t_enc = time.perf_counter() - t0
                Re, Ce = 200, 20
                rows = []
                for _ in range(8):
                    x = {"cdd_idx": torch.randint(0, NEWS_DEV, (Re, Ce), generator=gd).cuda(),
                         "his_idx": torch.randint(0, NEWS_DEV, (Re, L), generator=gd).cuda(),
                         "his_mask": torch.ones(Re, L, dtype=torch.int64).cuda()}
                    rows.append(x)
                for i in range(3):
                    model.infer(rows[i])
                fence()
                n_it = 50
                t0 = time.perf_counter()
                for i in range(n_it):
                    model.infer(rows[i % 8])
                fence()
                t_inf = time.perf_counter() - t0
                # whole impressions end to end (BASELINE config 4 shape, this rank's shard): score every <=20-candidate
                # chunk with large batches, then per-impression AUC/MRR/nDCG@5/10 on the device
                from utils.util import device_compute_metrics
                n_impr = 40000
                ncand = torch.randint(2, 81, (n_impr,), generator=gd)                   # MIND-like: ~40 candidates / impression
                nrow = (ncand + Ce - 1) // Ce
                impr_of_row = torch.repeat_interleave(torch.arange(n_impr), nrow)
                first = torch.cumsum(nrow, 0) - nrow
                within = torch.arange(int(nrow.sum())) - first[impr_of_row]
                valid = (within[:, None] * Ce + torch.arange(Ce)[None, :]) < ncand[impr_of_row][:, None]
                Rtot = impr_of_row.numel()
                # an impression never lists a news twice (duplicates score exact ties, which the metrics re-rank on the host
                # with numpy's tie order: 4 % of random 40-candidate lists would take that path, none of MIND's): candidate j
                # of impression i is news 1 + (s_i + 7919 j) mod (Nn - 1), distinct for j < 80
                s_i = torch.randint(0, NEWS_DEV - 1, (n_impr,), generator=gd)[impr_of_row]
                jj = within[:, None] * Ce + torch.arange(Ce)[None, :]
                cdd_all = ((1 + (s_i[:, None] + 7919 * jj) % (NEWS_DEV - 1)) * valid).cuda()
                his_u = torch.randint(1, NEWS_DEV, (n_impr, L), generator=gd)
                # history lengths as in the train batches (uniform 0..70 truncated to L; empty -> slot 0 kept, news 0)
                hl = torch.randint(0, 71, (n_impr,), generator=gd).clamp(max=L)
                live = torch.arange(L)[None, :] < hl[:, None]
                hm_u = live.long()
                hm_u[hl == 0, 0] = 1
                his_all = his_u[impr_of_row].cuda()
                his_mind = (his_u * live)[impr_of_row].cuda()
                hm_mind = hm_u[impr_of_row].cuda()
                lab_all = torch.where(valid, (torch.rand(Rtot, Ce, generator=gd) < 0.1).long(), torch.tensor(-1))
                lab_all[first, 0], lab_all[first, 1] = 1, 0                               # both classes in every impression
                lab_all, valid_d, impr_d = lab_all.cuda(), valid.cuda(), impr_of_row.cuda()
                hm = torch.ones(8192, L, dtype=torch.int64, device="cuda")

                # this rank's impressions as a list of 8192-row batches = the "loader" of the product's own _predict, which
                # scores them (infer), keeps the results on the device and, for N > 1, gathers them to rank 0 as flat typed
                # tensors; rank 0 then computes the per-impression metrics on the device.  Impression ids are made
                # distinct across ranks (rank * n_impr + i) as the reference's shards are.
                impr_g = impr_d + rank * n_impr
                def loader_of(his, mask_of):
                    return [{"cdd_idx": cdd_all[s:s + 8192], "his_idx": his[s:s + 8192], "his_mask": mask_of(s),
                             "impr_index": impr_g[s:s + 8192], "cdd_mask": valid_d[s:s + 8192], "label": lab_all[s:s + 8192]}
                            for s in range(0, Rtot, 8192)]
                loader_full = loader_of(his_all, lambda s: hm[:min(8192, Rtot - s)])          # every one of the 50 slots clicked
                loader_mind = loader_of(his_mind, lambda s: hm_mind[s:s + 8192])              # MIND-like history lengths
                mgr.impr_size = Ce

                def full_eval(loader):
                    res = model._predict(mgr, loader, with_labels=True, to_host=False)
                    if res is None:
                        return None
                    impr, masks, preds, labels = res
                    return device_compute_metrics(impr, masks, preds, labels, ["auc", "mean_mrr", "ndcg@5;10"])

                def timed_eval(loader):
                    full_eval(loader)
                    fence()
                    t0 = time.perf_counter()
                    m = full_eval(loader)
                    fence()
                    return time.perf_counter() - t0, m
                t_full, met = timed_eval(loader_full)
                t_mind, _ = timed_eval(loader_mind)
            if multi:
                t_enc, t_inf, t_full, t_mind = reduce_max([t_enc, t_inf, t_full, t_mind])
            out["eval"] = {"news_cache_rows": NEWS_DEV, "cache_build_news_per_s": round(NEWS_DEV / t_enc, 1),
                           "infer_rows_per_s_batch200": round(world * n_it * Re / t_inf, 1),
                           "eval_impressions_per_s": round(world * n_impr / t_full, 1),
                           "eval_rows_per_s": round(world * Rtot / t_full, 1),
                           "eval_impressions_per_s_mind_like_histories": round(world * n_impr / t_mind, 1),
                           "note": "cache build = TwoTowerBaseModel.encode_news over the 72 024-news corpus (sharded over the ranks + "
                                   "all-gather, tfm encoder); batch200 = the reference's -bse default; eval_impressions_per_s = 40 000 "
                                   "impressions per GPU (2..80 DISTINCT candidates each, L=50, tfm user encoder, ALL 50 history slots clicked) through "
                                   "_predict (8192-row batches, typed gather to rank 0) + device AUC/MRR/nDCG@5/10 on rank 0, end to end; "
                                   "..._mind_like_histories = the same with the train batches' history lengths (uniform 0..70 truncated "
                                   "to 50, SURVEY 8d): the user tower then runs over the clicked slots only (TfmUserEncoder.from_cache)",
                           "metrics_of_random_scores": met}

        except Exception as e:                 # noqa: BLE001 — reported in the JSON line
            out["eval"] = {"error": repr(e)[:300]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(B)
        except Exception as e:                 # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)[:300]}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        report(out)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
