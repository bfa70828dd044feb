"""world_size-2 tests of the N>1 path on CPU (gloo): sharded news-cache build + all-gather, sharded
impression scoring + typed gather + metrics.  The HIP kernels cannot run here, so the two compute
hooks (`_encode_news`, `infer`) are replaced by the CPU oracle: what is under test is the product's
sharding / collective / regrouping logic, which is device independent."""
import os
import socket
import sys
import tempfile
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(tmp, world, rank):
    from utils.manager import Manager
    from utils.synthetic import write_synthetic_caches
    from models.TwoTower import TwoTowerModel
    from models.modules import encoder as E
    from oracle import twotower_oracle as O
    m = Manager({"enable_fields": ["title"]}, command=f"-d cpu -s demo -m dev -ne cnn -ue attn -hd 20 -tl 16 -hs 12 -is 6 -bse 7 -ws {world} "
                                                      f"--synthetic-plm -cr {tmp}/cache -dr {tmp}/data".split())
    m.vocab_size, m.plm_dim = 5000, 24
    m.news_nums = {"MINDdemo_train": 300, "MINDdemo_dev": 200}
    if rank == 0:
        write_synthetic_caches(m, n_train_impr=8, n_dev_impr=33, seed=3, splits=("dev",))
    return m, TwoTowerModel, E, O


def _hook_oracle(model, O):
    P = {k: v.detach() for k, v in model.state_dict().items()}

    def enc(x, cdd=True):
        key = "cdd" if cdd else "his"
        return O.cnn_news_encoder(P, x[key + "_token_id"], x[key + "_attn_mask"])

    def infer(x):
        return O.infer(P, "attn", model.news_embeddings, x["cdd_idx"], x["his_idx"], x["his_mask"])
    model._encode_news = enc
    model._corpus_vectors = lambda x: enc(x)[1]
    model.infer = infer


def _worker(rank, world, tmp, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    for p in (helpers.__file__,):
        pass
    m, TwoTowerModel, E, O = _build(tmp, world, rank)
    m.setup(rank)
    m.barrier()
    torch.manual_seed(3407)
    model = TwoTowerModel(m, E.CnnNewsEncoder(m), E.AttnUserEncoder(m))
    _hook_oracle(model, O)
    loaders = m.prepare()
    metrics = model.dev(m, loaders)
    cache = model.news_embeddings.clone()
    shard = (loaders["dev"].sampler.start, loaders["dev"].sampler.end)
    q.put((rank, metrics, cache.numpy(), shard))
    m.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_eval_world2_equals_single_process():
    tmp = tempfile.mkdtemp(prefix="nr_dist_")
    # single process reference run
    m, TwoTowerModel, E, O = _build(tmp, 1, 0)
    m.setup(0)
    torch.manual_seed(3407)
    model = TwoTowerModel(m, E.CnnNewsEncoder(m), E.AttnUserEncoder(m))
    _hook_oracle(model, O)
    loaders = m.prepare()
    single = model.dev(m, loaders)
    cache1 = model.news_embeddings.numpy()
    assert cache1.shape == (201, 20) and set(single) >= {"auc", "mean_mrr", "ndcg@5", "ndcg@10", "main"}

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, tmp, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, met0, c0, s0), (_, met1, c1, s1) = got
    assert met1 is None and met0 == single                      # rank 0 reports, identical to 1 process
    assert np.array_equal(c0, cache1) and np.array_equal(c1, cache1)   # every rank holds the whole cache
    n = len(loaders["dev"].dataset)
    assert s0 == (0, round(n / 2)) and s1 == (round(n / 2), n)  # contiguous impression shards


def _grad_sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from utils.dp import FlatGradSync
    torch.manual_seed(100 + rank)                       # replicas start DIFFERENT: the wrapper must broadcast rank 0's weights
    class Net(torch.nn.Module):              # `embedding.weight` = a token table: the LATE piece of the flat buffer (utils/dp.py)
        def __init__(self):
            super().__init__()
            self.embedding = torch.nn.Embedding(50, 8)
            self.body = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1))

        def forward(self, ids):
            return self.body(self.embedding(ids))
    net = Net()
    dp = FlatGradSync(net, world)
    assert 0 < dp.n_early < dp.flat.numel() and dp.flat_late.numel() == 50 * 8 and dp.params[-1] is net.embedding.weight
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, 50, (8, 5), generator=g)     # the full batch; rank r takes rows r::world
    full = Net()
    full.load_state_dict(net.state_dict())
    full_opt = torch.optim.Adam(full.parameters(), lr=1e-2)
    for _ in range(3):
        dp.zero_grad()
        dp(ids[rank::world]).square().mean().backward()
        dp.sync()
        opt.step()
        full_opt.zero_grad()
        full(ids).square().mean().backward()            # mean over the whole batch = mean of the per-rank means
        full_opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ref = torch.cat([p.detach().reshape(-1) for p in full.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    q.put((rank, bool(torch.equal(both[0], both[1])), float((flat - ref).abs().max()), all(p.grad is v for p, v in zip(dp.params, dp.views))))
    dist.destroy_process_group()


def test_flat_grad_sync_two_ranks():
    """utils/dp.py (what replaces DistributedDataParallel): rank 0's weights are broadcast, every .grad stays a view of
    the flat buffer, one all-reduce per step averages it, and three Adam steps on two half batches equal three steps
    on the whole batch on one replica."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, same, err, views in res:
        assert same, "replicas diverged"
        assert err < 1e-6, err
        assert views


@pytest.mark.timeout(300)
def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start 2 ranks itself (fresh processes,
    torch.distributed environment on 127.0.0.1) and rank 0 must print ONE JSON line with n_gpus = 2.  NR_BENCH_STUB=1
    swaps the HIP step for a tiny torch model on gloo; the launcher, FlatGradSync step, fences, max-over-ranks timing
    and report are bench.py's own."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["NR_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch-size", "4"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                 # stdout is the JSON line alone (gloo's connection lines are on stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["stub"] is True
    assert out["replicas_identical"] and out["value"] > 0 and out["ms_per_step"] > 0
    # --gpus 1 stays in-process (no launcher, no process group)
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch-size", "4"],
                        env=env, capture_output=True, text=True, timeout=240)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_bench_launcher_propagates_a_rank_failure():
    """a rank that dies ends the others and the launcher returns its status (no hang on the survivors' barrier)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(2, ['--definitely-not-a-flag'], timeout=120))" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200)
    assert r.returncode != 0
