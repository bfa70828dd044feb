"""Multi-step 2-rank data-parallel rehearsal on ONE GPU (both ranks on cuda:0, gloo): several optimisation steps in
train mode, like bench.py's loop; at the end the replicas must be bit-identical.
    VARIANT=flat python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/ddp_steps.py
VARIANT=flat is the product path (utils/dp.py FlatGradSync, gloo collectives staged through host copies).  The other
variants (base | noview | keepgrad | eval | sameseed) wrap the model in torch's DistributedDataParallel over gloo's
device-tensor path, which hung at random in this two-processes-one-GPU setup (kept for the record).
"""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "news-recommendation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    variant = os.environ.get("VARIANT", "base")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    faulthandler.dump_traceback_later(40, exit=True, file=open(os.path.join(ROOT, "gpurun_out", f"ddp_steps_{variant}_rank{rank}.txt"), "w"))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    model, mgr = bench.build_model(0)
    model = model.cuda().train()
    if variant == "eval":
        model.eval()
    g = torch.Generator().manual_seed(5 if variant == "sameseed" else 5 + rank)
    ids, msk = bench.synth_table(g, 5000)
    X = {k: v.cuda() for k, v in bench.synth_train_batch(g, 8, 5000, ids, msk).items()}
    if variant == "flat":
        from utils.dp import FlatGradSync
        ddp = FlatGradSync(model, world)
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0,
                                                        gradient_as_bucket_view=(variant != "noview"))
    opt, _ = model.get_optimizer(mgr, 100)
    for it in range(4):
        if variant == "flat":
            ddp.zero_grad()
        else:
            opt.zero_grad(set_to_none=(variant != "keepgrad"))
        loss = ddp(X)
        loss.backward()
        if variant == "flat":
            torch.cuda.synchronize()
            print(f"[{variant}] rank {rank} step {it} backward drained", flush=True)
            ddp.sync()
            print(f"[{variant}] rank {rank} step {it} all-reduce issued", flush=True)
            torch.cuda.synchronize()
            print(f"[{variant}] rank {rank} step {it} all-reduce done", flush=True)
        opt.step()
        torch.cuda.synchronize()
        print(f"[{variant}] rank {rank} step {it} loss {loss.item():.5f}", flush=True)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    both = [t.cpu() for t in both]
    dist.all_gather(both, flat.cpu())
    assert torch.equal(both[0], both[1]), "replicas diverged"
    if rank == 0:
        print(f"[{variant}] ddp_steps ok: 4 steps, replicas bit-identical", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
