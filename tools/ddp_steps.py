"""Multi-step 2-rank data-parallel rehearsal on ONE GPU (both ranks on cuda:0, gloo staged through host copies): four
optimisation steps in train mode through the product's FlatGradSync (utils/dp.py); at the end the replicas must be
bit-identical.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/ddp_steps.py
(`NR_BENCH_REHEARSE_ONE_GPU=1 python bench.py --gpus 2` rehearses the whole bench the same way.)
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
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    faulthandler.dump_traceback_later(40, exit=True, file=open(os.path.join(ROOT, "gpurun_out", f"ddp_steps_flat_rank{rank}.txt"), "w"))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from utils.dp import FlatGradSync
    model, mgr = bench.build_model(0)
    model = model.cuda().train()
    g = torch.Generator().manual_seed(5 + rank)
    ids, msk = bench.synth_table(g, 5000)
    X = {k: v.cuda() for k, v in bench.synth_train_batch(g, 8, 5000, ids, msk).items()}
    net = FlatGradSync(model, world)
    opt, _ = model.get_optimizer(mgr, 100)
    for it in range(4):
        net.zero_grad()
        loss = net(X)
        loss.backward()
        net.sync()
        opt.step()
        torch.cuda.synchronize()
        print(f"[flat] rank {rank} step {it} loss {loss.item():.5f}", flush=True)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    if rank == 0:
        print("[flat] ddp_steps ok: 4 steps, replicas bit-identical", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
