"""2-rank data-parallel check on ONE GPU (both ranks on cuda:0, gloo collectives so no second device is
needed): DistributedDataParallel over the HIP autograd functions must give, on every rank, the gradient of
the full batch (= mean of the two half-batch gradients), and HipAdam must then move both replicas identically.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_check.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "news-recommendation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from helpers import relerr
    model, mgr = bench.build_model(0)
    model = model.cuda().eval()                      # dropout off: the two runs must be comparable
    g = torch.Generator().manual_seed(5)
    ids, msk = bench.synth_table(g, 5000)
    B = 8
    X = {k: v.cuda() for k, v in bench.synth_train_batch(g, B, 5000, ids, msk).items()}
    # reference: the whole batch on one replica
    model.zero_grad()
    model(X).backward()
    ref = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad()
    # data parallel: each rank half the batch
    half = B // world
    Xr = {k: v[rank * half:(rank + 1) * half].contiguous() for k, v in X.items()}
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, gradient_as_bucket_view=True)
    opt, _ = model.get_optimizer(mgr, 100)
    opt.zero_grad(set_to_none=True)
    ddp(Xr).backward()
    worst = 0.0
    gscale = max(float(r.abs().max()) for r in ref.values())
    for n, p in model.named_parameters():
        # analytically-zero gradients (key.bias: softmax shift invariance) are rounding noise: absolute floor
        err = (p.grad - ref[n]).abs().max().item()
        assert err <= 5e-5 * ref[n].abs().max().item() + 3e-6 * gscale, (n, err)
        if ref[n].abs().max().item() > 1e-3 * gscale:
            worst = max(worst, relerr(p.grad, ref[n]))
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged after one Adam step"
    if rank == 0:
        print(f"ddp_check ok: worst gradient rel. error vs full batch {worst:.2e}; replicas bit-identical after Adam")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
