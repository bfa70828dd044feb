"""soak: N train steps of the headline workload; device memory (allocated / reserved / peak) and loss at checkpoints —
a leak or a drifting allocation pattern shows as growth after the first few hundred steps.   usage: python tools/soak.py [steps] [dp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dp = len(sys.argv) > 2 and sys.argv[2] == "dp"          # through utils.dp.FlatGradSync (one rank, no process group: its
                                                         # gradient buffer, in-place table gradient and packing, no collectives)
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
batches = [{k: v.cuda() for k, v in bench.synth_train_batch(g, 32, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(16)]
model, mgr = bench.build_model(0, "nrms")
model = model.cuda().train()
opt, _ = model.get_optimizer(mgr, 1000)
net = model
if dp:
    from utils.dp import FlatGradSync
    net = FlatGradSync(model, world_size=1)
marks = sorted({100, 300, 1000, steps // 2, steps})
base = None
for i in range(1, steps + 1):
    if dp:
        net.zero_grad()
    else:
        opt.zero_grad(set_to_none=True)
    loss = net(batches[i % 16])
    model.plan_ahead(batches[(i + 1) % 16])
    loss.backward()
    if dp:
        net.sync()
    opt.step()
    if i in marks:
        torch.cuda.synchronize()
        a, r, p = torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, torch.cuda.max_memory_allocated() >> 20
        print(f"step {i:6d}  loss {float(loss.detach()):.5f}  allocated {a} MiB  reserved {r} MiB  peak {p} MiB", flush=True)
        if i == 300:
            base = (a, r)
        if base is not None and i > 300:
            assert a <= base[0] + 64 and r <= base[1] + 256, "device memory keeps growing"
        assert float(loss.detach()) == float(loss.detach()), "loss is NaN"
print("ok")
