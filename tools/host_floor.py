"""host cost of one train step: the same step at a batch so small that the GPU is never the limit (B = 2: the same
~100 launches, autograd nodes and allocations, a few microseconds of kernel time each), timed per phase on the host.
usage: python tools/host_floor.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.cuda.set_device(0)
model, mgr = bench.build_model(0, "nrms")
model = model.cuda().train()
opt, _ = model.get_optimizer(mgr, 1000)
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
batches = [{k: v.cuda() for k, v in bench.synth_train_batch(g, B, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(8)]
import gc
gc.collect(); gc.freeze()
acc = [0.0] * 5
def step(i, rec):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    loss = model(batches[i % 8])
    t2 = time.perf_counter()
    model.plan_ahead(batches[(i + 1) % 8])
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if rec:
        for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
            acc[k] += b - a
for i in range(20):
    step(i, False)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for i in range(N):
    step(i, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host loop {1e3 * (t1 - t0) / N:.3f} ms/step, wall {1e3 * (t2 - t0) / N:.3f} ms/step")
for name, v in zip(("zero_grad", "forward", "plan_ahead", "backward", "adam"), acc):
    print(f"  {name:10s} {1e3 * v / N:.3f} ms")
