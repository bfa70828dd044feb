"""Host-side cost of one train step (GPU box): cProfile over the bench's own step at the headline workload.
usage: python tools/host_profile.py [steps]    -> top functions by cumulative and by own time"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.cuda.set_device(0)
model, mgr = bench.build_model(0, "nrms")
model = model.cuda().train()
opt, _ = model.get_optimizer(mgr, 1000)
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
batches = [{k: v.cuda() for k, v in bench.synth_train_batch(g, B, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(8)]

def step(i):
    opt.zero_grad(set_to_none=True)
    loss = model(batches[i % 8])
    model.plan_ahead(batches[(i + 1) % 8])            # as bench.py / Manager._train: no device->host wait in the step
    loss.backward()
    opt.step()

for i in range(10):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / steps:.3f} ms/step, wall {1e3 * (t2 - t0) / steps:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).print_callers("_dict_flatten|tree_flatten|tree_map")
print(s.getvalue()[:3000])
