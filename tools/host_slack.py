"""is the train step host-bound?  adds a busy-wait of D us to every step and watches the wall time: a GPU-bound loop absorbs
it, a host-bound loop slows down by D.   usage: python tools/host_slack.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

torch.cuda.set_device(0)
model, mgr = bench.build_model(0, "nrms")
model = model.cuda().train()
opt, _ = model.get_optimizer(mgr, 1000)
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
batches = [{k: v.cuda() for k, v in bench.synth_train_batch(g, 32, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(8)]

import gc
_gc_t = {}
def _gc_cb(phase, info):
    if phase == "start":
        _gc_t["t"] = time.perf_counter()
    else:
        dt = time.perf_counter() - _gc_t["t"]
        if dt > 2e-3:
            print(f"    gc gen{info['generation']} took {dt * 1e3:.1f} ms (collected {info['collected']})")
gc.callbacks.append(_gc_cb)
if os.environ.get("NR_GC_FREEZE") == "1":
    gc.collect(); gc.freeze()


def step(i, delay):
    opt.zero_grad(set_to_none=True)
    loss = model(batches[i % 8])
    model.plan_ahead(batches[(i + 1) % 8])
    loss.backward()
    opt.step()
    if delay:
        t = time.perf_counter() + delay * 1e-6
        while time.perf_counter() < t:
            pass

for delay in [int(v) for v in (sys.argv[1:] or ["0", "100", "200", "400", "800"])]:
    for i in range(10):
        step(i, delay)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        step(i, delay)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"delay {delay:4d} us: host loop {1e3 * (t1 - t0) / 100:.3f} ms/step, wall {1e3 * (t2 - t0) / 100:.3f} ms/step")
