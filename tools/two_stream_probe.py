"""how much would micro-batching the towers over two streams buy?  forward+backward of one B=32 batch on one stream against
two B=16 halves running concurrently on two streams (same model, gradients accumulate; no optimizer step).
usage: python tools/two_stream_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

torch.cuda.set_device(0)
model, mgr = bench.build_model(0, "nrms")
model = model.cuda().train()
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
full = [{k: v.cuda() for k, v in bench.synth_train_batch(g, 32, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(4)]
halves = [({k: v[:16].contiguous() for k, v in b.items()}, {k: v[16:].contiguous() for k, v in b.items()}) for b in full]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one(i):
    model.zero_grad(set_to_none=True)
    model(full[i % 4]).backward()

def two(i):
    model.zero_grad(set_to_none=True)
    a, b = halves[i % 4]
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        la = model(a)
    with torch.cuda.stream(s2):
        lb = model(b)
    with torch.cuda.stream(s1):
        la.backward()
    with torch.cuda.stream(s2):
        lb.backward()
    cur.wait_stream(s1); cur.wait_stream(s2)

def seq_halves(i):
    model.zero_grad(set_to_none=True)
    a, b = halves[i % 4]
    model(a).backward()
    model(b).backward()

for name, fn in (("one stream, B=32", one), ("two halves, one stream", seq_halves), ("two halves, two streams", two)):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        fn(i)
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per fwd+bwd")
