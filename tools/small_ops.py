"""which Python lines issue the small torch kernels of a train step (fills, cats, elementwise): torch.profiler with stacks.
usage: python tools/small_ops.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.set_device(0)
model, mgr = bench.build_model(0, "nrms")
model = model.cuda().train()
opt, _ = model.get_optimizer(mgr, 1000)
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
batches = [{k: v.cuda() for k, v in bench.synth_train_batch(g, 32, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(8)]

def step(i):
    opt.zero_grad(set_to_none=True)
    loss = model(batches[i % 8])
    model.plan_ahead(batches[(i + 1) % 8])
    loss.backward()
    opt.step()

for i in range(6):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=8):
    if e.key.startswith("aten::") and e.device_time_total > 0 and e.key not in ("aten::to", "aten::_to_copy"):
        stack = [s for s in e.stack if "news-recommendation_amd" in s or "bench.py" in s or "small_ops" in s][:3]
        rows.append((e.device_time_total / steps, e.count / steps, e.key, " <- ".join(s.split("news-recommendation_amd/")[-1] for s in stack)))
rows.sort(reverse=True)
for us, cnt, key, st in rows[:40]:
    print(f"{us:8.1f} us/step {cnt:5.1f}/step {key:28s} {st}")
