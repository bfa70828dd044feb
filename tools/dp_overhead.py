"""what the data-parallel bookkeeping costs beside the collective: the train step with and without FlatGradSync at
world size 1 (drop gradients -> backward -> one multi-tensor copy into the flat buffer -> Adam on its views).
usage: python tools/dp_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from utils.dp import FlatGradSync

torch.cuda.set_device(0)
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
batches = [{k: v.cuda() for k, v in bench.synth_train_batch(g, 32, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(8)]
import gc
for wrap in (False, True):
    model, mgr = bench.build_model(0, "nrms")
    model = model.cuda().train()
    net = FlatGradSync(model, 1) if wrap else model
    opt, _ = model.get_optimizer(mgr, 1000)

    def step(i):
        if wrap:
            net.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        loss = net(batches[i % 8])
        model.plan_ahead(batches[(i + 1) % 8])
        loss.backward()
        if wrap:
            net.sync()
        opt.step()
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    for i in range(100):
        step(i)
    torch.cuda.synchronize()
    print(("FlatGradSync(world=1)" if wrap else "plain") + f": {(time.perf_counter() - t0) * 10:.3f} ms/step")
