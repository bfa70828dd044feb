"""repeat the train step many times with the second stream on and compare every step's loss / a gradient checksum with the
same steps run with it off (same seeds): any race shows up as a mismatch.   usage: python tools/side_stream_stress.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from newsrec_hip import ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(1000)
ids, msk = bench.synth_table(g, bench.NEWS_TRAIN)
batches = [{k: v.cuda() for k, v in bench.synth_train_batch(g, 32, bench.NEWS_TRAIN, ids, msk).items()} for _ in range(8)]

def run(on):
    ops.set_side_stream(on)
    torch.manual_seed(5)
    ops.dropout_stream.reset(0)
    model, mgr = bench.build_model(0, "nrms")
    model = model.cuda().train()
    opt, _ = model.get_optimizer(mgr, 1000)
    out = []
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = model(batches[i % 8])
        model.plan_ahead(batches[(i + 1) % 8])
        loss.backward()
        chk = torch.stack([p.grad.double().abs().sum() for p in model.parameters() if p.grad is not None and p.numel() < 2_000_000]).sum()
        opt.step()
        out.append((loss.detach().clone(), chk))
    torch.cuda.synchronize()
    return [(float(a), float(b)) for a, b in out]

a = run(False)
a2 = run(False)
b = run(True)
# the embedding scatter-add uses atomics, so two runs agree to rounding, not to the bit, after the first update: the
# baseline is off-vs-off; on-vs-off must look the same (and step 0, before any update, must agree exactly)
rel = lambda x, y: max(abs(p - q) / max(abs(p), 1e-30) for (p, _), (q, _) in zip(x, y))
relg = lambda x, y: max(abs(p - q) / max(abs(p), 1e-30) for (_, p), (_, q) in zip(x, y))
print(f"steps {steps}: off/off  loss rel {rel(a, a2):.2e}  grad-checksum rel {relg(a, a2):.2e}")
print(f"           on/off   loss rel {rel(a, b):.2e}  grad-checksum rel {relg(a, b):.2e}   step 0 identical: {a[0] == b[0]}")
ok = a[0] == b[0] and rel(a, b) < max(10 * rel(a, a2), 1e-5) and relg(a, b) < max(10 * relg(a, a2), 1e-5)
sys.exit(0 if ok else 1)
