"""Data-parallel gradient averaging for the two-tower model (one process per GPU, RCCL over xGMI).

The reference wraps the model in DistributedDataParallel (main/twotower.py:40-41).  Here the step is: drop the
gradients -> backward -> ONE multi-tensor copy of the fresh gradients into a flat fp32 buffer -> one all-reduce
(average) over the buffer -> Adam on views of the buffer.  No autograd hooks and no communication thread beside the
backward.  The design reason: of the 46 MB reduced per step at 300-d, 36.6 MB are the dense token-table gradient,
which is the LAST thing the backward produces, so DDP's bucket overlap had ~10 MB to hide, and one large collective
uses the xGMI links better than several 25 MB buckets.

What is known about DistributedDataParallel with this library (round 1 records, gpurun_out/ddp_steps_*): wrapped in
torch's DDP over GLOO with TWO PROCESSES ON ONE GPU, both ranks stalled in the first backward (main thread in
_engine_run_backward, the autograd engine thread inside C++ with no Python frame); a stock torch model passed in the
same set-up (ddp_plain.log).  The cause was not established and that configuration (gloo, a shared GPU, the
hook-driven reducer) is not a target: FlatGradSync is the product's data-parallel path by design, not a diagnosis of
that stall.  DDP over RCCL with one process per GPU has not been run with this library.
"""
import torch
import torch.distributed as dist


def _staged(t):
    """under gloo (CPU tests, the one-GPU rehearsal of bench.py) collectives on device tensors go through an explicit
    host copy, so the rehearsal exercises only gloo's host path.  RCCL ("nccl") is untouched."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_(t, op=dist.ReduceOp.SUM):
    if _staged(t):
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def broadcast_(t, src=0):
    if _staged(t):
        c = t.cpu()
        dist.broadcast(c, src)
        t.copy_(c)
    else:
        dist.broadcast(t, src)
    return t


def all_gather_into_tensor_(out, t):
    """out [world * n, ...] <- every rank's t [n, ...] (rank-major), the collective of the sharded news-cache build"""
    if _staged(t):
        co = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(co, t.cpu())
        out.copy_(co)
    else:
        dist.all_gather_into_tensor(out, t)
    return out


def all_gather_list(t, world):
    """list of every rank's tensor (all of t's shape)"""
    if _staged(t):
        bufs = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(bufs, t.cpu())
        return [b.to(t.device) for b in bufs]
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)
    return bufs


class FlatGradSync(torch.nn.Module):
    """wraps the model like DDP does (`.module`, same forward); call zero_grad() before and sync() after backward."""

    def __init__(self, module, world_size=None):
        super().__init__()
        self.module = module
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.world > 1:                       # replicas start from rank 0's weights (and buffers), like DDP
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    broadcast_(t.data, 0)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def zero_grad(self, set_to_none=False):
        """drop the gradients: the backward then writes fresh tensors (no `grad += new` kernel per parameter, no 46 MB
        zero fill) and sync() packs them into the flat buffer with one multi-tensor copy.  `set_to_none` is accepted for
        DDP / optimizer call-site compatibility; the gradients are always dropped."""
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def sync(self):
        """pack the gradients into the flat buffer, average it over the ranks in place, and leave every .grad a view of it"""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()                            # a parameter the loss did not reach contributes zeros to the average
            elif p.grad is not v:
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v
        if self.world <= 1:
            return
        all_reduce_(self.flat)                       # SUM, then one in-place scale (exact for 2, 4, 8 ranks): no reliance on
        self.flat.div_(self.world)                   # the backend offering an AVG reduction
