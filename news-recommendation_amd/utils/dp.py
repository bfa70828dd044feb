"""Data-parallel gradient averaging for the two-tower model (one process per GPU, RCCL over xGMI).

The reference wraps the model in DistributedDataParallel (main/twotower.py:40-41).  Here the step is: drop the
gradients -> backward of loss with its incoming gradient scaled by 1/N (so every gradient is born pre-divided, as DDP
pre-divides its buckets) -> one multi-tensor copy of the fresh dense gradients into a flat fp32 buffer laid out
[dense parameters | token table]; the HIP path scatter-adds the table gradient straight into its slice -> TWO SUM
all-reduces (the 9.4 MB dense part starts when the backward reaches the token table's scatter-add and overlaps it; the
36.6 MB table part follows) -> Adam on views of the buffer.
No per-parameter autograd hooks and no communication thread beside the backward.  The design reason: the token-table
gradient is the LAST thing the backward produces, so DDP's 25 MB bucket overlap had ~10 MB to hide; two large
collectives use the xGMI links better than several buckets and keep the step rehearsable on one GPU.

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


def all_reduce_async(t):
    """start a SUM all-reduce of `t`; returns a handle whose wait() orders the current stream behind it.  Under gloo on
    device tensors (rehearsal) it runs synchronously through the host and returns None."""
    if _staged(t):
        all_reduce_(t)
        return None
    return dist.all_reduce(t, async_op=True)


# token tables: their dense [V, E] gradient is the LAST thing a backward produces (the embedding is the first op of the
# forward) and by far the largest (36.6 of 46 MB at 300-d)
LATE_SUFFIXES = ("embedding.weight", "word_embeddings.weight")


class _ScaleGrad(torch.autograd.Function):
    """identity in the forward; the gradient that comes back is multiplied by `c`"""

    @staticmethod
    def forward(ctx, x, c):
        ctx.c = c
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.c, None


class FlatGradSync(torch.nn.Module):
    """wraps the model like DDP does (`.module`, same forward); call zero_grad() before and sync() after backward.

    The flat buffer is laid out [dense parameters | token tables] and reduced as TWO collectives: the dense part
    (9.4 MB at 300-d) is packed and its all-reduce started as soon as the backward reaches the token table's scatter-add
    (ops.table_backward_hooks: every other gradient is final by then, the autograd engine runs parameter accumulation
    ahead of other nodes), so it travels while the scatter-add runs; the table part follows in sync().  Without the hook
    (CPU tests, torch modules) sync() issues both, one after the other: same result."""

    def __init__(self, module, world_size=None, late_suffixes=LATE_SUFFIXES, always_reduce=False):
        super().__init__()
        self.module = module
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        # always_reduce: issue the collectives with a single rank too (the RCCL smoke run of a one-GPU box)
        self._reduce = self.world > 1 or (always_reduce and dist.is_initialized())
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        late = [(n, p) for n, p in named if n.endswith(tuple(late_suffixes))]
        early = [(n, p) for n, p in named if not n.endswith(tuple(late_suffixes))]
        self.params = [p for _, p in early + late]
        self.n_early_params = len(early)
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for i, p in enumerate(self.params):
            if i == self.n_early_params:
                self.n_early = off
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if not late:
            self.n_early = off
        self.flat_early, self.flat_late = self.flat[:self.n_early], self.flat[self.n_early:]
        self._early_done, self._works, self._dirty = False, [], False
        self._prescaled = False                  # this step's loss went out of forward() with its gradient scaled by 1/N
        self._claimed = set()                    # late parameters whose gradient was produced in place this step
        if self._reduce:                         # replicas start from rank 0's weights (and buffers), like DDP
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    broadcast_(t.data, 0)
        try:                                     # the HIP path announces "only the token table's gradient is left"
            from newsrec_hip import ops
            import weakref
            me = weakref.ref(self)

            def hook():
                obj = me()
                if obj is not None and obj.flat.is_cuda:
                    if obj._early_done:         # a second backward since zero_grad(): its gradients would miss the reduce
                        obj._dirty = True
                    obj.sync_early()
            self._hook = hook
            ops.table_backward_hooks.append(hook)
            # the token tables' gradients are scatter-added straight into their slices of the flat buffer
            self._dest_keys = []
            if self.flat.is_cuda:
                for i in range(self.n_early_params, len(self.params)):
                    def dest(i=i):
                        obj = me()
                        return obj._claim(i) if obj is not None else None
                    key = self.params[i].data_ptr()
                    ops.table_grad_dest[key] = dest
                    self._dest_keys.append((key, dest))
        except ImportError:
            self._hook, self._dest_keys = None, []

    def __del__(self):
        try:
            from newsrec_hip import ops
            if self._hook in ops.table_backward_hooks:
                ops.table_backward_hooks.remove(self._hook)
            for key, dest in self._dest_keys:
                if ops.table_grad_dest.get(key) is dest:
                    del ops.table_grad_dest[key]
        except Exception:       # noqa: BLE001 — interpreter shutdown
            pass

    def _claim(self, i):
        """the flat buffer's slice of late parameter i, for the FIRST gradient of the step that asks (a table looked up twice
        in one forward gets its second gradient as a fresh tensor, which autograd adds onto the first)"""
        p = self.params[i]
        if i in self._claimed or p.grad is not None or p.data_ptr() != self._dest_keys[i - self.n_early_params][0]:
            return None
        self._claimed.add(i)
        return self.views[i].view(p.shape)       # a fresh tensor object on the same memory: autograd keeps it as .grad

    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        # the mean over ranks, DDP's way: gradients are born divided by N (exact for 2, 4, 8 ranks) and the all-reduce sums
        # (every training forward: two losses in one step are both scaled; sync() clears the mark, so a caller that never
        # calls zero_grad() on the wrapper is scaled on every step too)
        if self._reduce and self.world > 1 and torch.is_tensor(out) and out.requires_grad:
            out = _ScaleGrad.apply(out, 1.0 / self.world)
            self._prescaled = True
        return out

    def zero_grad(self, set_to_none=False):
        """drop the gradients: the backward then writes fresh tensors (no `grad += new` kernel per parameter, no 46 MB
        zero fill) and sync() packs them into the flat buffer with one multi-tensor copy.  `set_to_none` is accepted for
        DDP / optimizer call-site compatibility; the gradients are always dropped."""
        for p in self.params:
            p.grad = None
        if self._works:                          # a step abandoned between backward and sync(): let its collective finish
            for w in self._works:
                if w is not None:
                    w.wait()
        self._early_done, self._works, self._dirty = False, [], False
        self._prescaled = False
        self._claimed.clear()

    def _pack(self, lo, hi):
        src, dst = [], []
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is None:
                v.zero_()                            # a parameter the loss did not reach contributes zeros to the average
            elif p.grad is not v and p.grad.data_ptr() != v.data_ptr():     # (same memory: produced in place, _claim)
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            p.grad = v

    @torch.no_grad()
    def sync_early(self):
        """pack the dense parameters' gradients and start their all-reduce (once per step; later calls do nothing)"""
        if self._early_done:
            return
        self._early_done = True
        self._pack(0, self.n_early_params)
        if self._reduce and self.n_early:
            self._works.append(all_reduce_async(self.flat_early))

    @torch.no_grad()
    def sync(self):
        """pack the gradients into the flat buffer, average it over the ranks in place, and leave every .grad a view of it"""
        if self._dirty:
            raise RuntimeError("FlatGradSync: two backward passes between zero_grad() and sync() — the dense parameters' "
                               "all-reduce started during the first one; call zero_grad() before every backward")
        self.sync_early()
        self._pack(self.n_early_params, len(self.params))
        if not self._reduce:
            return
        if self.flat_late.numel():
            self._works.append(all_reduce_async(self.flat_late))
        for w in self._works:
            if w is not None:
                w.wait()
        self._works = []
        if not self._prescaled and self.world > 1:   # a loss that did not come out of forward(): SUM, then one in-place scale
            self.flat.div_(self.world)               # (no reliance on the backend offering an AVG reduction)
        self._prescaled = False
