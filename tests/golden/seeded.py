"""Deterministic weights for the larger golden cases (shared by the generator
and the tests, so multi-MB weight blobs need not be committed).  CPU
torch.Generator streams are reproducible for one torch build; the fixture
stores an abs-sum checksum per tensor to detect drift."""
import math
import torch


def seeded_like(name, shape, gen):
    if "LayerNorm.weight" in name:
        return 1 + 0.1 * torch.randn(shape, generator=gen)
    if name.endswith("bias") or "bias_" in name:
        return 0.1 * torch.randn(shape, generator=gen)
    if name.endswith("embedding.weight"):
        return 0.5 * torch.randn(shape, generator=gen)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=gen) / math.sqrt(max(fan_in, 1))


def seeded_state(named_shapes, seed):
    """named_shapes: iterable of (name, shape) in a FIXED order."""
    gen = torch.Generator().manual_seed(seed)
    return {n: seeded_like(n, tuple(s), gen) for n, s in named_shapes}
