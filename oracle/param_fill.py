"""Deterministic, name-keyed parameter fill (TEST INFRASTRUCTURE).

Golden fixtures store only inputs and outputs, never the ~10^8 weights: both the reference model
(in tests/golden/make_golden.py) and the model under test are filled by this function, which
derives every tensor from (seed, parameter NAME, shape) alone -- independent of construction
order, RNG consumption during __init__ and of the implementing class."""
from __future__ import annotations

import zlib

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) % (2 ** 31))
    return g


def fill_params(module: torch.nn.Module, seed: int = 0, prefix_filter: str | None = None) -> None:
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if prefix_filter is not None and not name.startswith(prefix_filter):
                continue
            if not torch.is_floating_point(t):
                continue  # num_batches_tracked etc.
            g = _gen(name, seed)
            shape = tuple(t.shape)
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "running_var":
                v = 0.8 + 0.4 * torch.rand(shape, generator=g)
            elif leaf == "running_mean":
                v = 0.1 * torch.randn(shape, generator=g)
            elif t.dim() >= 2 and leaf != "logit_scale":
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                if t.dim() == 4:            # 1x1 conv of the SharedMLPs: keep activations O(1)
                    std = (2.0 / fan_in) ** 0.5
                elif "embeddings" in name or fan_in > 4096:
                    std = 0.02
                else:
                    std = min(0.05, 1.0 / fan_in ** 0.5)
                v = std * torch.randn(shape, generator=g)
            elif t.dim() == 0:
                continue                    # logit_scale keeps its init
            elif leaf == "weight":          # every 1-D `weight` is a LayerNorm / BatchNorm scale
                v = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:                           # biases and other 1-D parameters
                v = 0.05 * torch.randn(shape, generator=g)
            t.copy_(v.to(t.dtype))
