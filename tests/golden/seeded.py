"""Name-seeded parameter values shared by the golden generators (reference side) and the tests
(our side): parameter `name` of a module gets randn(seed = crc32(salt + name)) * scale, so big
state dicts need not be stored in the fixtures -- both sides rebuild identical weights as long as
the parameter NAMES and shapes agree (which is itself part of the drop-in contract)."""
import zlib

import torch


def fill_by_name_(module, salt: str, scale: float = 0.15, overrides=None, prefix: str = ""):
    """`prefix` = where `module` sits in the full model ("transformer.", "input_proj." ...), so
    that a sub-module filled on its own gets the values it would get as part of the whole."""
    overrides = overrides or {}
    with torch.no_grad():
        for local, p in module.named_parameters():
            name = prefix + local
            g = torch.Generator().manual_seed(zlib.crc32((salt + name).encode()) & 0x7FFFFFFF)
            s = scale
            for key, val in overrides.items():
                if key in name:
                    s = val
            p.copy_(torch.randn(p.shape, generator=g) * s)
    return module


def layernorm_weights_plus_one_(module):
    """LayerNorm gains around 1 instead of around 0 (every 1-D `*.weight` whose name contains "norm"):
    with gains ~ N(0, s) a deep random transformer collapses all tokens onto one direction, and the
    two-stage selection degenerates into a tie between the (identical) invalid proposals."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "norm" in name and name.endswith(".weight") and p.dim() == 1:
                p.add_(1.0)
    return module
