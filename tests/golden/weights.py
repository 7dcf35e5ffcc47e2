"""Deterministic, name-keyed weights so a golden vector never has to ship a state_dict.

``fill_(module, seed)`` overwrites every entry of ``module.state_dict()`` with values drawn from a
generator seeded by crc32(key) -- so the reference module (in make_golden.py) and the build's module
(in the tests) get bit-identical parameters as long as their state_dict KEYS and SHAPES agree, which
is itself part of the drop-in contract being tested.
"""
import zlib

import numpy as np
import torch


def tensor_for(key, shape, dtype, seed=0):
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    if dtype in (torch.int64, torch.int32):            # num_batches_tracked
        return torch.zeros(shape, dtype=dtype)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_var":
        a = rng.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        a = rng.uniform(-0.2, 0.2, size=shape)
    elif leaf in ("bias", "in_proj_bias"):
        a = rng.uniform(-0.1, 0.1, size=shape)
    elif leaf == "weight" and len(shape) == 1:          # norm scales
        a = rng.uniform(0.8, 1.2, size=shape)
    else:                                               # matrices / conv kernels / embeddings
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        bound = (3.0 / max(fan_in, 1)) ** 0.5
        a = rng.uniform(-bound, bound, size=shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dtype)


def fill_(module, seed=0, skip_prefixes=()):
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if any(k.startswith(p) for p in skip_prefixes):
            new[k] = v
        else:
            new[k] = tensor_for(k, tuple(v.shape), v.dtype, seed)
    module.load_state_dict(new, strict=True)
    return module
