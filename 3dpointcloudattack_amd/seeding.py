"""Seeded synthetic weights for victims (no checkpoint ships with the reference; SURVEY §8(c) "Weights for label
parity"): the same recipe builds the reference classes' weights when the golden fixtures are generated, the oracle's
and this package's mirrors, because it only looks at `state_dict` key names and shapes. Pure torch-CPU, no GPU."""
import numpy as np
import torch


def seeded_state_dict(model, seed, gain=2.45):
    """Deterministic non-trivial state for ANY module with the reference's key names: conv/linear weights
    U(+-gain/sqrt(fan_in)) (gain sqrt(6) = Kaiming-uniform, which makes the random-init victim input-sensitive
    enough for short attacks to succeed), BN gamma 1+0.1n, beta 0.1n, running_mean 0.1n, running_var 1+0.2u. Keys are processed in
    sorted order from one CPU generator, so the reference class, this oracle and the HIP mirror get identical
    tensors as long as their state_dict keys and shapes agree (which is itself part of the drop-in contract)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    ref = model.state_dict()
    for k in sorted(ref):
        v = ref[k]
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            sd[k] = 1.0 + 0.2 * torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif (k.rsplit(".", 1)[0] + ".running_mean") in ref:  # BatchNorm gamma / beta
            sd[k] = (1.0 + 0.1 * torch.randn(v.shape, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            sd[k] = (torch.rand(v.shape, generator=g) * 2 - 1) * (gain / np.sqrt(fan_in))
        else:  # conv / linear bias
            sd[k] = (torch.rand(v.shape, generator=g) * 2 - 1) * 0.05
    return sd


def state_sha256(sd):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()
