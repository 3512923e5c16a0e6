"""GPU parity: CurveNet mirror (kNN / FPS / ball query / gathers on the HIP kernels) vs the reference's golden logits and
input gradients. Parity for this victim is pinned by fixtures generated from the real reference (no separate oracle
restatement yet)."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ref_torch as ort

pytestmark = pytest.mark.gpu


def test_curvenet_logits_and_input_grad_vs_reference(dev):
    fx = np.load(os.path.join(GOLDEN, "curvenet.npz"))
    cn = importlib.import_module("3dpointcloudattack_amd.model.curvenet")
    m = cn.CurveNet(num_classes=40)
    sd = ort.seeded_state_dict(m, 9, gain=1.0)
    m.load_state_dict(sd)
    assert ort.state_sha256(sd) == str(fx["sha256"])
    m = m.eval().to(dev)
    for nm in ("n1024", "n2048"):
        x = torch.from_numpy(fx[f"{nm}_x"]).to(dev).requires_grad_()
        out = m(x)
        assert len(out) == 3
        logits, ref = out[0], fx[f"{nm}_logits"]
        # measured: 5e-8 absolute on logits of magnitude 0.16 (walks take hard arg-max decisions, so a flipped
        # neighbour would show up as a much larger jump)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5, err_msg=nm)
        assert np.array_equal(logits.argmax(1).cpu().numpy(), ref.argmax(1)), nm
        (logits * torch.from_numpy(fx[f"{nm}_w"]).to(dev)).sum().backward()
        got, gref = x.grad.cpu().numpy(), fx[f"{nm}_gx"]
        assert np.isfinite(got).all()
        assert np.linalg.norm(got - gref) / np.linalg.norm(gref) < 5e-3, nm     # measured 5e-4 / 4e-5
