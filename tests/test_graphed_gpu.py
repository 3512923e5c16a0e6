"""GPU: hipGraph replay of a victim's forward/backward (3dpointcloudattack_amd/graphed.py) returns what the eager
launches return, and never lets a replay overwrite memory an earlier forward still needs."""
import importlib
import types

import pytest
import torch

from oracle import ref_torch as ort

pytestmark = pytest.mark.gpu
graphed = importlib.import_module("3dpointcloudattack_amd.graphed")


def _victim(dev, name):
    if name == "curvenet":
        m = importlib.import_module("3dpointcloudattack_amd.model.curvenet").CurveNet(num_classes=40)
    elif name == "dgcnn":
        m = importlib.import_module("3dpointcloudattack_amd.model.dgcnn").DGCNN(
            types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
    else:
        m = importlib.import_module("3dpointcloudattack_amd.model.pointnet").PointNetCls(k=40, feature_transform=False)
    m.load_state_dict(ort.seeded_state_dict(m, 3, gain=1.0))
    return m.eval().to(dev)


def _grad(model, x, w):
    xa = x.clone().requires_grad_()
    out = model(xa)
    logits = out[0]
    (logits * w).sum().backward()
    return logits.detach().clone(), xa.grad.clone(), out


@pytest.mark.parametrize("name", ["curvenet", "dgcnn", "pointnet"])
def test_replay_matches_eager(dev, name):
    m = _victim(dev, name)
    g = graphed.wrap(m)
    assert isinstance(g, graphed.GraphedVictim) and graphed.wrap(m) is g
    gen = torch.Generator().manual_seed(0)
    w = torch.randn(4, 40, generator=gen).to(dev)
    for trial in range(3):                                  # capture on the first call, replays after it
        x = (torch.rand(4, 3, 1024, generator=gen) - 0.5).to(dev)
        le, ge, oe = _grad(m, x, w)
        lg, gg, og = _grad(g, x, w)
        assert len(oe) == len(og)
        torch.testing.assert_close(lg, le, rtol=1e-4, atol=1e-5)
        assert (gg - ge).norm() <= 2e-3 * ge.norm() + 1e-12, (trial, g.stats)
        with torch.no_grad():
            torch.testing.assert_close(g(x)[0], m(x)[0], rtol=1e-4, atol=1e-5)
    assert g.stats["replayed"] >= 6 and g.stats["eager"] == 0 and g.stats["captures"] == 2


def test_second_forward_before_backward_runs_eagerly(dev):
    m = _victim(dev, "dgcnn")
    g = graphed.wrap(m)
    gen = torch.Generator().manual_seed(1)
    x1 = (torch.rand(2, 3, 256, generator=gen) - 0.5).to(dev).requires_grad_()
    x2 = (torch.rand(2, 3, 256, generator=gen) - 0.5).to(dev).requires_grad_()
    w = torch.randn(2, 40, generator=gen).to(dev)
    g(x1.detach().clone().requires_grad_())[0].sum().backward()          # capture
    a = g(x1)[0]
    b = g(x2)[0]                                                         # a still waits for its backward
    assert g.stats["eager"] == 1
    ((a + b) * w).sum().backward()
    r1, r2 = x1.detach().clone().requires_grad_(), x2.detach().clone().requires_grad_()
    ((m(r1)[0] + m(r2)[0]) * w).sum().backward()
    for got, ref in ((x1.grad, r1.grad), (x2.grad, r2.grad)):
        assert (got - ref).norm() <= 2e-3 * ref.norm() + 1e-12
    del a, b
    g(x1)[0].sum().backward()                                            # free again: replayed
    assert g.stats["eager"] == 1


def test_recaptures_when_weights_change(dev):
    m = _victim(dev, "pointnet")
    g = graphed.wrap(m)
    x = (torch.rand(2, 3, 128, generator=torch.Generator().manual_seed(2)) - 0.5).to(dev)
    with torch.no_grad():
        before = g(x)[0].clone()
        m.load_state_dict(ort.seeded_state_dict(m, 4, gain=1.0))
        after = g(x)[0]
        torch.testing.assert_close(after, m(x)[0], rtol=1e-4, atol=1e-5)
    assert not torch.allclose(before, after)


def test_non_deterministic_victims_are_left_alone(dev):
    ssg = importlib.import_module("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg(num_classes=40)
    assert graphed.wrap(ssg) is ssg
