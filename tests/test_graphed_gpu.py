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
    assert g.stats["replayed"] >= 6 and g.stats["eager"] == 0 and g.stats["captures"] == 2, g.stats


def test_forwards_waiting_for_backward_keep_their_memory(dev):
    """Several forwards before one backward (expectation-over-transformation loops): each replays its own replica of
    the capture, beyond MAX_REPLICAS the call runs eagerly; the gradients are those of the eager victim. The first
    forward's OUTPUT tensor is dropped on purpose: what keeps a replica busy is its autograd node."""
    m = _victim(dev, "dgcnn")
    g = graphed.wrap(m)
    gen = torch.Generator().manual_seed(1)
    n = graphed.MAX_REPLICAS + 1
    xs = [(torch.rand(2, 3, 256, generator=gen) - 0.5).to(dev).requires_grad_() for _ in range(n)]
    w = torch.randn(2, 40, generator=gen).to(dev)
    loss = 0.
    for x in xs:
        loss = loss + (g(x)[0] * w).sum()
    assert g.stats["captures"] == graphed.MAX_REPLICAS and g.stats["eager"] == 1
    loss.backward()
    rs = [x.detach().clone().requires_grad_() for x in xs]
    sum((m(r)[0] * w).sum() for r in rs).backward()
    for x, r in zip(xs, rs):
        assert (x.grad - r.grad).norm() <= 2e-3 * r.grad.norm() + 1e-12
    before = dict(g.stats)
    g(xs[0].detach().clone().requires_grad_())[0].sum().backward()      # all replicas are free again: a replay
    assert g.stats["eager"] == before["eager"] and g.stats["captures"] == before["captures"]


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


def test_capture_survives_cache_invalidation(dev):
    """.eval() / .to(device) drop the victims' folded-weight caches even when nothing changed; a capture made before
    must keep replaying correctly (it keeps the tensors it was captured with alive)."""
    m = _victim(dev, "pointnet")
    g = graphed.wrap(m)
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(3, 3, 200, generator=gen) - 0.5).to(dev)
    w = torch.randn(3, 40, generator=gen).to(dev)
    _grad(g, x, w)
    ref_l, ref_g, _ = _grad(m, x, w)
    for _ in range(3):
        m.train(True), m.eval(), m.to(dev), m._invalidate()
        junk = [torch.full((1 << 16,), float("nan"), device=dev) for _ in range(64)]    # reuse whatever was freed
        l, gr, _ = _grad(g, x, w)
        del junk
        torch.testing.assert_close(l, ref_l, rtol=1e-4, atol=1e-5)
        assert (gr - ref_g).norm() <= 2e-3 * ref_g.norm() + 1e-12
    assert g.stats["captures"] == 1


def test_leaf_grad_accumulates_across_replays(dev):
    """the replayed backward must not hand its static grad buffer to AccumulateGrad — two backward passes
    into the same leaf without clearing .grad give g1 + g2, as eager does."""
    m = _victim(dev, "pointnet")
    g = graphed.wrap(m)
    gen = torch.Generator().manual_seed(8)
    x = (torch.rand(2, 3, 160, generator=gen) - 0.5).to(dev)
    w1, w2 = torch.randn(2, 40, generator=gen).to(dev), torch.randn(2, 40, generator=gen).to(dev)
    xe, xg = x.clone().requires_grad_(), x.clone().requires_grad_()
    for model, leaf in ((m, xe), (g, xg)):
        (model(leaf)[0] * w1).sum().backward()
        (model(leaf)[0] * w2).sum().backward()
    assert g.stats["replayed"] >= 2
    assert (xg.grad - xe.grad).norm() <= 2e-3 * xe.grad.norm()


def test_mode_switch_recaptures(dev):
    """The backward kernel flavour (ordered sums / float atomics) is baked into a capture: a graphed victim used with
    deterministic=False must not hand that capture to a later deterministic call of the same shape. After an
    atomic-mode pass, deterministic passes through the SAME wrapper are bit-equal to each other and to the eager victim."""
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    m = _victim(dev, "dgcnn")
    g = graphed.wrap(m)
    gen = torch.Generator().manual_seed(5)
    w = torch.randn(4, 40, generator=gen).to(dev)
    x = (torch.rand(4, 3, 1024, generator=gen) - 0.5).to(dev)
    with ops.deterministic(False):
        _grad(g, x, w)
        _grad(g, x, w)
    caps = g.stats["captures"]
    with ops.deterministic(True):
        _, g1, _ = _grad(g, x, w)
        _, g2, _ = _grad(g, x, w)
        _, ge, _ = _grad(m, x, w)
    assert g.stats["captures"] == caps + 1, g.stats       # its own capture, not the atomic one's
    assert torch.equal(g1, g2) and torch.equal(g1, ge)
