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


def _walk_case(dev, C, k, cn, L, N, B=3, seed=0):
    wk = importlib.import_module("3dpointcloudattack_amd.model.walk")
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    g = torch.Generator().manual_seed(seed)
    w = wk.Walk(C, k, cn, L)
    with torch.no_grad():
        for bn in (w.agent_mlp[1], w.momentum_mlp[1]):        # non-trivial eval-mode BatchNorm statistics
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(bn.bias.shape, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(bn.bias.shape, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(bn.bias.shape, generator=g) + 0.5)
        w.agent_mlp[0].weight.copy_(torch.randn(w.agent_mlp[0].weight.shape, generator=g) * 0.5)
        w.momentum_mlp[0].weight.copy_(torch.randn(w.momentum_mlp[0].weight.shape, generator=g) * 0.5)
    w = w.eval().to(dev)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    x = torch.randn(B, C, N, generator=g).to(dev)
    adj = ops.knn_raw(xyz, xyz, k + 1)[1][:, :, 1:].long()                     # self excluded, like CIC.forward
    start = torch.stack([torch.randperm(N, generator=g)[:cn] for _ in range(B)]).to(dev).unsqueeze(2)
    return w, xyz.transpose(1, 2).contiguous(), x, adj, start


@pytest.mark.parametrize("C,k,cn,L,N", [(16, 20, 100, 5, 1024), (32, 20, 100, 5, 512), (8, 7, 10, 30, 200),
                                        (64, 63, 5, 3, 128), (16, 1, 3, 4, 64), (32, 20, 17, 6, 300)])
def test_fused_walk_matches_step_by_step(dev, C, k, cn, L, N):
    """pc3d_curve_walk_fwd/bwd_f32 (one launch each) vs the step-by-step torch formulation of model/walk.py:74-153,
    which the golden CurveNet fixtures pin against the reference: same curves (the walk takes hard arg-max decisions, so
    a different pick would change a whole curve) and the same gradient with respect to the features."""
    w, xyz, x, adj, start = _walk_case(dev, C, k, cn, L, N)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    w.fused = True
    got = w(xyz, xa, adj, start)
    w.fused = False
    ref = w(xyz, xb, adj, start)
    assert got.shape == ref.shape == (x.shape[0], C, cn, L)
    same = ((got - ref).abs().amax(dim=(1, 3)) <= 1e-5 * (1 + ref.abs().amax(dim=(1, 3))))   # per (cloud, curve)
    assert same.float().mean() >= 0.99, same.float().mean()
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1)).to(dev) * same[:, None, :, None]
    (got * gout).sum().backward()
    (ref * gout).sum().backward()
    err = (xa.grad - xb.grad).norm() / xb.grad.norm()
    assert torch.isfinite(xa.grad).all() and err < 2e-4, err


def test_fused_walk_rejects_unsupported_shapes(dev):
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    f = torch.zeros(1, 8, 12, device=dev)
    with pytest.raises(ValueError):
        ops.curve_walk(f, torch.zeros(1, 8, 3, dtype=torch.int32, device=dev), torch.zeros(1, 2, dtype=torch.int32, device=dev),
                       torch.zeros(24, device=dev), torch.zeros(1, device=dev), torch.zeros(2, 24, device=dev),
                       torch.zeros(2, device=dev), 3)


@pytest.mark.parametrize("B,N,K,C", [(2, 100, 20, 16), (3, 257, 7, 32), (1, 1, 1, 4), (2, 64, 20, 128)])
def test_edge_act_and_act_mean_match_torch(dev, B, N, K, C):
    """pc3d_edge_act_f32 / pc3d_act_mean_f32 (+ backward) vs the same ops written with torch gather / leaky_relu / mean."""
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    g = torch.Generator().manual_seed(B * 1000 + N)
    A = torch.randn(B, N, C, generator=g).to(dev).requires_grad_()
    Bc = torch.randn(B, N, C, generator=g).to(dev).requires_grad_()
    idx = torch.randint(0, N, (B, N, K), generator=g).to(dev)
    A2, Bc2 = A.detach().clone().requires_grad_(), Bc.detach().clone().requires_grad_()
    E = ops.edge_act(A, Bc, idx.to(torch.int32), 0.2)
    gathered = torch.gather(A2.unsqueeze(1).expand(-1, N, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, C))
    E2 = torch.nn.functional.leaky_relu(gathered + Bc2.unsqueeze(2), 0.2)
    assert torch.equal(E, E2)
    W = torch.randn(C, C, generator=g).to(dev) / C ** 0.5
    out = ops.act_mean(torch.nn.functional.linear(E, W), 0.2)
    out2 = torch.nn.functional.leaky_relu(torch.nn.functional.linear(E2, W), 0.2).mean(dim=2)
    torch.testing.assert_close(out, out2, rtol=1e-5, atol=1e-6)
    gout = torch.randn(out.shape, generator=g).to(dev)
    out.backward(gout)
    out2.backward(gout)
    for a, b in ((A.grad, A2.grad), (Bc.grad, Bc2.grad)):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))


def test_curvenet_graph_cache_sees_in_place_updates(dev):
    """CIC blocks at one resolution share the kNN graph through the xyz tensor they pass along; an in-place update of
    the caller's input (what the attack optimisers do) must not be served a stale graph."""
    cn = importlib.import_module("3dpointcloudattack_amd.model.curvenet")
    m = cn.CurveNet(num_classes=40)
    m.load_state_dict(ort.seeded_state_dict(m, 9, gain=1.0))
    m = m.eval().to(dev)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(2, 3, 1024, generator=g) - 0.5).to(dev)      # N == npoint of the first blocks: no down-sampling
    with torch.no_grad():
        m(x)
        x.copy_((torch.rand(2, 3, 1024, generator=g) - 0.5).to(dev))
        again = m(x)[0]
        fresh = m(x.clone())[0]
    assert torch.equal(again, fresh)


@pytest.mark.parametrize("C,cn,cl,N", [(16, 100, 5, 1024), (32, 100, 5, 300), (16, 10, 30, 64), (8, 3, 1, 17)])
def test_curve_aggregation_kv_matches_step_by_step(dev, C, cn, cl, N):
    """pc3d_curve_agg_kv_f32 (+ backward) and the two batched GEMMs around it vs the reference's sequence of 1x1 convs,
    softmaxes and products (CurveAggregation.forward_steps, pinned by the golden CurveNet fixtures): same output, same
    gradients with respect to the point features and the curves."""
    cu = importlib.import_module("3dpointcloudattack_amd.model.curvenet_util")
    g = torch.Generator().manual_seed(C * 7 + cn)
    agg = cu.CurveAggregation(C)
    with torch.no_grad():
        for p in agg.parameters():
            if p.dim() > 1:      # 1/sqrt(fan_in): attention logits of order one, as in a trained block
                p.copy_(torch.randn(p.shape, generator=g) / float(p[0].numel()) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2 + 1.0)
        bn = agg.convd[1]
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    agg = agg.eval().to(dev)
    B = 3
    x = torch.randn(B, C, N, generator=g).to(dev)
    # curves as the walk hands them over: a [B,C,cn,cl] view of a channels-last buffer
    curves = torch.randn(B, cn, cl, C, generator=g).to(dev)
    xa, ca = x.clone().requires_grad_(), curves.clone().requires_grad_()
    xb, cb = x.clone().requires_grad_(), curves.clone().requires_grad_()
    agg.fused = True
    got = agg(xa, ca.permute(0, 3, 1, 2))
    agg.fused = False
    ref = agg(xb, cb.permute(0, 3, 1, 2))
    assert got.shape == ref.shape == (B, C, N)
    # convc / convd are applied to the keys / values instead of the points (exact algebra, different rounding)
    assert (got - ref).norm() <= 2e-5 * ref.norm() and (got - ref).abs().max() <= 2e-4 * ref.abs().max()
    gout = torch.randn(ref.shape, generator=g).to(dev)
    (got * gout).sum().backward()
    (ref * gout).sum().backward()
    for a, b in ((xa.grad, xb.grad), (ca.grad, cb.grad)):
        assert (a - b).norm() <= 5e-4 * b.norm() + 1e-12, ((a - b).norm(), b.norm())


def _load_block(mod, fx, prefix, dev):
    sd = {k[len(prefix) + 4:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(prefix + ".sd.")}
    mod.load_state_dict(sd)
    return mod.eval().to(dev)


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("fused", [True, False])
def test_walk_vs_reference_fixture(dev, fused):
    """K16 (and the step-by-step formulation it is tested against) vs the REAL reference Walk on non-trivial weights
    and BatchNorm statistics (tests/golden/curvenet_blocks.npz, generated by make_golden.py from model/walk.py): this
    is what pins the momentum path, including the reference's `.view(B,1,cn,2)` reinterpretation of the softmax."""
    fx = np.load(os.path.join(GOLDEN, "curvenet_blocks.npz"))
    wk = importlib.import_module("3dpointcloudattack_amd.model.walk")
    B, C, N = fx["walk.x"].shape
    k, (cn, cl) = fx["idx"].shape[2] - 1, fx["walk.curves"].shape[2:]
    w = _load_block(wk.Walk(C, k, cn, cl), fx, "walk", dev)
    w.fused = fused
    x = torch.from_numpy(fx["walk.x"]).to(dev).requires_grad_()
    xyz = torch.from_numpy(fx["xyz"]).to(dev)
    adj = torch.from_numpy(fx["idx"]).to(dev)[:, :, 1:]
    got = w(xyz, x, adj, torch.from_numpy(fx["walk.start"]).to(dev))
    ref = torch.from_numpy(fx["walk.curves"]).to(dev)
    same = (got - ref).abs().amax(dim=(1, 3)) <= 1e-5 * (1 + ref.abs().amax(dim=(1, 3)))        # per (cloud, curve)
    assert same.float().mean() >= 0.99, same.float().mean()
    (got * torch.from_numpy(fx["walk.w"]).to(dev)).sum().backward()
    if bool(same.all()):
        assert _rel(x.grad, fx["walk.gx"]) < 1e-4
    else:                                   # a flipped near-tie pick changes that curve's contribution only
        assert _rel(x.grad, fx["walk.gx"]) < 0.2


def test_curve_aggregation_vs_reference_fixture(dev):
    """K18 + the two batched GEMMs vs the real reference CurveAggregation (model/curvenet_util.py:379-437)."""
    fx = np.load(os.path.join(GOLDEN, "curvenet_blocks.npz"))
    cu = importlib.import_module("3dpointcloudattack_amd.model.curvenet_util")
    C = fx["agg.x"].shape[1]
    agg = _load_block(cu.CurveAggregation(C), fx, "agg", dev)
    for fused in (True, False):
        agg.fused = fused
        x = torch.from_numpy(fx["agg.x"]).to(dev).requires_grad_()
        cv = torch.from_numpy(fx["agg.curves"]).to(dev).requires_grad_()
        y = agg(x, cv)
        assert _rel(y, fx["agg.y"]) < 2e-5, fused
        (y * torch.from_numpy(fx["agg.w"]).to(dev)).sum().backward()
        assert _rel(x.grad, fx["agg.gx"]) < 1e-4 and _rel(cv.grad, fx["agg.gcurves"]) < 1e-4, fused


def test_lpfa_vs_reference_fixture(dev):
    """The first LPFA (EdgeConv identity -> pc3d_edge_max_f32) and an inner LPFA (K17: pc3d_edge_act_f32 + GEMM +
    pc3d_act_mean_f32) vs the real reference LPFA (model/curvenet_util.py:175-236) on the reference's own kNN graph."""
    fx = np.load(os.path.join(GOLDEN, "curvenet_blocks.npz"))
    cu = importlib.import_module("3dpointcloudattack_amd.model.curvenet_util")
    idx = torch.from_numpy(fx["idx"]).to(dev)
    k = idx.shape[2] - 1
    C = fx["lpfa1.x"].shape[1]
    for tag, mod in (("lpfa0", cu.LPFA(9, 32, k, mlp_num=1, initial=True)), ("lpfa1", cu.LPFA(C, C, k, mlp_num=1, initial=False))):
        mod = _load_block(mod, fx, tag, dev)
        pts = torch.from_numpy(fx["xyz"]).to(dev).requires_grad_()
        f = pts if tag == "lpfa0" else torch.from_numpy(fx["lpfa1.x"]).to(dev).requires_grad_()
        y = mod(f, pts, idx=idx[:, :, :k])
        assert _rel(y, fx[f"{tag}.y"]) < 2e-5, tag
        (y * torch.from_numpy(fx[f"{tag}.w"]).to(dev)).sum().backward()
        assert _rel(pts.grad, fx[f"{tag}.gxyz"]) < 2e-4, tag
        if tag == "lpfa1":
            assert _rel(f.grad, fx["lpfa1.gx"]) < 1e-4


def test_curvenet_stage_trace_vs_reference(dev):
    """Stage-by-stage against the REAL reference on Kaiming-scale weights (tests/golden/curvenet_trace.npz): under the
    gain-1.0 weights of curvenet.npz everything past the first blocks is bias-dominated, so a deep mistake could not
    show there. Per stage the relative L2 error of the features (fraction of points that agree) is checked; the
    walk takes hard arg-max decisions, so individual curves may differ on near-ties."""
    fx = np.load(os.path.join(GOLDEN, "curvenet_trace.npz"))
    cn = importlib.import_module("3dpointcloudattack_amd.model.curvenet")
    m = cn.CurveNet(num_classes=40)
    sd = ort.seeded_state_dict(m, 9)
    m.load_state_dict(sd)
    assert ort.state_sha256(sd) == str(fx["sha256"])
    m = m.eval().to(dev)
    got = {}

    def keep(name, idx=None):
        def fn(mod, inp, out):      # CurveNet.forward runs its blocks channels-last: back to the reference layout
            if isinstance(out, tuple):
                got[name] = (out[0].transpose(1, 2), out[1].transpose(1, 2))
            elif out.dim() == 4:
                got[name] = out.permute(0, 3, 1, 2)
            else:
                got[name] = out.transpose(1, 2)
        return fn

    hooks = [getattr(m, nm).register_forward_hook(keep(nm)) for nm in fx["stages"] if nm != "conv0"]
    hooks.append(m.cic11.curvegrouping.register_forward_hook(keep("cic11_curves")))
    hooks.append(m.cic11.curveaggregation.register_forward_hook(keep("cic11_agg")))
    hooks.append(m.cic11.lpfa.register_forward_hook(keep("cic11_lpfa")))
    with torch.no_grad():
        logits = m(torch.from_numpy(fx["x"]).to(dev))[0].cpu().numpy()
    for h in hooks:
        h.remove()
    report = []

    def rel(a, b):
        return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

    for nm in fx["stages"]:
        nm = str(nm)
        if nm == "conv0":
            continue
        out = got[nm]
        feats = (out[1] if isinstance(out, tuple) else out).cpu().numpy()[:, ::int(fx[f"{nm}_cstride"])]
        ref = fx[f"{nm}_feat"]
        per_pt = np.abs(feats - ref).max(axis=1) <= 1e-3 * (1 + np.abs(ref).max(axis=1))
        report.append((nm, rel(feats, ref), float(per_pt.mean())))
        if isinstance(out, tuple):
            np.testing.assert_allclose(out[0].cpu().numpy(), fx[f"{nm}_xyz"], atol=1e-6, err_msg=nm + " positions")
    curves = got["cic11_curves"].cpu().numpy()
    same_curve = np.abs(curves - fx["cic11_curves"]).max(axis=(1, 3)) <= 1e-3 * (1 + np.abs(fx["cic11_curves"]).max(axis=(1, 3)))
    report.append(("cic11_curves", rel(curves, fx["cic11_curves"]), float(same_curve.mean())))
    report.append(("cic11_agg", rel(got["cic11_agg"].cpu().numpy()[:, ::2], fx["cic11_agg"]), -1.0))
    report.append(("cic11_lpfa", rel(got["cic11_lpfa"].cpu().numpy()[:, ::2], fx["cic11_lpfa"]), -1.0))
    report.append(("logits", rel(logits, fx["logits"]), -1.0))
    print("\nstage, rel L2 error, fraction of points/curves equal:")
    for r in report:
        print("  %-14s %.3e  %.4f" % r)
    errs = dict((r[0], r[1]) for r in report)
    assert errs["lpfa"] < 1e-5
    assert max(errs.values()) < 5e-2, report


def test_geometry_side_stream_equals_inline(dev):
    """CurveNet.forward computes the FPS chain / ball queries / kNN graphs on a side stream beside the feature path
    (CurveNet._geometry): same logits bit for bit as the inline order, same input gradient up to the float-atomic
    order noise of the scatter kernels, and the CPU generator advances identically (the discarded FPS start draws)."""
    cn = importlib.import_module("3dpointcloudattack_amd.model.curvenet")
    m = cn.CurveNet(num_classes=40)
    m.load_state_dict(ort.seeded_state_dict(m, 9))
    m = m.eval().to(dev)
    g = torch.Generator().manual_seed(4)
    x0 = (torch.rand(3, 3, 2048, generator=g) - 0.5).to(dev)
    up = torch.randn(3, 40, generator=g).to(dev)
    res = []
    for side in (True, False, True):
        m.geometry_stream = side
        torch.manual_seed(11)
        x = x0.clone().requires_grad_()
        logits = m(x)[0]
        (logits * up).sum().backward()
        res.append((logits.detach().clone(), x.grad.clone(), torch.rand(1).item()))
    m.geometry_stream = True
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][0], res[2][0])
    assert res[0][2] == res[1][2]
    for a in (res[1][1], res[2][1]):
        assert float((a - res[0][1]).norm() / res[0][1].norm()) < 1e-4
