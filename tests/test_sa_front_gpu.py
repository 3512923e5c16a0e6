"""GPU: the per-point / per-centre form of a set-abstraction layer's first 1x1 convolution (csrc/sa_front.hip,
ops.affine3 / ops.sa_front, model/pointnet2_utils.py:113,118-135,190-197 of the reference), the plain group max
(pc3d_rows_max_f32, :198), the second gradient operand of the Adam launch, and the attack loops' "direct terms" form
(functor.per_sample_terms / per_sample) against the general autograd form they replace."""
import importlib

import numpy as np
import pytest
import torch

from helpers import hip_pointnet, unit_cloud
from oracle import ref_torch as ort

pytestmark = pytest.mark.gpu
M = importlib.import_module


@pytest.mark.parametrize("B,N,C,cf", [(2, 100, 64, True), (3, 257, 128, False), (1, 1, 4, True), (2, 64, 24, False),
                                      (2, 33, 512, True)])
def test_affine3_values_and_gradient_vs_float64(ops, dev, B, N, C, cf):
    rng = np.random.default_rng(B * 1000 + N + C)
    base = torch.from_numpy(rng.standard_normal((B, 3, N) if cf else (B, N, 3)).astype(np.float32)).to(dev).requires_grad_()
    x = base.permute(0, 2, 1) if cf else base
    w = torch.from_numpy(rng.standard_normal((C, 3)).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(dev)
    gw = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    for b_, sign in ((None, 1.0), (bias, -1.0), (bias, 1.0)):
        base.grad = None
        y = ops.affine3(x, w, b_, sign)
        (y * gw).sum().backward()
        x64 = x.detach().double().requires_grad_()
        y64 = sign * (x64 @ w.double().t()) + (b_.double() if b_ is not None else 0.0)
        (y64 * gw.double()).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), y64.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
        g = base.grad.permute(0, 2, 1) if cf else base.grad
        np.testing.assert_allclose(g.cpu().numpy(), x64.grad.cpu().numpy(), rtol=2e-5, atol=1e-4 * C ** 0.5)
        assert base.grad.is_contiguous()
    # run == run, and a cloud's rows do not depend on the batch it is in
    y1 = ops.affine3(x.detach(), w, bias, -1.0)
    y2 = ops.affine3(x.detach()[:1], w, bias, -1.0)
    assert torch.equal(y1[:1], y2) and torch.equal(y1, ops.affine3(x.detach(), w, bias, -1.0))


@pytest.mark.parametrize("B,N,S,D,C1", [(2, 256, 64, 0, 64), (3, 128, 32, 16, 128), (2, 200, 50, 131, 32)])
def test_sa_front_vs_separate_operators(ops, dev, B, N, S, D, C1):
    """new_xyz / P / Bc and the gradients of xyz and the features against the composition it replaces (gather of the
    centres, two products, a row gather and a subtraction, autograd between them), REPEATED centres included."""
    rng = np.random.default_rng(N + D)
    base = torch.from_numpy(rng.standard_normal((B, 3, N)).astype(np.float32)).to(dev)
    pts0 = torch.from_numpy(rng.standard_normal((B, N, D)).astype(np.float32)).to(dev) if D else None
    idx = torch.from_numpy(rng.integers(0, N, (B, S)).astype(np.int32)).to(dev)
    idx[:, 1] = idx[:, 0]                                              # a centre drawn twice
    wx = torch.from_numpy(rng.standard_normal((C1, 3)).astype(np.float32)).to(dev)
    wf = torch.from_numpy((rng.standard_normal((C1, D)) / max(D, 1) ** 0.5).astype(np.float32)).to(dev) if D else None
    b1 = torch.from_numpy(rng.standard_normal(C1).astype(np.float32)).to(dev)
    g_new = torch.from_numpy(rng.standard_normal((B, S, 3)).astype(np.float32)).to(dev)
    g_P = torch.from_numpy(rng.standard_normal((B, N, C1)).astype(np.float32)).to(dev)
    g_Bc = torch.from_numpy(rng.standard_normal((B, S, C1)).astype(np.float32)).to(dev)

    def run(fn):
        xb = base.clone().requires_grad_()
        p = pts0.clone().requires_grad_() if D else None
        new_xyz, P, Bc = fn(xb.permute(0, 2, 1), p)
        ((new_xyz * g_new).sum() + (P * g_P).sum() + (Bc * g_Bc).sum()).backward()
        return new_xyz.detach(), P.detach(), Bc.detach(), xb.grad, (p.grad if D else None)

    def fused(x, p):
        return ops.sa_front(x, p, idx, wx, wf, b1)

    def ref64(x, p):
        x = x.double()
        li = idx.long()
        new_xyz = torch.gather(x, 1, li[:, :, None].expand(-1, -1, 3))
        px = x @ wx.double().t()
        P = px if p is None else px + p.double() @ wf.double().t()
        Bc = b1.double() - torch.gather(px, 1, li[:, :, None].expand(-1, -1, C1))
        return new_xyz, P, Bc

    a = run(fused)
    b = run(lambda x, p: tuple(t.float() for t in ref64(x, p)))        # float64 arithmetic, float32 leaves
    for got, ref, nm in zip(a, b, ("new_xyz", "P", "Bc", "grad xyz", "grad pts")):
        if got is None:
            assert ref is None
            continue
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * max(C1, D, 1) ** 0.5, err_msg=nm)
    assert a[3].is_contiguous()                                        # the gradient comes back channels-first, no copy
    a2 = run(fused)
    assert all(torch.equal(u, v) for u, v in zip(a, a2) if u is not None)            # run == run (ordered scatter)


@pytest.mark.parametrize("G,ns,C", [(64, 128, 1024), (3, 5, 7), (1, 1, 1)])
def test_rows_max_vs_torch(ops, dev, G, ns, C):
    lib = M("3dpointcloudattack_amd._lib")
    rng = np.random.default_rng(G + ns)
    y = torch.from_numpy(rng.integers(-4, 5, (G, ns, C)).astype(np.float32)).to(dev)      # many ties
    out = torch.empty((G, C), dtype=torch.float32, device=dev)
    arg = torch.empty((G, C), dtype=torch.int64, device=dev)
    lib.call("pc3d_rows_max_f32", y.data_ptr(), G, ns, C, out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    ref = y.cpu().numpy()
    assert np.array_equal(out.cpu().numpy(), ref.max(1))
    assert np.array_equal(arg.cpu().numpy(), ref.argmax(1))            # numpy: the FIRST maximum


def test_ssg_forward_backward_front_forms_agree(dev):
    """PointNet++ SSG with the fused front (SA_FRONT / SPLIT_GROUP_ALL) against the separate-operator form it replaces:
    same FPS starts, logits and input gradient to fp32 rounding."""
    pu = M("3dpointcloudattack_amd.model.pointnet2_utils")
    ssg = M("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg(40)
    ssg.load_state_dict(ort.seeded_state_dict(ssg, 3))
    ssg = ssg.eval().to(dev)
    rng = np.random.default_rng(2)
    x0 = torch.from_numpy(np.stack([unit_cloud(rng, 1024) for _ in range(3)])).transpose(1, 2).contiguous().to(dev)
    gw = torch.from_numpy(rng.standard_normal((3, 40)).astype(np.float32)).to(dev)

    def run(front, split):
        pu.SA_FRONT, pu.SPLIT_GROUP_ALL = front, split
        try:
            torch.manual_seed(5)
            x = x0.clone().requires_grad_()
            out = ssg(x)[0]
            (out * gw).sum().backward()
            return out.detach(), x.grad
        finally:
            pu.SA_FRONT, pu.SPLIT_GROUP_ALL = True, True

    o1, g1 = run(True, True)
    o0, g0 = run(False, False)
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=1e-3, atol=1e-5 * float(g0.abs().max()) + 1e-7)
    o2, g2 = run(True, True)
    assert torch.equal(o1, o2) and torch.equal(g1, g2)


def test_adam_clip_step_second_gradient(ops, dev):
    rng = np.random.default_rng(9)
    B, K = 3, 300
    p0 = torch.from_numpy(rng.standard_normal((B, 3, K)).astype(np.float32)).to(dev)
    ga = torch.from_numpy(rng.standard_normal((B, 3, K)).astype(np.float32)).to(dev)
    gb = torch.from_numpy(rng.standard_normal((B, 3, K)).astype(np.float32)).to(dev)
    ori = p0 + 0.01
    outs = []
    for two in (True, False):
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        for t in (1, 2, 3):
            if two:
                ops.adam_clip_step(p, ga, m, v, t, 1e-2, ori=ori, budget=0.18, g2=gb)
            else:
                ops.adam_clip_step(p, ga + gb, m, v, t, 1e-2, ori=ori, budget=0.18)
        outs.append((p, m, v))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("dist_name", ["chamfer", "chamfer_knn", "hausdorff", "chamfer_both"])
def test_direct_terms_gradient_equals_general_form(ops, dev, dist_name):
    """One iteration's gradient of `adv_func(logits).mean() + dist_func(adv, ori).mean() * K` (attack/KNN/KNN_attack.py:
    117-125) assembled by autograd from the functors' scalar outputs against the direct form: the functors' per-sample terms
    with d loss / d term built in and a backward started from the terms."""
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
    du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    dist = {"chamfer": du.ChamferDist(), "chamfer_knn": du.ChamferkNNDist(), "hausdorff": du.HausdorffDist(),
            "chamfer_both": du.ChamferDist(method="both")}[dist_name]
    model, _ = hip_pointnet(0, dev)
    rng = np.random.default_rng(4)
    B, K = 4, 512
    ori = torch.from_numpy(np.stack([unit_cloud(rng, K) for _ in range(B)])).transpose(1, 2).contiguous().to(dev)
    x0 = (ori + 0.02 * torch.randn_like(ori)).contiguous()
    with torch.no_grad():
        target = model(ori)[0].argmax(1)
    af = adv.UntargetedLogitsAdvLoss(kappa=5.)
    for ratio in (1.0, 0.25):
        xa = x0.clone().requires_grad_()
        loss = af(model(xa)[0], target).mean() + dist(xa.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous()).mean() * K
        if ratio != 1.0:
            loss = loss * ratio
        loss.backward()
        xb = x0.clone().requires_grad_()
        alias = xb.detach().requires_grad_()
        up = np.float32(ratio) * np.float32(K) if ratio != 1.0 else np.float32(K)
        terms = [af.per_sample(model(xb)[0], target, np.float32(ratio))] + dist.per_sample_terms(alias, ori, up)
        torch.autograd.backward(terms, [ops.const_vec(dev, B, 1.0)] * len(terms))
        got = xb.grad + alias.grad
        np.testing.assert_allclose(got.cpu().numpy(), xa.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
        # the terms' VALUES are the functors' per-sample values
        ref_vec = dist(x0.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous(), batch_avg=False)
        tv = [t.detach() for t in terms[1:]]
        if dist_name == "chamfer_knn":
            val = tv[0] * dist.w1 + tv[1] * dist.w2
        elif dist_name == "chamfer_both":
            val = (tv[0] + tv[1]) / 2.
        else:
            val = tv[0]
        np.testing.assert_allclose(val.cpu().numpy(), ref_vec.cpu().numpy(), rtol=1e-6)


def test_knn_attack_direct_terms_vs_general_form(dev):
    """CWKNN on PointNet (deterministic victim), 30 iterations: the direct-terms loop and the general loop end within fp32
    rounding of each other, and the direct loop reproduces itself bit for bit."""
    knn = M("3dpointcloudattack_amd.attack.KNN.KNN_attack")
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
    du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    model, _ = hip_pointnet(0, dev)
    rng = np.random.default_rng(8)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, 512) for _ in range(4)]))
    with torch.no_grad():
        lab = model(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()

    def run(direct):
        atk = knn.CWKNN(model, None, None, None, None, None, adv.UntargetedLogitsAdvLoss(kappa=15.), du.ChamferkNNDist(),
                        cu.ProjectInnerClipLinf(budget=0.18), attack_lr=1e-2, num_iter=30)
        atk.direct_terms = direct
        torch.manual_seed(3)
        return atk.attack(pcs, lab)

    a, sa = run(True)
    b, sb = run(False)
    assert sa == sb
    d = np.abs(a - b)
    assert np.median(d) < 1e-5 and d.max() < 0.05, (np.median(d), d.max())
    a2, _ = run(True)
    assert np.array_equal(a, a2)


def test_geoa3_direct_terms_vs_general_form(dev):
    """geoA3_attack with the loss never formed (cfg.direct_terms, the default: the terms tensor and the cross-entropy carry
    d loss / d loss_n = 1 / B as constants) against the general `loss_n.mean().backward()` form: same per-iteration losses
    and best attack to fp32 rounding, for the batch mean and for a shard of a larger batch (cfg.global_batch)."""
    from test_oracle_golden import _geo_cfg
    ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
    net, _ = hip_pointnet(0, dev)
    rng = np.random.default_rng(3)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, 256) for _ in range(3)]))
    with torch.no_grad():
        labels = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
    for gb in (None, 12):
        res = []
        for direct in (True, False, True):
            cfg = _geo_cfg(iter_max_steps=12, binary_max_steps=2, npoint=256)
            cfg.direct_terms = direct
            if gb:
                cfg.global_batch = gb
            torch.manual_seed(1)
            np.random.seed(1)
            best, tgt, mask, steps, losses = ga.geoA3_attack(net, None, None, None, None, None, pcs, labels, cfg, 0, 1)
            res.append((best.cpu().numpy(), np.array(losses), mask))
        np.testing.assert_allclose(res[0][1], res[1][1], rtol=2e-4, atol=1e-5)
        assert np.array_equal(res[0][2], res[1][2])
        assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1])      # run == run


def test_cw_generic_direct_terms_vs_general_form(dev):
    """CW.attack on a victim that goes through autograd (DGCNN: no fused_loss_and_grad) with own functors: the direct-terms
    iteration against the general one (loss assembled by autograd): same success, best distances and clouds to rounding."""
    import types
    cwm = M("3dpointcloudattack_amd.attack.CW.CW_attack")
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
    du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    dg = M("3dpointcloudattack_amd.model.dgcnn")
    model = dg.DGCNN(types.SimpleNamespace(k=8, emb_dims=64, dropout=0.5), output_channels=10)
    model.load_state_dict(ort.seeded_state_dict(model, 3))
    model = model.eval().to(dev)
    rng = np.random.default_rng(6)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, 256) for _ in range(3)]))
    with torch.no_grad():
        lab = model(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
    for dist in (du.ChamferDist(), du.ChamferkNNDist()):
        res = []
        for direct in (True, False, True):
            atk = cwm.CW(model, model, adv_func=adv.UntargetedLogitsAdvLoss(kappa=0.), clip_func=cu.ClipPointsLinf(budget=0.18),
                         dist_func=dist, attack_lr=1e-2, init_weight=10., max_weight=80., binary_step=2, num_iter=15,
                         attack_method='untarget')
            atk.direct_terms = direct
            torch.manual_seed(2)
            np.random.seed(2)
            bd, ba, sn = atk.attack(pcs, lab)
            res.append((bd, ba, sn))
        assert res[0][2] == res[1][2]
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-3)
        d = np.abs(res[0][1] - res[1][1])
        assert np.median(d) < 1e-5, np.median(d)
        assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1])


@pytest.mark.parametrize("seeded", [False, True])
def test_knn_attack_predrawn_fps_starts_equal_live_draws(dev, seeded):
    """CWKNN on PointNet++ SSG: the loop's FPS start indices drawn ahead and uploaded once (PredrawnFpsStarts) against a
    draw + upload per sampling layer and forward — the same generator stream in the same order, so the attack's result is
    the same bit for bit, and the generator ends at the same position (the forwards after the loop draw the same values)."""
    knn = M("3dpointcloudattack_amd.attack.KNN.KNN_attack")
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
    du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    ssg = M("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg(40)
    ssg.load_state_dict(ort.seeded_state_dict(ssg, 3))
    ssg = ssg.eval().to(dev)
    rng = np.random.default_rng(12)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, 1024) for _ in range(3)]))
    lab = torch.tensor([3, 5, 7])

    def run(predraw):
        atk = knn.CWKNN(ssg, None, None, None, None, None, adv.UntargetedLogitsAdvLoss(kappa=5.), du.ChamferDist(),
                        cu.ProjectInnerClipLinf(budget=0.18), attack_lr=1e-2, num_iter=12,
                        sample_seeds=[11, 12, 13] if seeded else None)
        atk.predraw_starts = predraw
        torch.manual_seed(4)
        out, sn = atk.attack(pcs, lab)
        return out, sn, atk.attack_fail, torch.randint(0, 1 << 30, (4,)).tolist()

    a = run(True)
    b = run(False)
    assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]


@pytest.mark.parametrize("B,N,k", [(2, 300, 20), (1, 64, 5), (3, 1024, 40)])
def test_knn_graph_views_equal_slices(ops, dev, B, N, k):
    """pc3d_knn_graph_i32: the self-kNN graph and the two views CurveNet's blocks use (model/curvenet_util.py:10-17,
    idx[:, :, 1:], idx[:, :, :k]) from ONE launch — equal to the search + two slicing copies."""
    rng = np.random.default_rng(N + k)
    pts = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).to(dev)
    idx, noself, first = ops.knn_graph(pts, k)
    ref = ops.knn_raw(pts, pts, k + 1)[1]
    assert torch.equal(idx, ref) and torch.equal(noself, ref[:, :, 1:]) and torch.equal(first, ref[:, :, :k])
    assert noself.is_contiguous() and first.is_contiguous() and idx.dtype == torch.int32


def test_nn_int64_indices_from_the_search_launch(ops, dev):
    """knn_points (attack/GeoA3/knn_utils.py:22-55) hands out int64 positions: for K = 1 they come from the search launch
    (pc3d_nn_i64_f32), equal to the int32 indices; the distances' gradient is unchanged."""
    ku = M("3dpointcloudattack_amd.attack.GeoA3.knn_utils")
    rng = np.random.default_rng(1)
    a = torch.from_numpy(np.stack([unit_cloud(rng, 257) for _ in range(3)])).to(dev).requires_grad_()
    b = torch.from_numpy(np.stack([unit_cloud(rng, 100) for _ in range(3)])).to(dev)
    res = ku.knn_points(a, b, K=1)
    d32, i32 = ops.nn_raw(a.detach(), b)
    assert res.idx.dtype == torch.int64 and res.idx.shape == (3, 257, 1)
    assert torch.equal(res.idx.squeeze(-1), i32.long()) and torch.equal(res.dists.squeeze(-1).detach(), d32)
    res.dists.sum().backward()
    nearest = torch.gather(b, 1, i32.long()[..., None].expand(-1, -1, 3))
    np.testing.assert_allclose(a.grad.cpu().numpy(), (2 * (a.detach() - nearest)).cpu().numpy(), rtol=1e-5, atol=1e-6)
    r3 = ku.knn_points(a.detach(), b, K=3)
    assert r3.idx.dtype == torch.int64 and torch.equal(r3.idx[:, :, 0], i32.long())


@pytest.mark.parametrize("B,N,S", [(3, 512, 128), (2, 1000, 256), (2, 2048, 64), (1, 37, 37)])
def test_fps_same_indices_for_every_workgroup_size(ops, dev, B, N, S):
    """pc3d_fps_threads_f32: the sampling with 64 ... 1024 threads per cloud gives the indices of pc3d_fps_f32 (which picks
    the size by N) — the arg-max tie rule (lowest index) does not depend on how the points are spread over the lanes."""
    lib = M("3dpointcloudattack_amd._lib")
    rng = np.random.default_rng(N)
    x = torch.from_numpy(np.round(rng.standard_normal((B, N, 3)) * 4).astype(np.float32) / 4).to(dev)     # many exact ties
    start = torch.from_numpy(rng.integers(0, N, B).astype(np.int32)).to(dev)
    ref = ops.fps(x, S, start)
    for thr in (64, 128, 256, 512, 1024):
        if N > 32 * thr:
            continue
        out = torch.empty((B, S), dtype=torch.int32, device=dev)
        lib.call("pc3d_fps_threads_f32", thr, x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, N, S, start.data_ptr(),
                 out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert torch.equal(out, ref), thr
