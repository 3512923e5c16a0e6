"""GPU parity: FPS / ball query / grouping kernels and the PointNet++ SSG / MSG mirrors vs the reference's golden
outputs and the oracle."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import unit_cloud
from oracle import ref_torch as ort

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "pointnet2.npz"))


@pytest.fixture(scope="module")
def pu():
    return importlib.import_module("3dpointcloudattack_amd.model.pointnet2_utils")


def test_fps_indices_bit_exact_vs_reference(pu, dev, fx):
    for nm in fx["names"]:
        xyz = torch.from_numpy(fx[f"{nm}_xyz"]).to(dev)
        S = int(fx[f"{nm}_cfg"][0])
        torch.manual_seed(11)                       # same global-RNG stream as the reference's randint (:72)
        got = pu.farthest_point_sample(xyz, S)
        assert got.dtype == torch.int64
        assert np.array_equal(got.cpu().numpy(), fx[f"{nm}_fps"]), nm


@pytest.mark.parametrize("B,N,S", [(1, 1, 1), (2, 7, 7), (3, 257, 100), (2, 4096, 1024), (1, 8192, 16)])
def test_fps_edge_sizes_vs_oracle(ops, dev, B, N, S):
    rng = np.random.default_rng(N)
    xyz = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
    start = torch.from_numpy(rng.integers(0, N, size=B).astype(np.int32))
    ref = ort.farthest_point_sample(xyz, S, start)
    got = ops.fps(xyz.to(dev), S, start.to(dev))
    assert np.array_equal(got.cpu().numpy(), ref.numpy())
    # channel-first view and default start 0 (CurveNet convention)
    got0 = ops.fps(xyz.to(dev).transpose(1, 2).contiguous(), S, None, cf=True)
    assert np.array_equal(got0.cpu().numpy(), ort.farthest_point_sample(xyz, S, torch.zeros(B, dtype=torch.long)).numpy())


def test_fps_duplicate_points_lowest_index(ops, dev):
    xyz = torch.zeros(1, 10, 3)
    xyz[0, 5:] = 1.0                                # two clusters of identical points
    got = ops.fps(xyz.to(dev), 4, torch.tensor([2], dtype=torch.int32, device=dev)).cpu().numpy()
    assert got.tolist() == ort.farthest_point_sample(xyz, 4, torch.tensor([2])).numpy().tolist()


def test_ball_query_vs_reference_with_rim_band(pu, dev, fx):
    """Identical to the reference except for points within fp32 rounding of the ball surface (its expansion vs our
    direct difference); and bit-identical to the oracle evaluated in the direct form."""
    for nm in fx["names"]:
        xyz = torch.from_numpy(fx[f"{nm}_xyz"])
        S, r, ns = fx[f"{nm}_cfg"]
        S, ns, r = int(S), int(ns), float(r)
        new_xyz = ort.index_points(xyz, torch.from_numpy(fx[f"{nm}_fps"]))
        got = pu.query_ball_point(r, ns, xyz.to(dev), new_xyz.to(dev)).cpu().numpy()
        exact = ort.query_ball_point(r, ns, xyz, new_xyz, exact=True).numpy()
        assert np.array_equal(got, exact), nm
        ref = fx[f"{nm}_ball"]
        diff_rows = np.any(got != ref, axis=2)
        assert np.all(fx[f"{nm}_rim"][diff_rows] < 1e-6), nm      # only centroids with a point on the rim may differ
        assert diff_rows.mean() < 0.01


def test_ball_query_no_hit_and_padding(ops, dev):
    xyz = torch.tensor([[[0., 0, 0], [1, 0, 0], [0.05, 0, 0], [5, 5, 5]]], device=dev)
    ctr = torch.tensor([[[0., 0, 0], [9, 9, 9]]], device=dev)
    out = ops.ball_query(0.1, 4, xyz, ctr).cpu().numpy()
    assert out[0, 0].tolist() == [0, 2, 0, 0]       # two hits, padded with the first
    assert out[0, 1].tolist() == [4, 4, 4, 4]       # no hit: N, like the reference's sort leaves it


@pytest.mark.parametrize("N,S,r,ns,cf", [(4096, 1000, 0.05, 20, False), (4096, 1024, 0.05, 20, True), (1000, 70, 0.3, 64, False),
                                         (777, 1, 0.5, 96, True), (2048, 130, 0.4, 100, False), (300, 64, 2.0, 8, False)])
def test_ball_query_both_kernels_equal_the_direct_form(ops, dev, N, S, r, ns, cf):
    """The centre-per-lane kernel (nsample <= 96: quarters of the cloud per wavefront, early exit, ragged last tile of
    centres, rows concatenated from the quarters), the wavefront-per-centre kernel and the library's choice against the oracle's
    direct-difference form: same indices, order and padding; sparse (r = 0.05: most balls hold one point) and
    saturated (r = 2: every ball is full after nsample points) cases, both memory layouts."""
    g = torch.Generator().manual_seed(N + S)
    xyz = torch.rand(2, N, 3, generator=g) - 0.5
    ctr = xyz[:, torch.randperm(N, generator=g)[:S]].contiguous()
    ctr[:, -1] = 7.0                                  # a centre with no point in its ball
    exact = ort.query_ball_point(r, ns, xyz, ctr, exact=True).numpy()
    a, c = (xyz.transpose(1, 2).contiguous(), ctr.transpose(1, 2).contiguous()) if cf else (xyz, ctr)
    for kernel in (None, "wave") + (("lane",) if ns <= 96 else ()):
        got = ops.ball_query(r, ns, a.to(dev), c.to(dev), cf=cf, kernel=kernel).cpu().numpy()
        assert np.array_equal(got, exact), kernel
        assert (got[:, -1] == N).all()
    if ns > 96:
        with pytest.raises(Exception, match="centre-per-lane"):
            ops.ball_query(r, ns, a.to(dev), c.to(dev), cf=cf, kernel="lane")


def test_sample_and_group_matches_reference(pu, dev, fx):
    for nm in fx["names"]:
        xyz = torch.from_numpy(fx[f"{nm}_xyz"]).to(dev)
        feats = torch.from_numpy(fx[f"{nm}_feats"]).to(dev)
        S, r, ns = fx[f"{nm}_cfg"]
        torch.manual_seed(11)
        nx, npts = pu.sample_and_group(int(S), float(r), int(ns), xyz, feats)
        np.testing.assert_array_equal(nx.cpu().numpy(), fx[f"{nm}_sg_new_xyz"])
        ref = fx[f"{nm}_sg_new_points"]
        same = np.all(np.isclose(npts.cpu().numpy(), ref, rtol=0, atol=1e-6), axis=(2, 3))
        assert same.mean() > 0.99                      # rows with a rim point may group differently
        assert np.all(fx[f"{nm}_rim"][~same] < 1e-6)


@pytest.mark.parametrize("B,N,S,ns,D,mode", [(2, 50, 7, 5, 6, "rand"), (2, 300, 40, 16, 131, "rand"), (1, 100, 300, 16, 0, "rand"),
                                             (1, 64, 70, 33, 40, "same"), (2, 2048, 512, 32, 0, "rand"), (1, 90, 33, 8, 700, "rand")])
def test_group_gather_backward_vs_torch(ops, dev, B, N, S, ns, D, mode):
    """Forward values and the scatter-add backward (the xyz-only case, a heavily repeated index, BASELINE's SA1 size,
    a wide feature row) against torch indexing + autograd in float64."""
    torch.manual_seed(N + S)
    xyz = torch.randn(B, N, 3, device=dev, requires_grad=True)
    feat = torch.randn(B, N, D, device=dev, requires_grad=True) if D else None
    idx = torch.randint(0, N, (B, S, ns), device=dev, dtype=torch.int32)
    if mode == "same":
        idx[:, :, 3:] = 63                                   # padding-like repeats, in the last tile
    cidx = torch.randint(0, N, (B, S), device=dev, dtype=torch.int32)
    centers = torch.gather(xyz, 1, cidx.long()[:, :, None].expand(-1, -1, 3))
    out = ops.group_gather(xyz, feat, idx, centers=centers.detach(), center_idx=cidx)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    gx, gf = xyz.grad.clone(), (feat.grad.clone() if D else None)
    xyz.grad = None
    if D:
        feat.grad = None
    bi = torch.arange(B, device=dev)[:, None, None]
    parts = [xyz[bi, idx.long()] - xyz[torch.arange(B, device=dev)[:, None], cidx.long()][:, :, None, :]]
    if D:
        parts.append(feat[bi, idx.long()])
    ref = torch.cat(parts, dim=-1)
    torch.testing.assert_close(out, ref.detach())
    (ref.double() * w.double()).sum().backward()
    torch.testing.assert_close(gx, xyz.grad, rtol=1e-4, atol=1e-4)
    if D:
        torch.testing.assert_close(gf, feat.grad, rtol=1e-4, atol=1e-4)
    # float atomics: a second backward agrees to rounding (the summation order is not fixed)
    xyz.grad = None
    if D:
        feat.grad = None
    (ops.group_gather(xyz, feat, idx, centers=centers.detach(), center_idx=cidx) * w).sum().backward()
    torch.testing.assert_close(xyz.grad, gx, rtol=1e-4, atol=1e-4)


def test_group_gather_backward_features_only_and_free_centres(ops, dev):
    torch.manual_seed(4)
    B, N, S, ns, D = 2, 120, 30, 9, 24
    feat = torch.randn(B, N, D, device=dev, requires_grad=True)
    idx = torch.randint(0, N, (B, S, ns), device=dev, dtype=torch.int32)
    out = ops.group_gather(None, feat, idx)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g1 = feat.grad.clone()
    feat.grad = None
    (feat[torch.arange(B, device=dev)[:, None, None], idx.long()] * w).sum().backward()
    torch.testing.assert_close(g1, feat.grad, rtol=1e-5, atol=1e-5)
    # centres given as an independent tensor: their own gradient is minus the group sums
    xyz = torch.randn(B, N, 3, device=dev, requires_grad=True)
    ctr = torch.randn(B, S, 3, device=dev, requires_grad=True)
    out = ops.group_gather(xyz, None, idx, centers=ctr)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    torch.testing.assert_close(ctr.grad, -w.sum(dim=2), rtol=1e-5, atol=1e-5)


def _hip_model(name, dev):
    mod = importlib.import_module(f"3dpointcloudattack_amd.model.pointnet2_{name.upper()}")
    m = mod.PointNet_Ssg(40) if name == "ssg" else mod.PointNet_Msg(40, normal_channel=False)
    sd = ort.seeded_state_dict(m, 3)
    m.load_state_dict(sd)
    return m.eval().to(dev), ort.state_sha256(sd)


@pytest.mark.parametrize("name", ["ssg", "msg"])
def test_geometry_side_stream_equals_inline(dev, fx, name):
    """Sampling + grouping of both set-abstraction layers on the side stream (geometry_chain) against the same forward
    with everything on one stream: same FPS start draws, so logits and the input gradient are equal bit for bit up to
    the float atomics of the scatter kernels."""
    model, _ = _hip_model(name, dev)
    x0 = torch.from_numpy(fx[f"{name}_x"]).to(dev)
    res = []
    for side in (True, False):
        model.geometry_stream = side
        x = x0.clone().requires_grad_()
        torch.manual_seed(21)
        logp = model(x)[0]
        logp[:, 3].sum().backward()
        res.append((logp.detach().clone(), x.grad.clone()))
    model.geometry_stream = True
    assert torch.equal(res[0][0], res[1][0])
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["ssg", "msg"])
def test_classifier_logits_and_input_grad_vs_reference(dev, fx, name):
    model, sha = _hip_model(name, dev)
    assert sha == str(fx[f"{name}_sha256"])
    x = torch.from_numpy(fx[f"{name}_x"]).to(dev).requires_grad_()
    torch.manual_seed(21)
    out = model(x)
    assert len(out) == 3
    logp = out[0]
    ref = fx[f"{name}_logp"]
    # a rim point entering/leaving a group (see above) changes a max-pooled feature: compare with a loose bound and
    # demand equal predictions
    np.testing.assert_allclose(logp.detach().cpu().numpy(), ref, rtol=5e-3, atol=5e-3)
    assert np.array_equal(logp.argmax(1).cpu().numpy(), ref.argmax(1))
    (logp * torch.from_numpy(fx[f"{name}_w"]).to(dev)).sum().backward()
    got, gref = x.grad.cpu().numpy(), fx[f"{name}_gx"]
    assert np.linalg.norm(got - gref) / np.linalg.norm(gref) < 5e-2
    # strict check against the oracle run with the SAME (direct-difference) ball membership
    oc = (ort.PointNet_Ssg if name == "ssg" else ort.PointNet_Msg)(40, exact=True)
    oc.load_state_dict(ort.seeded_state_dict(oc, 3))
    oc.eval()
    xo = torch.from_numpy(fx[f"{name}_x"]).requires_grad_()
    torch.manual_seed(21)
    lo = oc(xo)[0]
    np.testing.assert_allclose(logp.detach().cpu().numpy(), lo.detach().numpy(), rtol=1e-4, atol=5e-5)
    (lo * torch.from_numpy(fx[f"{name}_w"])).sum().backward()
    go = xo.grad.numpy()
    close = np.isclose(got, go, rtol=5e-3, atol=2e-5 * np.abs(go).max())
    assert close.mean() > 0.97       # max-pool arg-max near-ties move single contributions (9 max-pools in MSG)
    assert np.linalg.norm(got - go) / np.linalg.norm(go) < 1e-2


@pytest.mark.parametrize("G,ns,C2,C3", [(7, 32, 64, 128), (3, 128, 512, 1024), (5, 16, 33, 40), (2, 1, 8, 8), (9, 64, 128, 256),
                                        (4, 128, 96, 128), (6, 16, 32, 64), (3, 50, 72, 96), (2, 1, 8, 32), (300, 32, 64, 128)])
def test_linear_relu_max_fwd_bwd_vs_autograd(dev, G, ns, C2, C3):
    """ops.linear_relu_max (GEMM epilogue + sparse backward through the max) vs relu(linear).max with autograd."""
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    g = torch.Generator().manual_seed(G * ns + C3)
    x = torch.randn(G, ns, C2, generator=g).to(dev).requires_grad_()
    w = (torch.randn(C3, C2, generator=g) / C2 ** 0.5).to(dev)
    b = torch.randn(C3, generator=g).to(dev)
    up = torch.randn(G, C3, generator=g).to(dev)
    out = ops.linear_relu_max(x, w, b)
    (out * up).sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    ref = torch.relu(torch.nn.functional.linear(x, w, b)).max(dim=1)[0]
    (ref * up).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g1, x.grad, rtol=1e-4, atol=1e-6)
    # deterministic: same bits on a second call
    x.grad = None
    (ops.linear_relu_max(x, w, b) * up).sum().backward()
    assert torch.equal(x.grad, g1)


@pytest.mark.parametrize("G,ns,dims", [(6, 32, (3, 64, 64, 128)), (4, 64, (131, 128, 128, 256)), (2, 128, (259, 256, 512, 1024)),
                                       (5, 16, (6, 32, 32, 64)), (3, 20, (9, 40))])
def test_mlp_relu_max_fwd_bwd_vs_autograd(dev, G, ns, dims):
    """ops.mlp_relu_max (GEMM epilogues + fused last layer/max + hand-written backward) vs the plain torch chain."""
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    g = torch.Generator().manual_seed(G * ns + len(dims))
    x = torch.randn(G, ns, dims[0], generator=g).to(dev).requires_grad_()
    layers = [((torch.randn(co, ci, generator=g) / ci ** 0.5).to(dev), (0.3 * torch.randn(co, generator=g)).to(dev))
              for ci, co in zip(dims[:-1], dims[1:])]
    out = ops.mlp_relu_max(x, layers)
    up = torch.randn(out.shape, generator=g).to(dev)
    (out * up).sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    h = x.double()
    for w, b in layers:
        h = torch.relu(torch.nn.functional.linear(h, w.double(), b.double()))
    ref = h.max(dim=1)[0]
    torch.testing.assert_close(out.double(), ref.detach(), rtol=2e-5, atol=2e-5)
    (ref * up.double()).sum().backward()
    torch.testing.assert_close(g1.double(), x.grad.double(), rtol=2e-3, atol=2e-4)


def test_empty_ball_and_nan_cloud_do_not_fault(dev):
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    """ADVICE r1: ball query marks "no point inside the radius" with N (what the reference's sort leaves there,
    model/pointnet2_utils.py:97-103) and a NaN cloud gives FPS no arg-max winner. The reference then dies in a device
    assert; here the indices stay in range / the gather reads such rows as zeros and sends no gradient through them."""
    g = torch.Generator().manual_seed(0)
    B, N, S, ns = 2, 200, 8, 16
    xyz = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev)
    far = torch.full((B, S, 3), 50.0, device=dev)
    far[:, 0] = xyz[:, 7]                                    # one centroid does have neighbours
    idx = ops.ball_query(0.2, ns, xyz, far)
    assert (idx[:, 1:] == N).all() and (idx[:, 0] < N).all()
    feat = torch.randn(B, N, 5, generator=g).to(dev).requires_grad_()
    x = xyz.clone().requires_grad_()
    out = ops.group_gather(x, feat, idx, centers=far)
    assert out.shape == (B, S, ns, 8) and (out[:, 1:] == 0).all() and torch.isfinite(out).all()
    out.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(x.grad).all() and torch.isfinite(feat.grad).all()
    nan_cloud = torch.full((B, N, 3), float("nan"), device=dev)
    fi = ops.fps(nan_cloud, 16, torch.tensor([3, 10_000], dtype=torch.int32, device=dev))   # 2nd start is out of range
    torch.cuda.synchronize()
    assert ((fi >= 0) & (fi < N)).all()
    nb = ops.group_gather(nan_cloud, None, ops.ball_query(0.2, ns, nan_cloud, nan_cloud[:, :S]))
    torch.cuda.synchronize()
    assert nb.shape == (B, S, ns, 3) and (nb == 0).all()     # NaN distances are never inside a ball


@pytest.mark.parametrize("B,NA,S,K,C", [(2, 50, 7, 5, 8), (3, 300, 40, 32, 64), (2, 512, 128, 64, 128), (1, 33, 33, 1, 4)])
@pytest.mark.parametrize("slope", [0.0, 0.2])
def test_group_act_fwd_bwd_vs_torch(dev, B, NA, S, K, C, slope):
    """pc3d_group_act_f32 / _bwd_f32: H = act(P[idx] + Bc), gradients to P (scatter-add, ball-query style padding with
    the group's first index merged on chip) and Bc (group sums) against plain torch indexing + autograd."""
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    g = torch.Generator().manual_seed(NA + K)
    P = torch.randn(B, NA, C, generator=g).to(dev)
    Bc = torch.randn(B, S, C, generator=g).to(dev)
    idx = torch.randint(0, NA, (B, S, K), generator=g)
    idx[:, :, K // 2:] = idx[:, :, :1]                       # padded tail: repeats of the first index, as ball query pads
    if K > 2:
        idx[0, 0, 1] = NA                                    # "no point" marker: zero row, no gradient
    idx = idx.int().to(dev)
    up = torch.randn(B, S, K, C, generator=g).to(dev)
    Pa, Ba = P.clone().requires_grad_(), Bc.clone().requires_grad_()
    H = ops.group_act(Pa, Ba, idx, slope)
    (H * up).sum().backward()
    Pr, Br = P.double().clone().requires_grad_(), Bc.double().clone().requires_grad_()
    ok = (idx < NA)
    gathered = torch.gather(Pr[:, None].expand(-1, S, -1, -1), 2, idx.clamp(max=NA - 1).long()[..., None].expand(-1, -1, -1, C))
    pre = gathered * ok[..., None] + Br[:, :, None, :]
    Hr = torch.where(pre > 0, pre, slope * pre)
    (Hr * up.double()).sum().backward()
    torch.testing.assert_close(H.double(), Hr, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(Ba.grad.double(), Br.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(Pa.grad.double(), Pr.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,NA,S,ns,C1,C2,C3", [(2, 300, 40, 32, 64, 64, 128), (3, 257, 17, 64, 128, 128, 256),
                                                (1, 90, 5, 16, 32, 48, 64), (2, 128, 9, 128, 12, 24, 32),
                                                (2, 200, 11, 128, 64, 96, 128), (2, 150, 20, 16, 32, 32, 64)])   # MSG widths
def test_grouped_mlp_max_equals_group_act_then_mlp(dev, B, NA, S, ns, C1, C2, C3):
    """pc3d_gemm_nt_gather_f32 (layer 1 generated while loading layer 2's operand) + the bit-mask backward against
    the two-operator form (pc3d_group_act_f32, then the MLP): outputs bit for bit (same GEMM, same operand values),
    gradients to P and Bc to float-atomic noise; padded tails and a "no point" index included."""
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    g = torch.Generator().manual_seed(NA + ns)
    P, Bc = torch.randn(B, NA, C1, generator=g).to(dev), torch.randn(B, S, C1, generator=g).to(dev)
    idx = torch.randint(0, NA, (B, S, ns), generator=g)
    idx[:, :, ns // 2:] = idx[:, :, :1]
    idx[0, 0, 1] = NA
    idx = idx.int().to(dev)
    layers = [((torch.randn(C2, C1, generator=g) / C1 ** 0.5).to(dev), torch.randn(C2, generator=g).to(dev)),
              ((torch.randn(C3, C2, generator=g) / C2 ** 0.5).to(dev), torch.randn(C3, generator=g).to(dev))]
    up = torch.randn(B, S, C3, generator=g).to(dev)
    assert ops.grouped_mlp_max_supported(C1, ns, layers)
    Pa, Ba = P.clone().requires_grad_(), Bc.clone().requires_grad_()
    out = ops.grouped_mlp_max(Pa, Ba, idx, layers)
    (out * up).sum().backward()
    Pr, Br = P.clone().requires_grad_(), Bc.clone().requires_grad_()
    ref = ops.mlp_relu_max(ops.group_act(Pr, Br, idx, 0.0), layers)
    (ref * up).sum().backward()
    # At these toy sizes the two-operator form's plain GEMM has few tiles; with K >= 128 it then splits the K steps over
    # groups of waves (csrc/gemm.hip, variants 11 / 12) and sums in another order than the gathering kernel. At the
    # victims' sizes (hundreds of thousands of rows) both run the same tiling and agree bit for bit.
    M = B * S * ns
    k_split = C1 >= 128 and -(-M // 128) * -(-C2 // (64 if C2 <= 64 else 128)) < 192
    if k_split:
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    else:
        assert torch.equal(out, ref)
    torch.testing.assert_close(Ba.grad, Br.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(Pa.grad, Pr.grad, rtol=1e-4, atol=1e-5)
    # the same with the backward gathering through the reverse index of the grouping (no float atomics)
    rev = ops.group_reverse(idx, NA)
    off, lst = rev[0].cpu(), rev[1].cpu()
    assert int(off[:, 0].abs().max()) == 0 and bool((off[:, 1:] >= off[:, :-1]).all())
    ic = idx.cpu()
    real = (ic < NA) & ((torch.arange(ns)[None, None, :] == 0) | (ic != ic[:, :, :1]))
    lead = (ic[:, :, 0] < NA).sum(1) if ns > 1 else torch.zeros(B, dtype=torch.long)
    assert torch.equal(off[:, -1].long(), real.sum((1, 2)) + lead)       # every real row once + one tail per led group
    for bits in (False, True):       # ... and with layer 2's sign bits kept instead of its output (ops.LAYER2_SIGN_BITS)
        ops.LAYER2_SIGN_BITS = bits
        try:
            Pg, Bg = P.clone().requires_grad_(), Bc.clone().requires_grad_()
            out_g = ops.grouped_mlp_max(Pg, Bg, idx, layers, rev=rev)
            (out_g * up).sum().backward()
        finally:
            ops.LAYER2_SIGN_BITS = True
        assert torch.equal(out_g, out)          # against the gathering kernel without the reverse index: always bit for bit
        torch.testing.assert_close(Bg.grad, Br.grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(Pg.grad, Pr.grad, rtol=1e-4, atol=1e-5)
        assert torch.isfinite(Pg.grad).all() and float(Pg.grad[0].abs().sum()) > 0


@pytest.mark.parametrize("mlp,in_feat", [([64, 64, 128], 0), ([32, 48, 64], 13), ([30, 40], 5), ([16], 0)])
def test_set_abstraction_vs_oracle_module(dev, mlp, in_feat):
    """PointNetSetAbstraction end to end (FPS + ball query + the grouped first layer as P[idx] + Bc + the rest of the MLP
    + max) against the oracle's restatement of model/pointnet2_utils.py:158-199 with the same weights and FPS start:
    values and the gradient to the input points / features. Widths the gather-with-activation kernel does not take
    (C1 % 4 != 0, one-layer MLPs) go through the plain gather + MLP form and must agree as well."""
    pu = importlib.import_module("3dpointcloudattack_amd.model.pointnet2_utils")
    B, N, S, ns = 2, 200, 24, 16
    hip = pu.PointNetSetAbstraction(S, 0.4, ns, in_feat + 3, mlp, False)
    ora = ort.PointNetSetAbstraction(S, 0.4, ns, in_feat + 3, mlp, False, exact=True)
    sd = ort.seeded_state_dict(hip, 11)
    hip.load_state_dict(sd), ora.load_state_dict(sd)
    hip, ora = hip.eval().to(dev), ora.eval()
    g = torch.Generator().manual_seed(3)
    xyz = (torch.rand(B, 3, N, generator=g) - 0.5)
    feat = torch.randn(B, in_feat, N, generator=g) if in_feat else None
    xa = xyz.clone().to(dev).requires_grad_()
    fa = feat.clone().to(dev).requires_grad_() if in_feat else None
    xo = xyz.clone().requires_grad_()
    fo = feat.clone().requires_grad_() if in_feat else None
    torch.manual_seed(5)
    nx, nf = hip(xa, fa)
    torch.manual_seed(5)
    ox, of = ora(xo, fo)
    torch.testing.assert_close(nx.cpu(), ox, rtol=0, atol=0)
    torch.testing.assert_close(nf.cpu(), of, rtol=1e-4, atol=1e-5)
    up = torch.randn(of.shape, generator=g)
    (nf * up.to(dev)).sum().backward()
    (of * up).sum().backward()
    assert (xa.grad.cpu() - xo.grad).norm() <= 2e-3 * xo.grad.norm() + 1e-8
    if in_feat:
        assert (fa.grad.cpu() - fo.grad).norm() <= 2e-3 * fo.grad.norm() + 1e-8
