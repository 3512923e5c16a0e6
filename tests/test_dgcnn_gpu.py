"""GPU parity: DGCNN dynamic-graph kernels (feature-space kNN on MFMA, neighbour gather-max) and the DGCNN mirror
(EdgeConv as P_j + Q_i) vs the reference's golden outputs."""
import importlib
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ref_torch as ort

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "dgcnn.npz"))


def _knn_sets_ok(x_cn, idx, ref_idx, k):
    """Neighbour lists must agree with the reference except where the k-th and (k+1)-th distances are within fp32
    rounding of each other (the reference ranks the rounded expansion -xx - inner - xx^T)."""
    B, C, N = x_cn.shape
    xt = np.transpose(x_cn, (0, 2, 1)).astype(np.float64)
    bad = 0
    for b in range(B):
        D = ((xt[b][:, None, :] - xt[b][None, :, :]) ** 2).sum(-1)
        for i in range(N):
            got, ref = set(idx[b, i].tolist()), set(ref_idx[b, i].tolist())
            if got != ref:
                kth = np.sort(D[i])[k - 1]
                scale = max(1e-6, (xt[b][i] ** 2).sum() + 1.0)
                for j in got ^ ref:
                    assert abs(D[i, j] - kth) < 2e-5 * scale * C, (b, i, j, D[i, j], kth)
                bad += 1
            assert idx[b, i, 0] == i or D[i, idx[b, i, 0]] <= 1e-6       # self (distance 0) first
    return bad


def test_knn_feat_matches_reference_topk(ops, dev, fx):
    feat = fx["feat"]                                           # [B,64,N]
    idx = ops.knn_feat(torch.from_numpy(feat).transpose(1, 2).contiguous().to(dev), 20).cpu().numpy()
    bad = _knn_sets_ok(feat, idx, fx["feat_knn"], 20)
    assert bad <= 0.01 * feat.shape[0] * feat.shape[2]
    # exact check of the ordering against float64 distances
    xt = np.transpose(feat, (0, 2, 1)).astype(np.float64)
    D = ((xt[:, :, None, :] - xt[:, None, :, :]) ** 2).sum(-1)
    picked = np.take_along_axis(D, idx.astype(np.int64), axis=2)
    assert np.all(np.diff(picked, axis=2) >= -1e-4)
    np.testing.assert_allclose(picked, np.sort(D, axis=2)[:, :, :20], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,N,C,K", [(1, 40, 8, 5), (2, 1024, 128, 20), (3, 333, 64, 20), (1, 20, 64, 20), (1, 2500, 64, 20), (1, 200, 72, 64)])
def test_knn_feat_sizes(ops, dev, B, N, C, K):
    torch.manual_seed(N + C)
    x = torch.randn(B, N, C)
    idx = ops.knn_feat(x.to(dev), K).cpu().numpy()
    D = (torch.cdist(x.double(), x.double(), compute_mode='donot_use_mm_for_euclid_dist') ** 2).numpy()
    picked = np.take_along_axis(D, idx.astype(np.int64), axis=2)
    np.testing.assert_allclose(picked, np.sort(D, axis=2)[:, :, :K], rtol=1e-4, atol=1e-3)
    assert all(len(set(r)) == K for r in idx.reshape(-1, K))


@pytest.mark.parametrize("C", [96, 64, 30, 256])      # 16-byte lanes (C % 4 == 0) and the scalar kernel
def test_gather_max_fwd_bwd_vs_torch(ops, dev, C):
    torch.manual_seed(1)
    B, N, K = 2, 70, 9
    P = torch.randn(B, N, C, device=dev, requires_grad=True)
    idx = torch.randint(0, N, (B, N, K), device=dev, dtype=torch.int32)
    out = ops.gather_max(P, idx)
    ref = torch.gather(P.unsqueeze(1).expand(-1, N, -1, -1), 2, idx.long()[..., None].expand(-1, -1, -1, C)).max(dim=2)[0]
    torch.testing.assert_close(out, ref.detach())
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g1 = P.grad.clone()
    P.grad = None
    (ref * w).sum().backward()
    torch.testing.assert_close(g1, P.grad, rtol=1e-5, atol=1e-6)


def test_get_graph_feature_matches_reference(dev, fx):
    dg = importlib.import_module("3dpointcloudattack_amd.model.dgcnn")
    x = torch.from_numpy(fx["x"][:, :, :64].copy()).to(dev)
    got = dg.get_graph_feature(x, k=8).cpu().numpy()
    ref = fx["graph_feature"]
    assert got.shape == ref.shape
    same = np.isclose(got, ref, atol=1e-6).all(axis=(1, 3))      # per (b, point): neighbour order may differ on ties
    assert same.mean() > 0.97


def test_dgcnn_logits_and_input_grad_vs_reference(dev, fx):
    dg = importlib.import_module("3dpointcloudattack_amd.model.dgcnn")
    m = dg.DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), 40)
    sd = ort.seeded_state_dict(m, 5)
    m.load_state_dict(sd)
    assert ort.state_sha256(sd) == str(fx["sha256"])
    m = m.eval().to(dev)
    x = torch.from_numpy(fx["x"]).to(dev).requires_grad_()
    out = m(x)
    logp = out[0]
    ref = fx["logp"]
    np.testing.assert_allclose(logp.detach().cpu().numpy(), ref, rtol=2e-3, atol=2e-3)
    assert np.array_equal(logp.argmax(1).cpu().numpy(), ref.argmax(1))
    (logp * torch.from_numpy(fx["w"]).to(dev)).sum().backward()
    got, gref = x.grad.cpu().numpy(), fx["gx"]
    assert np.linalg.norm(got - gref) / np.linalg.norm(gref) < 3e-2
    # neighbour lists of the first (xyz) layer vs the reference
    idx = dg.knn(torch.from_numpy(fx["x"]).to(dev), 20).cpu().numpy()
    assert idx.dtype == np.int64
    assert _knn_sets_ok(fx["x"], idx, fx["xyz_knn"], 20) <= 0.01 * idx.shape[0] * idx.shape[1]


@pytest.mark.parametrize("B,N", [(2, 256), (3, 1024)])
def test_trunk_as_one_autograd_node_equals_layer_by_layer(dev, B, N):
    """DGCNN's four EdgeConv layers + conv5 + pooling as ONE autograd node (no torch.cat, no autograd additions: the EdgeConv
    launches write their slice of conv5's input, the backward scatter adds an output's two gradients on load) give the
    logits and the input gradient of the layer-by-layer path bit for bit — the same launches on the same values, and the one
    fp32 add per element that autograd's accumulation performed."""
    dg = importlib.import_module("3dpointcloudattack_amd.model.dgcnn")
    m = dg.DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), 40)
    m.load_state_dict(ort.seeded_state_dict(m, 7))
    m = m.eval().to(dev)
    torch.manual_seed(N)
    x0 = torch.nn.functional.normalize(torch.randn(B, 3, N), dim=1).to(dev)
    w = torch.randn(B, 40, device=dev)
    res = {}
    for one in (False, True):
        dg.TRUNK_AS_ONE_FUNCTION = one
        try:
            x = x0.clone().requires_grad_()
            logp = m(x)[0]
            (logp * w).sum().backward()
            res[one] = (logp.detach().clone(), x.grad.clone())
        finally:
            dg.TRUNK_AS_ONE_FUNCTION = True
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


def test_edge_max_cat_and_backward_sum_entries(ops, dev):
    """pc3d_edge_max_cat_f32 writes the same values into a column slice of a wider buffer (and nothing else in it);
    pc3d_edge_max_bwd_sum_f32 (g, g2) == pc3d_edge_max_bwd_f32 (g + g2), deterministic form, with g read through a row stride."""
    lib = importlib.import_module("3dpointcloudattack_amd._lib")
    torch.manual_seed(3)
    B, N, C, K = 3, 512, 64, 20
    PQ = torch.randn(B, N, 2 * C, device=dev)
    idx = torch.randint(0, N, (B, N, K), device=dev, dtype=torch.int32)
    out0, arg0 = ops.edge_max_raw(PQ, idx, 0.2)
    cat = torch.full((B, N, 320), -7.0, device=dev)
    out1, arg1 = ops.edge_max_raw(PQ, idx, 0.2, cat, 128)
    assert torch.equal(out0, out1) and torch.equal(arg0, arg1) and torch.equal(cat[:, :, 128:192], out0)
    assert bool((cat[:, :, :128] == -7.0).all()) and bool((cat[:, :, 192:] == -7.0).all())
    gwide = torch.randn(B * N, 320, device=dev)
    g2 = torch.randn(B * N, C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.empty(B, N, 2 * C, device=dev)
    lib.call("pc3d_edge_max_bwd_sum_f32", gwide.data_ptr() + 4 * 128, 320, g2.data_ptr(), C, out0.data_ptr(), arg0.data_ptr(), B, N, C,
             0.2, a.data_ptr(), st)
    gsum = (gwide[:, 128:192] + g2).contiguous()
    b = torch.empty(B, N, 2 * C, device=dev)
    lib.call("pc3d_edge_max_bwd_f32", gsum.data_ptr(), C, out0.data_ptr(), arg0.data_ptr(), B, N, C, 0.2, b.data_ptr(), 1, st)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,N,C,ld,ld2", [(2, 333, 12, 12, 12), (1, 90, 6, 8, 7), (3, 1024, 256, 512, 256), (2, 4100, 32, 40, 32), (1, 5000, 8, 8, 12)])
def test_edge_max_backward_sum_shapes(dev, B, N, C, ld, ld2):
    """pc3d_edge_max_bwd_sum_f32 on ragged shapes: unaligned / odd row strides (scalar loads), C % 4 != 0, more points than the
    fp64 tiles hold (N > 4096: fp32 tiles, row tiles), against a float64 scatter of (g + g2) * mask; twice bit-equal."""
    lib = importlib.import_module("3dpointcloudattack_amd._lib")
    torch.manual_seed(N + C)
    out = torch.randn(B, N, C, device=dev)
    arg = torch.randint(0, N, (B, N, C), device=dev, dtype=torch.int32)
    g = torch.randn(B * N, ld, device=dev)
    g2 = torch.randn(B * N, ld2, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for _ in range(2):
        gPQ = torch.empty(B, N, 2 * C, device=dev)
        lib.call("pc3d_edge_max_bwd_sum_f32", g.data_ptr(), ld, g2.data_ptr(), ld2, out.data_ptr(), arg.data_ptr(), B, N, C, 0.2,
                 gPQ.data_ptr(), st)
        res.append(gPQ)
    assert torch.equal(res[0], res[1])
    w = ((g[:, :C] + g2[:, :C]).view(B, N, C) * torch.where(out > 0, 1.0, 0.2)).double()
    dP = torch.zeros(B, N, C, dtype=torch.float64, device=dev)
    dP.scatter_add_(1, arg.long(), w)
    torch.testing.assert_close(res[0][:, :, :C].double(), dP, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[0][:, :, C:].double(), w, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("C,N", [(64, 90), (256, 90), (12, 90), (64, 1024), (32, 2500), (8, 5000), (4, 33)])
def test_edge_max_fwd_bwd_vs_torch(ops, dev, C, N):
    """ops.edge_max = leaky(max_j P_j + Q_i) on [P | Q] rows, forward and backward, vs torch gather/max/leaky. The sizes
    walk the three backward kernels: 8 / 4 channels per workgroup accumulated in LDS, global atomics beyond 64 KB."""
    torch.manual_seed(C + N)
    B, K = 2, 7
    PQ = torch.randn(B, N, 2 * C, device=dev, requires_grad=True)
    idx = torch.randint(0, N, (B, N, K), device=dev, dtype=torch.int32)
    out = ops.edge_max(PQ, idx, 0.2)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g1 = PQ.grad.clone()
    PQ.grad = None
    P, Q = PQ[..., :C], PQ[..., C:]
    nb = torch.gather(P.unsqueeze(1).expand(-1, N, -1, -1), 2, idx.long()[..., None].expand(-1, -1, -1, C))
    ref = torch.nn.functional.leaky_relu(nb.max(dim=2)[0] + Q, 0.2)
    torch.testing.assert_close(out, ref.detach())
    (ref * w).sum().backward()
    torch.testing.assert_close(g1, PQ.grad, rtol=1e-5, atol=1e-6)
    # the upstream gradient as a column slice of a wider tensor (what torch.cat's backward hands over): read in place
    # through its row stride (C % 4 == 0) or copied (otherwise) — same result either way
    PQ.grad = None
    wide = torch.zeros(B, N, C + 24, device=dev)
    wide[..., 8:8 + C] = w
    (ops.edge_max(PQ, idx, 0.2) * 1.0).backward(wide[..., 8:8 + C])
    torch.testing.assert_close(PQ.grad, g1, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,N,C,slope", [(2, 100, 64, 0.2), (3, 1024, 1024, 0.2), (1, 3, 8, 0.0), (2, 257, 132, 0.0)])
def test_act_maxmean_pool_fwd_bwd_vs_torch(ops, dev, B, N, C, slope):
    torch.manual_seed(N + C)
    Y = torch.randn(B, N, C, device=dev, requires_grad=True)
    out = ops.act_maxmean_pool(Y, slope)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g1 = Y.grad.clone()
    Y.grad = None
    z = torch.nn.functional.leaky_relu(Y, slope)
    ref = torch.cat((z.max(dim=1)[0], z.mean(dim=1)), 1)
    torch.testing.assert_close(out, ref.detach(), rtol=1e-5, atol=1e-6)
    (ref * w).sum().backward()
    torch.testing.assert_close(g1, Y.grad, rtol=1e-5, atol=1e-7)
    assert torch.equal(ops.act_maxmean_pool(Y, slope), out)          # deterministic


@pytest.mark.parametrize("C", [64, 72])          # the static product loop (C = 64) and the run-time one
def test_knn_feat_nan_rows_are_never_neighbours(ops, dev, C):
    """A NaN feature row gives NaN distances; as keys they rank as +inf (knn_list.h), so no finite query lists the row
    and the kernel returns valid indices for the NaN query itself."""
    torch.manual_seed(C)
    B, N, K = 2, 300, 20
    x = torch.randn(B, N, C)
    x[0, 17] = float("nan")
    x[1, 255, 3] = float("nan")
    idx = ops.knn_feat(x.to(dev), K).cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < N
    ok0 = torch.ones(N, dtype=torch.bool); ok0[17] = False
    ok1 = torch.ones(N, dtype=torch.bool); ok1[255] = False
    assert not bool((idx[0][ok0] == 17).any()) and not bool((idx[1][ok1] == 255).any())
    # the finite queries still get their exact neighbours among the finite rows
    D = torch.cdist(x[0][ok0].double(), x[0][ok0].double(), compute_mode='donot_use_mm_for_euclid_dist') ** 2
    remap = torch.full((N,), -1, dtype=torch.long); remap[ok0] = torch.arange(int(ok0.sum()))
    picked = torch.gather(D, 1, remap[idx[0][ok0].long()])
    torch.testing.assert_close(picked, torch.sort(D, dim=1)[0][:, :K], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,N,K,C,slope", [(2, 384, 64, 128, 0.2), (3, 1024, 512, 1024, 0.2), (2, 128, 40, 36, 0.0),
                                           (1, 77, 40, 36, 0.0)])     # the last one: two-operator fallback (N % 128 != 0)
def test_linear_act_maxmean_pool_backward_in_one_gemm(ops, dev, B, N, K, C, slope):
    """conv + LeakyReLU + [max | mean] pooling with the pooling's backward generated inside the layer's backward GEMM
    (pc3d_gemm_nt_poolbwd_f32) against the two-operator form and a float64 torch formulation (model/dgcnn.py:317-320)."""
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(B, N, K, generator=g).to(dev)
    w, b = (torch.randn(C, K, generator=g) / K ** 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    up = torch.randn(B, 2 * C, generator=g).to(dev)
    xa = x.clone().requires_grad_()
    out = ops.linear_act_maxmean_pool(xa, w, b, slope)
    (out * up).sum().backward()
    xb = x.clone().requires_grad_()
    ref = ops.act_maxmean_pool(ops.linear_act(xb, w, b), slope)
    (ref * up).sum().backward()
    assert torch.equal(out, ref)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-5)
    xd = x.double().requires_grad_()
    z = torch.nn.functional.leaky_relu(xd @ w.double().t() + b.double(), slope)
    r64 = torch.cat([z.max(dim=1)[0], z.mean(dim=1)], dim=1)
    (r64 * up.double()).sum().backward()
    torch.testing.assert_close(out.double(), r64.detach(), rtol=1e-5, atol=1e-5)
    assert float((xa.grad.double() - xd.grad).norm() / xd.grad.norm()) < 1e-4
