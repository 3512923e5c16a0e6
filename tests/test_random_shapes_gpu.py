"""Seeded random-shape sweeps of the C-ABI operators against plain torch (float64 where cheap): the parametrised tests
pin known sizes; this file walks odd ones (primes, 1, just past a tile boundary) so that indexing and tail handling
are exercised broadly. Sizes stay small: the whole file runs in seconds."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 127, 129, 200, 257, 511, 513, 1000, 1025, 2049]


def _rng(seed):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("seed", range(12))
def test_nn_bidir_random(ops, dev, seed):
    r = _rng(seed)
    B, N, M = int(r.integers(1, 4)), int(r.choice(SIZES)), int(r.choice(SIZES))
    a = torch.from_numpy(r.standard_normal((B, N, 3)).astype(np.float32)).to(dev)
    b = torch.from_numpy(r.standard_normal((B, M, 3)).astype(np.float32)).to(dev)
    if seed % 3 == 0:          # channel-first strided views
        a = a.transpose(1, 2).contiguous().transpose(1, 2)
    dA, iA, dB, iB = ops.nn_bidir_raw(a, b, two_scan=bool(seed & 1))
    D = ((a.double()[:, :, None] - b.double()[:, None]) ** 2).sum(-1)
    rA, rB = D.min(dim=2), D.min(dim=1)
    torch.testing.assert_close(dA.double(), rA[0], rtol=2e-6, atol=1e-12)
    torch.testing.assert_close(dB.double(), rB[0], rtol=2e-6, atol=1e-12)
    torch.testing.assert_close(torch.gather(D, 2, iA.long()[..., None])[..., 0], rA[0], rtol=2e-6, atol=1e-12)
    torch.testing.assert_close(torch.gather(D, 1, iB.long()[:, None])[:, 0], rB[0], rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("seed", range(12))
def test_knn_random(ops, dev, seed):
    r = _rng(100 + seed)
    B, N, M = int(r.integers(1, 3)), int(r.choice(SIZES[:14])), int(r.choice(SIZES[2:]))
    K = int(r.integers(1, min(64, M) + 1))
    q = torch.from_numpy(r.standard_normal((B, N, 3)).astype(np.float32)).to(dev)
    x = torch.from_numpy(r.standard_normal((B, M, 3)).astype(np.float32)).to(dev)
    d, i = ops.knn_raw(q, x, K)
    D = ((q.double()[:, :, None] - x.double()[:, None]) ** 2).sum(-1)
    rd, _ = D.topk(K, dim=-1, largest=False)
    torch.testing.assert_close(d.double(), rd, rtol=2e-6, atol=1e-12)
    torch.testing.assert_close(torch.gather(D, 2, i.long()), rd, rtol=2e-6, atol=1e-12)
    assert all(len(set(row.tolist())) == K for row in i.reshape(-1, K)[:: max(1, (B * N) // 40)])


@pytest.mark.parametrize("seed", range(10))
def test_fps_ball_gather_random(ops, dev, seed):
    r = _rng(200 + seed)
    B, N = int(r.integers(1, 4)), int(r.choice(SIZES[3:]))
    S, ns, D = int(r.integers(1, N + 1)), int(r.integers(1, 40)), int(r.choice([0, 1, 7, 64]))
    x = torch.from_numpy(r.standard_normal((B, N, 3)).astype(np.float32)).to(dev)
    start = torch.from_numpy(r.integers(0, N, (B,)).astype(np.int32)).to(dev)
    i = ops.fps(x, S, start)
    assert i.shape == (B, S) and torch.equal(i[:, 0], start) and int(i.min()) >= 0 and int(i.max()) < N
    if S <= N:
        srt = i.sort(dim=1)[0]
        # distinct unless the cloud has exact duplicates (it does not: continuous random coordinates)
        assert torch.all(srt[:, 1:] != srt[:, :-1])
    ctr = torch.gather(x, 1, i.long()[..., None].expand(-1, -1, 3))
    radius = float(r.uniform(0.2, 1.5))
    g = ops.ball_query(radius, ns, x, ctr)
    assert g.shape == (B, S, ns) and int(g.min()) >= 0 and int(g.max()) < N       # every centre is one of the points
    dist = (torch.gather(x, 1, g.long().reshape(B, -1, 1).expand(-1, -1, 3)).view(B, S, ns, 3) - ctr[:, :, None]).norm(dim=-1)
    assert float(dist.max()) <= radius * (1 + 1e-5) + 1e-6
    inside = (torch.cdist(ctr.double(), x.double()) <= radius * (1 - 1e-6)).sum(-1)    # clear members only
    found = (g[..., 1:] != g[..., :1]).sum(-1) + 1
    assert torch.all(found <= ns) and torch.all(found.to(inside.dtype) <= inside.clamp(min=1) + 1)
    xr = x.clone().requires_grad_()
    feat = torch.randn(B, N, D, device=dev, requires_grad=True) if D else None
    out = ops.group_gather(xr, feat, g, centers=ctr, center_idx=i)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    bi = torch.arange(B, device=dev)[:, None, None]
    xd = x.double().requires_grad_()
    fd = feat.detach().double().requires_grad_() if D else None
    parts = [xd[bi, g.long()] - xd[torch.arange(B, device=dev)[:, None], i.long()][:, :, None]]
    if D:
        parts.append(fd[bi, g.long()])
    ref = torch.cat(parts, -1)
    torch.testing.assert_close(out.double(), ref.detach(), rtol=1e-6, atol=1e-6)
    (ref * w.double()).sum().backward()
    torch.testing.assert_close(xr.grad.double(), xd.grad, rtol=1e-4, atol=1e-4)
    if D:
        torch.testing.assert_close(feat.grad.double(), fd.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("seed", range(10))
def test_tower_random(ops, dev, seed):
    r = _rng(300 + seed)
    B, N, C3 = int(r.integers(1, 4)), int(r.choice(SIZES)), int(r.choice([32, 64, 256, 1024]))
    relu_last, use_T = bool(seed & 1), bool(seed & 2)
    g = torch.Generator().manual_seed(seed)
    w = tuple(((torch.rand(*s, generator=g) * 2 - 1) / np.sqrt(k)).to(dev) for s, k in
              (((64, 3), 3), ((64,), 3), ((128, 64), 64), ((128,), 64), ((C3, 128), 128), ((C3,), 128)))
    x = torch.from_numpy(r.standard_normal((B, 3, N)).astype(np.float32) * 0.5).to(dev)
    T = torch.from_numpy(r.standard_normal((B, 3, 3)).astype(np.float32) * 0.6).to(dev) if use_T else None
    pooled, idx, masks = ops.pointmlp3_max_fwd_raw(x, w, relu_last, T=T, want_masks=True)
    xr = x.double().requires_grad_()
    xt = xr if T is None else torch.bmm(xr.transpose(1, 2), T.double()).transpose(1, 2)
    h = F.relu(F.conv1d(xt, w[0].double()[:, :, None], w[1].double()))
    h = F.relu(F.conv1d(h, w[2].double()[:, :, None], w[3].double()))
    h = F.conv1d(h, w[4].double()[:, :, None], w[5].double())
    if relu_last:
        h = F.relu(h)
    ref = h.max(dim=2)[0]
    torch.testing.assert_close(pooled.double(), ref.detach(), rtol=3e-5, atol=3e-6)
    gup = torch.from_numpy(r.standard_normal((B, C3)).astype(np.float32)).to(dev)
    (ref * gup.double()).sum().backward()
    gp = gup * (pooled > 0) if relu_last else gup
    gx = ops.pointmlp3_max_bwd_raw(x, w + (w[2].t().contiguous(),), idx, gp, masks, T=T)
    rel = float((gx.double() - xr.grad).norm() / (xr.grad.norm() + 1e-30))
    assert rel < 5e-3, rel          # a tie at the max / a pre-activation at rounding distance from 0 moves single entries


@pytest.mark.parametrize("seed", range(12))
def test_linear_random(ops, dev, seed):
    r = _rng(400 + seed)
    B, K, O = int(r.integers(1, 70)), int(r.choice([9, 16, 40, 48, 64, 256, 512, 1024])), int(r.choice([1, 9, 40, 63, 64, 65, 256, 1000]))
    parts = int(r.choice([1, 1, 3, 8]))
    X = torch.from_numpy(r.standard_normal((B, parts, K) if parts > 1 else (B, K)).astype(np.float32)).to(dev)
    W = torch.from_numpy((r.standard_normal((O, K)) / np.sqrt(K)).astype(np.float32)).to(dev)
    b = torch.from_numpy(r.standard_normal(O).astype(np.float32)).to(dev) if seed & 1 else None
    gate = torch.from_numpy(r.standard_normal((B, O)).astype(np.float32)).to(dev) if seed & 2 else None
    relu = bool(seed & 4)
    y = ops.linear(X, W, b, relu=relu, gate=gate, parts=parts)
    Xs = X.double().sum(1) if parts > 1 else X.double()
    ref = Xs @ W.double().t() + (b.double() if b is not None else 0)
    if relu:
        ref = ref.clamp(min=0)
    if gate is not None:
        ref = torch.where(gate > 0, ref, torch.zeros_like(ref))
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("seed", range(8))
def test_dgcnn_and_sa_ops_random(ops, dev, seed):
    r = _rng(500 + seed)
    B, N, C, K = int(r.integers(1, 3)), int(r.choice(SIZES[3:12])), int(r.choice([4, 8, 64, 132])), int(r.integers(1, 21))
    K = min(K, N)
    PQ = torch.from_numpy(r.standard_normal((B, N, 2 * C)).astype(np.float32)).to(dev).requires_grad_()
    idx = torch.from_numpy(r.integers(0, N, (B, N, K)).astype(np.int32)).to(dev)
    out = ops.edge_max(PQ, idx, 0.2)
    nb = torch.gather(PQ[..., :C].unsqueeze(1).expand(-1, N, -1, -1), 2, idx.long()[..., None].expand(-1, -1, -1, C))
    ref = F.leaky_relu(nb.max(dim=2)[0] + PQ[..., C:], 0.2)
    torch.testing.assert_close(out, ref.detach())
    Y = torch.from_numpy(r.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    z = F.leaky_relu(Y, 0.2)
    torch.testing.assert_close(ops.act_maxmean_pool(Y, 0.2), torch.cat((z.max(1)[0], z.mean(1)), 1), rtol=1e-5, atol=1e-6)
    G, ns, C2, C3 = int(r.integers(1, 9)), int(r.choice([1, 16, 33, 128])), int(r.choice([3, 64, 131])), int(r.choice([8, 40, 256]))
    x = torch.from_numpy(r.standard_normal((G, ns, C2)).astype(np.float32)).to(dev).requires_grad_()
    w = torch.from_numpy((r.standard_normal((C3, C2)) / np.sqrt(C2)).astype(np.float32)).to(dev)
    bb = torch.from_numpy(r.standard_normal(C3).astype(np.float32)).to(dev)
    o = ops.linear_relu_max(x, w, bb)
    up = torch.randn_like(o)
    (o * up).sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    ro = torch.relu(F.linear(x, w, bb)).max(dim=1)[0]
    torch.testing.assert_close(o, ro.detach(), rtol=1e-5, atol=1e-6)
    (ro * up).sum().backward()
    torch.testing.assert_close(g1, x.grad, rtol=1e-4, atol=1e-5)
