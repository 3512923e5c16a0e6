"""GPU parity: clip / projection, fused Adam(+clip), dense pairwise and K-NN kernels vs the oracle / torch."""
import importlib

import numpy as np
import pytest
import torch

from helpers import unit_cloud
from oracle import ref_numpy as orc
from oracle import ref_torch as ort

pytestmark = pytest.mark.gpu


def _rand(dev, B=3, K=500, seed=0):
    g = torch.Generator().manual_seed(seed)
    ori = torch.randn(B, 3, K, generator=g)
    pc = ori + 0.2 * torch.randn(B, 3, K, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(B, 3, K, generator=g), dim=1)
    return pc, ori, nrm


def test_clip_functors_match_oracle(dev):
    clip = importlib.import_module("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    pc, ori, nrm = _rand(dev)
    nrm[:, :, :7] = (pc - ori)[:, :, :7] * -3.0   # exactly opposite to the normal -> the "opposite" branch
    cases = [(clip.ClipPointsLinf(0.18), ort.ClipPointsLinf(0.18), False),
             (clip.ClipPointsL2(1.5), ort.ClipPointsL2(1.5), False),
             (clip.ProjectInnerPoints(), ort.ProjectInnerPoints(), True),
             (clip.ProjectInnerClipLinf(0.18), ort.ProjectInnerClipLinf(0.18), True)]
    for hip, ora, with_n in cases:
        args_d = (pc.to(dev), ori.to(dev)) + ((nrm.to(dev),) if with_n else ())
        args_c = (pc, ori) + ((nrm,) if with_n else ())
        got = hip(*args_d).cpu()
        torch.testing.assert_close(got, ora(*args_c), rtol=1e-5, atol=1e-6)
    # no normal -> ProjectInnerPoints returns its input untouched (clip_utils.py:76-77)
    x = pc.to(dev)
    assert clip.ProjectInnerPoints()(x, ori.to(dev)) is x


def test_fused_adam_clip_matches_torch_adam(ops, dev):
    pc, ori, _ = _rand(dev, B=2, K=300, seed=3)
    p_ref = pc.clone().requires_grad_()
    opt = torch.optim.Adam([p_ref], lr=1e-2, weight_decay=0.)
    p = pc.clone().to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(9)
    clipper = ort.ClipPointsLinf(0.18)
    for t in range(1, 8):
        grad = torch.randn(pc.shape, generator=g) * (10.0 ** (-t % 3))
        opt.zero_grad()
        p_ref.grad = grad.clone()
        opt.step()
        p_ref.data = clipper(p_ref.data.clone(), ori)
        ops.i32_add(step_dev, 1)
        ops.adam_clip_step(p, grad.to(dev), m, v, step_dev if t % 2 else t, 1e-2, ori=ori.to(dev), budget=0.18)
        torch.testing.assert_close(p.cpu(), p_ref.data, rtol=2e-6, atol=2e-7)
    st = opt.state[p_ref]
    torch.testing.assert_close(m.cpu(), st["exp_avg"], rtol=1e-5, atol=1e-5)  # |grad| up to ~1e2: fp32 rounding
    torch.testing.assert_close(v.cpu(), st["exp_avg_sq"], rtol=1e-5, atol=1e-6)


def test_pairwise_matches_oracle(ops, dev, metrics_fx):
    fx = metrics_fx
    a, b = fx["np_rand_64_a"], fx["np_rand_64_b"]
    M = ops.pairwise(torch.from_numpy(a)[None].to(dev), torch.from_numpy(b)[None].to(dev), euclid=True)[0]
    np.testing.assert_allclose(M.cpu().numpy(), fx["np_rand_64_M"], rtol=2e-6, atol=1e-7)
    x = torch.randn(2, 37, 3, device=dev)
    y = torch.randn(2, 53, 3, device=dev)   # M % 4 != 0 -> scalar store tail
    P = ops.pairwise(x, y)
    ref = ((x.cpu().double()[:, :, None, :] - y.cpu().double()[:, None, :, :]) ** 2).sum(-1)
    np.testing.assert_allclose(P.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("B,N,M,K", [(2, 100, 100, 6), (1, 64, 300, 17), (2, 1024, 1024, 21), (1, 50, 40, 31),
                                     (1, 5000, 70, 1), (2, 33, 4500, 4), (1, 70, 64, 64), (2, 130, 2113, 40),
                                     (1, 20, 4100, 9)])
def test_knn_matches_exact_topk(ops, dev, B, N, M, K):
    rng = np.random.default_rng(N + M + K)
    r = np.stack([unit_cloud(rng, M) for _ in range(B)])
    q = r.copy() if N == M else np.stack([unit_cloud(rng, N) for _ in range(B)])
    d, i = ops.knn_raw(torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev), K)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    D = ((q.astype(np.float64)[:, :, None, :] - r.astype(np.float64)[:, None, :, :]) ** 2).sum(-1)
    ref = np.sort(D, axis=2)[:, :, :K]
    np.testing.assert_allclose(d, ref, rtol=2e-6, atol=1e-12)
    assert np.all(np.diff(d, axis=2) >= 0)
    np.testing.assert_allclose(np.take_along_axis(D, i.astype(np.int64), axis=2), ref, rtol=2e-6, atol=1e-12)
    for bb in range(B):                         # K distinct neighbours per query
        assert all(len(set(row)) == K for row in i[bb][:: max(1, N // 50)])
    if N == M:
        assert np.array_equal(i[:, :, 0], np.broadcast_to(np.arange(N), (B, N)))


def test_knn_ties_prefer_lower_index(ops, dev):
    """Duplicated reference points (equal distances): the lower index comes first, within a step, across steps and
    across LDS tiles — the order torch.topk / the reference's sort gives on its distance matrix."""
    rng = np.random.default_rng(5)
    base = unit_cloud(rng, 700)
    r = np.concatenate([base, base, base, base])[None]            # 2800 points: 4 copies, crosses the 2048-point tile
    q = base[None, :90]
    d, i = ops.knn_raw(torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev), 12)
    i = i.cpu().numpy()[0]
    D = ((q[0].astype(np.float64)[:, None] - r[0].astype(np.float64)[None]) ** 2).sum(-1)
    ref = np.lexsort((np.broadcast_to(np.arange(2800), D.shape), D), axis=1)[:, :12]
    # float32 distances of exact duplicates are bit-equal, so the groups of 4 must appear as j, j+700, j+1400, j+2100
    assert np.array_equal(i[:, :4], np.arange(90)[:, None] + 700 * np.arange(4)[None])
    assert np.array_equal(i % 700, ref % 700)
    for row in i:
        for g in range(3):
            assert np.all(np.diff(row[4 * g:4 * g + 4]) == 700)


def _knn_abi(dev, q, r, K, hint=None, alias=False, graph=False):
    """pc3d_knn_f32 / pc3d_knn_hint_f32 (or the graph entries) straight through the C ABI on device tensors q [B,N,3], r [B,M,3]."""
    lib = importlib.import_module("3dpointcloudattack_amd._lib")
    B, N, M = q.shape[0], q.shape[1], r.shape[1]
    d = torch.empty(B, N, K, device=dev)
    i = torch.empty(B, N, K, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    if alias:
        i.copy_(hint)
        hint = i
    if graph:
        ns, fi = torch.empty(B, N, K - 1, dtype=torch.int32, device=dev), torch.empty(B, N, K - 1, dtype=torch.int32, device=dev)
        if hint is None:
            lib.call("pc3d_knn_graph_i32", q.data_ptr(), *q.stride(), B, N, K, i.data_ptr(), ns.data_ptr(), fi.data_ptr(), K - 1, st)
        else:
            lib.call("pc3d_knn_graph_hint_i32", q.data_ptr(), *q.stride(), B, N, K, i.data_ptr(), ns.data_ptr(), fi.data_ptr(), K - 1,
                     hint.data_ptr(), st)
        return i, ns, fi
    if hint is None:
        lib.call("pc3d_knn_f32", q.data_ptr(), *q.stride(), r.data_ptr(), *r.stride(), B, N, M, K, d.data_ptr(), i.data_ptr(), st)
    else:
        lib.call("pc3d_knn_hint_f32", q.data_ptr(), *q.stride(), r.data_ptr(), *r.stride(), B, N, M, K, d.data_ptr(), i.data_ptr(),
                 hint.data_ptr(), st)
    return d, i


@pytest.mark.parametrize("B,N,M,K", [(2, 100, 100, 6), (2, 1024, 1024, 21), (1, 333, 4500, 20), (1, 70, 64, 64), (2, 130, 2113, 2),
                                     (1, 4096, 4096, 21), (3, 50, 700, 17), (1, 40, 40, 1)])
def test_knn_hint_never_changes_the_result(dev, B, N, M, K):
    """pc3d_knn_hint_f32: distances and indices are those of pc3d_knn_f32 bit for bit whatever the hint is — last iteration's
    neighbours (the intended use: points moved by 1e-2), the exact answer, the output buffer itself, random indices, rows with
    repeated / out-of-range / negative entries (detected per wavefront: unhinted scan), and a hint whose bound is far too wide."""
    rng = np.random.default_rng(N + M + K)
    r = torch.from_numpy(np.stack([unit_cloud(rng, M) for _ in range(B)])).to(dev)
    q = r.clone() if N == M else torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).to(dev)
    d0, i0 = _knn_abi(dev, q, r, K)
    moved_r = r + 1e-2 * torch.randn_like(r)
    _, prev = _knn_abi(dev, q if N != M else moved_r, moved_r, K)        # "last iteration"
    g = torch.Generator(device="cpu").manual_seed(1)
    rand = torch.randint(0, M, (B, N, K), generator=g, dtype=torch.int32).to(dev)       # repeats inside rows are likely
    wild = torch.randint(-5, M + 5, (B, N, K), generator=g, dtype=torch.int32).to(dev)
    far = torch.argsort(torch.rand(B, N, M, generator=g), dim=2)[:, :, :K].to(torch.int32).to(dev)      # distinct, arbitrary
    dup = i0.clone(); dup[:, ::3, K - 1] = dup[:, ::3, 0]                                 # one repeated entry in every third row
    for name, hint, alias in (("previous", prev, False), ("exact", i0, False), ("alias", prev, True), ("random", rand, False),
                              ("wild", wild, False), ("distinct-far", far, False), ("repeat", dup, False), ("alias-wild", wild, True)):
        if K == 1 and name == "repeat":
            continue
        d1, i1 = _knn_abi(dev, q, r, K, hint=hint, alias=alias)
        assert torch.equal(i1, i0), name
        assert torch.equal(d1.view(torch.int32), d0.view(torch.int32)), name


def test_knn_hint_keeps_the_tie_order(dev):
    """Duplicated points under a hint: the list starts empty and the equal keys at the bound are inserted in ascending index
    order, so the lower index still comes first (the hint below lists the HIGHER copies first)."""
    rng = np.random.default_rng(9)
    base = unit_cloud(rng, 700)
    r = torch.from_numpy(np.concatenate([base, base, base, base])[None]).to(dev)
    q = torch.from_numpy(base[None, :90]).to(dev)
    d0, i0 = _knn_abi(dev, q, r, 10)
    assert np.array_equal(i0.cpu().numpy()[0, :, :4], np.arange(90)[:, None] + 700 * np.arange(4)[None])
    rev = torch.flip(i0, dims=[2]).contiguous()
    shifted = ((i0 + 700) % 2800).contiguous()          # same distances through other copies
    for hint in (rev, shifted):
        d1, i1 = _knn_abi(dev, q, r, 10, hint=hint)
        assert torch.equal(i1, i0) and torch.equal(d1, d0)


def test_knn_graph_hint_views_and_replay(ops, dev):
    """pc3d_knn_graph_hint_i32 writes the graph and both views exactly as the unhinted entry; ops.knn_graph hints from the
    last call of the same shape and, inside a hipGraph, from its own output buffer: replays on moving points stay exact."""
    rng = np.random.default_rng(3)
    x = torch.from_numpy(np.stack([unit_cloud(rng, 1024) for _ in range(2)])).to(dev)
    ref = _knn_abi(dev, x, x, 21, graph=True)
    got = _knn_abi(dev, x, x, 21, hint=ref[0], graph=True)
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    buf = x.clone()
    for _ in range(2):
        ops.knn_graph(buf, 20)                         # eager warm-up (also fills the hint cache)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            out = ops.knn_graph(buf, 20)
    for step in range(4):
        buf.add_(1e-2 * torch.randn_like(buf))
        g.replay()
        torch.cuda.synchronize()
        want = _knn_abi(dev, buf, buf, 21, graph=True)
        assert all(torch.equal(a, b) for a, b in zip(want, out)), step
        eager = ops.knn_graph(buf, 20)
        assert all(torch.equal(a, b) for a, b in zip(want, eager)), step


@pytest.mark.parametrize("K", [5, 20])            # the selection-round seed (K <= 10) and the sorted seed
def test_knn_nan_points_are_never_neighbours(ops, dev, K):
    """A reference point with a NaN coordinate has NaN distances; their keys rank above +inf (knn_list.h), so it is in
    no finite query's list, whether it falls in the seed step (index < 64) or a later one; K = 1 goes through the same
    list code (candidate step without the re-filter)."""
    rng = np.random.default_rng(K)
    r = unit_cloud(rng, 500)[None].copy()
    r[0, 7, 1] = np.nan
    r[0, 300, 0] = np.nan
    q = unit_cloud(rng, 120)[None]
    d, i = ops.knn_raw(torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev), K)
    d, i = d.cpu().numpy()[0], i.cpu().numpy()[0]
    assert np.isfinite(d).all() and not np.isin(i, (7, 300)).any()
    ok = np.ones(500, dtype=bool); ok[[7, 300]] = False
    D = ((q[0].astype(np.float64)[:, None] - r[0][ok].astype(np.float64)[None]) ** 2).sum(-1)
    np.testing.assert_allclose(d, np.sort(D, axis=1)[:, :K], rtol=2e-6, atol=1e-12)
    # K = 1 straight through the C ABI (ops.knn_raw routes K = 1 to the nearest-neighbour kernel instead)
    lib = importlib.import_module("3dpointcloudattack_amd._lib")
    qd, rd = torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev)
    d1 = torch.empty(1, 120, 1, device=dev)
    i1 = torch.empty(1, 120, 1, dtype=torch.int32, device=dev)
    lib.call("pc3d_knn_f32", qd.data_ptr(), *qd.stride(), rd.data_ptr(), *rd.stride(), 1, 120, 500, 1, d1.data_ptr(),
             i1.data_ptr(), torch.cuda.current_stream().cuda_stream)
    np.testing.assert_allclose(d1.cpu().numpy()[0, :, 0], np.sort(D, axis=1)[:, 0], rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("det", [False, True])
def test_knn_dist_functor_value_and_grad(dev, det):
    dist = importlib.import_module("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    rng = np.random.default_rng(1)
    pc = np.stack([unit_cloud(rng, 400) for _ in range(3)])
    w = torch.tensor([1.0, 2.0, 0.5])
    x = torch.from_numpy(pc).double().requires_grad_()       # float64 oracle (its expansion is fp32-noisy)
    ref = ort.KNNDist(5, 1.05)(x, w.double(), batch_avg=False)
    ref.sum().backward()
    xd = torch.from_numpy(pc).to(dev).requires_grad_()
    got = dist.KNNDist(5, 1.05)(xd, w, batch_avg=False)
    got.sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), x.grad.numpy(), rtol=1e-3, atol=1e-7)
    # channel-first input gives the same value
    got_cf = dist.KNNDist(5, 1.05)(torch.from_numpy(pc).to(dev).transpose(1, 2).contiguous(), w, batch_avg=False)
    torch.testing.assert_close(got_cf, got.detach(), rtol=1e-6, atol=0)


@pytest.mark.parametrize("B,K,O", [(32, 1024, 512), (5, 512, 256), (32, 256, 40), (3, 256, 9), (40, 40, 256), (2, 16, 256)])
def test_linear_kernel_matches_torch(ops, dev, B, K, O):
    g = torch.Generator().manual_seed(B + K + O)
    X = torch.randn(B, K, generator=g).to(dev)
    W = (torch.randn(O, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(O, generator=g).to(dev)
    gate = torch.randn(B, O, generator=g).to(dev)
    ref = X.double() @ W.double().t() + bias.double()
    torch.testing.assert_close(ops.linear(X, W, bias).double(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ops.linear(X, W, bias, relu=True).double(), ref.clamp(min=0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ops.linear(X, W, None, gate=gate).double(),
                               torch.where(gate > 0, (ref - bias.double()), torch.zeros_like(ref)), rtol=1e-5, atol=1e-5)
    if K % 8 == 0:   # partial slabs summed on load
        Xp = torch.randn(B, 3, K, generator=g).to(dev)
        refp = Xp.double().sum(1) @ W.double().t()
        torch.testing.assert_close(ops.linear(Xp, W, None, parts=3).double(), refp, rtol=1e-5, atol=1e-5)


def test_ragged_k_linear(ops, dev):
    X = torch.randn(7, 9, device=dev)
    W = torch.randn(256, 9, device=dev)
    torch.testing.assert_close(ops.linear(X, W).double(), X.double() @ W.double().t(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind", ["untargeted_logits", "logits", "cross_entropy"])
def test_cls_loss_kernel_matches_torch(ops, dev, kind):
    adv = importlib.import_module("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
    torch.manual_seed(3)
    B, k = 9, 40
    z = (torch.randn(B, k, device=dev) * 3).requires_grad_()
    tgt = torch.randint(0, k, (B,), device=dev)
    fn = {"untargeted_logits": adv.UntargetedLogitsAdvLoss(2.0), "logits": adv.LogitsAdvLoss(2.0),
          "cross_entropy": adv.CrossEntropyAdvLoss()}[kind]
    # reference: the functors' torch formulation, which is what they run on CPU tensors (on the GPU they use the kernel)
    zc = z.detach().cpu().requires_grad_()
    logp_ref = torch.log_softmax(zc, dim=1)
    fn(logp_ref, tgt.cpu()).mean().backward()
    logp, pred, loss, g = ops.cls_loss(z.detach(), tgt, kind, 2.0, scale=1.0 / B)
    torch.testing.assert_close(logp.cpu(), logp_ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(pred, z.argmax(1))
    torch.testing.assert_close(g.cpu(), zc.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(loss.mean().cpu(), fn(logp_ref.detach(), tgt.cpu()), rtol=1e-5, atol=1e-6)
    # raw mode (the loss on the tensor as given — log-probabilities or raw logits), through the functor itself: GPU
    # (kernel) against CPU (torch formulation), value and gradient, with a tie for the runner-up and [B,1] targets
    raw = (torch.randn(B, k) * 3)
    raw[0, 5] = raw[0, 9] = raw[0].max() + 1.0
    rc, rg = raw.clone().requires_grad_(), raw.clone().to(dev).requires_grad_()
    fn(rc, tgt.cpu().view(-1, 1) if kind != "cross_entropy" else tgt.cpu()).backward()
    out = fn(rg, tgt.view(-1, 1) if kind != "cross_entropy" else tgt)
    out.backward()
    torch.testing.assert_close(out.cpu(), fn(raw, tgt.cpu().view(-1, 1) if kind != "cross_entropy" else tgt.cpu()), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(rg.grad.cpu(), rc.grad, rtol=1e-6, atol=1e-8)
