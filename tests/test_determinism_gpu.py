"""GPU: the backward kernels that scatter through an index are DETERMINISTIC (csrc/det.hip) — the property the reference's
autograd of index_points / get_graph_feature / knn_gather has on its CPU path (model/pointnet2_utils.py:41-57,
model/dgcnn.py:203-227, attack/GeoA3/knn_utils.py:58-86). For every such operator:
  * two backward passes of the same inputs give bit-identical gradients (ten repetitions),
  * a cloud's gradient is bit-identical alone and inside a batch (the sums never cross clouds, tilings never depend on B),
  * the ordered sum agrees with the float-atomic flavour (PC3D_DETERMINISTIC=0) to fp32 rounding and with a float64
    evaluation of the same scatter.
"""
import importlib

import numpy as np
import pytest
import torch

from helpers import unit_cloud

pytestmark = pytest.mark.gpu
REPS = 10


def _clouds(rng, B, N):
    return torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))


def _grads(fn, inputs):
    """fn(*leaf copies) -> scalar; returns the gradients with respect to the float inputs that require grad."""
    leaves = [t.detach().clone().requires_grad_(t.is_floating_point()) if torch.is_tensor(t) else t for t in inputs]
    out = fn(*leaves)
    out.backward()
    return [t.grad.clone() for t in leaves if torch.is_tensor(t) and t.is_floating_point() and t.grad is not None]


def _check(ops, make, B, wsum=True, rtol=2e-5, atol=1e-6):
    """make(b_slice) -> (fn, inputs) for the clouds in b_slice (a python slice over the batch)."""
    assert ops.DETERMINISTIC
    fn, inputs = make(slice(0, B))
    ref = _grads(fn, inputs)
    for _ in range(REPS):
        again = _grads(fn, inputs)
        for a, b in zip(ref, again):
            assert torch.equal(a, b), "two runs of the same backward differ"
    # a cloud alone == the cloud inside the batch
    for k in (0, B - 1):
        fn1, in1 = make(slice(k, k + 1))
        alone = _grads(fn1, in1)
        for a, b in zip(alone, ref):
            if b.shape[0] == B:
                assert torch.equal(a[0], b[k]), f"cloud {k}: alone != in the batch"
    # float-atomic flavour: same numbers to rounding
    with ops.deterministic(False):
        atom = _grads(fn, inputs)
    for a, b in zip(ref, atom):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=rtol, atol=atol * max(1.0, float(b.abs().max())))
    return ref


def test_scatter_rows_det_vs_float64(ops, dev):
    _lib = importlib.import_module("3dpointcloudattack_amd._lib")
    rng = np.random.default_rng(0)
    # (the last two: more destination rows than a CU's LDS holds — row tiles, every tile walking all records)
    for B, R, N, C, clamp in [(3, 1000, 257, 32, 0), (2, 5000, 64, 3, 0), (1, 77, 4096, 5, 1), (2, 300, 9000, 1, 0), (4, 2048, 1024, 131, 0),
                              (2, 30000, 50000, 6, 0), (1, 5000, 100001, 3, 1)]:
        tgt = torch.from_numpy(rng.integers(-2, N + 2, size=(B, R)).astype(np.int32)).to(dev)
        val = torch.from_numpy(rng.standard_normal((B, R, C)).astype(np.float32)).to(dev)
        act = torch.from_numpy(rng.standard_normal((B, R, C)).astype(np.float32)).to(dev)
        for use_act in (False, True):
            out = torch.full((B, N, C), float("nan"), device=dev)
            with torch.cuda.device(dev):
                _lib.call("pc3d_scatter_rows_det_f32", tgt.data_ptr(), val.data_ptr(), C, act.data_ptr() if use_act else 0, C,
                          0.2, B, R, N, C, out.data_ptr(), C, 0, clamp, torch.cuda.current_stream().cuda_stream)
            v = val.double()
            if use_act:
                v = torch.where(act > 0, v, 0.2 * v)
            t = tgt.long()
            ok = (t >= 0) & (t < N)
            if clamp:
                t, ok = t.clamp(0, N - 1), torch.ones_like(ok)
            ref = torch.zeros((B, N, C), dtype=torch.float64, device=dev)
            for b in range(B):
                ref[b].index_add_(0, t[b][ok[b]], v[b][ok[b]])
            np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
            out2 = torch.empty_like(out)
            with torch.cuda.device(dev):
                _lib.call("pc3d_scatter_rows_det_f32", tgt.data_ptr(), val.data_ptr(), C, act.data_ptr() if use_act else 0, C,
                          0.2, B, R, N, C, out2.data_ptr(), C, 0, clamp, torch.cuda.current_stream().cuda_stream)
            assert torch.equal(out, out2)


def test_arg_scatter_more_rows_than_lds(ops, dev):
    """The per-channel-target scatter (the backward of a max over gathered rows) with 50 000 destination rows: row tiles
    instead of the refusal of round 3; against a float64 scatter, twice bit-equal."""
    _lib = importlib.import_module("3dpointcloudattack_amd._lib")
    rng = np.random.default_rng(1)
    B, N, C = 2, 50000, 8
    g = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    arg = torch.from_numpy(rng.integers(0, N, size=(B, N, C)).astype(np.int32)).to(dev)
    outs = []
    for _ in range(2):
        gp = torch.full((B, N, C), float("nan"), device=dev)
        with torch.cuda.device(dev):
            _lib.call("pc3d_gather_max_bwd_f32", g.data_ptr(), arg.data_ptr(), B, N, C, gp.data_ptr(), 1,
                      torch.cuda.current_stream().cuda_stream)
        outs.append(gp)
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros((B, N, C), dtype=torch.float64, device=dev)
    ref.scatter_add_(1, arg.long(), g.double())
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,E,NA,C", [(3, 20480, 1024, 32), (2, 777, 64, 16), (2, 5000, 300, 3), (1, 64, 4096, 128)])
def test_rev_index_sorted_and_gather_vs_float64(ops, dev, B, E, NA, C):
    _lib = importlib.import_module("3dpointcloudattack_amd._lib")
    rng = np.random.default_rng(E)
    idx = torch.from_numpy(rng.integers(0, NA, size=(B, E)).astype(np.int32))
    idx[0, : E // 3] = 5                                      # one crowded row (a degenerate graph): > 64 entries in a segment
    idx = idx.to(dev)
    off, lst = ops.rev_index(idx, NA)
    off2, lst2 = ops.rev_index(idx, NA)
    assert torch.equal(off, off2) and torch.equal(lst, lst2)
    o, l, ix = off.cpu().numpy(), lst.cpu().numpy(), idx.cpu().numpy()
    for b in range(B):
        assert o[b, 0] == 0 and o[b, -1] == E
        for t in list(rng.integers(0, NA, size=50)) + [5]:
            seg = l[b, o[b, t]:o[b, t + 1]]
            assert np.array_equal(seg, np.nonzero(ix[b] == t)[0])          # exactly the entries that read row t, ascending
    val = torch.from_numpy(rng.standard_normal((B, E, C)).astype(np.float32)).to(dev)
    act = torch.from_numpy(rng.standard_normal((B, E, C)).astype(np.float32)).to(dev)
    for use_act in (False, True):
        out = torch.full((B, NA, C), float("nan"), device=dev)
        with torch.cuda.device(dev):
            _lib.call("pc3d_rev_gather_sum_f32", val.data_ptr(), C, act.data_ptr() if use_act else 0, C, 0.2, off.data_ptr(),
                      lst.data_ptr(), B, E, NA, C, out.data_ptr(), C, torch.cuda.current_stream().cuda_stream)
        v = val.double()
        if use_act:
            v = torch.where(act > 0, v, 0.2 * v)
        ref = torch.zeros((B, NA, C), dtype=torch.float64, device=dev)
        for b in range(B):
            ref[b].index_add_(0, idx[b].long(), v[b])
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=3e-4)     # (a 6826-term fp32 sum)


@pytest.mark.parametrize("B,N,C,K", [(4, 1024, 64, 20), (3, 300, 8, 7), (2, 4096, 32, 20)])
def test_edge_max_backward(ops, dev, B, N, C, K):
    rng = np.random.default_rng(N + C)
    PQ = torch.from_numpy(rng.standard_normal((B, N, 2 * C)).astype(np.float32)).to(dev)
    idx = torch.from_numpy(rng.integers(0, N, size=(B, N, K)).astype(np.int32)).to(dev)
    w = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda pq: (ops.edge_max(pq, idx[s], 0.2) * w[s]).sum(), [PQ[s]]), B)


@pytest.mark.parametrize("B,N,S,C,K", [(4, 1024, 256, 64, 20), (2, 4096, 1024, 32, 20)])
def test_gather_max_backward(ops, dev, B, N, S, C, K):
    rng = np.random.default_rng(N + S)
    P = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    idx = torch.from_numpy(rng.integers(0, N, size=(B, N, K)).astype(np.int32)).to(dev)
    idx_rows = torch.from_numpy(rng.integers(0, N, size=(B, S, K)).astype(np.int32)).to(dev)
    w = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda p: (ops.gather_max(p, idx[s]) * w[s]).sum(), [P[s]]), B)
    _check(ops, lambda s: (lambda p: (ops.gather_max_rows(p, idx_rows[s]) * w[s][:, :S]).sum(), [P[s]]), B)


@pytest.mark.parametrize("B,N,C,K", [(4, 1024, 32, 20), (2, 256, 64, 20), (3, 64, 128, 20), (2, 1024, 16, 20)])
def test_lpfa_backward(ops, dev, B, N, C, K):
    rng = np.random.default_rng(C)
    A = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    Bc = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    idx = torch.from_numpy(rng.integers(0, N, size=(B, N, K)).astype(np.int32)).to(dev)
    W = torch.from_numpy((rng.standard_normal((C, C)) / C ** 0.5).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda a, bc: (ops.lpfa_fused(a, bc, idx[s], W, bias) * w[s]).sum(), [A[s], Bc[s]]), B, rtol=1e-4, atol=1e-5)
    wE = torch.from_numpy(rng.standard_normal((B, N, K, C)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda a, bc: (ops.edge_act(a, bc, idx[s], 0.2) * wE[s]).sum(), [A[s], Bc[s]]), B, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,N,S,ns,D", [(4, 2048, 512, 32, 0), (3, 512, 128, 64, 128), (2, 300, 50, 16, 7)])
def test_group_gather_and_group_act_backward(ops, dev, B, N, S, ns, D):
    rng = np.random.default_rng(S)
    xyz = _clouds(rng, B, N).to(dev)
    feat = torch.from_numpy(rng.standard_normal((B, N, D)).astype(np.float32)).to(dev) if D else None
    cidx = torch.from_numpy(np.stack([rng.choice(N, S, replace=False) for _ in range(B)]).astype(np.int32)).to(dev)
    centers = torch.gather(xyz, 1, cidx.long()[..., None].expand(-1, -1, 3)).contiguous()
    idx = ops.ball_query(0.3, ns, xyz, centers)
    w = torch.from_numpy(rng.standard_normal((B, S, ns, 3 + D)).astype(np.float32)).to(dev)

    def make(s):
        ins = [xyz[s]] + ([feat[s]] if D else [])
        return (lambda x, f=None: (ops.group_gather(x, f, idx[s], centers[s], cidx[s]) * w[s]).sum()), ins
    _check(ops, make, B, rtol=1e-4, atol=1e-5)

    C1 = 64
    P = torch.from_numpy(rng.standard_normal((B, N, C1)).astype(np.float32)).to(dev)
    Bc = torch.from_numpy(rng.standard_normal((B, S, C1)).astype(np.float32)).to(dev)
    wH = torch.from_numpy(rng.standard_normal((B, S, ns, C1)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda p, bc: (ops.group_act(p, bc, idx[s], 0.0) * wH[s]).sum(), [P[s], Bc[s]]), B, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,N,S,ns,C1,C2,C3", [(4, 2048, 512, 32, 64, 64, 128), (3, 512, 128, 64, 128, 128, 256)])
def test_grouped_mlp_max_backward_through_sorted_reverse_index(ops, dev, B, N, S, ns, C1, C2, C3):
    rng = np.random.default_rng(C3)
    xyz = _clouds(rng, B, N).to(dev)
    cidx = torch.from_numpy(np.stack([rng.choice(N, S, replace=False) for _ in range(B)]).astype(np.int64)).to(dev)
    centers = torch.gather(xyz, 1, cidx[..., None].expand(-1, -1, 3)).contiguous()
    idx = ops.ball_query(0.25, ns, xyz, centers)
    P = torch.from_numpy(rng.standard_normal((B, N, C1)).astype(np.float32)).to(dev)
    Bc = torch.from_numpy(rng.standard_normal((B, S, C1)).astype(np.float32)).to(dev)
    layers = [(torch.from_numpy((rng.standard_normal((C2, C1)) / C1 ** 0.5).astype(np.float32)).to(dev),
               torch.from_numpy(rng.standard_normal(C2).astype(np.float32)).to(dev)),
              (torch.from_numpy((rng.standard_normal((C3, C2)) / C2 ** 0.5).astype(np.float32)).to(dev),
               torch.from_numpy(rng.standard_normal(C3).astype(np.float32)).to(dev))]
    if not ops.grouped_mlp_max_supported(C1, ns, layers):
        pytest.skip("shape not on the fused path")
    w = torch.from_numpy(rng.standard_normal((B, S, C3)).astype(np.float32)).to(dev)

    def make(s):
        rev = ops.group_reverse(idx[s].contiguous(), N)
        return (lambda p, bc: (ops.grouped_mlp_max(p, bc, idx[s].contiguous(), layers, rev=rev) * w[s]).sum()), [P[s], Bc[s]]
    _check(ops, make, B, rtol=1e-4, atol=1e-5)
    # the reverse index itself: every list ascending, identical from build to build
    off, lst = ops.group_reverse(idx.contiguous(), N)
    off2, lst2 = ops.group_reverse(idx.contiguous(), N)
    assert torch.equal(off, off2)
    o = off.cpu().numpy()
    l1, l2 = lst.cpu().numpy(), lst2.cpu().numpy()
    for b in range(B):
        n_valid = o[b, -1]
        assert np.array_equal(l1[b, :n_valid], l2[b, :n_valid])
        for p in rng.integers(0, N, size=64):
            seg = l1[b, o[b, p]:o[b, p + 1]]
            assert np.all(np.diff(seg) > 0)


@pytest.mark.parametrize("B,N,K1", [(4, 1024, 17), (2, 4096, 17), (3, 257, 4)])
def test_kappa_backward(ops, dev, B, N, K1):
    rng = np.random.default_rng(N)
    pts = _clouds(rng, B, N).to(dev)
    _, idx = ops.knn_raw(pts, pts, K1)
    nrm = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((B, N, 3)).astype(np.float32)), dim=2).to(dev)
    w = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda p: (ops.kappa(p, nrm[s], idx[s].contiguous()) * w[s]).sum(), [pts[s]]), B, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,N,C,k,cn,L", [(4, 1024, 32, 20, 100, 5), (3, 256, 16, 20, 100, 5), (2, 64, 64, 20, 10, 3)])
def test_curve_walk_backward(ops, dev, B, N, C, k, cn, L):
    rng = np.random.default_rng(C + N)
    feats = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dev)
    adj = torch.from_numpy(rng.integers(0, N, size=(B, N, k)).astype(np.int32)).to(dev)
    start = torch.from_numpy(np.stack([rng.choice(N, cn, replace=False) for _ in range(B)]).astype(np.int32)).to(dev)
    aw = torch.from_numpy((rng.standard_normal(2 * C) / C ** 0.5).astype(np.float32)).to(dev)
    ab = torch.from_numpy(rng.standard_normal(1).astype(np.float32)).to(dev)
    mw = torch.from_numpy((rng.standard_normal((2, 2 * C)) / C ** 0.5).astype(np.float32)).to(dev)
    mb = torch.from_numpy(rng.standard_normal(2).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.standard_normal((B, cn, L, C)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda f: (ops.curve_walk(f, adj[s], start[s], aw, ab, mw, mb, L) * w[s]).sum(), [feats[s]]), B,
           rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("B,N,M", [(4, 1024, 1024), (2, 4096, 3000), (3, 100, 257)])
def test_set_distance_and_knn_backward(ops, dev, B, N, M):
    rng = np.random.default_rng(M)
    b = _clouds(rng, B, M).to(dev)
    a = (_clouds(rng, B, N) * 0.97).to(dev)
    for red in ("mean", "max"):
        _check(ops, lambda s: (lambda x, y: sum(t.sum() for t in ops.set_distance(x, y, red)), [a[s], b[s]]), B, rtol=1e-4, atol=1e-5)
    K = 6
    w = torch.from_numpy(rng.standard_normal((B, N, K)).astype(np.float32)).to(dev)
    _check(ops, lambda s: (lambda x: (ops.knn(x, x, K)[0] * w[s]).sum(), [a[s]]), B, rtol=1e-4, atol=1e-5)


def test_attack_level_deterministic_keyword(ops, dev):
    """`deterministic=` on the attack constructors / cfg: the mode holds for the duration of attack() and the process-wide
    switch is restored afterwards; False really selects the float-atomic kernels (the two modes agree to rounding)."""
    M = importlib.import_module
    knn = M("3dpointcloudattack_amd.attack.KNN.KNN_attack")
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
    dist = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    clip = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    from oracle import ref_torch as ort
    ssg = M("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg(40)
    ssg.load_state_dict(ort.seeded_state_dict(ssg, 3))
    ssg = ssg.eval().to(dev)
    rng = np.random.default_rng(9)
    pcs = _clouds(rng, 2, 512)
    labels = torch.zeros(2, dtype=torch.long)
    seen = []
    orig = ops._det

    def spy(flag=None):
        seen.append(ops.DETERMINISTIC)
        return orig(flag)
    outs = {}
    for mode in (True, False):
        atk = knn.CWKNN(ssg, None, None, None, None, None, adv_func=adv.UntargetedLogitsAdvLoss(5.), dist_func=dist.ChamferkNNDist(),
                        clip_func=clip.ProjectInnerClipLinf(budget=0.18), attack_lr=1e-2, num_iter=3, deterministic=mode)
        seen.clear()
        ops._det = spy
        try:
            torch.manual_seed(4)
            outs[mode] = atk.attack(pcs, labels)[0]
        finally:
            ops._det = orig
        assert seen and all(v == mode for v in seen), (mode, set(seen))
        assert ops.DETERMINISTIC                       # restored
    np.testing.assert_allclose(outs[True], outs[False], rtol=0, atol=2e-3)
