"""GPU parity: HIP K1 (fused NN / Chamfer / Hausdorff) through the C-ABI vs the oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as orc

pytestmark = pytest.mark.gpu

REL = 1e-5  # north_star tolerance: Chamfer/Hausdorff within 1e-5 relative (fp32) of dis_utils_numpy


def _cloud(rng, n):
    g = rng.standard_normal((n, 3))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    p = g * rng.random((n, 1)) ** (1 / 3)
    p -= p.mean(0, keepdims=True)
    m = np.linalg.norm(p, axis=1).max()
    return (p / (m if m > 0 else 1.0) + (0.25 if n == 1 else 0.0)).astype(np.float32)


def _check_idx(q, r, d_gpu, i_gpu):
    """Indices must be bit-exact vs the fp32-order oracle except inside exact/near ties; then the distance of
    the returned index must still equal the minimum to fp32 rounding."""
    d32, i32 = orc.nn_sq_f32(q, r)
    exact = (i_gpu == i32)
    if not exact.all():
        bad = np.where(~exact)[0]
        dq = np.sum((q[bad].astype(np.float64) - r[i_gpu[bad]].astype(np.float64)) ** 2, axis=1)
        np.testing.assert_allclose(dq, d32[bad].astype(np.float64), rtol=4e-7, atol=1e-30)
    np.testing.assert_allclose(d_gpu, d32, rtol=2e-7, atol=1e-30)
    return exact.mean()


@pytest.mark.parametrize("B,N,M", [(1, 1, 1), (2, 64, 64), (3, 100, 257), (2, 1024, 1024), (1, 300, 5000),
                                   (2, 4100, 70), (1, 4096, 4096)])
@pytest.mark.parametrize("cf", [False, True])
@pytest.mark.parametrize("two_scan", [False, True])
def test_nn_bidir_matches_oracle(ops, dev, B, N, M, cf, two_scan):
    rng = np.random.default_rng(N * 7 + M)
    a = np.stack([_cloud(rng, N) for _ in range(B)])
    b = np.stack([_cloud(rng, M) for _ in range(B)])
    if N == M:
        a = (b + 0.01 * rng.standard_normal(b.shape)).astype(np.float32)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    if cf:  # channel-first views [B,3,N], the attack's native layout
        ta, tb = ta.transpose(1, 2).contiguous(), tb.transpose(1, 2).contiguous()
    dA, iA, dB, iB = ops.nn_bidir_raw(ta, tb, cf, cf, two_scan=two_scan)
    dA, iA, dB, iB = dA.cpu().numpy(), iA.cpu().numpy(), dB.cpu().numpy(), iB.cpu().numpy()
    for k in range(B):
        _check_idx(a[k], b[k], dA[k], iA[k])
        _check_idx(b[k], a[k], dB[k], iB[k])
        d64, _ = orc.nn_sq(a[k], b[k])
        np.testing.assert_allclose(dA[k], d64, rtol=1e-6, atol=1e-12)


def test_ties_resolve_to_lowest_index(ops, dev):
    r = torch.tensor([[[1., 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0]]], device=dev).repeat(1, 300, 1)  # many dups
    q = torch.tensor([[[1., 0, 0], [0, 1, 0], [0.5, 0.5, 0]]], device=dev)
    d, i = ops.nn_raw(q, r)
    assert i.cpu().tolist() == [[0, 2, 0]]
    assert d.cpu().tolist()[0][:2] == [0.0, 0.0]


def test_noncontiguous_strided_views(ops, dev):
    rng = np.random.default_rng(5)
    big = torch.from_numpy(rng.standard_normal((2, 200, 6)).astype(np.float32)).to(dev)
    a = big[:, ::2, :3]   # point stride 12, not contiguous
    b = big[:, 1::2, 3:]
    dA, iA, dB, iB = ops.nn_bidir_raw(a, b)
    an, bn = a.cpu().numpy(), b.cpu().numpy()
    for k in range(2):
        _check_idx(an[k], bn[k], dA[k].cpu().numpy(), iA[k].cpu().numpy())
        _check_idx(bn[k], an[k], dB[k].cpu().numpy(), iB[k].cpu().numpy())


def test_golden_numpy_metrics(ops, dev, metrics_fx):
    """dis_utils_numpy.{chamfer,sgd_hausdorff_dis,bid_hausdorff_dis} values produced by the reference."""
    fx = metrics_fx
    for nm in fx["np_names"]:
        a = torch.from_numpy(fx[f"np_{nm}_a"]).to(dev)[None]
        b = torch.from_numpy(fx[f"np_{nm}_b"]).to(dev)[None]
        dA, _, dB, _ = ops.nn_bidir_raw(a, b)
        ch = ops.rowreduce(dA, "mean", sqrt=True) + ops.rowreduce(dB, "mean", sqrt=True)
        hab = ops.rowreduce(dA, "max", sqrt=True)
        hba = ops.rowreduce(dB, "max", sqrt=True)
        got = [ch.item(), hab.item(), hba.item(), max(hab.item(), hba.item())]
        np.testing.assert_allclose(got, fx[f"np_{nm}_out"], rtol=REL, atol=1e-7, err_msg=str(nm))


def test_golden_cw_functors_and_grads(ops, dev, metrics_fx):
    """distance.py ChamferDistance/HausdorffDistance (loss1, loss2) + autograd gradients from the reference."""
    fx = metrics_fx
    for nm in fx["cw_names"]:
        p = torch.from_numpy(fx[f"cw_{nm}_preds"]).to(dev).requires_grad_()
        g = torch.from_numpy(fx[f"cw_{nm}_gts"]).to(dev).requires_grad_()
        w = torch.from_numpy(fx[f"cw_{nm}_w"]).float().to(dev)
        for det in (False, True):
            p.grad = g.grad = None
            l1, l2 = ops.set_distance(p, g, "mean", deterministic=det)
            np.testing.assert_allclose(torch.stack([l1, l2]).detach().cpu().numpy(), fx[f"cw_{nm}_chamfer"],
                                       rtol=REL, err_msg=str(nm))
            ((l1 * w[0]).sum() + (l2 * w[1]).sum()).backward()
            scale = np.abs(fx[f"cw_{nm}_chamfer_gpreds"]).max()
            np.testing.assert_allclose(p.grad.cpu().numpy(), fx[f"cw_{nm}_chamfer_gpreds"], rtol=1e-4,
                                       atol=1e-6 * scale, err_msg=str(nm))
            np.testing.assert_allclose(g.grad.cpu().numpy(), fx[f"cw_{nm}_chamfer_ggts"], rtol=1e-4,
                                       atol=1e-6 * scale, err_msg=str(nm))
        p.grad = None
        h1, h2 = ops.set_distance(p, g.detach(), "max")
        np.testing.assert_allclose(torch.stack([h1, h2]).detach().cpu().numpy(), fx[f"cw_{nm}_hausdorff"],
                                   rtol=REL, err_msg=str(nm))
        ((h1 * w[0]).sum() + (h2 * w[1]).sum()).backward()
        np.testing.assert_allclose(p.grad.cpu().numpy(), fx[f"cw_{nm}_hausdorff_gpreds"], rtol=1e-4,
                                   atol=1e-7, err_msg=str(nm))


def test_deterministic_backward_is_bitwise_reproducible(ops, dev):
    rng = np.random.default_rng(11)
    a = torch.from_numpy(np.stack([_cloud(rng, 700) for _ in range(3)])).to(dev).requires_grad_()
    b = torch.from_numpy(np.stack([_cloud(rng, 90) for _ in range(3)])).to(dev)  # many a->same b and b<-a
    outs = []
    for _ in range(3):
        a.grad = None
        l1, l2 = ops.set_distance(a, b, "mean", deterministic=True)
        (l1.sum() + 3 * l2.sum()).backward()
        outs.append(a.grad.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_full_size_properties(ops, dev):
    """BASELINE sizes B=32, N in {1024,2048,4096}: properties that need no O(N^2) oracle.
    (1) NN of a set in itself is (0, self); (2) symmetry dA(a,b)==dB(b,a); (3) a sampled row check."""
    rng = np.random.default_rng(99)
    for N in (1024, 2048, 4096):
        a = torch.from_numpy(rng.standard_normal((32, N, 3)).astype(np.float32)).to(dev)
        b = a + 0.01 * torch.randn_like(a)
        d, i = ops.nn_raw(a, a)
        assert torch.all(d == 0) and torch.equal(i.long(), torch.arange(N, device=dev).expand(32, N))
        dA, iA, dB, iB = ops.nn_bidir_raw(a, b)
        dA2, iA2, dB2, iB2 = ops.nn_bidir_raw(b, a)
        assert torch.equal(dA, dB2) and torch.equal(iA, iB2) and torch.equal(dB, dA2) and torch.equal(iB, iA2)
        for k in (0, 31):
            rows = rng.choice(N, 64, replace=False)
            d64, i64 = orc.nn_sq(a[k, rows].cpu().numpy(), b[k].cpu().numpy())
            np.testing.assert_allclose(dA[k, rows].cpu().numpy(), d64, rtol=1e-6)


@pytest.mark.parametrize("B,N,M", [(1, 1, 1), (1, 1, 300), (2, 300, 1), (3, 63, 65), (2, 129, 4097), (2, 4100, 70),
                                   (5, 256, 256), (32, 1024, 1024), (7, 2048, 1500), (32, 4096, 4096), (1, 9000, 8200)])
def test_shared_evaluation_equals_two_scan_bitwise(ops, dev, B, N, M):
    """pc3d_nn_bidir_shared_f32 (one distance evaluation feeds both directions, butterfly column reduction, partials
    folded by a second launch) against pc3d_nn_bidir_f32 (one scan per direction): identical distances bit for bit,
    identical indices (both resolve ties to the lowest index), for every work split the planner picks."""
    g = torch.Generator().manual_seed(N * 31 + M)
    b = torch.randn(B, M, 3, generator=g)
    a = torch.randn(B, N, 3, generator=g)
    if N == M:
        a = b + 0.01 * torch.randn(B, N, 3, generator=g)
    a, b = a.to(dev), b.to(dev)
    ref = ops.nn_bidir_raw(a, b, two_scan=True)
    got = ops.nn_bidir_raw(a, b, two_scan=False)
    for r, x, nm in zip(ref, got, ("dA", "iA", "dB", "iB")):
        assert torch.equal(r, x), (nm, int((r != x).sum()))
    dA, iA, dB, iB = ops.nn_bidir_raw(a, b, want_idx=False, two_scan=False)   # values only
    assert iA is None and iB is None and torch.equal(dA, ref[0]) and torch.equal(dB, ref[2])
    acf = a.transpose(1, 2).contiguous()                                # [B,3,N] layout of the attack loops
    got_cf = ops.nn_bidir_raw(acf, b, True, False, two_scan=False)
    assert all(torch.equal(r, x) for r, x in zip(ref, got_cf))


def test_shared_evaluation_ties_and_duplicates(ops, dev):
    """Exact ties (duplicated points on both sides) resolve to the LOWEST index in both directions."""
    base = torch.tensor([[1., 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]], device=dev)
    a = base.repeat(130, 1)[None]                                       # 520 points, every point 130 times
    b = base[[1, 3, 2]].repeat(90, 1)[None]                             # 270 points: (0,1,0), (1,0,0), (0,0,1), ...
    dA, iA, dB, iB = ops.nn_bidir_raw(a, b, two_scan=False)
    assert float(dA.abs().max()) == 0.0 and float(dB.abs().max()) == 0.0
    assert iA[0, :4].tolist() == [1, 0, 2, 1] and iB[0, :3].tolist() == [1, 0, 2]
    ref = ops.nn_bidir_raw(a, b, two_scan=True)
    assert torch.equal(iA, ref[1]) and torch.equal(iB, ref[3])
