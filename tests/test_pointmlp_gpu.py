"""GPU numerics: fused per-point MLP + max-pool (K8) vs a plain PyTorch fp32 reference of the same op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _weights(dev, C3, seed):
    g = torch.Generator().manual_seed(seed)

    def u(*s, k):
        return ((torch.rand(*s, generator=g) * 2 - 1) / np.sqrt(k)).to(dev)
    return (u(64, 3, k=3), u(64, k=3), u(128, 64, k=64), u(128, k=64), u(C3, 128, k=128), u(C3, k=128))


def _torch_full(x, w):
    W1, b1, W2, b2, W3, b3 = w
    h = F.relu(F.conv1d(x, W1[:, :, None], b1))
    h = F.relu(F.conv1d(h, W2[:, :, None], b2))
    return F.conv1d(h, W3[:, :, None], b3)


def _torch_ref(x, w, relu_last):
    h = _torch_full(x, w)
    if relu_last:
        h = F.relu(h)
    return h.max(dim=2)


@pytest.mark.parametrize("B,N,C3", [(2, 128, 1024), (3, 200, 256), (1, 1, 32), (2, 1024, 1024), (1, 1500, 1024)])
@pytest.mark.parametrize("relu_last", [False, True])
def test_forward_matches_torch(ops, dev, B, N, C3, relu_last):
    torch.manual_seed(N + C3)
    x = torch.randn(B, 3, N, device=dev) * 0.5
    w = _weights(dev, C3, 1)
    pooled, idx = ops.pointmlp3_max_fwd_raw(x, w, relu_last)
    ref, ridx = _torch_ref(x, w, relu_last)
    torch.testing.assert_close(pooled, ref, rtol=2e-5, atol=2e-6)
    # argmax: equal, or a tie within fp32 rounding of the max
    full = _torch_full(x, w)
    assert torch.all((idx >= 0) & (idx < N))
    picked = torch.gather(full, 2, idx.long()[:, :, None])[:, :, 0]
    torch.testing.assert_close(picked, full.max(dim=2)[0], rtol=2e-5, atol=2e-6)
    if not relu_last:
        assert (idx.long() == ridx).float().mean() > 0.98


@pytest.mark.parametrize("B,N,C3", [(2, 128, 1024), (2, 333, 256), (4, 1024, 1024)])
@pytest.mark.parametrize("relu_last", [False, True])
def test_backward_matches_torch_autograd(ops, dev, B, N, C3, relu_last):
    torch.manual_seed(7 * N + C3)
    x = (torch.randn(B, 3, N, device=dev) * 0.5).requires_grad_()
    w = _weights(dev, C3, 2)
    gout = torch.randn(B, C3, device=dev)
    pooled = ops.pointmlp3_max(x, w, relu_last)
    pooled.backward(gout)
    g_hip = x.grad.clone()
    x.grad = None
    ref, _ = _torch_ref(x, w, relu_last)
    ref.backward(gout)
    scale = x.grad.abs().max()
    torch.testing.assert_close(g_hip, x.grad, rtol=1e-3, atol=float(scale) * 2e-5)


def test_transform_argument(ops, dev):
    torch.manual_seed(3)
    x = torch.randn(2, 3, 300, device=dev)
    T = torch.randn(2, 3, 3, device=dev)
    w = _weights(dev, 1024, 3)
    pooled, _ = ops.pointmlp3_max_fwd_raw(x, w, False, T=T)
    xt = torch.bmm(x.transpose(1, 2), T).transpose(1, 2).contiguous()
    ref, _ = _torch_ref(xt, w, False)
    torch.testing.assert_close(pooled, ref, rtol=2e-5, atol=2e-6)


def test_backward_is_deterministic(ops, dev):
    torch.manual_seed(5)
    x = torch.randn(3, 3, 512, device=dev)
    w = _weights(dev, 1024, 4)
    pooled, idx, masks = ops.pointmlp3_max_fwd_raw(x, w, False, want_masks=True)
    g = torch.randn_like(pooled)
    a = ops.pointmlp3_max_bwd_raw(x, w, idx, g, masks)
    b = ops.pointmlp3_max_bwd_raw(x, w, idx, g, masks)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,N", [(2, 128), (3, 333), (1, 1), (2, 1024)])
def test_forward_relu_masks_match_torch(ops, dev, B, N):
    """mask1 / mask2 of pc3d_pointmlp3_max_fwd_f32 are the layer-1 / layer-2 ReLU decisions per point (the backward
    launch consumes them instead of recomputing): compare with torch's activations, allowing a pre-activation within
    fp32 rounding of zero to fall on either side."""
    torch.manual_seed(N)
    x = torch.randn(B, 3, N, device=dev) * 0.5
    w = _weights(dev, 256, 3)
    T = torch.randn(B, 3, 3, device=dev) * 0.5 if N > 1 else None
    _, _, (m1, m2) = ops.pointmlp3_max_fwd_raw(x, w, False, T=T, want_masks=True)
    assert m1.shape == (B, N) and m1.dtype == torch.int64 and m2.shape == (B, N, 4) and m2.dtype == torch.int32
    xt = x if T is None else torch.bmm(x.transpose(1, 2), T).transpose(1, 2)
    z1 = F.conv1d(xt.double(), w[0].double()[:, :, None], w[1].double())             # [B,64,N]
    z2 = F.conv1d(F.relu(z1), w[2].double()[:, :, None], w[3].double())               # [B,128,N]
    bits1 = ((m1.unsqueeze(-1) >> torch.arange(64, device=dev)) & 1).bool()           # [B,N,64]
    bits2 = ((m2.unsqueeze(-1) >> torch.arange(32, device=dev)) & 1).bool().reshape(B, N, 128)
    for bits, z in ((bits1, z1), (bits2, z2)):
        zt = z.transpose(1, 2)
        sure = zt.abs() > 1e-5
        assert torch.equal(bits[sure], (zt > 0)[sure])
        assert sure.float().mean() > 0.999


@pytest.mark.parametrize("B,N,K", [(3, 200, 256), (2, 1024, 256), (1, 70, 64)])
def test_transform_head_in_prologue(ops, dev, B, N, K):
    """pc3d_pointmlp3_max_fwd_th_f32: the input transform T = h @ W.T + b (STN3d's fc3 + identity, model/pointnet.py:45-47)
    evaluated in the launch's prologue == the two-launch form (linear, then the tower with T given): T to fp32 rounding
    of a different summation order, pooled values accordingly, and the same arg-max points."""
    g = torch.Generator().manual_seed(K + N)
    x = (torch.randn(B, 3, N, generator=g) * 0.5).to(dev)
    w = _weights(dev, 256, 2)
    h = torch.randn(B, K, generator=g).to(dev)
    Wt = (torch.randn(9, K, generator=g) / K ** 0.5).to(dev)
    bt = (torch.randn(9, generator=g) * 0.1 + torch.eye(3).reshape(-1)).to(dev)
    pooled, idx, masks, T = ops.pointmlp3_max_fwd_raw(x, w, False, want_masks=True, T_head=(h, Wt, bt))
    T_ref = (h.double() @ Wt.double().t() + bt.double()).float()
    torch.testing.assert_close(T, T_ref, rtol=1e-5, atol=1e-6)
    pooled2, idx2, masks2 = ops.pointmlp3_max_fwd_raw(x, w, False, T=T.view(B, 3, 3), want_masks=True)
    assert torch.equal(pooled, pooled2) and torch.equal(idx, idx2)            # same T -> bit-identical tower
    assert torch.equal(masks[0], masks2[0]) and torch.equal(masks[1], masks2[1])


@pytest.mark.parametrize("B,P,K,O", [(32, 32, 256, 512), (5, 3, 64, 40), (33, 8, 128, 17)])
def test_linear_pre_matches_two_launches(ops, dev, B, P, K, O):
    """pc3d_linear_pre_f32 (the backward of STN3d's fc3 folded into the launch of fc2's backward) against plain torch:
    X = relu'(gate_pre) * (sum_p parts[:, p, :9] @ Wp), Y = relu'(gate) * (X @ W.T)."""
    g = torch.Generator().manual_seed(B + K)
    parts = torch.randn(B, P, 16, generator=g).to(dev)
    Wp = torch.randn(9, K, generator=g).to(dev)
    gate_pre = torch.randn(B, K, generator=g).to(dev)
    W = (torch.randn(O, K, generator=g) / K ** 0.5).to(dev)
    gate = torch.randn(B, O, generator=g).to(dev)
    Y = ops.linear_pre(parts, 9, Wp, gate_pre, W, gate=gate)
    S = parts.double().sum(1)[:, :9]
    X = torch.where(gate_pre > 0, S @ Wp.double(), torch.zeros((), dtype=torch.float64, device=dev))
    ref = torch.where(gate > 0, X @ W.double().t(), torch.zeros((), dtype=torch.float64, device=dev))
    torch.testing.assert_close(Y.double(), ref, rtol=1e-4, atol=1e-4)
    Y0 = ops.linear_pre(parts, 9, Wp, gate_pre, W)                            # no output gate
    torch.testing.assert_close(Y0.double(), X @ W.double().t(), rtol=1e-4, atol=1e-4)
