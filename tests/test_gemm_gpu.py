"""GPU: the fp32-MFMA point-wise layer (pc3d_gemm_nt_f32) vs plain torch fp32/fp64 — values for every activation, ragged
shapes (K = 3 / 131 / 259 as in PointNet++'s first layers, M / N off the 128-tile, strided views) and the backward to the
input through its autograd wrapper (same kernel on W^T with the activation's derivative applied on load)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dpointcloudattack_amd.ops")


def _ref(x, w, b, act, slope):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    if act == "relu":
        y = torch.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, slope)
    return y


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 40, 3), (200, 64, 3), (1000, 128, 131), (257, 130, 259), (4096, 64, 64),
                                   (32768, 128, 64), (3000, 1024, 512), (129, 129, 33),
                                   # few 128-row tiles, long K (CurveNet's deep levels)
                                   (8192, 128, 256), (2048, 512, 128), (2048, 128, 512), (2050, 130, 70)])
@pytest.mark.parametrize("act", [None, "relu", "leaky"])
def test_gemm_values(dev, M, N, K, act):
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    y = ops.gemm_nt(x, w, b, act, 0.2)
    ref = _ref(x, w, b, act, 0.2)
    torch.testing.assert_close(y.double(), ref, rtol=2e-5, atol=2e-5)
    y0 = ops.gemm_nt(x, w, None, act, 0.2)
    torch.testing.assert_close(y0.double(), _ref(x, w, None, act, 0.2), rtol=2e-5, atol=2e-5)


def test_gemm_exact_fp32_products(dev):
    """fp32-in MFMA keeps full fp32 products (no tf32-style truncation): 11-bit integers multiply to 22-bit products,
    and 40 of them sum below 2^24 — every intermediate is exact in fp32 and would not be with a 10-bit mantissa."""
    g = torch.Generator().manual_seed(0)
    x = torch.randint(-600, 600, (300, 40), generator=g).float().to(dev) * 2 + 1
    w = torch.randint(-600, 600, (70, 40), generator=g).float().to(dev) * 2 + 1
    y = ops.gemm_nt(x, w)
    assert torch.equal(y.double(), x.double() @ w.double().t())


def test_gemm_strided_rows_and_gate(dev):
    g = torch.Generator().manual_seed(5)
    big = torch.randn(500, 200, generator=g).to(dev)
    x = big[:, 8:8 + 96]                        # row stride 200, contiguous last dim
    w = torch.randn(72, 96, generator=g).to(dev) * 0.1
    gate = torch.randn(500, 96, generator=g).to(dev)
    out = torch.zeros(500, 100, device=dev)
    y = ops.gemm_nt(x, w, None, None, 0.0, gate=gate, gate_slope=0.2, out=out[:, 4:76])
    xg = torch.where(gate > 0, x, 0.2 * x)
    torch.testing.assert_close(y.double(), xg.double() @ w.double().t(), rtol=2e-5, atol=2e-5)
    assert float(out[:, :4].abs().max()) == 0.0 and float(out[:, 76:].abs().max()) == 0.0


@pytest.mark.parametrize("act", [None, "relu", "leaky"])
@pytest.mark.parametrize("shape,N", [((4, 300, 131), 128), ((2, 64, 32, 67), 64), ((700, 259), 256)])
def test_linear_act_backward(dev, act, shape, N):
    g = torch.Generator().manual_seed(N)
    K = shape[-1]
    x = torch.randn(*shape, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    gy = torch.randn(*shape[:-1], N, generator=g).to(dev)
    xa = x.clone().requires_grad_()
    ya = ops.linear_act(xa, w, b, act, 0.2)
    ya.backward(gy)
    xr = x.clone().double().requires_grad_()
    yr = F.linear(xr, w.double(), b.double())
    yr = torch.relu(yr) if act == "relu" else (F.leaky_relu(yr, 0.2) if act == "leaky" else yr)
    yr.backward(gy.double())
    torch.testing.assert_close(ya.double(), yr, rtol=2e-5, atol=2e-5)
    # activation kinks: the fp32 forward may put a pre-activation within rounding of zero on the other side
    bad = (xa.grad.double() - xr.grad).abs() > 2e-5 + 2e-5 * xr.grad.abs()
    assert float(bad.float().mean()) < 1e-3
    assert float((xa.grad.double() - xr.grad).norm() / xr.grad.norm()) < 1e-4
