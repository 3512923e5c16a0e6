"""GPU: the pruned one-wavefront farthest-point sampling (csrc/fps_pruned.hip, pc3d_fps_pruned_f32) returns the index
sequence of the full-update kernel (pc3d_fps_threads_f32, itself bit-exact against the reference's golden in
test_pointnet2_gpu.py) bit for bit — model/pointnet2_utils.py:60-81, model/curvenet_util.py:69-90."""
import importlib

import numpy as np
import pytest
import torch

from helpers import unit_cloud
from oracle import ref_torch as ort

pytestmark = pytest.mark.gpu
lib = importlib.import_module("3dpointcloudattack_amd._lib")


def _call(name, x, S, start, *pre):
    B, N = x.shape[0], x.shape[1]
    out = torch.empty((B, S), dtype=torch.int32, device=x.device)
    lib.call(name, *pre, x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, N, S,
             start.data_ptr() if start is not None else 0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out


def _full(x, S, start):
    return _call("pc3d_fps_threads_f32", x, S, start, 256)


def _pruned(x, S, start):
    return _call("pc3d_fps_pruned_f32", x, S, start)


def _clouds(kind, rng, B, N):
    if kind == "ball":
        return np.stack([unit_cloud(rng, N) for _ in range(B)])
    if kind == "gauss":
        return rng.standard_normal((B, N, 3)).astype(np.float32)
    if kind == "lattice":       # many exactly equidistant points: every tie rule is exercised
        return (np.round(rng.standard_normal((B, N, 3)) * 4) / 4).astype(np.float32)
    if kind == "plane":         # zero extent along z: that dimension is never split
        p = rng.random((B, N, 3)).astype(np.float32)
        p[..., 2] = 0.25
        return p
    if kind == "line":
        p = np.zeros((B, N, 3), np.float32)
        p[..., 0] = rng.random((B, N)).astype(np.float32)
        return p
    if kind == "dupes":         # a quarter of the cloud are copies of other points
        p = rng.standard_normal((B, N, 3)).astype(np.float32)
        src = rng.integers(0, N, (B, N // 4))
        dst = rng.integers(0, N, (B, N // 4))
        for b in range(B):
            p[b, dst[b]] = p[b, src[b]]
        return p
    if kind == "point":         # all points identical: distances all 0 after the first step
        return np.full((B, N, 3), 0.5, np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["ball", "gauss", "lattice", "plane", "line", "dupes", "point"])
@pytest.mark.parametrize("B,N,S", [(3, 4096, 1024), (4, 2048, 512), (3, 1024, 256), (2, 512, 128), (2, 1000, 300),
                                   (2, 2500, 100), (2, 200, 200), (3, 64, 64), (2, 37, 37), (1, 1, 1), (2, 65, 10)])
def test_pruned_fps_equals_full_update_kernel(dev, kind, B, N, S):
    rng = np.random.default_rng(N * 7 + S)
    x = torch.from_numpy(_clouds(kind, rng, B, N)).to(dev)
    start = torch.from_numpy(rng.integers(0, N, B).astype(np.int32)).to(dev)
    ref = _full(x, S, start)
    for trial in range(2):      # the internal layout depends on LDS atomics' arrival order; the picks must not
        got = _pruned(x, S, start)
        assert torch.equal(got, ref), (kind, N, S, trial, (got != ref).nonzero()[:4].tolist())
    assert torch.equal(_pruned(x, S, None), _full(x, S, None))            # start 0 (CurveNet's convention)
    xt = x.transpose(1, 2).contiguous().transpose(1, 2)                   # channels-first storage, same values
    assert torch.equal(_pruned(xt, S, start), ref)


def test_pruned_fps_vs_oracle_and_odd_inputs(dev):
    rng = np.random.default_rng(3)
    x = torch.from_numpy(np.stack([unit_cloud(rng, 777) for _ in range(2)]))
    start = torch.tensor([5, 700], dtype=torch.int32)
    want = ort.farthest_point_sample(x, 200, start.long())
    got = _pruned(x.to(dev), 200, start.to(dev))
    assert torch.equal(got.cpu().long(), want)
    # NaN / inf coordinates and out-of-range starts: the same picks as the full-update kernel, no fault
    bad = x.clone().to(dev)
    bad[0, 10] = float("nan")
    bad[0, 200, 1] = float("inf")
    bad[1, :, :] = float("nan")
    st = torch.tensor([3, 10_000], dtype=torch.int32, device=dev)
    assert torch.equal(_pruned(bad, 64, st), _full(bad, 64, st))
    with pytest.raises(lib.Pc3dError):
        _pruned(torch.zeros(1, 5000, 3, device=dev), 8, None)


def test_fps_default_choice_matches_full_update(ops, dev):
    """ops.fps (pc3d_fps_f32 picks the kernel by N) against the full-update kernel at the attack configs' shapes."""
    rng = np.random.default_rng(11)
    for B, N, S in ((4, 4096, 1024), (4, 2048, 512), (4, 512, 128), (4, 1024, 256), (4, 256, 64), (2, 6000, 64)):
        x = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).to(dev)
        start = torch.from_numpy(rng.integers(0, N, B).astype(np.int32)).to(dev)
        assert torch.equal(ops.fps(x, S, start), _full(x, S, start)), (N, S)
