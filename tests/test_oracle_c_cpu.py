"""CPU: the plain-C restatement (oracle/ref_c.c) against the reference's own golden vectors, against the numpy oracle
on seeded inputs, and its AddressSanitizer + UBSan self-test run (sanitizers on the CPU build)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import ref_numpy as orc

ODIR = os.path.join(ROOT, "oracle")
c_dp, c_fp, c_ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)


@pytest.fixture(scope="module")
def refc():
    subprocess.run(["make", "-C", ODIR], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(ODIR, "_build", "libref_c.so"))
    for name in ("refc_chamfer", "refc_sgd_hausdorff", "refc_bid_hausdorff"):
        getattr(lib, name).restype = ctypes.c_double
    return lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_fp)


def _nn(lib, a, b):
    (a, pa), (b, pb) = _f(a), _f(b)
    N, M = len(a), len(b)
    dA, dB = np.empty(N), np.empty(M)
    iA, iB = np.empty(N, np.int64), np.empty(M, np.int64)
    lib.refc_nn_bidir(pa, N, pb, M, dA.ctypes.data_as(c_dp), iA.ctypes.data_as(c_ip), dB.ctypes.data_as(c_dp),
                      iB.ctypes.data_as(c_ip))
    return dA, iA, dB, iB


def _metrics(lib, a, b):
    (a, pa), (b, pb) = _f(a), _f(b)
    N, M = len(a), len(b)
    w1, w2, i1, i2 = np.empty(N), np.empty(M), np.empty(N, np.int64), np.empty(M, np.int64)
    args = (pa, N, pb, M, w1.ctypes.data_as(c_dp), i1.ctypes.data_as(c_ip), w2.ctypes.data_as(c_dp), i2.ctypes.data_as(c_ip))
    return lib.refc_chamfer(*args), lib.refc_sgd_hausdorff(*args), lib.refc_bid_hausdorff(*args)


def test_c_metrics_vs_reference_golden(refc, metrics_fx):
    """chamfer / one-sided / bidirectional Hausdorff of utils/dis_utils_numpy.py on every stored pair, incl. the worked
    examples the reference documents in its comments (:40-46), a single point and ragged sizes."""
    fx = metrics_fx
    for nm in fx["np_names"]:
        out = fx[f"np_{nm}_out"]
        ch, h_ab, h_bid = _metrics(refc, fx[f"np_{nm}_a"], fx[f"np_{nm}_b"])
        h_ba = _metrics(refc, fx[f"np_{nm}_b"], fx[f"np_{nm}_a"])[1]
        # stored as [chamfer, sgd(a,b), sgd(b,a), bid]; the C entry points take fp32 clouds, hence 1e-6
        np.testing.assert_allclose([ch, h_ab, h_ba, h_bid], out, rtol=1e-6, atol=1e-7, err_msg=str(nm))


def _lcg_cloud(n, seed, dup=True):
    """The generator of oracle/ref_c_selftest.c."""
    st = np.uint32(seed)
    vals = np.empty(3 * n, np.float32)
    with np.errstate(over="ignore"):
        for i in range(3 * n):
            st = np.uint32(st * np.uint32(1664525) + np.uint32(1013904223))
            vals[i] = np.float32(np.float32(st >> np.uint32(8)) / np.float32(16777216.0)) - np.float32(0.5)
    p = vals.reshape(n, 3)
    if dup and n > 3:
        p[n - 1] = p[0]
        p[2] = p[1]
    return p


def test_c_vs_numpy_oracle_seeded(refc):
    rng = np.random.default_rng(0)
    for N, M in ((1, 1), (5, 300), (257, 64), (700, 700)):
        a, b = rng.normal(size=(N, 3)).astype(np.float32), rng.normal(size=(M, 3)).astype(np.float32)
        if N > 4:
            a[3] = a[1]                                         # exact ties: lowest index must win
        dA, iA, dB, iB = _nn(refc, a, b)
        rA, riA = orc.nn_sq(a, b)
        rB, riB = orc.nn_sq(b, a)
        np.testing.assert_allclose(dA, rA, rtol=1e-14), np.testing.assert_allclose(dB, rB, rtol=1e-14)
        assert np.array_equal(iA, riA) and np.array_equal(iB, riB)
        K = min(M, 7)
        d, idx = np.empty((N, K)), np.empty((N, K), np.int64)
        (_, pa), (_, pb) = _f(a), _f(b)
        refc.refc_knn(pa, N, pb, M, K, d.ctypes.data_as(c_dp), idx.ctypes.data_as(c_ip))
        D = ((a[:, None, :].astype(np.float64) - b[None].astype(np.float64)) ** 2).sum(-1)
        order = np.argsort(D, axis=1, kind="stable")[:, :K]
        assert np.array_equal(idx, order)
        np.testing.assert_allclose(d, np.take_along_axis(D, order, 1), rtol=1e-14)


def test_c_fps_ball_vs_torch_oracle(refc):
    """FPS index sequence and ball-query lists against the oracle's restatement of model/pointnet2_utils.py:60-104
    (itself bit-exact against the reference's golden FPS sequence)."""
    import torch
    from oracle import ref_torch as ort
    rng = np.random.default_rng(3)
    x = (rng.random((500, 3)).astype(np.float32) - 0.5)
    S, ns = 64, 16
    out, ws = np.empty(S, np.int64), np.empty(500, np.float32)
    (_, px) = _f(x)
    refc.refc_fps(px, 500, S, 7, out.ctypes.data_as(c_ip), ws.ctypes.data_as(c_fp))
    ref = ort.farthest_point_sample(torch.from_numpy(x)[None], S, start=torch.tensor([7]))[0].numpy()
    assert np.array_equal(out, ref)
    ctr = np.ascontiguousarray(x[out])
    bq = np.empty((S, ns), np.int64)
    refc.refc_ball_query(px, 500, ctr.ctypes.data_as(c_fp), S, ctypes.c_float(0.2), ns, bq.ctypes.data_as(c_ip))
    rbq = ort.query_ball_point(0.2, ns, torch.from_numpy(x)[None], torch.from_numpy(ctr)[None], exact=True)[0].numpy()
    assert np.array_equal(bq, rbq)


def test_c_selftest_under_address_sanitizer(refc):
    """The ASan + UBSan build runs every function on exactly-sized heap buffers (one point, ragged sizes, K == M,
    duplicates, an empty ball): exit code 0, no sanitizer report, and its checksums equal the numpy oracle's on the
    same seeded inputs."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(ODIR, "_build", "ref_c_asan")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr
    shapes = [(1, 1), (1, 7), (7, 1), (64, 64), (100, 37), (257, 512), (1024, 1024)]
    lines = [l.split() for l in r.stdout.splitlines()]
    nn_lines = [l for l in lines if l and l[0] == "nn"]
    assert len(nn_lines) == len(shapes)
    for c, ((N, M), l) in enumerate(zip(shapes, nn_lines)):
        a, b = _lcg_cloud(N, 11 + c), _lcg_cloud(M, 101 + c)
        assert (int(l[1]), int(l[2])) == (N, M)
        dA, iA = orc.nn_sq(a, b)
        dB, iB = orc.nn_sq(b, a)
        np.testing.assert_allclose(float(l[3]), dA.sum() + 2 * dB.sum(), rtol=1e-12)
        assert int(l[4]) == int(iA.sum() + 3 * iB.sum())
        np.testing.assert_allclose([float(l[5]), float(l[6]), float(l[7])],
                                   [orc.chamfer(a, b), orc.sgd_hausdorff_dis(a, b), orc.bid_hausdorff_dis(a, b)], rtol=1e-12)
