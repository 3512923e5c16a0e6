"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product package (3dpointcloudattack_amd/).

CPU restatement (numpy, float64 unless stated) of the reference's point-set metrics. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the checker.

Pinned against the real reference: tests/golden/make_golden.py imports /root/reference in the build
container and stores inputs + the reference's own outputs in tests/golden/*.npz; tests/test_oracle_golden.py
checks every function here against those vectors (incl. the worked examples commented at
utils/dis_utils_numpy.py:40-46 and utils/dis_utils_torch.py:30-35).
"""
import numpy as np


def pairwise_distances(a, b, chunk=1024):
    """Euclidean distance matrix [N,M], float64 — utils/dis_utils_numpy.py:13-20 (scipy distance_matrix:
    sum(|x-y|**2)**0.5, i.e. the direct-difference form)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
    for s in range(0, a.shape[0], chunk):
        d = a[s:s + chunk, None, :] - b[None, :, :]
        out[s:s + chunk] = np.sqrt(np.sum(d * d, axis=-1))
    return out


def chamfer(a, b):
    """mean_i min_j M + mean_j min_i M (non-squared, no 1/2) — utils/dis_utils_numpy.py:23-26."""
    M = pairwise_distances(a, b)
    return float(np.mean(np.min(M, axis=1)) + np.mean(np.min(M, axis=0)))


def sgd_hausdorff_dis(a, b):
    """max_i min_j M — utils/dis_utils_numpy.py:29-32."""
    M = pairwise_distances(a, b)
    return float(np.max(np.min(M, axis=1)))


def bid_hausdorff_dis(a, b):
    """max of both one-sided values — utils/dis_utils_numpy.py:35-38."""
    return float(max(sgd_hausdorff_dis(a, b), sgd_hausdorff_dis(b, a)))


# ----------------------------------------------------------------------------------------------------------
# squared NN (what the HIP kernel K1 returns), float64 and an fp32-FMA-order emulation for index parity
# ----------------------------------------------------------------------------------------------------------
def nn_sq(q, r, chunk=512):
    """(min_j |q_i-r_j|^2 [N] float64, argmin [N] int64; lowest index on ties) for one cloud pair."""
    q = np.asarray(q, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    dmin = np.empty(q.shape[0])
    imin = np.empty(q.shape[0], dtype=np.int64)
    for s in range(0, q.shape[0], chunk):
        d = q[s:s + chunk, None, :] - r[None, :, :]
        d2 = np.sum(d * d, axis=-1)
        imin[s:s + chunk] = np.argmin(d2, axis=1)
        dmin[s:s + chunk] = np.min(d2, axis=1)
    return dmin, imin


def nn_sq_f32(q, r, chunk=512):
    """Same as nn_sq but following the kernel's fp32 operation order: d = fma(dz,dz, fma(dy,dy, dx*dx)) with
    dx = fl32(r-q). Emulated through float64 (products of two fp32 are exact in fp64; the fused add is
    rounded once to fp32), which reproduces the fp32 FMA chain except for rare double-rounding cases."""
    q = np.asarray(q, dtype=np.float32)
    r = np.asarray(r, dtype=np.float32)
    dmin = np.empty(q.shape[0], dtype=np.float32)
    imin = np.empty(q.shape[0], dtype=np.int64)
    for s in range(0, q.shape[0], chunk):
        d = (r[None, :, :] - q[s:s + chunk, None, :]).astype(np.float64)  # fp32 subtract, widened
        acc = (d[..., 0] * d[..., 0]).astype(np.float32)
        acc = (d[..., 1] * d[..., 1] + acc.astype(np.float64)).astype(np.float32)
        acc = (d[..., 2] * d[..., 2] + acc.astype(np.float64)).astype(np.float32)
        imin[s:s + chunk] = np.argmin(acc, axis=1)
        dmin[s:s + chunk] = np.min(acc, axis=1)
    return dmin, imin


def nn_bidir_batch(a, b):
    """Batched float64 oracle: a [B,N,3], b [B,M,3] -> dA [B,N], iA, dB [B,M], iB."""
    a = np.asarray(a)
    b = np.asarray(b)
    dA, iA, dB, iB = [], [], [], []
    for k in range(a.shape[0]):
        d, i = nn_sq(a[k], b[k])
        dA.append(d), iA.append(i)
        d, i = nn_sq(b[k], a[k])
        dB.append(d), iB.append(i)
    return np.stack(dA), np.stack(iA), np.stack(dB), np.stack(iB)


# ----------------------------------------------------------------------------------------------------------
# torch-twin semantics of utils/dis_utils_torch.py on [B,3,N] inputs, restated in numpy float64
# ----------------------------------------------------------------------------------------------------------
def torch_pairwise_distances(a, b):
    """utils/dis_utils_torch.py:8-11 — inputs [B,3,N]/[B,3,M] -> [B,N,M] Euclidean."""
    a = np.asarray(a, dtype=np.float64).transpose(0, 2, 1)
    b = np.asarray(b, dtype=np.float64).transpose(0, 2, 1)
    return np.stack([pairwise_distances(x, y) for x, y in zip(a, b)])


def torch_euclidean_distances(a, b):
    """utils/dis_utils_torch.py:4-5 — sum of the diagonal of cdist(a, b) taken over the LAST TWO dims of the
    raw inputs (no permute there): a,b are [B,P,D]; torch.diagonal defaults to dims (0,1)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    M = np.stack([pairwise_distances(x, y) for x, y in zip(a, b)])  # [B,P,R]
    return float(np.sum(np.diagonal(M, axis1=0, axis2=1)))


def torch_chamfer(a, b):
    """utils/dis_utils_torch.py:14-16 — batch element 0 only, and normalised by a.shape[1] / b.shape[1], which
    is 3 (channels) for the documented [B,3,N] input."""
    M = torch_pairwise_distances(a, b)
    v = M.min(axis=1).sum(axis=1) / np.asarray(a).shape[1] + M.min(axis=2).sum(axis=1) / np.asarray(b).shape[1]
    return float(v[0])


def torch_sgd_hausdorff_dis(a, b):
    """utils/dis_utils_torch.py:19-22 — element 0: max_i min_j."""
    M = torch_pairwise_distances(a, b)
    return float(np.max(M[0].min(axis=1)))


def torch_bid_hausdorff_dis(a, b):
    """utils/dis_utils_torch.py:25-28."""
    return float(max(torch_sgd_hausdorff_dis(a, b), torch_sgd_hausdorff_dis(b, a)))


# ----------------------------------------------------------------------------------------------------------
# CW functor semantics (attack/CW/CW_utils/distance.py), float64, true squared distances
# ----------------------------------------------------------------------------------------------------------
def cw_chamfer(preds, gts):
    """ChamferDistance.forward (distance.py:40-50): loss1[b] = mean over preds of min over gts (pred->gt),
    loss2[b] = mean over gts of min over preds; squared distances."""
    dA, _, dB, _ = nn_bidir_batch(preds, gts)
    return dA.mean(axis=1), dB.mean(axis=1)


def cw_hausdorff(preds, gts):
    """HausdorffDistance.forward (distance.py:58-70): same with max."""
    dA, _, dB, _ = nn_bidir_batch(preds, gts)
    return dA.max(axis=1), dB.max(axis=1)
