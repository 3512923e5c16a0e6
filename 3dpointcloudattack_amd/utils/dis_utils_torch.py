"""MI355X mirror of utils/dis_utils_torch.py — torch twins of the metrics on [B,3,N] tensors, differentiable.

The reference's quirks are part of its observable behaviour and are kept: ``chamfer`` normalises by
``a.shape[1]`` / ``b.shape[1]`` (= 3 for the documented [B,3,N] input, not N) and returns batch element 0 only;
the Hausdorff functions look at element 0 only. Since only element 0 is returned, only element 0 is computed.
"""
import torch

from .. import ops


class _SqrtNN(torch.autograd.Function):
    """sqrt of the squared NN distances with cdist's backward convention: zero gradient at zero distance."""

    @staticmethod
    def forward(ctx, d2):
        d = torch.sqrt(d2)
        ctx.save_for_backward(d)
        return d

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return torch.where(d > 0, g / (2 * d), torch.zeros_like(g))


def euclidean_distances(a: torch.Tensor, b: torch.Tensor, p=2):
    """utils/dis_utils_torch.py:4-5 — sum of torch.diagonal(cdist(a, b)) with diagonal taken over dims (0,1)."""
    # The reference calls cdist on the RAW tensors, i.e. it reads [B,3,N] as 3 "points" of dimension N, then takes
    # torch.diagonal over dims (0,1) of the [B,3,3] result. Kept literally; a [B,3,3,N] difference is tiny.
    M = torch.sqrt(torch.sum((a.float()[:, :, None, :] - b.float()[:, None, :, :]) ** 2, dim=-1))
    return torch.sum(torch.diagonal(M))


def pairwise_distances(a: torch.Tensor, b: torch.Tensor, p=2):
    """:8-11 — [B,3,N],[B,3,M] -> [B,N,M] Euclidean (dense; for callers that want the matrix)."""
    return ops.pairwise(a.float(), b.float(), x_cf=True, y_cf=True, euclid=True)


def _nn0(a, b):
    dA, dB, _, _ = ops.nn_bidir(a[:1].float(), b[:1].float(), a_cf=True, b_cf=True)
    return _SqrtNN.apply(dA)[0], _SqrtNN.apply(dB)[0]


def chamfer(a, b):
    """:14-16 — (sum_j min_i M)/a.shape[1] + (sum_i min_j M)/b.shape[1], element 0."""
    da, db = _nn0(a, b)     # da[i] = min_j M[i,j] (M.min(2)), db[j] = min_i M[i,j] (M.min(1))
    return db.sum() / a.shape[1] + da.sum() / b.shape[1]


def sgd_hausdorff_dis(a, b):
    """:19-22 — max_i min_j M[0]."""
    da, _ = _nn0(a, b)
    return torch.max(da)


def bid_hausdorff_dis(a, b):
    """:25-28."""
    da, db = _nn0(a, b)
    return torch.max(torch.max(da), torch.max(db))
