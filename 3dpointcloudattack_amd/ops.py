"""Host-side operators over the C-ABI (include/pc3d.h): thin ctypes shims + torch.autograd.Functions.

PyTorch is plumbing here (device memory, streams, autograd graph); all arithmetic of these ops runs in the
hand-written gfx950 kernels of libpc3d_hip.so. There is no CPU/eager fallback: tensors must live on the GPU.
"""
import contextlib
import os

import numpy as np
import torch

from . import _lib
from . import graphed as _graphed

# Deterministic gradients (default): every backward that scatters through an index (kNN graphs, ball-query groups,
# arg-max routes) sums in a fixed order — one wavefront owns the LDS accumulator (csrc/det.hip) or gathers through a
# sorted reverse index — instead of float atomics whose order changes from run to run. Same seed => the same
# adversarial cloud, bit for bit: run == run, hipGraph replay == eager, a cloud in a batch == the cloud alone.
# PC3D_DETERMINISTIC=0 (or ops.set_deterministic(False)) selects the float-atomic kernels of rounds 1-2.
DETERMINISTIC = os.environ.get("PC3D_DETERMINISTIC", "1") != "0"


def set_deterministic(on):
    """Process-wide switch between the ordered and the float-atomic backward kernels; returns the previous value."""
    global DETERMINISTIC
    was, DETERMINISTIC = DETERMINISTIC, bool(on)
    return was


@contextlib.contextmanager
def deterministic(on=True):
    was = set_deterministic(on)
    try:
        yield
    finally:
        set_deterministic(was)


def _det(flag=None):
    return 1 if (DETERMINISTIC if flag is None else flag) else 0


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.Pc3dError(f"{name}: tensor is on {t.device}; pc3d ops run on the GPU only (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")


def _pts(t, channel_first, name="points"):
    """(ptr, batch_stride, point_stride, channel_stride, B, N) of a [B,N,3] or [B,3,N] fp32 GPU tensor."""
    _check(t, name)
    if t.dim() != 3:
        raise ValueError(f"{name}: expected 3 dims, got shape {tuple(t.shape)}")
    if channel_first:
        B, C, N = t.shape
        bs, cs, ps = t.stride()
    else:
        B, N, C = t.shape
        bs, ps, cs = t.stride()
    if C != 3:
        raise ValueError(f"{name}: expected 3 coordinates, got shape {tuple(t.shape)} (channel_first={channel_first})")
    return t.data_ptr(), bs, ps, cs, B, N


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------------
# raw (non-differentiable) ops
# ------------------------------------------------------------------------------------------------------
def nn_raw(q, r, q_cf=False, r_cf=False, want_idx=True, want_i64=False):
    """Nearest reference point of every query: (min_d2 [B,N] f32, idx [B,N] i32); with want_i64 a third tensor, the same
    indices as int64 written by the same launch."""
    qp, qbs, qps, qcs, B, N = _pts(q, q_cf, "q")
    rp, rbs, rps, rcs, B2, M = _pts(r, r_cf, "r")
    if B != B2:
        raise ValueError("q and r must have the same batch dimension")
    if M < 1:
        raise ValueError("reference set is empty")
    d = torch.empty((B, N), dtype=torch.float32, device=q.device)
    i = torch.empty((B, N), dtype=torch.int32, device=q.device) if want_idx else None
    if want_i64:
        i64 = torch.empty((B, N), dtype=torch.int64, device=q.device)
        with torch.cuda.device(q.device):
            _lib.call("pc3d_nn_i64_f32", qp, qbs, qps, qcs, rp, rbs, rps, rcs, B, N, M, d.data_ptr(), _ptr(i), i64.data_ptr(),
                      _stream())
        return d, i, i64
    with torch.cuda.device(q.device):
        _lib.call("pc3d_nn_f32", qp, qbs, qps, qcs, rp, rbs, rps, rcs, B, N, M, d.data_ptr(), _ptr(i), _stream())
    return d, i


_NN_WS = {}     # (device index, stream) -> scratch tensor of the shared-evaluation search, grown on demand


def _nn_workspace(dev, nbytes):
    """Scratch for pc3d_nn_bidir_shared_f32: one cached buffer per (device, stream) — work on one stream is ordered, so
    consecutive calls may reuse it; it only ever grows (a captured hipGraph keeps pointing at a buffer that stays
    alive and that later, smaller calls on the same stream still fit into)."""
    if torch.cuda.is_current_stream_capturing():
        # every capture runs on torch's one capture stream: a cached buffer would be baked into ALL captured graphs, and
        # two of them replayed on different streams would race on it. A capture gets its own scratch, owned by the
        # graph's keep-alive list (graphed.note_captured) and never entered into the cache.
        ws = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=dev)
        _graphed.note_captured(ws)
        if not _graphed.capturing():
            _NN_KEEP.append(ws)          # a capture outside capture_guard(): nobody else would own the buffer
        return ws
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
    ws = _NN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=dev)
        _NN_WS[key] = ws
    return ws


_NN_KEEP = []


NN_SHARED_MIN_PAIRS = 1 << 25    # B*N*M from which the shared-evaluation search wins. Measured on MI355X inside replayed
                                 # hipGraphs (tools/bench_nn_small.py; eager calls at these sizes time the host): B=32,
                                 # N=1024 (2^25 pairs) 14.2 us with indices / 11.8 values only against 14.4 for the two-scan
                                 # kernel; B=32, N=512 8.8 against 5.9; B=8, N=1024 8.9 against 6.4


def nn_bidir_raw(a, b, a_cf=False, b_cf=False, want_idx=True, two_scan=None):
    """Both directions: (dA [B,N], iA [B,N], dB [B,M], iB [B,M]); iA/iB are None with want_idx=False. Two kernels
    with bit-identical results: pc3d_nn_bidir_shared_f32 (one evaluation of every distance feeds both directions; a
    scan + a fold launch) and pc3d_nn_bidir_f32 (one scan per direction in one launch). two_scan=None picks by size
    (small problems are launch-bound: one launch wins), True / False force one."""
    ap, abs_, aps, acs, B, N = _pts(a, a_cf, "a")
    bp, bbs, bps, bcs, B2, M = _pts(b, b_cf, "b")
    if B != B2:
        raise ValueError("a and b must have the same batch dimension")
    if N < 1 or M < 1:
        raise ValueError("empty point set")
    dev = a.device
    dA = torch.empty((B, N), dtype=torch.float32, device=dev)
    dB = torch.empty((B, M), dtype=torch.float32, device=dev)
    iA = torch.empty((B, N), dtype=torch.int32, device=dev) if want_idx else None
    iB = torch.empty((B, M), dtype=torch.int32, device=dev) if want_idx else None
    if two_scan is None:
        two_scan = B * N * M < NN_SHARED_MIN_PAIRS
    with torch.cuda.device(dev):
        if two_scan:
            _lib.call("pc3d_nn_bidir_f32", ap, abs_, aps, acs, bp, bbs, bps, bcs, B, N, M,
                      dA.data_ptr(), _ptr(iA), dB.data_ptr(), _ptr(iB), _stream())
        else:
            need = int(_lib.load().pc3d_nn_bidir_shared_ws_bytes(B, N, M))
            ws = _nn_workspace(dev, need)
            _lib.call("pc3d_nn_bidir_shared_f32", ap, abs_, aps, acs, bp, bbs, bps, bcs, B, N, M,
                      dA.data_ptr(), _ptr(iA), dB.data_ptr(), _ptr(iB), ws.data_ptr(), ws.numel(), _stream())
    return dA, iA, dB, iB


_OPS = {"mean": 0, "max": 1, "sum": 2}


def rowreduce(x, op="mean", sqrt=False):
    """[B,N] f32 -> [B]; op in {mean,max,sum}; sqrt=True applies sqrt(max(x,0)) per element first."""
    _check(x, "x")
    x = x.contiguous()
    B, N = x.shape
    out = torch.empty((B,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("pc3d_rowreduce_f32", x.data_ptr(), B, N, _OPS[op], 1 if sqrt else 0, out.data_ptr(), _stream())
    return out


def _nn_bwd(a, a_cf, b, b_cf, iA, wA, sA, iB, wB, sB, need_a, need_b, deterministic):
    ap, abs_, aps, acs, B, N = _pts(a, a_cf, "a")
    bp, bbs, bps, bcs, _, M = _pts(b, b_cf, "b")
    ga = torch.empty_like(a, memory_format=torch.contiguous_format) if need_a else None
    gb = torch.empty_like(b, memory_format=torch.contiguous_format) if need_b else None

    def gview(g, cf):
        if g is None:
            return (0, 0, 0, 0)
        p, bs, ps, cs, _, _ = _pts(g, cf, "grad")
        return (p, bs, ps, cs)

    def wview(w):
        if w is None:
            return (0, 0, 0)
        _check(w, "upstream grad")
        return (w.data_ptr(), w.stride(0), w.stride(1))

    with torch.cuda.device(a.device):
        _lib.call("pc3d_nn_bwd_f32", ap, abs_, aps, acs, bp, bbs, bps, bcs, B, N, M,
                  _ptr(iA), *wview(wA), float(sA), _ptr(iB), *wview(wB), float(sB),
                  *gview(ga, a_cf), *gview(gb, b_cf), _det(deterministic), _stream())
    return ga, gb


_CONST_VECS = {}


def const_vec(device, n, value):
    """A cached [n] fp32 tensor filled with `value` (never written again): the fixed upstream gradient of a per-sample
    loss term. Created once per (device, n, value) — no fill launch inside an attack loop."""
    device = torch.device(device)
    key = (device.type, device.index, int(n), float(np.float32(value)))
    t = _CONST_VECS.get(key)
    if t is None and torch.cuda.is_current_stream_capturing():
        # a miss inside a capture: the fill below is only RECORDED, so the tensor is valid inside this graph's replays and
        # nowhere else — it belongs to the capture (keep-alive list) and never enters the process-wide cache
        t = torch.full((int(n),), float(np.float32(value)), dtype=torch.float32, device=device)
        _graphed.note_captured(t)
        return t
    if t is None:
        if len(_CONST_VECS) >= 256:
            _CONST_VECS.pop(next(iter(_CONST_VECS)))
        t = torch.full((int(n),), float(np.float32(value)), dtype=torch.float32, device=device)
        _CONST_VECS[key] = t
    _graphed.note_captured(t)
    return t


def scaled_gvec(up, factor, n, device, mean=True):
    """Upstream gradient of a per-sample term that enters a loss as mean_b(term_b) * factor * up: `up` a Python / numpy
    float (the same for every sample: a cached constant vector, no launch) or a [n] tensor (per-sample weights). The
    arithmetic follows autograd's order for `(term.mean() * factor) * up`: ((up * factor) / n)."""
    if torch.is_tensor(up):
        g = up * float(factor) if factor != 1.0 else up           # (factor 1, mean False: `up` itself, no launch)
        return g / float(n) if mean else g
    v = np.float32(up) * np.float32(factor) if factor != 1.0 else np.float32(up)
    if mean:
        v = v / np.float32(n)
    return const_vec(device, n, v)


# ------------------------------------------------------------------------------------------------------
# differentiable ops
# ------------------------------------------------------------------------------------------------------
class _NNBidirFn(torch.autograd.Function):
    """(a, b) -> (dA, dB, iA, iB): per-point squared NN distances both ways, differentiable in a and b."""

    @staticmethod
    def forward(ctx, a, b, a_cf, b_cf, deterministic):
        dA, iA, dB, iB = nn_bidir_raw(a, b, a_cf, b_cf)
        ctx.save_for_backward(a, b, iA, iB)
        ctx.cfg = (a_cf, b_cf, deterministic)
        ctx.mark_non_differentiable(iA, iB)
        ctx.set_materialize_grads(False)
        return dA, dB, iA, iB

    @staticmethod
    def backward(ctx, gA, gB, _gia, _gib):
        a, b, iA, iB = ctx.saved_tensors
        a_cf, b_cf, det = ctx.cfg
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if gA is None and gB is None:
            return None, None, None, None, None
        ga, gb = _nn_bwd(a, a_cf, b, b_cf, iA if gA is not None else None, gA, 1.0,
                         iB if gB is not None else None, gB, 1.0, need_a, need_b, det)
        return ga, gb, None, None, None


def nn_bidir(a, b, a_cf=False, b_cf=False, deterministic=None):
    """Differentiable bidirectional NN: returns (dA [B,N], dB [B,M], iA, iB)."""
    return _NNBidirFn.apply(a, b, a_cf, b_cf, deterministic)


class _SetDistFn(torch.autograd.Function):
    """Fused (a,b) -> (loss_a2b [B], loss_b2a [B]) with reduce in {mean (Chamfer), max (Hausdorff)} over
    squared NN distances — distance.py:40-50 (ChamferDistance) / :58-70 (HausdorffDistance)."""

    @staticmethod
    def forward(ctx, a, b, a_cf, b_cf, reduce, deterministic, gvec=None):
        dA, iA, dB, iB = nn_bidir_raw(a, b, a_cf, b_cf)
        l1 = rowreduce(dA, reduce)
        l2 = rowreduce(dB, reduce)
        if reduce == "max":
            # gradient flows to the arg-max point only (torch.max backward); keep one-hot weights
            ctx.hot = (dA == l1[:, None]), (dB == l2[:, None])
        ctx.save_for_backward(a, b, iA, iB, gvec)
        ctx.cfg = (a_cf, b_cf, reduce, deterministic, dA.shape[1], dB.shape[1])
        ctx.set_materialize_grads(False)
        return l1, l2

    @staticmethod
    def backward(ctx, g1, g2):
        a, b, iA, iB, gvec = ctx.saved_tensors
        a_cf, b_cf, reduce, det, N, M = ctx.cfg
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if (g1 is None and g2 is None) or not (need_a or need_b):
            return None, None, None, None, None, None, None
        if gvec is not None:               # the caller fixed d loss / d term per sample: the incoming values are not read
            g1 = gvec if g1 is not None else None
            g2 = gvec if g2 is not None else None
        if reduce == "mean":
            wA = None if g1 is None else g1.contiguous().view(-1, 1).expand(-1, N)  # stride (1,0): no copy
            wB = None if g2 is None else g2.contiguous().view(-1, 1).expand(-1, M)
            sA, sB = 1.0 / N, 1.0 / M
        else:
            hotA, hotB = ctx.hot
            # first arg-max only, like torch.max(dim) backward
            def first_hot(h, g):
                if g is None:
                    return None
                first = (h.int().cumsum(1) == 1) & h
                return first.float() * g.view(-1, 1)
            wA, wB = first_hot(hotA, g1), first_hot(hotB, g2)
            sA = sB = 1.0
        ga, gb = _nn_bwd(a, a_cf, b, b_cf, iA if wA is not None else None, wA, sA,
                         iB if wB is not None else None, wB, sB, need_a, need_b, det)
        return ga, gb, None, None, None, None, None


class _SetDistOneFn(torch.autograd.Function):
    """ONE direction of _SetDistFn: loss [B] = reduce_i min_j |a_i - b_j|^2 (a -> b). The distance functors of the attacks
    default to method='adv2ori' (attack/CW/CW_utils/dist_utils.py:40,62-63): the other direction's search, its reduction
    and — in the backward — its scatter are not needed at all."""

    @staticmethod
    def forward(ctx, a, b, a_cf, b_cf, reduce, deterministic, gvec=None):
        dA, iA = nn_raw(a, b, a_cf, b_cf)
        l1 = rowreduce(dA, reduce)
        if reduce == "max":
            ctx.hot = dA == l1[:, None]
        ctx.save_for_backward(a, b, iA, gvec)
        ctx.cfg = (a_cf, b_cf, reduce, deterministic, dA.shape[1])
        return l1

    @staticmethod
    def backward(ctx, g1):
        a, b, iA, gvec = ctx.saved_tensors
        a_cf, b_cf, reduce, det, N = ctx.cfg
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_a or need_b):
            return None, None, None, None, None, None, None
        if gvec is not None:
            g1 = gvec
        if reduce == "mean":
            wA, sA = g1.contiguous().view(-1, 1).expand(-1, N), 1.0 / N
        else:
            h = ctx.hot
            wA, sA = ((h.int().cumsum(1) == 1) & h).float() * g1.view(-1, 1), 1.0
        ga, gb = _nn_bwd(a, a_cf, b, b_cf, iA, wA, sA, None, None, 1.0, need_a, need_b, det)
        return ga, gb, None, None, None, None, None


def set_distance_one(a, b, reduce="mean", a_cf=False, b_cf=False, deterministic=None, gvec=None):
    """loss_a2b [B] only: squared Chamfer (reduce='mean') / Hausdorff ('max') term from a to b, one search. gvec [B]:
    the backward uses it as d loss / d term and ignores the incoming gradient (see scaled_gvec)."""
    return _SetDistOneFn.apply(a, b, a_cf, b_cf, reduce, deterministic, gvec)


def set_distance(a, b, reduce="mean", a_cf=False, b_cf=False, deterministic=None, gvec=None):
    """(loss_a2b [B], loss_b2a [B]) — squared Chamfer (reduce='mean') or Hausdorff (reduce='max') terms. gvec: as in
    set_distance_one, for BOTH outputs."""
    return _SetDistFn.apply(a, b, a_cf, b_cf, reduce, deterministic, gvec)


# ------------------------------------------------------------------------------------------------------
# K8: fused PointNet per-point MLP + max-pool
# ------------------------------------------------------------------------------------------------------
_PM_TILE = None


def _pm_tile():
    global _PM_TILE
    if _PM_TILE is None:
        _PM_TILE = _lib.load().pc3d_pointmlp3_tile_points()
    return _PM_TILE


def pointmlp3_max_fwd_raw(x, weights, relu_last, T=None, x_cf=True, fold=True, want_masks=False, T_head=None):
    """x [B,3,N] (x_cf) or [B,N,3]; weights = (W1[64,3], b1, W2[128,64], b2, W3[C3,128], b3) with eval-BN folded.
    Returns (pooled [B,C3] f32, argidx [B,C3] i32) and, with want_masks, a third item (mask1 [B,N] i64, mask2 [B,N,4]
    i32): the per-point ReLU decisions of layers 1 and 2 as bit masks, which pointmlp3_max_bwd_raw consumes.
    T_head = (h [B,K], W [9,K], b [9]): the input transform T = h @ W.T + b is computed in the launch's prologue
    (no launch of its own) and returned as a last extra item [B,9]."""
    xp, xbs, xps, xcs, B, N = _pts(x, x_cf, "x")
    W1, b1, W2, b2, W3, b3 = weights[:6]
    for w in weights:
        _check(w, "weight")
        if not w.is_contiguous():
            raise ValueError("pointmlp3 weights must be contiguous")
    C1, C2, C3 = W1.shape[0], W2.shape[0], W3.shape[0]
    if T is not None:
        _check(T, "T")
        T = T.contiguous()
    ntiles = (N + _pm_tile() - 1) // _pm_tile()
    dev = x.device
    part_val = torch.empty((B, ntiles, C3), dtype=torch.float32, device=dev)
    part_idx = torch.empty((B, ntiles, C3), dtype=torch.int32, device=dev)
    pooled = torch.empty((B, C3), dtype=torch.float32, device=dev)
    argidx = torch.empty((B, C3), dtype=torch.int32, device=dev)
    masks = None
    if want_masks:
        masks = (torch.empty((B, N), dtype=torch.int64, device=dev), torch.empty((B, N, 4), dtype=torch.int32, device=dev))
    tail = (W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr(),
            C1, C2, C3, 1 if relu_last else 0, part_val.data_ptr(), part_idx.data_ptr(),
            pooled.data_ptr() if fold else 0, argidx.data_ptr() if fold else 0,
            masks[0].data_ptr() if masks else 0, masks[1].data_ptr() if masks else 0, _stream())
    T_out = None
    with torch.cuda.device(dev):
        if T_head is not None:
            if T is not None:
                raise ValueError("pointmlp3_max_fwd_raw: give T or T_head, not both")
            h, Wt, bt = T_head
            if h.shape[0] != B or Wt.shape != (9, h.shape[1]) or bt.numel() != 9 or not (h.is_contiguous() and Wt.is_contiguous()):
                raise ValueError("pointmlp3_max_fwd_raw: T_head = (h [B,K], W [9,K], b [9]) contiguous expected")
            T_out = torch.empty((B, 9), dtype=torch.float32, device=dev)
            _lib.call("pc3d_pointmlp3_max_fwd_th_f32", xp, xbs, xps, xcs, B, N, h.data_ptr(), Wt.data_ptr(), bt.data_ptr(),
                      h.shape[1], T_out.data_ptr(), *tail)
        else:
            _lib.call("pc3d_pointmlp3_max_fwd_f32", xp, xbs, xps, xcs, B, N, _ptr(T), *tail)
    if not fold:
        return (part_val, part_idx) if T_out is None else (part_val, part_idx, T_out)
    res = (pooled, argidx, masks) if want_masks else (pooled, argidx)
    return res if T_out is None else res + (T_out,)


def pointmlp3_max_bwd_raw(x, weights, argidx, g_pooled, masks, T=None, x_cf=True, out=None, accumulate=False,
                          want_gT=False):
    """Gradient w.r.t. the tower input (T None) or w.r.t. the raw points through x' = x @ T (T given); same layout
    as x. masks: the (mask1, mask2) pair of the forward launch on the same x/T (want_masks=True).
    want_gT: also return the per-tile partials [B, ntiles, 16] of dL/dT. out/accumulate: add into `out`."""
    xp, xbs, xps, xcs, B, N = _pts(x, x_cf, "x")
    W1, b1, W2, b2, W3, b3 = weights[:6]
    W2T = weights[6] if len(weights) > 6 else W2.t().contiguous()
    C1, C2, C3 = W1.shape[0], W2.shape[0], W3.shape[0]
    _check(g_pooled, "g_pooled")
    g_pooled = g_pooled.contiguous()
    m1, m2 = masks
    if m1.shape != (B, N) or m1.dtype != torch.int64 or m2.shape != (B, N, 4) or m2.dtype != torch.int32 \
            or not (m1.is_contiguous() and m2.is_contiguous()):
        raise ValueError("pointmlp3_max_bwd_raw: masks must be the (int64 [B,N], int32 [B,N,4]) pair of the forward launch")
    gx = out if out is not None else torch.empty((B, 3, N) if x_cf else (B, N, 3), dtype=torch.float32,
                                                 device=x.device)
    gp, gbs, gps, gcs, _, _ = _pts(gx, x_cf, "grad_x")
    part_gT = None
    if want_gT:
        bt = _lib.load().pc3d_pointmlp3_bwd_tile_points()
        part_gT = torch.empty((B, (N + bt - 1) // bt, 16), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("pc3d_pointmlp3_max_bwd_f32", xp, xbs, xps, xcs, B, N, _ptr(T),
                  W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), W2T.data_ptr(),
                  C1, C2, C3, argidx.data_ptr(), m1.data_ptr(), m2.data_ptr(), g_pooled.data_ptr(), gp, gbs, gps, gcs,
                  _ptr(part_gT),
                  1 if accumulate else 0, _stream())
    return (gx, part_gT) if want_gT else gx


class _PointMLP3MaxFn(torch.autograd.Function):
    """x [B,3,N] -> pooled [B,C3] through the fused tower; differentiable in x only (frozen weights)."""

    @staticmethod
    def forward(ctx, x, relu_last, W1, b1, W2, b2, W3, b3, W2T):
        weights = (W1, b1, W2, b2, W3, b3, W2T)
        pooled, argidx, masks = pointmlp3_max_fwd_raw(x, weights, relu_last, want_masks=True)
        ctx.save_for_backward(x, argidx, pooled, masks[0], masks[1], *weights)
        ctx.relu_last = relu_last
        return pooled

    @staticmethod
    def backward(ctx, g):
        x, argidx, pooled, m1, m2, *weights = ctx.saved_tensors
        if ctx.relu_last:
            g = g * (pooled > 0)
        gx = pointmlp3_max_bwd_raw(x, tuple(weights), argidx, g, (m1, m2))
        return (gx,) + (None,) * 8


def pointmlp3_max(x, weights, relu_last):
    """weights: (W1,b1,W2,b2,W3,b3[,W2T]) with eval-BN folded; W2T is derived when absent."""
    if len(weights) == 6:
        weights = tuple(weights) + (weights[2].t().contiguous(),)
    return _PointMLP3MaxFn.apply(x, relu_last, *weights)


# ------------------------------------------------------------------------------------------------------
# K9 / K10 / pairwise
# ------------------------------------------------------------------------------------------------------
def _pv(t, cf, name):
    """(ptr, bs, ps, cs) or four zeros for None."""
    if t is None:
        return (0, 0, 0, 0)
    p, bs, ps, cs, _, _ = _pts(t, cf, name)
    return (p, bs, ps, cs)


def clip(pc, ori, normal=None, budget=0.0, mode="point", cf=True, out=None):
    """clip_utils.py semantics in one launch. mode 'point': optional inner-point projection (normal given) then
    per-point L2 clip (budget<=0: none). mode 'global': ClipPointsL2. Tensors [B,3,K] (cf) or [B,K,3]."""
    _, _, _, _, B, K = _pts(pc, cf, "pc")
    if out is None:
        out = torch.empty(pc.shape, dtype=torch.float32, device=pc.device)
    with torch.cuda.device(pc.device):
        _lib.call("pc3d_clip_f32", *_pv(pc, cf, "pc"), *_pv(ori, cf, "ori"), *_pv(normal, cf, "normal"),
                  B, K, 0 if mode == "point" else 1, float(budget), *_pv(out, cf, "out"), _stream())
    return out


def adam_clip_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, ori=None, normal=None, budget=0.0,
                   cf=True, g2=None):
    """In-place Adam step on p (+ fused per-point clip/projection against ori). `step` is an int (host) or a
    1-element int32 GPU tensor holding t. g2: a second gradient tensor, summed with g inside the launch."""
    _, _, _, _, B, K = _pts(p, cf, "p")
    if m.stride() != p.stride() or v.stride() != p.stride():
        raise ValueError("adam state must share the parameter's strides")
    if isinstance(step, torch.Tensor):
        step_dev, step_host = step.data_ptr(), 0
    else:
        step_dev, step_host = 0, int(step)
    with torch.cuda.device(p.device):
        _lib.call("pc3d_adam_clip_step_f32", *_pv(p, cf, "p"), *_pv(g, cf, "g"), *_pv(g2, cf, "g2"), m.data_ptr(), v.data_ptr(),
                  *_pv(ori, cf, "ori"), *_pv(normal, cf, "normal"), B, K, float(lr), float(betas[0]),
                  float(betas[1]), float(eps), float(budget), step_dev, step_host, _stream())
    return p


def i32_add(ctr, delta=1):
    with torch.cuda.device(ctr.device):
        _lib.call("pc3d_i32_add", ctr.data_ptr(), int(delta), _stream())


def pairwise(x, y, x_cf=False, y_cf=False, euclid=False):
    """Dense [B,N,M] (squared) distance matrix, direct-difference fp32."""
    xp, xbs, xps, xcs, B, N = _pts(x, x_cf, "x")
    yp, ybs, yps, ycs, B2, M = _pts(y, y_cf, "y")
    if B != B2:
        raise ValueError("x and y must have the same batch dimension")
    out = torch.empty((B, N, M), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("pc3d_pairwise_f32", xp, xbs, xps, xcs, yp, ybs, yps, ycs, B, N, M, 1 if euclid else 0,
                  out.data_ptr(), _stream())
    return out


# ------------------------------------------------------------------------------------------------------
# K2/K4: K nearest neighbours
# ------------------------------------------------------------------------------------------------------
# Hints for the K-NN searches (pc3d_knn_hint_f32): the index tensor the LAST search of the same shape returned. An attack
# loop repeats every search on points that moved by ~1e-2; last iteration's neighbours bound this iteration's K-th distance,
# and the kernel then inserts K + a few candidates per query instead of K ln(M / K). The result never depends on the hint
# (a bad one is detected and ignored), so a stale or foreign entry — another call site with the same shape — only costs time.
# Inside a hipGraph capture the hint is the output buffer itself: it is private to the graph and holds the previous
# replay's result (garbage before the first replay: detected).
KNN_HINTS = os.environ.get("PC3D_KNN_HINTS", "1") == "1"
_knn_hints = {}


KNN_HINT_MIN_M = 512      # below it the seed sort the hint replaces is all there is to save (M = 256: 16.5 us unhinted, 17.4-18.6 hinted)


def _knn_hint(key, out, M=None):
    """The hint to launch with for this search (an int32 tensor shaped like `out`, `out` itself inside a capture, or None)."""
    if not KNN_HINTS or (M is not None and M < KNN_HINT_MIN_M):
        return None
    if torch.cuda.is_current_stream_capturing():
        return out
    h = _knn_hints.get(key)
    _knn_hints[key] = out
    if len(_knn_hints) > 64:
        _knn_hints.pop(next(iter(_knn_hints)))
    return h if h is not None and h.shape == out.shape and h.device == out.device else None


def knn_raw(q, r, K, q_cf=False, r_cf=False):
    """(dists [B,N,K] f32 ascending, idx [B,N,K] i32)."""
    qp, qbs, qps, qcs, B, N = _pts(q, q_cf, "q")
    rp, rbs, rps, rcs, B2, M = _pts(r, r_cf, "r")
    if B != B2:
        raise ValueError("q and r must have the same batch dimension")
    if not (1 <= K <= min(64, M)):
        raise ValueError(f"K={K} out of range [1, min(64, M={M})]")
    if K == 1:  # the dedicated nearest-neighbour kernel (split-M, several queries per lane) is faster
        d1, i1 = nn_raw(q, r, q_cf, r_cf)
        return d1.unsqueeze(-1), i1.unsqueeze(-1)
    d = torch.empty((B, N, K), dtype=torch.float32, device=q.device)
    i = torch.empty((B, N, K), dtype=torch.int32, device=q.device)
    h = _knn_hint(("knn", B, N, M, K, q.device), i, M)
    with torch.cuda.device(q.device):
        _lib.call("pc3d_knn_hint_f32", qp, qbs, qps, qcs, rp, rbs, rps, rcs, B, N, M, K, d.data_ptr(), i.data_ptr(),
                  h.data_ptr() if h is not None else 0, _stream())
    return d, i


def knn_graph(pts, k, cf=False):
    """(idx [B,N,k+1] self first, idx[:, :, 1:], idx[:, :, :k]) — int32, contiguous — of the self-kNN graph of pts [B,N,3]:
    the graph and the two views CurveNet's blocks gather through, written by ONE launch (pc3d_knn_graph_i32)."""
    p, bs, ps, cs, B, N = _pts(pts, cf, "pts")
    K = int(k) + 1
    if not (2 <= K <= min(64, N)):
        raise ValueError(f"knn_graph: k + 1 = {K} out of range [2, min(64, N={N})]")
    idx = torch.empty((B, N, K), dtype=torch.int32, device=pts.device)
    noself = torch.empty((B, N, K - 1), dtype=torch.int32, device=pts.device)
    first = torch.empty((B, N, K - 1), dtype=torch.int32, device=pts.device)
    h = _knn_hint(("graph", B, N, K, pts.device), idx, N)
    with torch.cuda.device(pts.device):
        _lib.call("pc3d_knn_graph_hint_i32", p, bs, ps, cs, B, N, K, idx.data_ptr(), noself.data_ptr(), first.data_ptr(), K - 1,
                  h.data_ptr() if h is not None else 0, _stream())
    return idx, noself, first


class _KnnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, r, K, q_cf, r_cf, deterministic, idx64=False):
        if idx64 and K == 1:                   # int64 indices from the search launch itself (no conversion launch)
            d1, i1, i64 = nn_raw(q, r, q_cf, r_cf, want_i64=True)
            d, i, iout = d1.unsqueeze(-1), i1.unsqueeze(-1), i64.unsqueeze(-1)
        else:
            d, i = knn_raw(q, r, K, q_cf, r_cf)
            iout = i.long() if idx64 else i
        ctx.save_for_backward(q, r, i)
        ctx.cfg = (K, q_cf, r_cf, deterministic)
        ctx.mark_non_differentiable(iout)
        ctx.set_materialize_grads(False)      # no zero tensor for the index output's "gradient" (a fill launch per call)
        return d, iout

    @staticmethod
    def backward(ctx, gd, _gi):
        if gd is None:
            return None, None, None, None, None, None, None
        q, r, idx = ctx.saved_tensors
        K, q_cf, r_cf, det = ctx.cfg
        need_q, need_r = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gd = gd.contiguous()
        idx = idx.contiguous()
        _, _, _, _, B, N = _pts(q, q_cf, "q")
        _, _, _, _, _, M = _pts(r, r_cf, "r")
        gq = torch.empty(q.shape, dtype=torch.float32, device=q.device) if need_q else None
        gr = torch.empty(r.shape, dtype=torch.float32, device=r.device) if need_r else None
        ws = torch.empty((B, N * K, 3), dtype=torch.float32, device=q.device) if (_det(det) and need_r) else None
        with torch.cuda.device(q.device):
            _lib.call("pc3d_knn_bwd_f32", *_pv(q, q_cf, "q"), *_pv(r, r_cf, "r"), B, N, M, K, idx.data_ptr(),
                      gd.data_ptr(), *_pv(gq, q_cf, "gq"), *_pv(gr, r_cf, "gr"), _det(det), _ptr(ws), _stream())
        return gq, gr, None, None, None, None, None


def knn(q, r, K, q_cf=False, r_cf=False, deterministic=None, idx64=False):
    """Differentiable K-NN: (dists [B,N,K], idx [B,N,K] int32 — int64 with idx64). For self-kNN pass the same tensor twice
    (autograd sums the two gradient roles)."""
    return _KnnFn.apply(q, r, K, q_cf, r_cf, deterministic, idx64)


class _KnnOutlierFn(torch.autograd.Function):
    """KNNDist (attack/CW/CW_utils/dist_utils.py:112-160) per sample: self-kNN search, the outlier-masked mean of the mean
    neighbour distances, and its gradient to the points — search + loss launch forward, q / edges / scatter backward,
    instead of ~15 ATen launches each way around the search."""

    @staticmethod
    def forward(ctx, pc, k, alpha, cf, gvec=None):
        d, idx = knn_raw(pc, pc, k + 1, cf, cf)
        B, N, K1 = d.shape
        loss = torch.empty((B,), dtype=torch.float32, device=pc.device)
        w = torch.empty_like(d)
        with torch.cuda.device(pc.device):
            _lib.call("pc3d_knn_outlier_loss_f32", d.data_ptr(), B, N, K1, float(alpha), loss.data_ptr(), w.data_ptr(), _stream())
        ctx.save_for_backward(pc, idx, w, gvec)
        ctx.cf = cf
        return loss

    @staticmethod
    def backward(ctx, g):
        pc, idx, w, gvec = ctx.saved_tensors
        B, N, K1 = w.shape
        g = gvec if gvec is not None else g.contiguous()
        grad = torch.empty(pc.shape, dtype=torch.float32, device=pc.device)
        det = _det()
        ws = torch.empty((B, N * K1, 3), dtype=torch.float32, device=pc.device) if det else None
        with torch.cuda.device(pc.device):
            _lib.call("pc3d_knn_self_bwd_f32", *_pv(pc, ctx.cf, "pc"), B, N, K1, idx.data_ptr(), w.data_ptr(), g.data_ptr(),
                      *_pv(grad, ctx.cf, "grad"), det, _ptr(ws), _stream())
        return grad, None, None, None, None


def knn_outlier_loss(pc, k=5, alpha=1.05, cf=False, gvec=None):
    """Per-sample kNN-distance outlier penalty [B] of pc ([B,N,3], or [B,3,N] with cf); differentiable in pc. gvec [B]:
    fixed d loss / d term (the backward then ignores the incoming gradient, see scaled_gvec)."""
    _check(pc, "pc")
    return _KnnOutlierFn.apply(pc, int(k), float(alpha), bool(cf), gvec)


# ------------------------------------------------------------------------------------------------------
# classifier heads
# ------------------------------------------------------------------------------------------------------
def linear(X, W, bias=None, relu=False, gate=None, parts=1, out=None, slope=0.0, gate_slope=0.0):
    """Y = epi(X @ W.T + bias). X [B,K] (or [B,parts,K] summed over parts), W [O,K] contiguous fp32. relu: (Leaky)ReLU
    with `slope` in the epilogue; gate [B,O]: Y = gate > 0 ? Y : gate_slope * Y afterwards (a saved activation's mask)."""
    _check(X, "X")
    _check(W, "W")
    if parts == 1:
        B, K = X.shape
        ldx = X.stride(0)
        if X.stride(1) != 1:
            raise ValueError("linear: X rows must be contiguous")
    else:
        B, P, K = X.shape
        if not X.is_contiguous() or P != parts:
            raise ValueError("linear: partial slabs must be a contiguous [B,parts,K] tensor")
        ldx = P * K
    O = W.shape[0]
    if W.shape[1] != K or not W.is_contiguous():
        raise ValueError(f"linear: W must be contiguous [O,{K}], got {tuple(W.shape)}")
    Y = out if out is not None else torch.empty((B, O), dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        _lib.call("pc3d_linear_f32", X.data_ptr(), ldx, parts, B, K, W.data_ptr(), _ptr(bias), O, 1 if relu else 0,
                  float(slope), _ptr(gate), gate.stride(0) if gate is not None else 0, float(gate_slope), Y.data_ptr(),
                  Y.stride(0), _stream())
    return Y


def linear_pre(parts, J, Wp, gate_pre, W, gate=None):
    """Y = gate(X @ W.T) with X[b,k] = gate_pre[b,k] > 0 ? sum_{j<J} (sum_p parts[b,p,j]) Wp[j,k] : 0 in ONE launch
    (pc3d_linear_pre_f32): parts [B,P,Jp] contiguous, Wp [J,K], gate_pre [B,K], W [O,K], gate [B,O] or None."""
    _check(parts, "parts"), _check(Wp, "Wp"), _check(gate_pre, "gate_pre"), _check(W, "W")
    B, P, Jp = parts.shape
    K, O = Wp.shape[1], W.shape[0]
    if not (parts.is_contiguous() and Wp.is_contiguous() and W.is_contiguous()) or Wp.shape[0] < J or W.shape[1] != K \
            or gate_pre.shape != (B, K) or gate_pre.stride(1) != 1 or K % 16:
        raise ValueError("linear_pre: parts [B,P,Jp], Wp [>=J,K], gate_pre [B,K], W [O,K] (K % 16 == 0) expected")
    Y = torch.empty((B, O), dtype=torch.float32, device=parts.device)
    with torch.cuda.device(parts.device):
        _lib.call("pc3d_linear_pre_f32", parts.data_ptr(), P, Jp, int(J), Wp.data_ptr(), gate_pre.data_ptr(),
                  gate_pre.stride(0), B, K, W.data_ptr(), O, _ptr(gate), gate.stride(0) if gate is not None else 0,
                  Y.data_ptr(), Y.stride(0), _stream())
    return Y


LOSS_KINDS = {"untargeted_logits": 0, "logits": 1, "cross_entropy": 2}


def cls_loss(logits, target, kind, kappa=0.0, scale=1.0, want_grad=True, pred_out=None):
    """(logp [B,k], pred [B] int64, loss [B], g_logits [B,k] or None) — log_softmax + adversarial loss, one launch.
    pred_out: a persistent int64 [B] tensor to receive the prediction."""
    _check(logits, "logits")
    B, ncls = logits.shape
    dev = logits.device
    logp = torch.empty((B, ncls), dtype=torch.float32, device=dev)
    if pred_out is not None and (pred_out.dtype != torch.int64 or pred_out.shape != (B,) or not pred_out.is_contiguous()):
        raise TypeError("cls_loss: pred_out must be a contiguous int64 [B] tensor")
    pred = pred_out if pred_out is not None else torch.empty((B,), dtype=torch.int64, device=dev)
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    g = torch.empty((B, ncls), dtype=torch.float32, device=dev) if want_grad else None
    if target.dtype != torch.int64 or not target.is_cuda:
        raise TypeError("cls_loss: target must be an int64 GPU tensor")
    with torch.cuda.device(dev):
        _lib.call("pc3d_cls_loss_f32", logits.data_ptr(), logits.stride(0), B, ncls, target.data_ptr(),
                  LOSS_KINDS[kind] if isinstance(kind, str) else int(kind), float(kappa), float(scale),
                  logp.data_ptr(), pred.data_ptr(), loss.data_ptr(), _ptr(g), _stream())
    return logp, pred, loss, g


def cls_tail(c2, w3, b3, target, kind, kappa=0.0, scale=1.0, pred_out=None, step=None, want_logp=True):
    """fc3 + cls_loss + fc3-backward in one launch (see pc3d_cls_tail_f32): (logp or None, pred, loss, g_c2).
    pred_out: persistent int64 [B] to write the prediction into; step: int32 [1] device word to advance."""
    _check(c2, "c2"), _check(w3, "w3"), _check(b3, "b3")
    B, K2 = c2.shape
    ncls = w3.shape[0]
    dev = c2.device
    if target.dtype != torch.int64 or not target.is_cuda:
        raise TypeError("cls_tail: target must be an int64 GPU tensor")
    logp = torch.empty((B, ncls), dtype=torch.float32, device=dev) if want_logp else None
    pred = pred_out if pred_out is not None else torch.empty((B,), dtype=torch.int64, device=dev)
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    g_c2 = torch.empty((B, K2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pc3d_cls_tail_f32", c2.data_ptr(), B, K2, w3.data_ptr(), b3.data_ptr(), ncls, target.data_ptr(),
                  LOSS_KINDS[kind] if isinstance(kind, str) else int(kind), float(kappa), float(scale), _ptr(logp),
                  pred.data_ptr(), loss.data_ptr(), g_c2.data_ptr(), _ptr(step), _stream())
    return logp, pred, loss, g_c2


# ------------------------------------------------------------------------------------------------------
# CW-family loop: bookkeeping + fused update
# ------------------------------------------------------------------------------------------------------
CW_UPDATE_MAX_POINTS = 8192     # pc3d_cw_update_f32 keeps a sample's points in registers (8 per thread x 1024 threads)


def cw_update(adv, ori, pred, label, untarget, bestdist, bestscore, o_bestdist, o_bestscore, o_bestattack, g, m, v,
              step, lr, budget, input_val=None, dist_val=None, dist_kind=0, w=None, nn_idx=None,
              betas=(0.9, 0.999), eps=1e-8, cf=True):
    """cw_bookkeep + cw_step as one launch (see pc3d_cw_update_f32). `step` (int32 device word or int) is only READ."""
    _, _, _, _, B, K = _pts(adv, cf, "adv")
    for nm, t in (("o_bestattack", o_bestattack), ("m", m), ("v", v), ("input_val", input_val)):
        if t is not None and (t.shape != adv.shape or t.stride() != adv.stride()):
            raise ValueError(f"cw_update: {nm} must share adv's shape and strides")
    if isinstance(step, torch.Tensor):
        step_dev, step_host = step.data_ptr(), 0
    else:
        step_dev, step_host = 0, int(step)
    with torch.cuda.device(adv.device):
        _lib.call("pc3d_cw_update_f32", *_pv(adv, cf, "adv"), *_pv(ori, cf, "ori"), B, K, pred.data_ptr(),
                  label.data_ptr(), 1 if untarget else 0, bestdist.data_ptr(), bestscore.data_ptr(),
                  o_bestdist.data_ptr(), o_bestscore.data_ptr(), o_bestattack.data_ptr(), _ptr(input_val),
                  _ptr(dist_val), *_pv(g, cf, "g"), m.data_ptr(), v.data_ptr(), float(lr), float(betas[0]),
                  float(betas[1]), float(eps), float(budget), step_dev, step_host, int(dist_kind), _ptr(w),
                  _ptr(nn_idx), _stream())
    return adv


def cw_bookkeep(adv, ori, pred, label, untarget, bestdist, bestscore, o_bestdist, o_bestscore, o_bestattack,
                input_val=None, dist_val=None, step=None, cf=True):
    """In-place update of the best-attack state for one iteration (see pc3d_cw_bookkeep_f32)."""
    _, _, _, _, B, K = _pts(adv, cf, "adv")
    with torch.cuda.device(adv.device):
        _lib.call("pc3d_cw_bookkeep_f32", *_pv(adv, cf, "adv"), *_pv(ori, cf, "ori"), B, K, pred.data_ptr(),
                  label.data_ptr(), 1 if untarget else 0, bestdist.data_ptr(), bestscore.data_ptr(),
                  o_bestdist.data_ptr(), o_bestscore.data_ptr(), *_pv(o_bestattack, cf, "o_bestattack"),
                  *_pv(input_val, cf, "input_val"), _ptr(dist_val), _ptr(step), _stream())


def cw_step(p, g, m, v, step, lr, ori, budget, dist_kind=0, w=None, l2norm=None, nn_idx=None,
            betas=(0.9, 0.999), eps=1e-8, cf=True):
    """Fused total-gradient + Adam + clip (see pc3d_cw_step_f32). dist_kind: 0 none, 1 L2Dist, 2 Chamfer adv2ori."""
    _, _, _, _, B, K = _pts(p, cf, "p")
    if isinstance(step, torch.Tensor):
        step_dev, step_host = step.data_ptr(), 0
    else:
        step_dev, step_host = 0, int(step)
    with torch.cuda.device(p.device):
        _lib.call("pc3d_cw_step_f32", *_pv(p, cf, "p"), *_pv(g, cf, "g"), m.data_ptr(), v.data_ptr(),
                  *_pv(ori, cf, "ori"), B, K, float(lr), float(betas[0]), float(betas[1]), float(eps), float(budget),
                  step_dev, step_host, int(dist_kind), _ptr(w), _ptr(l2norm), _ptr(nn_idx), _stream())
    return p


def estimate_normal(pts, idx, cf=False):
    """Normals [B,N,3] of pts [B,N,3] (or [B,3,N] with cf) from neighbour lists idx [B,N,k+1] int32 (self first)."""
    p, bs, ps, cs, B, N = _pts(pts, cf, "pts")
    if idx.dtype != torch.int32 or not idx.is_contiguous() or idx.shape[:2] != (B, N):
        raise ValueError("estimate_normal: idx must be a contiguous int32 [B,N,k+1] tensor")
    out = torch.empty((B, N, 3), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        _lib.call("pc3d_estimate_normal_f32", p, bs, ps, cs, idx.data_ptr(), B, N, idx.shape[2], out.data_ptr(),
                  3 * N, 3, 1, _stream())
    return out


class _KappaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, normal, idx, cf):
        p, bs, ps, cs, B, N = _pts(pts, cf, "pts")
        n, nbs, nps, ncs, _, _ = _pts(normal, cf, "normal")
        out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            _lib.call("pc3d_kappa_f32", p, bs, ps, cs, n, nbs, nps, ncs, idx.data_ptr(), B, N, idx.shape[2],
                      out.data_ptr(), _stream())
        ctx.save_for_backward(pts, normal, idx)
        ctx.cf = cf
        return out

    @staticmethod
    def backward(ctx, g):
        pts, normal, idx = ctx.saved_tensors
        p, bs, ps, cs, B, N = _pts(pts, ctx.cf, "pts")
        n, nbs, nps, ncs, _, _ = _pts(normal, ctx.cf, "normal")
        g = g.contiguous()
        gx = torch.empty((B, N, 3), dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            _lib.call("pc3d_kappa_bwd_f32", p, bs, ps, cs, n, nbs, nps, ncs, idx.data_ptr(), g.data_ptr(), B, N,
                      idx.shape[2], gx.data_ptr(), _det(), _stream())
        return (gx.transpose(1, 2) if ctx.cf else gx), None, None, None


def kappa(pts, normal, idx, cf=False):
    """GeoA3's curvature proxy: mean_j |<normalize(p_j - p_i), n_i>| over idx[b,i,1:] (the first entry is the point
    itself) -> [B,N]; differentiable in pts ([B,N,3], or [B,3,N] with cf), normal is a constant."""
    if normal.requires_grad:
        raise NotImplementedError("kappa: no gradient to the normals (they are gathered constants on the attack path)")
    if idx.dtype != torch.int32 or idx.dim() != 3 or idx.shape[2] < 2:
        raise ValueError("kappa: idx must be int32 [B,N,K+1] (self first)")
    return _KappaFn.apply(pts, normal, idx.contiguous(), cf)


class _KappaGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, normal_src, nidx, idx):
        p, bs, ps, cs, B, N = _pts(pts, True, "pts")
        n, nbs, nps, ncs, _, M = _pts(normal_src, True, "normal_src")
        out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
        nout = torch.empty((B, 3, N), dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            _lib.call("pc3d_kappa_gather_f32", p, bs, ps, cs, n, nbs, nps, ncs, M, nidx.data_ptr(), idx.data_ptr(), B, N,
                      idx.shape[2], out.data_ptr(), nout.data_ptr(), _stream())
        ctx.save_for_backward(pts, nout, idx)
        ctx.mark_non_differentiable(nout)
        ctx.set_materialize_grads(False)
        return out, nout

    @staticmethod
    def backward(ctx, g, _g_nout):
        if g is None:
            return None, None, None, None
        pts, nout, idx = ctx.saved_tensors
        p, bs, ps, cs, B, N = _pts(pts, True, "pts")
        n, nbs, nps, ncs, _, _ = _pts(nout, True, "normal")
        g = g.contiguous()
        gx = torch.empty((B, N, 3), dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            _lib.call("pc3d_kappa_bwd_f32", p, bs, ps, cs, n, nbs, nps, ncs, idx.data_ptr(), g.data_ptr(), B, N,
                      idx.shape[2], gx.data_ptr(), _det(), _stream())
        return gx.transpose(1, 2), None, None, None


def kappa_gather(pts, normal_src, nidx, idx):
    """GeoA3's curvature proxy of pts [B,3,N] with point i taking normal_src[:, :, nidx[b,i]] ([B,3,M] constants, nidx
    int64 [B,N]): (kappa [B,N], the gathered normals [B,3,N]) in one launch; differentiable in pts."""
    if normal_src.requires_grad:
        raise NotImplementedError("kappa_gather: no gradient to the normals (constants of the attack)")
    if nidx.dtype != torch.int64 or nidx.shape != (pts.shape[0], pts.shape[2]) or not nidx.is_contiguous():
        raise ValueError("kappa_gather: nidx must be a contiguous int64 [B,N]")
    if idx.dtype != torch.int32 or idx.dim() != 3 or idx.shape[2] < 2:
        raise ValueError("kappa_gather: idx must be int32 [B,N,K+1] (self first)")
    return _KappaGatherFn.apply(pts, normal_src, nidx, idx.contiguous())


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, sign, gscale):
        ctx.fixed = gscale is not None
        if sign == -1.0:                  # minus the cross-entropy is a loss kind of the kernel: no scaling launch
            _, _, loss, g = cls_loss(logits, target, 3, 0.0, 1.0 if gscale is None else gscale, want_grad=True)
            ctx.save_for_backward(g)
            return loss
        _, _, loss, g = cls_loss(logits, target, 2, 0.0, sign if gscale is None else sign * gscale, want_grad=True)
        ctx.save_for_backward(g)
        return loss * sign if sign != 1.0 else loss

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return (g if ctx.fixed else g * gout[:, None]), None, None, None


class _AdvLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, kind, kappa, gscale):
        _, _, loss, g = cls_loss(logits, target, kind, kappa, 1.0 if gscale is None else gscale, want_grad=True)
        ctx.save_for_backward(g)
        ctx.fixed = gscale is not None
        return loss

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return (g if ctx.fixed else g * gout[:, None]), None, None, None, None


def adv_loss_raw(logits, target, kind, kappa=0.0, gscale=None):
    """Per-sample adversarial loss [B] of attack/CW/CW_utils/adv_utils.py on `logits` AS GIVEN (no log-softmax): kind
    "untargeted_logits" / "logits" = clamp(+-(real - other) + kappa, 0), "cross_entropy" = -logits[target]; one launch
    each way (LOSS_KINDS). gscale (float): d loss / d term, the same for every sample, multiplied into the gradient by
    the forward launch — the backward then returns it as is and does not read the incoming gradient."""
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("adv_loss_raw: logits must be [B,k] with unit column stride")
    return _AdvLossFn.apply(logits, target, LOSS_KINDS[kind] + 4, float(kappa),
                            None if gscale is None else float(np.float32(gscale)))


def cross_entropy(logits, target, sign=1.0, gscale=None):
    """sign * CrossEntropyLoss(reduction='none')(logits, target) -> [B], one launch each way (pc3d_cls_loss_f32, kind 2).
    gscale (float): d loss / d term, the same for every sample, multiplied into the gradient by the forward launch; the
    backward then does not read the incoming gradient."""
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("cross_entropy: logits must be [B,k] with unit column stride")
    return _CrossEntropyFn.apply(logits, target, float(sign), None if gscale is None else float(np.float32(gscale)))


def geoa3_record(logits, target, targeted, metric, iterate, search_step, step, best_loss, best_attack, best_bs, best_step,
                 iter_best_loss, iter_best_score):
    """One GeoA3 iteration's best-attack bookkeeping in place (pc3d_geoa3_record_f32); returns the predicted labels [B]."""
    _check(logits, "logits"), _check(metric, "metric"), _check(iterate, "iterate")
    B, ncls = logits.shape
    for nm, t, dt in (("target", target, torch.int64), ("best_loss", best_loss, torch.float32),
                      ("best_bs", best_bs, torch.int64), ("best_step", best_step, torch.int64),
                      ("iter_best_loss", iter_best_loss, torch.float32), ("iter_best_score", iter_best_score, torch.int64),
                      ("metric", metric, torch.float32)):
        if t.dtype != dt or t.shape != (B,) or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f"geoa3_record: {nm} must be a contiguous {dt} GPU tensor of shape [{B}]")
    if logits.stride(1) != 1 or not iterate.is_contiguous() or best_attack.shape != iterate.shape \
            or not best_attack.is_contiguous() or best_attack.dtype != torch.float32:
        raise ValueError("geoa3_record: logits rows / iterate / best_attack must be contiguous float32 of matching shapes")
    label = torch.empty((B,), dtype=torch.int64, device=logits.device)
    with torch.cuda.device(logits.device):
        _lib.call("pc3d_geoa3_record_f32", logits.data_ptr(), logits.stride(0), B, ncls, target.data_ptr(),
                  1 if targeted else 0, metric.data_ptr(), iterate.data_ptr(), iterate[0].numel(), int(search_step),
                  int(step), best_loss.data_ptr(), best_attack.data_ptr(), best_bs.data_ptr(), best_step.data_ptr(),
                  iter_best_loss.data_ptr(), iter_best_score.data_ptr(), label.data_ptr(), _stream())
    return label


class _GeoTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d_ao, d_oa, k_adv, k_ori, idx_ao, cls, scale, w, gfix=None):
        B, N = d_ao.shape
        M = d_oa.shape[1] if d_oa is not None else (k_ori.shape[1] if k_ori is not None else N)
        dev = d_ao.device
        out = torch.empty((5, B), dtype=torch.float32, device=dev)
        hd_arg = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("pc3d_geoa3_terms_f32", d_ao.data_ptr(), _ptr(d_oa), _ptr(k_adv), _ptr(k_ori), _ptr(idx_ao),
                      cls.data_ptr(), scale.data_ptr(), B, N, M, w[0], w[1], w[2], out.data_ptr(), hd_arg.data_ptr(),
                      _stream())
        ctx.save_for_backward(k_adv, k_ori, idx_ao, scale, hd_arg, gfix)
        ctx.dims, ctx.w, ctx.has_oa = (B, N, M), w, d_oa is not None
        return out

    @staticmethod
    def backward(ctx, g):
        k_adv, k_ori, idx_ao, scale, hd_arg, gfix = ctx.saved_tensors
        B, N, M = ctx.dims
        dev = g.device
        g = gfix if gfix is not None else g.contiguous()
        g_ao = torch.empty((B, N), dtype=torch.float32, device=dev)
        g_oa = torch.empty((B, M), dtype=torch.float32, device=dev) if ctx.has_oa else None
        g_k = torch.empty((B, N), dtype=torch.float32, device=dev) if k_adv is not None else None
        g_cls = torch.empty((B,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("pc3d_geoa3_terms_bwd_f32", g.data_ptr(), _ptr(k_adv), _ptr(k_ori), _ptr(idx_ao), scale.data_ptr(),
                      hd_arg.data_ptr(), B, N, M, ctx.w[0], ctx.w[1], ctx.w[2], g_ao.data_ptr(), _ptr(g_oa), _ptr(g_k),
                      g_cls.data_ptr(), _stream())
        return g_ao, g_oa, g_k, None, None, g_cls, None, None, None


_GEO_GFIX = {}


def geoa3_loss_grad(device, B, value):
    """The [5,B] upstream gradient of geoa3_terms' output when only its last row enters the loss, as value * sum_b loss_n[b]
    (value = 1/B for the batch mean): cached, never written again."""
    key = (torch.device(device).index, int(B), float(np.float32(value)))
    t = _GEO_GFIX.get(key)
    if t is None and torch.cuda.is_current_stream_capturing():      # as const_vec: owned by the capture, not cached
        t = torch.zeros((5, int(B)), dtype=torch.float32, device=device)
        t[4].fill_(float(np.float32(value)))
        _graphed.note_captured(t)
        return t
    if t is None:
        if len(_GEO_GFIX) >= 64:
            _GEO_GFIX.pop(next(iter(_GEO_GFIX)))
        t = torch.zeros((5, int(B)), dtype=torch.float32, device=device)
        t[4].fill_(float(np.float32(value)))
        _GEO_GFIX[key] = t
    _graphed.note_captured(t)
    return t


def geoa3_terms(d_ao, d_oa, k_adv, k_ori, idx_ao, cls, scale, w_dis, w_hd, w_curv, gfix=None):
    """GeoA3's loss assembly in one launch (pc3d_geoa3_terms_f32): returns [5,B] = (dis, hd, curv, constrain, loss_n)
    rows; differentiable in d_ao, d_oa, k_adv and cls. d_oa None = pseudo-Chamfer, k_adv None = no curvature term.
    gfix [5,B] (geoa3_loss_grad): the backward uses it as the output's gradient and does not read the incoming one."""
    for nm, t in (("d_ao", d_ao), ("d_oa", d_oa), ("k_adv", k_adv), ("k_ori", k_ori), ("cls", cls), ("scale", scale)):
        if t is not None:
            _check(t, nm)
            if not t.is_contiguous():
                raise ValueError(f"geoa3_terms: {nm} must be contiguous")
    B, N = d_ao.shape
    if cls.shape != (B,) or scale.shape != (B,) or (d_oa is not None and d_oa.shape[0] != B):
        raise ValueError("geoa3_terms: cls / scale / d_oa do not match d_ao's batch")
    if k_adv is not None:
        if k_ori is None or idx_ao is None or idx_ao.dtype != torch.int64 or not idx_ao.is_contiguous():
            raise ValueError("geoa3_terms: the curvature term needs k_ori and a contiguous int64 idx_ao")
        if k_adv.shape != (B, N) or idx_ao.shape != (B, N) or (d_oa is not None and k_ori.shape != d_oa.shape):
            raise ValueError("geoa3_terms: k_adv / idx_ao / k_ori shapes do not match the distances")
        if k_ori.requires_grad:
            raise NotImplementedError("geoa3_terms: k_ori is a constant of the attack")
    else:
        k_ori = idx_ao = None
    return _GeoTermsFn.apply(d_ao, d_oa, k_adv, k_ori, idx_ao, cls, scale, (float(w_dis), float(w_hd), float(w_curv)), gfix)


# ------------------------------------------------------------------------------------------------------
# K8b: point-wise dense layers (frozen weights) on the fp32-MFMA GEMM
# ------------------------------------------------------------------------------------------------------
_ACTS = {None: 0, "none": 0, "relu": 1, "leaky": 2}
GEMM_SMALL_M = 64
GEMM_HEAD_MAX_M = 4096      # sample rows up to which a head runs on the small-batch kernel (32-row tiles)


def F_leaky(y, slope):
    return torch.nn.functional.leaky_relu_(y, slope)


_WT_CACHE = {}      # (data_ptr, version, shape) -> (w, W^T): the backward's operand; weights are frozen
WT_CACHE_MAX = 512


def _w_transposed(w):
    """W^T (contiguous) of a frozen weight, computed once per (storage address, version, shape). The entry keeps a
    reference to `w` — an alias is enough — so the storage cannot be freed and its address reused while the entry lives;
    in-place updates bump the version, re-folded / moved weights have another address. (Until late round 2 the entry
    held a WEAK reference to the tensor OBJECT; callers pass `w.detach()`, a new object per call, so every backward of
    every point-wise layer re-transposed its weight: 7 copies per DGCNN backward, ~60 per CurveNet backward.)"""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _WT_CACHE.get(key)
    if hit is None and torch.cuda.is_current_stream_capturing():    # as const_vec: the transposing copy is only recorded
        wt = w.t().contiguous()
        _graphed.note_captured(w, wt)
        return wt
    if hit is None:
        while len(_WT_CACHE) >= WT_CACHE_MAX:             # oldest entry out, one at a time — never wholesale: a captured
            _WT_CACHE.pop(next(iter(_WT_CACHE)))          # graph that used it holds its own reference (note_captured)
        hit = (w, w.t().contiguous())
        _WT_CACHE[key] = hit
    _graphed.note_captured(hit[0], hit[1])
    return hit[1]


GEMM_NOMINAL_BATCH = 32      # clouds per launch the K-split rule below is tuned for (one GPU's share of every config)


def _unit_rows(shape):
    """Rows per cloud of a [B, ..., K] operand (None for a plain [M, K] matrix: no K split)."""
    if len(shape) < 3:
        return None
    n = 1
    for d in shape[1:-1]:
        n *= int(d)
    return n


def gemm_variant(unit_rows, N, K):
    """Tile variant of pc3d_gemm_nt_tiled_f32 for a layer with `unit_rows` rows per cloud: -1 (the library's default
    tiling) or one of the K-split variants 11 / 12 for launches with few tiles and a long K (CurveNet's deep levels).
    The K split changes the order in which an output element's products are summed, so the choice is made from the
    PER-CLOUD shape and a nominal batch of GEMM_NOMINAL_BATCH clouds — never from the actual row count: a cloud's
    arithmetic is the same alone, in a batch of 32 or in a shard of any size."""
    if unit_rows is None or K < 128:
        return -1
    M = GEMM_NOMINAL_BATCH * unit_rows
    cd = lambda a, b: (a + b - 1) // b
    if cd(M, 128) * cd(N, 64 if N <= 64 else 128) >= 192:
        return -1
    return 12 if (cd(M, 64) * cd(N, 64) < 192 and K >= 256) else 11


def gemm_nt(x2d, w, bias=None, act=None, slope=0.0, gate=None, gate_slope=0.0, out=None, unit_rows=None, head=False):
    """Y[M,N] = act(gate(x2d)[M,K] @ w[N,K]^T + bias) on pc3d_gemm_nt_f32 (exact fp32 MFMA). x2d / gate / out may be
    row-strided views (last dimension contiguous). unit_rows: rows per cloud (see gemm_variant). head: the rows are
    SAMPLES (a classifier head, [B, K]) — those go to the small-batch kernel whatever B is, so that a sample's logits
    do not depend on the batch size it is evaluated in."""
    _check(x2d, "x")
    _check(w, "w")
    M, K = x2d.shape
    N = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous():
        raise ValueError("gemm_nt: w must be a contiguous [N,K] matrix matching x's K")
    if x2d.stride(1) != 1 or (gate is not None and (gate.stride(1) != 1 or gate.shape != x2d.shape)):
        raise ValueError("gemm_nt: x / gate need a contiguous last dimension and equal shapes")
    if head and K % 8 == 0 and x2d.stride(0) % 4 == 0 and out is None and M <= GEMM_HEAD_MAX_M:
        # a handful of rows (the classifier heads: M = batch): the 128-row tile would leave all but N/128 CUs idle and
        # walk K serially; the small-batch kernel (32 x 16 tiles, K split over the 8 waves of a workgroup) is 10-30x
        # faster there. It has ReLU in its epilogue; the LeakyReLU / the input mask of the backward are elementwise
        # passes over [M, N] / [M, K] with M <= 64 rows.
        xs = x2d if gate is None else torch.where(gate > 0, x2d, gate_slope * x2d)
        return linear(xs, w, bias, relu=act in ("relu", "leaky"), slope=slope if act == "leaky" else 0.0)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
    variant = gemm_variant(unit_rows, N, K)
    with torch.cuda.device(x2d.device):
        if variant < 0:
            _lib.call("pc3d_gemm_nt_f32", x2d.data_ptr(), x2d.stride(0), w.data_ptr(), _ptr(bias), _ptr(gate),
                      gate.stride(0) if gate is not None else 0, float(gate_slope), M, N, K, _ACTS[act], float(slope),
                      out.data_ptr(), out.stride(0), _stream())
        else:
            _lib.call("pc3d_gemm_nt_tiled_f32", x2d.data_ptr(), x2d.stride(0), w.data_ptr(), _ptr(bias), _ptr(gate),
                      gate.stride(0) if gate is not None else 0, float(gate_slope), M, N, K, _ACTS[act], float(slope),
                      out.data_ptr(), out.stride(0), variant, _stream())
    return out


class _LinearActFn(torch.autograd.Function):
    """act(x @ w.T + b) with the weights frozen: backward = ONE launch of the same kernel on W^T with the activation's
    derivative applied to dY on load (gate = the saved output)."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        y = gemm_nt(x2, w, b, act, slope, unit_rows=_unit_rows(shp), head=len(shp) == 2)
        ctx.act, ctx.slope, ctx.shp = act, slope, shp
        ctx.save_for_backward(y if act in ("relu", "leaky") else None, w)
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        y, w = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1])
        if g2.stride(1) != 1:
            g2 = g2.contiguous()
        gx = gemm_nt(g2, _w_transposed(w), None, None, 0.0, gate=y,
                     gate_slope=ctx.slope if ctx.act == "leaky" else 0.0, unit_rows=_unit_rows(ctx.shp),
                     head=len(ctx.shp) == 2)
        return gx.view(ctx.shp), None, None, None, None


class _HeadMLPFn(torch.autograd.Function):
    """A classifier head — Linear + (Leaky)ReLU layers over the SAMPLES of a batch, [B,K0] -> [B,Kn] — as one launch per
    layer each way: the activation sits in the forward launch's epilogue and the backward chains
    dX_{l-1} = act'_{l-1}(dX_l . W_l) with the previous layer's saved output as the launch's gate (pc3d_linear_f32), instead
    of an activation pass forward and a compare / scale / select triple backward per layer."""

    @staticmethod
    def forward(ctx, x, acts, *wb):
        n = len(wb) // 2
        ys, h = [], x
        for l in range(n):
            name, slope = acts[l]
            h = linear(h, wb[2 * l], wb[2 * l + 1], relu=name is not None, slope=slope)
            ys.append(h)
        ctx.acts, ctx.n = acts, n
        ctx.save_for_backward(*[ys[l] if acts[l][0] is not None else None for l in range(n - 1)], *[wb[2 * l] for l in range(n)])
        ctx.last = ys[-1] if acts[-1][0] is not None else None
        return ys[-1]

    @staticmethod
    def backward(ctx, g):
        n, acts = ctx.n, ctx.acts
        saved = ctx.saved_tensors
        ys, ws = saved[:n - 1], saved[n - 1:]
        g = g.contiguous()
        if ctx.last is not None:                    # an activation on the LAST layer: mask the incoming gradient once
            g = gate(g, ctx.last, acts[-1][1])
        for l in range(n - 1, -1, -1):
            prev = ys[l - 1] if l > 0 else None
            g = linear(g, _w_transposed(ws[l]), None, gate=prev, gate_slope=acts[l - 1][1] if prev is not None else 0.0)
        return (g, None) + (None,) * (2 * n)


def head_mlp(x, layers):
    """x [B,K0] through layers = [(W [O,K], b or None, act in {None,"relu","leaky"}, slope), ...] of a frozen classifier
    head; differentiable in x. One pc3d_linear_f32 launch per layer forward and backward."""
    _check(x, "x")
    if x.dim() != 2:
        raise ValueError("head_mlp: x must be [B,K] (rows are samples)")
    acts, wb = [], []
    for (w, b, act, slope) in layers:
        w = w.detach()
        wb += [w if w.is_contiguous() else w.contiguous(), b.detach() if b is not None else None]
        acts.append((act, float(slope) if act == "leaky" else 0.0))
        if w.shape[1] % 8 or act not in (None, "relu", "leaky"):
            raise ValueError("head_mlp: K % 8 == 0 and act in {None, relu, leaky} expected")
    x = x if x.stride(1) == 1 and x.stride(0) % 4 == 0 else x.contiguous()
    return _HeadMLPFn.apply(x, tuple(acts), *wb)


def linear_act(x, w, b=None, act=None, slope=0.0):
    """[..., K] -> [..., N]: act(x @ w.T + b) for frozen (w, b); differentiable in x. act in {None, "relu", "leaky"}."""
    w = w.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    return _LinearActFn.apply(x, w, b.detach() if b is not None else None, act, float(slope))


# ------------------------------------------------------------------------------------------------------
# K5/K6/K7: PointNet++ sampling & grouping
# ------------------------------------------------------------------------------------------------------
FPS_THREADS_OVERRIDE = int(os.environ.get("PC3D_FPS_THREADS", "0"))     # 0: the library's choice


def fps(xyz, npoint, start=None, cf=False):
    """Farthest-point sampling: xyz [B,N,3] (or [B,3,N] with cf) -> int32 [B,npoint]. start: int32 [B] or None (0)."""
    p, bs, ps, cs, B, N = _pts(xyz, cf, "xyz")
    out = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    if start is not None and (start.dtype != torch.int32 or not start.is_cuda):
        raise TypeError("fps: start must be an int32 GPU tensor")
    with torch.cuda.device(xyz.device):
        if FPS_THREADS_OVERRIDE and N <= 32 * FPS_THREADS_OVERRIDE:     # diagnostics only (tools/exp/cw_curvenet_race.py)
            _lib.call("pc3d_fps_threads_f32", FPS_THREADS_OVERRIDE, p, bs, ps, cs, B, N, int(npoint), _ptr(start), out.data_ptr(),
                      _stream())
        else:
            _lib.call("pc3d_fps_f32", p, bs, ps, cs, B, N, int(npoint), _ptr(start), out.data_ptr(), _stream())
    return out


def ball_query(radius, nsample, xyz, new_xyz, cf=False, kernel=None):
    """int32 [B,S,nsample]: first nsample in-radius indices in ascending order, padded with the first. kernel: None (the
    library chooses by problem size), "wave" (a wavefront per centre) or "lane" (a centre per lane) — same results."""
    p, bs, ps, cs, B, N = _pts(xyz, cf, "xyz")
    c, cbs, cps, ccs, B2, S = _pts(new_xyz, cf, "new_xyz")
    out = torch.empty((B, S, nsample), dtype=torch.int32, device=xyz.device)
    args = (p, bs, ps, cs, c, cbs, cps, ccs, B, N, S, float(radius), int(nsample), out.data_ptr(), _stream())
    with torch.cuda.device(xyz.device):
        if kernel is None:
            _lib.call("pc3d_ball_query_f32", *args)
        else:
            _lib.call("pc3d_ball_query_kernel_f32", {"wave": 1, "lane": 2}[kernel], *args)
    return out


class _GroupGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, feat, idx, centers, center_idx):
        # xyz [B,N,3] or None; feat [B,N,D] or None; idx int32 [B,S,ns]; centers [B,S,3] or None
        B, S, ns = idx.shape
        ref = xyz if xyz is not None else feat
        N = ref.shape[1]
        D = 0 if feat is None else feat.shape[2]
        if feat is not None:
            _check(feat, "feat")
            feat = feat.contiguous()
        out = torch.empty((B, S, ns, (3 if xyz is not None else 0) + D), dtype=torch.float32, device=ref.device)
        with torch.cuda.device(ref.device):
            _lib.call("pc3d_group_gather_f32", *_pv(xyz, False, "xyz"), _ptr(feat), D, idx.data_ptr(),
                      *_pv(centers, False, "centers"), B, N, S, ns, out.data_ptr(), _stream())
        ctx.save_for_backward(idx, center_idx)
        ctx.cfg = (B, N, S, ns, D, xyz is not None, feat is not None, centers is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, center_idx = ctx.saved_tensors
        B, N, S, ns, D, has_x, has_f, has_c = ctx.cfg
        g = g.contiguous()
        need_x = has_x and ctx.needs_input_grad[0]
        need_f = has_f and ctx.needs_input_grad[1]
        need_c = has_c and ctx.needs_input_grad[3]
        gx = torch.empty((B, N, 3), dtype=torch.float32, device=g.device) if (need_x or need_c) else None
        gf = torch.empty((B, N, D), dtype=torch.float32, device=g.device) if need_f else None
        gc = None
        if need_c and center_idx is None:
            # centres are an independent tensor: their gradient is minus the group sums
            gc = -g[..., :3].sum(dim=2)
        det_ws = torch.empty((B, S, 3), dtype=torch.float32, device=g.device) if _det() else None
        with torch.cuda.device(g.device):
            _lib.call("pc3d_group_gather_bwd_f32", g.data_ptr(), idx.data_ptr(),
                      _ptr(center_idx) if (has_c and center_idx is not None) else 0, B, N, S, ns, D,
                      1 if has_x else 0, _ptr(gx) if need_x or (need_c and center_idx is not None) else 0, _ptr(gf),
                      _ptr(det_ws), _stream())
        return (gx if need_x else None), gf, None, gc, None


def group_gather(xyz, feat, idx, centers=None, center_idx=None):
    """[B,S,ns,(3)+D] = [xyz[idx]-centers, feat[idx]]. When the centres are xyz[center_idx] pass center_idx (int32
    [B,S]) so their gradient is folded into grad_xyz inside the kernel; otherwise centers gets its own gradient."""
    return _GroupGatherFn.apply(xyz, feat, idx, centers, center_idx)


# ------------------------------------------------------------------------------------------------------
# K3: DGCNN dynamic graph
# ------------------------------------------------------------------------------------------------------
class _GroupActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, Bc, idx, slope):
        B, NA, C = P.shape
        S, K = idx.shape[1], idx.shape[2]
        H = torch.empty((B, S, K, C), dtype=torch.float32, device=P.device)
        with torch.cuda.device(P.device):
            _lib.call("pc3d_group_act_f32", P.data_ptr(), Bc.data_ptr(), idx.data_ptr(), B, NA, S, K, C, float(slope),
                      H.data_ptr(), _stream())
        ctx.save_for_backward(H, idx)
        ctx.meta = (NA, float(slope))
        return H

    @staticmethod
    def backward(ctx, gH):
        H, idx = ctx.saved_tensors
        NA, slope = ctx.meta
        B, S, K, C = H.shape
        gH = gH.contiguous()
        gP = torch.empty((B, NA, C), dtype=torch.float32, device=H.device)
        gBc = torch.empty((B, S, C), dtype=torch.float32, device=H.device)
        with torch.cuda.device(H.device):
            _lib.call("pc3d_group_act_bwd_f32", gH.data_ptr(), H.data_ptr(), idx.data_ptr(), B, NA, S, K, C, slope,
                      gP.data_ptr(), gBc.data_ptr(), _det(), _stream())
        return gP, gBc, None, None


GROUP_ACT_MAX_C = 512


def group_act(P, Bc, idx, slope=0.0):
    """H[b,s,j,:] = act(P[b,idx[b,s,j],:] + Bc[b,s,:]) — P [B,NA,C] per point, Bc [B,S,C] per group, idx [B,S,K] int32
    -> [B,S,K,C]; act = LeakyReLU(slope) (0 = ReLU). Differentiable in P and Bc. C % 4 == 0, C <= 512."""
    _check(P, "P"), _check(Bc, "Bc")
    if (P.dim() != 3 or Bc.dim() != 3 or idx.dim() != 3 or P.shape[2] % 4 or P.shape[2] > GROUP_ACT_MAX_C
            or Bc.shape != (P.shape[0], idx.shape[1], P.shape[2]) or idx.dtype != torch.int32):
        raise ValueError("group_act: P [B,NA,C], Bc [B,S,C] (C % 4 == 0, C <= 512), idx [B,S,K] int32 expected")
    return _GroupActFn.apply(P.contiguous(), Bc.contiguous(), idx.contiguous(), slope)


def knn_feat(x, K):
    """x [B,N,C] channels-last features -> int32 [B,N,K] nearest (self first) in feature space."""
    _check(x, "x")
    x = x.contiguous()
    B, N, C = x.shape
    idx = torch.empty((B, N, K), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("pc3d_knn_feat_f32", x.data_ptr(), B, N, C, int(K), idx.data_ptr(), _stream())
    return idx


class _GatherMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, idx):
        P = P.contiguous()
        B, N, C = P.shape
        K = idx.shape[2]
        out = torch.empty_like(P)
        arg = torch.empty((B, N, C), dtype=torch.int32, device=P.device)
        with torch.cuda.device(P.device):
            _lib.call("pc3d_gather_max_f32", P.data_ptr(), idx.data_ptr(), 0, B, N, C, K, out.data_ptr(),
                      arg.data_ptr(), _stream())
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        g = g.contiguous()
        B, N, C = g.shape
        gP = torch.empty_like(g)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_gather_max_bwd_f32", g.data_ptr(), arg.data_ptr(), B, N, C, gP.data_ptr(), _det(), _stream())
        return gP, None


def gather_max(P, idx):
    """out[b,i,c] = max_j P[b, idx[b,i,j], c]; differentiable in P (gradient to the arg-max neighbour)."""
    _check(P, "P")
    return _GatherMaxFn.apply(P, idx.contiguous())


class _ActPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, slope):
        B, N, C = Y.shape
        out = torch.empty((B, 2 * C), dtype=torch.float32, device=Y.device)
        arg = torch.empty((B, C), dtype=torch.int32, device=Y.device)
        with torch.cuda.device(Y.device):
            _lib.call("pc3d_act_pool_f32", Y.data_ptr(), B, N, C, float(slope), out.data_ptr(), arg.data_ptr(), _stream())
        ctx.save_for_backward(Y, arg)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        Y, arg = ctx.saved_tensors
        B, N, C = Y.shape
        g = g.contiguous()
        gY = torch.empty_like(Y)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_act_pool_bwd_f32", Y.data_ptr(), g.data_ptr(), arg.data_ptr(), B, N, C, ctx.slope,
                      gY.data_ptr(), _stream())
        return gY, None


class _LinearActPoolFn(torch.autograd.Function):
    """act_maxmean_pool(x @ w.T + b) for a frozen layer: forward = the point-wise GEMM + the pooling launch; backward = ONE
    GEMM on W^T whose dY operand is generated on load from the saved pre-activation, the upstream [B,2C] gradient and the
    arg-max rows (pc3d_gemm_nt_poolbwd_f32) — no pooling-backward launch and no [B,N,C] gradient tensor."""

    @staticmethod
    def forward(ctx, x, w, b, slope):
        B, N, K = x.shape
        C = w.shape[0]
        Y = gemm_nt(x.reshape(B * N, K), w, b, unit_rows=N)
        out = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_act_pool_f32", Y.data_ptr(), B, N, C, float(slope), out.data_ptr(), arg.data_ptr(), _stream())
        ctx.save_for_backward(Y, arg, w)
        ctx.meta = (B, N, K, C, float(slope))
        return out

    @staticmethod
    def backward(ctx, g):
        Y, arg, w = ctx.saved_tensors
        B, N, K, C, slope = ctx.meta
        g = g.contiguous()
        wt = _w_transposed(w)                                   # [K, C]
        gx = torch.empty((B * N, K), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_gemm_nt_poolbwd_f32", Y.data_ptr(), Y.stride(0), g.data_ptr(), arg.data_ptr(), B, N, slope,
                      wt.data_ptr(), K, C, gx.data_ptr(), K, _stream())
        return gx.view(B, N, K), None, None, None


def linear_act_maxmean_pool(x, w, b, slope):
    """[max_n z | mean_n z] of z = leaky_relu(x @ w.T + b, slope) for x [B,N,K] and a frozen (w [C,K], b): [B,2C].
    Differentiable in x (see _LinearActPoolFn). C % 4 == 0; a handful of rows, or N not a multiple of 128, goes through
    the two-operator form."""
    _check(x, "x"), _check(w, "w")
    if x.dim() != 3 or w.shape[0] % 4 or x.shape[1] % 128:
        return act_maxmean_pool(linear_act(x, w, b), slope)      # (the fused backward wants whole 128-row tiles per cloud)
    w = w.detach()
    return _LinearActPoolFn.apply(x.contiguous(), w if w.is_contiguous() else w.contiguous(),
                                  b.detach() if b is not None else None, slope)


def act_maxmean_pool(Y, slope):
    """[max_i z | mean_i z] over dim 1 of z = leaky_relu(Y, slope) (slope 0: ReLU) for Y [B,N,C], C % 4 == 0 -> [B,2C]."""
    _check(Y, "Y")
    if Y.dim() != 3 or Y.shape[2] % 4:
        raise ValueError("act_maxmean_pool: Y must be [B,N,C] with C % 4 == 0")
    return _ActPoolFn.apply(Y.contiguous(), slope)


class _EdgeMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, PQ, idx, slope):
        B, N, C2 = PQ.shape
        C = C2 // 2
        out = torch.empty((B, N, C), dtype=torch.float32, device=PQ.device)
        arg = torch.empty((B, N, C), dtype=torch.int32, device=PQ.device)
        with torch.cuda.device(PQ.device):
            _lib.call("pc3d_edge_max_f32", PQ.data_ptr(), idx.data_ptr(), B, N, C, idx.shape[2], float(slope),
                      out.data_ptr(), arg.data_ptr(), _stream())
        ctx.save_for_backward(out, arg)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        out, arg = ctx.saved_tensors
        B, N, C = out.shape
        # a column slice of a wider gradient (torch.cat's backward) is read in place through its row stride
        if not (g.stride(2) == 1 and g.stride(0) == N * g.stride(1) and g.stride(1) >= C):
            g = g.contiguous()
        gPQ = torch.empty((B, N, 2 * C), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_edge_max_bwd_f32", g.data_ptr(), g.stride(1), out.data_ptr(), arg.data_ptr(), B, N, C,
                      ctx.slope, gPQ.data_ptr(), _det(), _stream())
        return gPQ, None, None


def edge_max_raw(PQ, idx, slope, cat=None, off=0):
    """(out [B,N,C], arg int32 [B,N,C]) of pc3d_edge_max_f32, no autograd; with `cat` [B,N,Ct] the launch also writes out into
    cat[:, :, off:off+C] (pc3d_edge_max_cat_f32)."""
    B, N, C2 = PQ.shape
    C = C2 // 2
    out = torch.empty((B, N, C), dtype=torch.float32, device=PQ.device)
    arg = torch.empty((B, N, C), dtype=torch.int32, device=PQ.device)
    with torch.cuda.device(PQ.device):
        if cat is None:
            _lib.call("pc3d_edge_max_f32", PQ.data_ptr(), idx.data_ptr(), B, N, C, idx.shape[2], float(slope),
                      out.data_ptr(), arg.data_ptr(), _stream())
        else:
            if not cat.is_contiguous() or cat.shape[:2] != (B, N) or off % 4 or off + C > cat.shape[2]:
                raise ValueError("edge_max_raw: cat must be a contiguous [B,N,Ct] buffer and off a multiple of 4 inside it")
            _lib.call("pc3d_edge_max_cat_f32", PQ.data_ptr(), idx.data_ptr(), B, N, C, idx.shape[2], float(slope),
                      out.data_ptr(), arg.data_ptr(), cat.data_ptr() + 4 * off, cat.shape[2], _stream())
    return out, arg


def edge_max_bwd_raw(gcat, off, ld, out, arg, B, N, C, slope, g2=None):
    """gPQ [B,N,2C] of pc3d_edge_max_bwd_f32 for the upstream gradient gcat[:, off:off+C] read in place (row stride ld);
    g2 [B*N, C] contiguous or None: a second upstream gradient, added on load (deterministic mode) or by a launch of its own."""
    gPQ = torch.empty((B, N, 2 * C), dtype=torch.float32, device=gcat.device)
    with torch.cuda.device(gcat.device):
        if g2 is not None and _det():
            _lib.call("pc3d_edge_max_bwd_sum_f32", gcat.data_ptr() + 4 * off, ld, g2.data_ptr(), C, out.data_ptr(), arg.data_ptr(),
                      B, N, C, float(slope), gPQ.data_ptr(), _stream())
            return gPQ
        if g2 is not None:
            gsum = gcat.view(B, N, ld)[:, :, off:off + C] + g2.view(B, N, C)
            _lib.call("pc3d_edge_max_bwd_f32", gsum.data_ptr(), C, out.data_ptr(), arg.data_ptr(), B, N, C,
                      float(slope), gPQ.data_ptr(), 0, _stream())
            return gPQ
        _lib.call("pc3d_edge_max_bwd_f32", gcat.data_ptr() + 4 * off, ld, out.data_ptr(), arg.data_ptr(), B, N, C,
                  float(slope), gPQ.data_ptr(), _det(), _stream())
    return gPQ


def act_pool_raw(Y, B, N, C, slope):
    """([max_n z | mean_n z] [B,2C], arg-max rows int32 [B,C]) of z = leaky(Y) for Y [B*N, C] (pc3d_act_pool_f32), no autograd."""
    out = torch.empty((B, 2 * C), dtype=torch.float32, device=Y.device)
    arg = torch.empty((B, C), dtype=torch.int32, device=Y.device)
    with torch.cuda.device(Y.device):
        _lib.call("pc3d_act_pool_f32", Y.data_ptr(), B, N, C, float(slope), out.data_ptr(), arg.data_ptr(), _stream())
    return out, arg


def pool_bwd_gemm_raw(Y, g, arg, B, N, slope, w, K, C):
    """d/dx [B*N, K] of act_maxmean_pool(x @ w.T + b) from the saved pre-activation Y [B*N, C], the upstream g [B,2C] and the
    arg-max rows: ONE GEMM on W^T whose dY operand is generated on load (pc3d_gemm_nt_poolbwd_f32), no autograd."""
    wt = _w_transposed(w)
    gx = torch.empty((B * N, K), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.call("pc3d_gemm_nt_poolbwd_f32", Y.data_ptr(), Y.stride(0), g.data_ptr(), arg.data_ptr(), B, N, float(slope),
                  wt.data_ptr(), K, C, gx.data_ptr(), K, _stream())
    return gx


def gemm_nt_res_into(x2d, w, buf, off, width, unit_rows=None):
    """buf[:, off:off+width] += x2d [M,K] @ w [width,K]^T in the GEMM's epilogue (pc3d_gemm_nt_res_f32 with R = Y = the slice of
    the contiguous [M, ld] buffer): a gradient accumulated where it already lives, without a product tensor and an add."""
    M, K = x2d.shape
    if w.shape != (width, K) or not w.is_contiguous() or not buf.is_contiguous() or buf.shape[0] != M or off % 4:
        raise ValueError("gemm_nt_res_into: w [width,K] contiguous, buf [M,ld] contiguous, off % 4 == 0 expected")
    ld = buf.shape[1]
    ptr = buf.data_ptr() + 4 * off
    with torch.cuda.device(x2d.device):
        _lib.call("pc3d_gemm_nt_res_f32", x2d.data_ptr(), x2d.stride(0), w.data_ptr(), 0, ptr, ld, M, width, K, _ACTS[None], 0.0,
                  ptr, ld, gemm_variant(unit_rows, width, K), _stream())


def edge_max(PQ, idx, slope=0.2):
    """out[b,i,c] = leaky(max_j P[b,idx[b,i,j],c] + Q[b,i,c]) for PQ [B,N,2C] = [P | Q] (C % 4 == 0): the neighbour
    reduction, the centre term and the activation of an EdgeConv layer in one launch; differentiable in PQ."""
    _check(PQ, "PQ")
    if PQ.shape[2] % 8:
        raise ValueError("edge_max: PQ must be [B,N,2C] with C % 4 == 0")
    return _EdgeMaxFn.apply(PQ.contiguous(), idx.contiguous(), slope)


# ------------------------------------------------------------------------------------------------------
# K17: CurveNet local point-feature aggregation (edge activation, activation + neighbour mean)
# ------------------------------------------------------------------------------------------------------
def rev_index(idx, NA, clamp=True):
    """Sorted reverse index of a gather through idx [B, ...] int32 (flattened per cloud to E entries, values in [0, NA)):
    (off [B,NA+1], lst [B,E]) with lst[b, off[b,t]:off[b,t+1]] = the entries that read row t, ascending. The
    deterministic backward of the gather sums through it (pc3d_rev_gather_sum_f32)."""
    if idx.dtype != torch.int32 or not idx.is_cuda:
        raise ValueError("rev_index: idx must be an int32 GPU tensor")
    idx = idx.contiguous()
    B = idx.shape[0]
    E = idx[0].numel()
    dev = idx.device
    cnt = torch.empty((B, NA), dtype=torch.int32, device=dev)
    off = torch.empty((B, NA + 1), dtype=torch.int32, device=dev)
    lst = torch.empty((B, E), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pc3d_rev_index_i32", idx.data_ptr(), B, E, int(NA), 1 if clamp else 0, cnt.data_ptr(), off.data_ptr(),
                  lst.data_ptr(), _stream())
    return off, lst


def attach_rev_index(idx, NA):
    """Build the sorted reverse index of a neighbour graph once, where the graph is built (a victim's geometry stream),
    and hang it on the tensor: every operator that later gathers through this idx object (lpfa_fused, edge_act) finds
    it there instead of building its own. No-op outside deterministic mode."""
    if DETERMINISTIC and getattr(idx, "_pc3d_rev", None) is None:
        idx._pc3d_rev = rev_index(idx, NA)
    return idx


def _rev_of(idx, NA, needs_grad):
    """The reverse index an operator's deterministic backward will gather through, or None."""
    if not (DETERMINISTIC and needs_grad and torch.is_grad_enabled()):
        return None
    rev = getattr(idx, "_pc3d_rev", None)
    return rev if rev is not None else rev_index(idx, NA)


class _EdgeActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, Bc, idx, slope, rev_off, rev_lst):
        B, N, C = A.shape
        K = idx.shape[2]
        E = torch.empty((B, N, K, C), dtype=torch.float32, device=A.device)
        with torch.cuda.device(A.device):
            _lib.call("pc3d_edge_act_f32", A.data_ptr(), Bc.data_ptr(), idx.data_ptr(), B, N, K, C, float(slope),
                      E.data_ptr(), _stream())
        ctx.save_for_backward(E, idx, rev_off, rev_lst)
        ctx.slope = float(slope)
        return E

    @staticmethod
    def backward(ctx, gE):
        E, idx, rev_off, rev_lst = ctx.saved_tensors
        B, N, K, C = E.shape
        gE = gE.contiguous()
        det = 1 if (_det() or rev_off is not None) else 0
        gA = (torch.empty if det else torch.zeros)((B, N, C), dtype=torch.float32, device=E.device)
        gBc = torch.empty((B, N, C), dtype=torch.float32, device=E.device)
        with torch.cuda.device(E.device):
            _lib.call("pc3d_edge_act_bwd_f32", gE.data_ptr(), E.data_ptr(), idx.data_ptr(), B, N, K, C, ctx.slope,
                      gA.data_ptr(), gBc.data_ptr(), det, _ptr(rev_off), _ptr(rev_lst), _stream())
        return gA, gBc, None, None, None, None


def edge_act(A, Bc, idx, slope=0.2):
    """E[b,i,j,:] = leaky(A[b,idx[b,i,j],:] + Bc[b,i,:]): A, Bc [B,N,C] (C % 4 == 0), idx [B,N,K] int32 -> [B,N,K,C];
    differentiable in A and Bc."""
    _check(A, "A"), _check(Bc, "Bc")
    if A.shape != Bc.shape or A.shape[2] % 4 or idx.dtype != torch.int32 or idx.shape[:2] != A.shape[:2]:
        raise ValueError("edge_act: A, Bc [B,N,C] with C % 4 == 0 and idx [B,N,K] int32 expected")
    rev = _rev_of(idx, A.shape[1], A.requires_grad)
    return _EdgeActFn.apply(A.contiguous(), Bc.contiguous(), idx.contiguous(), slope, *(rev or (None, None)))


class _ActMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Z, slope):
        B, N, K, C = Z.shape
        out = torch.empty((B, N, C), dtype=torch.float32, device=Z.device)
        with torch.cuda.device(Z.device):
            _lib.call("pc3d_act_mean_f32", Z.data_ptr(), B, N, K, C, float(slope), out.data_ptr(), _stream())
        ctx.save_for_backward(Z)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        (Z,) = ctx.saved_tensors
        B, N, K, C = Z.shape
        g = g.contiguous()
        gZ = torch.empty_like(Z)
        with torch.cuda.device(Z.device):
            _lib.call("pc3d_act_mean_bwd_f32", Z.data_ptr(), g.data_ptr(), B, N, K, C, ctx.slope, gZ.data_ptr(), _stream())
        return gZ, None


def act_mean(Z, slope=0.2):
    """out[b,i,:] = mean_j leaky(Z[b,i,j,:]) for Z [B,N,K,C] (C % 4 == 0) in one pass; differentiable."""
    _check(Z, "Z")
    if Z.dim() != 4 or Z.shape[3] % 4:
        raise ValueError("act_mean: Z must be [B,N,K,C] with C % 4 == 0")
    return _ActMeanFn.apply(Z.contiguous(), slope)


LPFA_FUSED_CHANNELS = (16, 32, 64, 128)
LPFA_FUSED_MAX_K = 30


class _LpfaFusedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, Bc, idx, W, b, s1, s2, rev_off, rev_lst):
        B, N, C = A.shape
        out = torch.empty_like(A)
        with torch.cuda.device(A.device):
            _lib.call("pc3d_lpfa_fused_f32", A.data_ptr(), Bc.data_ptr(), idx.data_ptr(), W.data_ptr(), _ptr(b), B, N,
                      idx.shape[2], C, float(s1), float(s2), out.data_ptr(), _stream())
        ctx.save_for_backward(A, Bc, idx, W, b, rev_off, rev_lst)
        ctx.slopes = (float(s1), float(s2))
        return out

    @staticmethod
    def backward(ctx, g):
        A, Bc, idx, W, b, rev_off, rev_lst = ctx.saved_tensors
        B, N, C = A.shape
        g = g.contiguous()
        gA, gBc = torch.empty_like(A), torch.empty_like(A)
        # deterministic: the per-edge gradients [B,N,K,C] go through memory and are summed in edge order (csrc/det.hip)
        det = _det() or rev_off is not None
        scratch = torch.empty((B, N, idx.shape[2], C), dtype=torch.float32, device=A.device) if det else None
        with torch.cuda.device(A.device):
            _lib.call("pc3d_lpfa_fused_bwd_f32", g.data_ptr(), A.data_ptr(), Bc.data_ptr(), idx.data_ptr(), W.data_ptr(),
                      _w_transposed(W).data_ptr(), _ptr(b), B, N, idx.shape[2], C, ctx.slopes[0], ctx.slopes[1],
                      gA.data_ptr(), gBc.data_ptr(), _ptr(scratch), _ptr(rev_off), _ptr(rev_lst), _stream())
        return gA, gBc, None, None, None, None, None, None, None


def lpfa_fused_supported(C, Cout, K):
    return C == Cout and C in LPFA_FUSED_CHANNELS and 1 <= K <= LPFA_FUSED_MAX_K


def lpfa_fused(A, Bc, idx, W, b, slope1=0.2, slope2=0.2):
    """mean_j leaky_s2(leaky_s1(A[idx[:, :, j]] + Bc) @ W.T + b) for A, Bc [B,N,C], idx [B,N,K] int32, frozen W [C,C] /
    b [C]: CurveNet's LPFA block with one MLP layer in one launch each way (no [B,N,K,C] tensor); differentiable in A
    and Bc."""
    _check(A, "A"), _check(Bc, "Bc")
    B, N, C = A.shape
    if A.shape != Bc.shape or idx.dtype != torch.int32 or idx.shape[:2] != (B, N) or not lpfa_fused_supported(C, W.shape[0], idx.shape[2]) \
            or W.shape != (C, C):
        raise ValueError("lpfa_fused: A, Bc [B,N,C] (C in 16/32/64/128), idx int32 [B,N,K<=30], W [C,C] expected")
    W = W.detach().contiguous().float()
    rev = _rev_of(idx, N, A.requires_grad or Bc.requires_grad)
    return _LpfaFusedFn.apply(A.contiguous(), Bc.contiguous(), idx.contiguous(), W,
                              b.detach().contiguous().float() if b is not None else None, slope1, slope2,
                              *(rev or (None, None)))


def h2d(t, device, dtype=None):
    """Host tensor -> device WITHOUT stalling the launch queue. A plain `.to(device)` of a pageable host tensor is a
    synchronous copy that first waits for everything already queued on the stream, so one such call per iteration
    (an FPS start index, a jitter draw, a search-step constant) removes the host's run-ahead: measured 0.6 ms per
    iteration in GeoA3 on DGCNN. Staged through pinned memory the copy is just another stream operation."""
    device = torch.device(device)
    if dtype is not None:
        t = t.to(dtype)
    if t.device.type != "cpu" or device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


# ------------------------------------------------------------------------------------------------------
# K18: CurveNet curve aggregation, per-cloud half (attention keys / values from the curves)
# ------------------------------------------------------------------------------------------------------
CURVE_AGG_LDS_LIMIT = 160 * 1024      # the CU's whole LDS (csrc/curve_agg.hip raises the kernels' window to it)


def curve_agg_lds_bytes(cn, cl, C, mid, backward=True):
    return int(_lib.load().pc3d_curve_agg_lds_bytes(int(cn), int(cl), int(C), int(mid), int(bool(backward))))


class _CurveAggKVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, curves, *ws):
        B, cn, cl, C = curves.shape
        mid = ws[1].shape[0]
        R = cn + cl
        Kp = torch.empty((B, C, R), dtype=torch.float32, device=curves.device)
        Vp = torch.empty((B, R, C), dtype=torch.float32, device=curves.device)
        with torch.cuda.device(curves.device):
            _lib.call("pc3d_curve_agg_kv_f32", curves.data_ptr(), *[w.data_ptr() for w in ws], B, cn, cl, C, mid,
                      Kp.data_ptr(), Vp.data_ptr(), _stream())
        ctx.save_for_backward(curves, *ws)
        return Kp, Vp

    @staticmethod
    def backward(ctx, gKp, gVp):
        curves, *ws = ctx.saved_tensors
        B, cn, cl, C = curves.shape
        mid = ws[1].shape[0]
        gKp, gVp = gKp.contiguous(), gVp.contiguous()
        gc = torch.empty_like(curves)
        with torch.cuda.device(curves.device):
            _lib.call("pc3d_curve_agg_kv_bwd_f32", gKp.data_ptr(), gVp.data_ptr(), curves.data_ptr(),
                      *[w.data_ptr() for w in ws], B, cn, cl, C, mid, gc.data_ptr(), _stream())
        return (gc,) + (None,) * len(ws)


def curve_agg_kv(curves, w_att, Wa, Wb, Wn, Wl, Wc, Wd, bd):
    """Attention keys Kp [B,C,cn+cl] and values Vp [B,cn+cl,C] of CurveNet's curve aggregation from channels-last
    curves [B,cn,cl,C] and the block's (frozen, folded) weights — see include/pc3d.h K18; differentiable in curves."""
    _check(curves, "curves")
    B, cn, cl, C = curves.shape
    mid = Wa.shape[0]
    shapes = ((C,), (mid, C), (mid, C), (mid, mid), (mid, mid), (mid, C), (C, 2 * mid), (C,))
    ws = [w.detach().contiguous().float() for w in (w_att.reshape(-1), Wa, Wb, Wn, Wl, Wc, Wd, bd)]
    if any(tuple(w.shape) != sh for w, sh in zip(ws, shapes)):
        raise ValueError(f"curve_agg_kv: weight shapes must be {shapes}")
    if curve_agg_lds_bytes(cn, cl, C, mid) > CURVE_AGG_LDS_LIMIT:
        raise ValueError(f"curve_agg_kv: cn={cn} cl={cl} C={C} mid={mid} does not fit the CU's 160 KB of LDS")
    return _CurveAggKVFn.apply(curves.contiguous(), *ws)


# ------------------------------------------------------------------------------------------------------
# K19: channels-last glue of a CurveNet CIC block
# ------------------------------------------------------------------------------------------------------
def gate(g, y, slope):
    """y > 0 ? g : slope * g (one launch): the (Leaky)ReLU derivative taken from the activation's output."""
    g, y = g.contiguous(), y.contiguous()
    out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        _lib.call("pc3d_gate_f32", g.data_ptr(), y.data_ptr(), g.numel(), float(slope), out.data_ptr(), _stream())
    return out


class _LinearResActFn(torch.autograd.Function):
    """act(x @ w.T + b + r): the residual tail of a block in one launch; backward = gate + one GEMM on W^T."""

    @staticmethod
    def forward(ctx, x, w, b, r, act, slope):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        N = w.shape[0]
        r2 = r.reshape(-1, N)
        if r2.stride(1) != 1:
            r2 = r2.contiguous()
        M, K = x2.shape
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_gemm_nt_res_f32", x2.data_ptr(), x2.stride(0), w.data_ptr(), _ptr(b), r2.data_ptr(),
                      r2.stride(0), M, N, K, _ACTS[act], float(slope), y.data_ptr(), y.stride(0),
                      gemm_variant(_unit_rows(shp), N, K), _stream())
        ctx.act, ctx.slope, ctx.shp, ctx.rshp = act, slope, shp, r.shape
        ctx.save_for_backward(y if act in ("relu", "leaky") else None, w)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        y, w = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1])
        if y is not None:
            g2 = gate(g2, y, ctx.slope if ctx.act == "leaky" else 0.0)
        elif not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = gemm_nt(g2, _w_transposed(w), unit_rows=_unit_rows(ctx.shp)) if ctx.needs_input_grad[0] else None
        return (gx.view(ctx.shp) if gx is not None else None), None, None, g2.view(ctx.rshp), None, None


def linear_res_act(x, w, b, r, act=None, slope=0.0):
    """[..., K] -> [..., N]: act(x @ w.T + b + r) with r [..., N]; frozen (w, b), differentiable in x and r."""
    _check(x, "x"), _check(r, "r")
    w = w.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    return _LinearResActFn.apply(x, w, b.detach() if b is not None else None, r, act, float(slope))


class _GatherMaxRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, idx):
        B, N, C = P.shape
        S, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, S, C), dtype=torch.float32, device=P.device)
        arg = torch.empty((B, S, C), dtype=torch.int32, device=P.device)
        with torch.cuda.device(P.device):
            _lib.call("pc3d_gather_max_rows_f32", P.data_ptr(), idx.data_ptr(), B, N, S, C, K, out.data_ptr(),
                      arg.data_ptr(), _stream())
        ctx.save_for_backward(arg)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        g = g.contiguous()
        B, S, C = g.shape
        gP = torch.empty((B, ctx.N, C), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_gather_max_rows_bwd_f32", g.data_ptr(), arg.data_ptr(), B, ctx.N, S, C, gP.data_ptr(), _det(),
                      _stream())
        return gP, None


def gather_max_rows(P, idx):
    """out[b,s,c] = max_j P[b, idx[b,s,j], c] for P [B,N,C], idx [B,S,K] int32 (clamped to [0,N-1]) -> [B,S,C]."""
    _check(P, "P")
    if idx.dtype != torch.int32 or idx.dim() != 3 or idx.shape[0] != P.shape[0]:
        raise ValueError("gather_max_rows: idx must be int32 [B,S,K]")
    return _GatherMaxRowsFn.apply(P.contiguous(), idx.contiguous())


class _AttScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        C = x.shape[-1]
        M = x.numel() // C
        xs = torch.empty_like(x)
        att = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_att_scale_f32", x.data_ptr(), w.data_ptr(), M, C, xs.data_ptr(), att.data_ptr(), _stream())
        ctx.save_for_backward(x, w, att)
        ctx.mark_non_differentiable(att)
        ctx.set_materialize_grads(False)      # no zero tensor for the score output's "gradient" (a fill launch per call)
        return xs, att

    @staticmethod
    def backward(ctx, g, _gatt):
        if g is None:
            return None, None
        x, w, att = ctx.saved_tensors
        C = x.shape[-1]
        g = g.contiguous()
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_att_scale_bwd_f32", g.data_ptr(), x.data_ptr(), att.data_ptr(), w.data_ptr(), x.numel() // C,
                      C, gx.data_ptr(), _stream())
        return gx, None


def att_scale(x, w):
    """(x * att, att) with att = sigmoid(x @ w) for x [..., C] (C % 4 == 0) and a frozen w [C]; the scaled features
    are differentiable in x, att is returned for the (non-differentiable) start-point selection only."""
    _check(x, "x")
    if x.shape[-1] % 4 or w.numel() != x.shape[-1]:
        raise ValueError("att_scale: x [..., C] with C % 4 == 0 and w [C] expected")
    return _AttScaleFn.apply(x.contiguous(), w.detach().reshape(-1).contiguous().float())


TOPK_MAX_N = 8192


def topk_desc(score, K):
    """score [B,N] fp32 (N <= 8192) -> int32 [B,K]: indices of the K largest, descending, ties to the lower index."""
    _check(score, "score")
    score = score.detach().contiguous()
    B, N = score.shape
    idx = torch.empty((B, K), dtype=torch.int32, device=score.device)
    with torch.cuda.device(score.device):
        _lib.call("pc3d_topk_desc_f32", score.data_ptr(), B, N, int(K), idx.data_ptr(), _stream())
    return idx


CURVE_ATTN_CHANNELS = (8, 16, 32, 64)
CURVE_ATTN_MAX_R = 128


def curve_attn_supported(C, R):
    """Shapes pc3d_curve_attn_f32 takes: the keys and values of one cloud (2 R (C+4) floats) must fit 64 KiB of LDS."""
    return C in CURVE_ATTN_CHANNELS and 1 <= R <= CURVE_ATTN_MAX_R and 8 * R * (C + 4) <= 64 * 1024


class _CurveAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Kp, Vp, cn, slope):
        B, N, C = x.shape
        R = Kp.shape[2]
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_curve_attn_f32", x.data_ptr(), Kp.data_ptr(), Vp.data_ptr(), B, N, C, cn, R - cn, float(slope),
                      out.data_ptr(), _stream())
        ctx.save_for_backward(x, Kp, Vp, out)
        ctx.cn, ctx.slope = cn, float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        x, Kp, Vp, out = ctx.saved_tensors
        B, N, C = x.shape
        R = Kp.shape[2]
        g = g.contiguous()
        gx, gKp, gVp = torch.empty_like(x), torch.empty_like(Kp), torch.empty_like(Vp)
        ws = torch.empty(int(_lib.load().pc3d_curve_attn_bwd_ws_floats(B, N, C, ctx.cn, R - ctx.cn)), dtype=torch.float32,
                         device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_curve_attn_bwd_f32", g.data_ptr(), out.data_ptr(), x.data_ptr(), Kp.data_ptr(), Vp.data_ptr(),
                      B, N, C, ctx.cn, R - ctx.cn, ctx.slope, gx.data_ptr(), gKp.data_ptr(), gVp.data_ptr(), ws.data_ptr(),
                      _stream())
        return gx, gKp, gVp, None, None


def curve_attn(x, Kp, Vp, cn, slope=0.2):
    """leaky(x + softmax(x Kp[:, :, :cn]) Vp[:, :cn] + softmax(x Kp[:, :, cn:]) Vp[:, cn:]) for x [B,N,C], Kp [B,C,R],
    Vp [B,R,C] in one launch; differentiable in all three."""
    _check(x, "x"), _check(Kp, "Kp"), _check(Vp, "Vp")
    B, N, C = x.shape
    R = Kp.shape[2]
    if not curve_attn_supported(C, R) or Kp.shape != (B, C, R) or Vp.shape != (B, R, C) or not 0 <= cn <= R:
        raise ValueError("curve_attn: x [B,N,C] (C in 8/16/32/64), Kp [B,C,R], Vp [B,R,C] with R <= 128 expected")
    return _CurveAttnFn.apply(x.contiguous(), Kp.contiguous(), Vp.contiguous(), int(cn), slope)


class _LpfaPrepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pts, G1, G2, t):
        C = x.shape[-1]
        M = x.numel() // C
        A, Bc = torch.empty_like(x), torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_lpfa_prep_f32", x.data_ptr(), pts.data_ptr(), G1.data_ptr(), G2.data_ptr(), t.data_ptr(), M, C,
                      A.data_ptr(), Bc.data_ptr(), _stream())
        ctx.save_for_backward(G1, G2)
        ctx.pshape = pts.shape
        return A, Bc

    @staticmethod
    def backward(ctx, gA, gBc):
        G1, G2 = ctx.saved_tensors
        gA, gBc = gA.contiguous(), gBc.contiguous()
        C = gA.shape[-1]
        M = gA.numel() // C
        gx = torch.empty_like(gA)
        gp = torch.empty(ctx.pshape, dtype=torch.float32, device=gA.device)
        with torch.cuda.device(gA.device):
            _lib.call("pc3d_lpfa_prep_bwd_f32", gA.data_ptr(), gBc.data_ptr(), G1.data_ptr(), G2.data_ptr(), M, C,
                      gx.data_ptr(), gp.data_ptr(), _stream())
        return gx, gp, None, None, None


def lpfa_prep(x, pts, G1, G2, t):
    """(A, Bc) = (x + pts @ G1.T, pts @ G2.T + t - x) for x [..., C] (C % 4 == 0), pts [..., 3], frozen G1, G2 [C,3] and
    t [C]; differentiable in x and pts."""
    _check(x, "x"), _check(pts, "pts")
    C = x.shape[-1]
    if C % 4 or pts.shape[-1] != 3 or pts.shape[:-1] != x.shape[:-1] or G1.shape != (C, 3) or G2.shape != (C, 3):
        raise ValueError("lpfa_prep: x [..., C] (C % 4 == 0), pts [..., 3], G1, G2 [C,3] expected")
    f = lambda w: w.detach().contiguous().float()          # noqa: E731
    return _LpfaPrepFn.apply(x.contiguous(), pts.contiguous(), f(G1), f(G2), f(t))


# ------------------------------------------------------------------------------------------------------
# K16: CurveNet guided walk
# ------------------------------------------------------------------------------------------------------
CURVE_WALK_CHANNELS = (8, 16, 32, 64)


class _CurveWalkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, adj, start, aw, ab, mw, mb, L):
        B, N, C = feats.shape
        k, cn = adj.shape[2], start.shape[1]
        dev = feats.device
        curves = torch.empty((B, cn, L, C), dtype=torch.float32, device=dev)
        pre = torch.empty((B, cn, L, C), dtype=torch.float32, device=dev)
        mom = torch.empty((B, cn, L, 2), dtype=torch.float32, device=dev)
        nodes = torch.empty((B, cn, L), dtype=torch.int32, device=dev)
        pick = torch.empty((B, cn, L), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("pc3d_curve_walk_fwd_f32", feats.data_ptr(), adj.data_ptr(), start.data_ptr(), aw.data_ptr(),
                      ab.data_ptr(), mw.data_ptr(), mb.data_ptr(), B, N, C, k, cn, L, curves.data_ptr(),
                      nodes.data_ptr(), pick.data_ptr(), pre.data_ptr(), mom.data_ptr(), _stream())
        ctx.save_for_backward(feats, adj, aw, ab, mw, mb, curves, nodes, pick, pre, mom)
        ctx.L = L
        return curves

    @staticmethod
    def backward(ctx, g):
        feats, adj, aw, ab, mw, mb, curves, nodes, pick, pre, mom = ctx.saved_tensors
        B, N, C = feats.shape
        k, cn = adj.shape[2], nodes.shape[1]
        g = g.contiguous()
        det = _det()
        # (deterministic: both accumulators are overwritten by the ordered scatter; else one fill for both)
        acc = (torch.empty if det else torch.zeros)(B * N * (C + 1), dtype=torch.float32, device=g.device)
        gF, coef = acc[:B * N * C].view(B, N, C), acc[B * N * C:].view(B, N)
        ws = torch.empty(int(_lib.load().pc3d_curve_walk_bwd_ws_floats(B, cn, C, ctx.L, k, det)), dtype=torch.float32,
                         device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_curve_walk_bwd_f32", g.data_ptr(), feats.data_ptr(), adj.data_ptr(), aw.data_ptr(),
                      ab.data_ptr(), mw.data_ptr(), mb.data_ptr(), B, N, C, k, cn, ctx.L, curves.data_ptr(),
                      nodes.data_ptr(), pick.data_ptr(), pre.data_ptr(), mom.data_ptr(), gF.data_ptr(),
                      coef.data_ptr(), ws.data_ptr(), det, _stream())
        if not det:                                  # (deterministic: added by the ordered scatter on its way out)
            gF.addcmul_(coef.unsqueeze(-1), aw[:C])  # the rank-1 score term: every candidate row gets coef * w_nbr
        return gF, None, None, None, None, None, None, None


def curve_walk(feats, adj, start, agent_w, agent_b, mom_w, mom_b, length):
    """Guided walk of CurveNet (model/walk.py:74-153), one launch per step and direction: feats [B,N,C] (C in
    CURVE_WALK_CHANNELS), adj [B,N,k] int32 (k <= 64), start [B,cn] int32, folded agent / momentum weights
    ([2C], [1], [2,2C], [2]) -> curves [B,cn,length,C]; differentiable in feats (weights are frozen)."""
    _check(feats, "feats")
    B, N, C = feats.shape
    if C not in CURVE_WALK_CHANNELS or adj.shape[2] > 64:
        raise ValueError(f"curve_walk: C={C}, k={adj.shape[2]} (supported: C in {CURVE_WALK_CHANNELS}, k <= 64)")
    if adj.dtype != torch.int32 or start.dtype != torch.int32:
        raise TypeError("curve_walk: adj / start must be int32")
    if agent_w.numel() != 2 * C or agent_b.numel() != 1 or tuple(mom_w.shape) != (2, 2 * C) or mom_b.numel() != 2:
        raise ValueError("curve_walk: weight shapes must be [2C], [1], [2,2C], [2]")
    ws = [w.detach().contiguous().float() for w in (agent_w.reshape(-1), agent_b.reshape(-1), mom_w, mom_b.reshape(-1))]
    return _CurveWalkFn.apply(feats.contiguous(), adj.contiguous(), start.contiguous(), *ws, int(length))


# ------------------------------------------------------------------------------------------------------
# K12: AOF spectral front-end
# ------------------------------------------------------------------------------------------------------
def graph_laplacian(xyz, k=30, cf=True):
    """Dense L = D - A [B,N,N] of the symmetrised kNN-k Gaussian graph of xyz ([B,3,N] with cf, else [B,N,3])."""
    p, bs, ps, cs, B, N = _pts(xyz, cf, "xyz")
    _, idx = knn_raw(xyz, xyz, min(k, N), q_cf=cf, r_cf=cf)
    L = torch.empty((B, N, N), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.call("pc3d_graph_laplacian_f32", p, bs, ps, cs, idx.data_ptr(), B, N, idx.shape[2], L.data_ptr(), _stream())
    return L


def spectral_reproject(adv, V, Vt, low_pass, lfc=None, hfc=None, coeff=None):
    """AOF's re-projection of adv [B,3,N] onto the graph-frequency bands of the basis V [B,N,N] (Vt = its transpose, kept
    beside it): returns (lfc, hfc) = (adv V)[..., :lp] V[..., :lp]^T, (adv V)[..., lp:] V[..., lp:]^T, written into the
    given buffers when passed (TAOF_attack.py:114-126,164-170). No gradient (the reference runs it under no_grad)."""
    for nm, t in (("adv", adv), ("V", V), ("Vt", Vt)):
        _check(t, nm)
        if not t.is_contiguous():
            raise ValueError(f"spectral_reproject: {nm} must be contiguous")
    B, three, N = adv.shape
    if three != 3 or V.shape != (B, N, N) or Vt.shape != (B, N, N):
        raise ValueError("spectral_reproject: adv [B,3,N], V and Vt [B,N,N] expected")
    mk = lambda t: torch.empty((B, 3, N), dtype=torch.float32, device=adv.device) if t is None else t
    lfc, hfc, coeff = mk(lfc), mk(hfc), mk(coeff)
    for nm, t in (("lfc", lfc), ("hfc", hfc), ("coeff", coeff)):
        if t.shape != (B, 3, N) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != adv.device:
            raise ValueError(f"spectral_reproject: {nm} must be a contiguous fp32 [B,3,N] tensor on adv's device")
    with torch.cuda.device(adv.device):
        _lib.call("pc3d_spectral_reproject_f32", adv.data_ptr(), V.data_ptr(), Vt.data_ptr(), B, N, int(low_pass),
                  coeff.data_ptr(), lfc.data_ptr(), hfc.data_ptr(), _stream())
    return lfc, hfc


# ------------------------------------------------------------------------------------------------------
# Last 1x1 conv + ReLU + max over the group of a set-abstraction layer, with the sparse backward
# ------------------------------------------------------------------------------------------------------
GROUP_MAX_NS = 128


class _LinearReLUMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        # x [G,ns,C2] -> out [G,C3]; GEMM with bias+ReLU epilogue, then torch's max (values + winning row)
        G, ns, C2 = x.shape
        out, arg = _group_linear_max_fwd(x, w, b)      # fused fp32-MFMA kernel: no [G*ns, C3] activation
        ctx.save_for_backward(out, arg, w)
        ctx.shape = (G, ns, C2)
        return out

    @staticmethod
    def backward(ctx, g):
        out, arg, w = ctx.saved_tensors
        G, ns, C2 = ctx.shape
        g = g.contiguous()
        gx = torch.empty((G, ns, C2), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_group_max_linear_bwd_f32", g.data_ptr(), out.data_ptr(), arg.data_ptr(), w.data_ptr(),
                      G, ns, C2, w.shape[0], 0, gx.data_ptr(), _stream())
        return gx, None, None


def _group_linear_max_fwd(x, w, b):
    """(out [G,C3], arg [G,C3] int64) of max_r relu(x[g,r,:] @ w.T + b): fused MFMA kernel when the shape fits it."""
    G, ns, C2 = x.shape
    C3 = w.shape[0]
    if C2 % 8 == 0 and (C2 <= 128 or ns in (32, 64, 128)) and C3 % 32 == 0 and C3 <= 4096:
        out = torch.empty((G, C3), dtype=torch.float32, device=x.device)
        arg = torch.empty((G, C3), dtype=torch.int64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("pc3d_group_linear_max_f32", x.data_ptr(), w.data_ptr(), b.data_ptr(), G, ns, C2, C3,
                      out.data_ptr(), arg.data_ptr(), _stream())
        return out, arg
    y = gemm_nt(x.reshape(G * ns, C2), w, b, "relu")
    out = torch.empty((G, C3), dtype=torch.float32, device=x.device)
    arg = torch.empty((G, C3), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("pc3d_rows_max_f32", y.data_ptr(), G, ns, C3, out.data_ptr(), arg.data_ptr(), _stream())
    return out, arg


def gmb_ksplit(groups_per_cloud, C2, C3):
    """Channel split (1 or 4) of the max-backward launch for a layer with `groups_per_cloud` groups per cloud: 4 when a
    nominal batch of GEMM_NOMINAL_BATCH clouds leaves the chip mostly idle (a group-all layer). From the per-cloud shape
    only: the split changes the order of a row's sum."""
    if C3 % 32 or groups_per_cloud is None:
        return 1
    wgs = GEMM_NOMINAL_BATCH * int(groups_per_cloud) * ((C2 + (63 if C2 <= 64 else 127)) // (64 if C2 <= 64 else 128))
    return 4 if wgs < 512 else 1


class _MLPReLUMaxFn(torch.autograd.Function):
    """The whole shared MLP of a set-abstraction layer + the max over the group, x [G,ns,C0] -> [G,C_last], with a
    hand-written backward: hidden layers run on pc3d_gemm_nt_f32 (fp32 MFMA, bias + ReLU in the epilogue); the last
    layer + max is the fused MFMA launch; backward = sparse row accumulation through the max (which also applies the
    last hidden ReLU's mask), then per hidden layer ONE launch of the same kernel on W^T that applies the previous
    ReLU's mask while loading dY — no separate mask pass over the [G*ns, C] tensors. Frozen weights: only dL/dx."""

    @staticmethod
    def forward(ctx, x, gpc, *wb):
        G, ns, C0 = x.shape
        ctx.gpc = gpc
        ws, bs = wb[0::2], wb[1::2]
        acts = [x.reshape(G * ns, C0)]
        for w, b in zip(ws[:-1], bs[:-1]):
            acts.append(gemm_nt(acts[-1], w, b, "relu"))
        out, arg = _group_linear_max_fwd(acts[-1].view(G, ns, -1), ws[-1], bs[-1])
        ctx.save_for_backward(out, arg, *acts[1:], *ws)
        ctx.meta = (G, ns, C0, len(ws))
        return out

    @staticmethod
    def backward(ctx, g):
        G, ns, C0, nl = ctx.meta
        saved = ctx.saved_tensors
        out, arg = saved[0], saved[1]
        acts, ws = saved[2:2 + nl - 1], saved[2 + nl - 1:]
        g = g.contiguous()
        C2 = ws[-1].shape[1]
        gx = torch.empty((G * ns, C2), dtype=torch.float32, device=g.device)
        last_hidden = acts[-1] if nl > 1 else None                     # ReLU output feeding the last layer (or raw x)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_group_max_linear_bwd_ks_f32", g.data_ptr(), out.data_ptr(), arg.data_ptr(), ws[-1].data_ptr(),
                      G, ns, C2, ws[-1].shape[0], _ptr(last_hidden), gx.data_ptr(), gmb_ksplit(ctx.gpc, C2, ws[-1].shape[0]),
                      _stream())
        for li in range(nl - 2, -1, -1):                                # hidden layers, last to first
            # gradient wrt that layer's input; the incoming gradient is masked by the layer's own ReLU on load (the
            # last hidden layer's mask was applied by the max-backward kernel above)
            gx = gemm_nt(gx, _w_transposed(ws[li]), gate=(acts[li] if li < nl - 2 else None), gate_slope=0.0)
        return (gx.view(G, ns, C0), None) + (None,) * (2 * nl)


def group_reverse(idx, NA):
    """Reverse index of a grouping idx [B,S,K] int32 over NA points per cloud: (off [B,NA+1], lst [B,S*K+S]) int32 — per
    point the grouped rows that reference it (pc3d_group_reverse_i32). Depends on idx only; the set-abstraction layers
    build it once per forward on the geometry stream and their backward gathers through it instead of scattering with
    float atomics."""
    if idx.dtype != torch.int32 or idx.dim() != 3 or not idx.is_cuda:
        raise ValueError("group_reverse: idx must be an int32 [B,S,K] GPU tensor")
    idx = idx.contiguous()
    B, S, K = idx.shape
    dev = idx.device
    cnt = torch.empty((B, NA), dtype=torch.int32, device=dev)
    off = torch.empty((B, NA + 1), dtype=torch.int32, device=dev)
    lst = torch.empty((B, S * K + S), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pc3d_group_reverse_i32", idx.data_ptr(), B, NA, S, K, cnt.data_ptr(), off.data_ptr(), lst.data_ptr(),
                  _stream())
    return off, lst


def group_act_bwd_rev(gH, H, mask, idx, rev, NA, slope=0.0):
    """(gP [B,NA,C], gBc [B,S,C]) of group_act through the reverse index rev = group_reverse(idx, NA): no float atomics,
    every gP row written once. The activation's sign comes from H [B,S,K,C] or, with H None, from the bit mask."""
    off, lst = rev
    B, S, K = idx.shape
    C = gH.shape[-1]
    gH = gH.contiguous()
    gP = torch.empty((B, NA, C), dtype=torch.float32, device=gH.device)
    gBc = torch.empty((B, S, C), dtype=torch.float32, device=gH.device)
    tail = torch.empty((B, S, C), dtype=torch.float32, device=gH.device)
    with torch.cuda.device(gH.device):
        _lib.call("pc3d_group_act_bwd_rev_f32", gH.data_ptr(), _ptr(H), _ptr(mask), idx.data_ptr(), off.data_ptr(),
                  lst.data_ptr(), B, NA, S, K, C, float(slope), gP.data_ptr(), gBc.data_ptr(), tail.data_ptr(), _stream())
    return gP, gBc


# ------------------------------------------------------------------------------------------------------
# The front of a set-abstraction layer: per-point / per-centre forms of the first 1x1 convolution
# ------------------------------------------------------------------------------------------------------
def _xyz_view(t, name):
    """(ptr, bs, ps, cs, B, N) of a [B,N,3] fp32 view with any strides (a permuted channels-first tensor is read in place)."""
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.shape[2] == 3):
        raise TypeError(f"{name}: expected a float32 [B,N,3] GPU tensor, got {tuple(t.shape)} {t.dtype}")
    return t.data_ptr(), t.stride(0), t.stride(1), t.stride(2), t.shape[0], t.shape[1]


def _affine3_raw(x, w, bias, sign):
    p, bs, ps, cs, B, N = _xyz_view(x, "affine3")
    C = w.shape[0]
    out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("pc3d_affine3_f32", p, bs, ps, cs, B, N, w.data_ptr(), _ptr(bias), float(sign), C, out.data_ptr(), C, _stream())
    return out


def _affine3_bwd_raw(g, w, sign, add, out):
    """out [B,N,3] view (+)= sign * g [B,N,C] @ w [C,3] (+ add [B,N,3] view)."""
    B, N, C = g.shape
    g = g if g.is_contiguous() else g.contiguous()
    ap = (0, 0, 0, 0) if add is None else _xyz_view(add, "affine3_bwd add")[:4]
    op = _xyz_view(out, "affine3_bwd out")[:4]
    with torch.cuda.device(g.device):
        _lib.call("pc3d_affine3_bwd_f32", g.data_ptr(), C, B, N, C, w.data_ptr(), float(sign), *ap, *op, _stream())
    return out


def _cf_grad(B, N, device):
    """A [B,N,3] gradient laid out as a contiguous [B,3,N] tensor: what the permute of a channels-first input hands back
    to its producer without a copy."""
    return torch.empty((B, 3, N), dtype=torch.float32, device=device).permute(0, 2, 1)


class _Affine3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, sign):
        ctx.save_for_backward(w)
        ctx.sign = sign
        return _affine3_raw(x, w, bias, sign)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        B, N, _ = g.shape
        return _affine3_bwd_raw(g, w, ctx.sign, None, _cf_grad(B, N, g.device)), None, None, None


def affine3(x, w, bias=None, sign=1.0):
    """bias + sign * (x @ w.T) for a [B,N,3] view x (any strides) and a frozen w [C,3] (C % 4 == 0): [B,N,C]. The coordinate
    columns of a 1x1 convolution (pc3d_affine3_f32); differentiable in x."""
    _check(w, "w")
    if w.dim() != 2 or w.shape[1] != 3 or w.shape[0] % 4 or not w.is_contiguous():
        raise ValueError("affine3: w must be a contiguous [C,3] matrix with C % 4 == 0")
    return _Affine3Fn.apply(x, w.detach(), None if bias is None else bias.detach().contiguous(), float(sign))


class _SAFrontFn(torch.autograd.Function):
    """(new_xyz [B,S,3], P [B,N,C1], Bc [B,S,C1]) of a set-abstraction layer from xyz [B,N,3] (any strides), pts [B,N,D] or
    None, the centre indices fps_idx [B,S] and the split first layer (Wx [C1,3], Wf [C1,D], b1):
        new_xyz = xyz[fps_idx],  P = Wx x + Wf f,  Bc = b1 - Wx new_xyz
    (model/pointnet2_utils.py:113,118-135,190-197). Forward 3-4 launches; backward: the two three-column products, the
    centres' gradient (their own + Bc's) scattered into xyz's in ONE ordered launch, one GEMM for the features — no
    gather of P's rows, no negation, no autograd accumulation launches. The gradient of xyz is laid out channels-first."""

    @staticmethod
    def forward(ctx, xyz, pts, fps_idx, wx, wf, b1):
        p, bs, ps, cs, B, N = _xyz_view(xyz, "sa_front xyz")
        S = fps_idx.shape[1]
        C1 = wx.shape[0]
        dev = xyz.device
        new_xyz = torch.empty((B, S, 1, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("pc3d_group_gather_f32", p, bs, ps, cs, 0, 0, fps_idx.data_ptr(), 0, 0, 0, 0, B, N, S, 1,
                      new_xyz.data_ptr(), _stream())
        new_xyz = new_xyz.view(B, S, 3)
        px = _affine3_raw(xyz, wx, None, 1.0)
        if pts is None:
            P = px
        else:
            D = pts.shape[2]
            x2 = pts.reshape(B * N, D)
            P = torch.empty((B, N, C1), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.call("pc3d_gemm_nt_res_f32", x2.data_ptr(), x2.stride(0), wf.data_ptr(), 0, px.data_ptr(), C1,
                          B * N, C1, D, 0, 0.0, P.data_ptr(), C1, gemm_variant(N, C1, D), _stream())
        Bc = _affine3_raw(new_xyz, wx, b1, -1.0)
        ctx.save_for_backward(fps_idx, wx, wf)
        ctx.dims = (B, N, S, C1, pts is not None)
        ctx.set_materialize_grads(False)
        return new_xyz, P, Bc

    @staticmethod
    def backward(ctx, g_new, gP, gBc):
        fps_idx, wx, wf = ctx.saved_tensors
        B, N, S, C1, has_pts = ctx.dims
        dev = fps_idx.device
        gxyz = None
        if ctx.needs_input_grad[0] and (g_new is not None or gP is not None or gBc is not None):
            gxyz = _cf_grad(B, N, dev)
            if gP is not None:
                _affine3_bwd_raw(gP, wx, 1.0, None, gxyz)
            else:
                gxyz.zero_()
            gc = None
            if gBc is not None:
                gc = _affine3_bwd_raw(gBc, wx, -1.0, g_new, torch.empty((B, S, 3), dtype=torch.float32, device=dev))
            elif g_new is not None:
                gc = g_new.contiguous()
            if gc is not None:
                with torch.cuda.device(dev):
                    _lib.call("pc3d_scatter_points_det_f32", fps_idx.data_ptr(), gc.data_ptr(), B, S, N, gxyz.data_ptr(),
                              gxyz.stride(0), gxyz.stride(1), gxyz.stride(2), _stream())
        gpts = None
        if has_pts and ctx.needs_input_grad[1] and gP is not None:
            gP2 = gP.reshape(B * N, C1)
            gpts = gemm_nt(gP2 if gP2.stride(1) == 1 else gP2.contiguous(), _w_transposed(wf), unit_rows=N).view(B, N, -1)
        return gxyz, gpts, None, None, None, None


def sa_front(xyz, pts, fps_idx, wx, wf, b1):
    """(new_xyz [B,S,3], P [B,N,C1], Bc [B,S,C1]) — see _SAFrontFn. xyz [B,N,3] view with any strides, pts [B,N,D]
    contiguous or None, fps_idx int32 [B,S]; wx [C1,3] (C1 % 4 == 0), wf [C1,D], b1 [C1] frozen."""
    if fps_idx.dtype != torch.int32 or fps_idx.dim() != 2 or not fps_idx.is_contiguous():
        raise ValueError("sa_front: fps_idx must be a contiguous int32 [B,S] tensor")
    if wx.dim() != 2 or wx.shape[1] != 3 or wx.shape[0] % 4 or not wx.is_contiguous():
        raise ValueError("sa_front: wx must be a contiguous [C1,3] matrix with C1 % 4 == 0")
    if pts is not None:
        _check(pts, "pts")
        pts = pts if pts.is_contiguous() else pts.contiguous()
        if wf is None or wf.shape != (wx.shape[0], pts.shape[2]) or not wf.is_contiguous():
            raise ValueError("sa_front: wf must be a contiguous [C1,D] matrix")
    return _SAFrontFn.apply(xyz, pts, fps_idx, wx.detach(), None if pts is None else wf.detach(), b1.detach().contiguous())


# Keep layer 2's sign bits (pc3d_gemm_nt_gather_f32's ymask) instead of its output for the backward: 0.5 GB less per
# forward at SSG's sizes at the same speed (cfg4 3.43 / 3.44 ms per iteration with / without, once
# pc3d_group_max_linear_bwd_mask_f32 loads a wave's mask words in one go; with a word load per row and thread it was
# 30 us per level slower).
LAYER2_SIGN_BITS = True
# Layers 1-2-3 + group max as ONE launch (pc3d_sa_chain_f32): the layer-2 output never leaves the chip. False = the
# two-launch form of round 2 (bit-identical results; kept for A/B timing and as the fallback for other shapes).
SA_CHAIN = True


# The chain's backward GEMM on W2^T and the groups pass of the first layer's backward as one launch
# (pc3d_gemm_nt_groupsum_f32: the [B*S*ns, C1] gradient is written once and not read back for the sums), then the points
# pass — bit-identical to the separate launches; False = those, for A/B timing.
SA_CHAIN_BWD = True
# ... on the rows that won a channel of the group max only (pc3d_group_max_linear_bwd_sparse_f32 and the amask arguments
# of the two launches after it): the other rows carry exact zeros. Bit-identical; False = full tensors, for A/B timing.
SA_BWD_SPARSE = True
# ... with the GEMM + groups launch over tiles PACKED with whole groups' active rows (pc3d_gemm_nt_groupsum_packed_f32)
# instead of 128 consecutive rows. Bit-identical; False = consecutive rows, for A/B timing.
SA_BWD_PACKED = True
# The chain launch over a unit table (ops.sa_blocks): 8- / 16- / 32-row units of nothing but padding copies are left out and the rest
# packed into fewer tiles (same results; False = every block, for A/B timing).
SA_BLOCK_TABLE = True


def sa_chain_table_unit(S, ns, C1, C2, C3):
    """Rows per unit (8 / 16 / 32) of the table the chain launch takes for this shape, or 0 when a table would change nothing
    (pc3d_sa_chain_table_unit: the library's own dispatch rule)."""
    return int(_lib.load().pc3d_sa_chain_table_unit(int(S), int(ns), int(C1), int(C2), int(C3)))


def sa_blocks(idx, unit):
    """Unit table of a grouping idx [B,S,ns] int32 for the chain launch: (tb, ntiles, unit) — the 8- / 16- / 32-row units
    that hold at least one listed point, packed eight (8-row units) or four to a tile, whole groups per tile
    (pc3d_sa_blocks_i32). Depends on idx only: the geometry chain builds it right after the ball query. None when B * S
    exceeds what the packing launches take (64 K groups; 48 K for 32-row blocks)."""
    if idx.dtype != torch.int32 or idx.dim() != 3 or not idx.is_cuda or not idx.is_contiguous():
        raise ValueError("sa_blocks: idx must be a contiguous int32 [B,S,ns] GPU tensor")
    B, S, ns = idx.shape
    dev = idx.device
    if B * S > (48 if unit == 32 else 64) * 1024:         # the packing launches' limit: no table, the chain takes every row
        return None
    flags = torch.empty((B * S,), dtype=torch.uint8, device=dev)
    # 8-row units: 64-row tiles of eight slots, 16-row units: of four — at most one tile per group; 32-row blocks: 128-row
    # tiles of four
    n_tb = B * S * (8 if unit == 8 else 4) if unit != 32 else ((B * S * ns + 127) // 128) * 4
    tb = torch.empty((n_tb,), dtype=torch.int32, device=dev)
    nt = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("pc3d_sa_blocks_i32", idx.data_ptr(), B, S, ns, int(unit), flags.data_ptr(), tb.data_ptr(), nt.data_ptr(), _stream())
    return tb, nt, int(unit)


def sa_chain_supported(C1, C2, C3, ns):
    return (SA_CHAIN and ns in (32, 64, 128) and C1 in (32, 64, 128) and C2 % 32 == 0 and 32 <= C2 <= 128
            and C3 % 32 == 0 and LAYER2_SIGN_BITS)


class _GroupedMLPMaxFn(torch.autograd.Function):
    """A three-layer set-abstraction MLP + group max from the per-point form of its first layer, WITHOUT the layer-1
    output: H2 = relu(W2 relu(P[idx] + Bc) + b2) in one launch (pc3d_gemm_nt_gather_f32 generates the rows of its X
    operand on load), then the last layer + max (pc3d_group_linear_max_f32). Backward: sparse rows through the max,
    one GEMM on W2^T, and the scatter to P / Bc with the first ReLU's sign from the bit mask the forward GEMM wrote
    (pc3d_group_act_bwd_mask_f32). One [B,S,ns,C1] write and two reads fewer than group_act + mlp_relu_max."""

    @staticmethod
    def forward(ctx, P, Bc, idx, w2, b2, w3, b3, rev_off, rev_lst, rev_event=None, blocks=None):
        ctx.rev_event = rev_event
        B, NA, C1 = P.shape
        S, ns = idx.shape[1], idx.shape[2]
        C2 = w2.shape[0]
        mask = torch.empty((B * S * ns, C1 // 4), dtype=torch.uint8, device=P.device)
        if sa_chain_supported(C1, C2, w3.shape[0], ns) and P.stride(1) % 4 == 0:
            C3 = w3.shape[0]
            m2 = torch.empty((B * S * ns, C2 // 32), dtype=torch.int32, device=P.device)
            out = torch.empty((B * S, C3), dtype=torch.float32, device=P.device)
            arg = torch.empty((B * S, C3), dtype=torch.int64, device=P.device)
            tb = nt = None
            unit = 0
            if blocks is not None and blocks[2] == sa_chain_table_unit(S, ns, C1, C2, C3):
                tb, nt, unit = blocks[:3]
                if len(blocks) > 3 and blocks[3] is not None:      # built on the geometry stream: its own event
                    torch.cuda.current_stream(P.device).wait_event(blocks[3])
            with torch.cuda.device(P.device):
                _lib.call("pc3d_sa_chain_tb_f32", P.data_ptr(), P.stride(1), Bc.data_ptr(), idx.data_ptr(), B, NA, S, ns,
                          w2.data_ptr(), b2.data_ptr(), C1, C2, w3.data_ptr(), b3.data_ptr(), C3, mask.data_ptr(),
                          m2.data_ptr(), out.data_ptr(), arg.data_ptr(), _ptr(tb), _ptr(nt), unit, _stream())
            ctx.save_for_backward(out, arg, None, m2, mask, idx, w2, w3, rev_off, rev_lst)
            ctx.dims = (B, NA, C1)
            return out.view(B, S, -1)
        H2 = torch.empty((B * S * ns, C2), dtype=torch.float32, device=P.device)
        # layer 2's own sign bits (C2 % 32 == 0): H2 is then not kept for the backward at all
        m2 = torch.empty((B * S * ns, C2 // 32), dtype=torch.int32, device=P.device) if (C2 % 32 == 0 and LAYER2_SIGN_BITS) else None
        with torch.cuda.device(P.device):
            _lib.call("pc3d_gemm_nt_gather_f32", P.data_ptr(), C1, Bc.data_ptr(), idx.data_ptr(), B, NA, S, ns, 0.0,
                      w2.data_ptr(), b2.data_ptr(), C2, C1, _ACTS["relu"], 0.0, H2.data_ptr(), C2, mask.data_ptr(),
                      _ptr(m2), _stream())
        out, arg = _group_linear_max_fwd(H2.view(B * S, ns, C2), w3, b3)
        ctx.save_for_backward(out, arg, H2 if m2 is None else None, m2, mask, idx, w2, w3, rev_off, rev_lst)
        ctx.dims = (B, NA, C1)
        return out.view(B, S, -1)

    @staticmethod
    def backward(ctx, g):
        out, arg, H2, m2, mask, idx, w2, w3, rev_off, rev_lst = ctx.saved_tensors
        B, NA, C1 = ctx.dims
        S, ns = idx.shape[1], idx.shape[2]
        C2 = w2.shape[0]
        g = g.contiguous()
        fused = SA_CHAIN_BWD and rev_off is not None and ns in (32, 64, 128) and C1 in (32, 64, 128) and C2 % 32 == 0
        sparse = fused and m2 is not None and SA_BWD_SPARSE
        gz = torch.empty((B * S * ns, C2), dtype=torch.float32, device=g.device)
        amask = torch.empty((B * S, (ns + 31) // 32), dtype=torch.int32, device=g.device) if sparse else None
        with torch.cuda.device(g.device):
            if sparse:                 # only the rows that won a channel of the max are written (and read below)
                _lib.call("pc3d_group_max_linear_bwd_sparse_f32", g.data_ptr(), out.data_ptr(), arg.data_ptr(),
                          w3.data_ptr(), B * S, ns, C2, w3.shape[0], m2.data_ptr(), gz.data_ptr(), amask.data_ptr(), _stream())
            elif m2 is not None:
                _lib.call("pc3d_group_max_linear_bwd_mask_f32", g.data_ptr(), out.data_ptr(), arg.data_ptr(),
                          w3.data_ptr(), B * S, ns, C2, w3.shape[0], m2.data_ptr(), gz.data_ptr(), _stream())
            else:
                _lib.call("pc3d_group_max_linear_bwd_f32", g.data_ptr(), out.data_ptr(), arg.data_ptr(), w3.data_ptr(),
                          B * S, ns, C2, w3.shape[0], H2.data_ptr(), gz.data_ptr(), _stream())
        if fused:
            dev = g.device
            gh1 = torch.empty((B * S * ns, C1), dtype=torch.float32, device=dev)
            gBc = torch.empty((B, S, C1), dtype=torch.float32, device=dev)
            tail = torch.empty((B, S, C1), dtype=torch.float32, device=dev)
            gP = torch.empty((B, NA, C1), dtype=torch.float32, device=dev)
            w2t = _w_transposed(w2)
            with torch.cuda.device(dev):
                if sparse and SA_BWD_PACKED and ns <= 64:
                    G = B * S
                    scratch = torch.empty((G + 4 + (G + 3) // 4,), dtype=torch.int32, device=dev)
                    _lib.call("pc3d_gemm_nt_groupsum_packed_f32", gz.data_ptr(), C2, w2t.data_ptr(), mask.data_ptr(),
                              idx.data_ptr(), amask.data_ptr(), B, S, ns, C1, C2, gh1.data_ptr(), gBc.data_ptr(),
                              tail.data_ptr(), scratch.data_ptr(), _stream())
                else:
                    _lib.call("pc3d_gemm_nt_groupsum_f32", gz.data_ptr(), C2, w2t.data_ptr(), mask.data_ptr(), idx.data_ptr(),
                              _ptr(amask), B, S, ns, C1, C2, gh1.data_ptr(), gBc.data_ptr(), tail.data_ptr(), _stream())
                if ctx.rev_event is not None:    # built on the geometry stream, after the sampling chain (pointnet2_utils)
                    torch.cuda.current_stream(dev).wait_event(ctx.rev_event)
                # (no amask here: the rows the sparse launches leave unwritten are the ball query's padding copies, which the
                # reverse lists do not contain — the few other inactive rows were written as zeros above)
                _lib.call("pc3d_group_act_bwd_points_f32", gh1.data_ptr(), mask.data_ptr(), tail.data_ptr(), rev_off.data_ptr(),
                          rev_lst.data_ptr(), 0, B, NA, S, ns, C1, 0.0, gP.data_ptr(), _stream())
            return gP, gBc, None, None, None, None, None, None, None, None, None
        gh1 = gemm_nt(gz, _w_transposed(w2))
        if rev_off is not None:        # gather through the reverse index of the grouping: no float atomics
            if ctx.rev_event is not None:        # built on the geometry stream, after the sampling chain (pointnet2_utils)
                torch.cuda.current_stream(g.device).wait_event(ctx.rev_event)
            gP, gBc = group_act_bwd_rev(gh1.view(B, S, ns, C1), None, mask, idx, (rev_off, rev_lst), NA, 0.0)
            return gP, gBc, None, None, None, None, None, None, None, None, None
        gP = torch.empty((B, NA, C1), dtype=torch.float32, device=g.device)
        gBc = torch.empty((B, S, C1), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("pc3d_group_act_bwd_mask_f32", gh1.data_ptr(), mask.data_ptr(), idx.data_ptr(), B, NA, S, ns, C1,
                      0.0, gP.data_ptr(), gBc.data_ptr(), _det(), _stream())
        return gP, gBc, None, None, None, None, None, None, None, None, None


def grouped_mlp_max_supported(C1, ns, layers):
    """Shapes pc3d_gemm_nt_gather_f32 + the fused last layer take: [(W2,b2),(W3,b3)] after the per-point first layer."""
    if len(layers) != 2 or C1 % 4 or C1 > GROUP_ACT_MAX_C or ns > GROUP_MAX_NS:
        return False
    C2, C3 = layers[0][0].shape[0], layers[1][0].shape[0]
    return C2 % 8 == 0 and C2 <= 128 and C3 % 32 == 0 and C3 <= 4096


def grouped_mlp_max(P, Bc, idx, layers, rev=None, blocks=None):
    """max_j relu(W3 relu(W2 relu(P[b,idx[b,s,j]] + Bc[b,s]) + b2) + b3) -> [B,S,C3]; P [B,NA,C1], Bc [B,S,C1], idx
    [B,S,ns] int32, layers = [(W2,b2),(W3,b3)] frozen. Differentiable in P and Bc. rev = group_reverse(idx, NA): the
    backward gathers through it instead of scattering with float atomics."""
    _check(P, "P"), _check(Bc, "Bc")
    if not grouped_mlp_max_supported(P.shape[2], idx.shape[2], layers) or idx.dtype != torch.int32 \
            or Bc.shape != (P.shape[0], idx.shape[1], P.shape[2]):
        raise ValueError("grouped_mlp_max: unsupported shapes (see grouped_mlp_max_supported)")
    (w2, b2), (w3, b3) = layers
    r0, r1 = (rev[0], rev[1]) if rev is not None else (None, None)
    ev = rev[2] if rev is not None and len(rev) > 2 else None       # (off, lst[, event recorded after they were built])
    return _GroupedMLPMaxFn.apply(P.contiguous(), Bc.contiguous(), idx.contiguous(), w2.detach().contiguous(),
                                  b2.detach().contiguous(), w3.detach().contiguous(), b3.detach().contiguous(), r0, r1, ev,
                                  blocks if SA_BLOCK_TABLE else None)


def mlp_relu_max(x, layers):
    """x [..., ns, C0], layers [(w, b), ...] frozen -> max over dim -2 of the ReLU MLP's output: [..., C_last]."""
    _check(x, "x")
    lead, ns, C0 = x.shape[:-2], x.shape[-2], x.shape[-1]
    if ns > GROUP_MAX_NS:
        raise ValueError(f"mlp_relu_max: group size {ns} exceeds {GROUP_MAX_NS}")
    flat = [t.contiguous() for wb in layers for t in wb]
    gpc = 1
    for d in lead[1:]:
        gpc *= int(d)
    out = _MLPReLUMaxFn.apply(x.reshape(-1, ns, C0).contiguous(), gpc if len(lead) >= 2 else None, *flat)
    return out.view(*lead, layers[-1][0].shape[0])


def linear_relu_max(x, w, b):
    """max over dim -2 of relu(x @ w.T + b) for x [..., ns, C2] with frozen (w [C3,C2], b): [..., C3]. The backward
    routes each channel's gradient to its winning row only (pc3d_group_max_linear_bwd_f32)."""
    _check(x, "x"), _check(w, "w"), _check(b, "b")
    lead, ns, C2 = x.shape[:-2], x.shape[-2], x.shape[-1]
    if ns > GROUP_MAX_NS:
        raise ValueError(f"linear_relu_max: group size {ns} exceeds {GROUP_MAX_NS}")
    out = _LinearReLUMaxFn.apply(x.reshape(-1, ns, C2).contiguous(), w.contiguous(), b.contiguous())
    return out.view(*lead, w.shape[0])
