"""hipGraph replay of a frozen victim's forward and backward.

The victims with many small layers (CurveNet: ~900 launches per forward+backward at B=32, N=4096) are bound by the
host's launch rate, not by the GPU, once their hot ops are single kernels. ``GraphedVictim`` captures one hipGraph for
the forward and one for the backward of ``model(x)`` per input shape (``torch.cuda.make_graphed_callables``) and
replays them from inside ordinary autograd code, so the attack loops (attack/{CW,GeoA3,KNN}) stay as they are.

What makes a replay safe here, and what is checked:
  * only victims that declare ``deterministic_forward`` (a pure function of the input: no RNG, no data-dependent
    shapes) are wrapped — ``wrap()`` returns any other model unchanged;
  * a graph holds its intermediates in its own memory, so a second forward would overwrite what the first one's
    backward needs. Calls with gradients enabled and calls under ``no_grad`` replay separate captures, outputs are
    returned as copies, and a gradient-enabled call that arrives while an earlier one still waits for its backward
    (its autograd node is alive) replays another replica of the capture (up to MAX_REPLICAS), then runs eagerly;
  * the capture is keyed on the version counters of the victim's parameters and buffers: loading other weights or
    moving the model re-captures.
"""
import contextlib
import gc
import weakref

import torch
import torch.nn as nn

from . import streams as _streams


_KEEP_STACK = []     # one list per capture in progress: tensors handed out by process-wide caches during it


def note_captured(*tensors):
    """Called by the operator layer's process-wide caches (transposed weights, search workspaces) for every tensor they
    hand out: while a capture_guard() is open the tensors join that capture's keep-alive list, so the graph that bakes
    their addresses in owns a reference — the cache itself may then evict the entry."""
    if _KEEP_STACK:
        _KEEP_STACK[-1].extend(t for t in tensors if t is not None)


def capturing():
    """True while a capture_guard() of this module is open (warm-up passes of a capture included)."""
    return bool(_KEEP_STACK)


@contextlib.contextmanager
def capture_guard():
    """Run a hipGraph capture with Python's cyclic garbage collector off (after one explicit collection); yields the
    capture's keep-alive list (see note_captured) — whoever owns the graph must hold on to it.
    A victim and its wrapper reference each other, as do an attack's state and its graph runner, so captured graphs
    die in the CYCLE collector, at whatever allocation happens to trigger it — if that is inside another capture, the
    graph's destructor runs HIP calls that are illegal while a stream is capturing and the process aborts (seen once
    in a long test session: "Fatal Python error: Aborted ... Garbage-collecting" under torch.cuda.current_stream)."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    keep = []
    _KEEP_STACK.append(keep)
    try:
        yield keep
    finally:
        assert _KEEP_STACK and _KEEP_STACK[-1] is keep      # guards nest strictly; pop by identity, never by list equality
        _KEEP_STACK.pop()
        if was:
            gc.enable()

MAX_CAPTURES = 4     # distinct (shape, mode, weights) keys per wrapper; the oldest is dropped beyond this
MAX_REPLICAS = 4     # captures of ONE key that may wait for their backward at the same time (EOT-style loops that
                     # run several forwards before one backward); further forwards run eagerly


class _Token:
    """Lives exactly as long as the autograd node of one replayed forward (it hangs off that node's ctx)."""
    __slots__ = ("__weakref__",)


class _Guard(torch.autograd.Function):
    """Copies the graph's static outputs and ties the backward to the capture slot that produced them."""

    @staticmethod
    def forward(ctx, slot, ticket, *outs):
        ctx.slot, ctx.ticket = slot, ticket
        ctx.token = _Token()
        slot.token_ref = weakref.ref(ctx.token)
        return tuple(o.clone() for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        slot = ctx.slot
        if slot.ticket != ctx.ticket:
            raise RuntimeError("GraphedVictim: backward of a forward whose graph memory has since been reused")
        slot.pending = False
        return (None, None) + grads


class _OwnGrad(torch.autograd.Function):
    """Identity whose backward clones: ``make_graphed_callables`` hands the graph's STATIC grad-input buffer to
    autograd, and AccumulateGrad may adopt it as ``x.grad`` of a leaf input — a caller that accumulates over two
    backward passes without clearing ``.grad`` would then read the second replay's values twice. One [B,3,N] copy."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.clone()


def note_input_knn(model, x, idx):
    """Called by a victim's forward that has built a neighbour graph of its INPUT cloud x [B,3,N]: idx int32 [B,N,K], the K
    nearest points of every point in ascending distance, itself first. An attack that needs the same graph with fewer
    columns (GeoA3's curvature term: 17 of DGCNN's / CurveNet's 20) takes it from here instead of searching again."""
    object.__setattr__(model, "_input_knn", (x.data_ptr(), x._version, tuple(x.shape), idx))


def input_knn(net, x, k1):
    """idx[:, :, :k1] (contiguous int32) of the neighbour graph `net` built for exactly this x in its last forward, or None."""
    rec = net.__dict__.get("_input_knn") if isinstance(net, nn.Module) else None
    if rec is None or rec[3] is None or rec[0] != x.data_ptr() or rec[1] != x._version or rec[2] != tuple(x.shape):
        return None
    idx = rec[3]
    if idx.dim() != 3 or idx.shape[0] != x.shape[0] or idx.shape[1] != x.shape[2] or idx.shape[2] < k1 or idx.dtype != torch.int32:
        return None
    return idx[:, :, :k1].contiguous()


class _Slot:
    def __init__(self, fn):
        self.fn = fn
        self.ticket = 0
        self.pending = False
        self.token_ref = None

    def busy(self):
        """A gradient-enabled forward was replayed here and its autograd node is still alive without having run its
        backward: the graph's memory holds what that backward will read."""
        return self.pending and self.token_ref is not None and self.token_ref() is not None


def _cached_tensors(model):
    """Every tensor reachable from plain attributes of the model's modules (weight caches; not parameters/buffers)."""
    found, seen = [], set()

    def walk(v, depth=0):
        if torch.is_tensor(v):
            if id(v) not in seen:
                seen.add(id(v))
                found.append(v)
        elif isinstance(v, (tuple, list)) and depth < 6:
            for e in v:
                walk(e, depth + 1)
        elif isinstance(v, dict) and depth < 6:
            for e in v.values():
                walk(e, depth + 1)

    for mod in model.modules():
        for name, val in vars(mod).items():
            if name not in ("_parameters", "_buffers", "_modules") and not isinstance(val, nn.Module):
                walk(val)
    return found


class GraphedVictim(nn.Module):
    """``model`` with its forward/backward replayed from hipGraphs. Attribute access falls through to the model
    (``fused_attack_grad``, ``deterministic_forward``, ...)."""

    def __init__(self, model, warmup=3):
        super().__init__()
        self.model = model
        self._warmup = warmup
        self._slots = {}
        self.stats = {"replayed": 0, "eager": 0, "captures": 0}

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("model"), name)

    # captures are tied to this process's device memory: copies and pickles of the wrapper (a deepcopy of the victim
    # takes its cached wrapper along) start without them
    def __deepcopy__(self, memo):
        import copy
        return GraphedVictim(copy.deepcopy(self.model, memo), warmup=self._warmup)

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_slots"] = {}
        st["stats"] = {"replayed": 0, "eager": 0, "captures": 0}
        return st

    def state_dict(self, *a, **k):
        return self.model.state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        return self.model.load_state_dict(*a, **k)

    def _weights_key(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.model.parameters()) + list(self.model.buffers()))

    def _capture(self, x, with_grad):
        """Capture under a saved host RNG state: the probe / warm-up / capture passes run the Python forward several
        times, the call they serve counts as ONE forward (its draws are re-issued by consume_forward_rng)."""
        rng = torch.get_rng_state()
        try:
            with capture_guard() as keep:
                slot = self._capture_impl(x, with_grad)
                slot.keepalive = slot.keepalive + keep     # W^T / workspace tensors the graphs point at
                return slot
        finally:
            torch.set_rng_state(rng)

    def _capture_impl(self, x, with_grad):
        model = self.model
        probe = x.detach().clone().requires_grad_(with_grad)
        with torch.set_grad_enabled(with_grad):
            out = model(probe)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        uniq, where = [], []                 # where[i]: index into the graphed outputs, or ("const", value)
        for o in outs:                       # victims return the same tensor several times (x, x, x)
            if not torch.is_tensor(o):
                where.append(("const", o))
                continue
            for i, u in enumerate(uniq):
                if u is o:
                    where.append(i)
                    break
            else:
                where.append(len(uniq))
                uniq.append(o)
        picks = [next(i for i, o in enumerate(outs) if o is u) for u in uniq]
        is_seq = isinstance(out, (tuple, list))

        def fn(inp):
            o = model(inp)
            o = o if isinstance(o, (tuple, list)) else (o,)
            return tuple(o[i] for i in picks)

        if with_grad:
            g = torch.cuda.make_graphed_callables(fn, (x.detach().clone().requires_grad_(True),),
                                                  num_warmup_iters=self._warmup)
        else:
            static_in = x.detach().clone()
            side = _streams.side_stream(x.device, _streams.TERMS)     # warm-up passes; ONE per process (streams.py)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self._warmup):
                    fn(static_in)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph):
                static_out = fn(static_in)

            def g(inp):
                static_in.copy_(inp)
                graph.replay()
                return static_out
        slot = _Slot(g)
        slot.where, slot.is_seq = where, is_seq
        rec = model.__dict__.get("_input_knn")           # the capture pass's neighbour graph: static memory of this capture,
        slot.input_knn = rec[3] if rec else None         # rewritten by every replay (see note_input_knn)
        # The graphs hold raw pointers to every tensor the forward read, including the victims' folded-weight caches
        # (created lazily, re-folded when weights change): keep those alive with the capture.
        slot.keepalive = _cached_tensors(model)
        self.stats["captures"] += 1
        return slot

    def forward(self, x):
        model = self.model
        if (not torch.is_tensor(x) or not x.is_cuda or model.training or not getattr(model, "deterministic_forward", False)
                or torch.cuda.is_current_stream_capturing()):
            out = model(x)
            object.__setattr__(self, "_input_knn", model.__dict__.get("_input_knn"))
            return out
        with_grad = torch.is_grad_enabled() and x.requires_grad
        from . import ops as _ops               # the backward kernel flavour (ordered sums / float atomics) is baked into a capture
        key = (tuple(x.shape), x.dtype, x.device, with_grad, bool(_ops._det()), self._weights_key())
        slots = self._slots.get(key)
        if slots is None:
            while len(self._slots) >= MAX_CAPTURES:
                self._slots.pop(next(iter(self._slots)))
            with torch.cuda.device(x.device):            # capture on the input's device, whatever the current one is
                slots = self._slots[key] = [self._capture(x, with_grad)]
        slot = next((s for s in slots if not s.busy()), None) if with_grad else slots[0]
        if slot is None and len(slots) < MAX_REPLICAS:
            with torch.cuda.device(x.device):
                slot = self._capture(x, with_grad)
            slots.append(slot)
        if slot is None:
            self.stats["eager"] += 1         # every replica still waits for its backward: do not touch their memory
            out = model(x)
            object.__setattr__(self, "_input_knn", model.__dict__.get("_input_knn"))
            return out
        self.stats["replayed"] += 1
        hook = getattr(model, "consume_forward_rng", None)
        if hook is not None:
            hook(x)                          # host-side RNG draws of the forward the replay stands in for
        with torch.cuda.device(x.device):
            outs = slot.fn(_OwnGrad.apply(x) if with_grad else x)
        if with_grad:
            slot.ticket += 1
            slot.pending = True
            outs = _Guard.apply(slot, slot.ticket, *outs)
        else:
            outs = tuple(o.clone() for o in outs)
        object.__setattr__(self, "_input_knn", (x.data_ptr(), x._version, tuple(x.shape), slot.input_knn))
        res = tuple(w[1] if isinstance(w, tuple) else outs[w] for w in slot.where)
        return res if slot.is_seq else res[0]


def wrap(model, enable=True):
    """``GraphedVictim(model)`` for victims that declare a deterministic forward, the model itself otherwise. enable:
    True / False, or None = the victim's own ``graph_replay_default`` (True unless it says otherwise: for a victim of a
    few dozen chip-filling launches the replay's static-buffer copies and per-node cost outweigh what the host saves)."""
    if enable is None:
        enable = getattr(model, "graph_replay_default", True)
    if not enable or isinstance(model, GraphedVictim) or not getattr(model, "deterministic_forward", False):
        return model
    g = model.__dict__.get("_pc3d_graphed")
    if g is None:                            # kept on the model (not as a sub-module) so that repeated attack calls
        g = GraphedVictim(model)             # on the same victim reuse the captures
        object.__setattr__(model, "_pc3d_graphed", g)
    return g
