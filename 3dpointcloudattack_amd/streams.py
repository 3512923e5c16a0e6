"""Process-wide side streams.

ROCm maps HIP streams onto a small number of hardware queues (four by default); streams that share a queue run their
work in submission order, so a "side" stream created on every call sooner or later lands on the main stream's queue and
the overlap it was created for turns into serialisation. The package therefore uses exactly two side streams per device, created once:

    GEOMETRY  farthest-point sampling, ball queries and kNN graphs of a victim's forward (they depend on coordinates only)
    TERMS     an attack's distance / regulariser terms, beside the victim's forward and backward
"""
import torch

GEOMETRY, TERMS = 0, 1
_STREAMS = {}


def side_stream(device, slot):
    """The process-wide side stream `slot` (GEOMETRY or TERMS) of a CUDA/HIP device."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _STREAMS.get((idx, slot))
    if st is None:
        # Ordinary priority on purpose. A high-priority geometry stream was tried (its first kernel gates the feature
        # path): it made a graphed CurveNet forward + backward with the forked geometry branch 1.8x SLOWER in a fresh
        # process (GeoA3 on CurveNet 7.1 -> 12.8 ms per iteration, tools/exp/geoa3_curvenet_stats.py) and did nothing
        # measurable for PointNet++.
        st = _STREAMS[(idx, slot)] = torch.cuda.Stream(device=torch.device("cuda", idx))
    return st
