"""Process-wide side streams.

ROCm maps HIP streams onto a small number of hardware queues (four by default); streams that share a queue run their
work in submission order, so a "side" stream created on every call sooner or later lands on the main stream's queue and
the overlap it was created for turns into serialisation (measured: cfg4 3.71 ms/iter with the first pair of side streams
of a process, 4.12 with the third). The package therefore uses exactly two side streams per device, created once:

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
        # the geometry chain gates the victim's feature path (its first kernel is a long one-workgroup-per-cloud chain):
        # high priority, so its workgroups are placed ahead of the bulk kernels of the other two streams
        st = _STREAMS[(idx, slot)] = torch.cuda.Stream(device=torch.device("cuda", idx),
                                                       priority=-1 if slot == GEOMETRY else 0)
    return st
