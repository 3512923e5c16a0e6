"""EXPERIMENT: do small kernels on the main stream start while ONE long, narrow kernel (farthest-point sampling: 64
workgroups for ~330 us) runs on a side stream? For the k-th stream created in the process as the side stream: time from
queueing five small launches on the main stream to their completion, with and without the long kernel beside them."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((64, 2048, 3)).astype(np.float32)).to(dev)
w = torch.from_numpy(rng.standard_normal((64, 3)).astype(np.float32)).to(dev)
start = torch.zeros(64, dtype=torch.int32, device=dev)
main = torch.cuda.current_stream()
def small():
    for _ in range(5):
        ops.affine3(x, w)
def measure(side, with_long):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(main)
    if with_long:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.fps(x, 512, start)
            e2.record(side)
    small()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, (e0.elapsed_time(e2) * 1e3 if with_long else 0.0)
streams = [torch.cuda.Stream() for _ in range(8)]
for _ in range(3):
    measure(streams[0], True)
print("small launches alone: %.1f us" % measure(streams[0], False)[0])
for k, s in enumerate(streams):
    r = [measure(s, True) for _ in range(5)]
    print("side = stream #%d (%s): small launches done after %.1f us, long kernel after %.1f us"
          % (k, hex(s.cuda_stream)[-6:], sorted(a for a, _ in r)[2], sorted(b for _, b in r)[2]))
for prio in (-1, 0):
    s = torch.cuda.Stream(priority=prio)
    r = [measure(s, True) for _ in range(5)]
    print("side priority %d: small after %.1f us, long after %.1f us" % (prio, sorted(a for a, _ in r)[2], sorted(b for _, b in r)[2]))
