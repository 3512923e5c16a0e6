"""Experiment: do small kernels on a second stream make progress while the forward tower kernel fills the chip?"""
import importlib, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
ws = tuple(t.to(dev) for t in (torch.randn(64,3), torch.randn(64), torch.randn(128,64)/8, torch.randn(128), torch.randn(1024,128)/11, torch.randn(1024)))
x = torch.randn(32, 3, 1024, device=dev)
a = torch.randn(32, 512, device=dev); w = torch.randn(256, 512, device=dev); b = torch.randn(256, device=dev)
p = torch.randn(32, 8, 1024, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def towers(n):
    for _ in range(n): ops.pointmlp3_max_fwd_raw(x, ws, False)
ctr = torch.zeros(1, dtype=torch.int32, device=dev)
cl_out = torch.empty_like(x)
def smalls(n):
    for _ in range(n):
        if os.environ.get("SMALL", "i32") == "i32": ops.i32_add(ctr, 1)
        else: ops.clip(x, x, budget=0.1, out=cl_out)
def timed(fA, fB):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    if fA:
        with torch.cuda.stream(sA):
            eb[0].record(); fA(); eb[1].record()
    if fB:
        with torch.cuda.stream(sB):
            eb[2].record(); fB(); eb[3].record()
    torch.cuda.synchronize()
    return (eb[0].elapsed_time(eb[1]) if fA else None, eb[2].elapsed_time(eb[3]) if fB else None, (time.perf_counter()-t0)*1e3)
towers(3); smalls(3)
# use graphs so launch overhead does not hide the effect
def graph_of(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g
gA = graph_of(lambda: towers(40), sA)
gB = graph_of(lambda: smalls(400), sB)
def rA():
    gA.replay()
def rB():
    gB.replay()
print("towers alone (ms A, B, wall):", timed(rA, None))
print("smalls alone:", timed(None, rB))
print("both:", timed(rA, rB))
print("both:", timed(rA, rB))
