"""Does a hipMemsetAsync issued inside a captured region (as pc3d_edge_max_bwd_f32 does) run on every replay?"""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
_lib = ops._lib
dev = torch.device("cuda:0")
B, N, C = 4, 1024, 64
gen = torch.Generator().manual_seed(0)
out = torch.randn(B, N, C, generator=gen).to(dev)
arg = torch.randint(0, N, (B, N, C), generator=gen).to(dev).to(torch.int32)
g = torch.randn(B, N, C, generator=gen).to(dev)
gPQ = torch.full((B, N, 2 * C), 7.0, device=dev)
def run(stream):
    _lib.call("pc3d_edge_max_bwd_f32", g.data_ptr(), C, out.data_ptr(), arg.data_ptr(), B, N, C, 0.2, gPQ.data_ptr(), stream)
run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
ref = gPQ.clone()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    run(torch.cuda.current_stream().cuda_stream)
for t in range(3):
    gPQ.fill_(7.0)
    graph.replay(); torch.cuda.synchronize()
    print("replay", t, "max diff vs eager", float((gPQ - ref).abs().max()), flush=True)
graph.replay(); graph.replay(); torch.cuda.synchronize()
print("3 replays without refill", float((gPQ - ref).abs().max()), flush=True)
