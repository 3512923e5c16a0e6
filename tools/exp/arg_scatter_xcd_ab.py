"""The deterministic EdgeConv backward (pc3d_edge_max_bwd_slice_f32) with its workgroups in XCD bands and in launch order, a few
launches each: the workload for a TCC counter pass — rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum (or FETCH_SIZE) -- python3 ..."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
L = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
for B, N, C in ((32, 1024, 64), (32, 1024, 256)):
    g = torch.randn(B, N, C, device=dev); out = torch.randn(B, N, C, device=dev)
    arg = torch.randint(0, N, (B, N, C), device=dev, dtype=torch.int32); gPQ = torch.empty(B, N, 2 * C, device=dev)
    for sl in (8, 8 + 512):          # banded, launch order
        for _ in range(6):
            L.call("pc3d_edge_max_bwd_slice_f32", g.data_ptr(), C, out.data_ptr(), arg.data_ptr(), B, N, C, 0.2, gPQ.data_ptr(), sl, st())
        torch.cuda.synchronize()
