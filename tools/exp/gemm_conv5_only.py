"""conv5's two GEMM shapes on the tile variants of pc3d_gemm_nt_tiled_f32 and on the library, a few launches each: the workload for
a counter pass (tools/prof_sq.sh tools/exp/gemm_conv5_only.py)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
L = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")
for M, N, K in ((32768, 1024, 512), (32768, 512, 1024)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for v in (5, 6, 8):
        for _ in range(6):
            L.call("pc3d_gemm_nt_tiled_f32", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), 0, 0, 0.0, M, N, K, 1, 0.0, out.data_ptr(), N, v,
                   torch.cuda.current_stream().cuda_stream)
    for _ in range(6):
        torch._addmm_activation(b, x, w.t(), use_gelu=False)
torch.cuda.synchronize()
