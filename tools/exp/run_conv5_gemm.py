"""Runs the DGCNN conv5-shaped point-wise GEMM (M=32768, N=1024, K=512) 30 times: a target for tools/prof_sq.sh."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
x = torch.randn(32768, 512, device=dev); w = torch.randn(1024, 512, device=dev) / 22; b = torch.randn(1024, device=dev)
for _ in range(30): ops.gemm_nt(x, w, b, "leaky", 0.2)
torch.cuda.synchronize()
