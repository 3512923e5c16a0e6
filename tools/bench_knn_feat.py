"""pc3d_knn_feat_f32 at the DGCNN shapes: us per call."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N, C, K in ((32, 1024, 64, 20), (32, 1024, 128, 20), (32, 1024, 64, 1), (32, 2048, 64, 20), (8, 1024, 64, 20)):
    torch.manual_seed(C + N)
    x = torch.randn(B, N, C, device=dev)
    for _ in range(3): ops.knn_feat(x, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.knn_feat(x, K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(json.dumps({"B": B, "N": N, "C": C, "K": K, "us": round(us, 1), "TFLOPs": round(2.0 * B * N * N * C / us / 1e6, 1)}), flush=True)
