"""pc3d_knn_f32 (xyz kNN, K-lists across the lanes) at the shapes of the attacks: us per call."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N, K in ((32, 4096, 20), (32, 1024, 21), (32, 1024, 20), (64, 2048, 5), (32, 4096, 21), (32, 1024, 17)):
    torch.manual_seed(N)
    x = torch.rand(B, N, 3, device=dev)
    for _ in range(3): ops.knn_raw(x, x, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.knn_raw(x, x, K)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"B": B, "N": N, "K": K, "us": round(e0.elapsed_time(e1) / 20 * 1e3, 1)}), flush=True)
