import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N, S in ((64, 2048, 512), (64, 512, 128), (32, 4096, 1024), (32, 1024, 512)):
    x = torch.randn(B, N, 3, device=dev)
    for _ in range(2): i = ops.fps(x, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.fps(x, S)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"B": B, "N": N, "S": S, "us": e0.elapsed_time(e1) * 200, "us_per_step": e0.elapsed_time(e1) * 200 / S}))
