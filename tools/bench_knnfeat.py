import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N, C, K in ((32, 1024, 64, 20), (32, 1024, 128, 20), (8, 2048, 64, 20), (2, 300, 24, 7)):
    x = torch.randn(B, N, C, device=dev)
    for _ in range(2): i = ops.knn_feat(x, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.knn_feat(x, K)
    e1.record(); torch.cuda.synchronize()
    xd = x[0].double()
    D = ((xd[:, None] - xd[None]) ** 2).sum(-1)
    ri = D.topk(K, dim=-1, largest=False)[1]
    same = (i[0].long().sort(1)[0] == ri.sort(1)[0]).float().mean().item()
    print(json.dumps({"B": B, "N": N, "C": C, "K": K, "us": e0.elapsed_time(e1) * 100, "set_agreement": same}))
