import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N, K in ((32, 4096, 21), (32, 1024, 21), (32, 1024, 17), (64, 2048, 6), (32, 1024, 30), (32, 256, 21), (32, 1024, 4)):
    x = torch.randn(B, N, 3, device=dev)
    for _ in range(2): d, i = ops.knn_raw(x, x, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.knn_raw(x, x, K)
    e1.record(); torch.cuda.synchronize()
    # exact check vs float64 brute force on one batch element
    D = ((x[0].double()[:, None] - x[0].double()[None]) ** 2).sum(-1)
    rd, ri = D.topk(K, dim=-1, largest=False)
    ok = bool(torch.equal(i[0].long(), ri)) or float((d[0].double() - rd).abs().max()) < 1e-6
    print(json.dumps({"B": B, "N": N, "K": K, "us": e0.elapsed_time(e1) * 100, "idx_equal": bool(torch.equal(i[0].long(), ri)), "ok": ok}))
