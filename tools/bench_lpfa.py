"""pc3d_lpfa_fused_f32 forward at the CIC shapes of cfg5 (B=32, k=20): us per call (HIP events around eager calls; the
forward+backward figure is dominated by the host's autograd bookkeeping at these sizes and is printed for reference)."""
import importlib, sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
lib = importlib.import_module("3dpointcloudattack_amd._lib").load()
dev = torch.device("cuda:0")
for B, N, C in ((32, 1024, 16), (32, 1024, 32), (32, 256, 64), (32, 64, 128)):
    K = 20
    A = torch.randn(B, N, C, device=dev, requires_grad=True); Bc = torch.randn(B, N, C, device=dev, requires_grad=True)
    idx = torch.randint(0, N, (B, N, K), device=dev, dtype=torch.int32)
    W = torch.randn(C, C, device=dev) / C ** 0.5; b = torch.randn(C, device=dev); up = torch.randn(B, N, C, device=dev)
    row = {"B": B, "N": N, "C": C}
    for T in (256,):
        def fwd(): return ops.lpfa_fused(A, Bc, idx, W, b, 0.2, 0.2)
        def both():
            A.grad = Bc.grad = None
            fwd().backward(up)
        for name, fn in (("fwd", fwd), ("fwd+bwd", both)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            row[f"{name}_T{T}"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps(row), flush=True)
