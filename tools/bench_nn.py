"""Micro-benchmark of the fused NN kernel (Chamfer core) at BASELINE sizes. HIP-event timing on torch's stream."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N in ((32, 1024), (32, 2048), (32, 4096), (64, 2048), (256, 4096)):
    a = torch.randn(B, N, 3, device=dev); b = a + 0.01 * torch.randn_like(a)
    for _ in range(3): ops.nn_bidir_raw(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it): ops.nn_bidir_raw(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    pairs = 2.0 * B * N * N
    print(json.dumps({"B": B, "N": N, "us": ms * 1e3, "Gpairs_per_s": pairs / ms / 1e6,
                      "alg_GBps": B * 2 * N * 20 / ms / 1e6}))
