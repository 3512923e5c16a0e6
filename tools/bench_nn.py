"""Micro-benchmark of the bidirectional NN search (Chamfer core) at BASELINE sizes: the shared-evaluation kernel pair
(scan + finalize) beside the two-scan kernel. HIP-event timing on torch's stream; valu_frac = 10*N^2 lane-ops per cloud
pair / time / 78.6 T lane-op/s."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
PEAK = 256 * 4 * 32 * 2.4e9
for B, N in ((32, 1024), (32, 2048), (32, 4096), (64, 2048), (256, 4096)):
    a = torch.randn(B, N, 3, device=dev); b = a + 0.01 * torch.randn_like(a)
    row = {"B": B, "N": N}
    for tag, kw in (("shared", {"two_scan": False}), ("shared_noidx", {"want_idx": False, "two_scan": False}), ("two_scan", {"two_scan": True})):
        for _ in range(3): ops.nn_bidir_raw(a, b, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 30
        e0.record()
        for _ in range(it): ops.nn_bidir_raw(a, b, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        row[tag + "_us"] = round(ms * 1e3, 2)
        row[tag + "_valu_frac"] = round(10.0 * B * N * N / (ms * 1e-3) / PEAK, 3)
    print(json.dumps(row), flush=True)
