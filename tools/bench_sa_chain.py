"""pc3d_sa_chain_f32 (gather -> layers 1-2-3 -> group max in one launch) beside the two-launch form at SSG's sizes: us per
forward of ops.grouped_mlp_max, HIP events."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for name, B, NA, S, ns, C1, C2, C3 in (("ssg sa1", 64, 2048, 512, 32, 64, 64, 128), ("ssg sa2", 64, 512, 128, 64, 128, 128, 256),
                                      ("msg sa1 s3", 32, 1024, 512, 128, 64, 96, 128)):
    P = torch.randn(B, NA, C1, device=dev)
    Bc = torch.randn(B, S, C1, device=dev)
    idx = torch.randint(0, NA, (B, S, ns), device=dev, dtype=torch.int32)
    idx = torch.sort(idx, dim=2)[0].contiguous()
    layers = [(torch.randn(C2, C1, device=dev) / C1 ** 0.5, torch.randn(C2, device=dev)),
              (torch.randn(C3, C2, device=dev) / C2 ** 0.5, torch.randn(C3, device=dev))]
    flops = 2.0 * B * S * ns * (C1 * C2 + C2 * C3)
    row = {"layer": name, "GFLOP": round(flops / 1e9, 1)}
    for tag, chain in (("chain", True), ("two_launch", False)):
        ops.SA_CHAIN = chain
        with torch.no_grad():
            for _ in range(3):
                ops.grouped_mlp_max(P, Bc, idx, layers)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.grouped_mlp_max(P, Bc, idx, layers)
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row[tag + "_us"] = round(us, 1)
        row[tag + "_frac_of_mfma_peak"] = round(flops / us / 1e6 / 157.3, 3)
    ops.SA_CHAIN = True
    print(json.dumps(row), flush=True)
