"""Deterministic scatter kernels (csrc/det.hip) at the victims' sizes: us per call, HIP events, beside the float-atomic
flavour. edge_max backward per channel-slice width (pc3d_edge_max_bwd_slice_f32)."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
L = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")


def us_of(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / it * 1e3, 1)


st = lambda: torch.cuda.current_stream().cuda_stream
for B, N, C in ((32, 1024, 64), (32, 1024, 128), (32, 1024, 256), (32, 4096, 32), (1, 1024, 256)):
    g = torch.randn(B, N, C, device=dev)
    out = torch.randn(B, N, C, device=dev)
    arg = torch.randint(0, N, (B, N, C), device=dev, dtype=torch.int32)
    gPQ = torch.empty(B, N, 2 * C, device=dev)
    row = {"op": "edge_max_bwd", "B": B, "N": N, "C": C}
    row["atomic_us"] = us_of(lambda: L.call("pc3d_edge_max_bwd_f32", g.data_ptr(), C, out.data_ptr(), arg.data_ptr(), B, N, C, 0.2,
                                            gPQ.data_ptr(), 0, st()))
    ref = None
    for sl in (0, 2, 4, 8, 16):
        try:
            row[f"slice{sl}_us"] = us_of(lambda: L.call("pc3d_edge_max_bwd_slice_f32", g.data_ptr(), C, out.data_ptr(), arg.data_ptr(),
                                                        B, N, C, 0.2, gPQ.data_ptr(), sl, st()))
            if ref is None:
                ref = gPQ.clone()
            row[f"slice{sl}_same"] = bool(torch.equal(ref, gPQ))
        except Exception as e:
            row[f"slice{sl}_us"] = str(e)[:60]
    print(json.dumps(row), flush=True)
