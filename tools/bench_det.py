"""Deterministic scatter kernels (csrc/det.hip) at the victims' sizes: us per call, HIP events, beside the float-atomic
flavour. edge_max backward per channel-slice width (pc3d_edge_max_bwd_slice_f32)."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
L = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")


def us_of(fn, it=20):
    """us per call as 10 calls inside ONE replayed hipGraph (bench.graph_ms): eager calls through ctypes cost ~10-20 us of host
    time each, more than several of these kernels run."""
    import bench
    return round(bench.graph_ms(fn, per=10, reps=10) * 1e3, 1)


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
    for sl in (0, 4, 8, 16, 0 + 256, 4 + 256, 8 + 256, 16 + 256):      # + 256: fp32 tiles
        try:
            row[f"slice{sl & 255}_w{sl >> 8}_us"] = us_of(lambda: L.call("pc3d_edge_max_bwd_slice_f32", g.data_ptr(), C, out.data_ptr(), arg.data_ptr(),
                                                        B, N, C, 0.2, gPQ.data_ptr(), sl, st()))
            if (sl >> 8) == 0:      # same bits whatever the width
                if ref is None:
                    ref = gPQ.clone()
                row[f"slice{sl & 255}_w{sl >> 8}_same"] = bool(torch.equal(ref, gPQ))
        except Exception as e:
            row[f"slice{sl & 255}_w{sl >> 8}_us"] = str(e)[:60]
    print(json.dumps(row), flush=True)
