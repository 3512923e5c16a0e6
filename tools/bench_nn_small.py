"""Small-problem NN search timed INSIDE a replayed hipGraph (20 calls per graph): at N <= 1024 an eager Python call
costs more host time than the kernel runs, so event timing around eager calls measures the host."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
PEAK = 256 * 4 * 32 * 2.4e9

def graph_us(fn, per=20, reps=10):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(per): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * per) * 1e3

for B, N in ((32, 1024), (32, 512), (8, 1024), (32, 2048), (32, 4096)):
    a = torch.randn(B, N, 3, device=dev); b = a + 0.01 * torch.randn_like(a)
    row = {"B": B, "N": N}
    row["two_scan"] = round(graph_us(lambda: ops.nn_bidir_raw(a, b, two_scan=True)), 2)
    row["one_dir"] = round(graph_us(lambda: ops.nn_raw(a, b)), 2)
    row["shared_values"] = round(graph_us(lambda: ops.nn_bidir_raw(a, b, want_idx=False, two_scan=False)), 2)
    row["shared_idx"] = round(graph_us(lambda: ops.nn_bidir_raw(a, b, two_scan=False)), 2)
    best = min(row["two_scan"], row["shared_values"])
    row["best_valu_frac"] = round(10.0 * B * N * N / (best * 1e-6) / PEAK, 3)
    print(json.dumps(row), flush=True)
