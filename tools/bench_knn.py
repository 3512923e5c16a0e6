"""pc3d_knn_f32 (xyz kNN, K-lists across the lanes) at the shapes of the attacks: us per call."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for B, N, K in ((32, 4096, 20), (32, 1024, 21), (32, 1024, 20), (64, 2048, 5), (32, 4096, 21), (32, 1024, 17)):
    torch.manual_seed(N)
    x = torch.rand(B, N, 3, device=dev)
    for _ in range(3): ops.knn_raw(x, x, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.knn_raw(x, x, K)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"B": B, "N": N, "K": K, "us": round(e0.elapsed_time(e1) / 20 * 1e3, 1)}), flush=True)

# ---- CurveNet's graphs: the plain search (values + indices) against pc3d_knn_graph_i32 (indices + the two views its blocks
# gather through, written by the same launch) — both as 10 calls per replayed hipGraph (round-3 review: knn_wave_kernel<4,true>
# went 124 -> 160 us average inside the cfg5 loop when the views moved into the launch; stand-alone A/B)
def graph_us(fn, reps=10, rounds=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


for B, N in ((32, 4096), (32, 1024), (32, 256), (32, 64)):
    torch.manual_seed(N)
    x = torch.rand(B, N, 3, device=dev)
    t_plain = graph_us(lambda: ops.knn_raw(x, x, 21))
    t_graph = graph_us(lambda: ops.knn_graph(x, 20))
    print(json.dumps({"B": B, "N": N, "k": 20, "knn_raw_us": round(t_plain, 1), "knn_graph_us": round(t_graph, 1)}), flush=True)

# ---- hints (pc3d_knn_hint_f32): the same search unhinted, with last call's result as the hint (what a replayed graph does:
# every captured call reads its own previous output) and with the hint of points that have since moved by 1e-2 per coordinate
ops.KNN_HINTS = False
for B, N, K in ((32, 4096, 21), (32, 1024, 21), (64, 2048, 6), (32, 1024, 20), (32, 256, 21)):
    torch.manual_seed(N)
    x = torch.rand(B, N, 3, device=dev)
    row = {"B": B, "N": N, "K": K}
    ops.KNN_HINTS = False
    row["unhinted_us"] = round(graph_us(lambda: ops.knn_raw(x, x, K)), 1)
    ops.KNN_HINTS = True
    row["hint_same_points_us"] = round(graph_us(lambda: ops.knn_raw(x, x, K)), 1)
    L = importlib.import_module("3dpointcloudattack_amd._lib")
    _, stale = ops.knn_raw(x + 1e-2 * torch.randn_like(x), x + 1e-2 * torch.randn_like(x), K)
    d = torch.empty(B, N, K, device=dev)
    i = torch.empty(B, N, K, dtype=torch.int32, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    row["hint_moved_1e-2_us"] = round(graph_us(lambda: L.call("pc3d_knn_hint_f32", x.data_ptr(), *x.stride(), x.data_ptr(), *x.stride(), B, N, N, K,
                                                              d.data_ptr(), i.data_ptr(), stale.data_ptr(), st())), 1)
    print(json.dumps(row), flush=True)
