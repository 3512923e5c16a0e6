"""conv5's forward GEMM timed as graphs of different length: does the launch time depend on how long the matrix pipe has been busy?
(bench.py's sweep reads 351-354 us for it as 5 calls x 8 replays back to back; the GeoA3 loop profile and tools/bench_gemm.py read 321-322.)"""
import importlib, os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = sys.argv[:1]
import torch, bench
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
M, N, K = 32768, 1024, 512
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
xz = torch.relu(torch.randn(M, K, device=dev))            # half zeros, like a post-activation operand
for name, xx in (("gaussian operand", x), ("half-zero operand", xz)):
    for per, reps in ((1, 1), (1, 4), (5, 2), (5, 8), (5, 40)):
        torch.cuda.synchronize(); time.sleep(0.3)          # let the device idle before every measurement
        us = bench.graph_ms(lambda: ops.gemm_nt(xx, w, None, "leaky", 0.2), per=per, reps=reps) * 1e3
        print(json.dumps({"operand": name, "calls_per_graph": per, "replays": reps, "busy_ms": round(per * reps * us / 1e3, 1), "us_per_call": round(us, 1)}), flush=True)
