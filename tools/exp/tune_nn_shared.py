import importlib, sys, os, json, ctypes
sys.path.insert(0, "/root/repo")
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
lib = importlib.import_module("3dpointcloudattack_amd._lib").load()
dev = torch.device("cuda:0")
def graph_us(fn, per=20, reps=10):
    s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(per): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * per) * 1e3
for B, N in ((32, 1024), (32, 2048), (32, 4096)):
    a = torch.randn(B, N, 3, device=dev); b = a + 0.01 * torch.randn_like(a)
    ref = ops.nn_bidir_raw(a, b, two_scan=True)
    for q, wgs, mb in ((0, 0, 0),):
        pass  # (the tuning hook this script drove has been removed: the measured rule is in nn_shared_plan)
        ops._NN_WS.clear()
        got = ops.nn_bidir_raw(a, b, two_scan=False)
        ok = all(torch.equal(x, y) for x, y in zip(got, ref))
        print(json.dumps({"B": B, "N": N, "Q": q, "wgs": wgs, "minbf": mb, "values": round(graph_us(lambda: ops.nn_bidir_raw(a, b, want_idx=False, two_scan=False)), 2),
                          "idx": round(graph_us(lambda: ops.nn_bidir_raw(a, b, two_scan=False)), 2), "ok": ok}), flush=True)
