import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
def w(C3):
    return tuple(t.to(dev) for t in (torch.randn(64,3), torch.randn(64), torch.randn(128,64)/8, torch.randn(128), torch.randn(C3,128)/11, torch.randn(C3)))
for B, N in ((32, 1024), (32, 2048), (64, 2048)):
    x = torch.randn(B, 3, N, device=dev); ws = w(1024)
    for _ in range(3): p, i, mk = ops.pointmlp3_max_fwd_raw(x, ws, False, want_masks=True)
    g = torch.randn_like(p)
    for _ in range(3): ops.pointmlp3_max_bwd_raw(x, ws, i, g, mk)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    it = 20
    e[0].record()
    for _ in range(it): ops.pointmlp3_max_fwd_raw(x, ws, False)
    e[1].record()
    for _ in range(it): ops.pointmlp3_max_bwd_raw(x, ws, i, g, mk)
    e[2].record(); torch.cuda.synchronize()
    f = e[0].elapsed_time(e[1]) / it; bw = e[1].elapsed_time(e[2]) / it
    flop = 2.0 * B * N * (3*64 + 64*128 + 128*1024)
    print(json.dumps({"B": B, "N": N, "fwd_us": f*1e3, "fwd_TFLOPs": flop / f / 1e9, "bwd_us": bw*1e3}))
