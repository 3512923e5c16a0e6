"""Time pc3d_curve_agg_kv_f32 / its backward at the classifier's block sizes (B=32, 100 curves of 5 points, C=16 / 32)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
for C in (16, 32):
    B, cn, cl, mid = 32, 100, 5, C // 2
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    curves = r(B, cn, cl, C).requires_grad_()
    ws = [r(C), r(mid, C), r(mid, C), r(mid, mid), r(mid, mid), r(mid, C), r(C, 2 * mid), r(C)]
    Kp, Vp = ops.curve_agg_kv(curves, *ws)
    gk, gv = torch.randn_like(Kp), torch.randn_like(Vp)
    def fwd(): return ops.curve_agg_kv(curves, *ws)
    def both():
        k, v = ops.curve_agg_kv(curves, *ws)
        torch.autograd.grad([k, v], [curves], [gk, gv])
    out = {}
    for name, fn in (("fwd", fwd), ("fwd+bwd", both)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"C={C}: fwd {out['fwd']:.1f} us, bwd {out['fwd+bwd'] - out['fwd']:.1f} us (incl. launch overheads)", flush=True)
