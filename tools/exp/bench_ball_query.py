"""Time pc3d_ball_query_f32 at CurveNet's three pooling levels and PointNet++ SSG's two (kernel time from rocprofv3)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, B, N, S, r, ns in (("curvenet L1", 32, 4096, 1024, 0.05, 20), ("curvenet L2", 32, 1024, 256, 0.1, 20),
                             ("curvenet L3", 32, 256, 64, 0.2, 20), ("ssg SA1", 64, 2048, 512, 0.2, 32), ("ssg SA2", 64, 512, 128, 0.4, 64)):
    x = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=2).to(dev) * torch.rand(B, N, 1, generator=g).to(dev) ** (1 / 3)
    idx = ops.fps(x, S, None)
    c = torch.gather(x, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    t = {}
    for kernel in ("wave", "lane", None):
        for _ in range(3): ops.ball_query(r, ns, x, c, kernel=kernel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): out = ops.ball_query(r, ns, x, c, kernel=kernel)
        e1.record(); torch.cuda.synchronize()
        t[kernel] = e0.elapsed_time(e1) / 20 * 1e3
    filled = (out != out[:, :, :1]).sum(-1).float().mean().item() + 1
    print(f"{name}: wave {t['wave']:.1f} / lane {t['lane']:.1f} / chosen {t[None]:.1f} us per call (B={B} N={N} S={S} r={r} ns={ns}; "
          f"mean distinct hits {filled:.1f})", flush=True)
