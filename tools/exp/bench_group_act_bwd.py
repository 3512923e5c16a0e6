"""EXPERIMENT: pc3d_group_act_bwd_f32 (sign from H) against pc3d_group_act_bwd_mask_f32 (sign from the bit mask) on the
SSG set-abstraction shapes, same inputs; us per call."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
lib = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
for B, NA, S, K, C in ((64, 2048, 512, 32, 64), (64, 512, 128, 64, 128)):
    g = torch.Generator().manual_seed(0)
    P = torch.randn(B, NA, C, device=dev); Bc = torch.randn(B, S, C, device=dev)
    idx = torch.randint(0, NA, (B, S, K), generator=g)
    idx[:, :, K // 2:] = idx[:, :, :1]
    idx = idx.int().to(dev)
    H = ops.group_act(P, Bc, idx, 0.0)
    gH = torch.randn_like(H)
    bits = (H > 0).view(B * S * K, C // 4, 4).to(torch.uint8)
    mask = (bits[..., 0] | (bits[..., 1] << 1) | (bits[..., 2] << 2) | (bits[..., 3] << 3)).contiguous()
    gP = torch.empty(B, NA, C, device=dev); gBc = torch.empty(B, S, C, device=dev)
    gP2 = torch.empty_like(gP); gBc2 = torch.empty_like(gBc)
    def a():
        lib.call("pc3d_group_act_bwd_f32", gH.data_ptr(), H.data_ptr(), idx.data_ptr(), B, NA, S, K, C, 0.0, gP.data_ptr(), gBc.data_ptr(), st())
    def b():
        lib.call("pc3d_group_act_bwd_mask_f32", gH.data_ptr(), mask.data_ptr(), idx.data_ptr(), B, NA, S, K, C, 0.0, gP2.data_ptr(), gBc2.data_ptr(), st())
    a(); b(); torch.cuda.synchronize()
    print("max |gBc diff|", float((gBc - gBc2).abs().max()), "max |gP diff|", float((gP - gP2).abs().max()))
    for name, f in (("H", a), ("mask", b), ("H", a), ("mask", b)):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(B, NA, S, K, C, name, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us", flush=True)
