"""pc3d_kappa_f32 / pc3d_kappa_bwd_f32 (GeoA3's curvature proxy) through the C ABI: us per call (HIP events around 50
direct calls; the autograd wrappers add host time that hides kernels this small)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
lib = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
for B, N, k in ((32, 1024, 16), (32, 1024, 2), (32, 4096, 16), (8, 8192, 16)):
    x = (torch.rand(B, 3, N, device=dev) - 0.5)
    nrm = torch.nn.functional.normalize(torch.randn(B, 3, N, device=dev), dim=1)
    idx = ops.knn_raw(x, x, k + 1, q_cf=True, r_cf=True)[1]
    out = torch.empty(B, N, device=dev); g = torch.randn(B, N, device=dev); gx = torch.empty(B, N, 3, device=dev)
    xs, ns = (x.stride(0), x.stride(2), x.stride(1)), (nrm.stride(0), nrm.stride(2), nrm.stride(1))
    def fwd(): lib.call("pc3d_kappa_f32", x.data_ptr(), *xs, nrm.data_ptr(), *ns, idx.data_ptr(), B, N, k + 1, out.data_ptr(), st())
    def bwd(): lib.call("pc3d_kappa_bwd_f32", x.data_ptr(), *xs, nrm.data_ptr(), *ns, idx.data_ptr(), g.data_ptr(), B, N, k + 1, gx.data_ptr(), st())
    res = []
    for f in (fwd, bwd):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"B={B} N={N} k={k}: fwd {res[0]:.1f} us, bwd {res[1]:.1f} us", flush=True)
