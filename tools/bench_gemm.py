"""pc3d_gemm_nt_f32 beside torch (hipBLASLt) on the victims' layer shapes: TFLOP/s, HIP events on torch's stream."""
import importlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")


def ms_of(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def tiled(x, w, b, variant, out=[None]):
    """pc3d_gemm_nt_tiled_f32 with an explicit tile variant (ReLU epilogue)."""
    M, K = x.shape
    N = w.shape[0]
    if out[0] is None or out[0].shape != (M, N):
        out[0] = torch.empty((M, N), device=x.device)
    L.call("pc3d_gemm_nt_tiled_f32", x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr(), 0, 0, 0.0, M, N, K, 1, 0.0,
           out[0].data_ptr(), N, int(variant), torch.cuda.current_stream().cuda_stream)
    return out[0]


L = importlib.import_module("3dpointcloudattack_amd._lib")
shapes = [("dgcnn conv5", 32768, 1024, 512), ("dgcnn edge4", 32768, 512, 128), ("dgcnn edge2", 32768, 128, 64),
          ("ssg sa1 l2", 1048576, 64, 64), ("ssg sa1 l1", 1048576, 64, 3), ("ssg sa2 l1", 524288, 128, 131),
          ("ssg sa2 l2", 524288, 128, 128), ("ssg sa3 l2", 8192, 512, 256), ("curvenet pw", 131072, 64, 64)]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    tv = {}
    for v in (0, 2, 4, 5, 6, 8):
        if v in (2, 3) and N > 64 * 64:
            continue
        tv[v] = round(ms_of(lambda: tiled(x, w, b, v)) * 1e3, 1)
    t_own = ms_of(lambda: ops.gemm_nt(x, w, b, "relu"))
    t_lib = ms_of(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
    fl = 2.0 * M * N * K
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "own_us": round(t_own * 1e3, 1), "lib_us": round(t_lib * 1e3, 1),
                      "own_TFLOPs": round(fl / t_own / 1e9, 1), "lib_TFLOPs": round(fl / t_lib / 1e9, 1),
                      "own_GBps": round((M * K + M * N) * 4 / t_own / 1e6, 0), "variants_us": tv}), flush=True)
