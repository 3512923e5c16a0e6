"""pc3d_group_linear_max_f32 (last SA layer + ReLU + group max) on the SSG / MSG shapes: us and TFLOP/s of the
one-workgroup-per-group kernel and of the GEMM main loop with the group-max epilogue (the default for ns = 32 / 64 / 128)."""
import importlib, sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
L = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")
for name, G, ns, C2, C3 in (("ssg sa1", 64 * 512, 32, 64, 128), ("ssg sa2", 64 * 128, 64, 128, 256),
                            ("msg sa1 s3", 32 * 512, 128, 96, 128), ("msg sa1 s1", 32 * 512, 16, 32, 64)):
    x = torch.relu(torch.randn(G, ns, C2, device=dev))
    w = torch.randn(C3, C2, device=dev) / C2 ** 0.5
    b = torch.randn(C3, device=dev)
    row = {"layer": name, "G": G, "ns": ns, "C2": C2, "C3": C3}
    ref = torch.relu(x @ w.t() + b).max(dim=1)[0]
    out = torch.empty((G, C3), device=dev)
    arg = torch.empty((G, C3), dtype=torch.int64, device=dev)

    def run(kernel):
        L.call("pc3d_group_linear_max_kernel_f32", kernel, x.data_ptr(), w.data_ptr(), b.data_ptr(), G, ns, C2, C3,
               out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out

    for tag, kernel in (("per_group_kernel", 1), ("gemm_epilogue", 2)):
        if kernel == 2 and ns not in (32, 64, 128):
            continue
        for _ in range(3): run(kernel)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(kernel)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        row[tag + "_us"] = round(ms * 1e3, 1)
        row[tag + "_TFLOPs"] = round(2.0 * G * ns * C2 * C3 / ms / 1e9, 1)
        row[tag + "_ok"] = bool(torch.allclose(run(kernel), ref, rtol=1e-4, atol=1e-4))
    print(json.dumps(row), flush=True)
