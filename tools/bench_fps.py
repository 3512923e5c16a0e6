"""pc3d_fps_f32 at the shapes of the attack configs: us per call and per sampling step."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
import numpy as np
from tests.helpers import unit_cloud
for B, N, S in ((32, 4096, 1024), (64, 2048, 512), (64, 512, 128), (32, 1024, 256), (32, 256, 64)):
    rng = np.random.default_rng(N)
    x = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).to(dev)
    for _ in range(3): ops.fps(x, S, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.fps(x, S, None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"B={B} N={N} S={S}: {us:.1f} us, {us / S * 1e3:.0f} ns per step", flush=True)
    lib = importlib.import_module("3dpointcloudattack_amd._lib")
    out = torch.empty((B, S), dtype=torch.int32, device=dev)
    ref = ops.fps(x, S, None)
    call = lambda: lib.call("pc3d_fps_pruned_f32", x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, N, S, 0,
                            out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if N <= 4096:
        for _ in range(3): call()
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"    pruned (pc3d_fps_pruned_f32): {us:.1f} us, {us / S * 1e3:.0f} ns per step, equal {bool(torch.equal(out, ref))}", flush=True)
        call1 = lambda: lib.call("pc3d_fps_pruned_f32", x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, N, 1, 0,
                                 out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for _ in range(3): call1()
        e0.record()
        for _ in range(10): call1()
        e1.record(); torch.cuda.synchronize()
        print(f"    pruned, S=1 (the setup): {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", flush=True)
    for thr in (64, 128, 256, 512):
        if N > thr * 32:
            continue
        call = lambda: lib.call("pc3d_fps_threads_f32", thr, x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, N, S, 0,
                                out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for _ in range(3): call()
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"    {thr} threads: {us:.1f} us, {us / S * 1e3:.0f} ns per step, equal {bool(torch.equal(out, ref))}", flush=True)
