#!/usr/bin/env python3
"""GPU box: is the sampling kernel a pure function of its input while other work shares the chip? 200 launches of every
workgroup size at N = 1024 / 4096 beside a stream of GEMMs, each compared with the first."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = importlib.import_module("3dpointcloudattack_amd._lib")
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)
for N, S in ((1024, 256), (4096, 1024)):
    x = torch.rand(32, N, 3, device=dev)
    for thr in (64, 128, 256):
        if N > 32 * thr:
            continue
        ref = None
        bad = 0
        for it in range(200):
            with torch.cuda.stream(side):
                for _ in range(2):
                    a @ a
            out = torch.empty((32, S), dtype=torch.int32, device=dev)
            lib.call("pc3d_fps_threads_f32", thr, x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), 32, N, S, 0, out.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
            if ref is None:
                ref = out.clone()
            elif not torch.equal(out, ref):
                bad += 1
        torch.cuda.synchronize()
        print(f"N={N} threads={thr}: {bad} of 199 launches differ from the first", flush=True)
