#!/usr/bin/env python3
"""GPU box: per-phase clocks of the pruned FPS chain (diagnostic build of csrc/fps_pruned.hip with -DFPSP_DIAG, compiled
here with hipcc into /tmp; the product library carries no stamps). Phases per wave, summed over the S steps:
0 centre lookup + bound, 1 row updates, 2 the wave's best row, 3 post + poll, 4 winner selection; [5] active rows, [6] polls."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import unit_cloud
src = os.path.join(ROOT, "3dpointcloudattack_amd", "csrc")
so = "/tmp/libfps_diag.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                "-DFPSP_DIAG", os.path.join(src, "fps_pruned.hip"), os.path.join(src, "api_common.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
for B, N, S in ((32, 4096, 1024), (64, 2048, 512), (32, 1024, 256), (32, 256, 64)):
    rng = np.random.default_rng(N)
    x = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).to(dev)
    out = torch.empty((B, S), dtype=torch.int32, device=dev)
    diag = torch.zeros((B, 4, 8), dtype=torch.int64, device=dev)
    lib.fpsp_set_diag(ctypes.c_void_p(diag.data_ptr()))
    for _ in range(3):
        rc = lib.pc3d_fps_pruned_f32(ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(x.stride(0)), ctypes.c_int64(x.stride(1)),
                                     ctypes.c_int64(x.stride(2)), B, N, S, None, ctypes.c_void_p(out.data_ptr()), None)
        assert rc == 0
    torch.cuda.synchronize()
    d = diag.float().mean(dim=(0,)).cpu().numpy() / S           # per wave, per step
    print(f"N={N} S={S}: per step and wave (memtime ticks = 100 MHz? see total):")
    for w in range(4):
        print("   wave", w, " ".join(f"{v:8.1f}" for v in d[w]))
