#!/usr/bin/env python3
"""GPU box: per-phase clocks of the full-update sampling kernel (diagnostic build of csrc/group.hip with -DFPS_DIAG, compiled
here into /tmp; the product library carries no stamps). Per wave and step: [0] pick bookkeeping + centre lookup + row updates,
[1] wave arg-max, [2] key write + barrier, [3] key reads + selection."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import unit_cloud
src = os.path.join(ROOT, "3dpointcloudattack_amd", "csrc")
so = "/tmp/libfps_full_diag.so"
import glob
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-DFPS_DIAG", "-c",
                os.path.join(src, "group.hip"), "-o", "/tmp/group_diag.o"], check=True)
objs = [o for o in glob.glob(os.path.join(src, "*.o")) if os.path.basename(o) != "group.o"]     # the product's other objects
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/group_diag.o"] + objs, check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
for B, N, S in ((32, 4096, 1024), (64, 2048, 512), (32, 256, 64)):
    rng = np.random.default_rng(N)
    x = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).to(dev)
    out = torch.empty((B, S), dtype=torch.int32, device=dev)
    for thr in (64, 128, 256, 512):
        if N > 32 * thr:
            continue
        W = thr // 64
        diag = torch.zeros((B, W, 4), dtype=torch.int64, device=dev)
        lib.fps_set_diag(ctypes.c_void_p(diag.data_ptr()))
        for _ in range(2):
            rc = lib.pc3d_fps_threads_f32(thr, ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(x.stride(0)), ctypes.c_int64(x.stride(1)),
                                          ctypes.c_int64(x.stride(2)), B, N, S, None, ctypes.c_void_p(out.data_ptr()), None)
            assert rc == 0
        torch.cuda.synchronize()
        d = diag.float().mean(dim=0).cpu().numpy() / S
        print(f"N={N} threads={thr}: per step, mean over waves " + " ".join(f"{v:7.1f}" for v in d.mean(0)) + f"  total {d.mean(0).sum():7.1f}"
              + "   | slowest wave in [2]: %.1f, fastest %.1f" % (d[:, 2].max(), d[:, 2].min()))
