#!/usr/bin/env python3
"""GPU box: AOF's spectral re-projection (pc3d_spectral_reproject_f32) against the three torch.bmm it replaces, both timed
as 10 calls per replayed hipGraph; algorithmic bytes = V and V^T read once each (2 B N^2 4)."""
import importlib
import json
import sys

import torch

sys.path.insert(0, ".")
ops = importlib.import_module("3dpointcloudattack_amd.ops")


def graph_us(fn, reps=10, rounds=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


def main():
    dev = torch.device("cuda:0")
    out = []
    for B, N, lp in ((32, 1024, 100), (32, 512, 100), (8, 1024, 100), (8, 2048, 100)):
        V = torch.linalg.qr(torch.randn(B, N, N, device=dev))[0].contiguous()
        Vt = V.transpose(1, 2).contiguous()
        adv = torch.randn(B, 3, N, device=dev)
        lfc, hfc, co = (torch.empty_like(adv) for _ in range(3))
        Vlo_t, Vhi_t = Vt[:, :lp].contiguous(), Vt[:, lp:].contiguous()

        def own():
            ops.spectral_reproject(adv, V, Vt, lp, lfc, hfc, co)

        def lib():
            c = torch.bmm(adv, V)
            hfc.copy_(torch.bmm(c[..., lp:], Vhi_t))
            lfc.copy_(torch.bmm(c[..., :lp], Vlo_t))

        t_own, t_lib = graph_us(own), graph_us(lib)
        nbytes = 2 * B * N * N * 4
        out.append({"B": B, "N": N, "lp": lp, "own_us": round(t_own, 2), "bmm_us": round(t_lib, 2),
                    "own_GBps": round(nbytes / t_own / 1e3, 1), "hbm_frac": round(nbytes / t_own / 1e3 / 8000, 3)})
        print(out[-1], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
