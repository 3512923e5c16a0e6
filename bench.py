#!/usr/bin/env python3
"""Headline benchmark: attack iterations per second of the CW attack on PointNet (B=32, N=1024, fp32) with the
Chamfer regulariser, on N GPUs of one node (BASELINE.json configs[1]; one rank per GPU, independent attack
instances per rank, one RCCL broadcast of the frozen weights, no per-iteration collective).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot-loop body for the whole batch: victim forward, adversarial + Chamfer loss,
backward to the input points, Adam step, per-point clip, best-attack bookkeeping — inputs already resident in HBM.
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel, measured live
with HIP events) and `cpu_baseline` (the oracle's torch-CPU restatement of the same loop, timed on the host cores).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, NPTS, NCLS = 32, 1024, 40
KAPPA, BUDGET, LR = 30.0, 0.18, 1e-2
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = vector peak
VALU_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9   # lanes issued per second (v_fma_f32 wave64 = 2 cycles / SIMD)


def unit_cloud(rng, n):
    g = rng.standard_normal((n, 3))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    p = g * rng.random((n, 1)) ** (1.0 / 3.0)
    p = p - p.mean(axis=0, keepdims=True)
    return (p / np.max(np.linalg.norm(p, axis=1))).astype(np.float32)


def seeded_state(model, seed):
    return importlib.import_module("3dpointcloudattack_amd.seeding").seeded_state_dict(model, seed)


def ev_ms(fn, iters, stream):
    """Average ms per call of fn() over `iters` back-to-back calls, HIP events on `stream`. A runner that batches
    calls (CW's graph runner) is flushed inside the timed region, so exactly `iters` calls are measured."""
    flush = getattr(fn, "flush", None)
    if flush is not None:
        flush()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    if flush is not None:
        flush()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def geo_cfg(**over):
    """GeoA3's default flags (attack/GeoA3/Eval_GeoA3.py:109-204: CE classification loss, CD + 0.1 HD + curvature
    consistency with 16 neighbours, Adam lr 1e-2) as the namespace geoA3_attack reads (GeoA3_attack.py:186-190)."""
    import types
    base = dict(attack_method='untarget', curv_loss_weight=1.0, curv_loss_knn=16, initial_const=10, iter_max_steps=12,
                binary_max_steps=1, is_partial_var=False, optim='adam', lr=0.01, npoint=1024, is_subsample_opt=False,
                eval_num=1, is_pre_jitter_input=False, cls_loss_type='CE', classes=NCLS, confidence=0, dis_loss_type='CD',
                is_cd_single_side=False, dis_loss_weight=1.0, hd_loss_weight=0.1, uniform_loss_weight=0.0,
                is_use_lr_scheduler=False, is_debug=False, is_pro_grad=False, cc_linf=0.0, binary_step=1, num_iter=12,
                is_real_offset=False, knn_range=3, calculate_project_jitter_noise_iter=50, jitter_k=16,
                jitter_sigma=0.01, jitter_clip=0.05)
    base.update(over)
    return types.SimpleNamespace(**base)


def _graphed():
    import importlib
    return importlib.import_module("3dpointcloudattack_amd.graphed")


CLOCK_WARM_MS = 150.0     # device time of sustained load before the sweep's per-kernel measurements (graph_ms, warm_ms)


def graph_ms(fn, per=10, reps=20, warm_ms=0.0):
    """ms per call of fn, timed as `per` calls captured into ONE hipGraph and replayed `reps` times (HIP events
    on the replay stream). At N <= 2048 an eager Python call costs more host time than these kernels run, so
    event timing around eager calls measures the host; the attack loops replay graphs as well."""
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with _graphed().capture_guard(), torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per):
                fn()
        g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if warm_ms > 0:      # sustained load first: MI355X raises its clocks over tens of ms of load (conv5's forward GEMM: 370 us in
                             # the first 4 ms after an idle gap, 357 after 14 ms, 324 after 65 ms — tools/exp/gemm_sustain.py), and the
                             # attack loops these kernels sit in run for seconds (GeoA3 loop profile: 322 us)
            e0.record(side)
            g.replay()
            e1.record(side)
            e1.synchronize()
            for _ in range(int(warm_ms / max(e0.elapsed_time(e1), 1e-3)) + 1):
                g.replay()
        e0.record(side)
        for _ in range(reps):
            g.replay()
        e1.record(side)
        e1.synchronize()
    return e0.elapsed_time(e1) / (per * reps)


def slope_ms(run_attack, base, inc):
    """ms per iteration of a whole attack call as the MEDIAN of three slopes: wall time of runs with base, base + inc,
    base + 2 inc, base + 3 inc iterations (after one untimed run of `base` that pays the captures); everything a call
    does besides its loop (upload, clean forward, final checks, D2H) cancels in the differences, and a one-off stall
    moves only one of the three."""
    run_attack(base)
    ts = []
    for it in (base, base + inc, base + 2 * inc, base + 3 * inc):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_attack(it)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    slopes = [(ts[i + 1] - ts[i]) / inc * 1e3 for i in range(3)]
    return sorted(slopes)[1], slopes


def pmc_provenance(counters):
    """Where and when the committed PMC file was collected (its "_meta" entry: the commit of the kernels it describes and
    the date, written when the file is copied from gpurun_out/ into profiles/): printed next to `traffic` so that a figure
    older than the kernels it describes is visible. The counters are collected under the rocprofv3 wrapper, not here."""
    meta = counters.get("_meta") if isinstance(counters, dict) else None
    return meta if isinstance(meta, dict) else None


def pmc_commit_distance(counters):
    """Commits between the commit the PMC file describes (its "_meta") and HEAD, when this copy of the repo has its history
    (the GPU box gets a snapshot without .git: None there — the judge's clone can run the same command)."""
    meta = pmc_provenance(counters)
    c = (meta or {}).get("commit")
    if not c or not os.path.isdir(os.path.join(ROOT, ".git")):
        return None
    try:
        import subprocess
        r = subprocess.run(["git", "-C", ROOT, "rev-list", "--count", f"{c}..HEAD"], capture_output=True, text=True, timeout=10)
        return int(r.stdout.strip()) if r.returncode == 0 else None
    except Exception:
        return None


def chamfer_cpu_baseline():
    """SURVEY §8(d) "CPU baseline beside it", metric 2: the reference's two CPU Chamfer paths per cloud pair, via the
    oracle's restatements — float64 direct-difference matrix + two min-reductions (utils/dis_utils_numpy.py:13-26) and
    torch-CPU cdist + min (utils/dis_utils_torch.py:8-16) — median of 3 after two warm-ups, all host cores."""
    from oracle import ref_numpy as orn
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, ncpu)))
    res = {"numpy_f64_ms_per_pair": {}, "torch_cpu_ms_per_pair": {}, "c_scalar_ms_per_pair": {},
           "cores": torch.get_num_threads(), "c_scalar_cores": 1, "kind": "port"}
    clib = None
    cso = os.path.join(ROOT, "oracle", "_build", "libref_c.so")      # the plain-C restatement (oracle/ref_c.c), one thread
    if os.path.exists(cso):
        import ctypes
        clib = ctypes.CDLL(cso)
        clib.refc_chamfer.restype = ctypes.c_double
    for n in (1024, 2048, 4096):
        rs = np.random.default_rng(99 + n)
        a = unit_cloud(rs, n)
        b = (a + 0.01 * rs.standard_normal(a.shape)).astype(np.float32)
        ta, tb = torch.from_numpy(a)[None], torch.from_numpy(b)[None]

        def torch_path():
            m = torch.cdist(ta, tb, p=2)
            return float(m.min(dim=2)[0].mean() + m.min(dim=1)[0].mean())
        paths = [("numpy_f64_ms_per_pair", lambda: orn.chamfer(a, b)), ("torch_cpu_ms_per_pair", torch_path)]
        if clib is not None:
            w1, w2 = np.empty(n), np.empty(n)
            i1, i2 = np.empty(n, np.int64), np.empty(n, np.int64)
            ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)                      # noqa: E731
            paths.append(("c_scalar_ms_per_pair", lambda: clib.refc_chamfer(ptr(a), n, ptr(b), n, ptr(w1), ptr(i1), ptr(w2), ptr(i2))))
        for key, fn in paths:
            fn(), fn()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            res[key][f"N{n}"] = sorted(ts)[1] * 1e3
    return res


def cpu_baseline(pcs, labels, seconds_budget=20.0):
    """The oracle's CPU restatement of the same CW iteration (torch-CPU, all host cores), bounded sample."""
    from oracle import ref_torch as ort
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, ncpu)))   # the GPU box gives one GPU a 16-core share
    model = ort.PointNetCls(k=NCLS)
    model.load_state_dict(seeded_state(model, 0))
    model.eval()
    data = torch.from_numpy(pcs).transpose(1, 2).contiguous()
    ori = data.clone()
    adv = (ori + torch.randn_like(ori) * 1e-7).requires_grad_()
    opt = torch.optim.Adam([adv], lr=LR)
    adv_func, dist, clip = ort.UntargetedLogitsAdvLoss(KAPPA), ort.ChannelFirst(ort.ChamferDist()), ort.ClipPointsLinf(BUDGET)
    w = torch.full((B,), 10.0, dtype=torch.float64)
    bestdist = np.full((B,), 1e10)
    n, t0 = 0, None
    while True:
        if n == 1:
            t0 = time.perf_counter()  # first iteration = warm-up
        logits = model(adv)[0]
        pred = logits.argmax(1).numpy()
        dv = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).detach().numpy()
        upd = (pred != labels.numpy()) & (dv < bestdist)
        bestdist = np.where(upd, dv, bestdist)
        loss = adv_func(logits, labels).mean() + dist(adv, ori, w).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        adv.data = clip(adv.clone().detach(), ori)
        n += 1
        if n > 1 and (time.perf_counter() - t0 > seconds_budget or n >= 41):
            break
    el = time.perf_counter() - t0
    return {"value": (n - 1) / el, "unit": "iters/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n - 1} iterations of the same CW/PointNet/Chamfer step (B={B}, N={NPTS}) via oracle/ref_torch.py on torch-CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if os.environ.get("PC3D_BENCH_REHEARSAL"):      # rehearsal on a one-GPU box: every rank on cuda:0, gloo collectives
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # launched by torch.distributed.run
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("PC3D_BENCH_REHEARSAL"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI

    pc3d = importlib.import_module("3dpointcloudattack_amd")
    pc3d.load()  # fail loudly if the HIP library is missing
    ops = importlib.import_module("3dpointcloudattack_amd.ops")
    M = importlib.import_module
    PointNetCls = M("3dpointcloudattack_amd.model.pointnet").PointNetCls
    CW = M("3dpointcloudattack_amd.attack.CW.CW_attack").CW
    adv_utils = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils")
    dist_utils = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    clip_utils = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    sharding = M("3dpointcloudattack_amd.sharding")
    graphed = M("3dpointcloudattack_amd.graphed")

    # frozen victim weights: built on rank 0, one RCCL broadcast of the flattened blob, never touched again
    model = PointNetCls(k=NCLS, feature_transform=False)
    trans_model = PointNetCls(k=NCLS, feature_transform=False)
    if rank == 0:
        model.load_state_dict(seeded_state(model, 0))
        trans_model.load_state_dict(seeded_state(trans_model, 1))
    model, trans_model = model.to(dev).eval(), trans_model.to(dev).eval()
    if dist_on:
        sharding.broadcast_frozen_weights([model, trans_model], src=0)

    # synthetic inputs: every rank attacks its own B clouds (weak scaling, per-GPU work fixed)
    rng = np.random.default_rng(1234 + 1 + 1000 * rank)
    pcs = np.stack([unit_cloud(rng, NPTS) for _ in range(B)])
    data = torch.from_numpy(pcs)
    with torch.no_grad():
        labels = model(data.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()

    attacker = CW(model, trans_model, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=KAPPA),
                  clip_func=clip_utils.ClipPointsLinf(budget=BUDGET), dist_func=dist_utils.ChamferDist(),
                  attack_lr=LR, binary_step=10, num_iter=500, device=dev)
    torch.manual_seed(1000 + rank)
    st = attacker._begin(data, labels)
    attacker._begin_binary_step(st)
    run = attacker._make_runner(st) if hasattr(attacker, "_make_runner") else None

    def step(i):
        if run is not None:
            run(i)
        else:
            attacker._iterate(st, i)

    stream = torch.cuda.current_stream()
    # ---- kernel-level measurements on rank 0, right BEFORE the headline loop: same process, same thermal state as the
    # headline (the sweep below runs several seconds of other work), and the GPU enters the W warm-up steps at its
    # working clocks instead of from idle (a 20-step timed region is 6 ms long: a clock ramp is 5 % of it)
    roofline = chamfer = None
    if rank == 0:
        # ---- the Chamfer kernel at N=4096 (north_star's second figure): VALU-bound, HBM share reported too.
        # Three timings per size: `values` = what Chamfer / Hausdorff VALUES need (utils/dis_utils_*.py, the metric's
        # kernel: scan + fold launches, no arg-min), `with_idx` = values + both arg-min index arrays (what the
        # backward of the distance functors needs), `two_scan` = the round-1 kernel (one scan per direction).
        pmc_all, pmc_file = {}, None
        for cand in ("r04_pmc_hbm_counters.json", "r03_pmc_hbm_counters.json", "r02_pmc_hbm_counters.json"):
            if os.path.exists(os.path.join(ROOT, "profiles", cand)):
                pmc_file = os.path.join("profiles", cand)
                pmc_all = json.load(open(os.path.join(ROOT, pmc_file)))
                break
        pmc_src = {"file": pmc_file, "collected_at": pmc_provenance(pmc_all), "commits_behind_head": pmc_commit_distance(pmc_all),
                   "note": "HBM counters come from separate rocprofv3 --pmc passes of this command (tools/prof_pmc.sh), "
                           "not from this run"}

        def counter_bytes(keys):
            """HBM bytes per call (sum over the launches of one call) from the committed PMC passes of THIS bench
            command (tools/prof_pmc.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs, mean per dispatch; KiB units;
            FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note for 16-B-per-lane reads), or None when a kernel of
            the call is not covered. keys: "kernel name|grid=<work-items>|run=<k>" as that script writes them (run k = the
            k-th change of that kernel's grid in dispatch order: this function's call order below fixes it)."""
            tot = 0.0
            for k in keys:
                c = pmc_all.get(k)
                if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                    return None
                tot += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            return tot

        def chamfer_point(Nc):
            a = torch.randn(B, Nc, 3, device=dev)
            b = a + 0.01 * torch.randn_like(a)
            res = {}
            for tag, kw in (("values", {"want_idx": False, "two_scan": False}), ("with_idx", {"two_scan": False}),
                            ("two_scan", {"two_scan": True})):
                res[tag] = graph_ms(lambda: ops.nn_bidir_raw(a, b, **kw))
            c_ms = res["values"]
            alg_bytes = B * (2 * Nc * 12 + 2 * Nc * 8)          # 40*N bytes per cloud pair
            alg_ops = 10.0 * B * Nc * Nc                         # 8 (shared distance) + 2 (running mins) per pair
            issued_ops = 8.4 * B * Nc * Nc                       # instructions the shared-evaluation scan issues per pair
            return {"kernel": "nn_shared_kernel + nn_shared_finalize_kernel", "config": f"B={B} N=M={Nc} bidirectional",
                    "timing": "10 calls per replayed hipGraph, HIP events around 20 replays",
                    "launch_us": c_ms * 1e3, "with_idx_us": res["with_idx"] * 1e3, "two_scan_us": res["two_scan"] * 1e3,
                    "hbm_alg_GBps": alg_bytes / (c_ms * 1e-3) / 1e9,
                    "hbm_frac": alg_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "valu_frac": alg_ops / (c_ms * 1e-3) / VALU_LANE_OPS_PEAK,
                    "valu_frac_with_idx": alg_ops / (res["with_idx"] * 1e-3) / VALU_LANE_OPS_PEAK,
                    "valu_frac_two_scan": alg_ops / (res["two_scan"] * 1e-3) / VALU_LANE_OPS_PEAK,
                    "valu_issue_frac": issued_ops / (c_ms * 1e-3) / VALU_LANE_OPS_PEAK, "bound": "valu"}
        chamfer = chamfer_point(4096)
        # B=32, N=M=4096 is the FIRST Chamfer point measured (run 0): scan grid 16 tiles x 32 clouds x 8 splits of 256
        # threads, fold grid 16 x 32 x 2 of 256 (the fold's counters average its values-only and with-index calls)
        cb = counter_bytes(("void pc3d::nn_shared_kernel<4, false>|grid=1048576|run=0",
                            "pc3d::nn_shared_finalize_kernel|grid=262144|run=0"))
        chamfer["hbm_counter_bytes"] = cb                      # FETCH+WRITE of the values path's launches, per call
        chamfer["hbm_counter_GBps"] = (cb / (chamfer["launch_us"] * 1e-6) / 1e9) if cb else None
        chamfer["hbm_counter_source"] = pmc_src
        chamfer["other_sizes"] = {f"N{n}": {k: v for k, v in chamfer_point(n).items()
                                            if k in ("launch_us", "with_idx_us", "two_scan_us", "hbm_alg_GBps", "valu_frac", "timing",
                                                     "valu_frac_with_idx", "valu_frac_two_scan")}
                                  for n in (1024, 2048)}
        # ---- roofline of the dominant kernel: the fused per-point MLP + max forward (fp32 MFMA), measured immediately
        # before the headline loop after 50 untimed launches — deliberately NOT after a long sustained run of itself: 150 ms of
        # back-to-back tower launches raise the clocks and the launch drops to 73.7 us (0.78 of the peak), but inside the
        # iteration, between a dozen small launches, it runs at 79 us (rocprof, profiles/r04_bench_kernel_stats.*), and that is
        # the figure the roofline of the STEP should carry
        x = st["adv"].detach()
        tower = model.feat.folded()
        flops = 2.0 * B * NPTS * (3 * 64 + 64 * 128 + 128 * 1024)   # algorithmic flops per launch
        for _ in range(50):
            ops.pointmlp3_max_fwd_raw(x, tower, False, fold=False)
        k_ms = ev_ms(lambda: ops.pointmlp3_max_fwd_raw(x, tower, False, fold=False), 50, stream)
        ach = flops / (k_ms * 1e-3) / 1e12
        traffic = None
        if pmc_all:
            # HBM bytes per launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
            # runs of this same command at this round's kernels, KiB units). gfx950 correction (MI355X_MICROARCH.md
            # §HBM): FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane reads -> doubled; WRITE_SIZE is exact.
            # (PMC collection needs the rocprofv3 wrapper, so it cannot run inside this process; the file is regenerated
            # by tools/prof_pmc.sh whenever a kernel on this line changes.)
            c = pmc_all.get("pc3d::pointmlp3_max_fwd_kernel|grid=131072|run=0")
            if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        roofline = {"kernel": "pointmlp3_max_fwd_kernel", "bound": "mfma", "achieved": ach,
                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_source": pmc_src, "launch_us": k_ms * 1e3, "launches_per_step": 2}

    it = 0
    for _ in range(args.warmup):
        step(it)
        it += 1
    if hasattr(run, "flush"):
        run.flush()        # warm-up iterations all executed before the clock starts
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(it)
        it += 1
    if hasattr(run, "flush"):
        run.flush()        # EXACTLY `steps` iterations are inside the timed region
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    def across_ranks(x):
        """(max over ranks, [value of every rank]) of a host scalar; one all_reduce (works on RCCL and on gloo)."""
        if not dist_on:
            return x, [x]
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = x
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per = [float(v) for v in t.cpu()]
        return max(per), per

    elapsed, per_rank_elapsed = across_ranks(elapsed)
    ms_per_step = elapsed * 1e3 / args.steps
    iters_per_s = world * args.steps / elapsed     # whole job: every rank advances its own batch each step

    # ---- the other shapes of the same path, on EVERY rank (the collectives inside line up): north_star's N = 2048 /
    # 4096 clouds, the reference's default L2 regulariser, and one GPU's share of BASELINE configs[4] (CW on CurveNet,
    # 32 of the 256 clouds, N=4096). Reported as whole-job rates over the slowest rank, with the per-rank times.
    sweep = None
    if not args.no_sweep:
        sweep = {}

        def time_cw(victim, trans, npts, dname, warm, iters, seed):
            rs = np.random.default_rng(seed + 1000 * rank)
            d2 = torch.from_numpy(np.stack([unit_cloud(rs, npts) for _ in range(B)]))
            with torch.no_grad():
                l2 = victim(d2.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
            atk = CW(victim, trans, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=KAPPA),
                     clip_func=clip_utils.ClipPointsLinf(budget=BUDGET),
                     dist_func=dist_utils.L2Dist() if dname == "l2" else dist_utils.ChamferDist(),
                     attack_lr=LR, binary_step=10, num_iter=500, device=dev)
            torch.manual_seed(7 + rank)
            s2 = atk._begin(d2, l2)
            atk._begin_binary_step(s2)
            r2 = atk._make_runner(s2)
            for i in range(warm):
                r2(i)
            ms, per = across_ranks(ev_ms(r2, iters, stream))
            return {"iters_per_s": world * 1e3 / ms, "ms_per_step": ms, "per_rank_ms": per}

        for npts, dname in ((NPTS, "l2"), (2048, "chamfer"), (4096, "chamfer")):
            sweep[f"cw_pointnet_{dname}_B{B}_N{npts}"] = time_cw(model, trans_model, npts, dname, 10, 60, 4321 + npts)
        CurveNet = M("3dpointcloudattack_amd.model.curvenet").CurveNet
        cnet = CurveNet(num_classes=NCLS)
        if rank == 0:
            cnet.load_state_dict(seeded_state(cnet, 9))
        cnet = cnet.to(dev).eval()
        if dist_on:
            sharding.broadcast_frozen_weights([cnet], src=0)
        sweep[f"cfg5_share_cw_curvenet_chamfer_B{B}_N4096"] = time_cw(cnet, cnet, 4096, "chamfer", 4, 12, 555)

        # ---- BASELINE configs[2], configs[3] and the GeoA3 half of configs[4] as WHOLE attack calls (median of three
        # slopes, see slope_ms), each with the roofline of the kernel that dominates its loop — measured stand-alone with
        # HIP events at the layer's shape; every rank runs them, the slowest rank is reported
        def kernel_roof(name, fn, flops=None, lane_ops=None, hbm_bytes=None, note=None, per=5, reps=8):
            """Launch time of the layer-shaped call as `per` calls inside ONE replayed hipGraph (graph_ms) — how the attack loops
            launch it and how the Chamfer points above are timed; HIP events around eager Python calls time the host for
            anything under ~100 us and read 5-15 % high for the rest (round-3 review: conv5 372 us here vs 341 in the loop
            profile). Falls back to eager events (and says so) if the call cannot be captured."""
            for _ in range(3):
                fn()
            try:
                us, how = (graph_ms(fn, per=per, reps=reps, warm_ms=CLOCK_WARM_MS) * 1e3,
                           f"{per} calls per replayed hipGraph, HIP events around {reps} replays after {CLOCK_WARM_MS:.0f} ms of the same replays "
                           "(clock ramp)")
            except Exception as e:      # noqa: BLE001 — a call that syncs with the host cannot be captured
                torch.cuda.synchronize()
                us, how = ev_ms(fn, 20, stream) * 1e3, f"eager calls, HIP events (capture failed: {type(e).__name__})"
            r = {"kernel": name, "launch_us": us, "timing": how}
            if flops is not None:
                r.update(bound="mfma", alg_flops=flops, achieved=flops / us / 1e6, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                         frac=flops / us / 1e6 / MFMA_F32_PEAK_TFLOPS)
            if lane_ops is not None:
                r.update(bound="valu", alg_lane_ops=lane_ops, achieved=lane_ops / us / 1e6, peak=VALU_LANE_OPS_PEAK / 1e12,
                         unit="T lane-op/s", frac=lane_ops / (us * 1e-6) / VALU_LANE_OPS_PEAK)
            if hbm_bytes is not None:
                r.update(bound="hbm", alg_bytes=hbm_bytes, achieved=hbm_bytes / us / 1e3, peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=hbm_bytes / us / 1e3 / HBM_PEAK_GBS)
            if note:
                r["note"] = note
            return r

        def whole(tag, run_attack, base, inc, roof):
            med, slopes = slope_ms(run_attack, base, inc)
            ms, per = across_ranks(med)
            sweep[tag] = {"ms_per_iter": ms, "iters_per_s": world * 1e3 / ms, "per_rank_ms": per, "slopes_ms_rank0": slopes,
                          "timing": f"whole attack() calls, wall clock, median of three {inc}-iteration slopes", "roofline": roof}

        import types
        ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
        DGCNN = M("3dpointcloudattack_amd.model.dgcnn").DGCNN
        dg = DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=NCLS)
        if rank == 0:
            dg.load_state_dict(seeded_state(dg, 5))
        dg = dg.to(dev).eval()
        if dist_on:
            sharding.broadcast_frozen_weights([dg], src=0)

        def clouds(n, npts, seed):
            rs = np.random.default_rng(seed + 1000 * rank)
            d = torch.from_numpy(np.stack([unit_cloud(rs, npts) for _ in range(n)]))
            return d

        def labels_of(net, d):
            with torch.no_grad():
                return net(d.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()

        def geo_runner(net, d, lab, npts):
            def run_attack(iters):
                torch.manual_seed(3 + rank)
                np.random.seed(3 + rank)
                ga.geoA3_attack(net, None, None, None, None, None, d, lab, geo_cfg(iter_max_steps=iters, npoint=npts), 0, 1)
            return run_attack

        d3 = clouds(B, 1024, 777)
        xg = torch.randn(B * 1024, 512, device=dev)
        wg = torch.randn(1024, 512, device=dev) / 512 ** 0.5
        whole(f"cfg3_geoa3_dgcnn_B{B}_N1024", geo_runner(dg, d3, labels_of(dg, d3), 1024), 8, 40,
              kernel_roof("gemm_nt_kernel (conv5 forward, [32768,1024,512])", lambda: ops.gemm_nt(xg, wg, None, "leaky", 0.2),
                          flops=2.0 * B * 1024 * 1024 * 512))
        del xg, wg

        knn_mod = M("3dpointcloudattack_amd.attack.KNN.KNN_attack")
        SSG = M("3dpointcloudattack_amd.model.pointnet2_SSG").PointNet_Ssg
        ssg = SSG(NCLS)
        if rank == 0:
            ssg.load_state_dict(seeded_state(ssg, 3))
        ssg = ssg.to(dev).eval()
        if dist_on:
            sharding.broadcast_frozen_weights([ssg], src=0)
        B4 = 64
        d4 = clouds(B4, 2048, 888)
        torch.manual_seed(11 + rank)
        l4 = labels_of(ssg, d4)

        def knn_run(iters):
            atk = knn_mod.CWKNN(ssg, None, None, None, None, None, adv_utils.UntargetedLogitsAdvLoss(kappa=15.),
                                dist_utils.ChamferDist(method='adv2ori'), clip_utils.ProjectInnerClipLinf(budget=BUDGET),
                                attack_lr=LR, num_iter=iters, device=dev)
            torch.manual_seed(5 + rank)
            np.random.seed(5 + rank)
            atk.attack(d4, l4)
        # dominant kernel of the iteration: the second level's set-abstraction chain (gather -> layers 1-2-3 -> group max, one
        # launch over the table of 8-row units that hold listed points), stand-alone on a grouping with the real padding
        # structure: 512-point clouds, 128 FPS centres, ball query r = 0.4 / 64 samples
        c4 = clouds(B4, 512, 4242).to(dev)
        with torch.no_grad():
            f4 = ops.fps(c4, 128, None)
            ctr4 = torch.gather(c4, 1, f4.long()[..., None].expand(-1, -1, 3)).contiguous()
            idx4 = ops.ball_query(0.4, 64, c4, ctr4)
            P4, Bc4 = torch.randn(B4, 512, 128, device=dev), torch.randn(B4, 128, 128, device=dev)
            L4 = [(torch.randn(128, 128, device=dev) / 128 ** 0.5, torch.randn(128, device=dev)),
                  (torch.randn(256, 128, device=dev) / 128 ** 0.5, torch.randn(256, device=dev))]
            unit4 = ops.sa_chain_table_unit(128, 64, 128, 128, 256)        # rows per unit of the table this shape takes (8)
            blk4 = ops.sa_blocks(idx4, unit4)
            kept4 = int((blk4[0][:int(blk4[1].item()) * (8 if unit4 == 8 else 4)] >= 0).sum().item())
        per_row = 2.0 * (128 * 128 + 128 * 256)

        def sa2_chain():
            with torch.no_grad():
                ops.grouped_mlp_max(P4, Bc4, idx4, L4, blocks=blk4)
        roof4 = kernel_roof("sa_chain_u8_kernel<2> (SSG SA2: 8192 groups x 64 rows, 128 -> 128 -> 256, over the unit table)",
                            sa2_chain, flops=per_row * unit4 * kept4,
                            note=f"flops of the {kept4} kept {unit4}-row units of {B4 * 128 * 64 // unit4} (the others are padding "
                                 "copies of a group's first point); the reference's count for every row is alg_flops_all_rows")
        roof4["alg_flops_all_rows"] = per_row * B4 * 128 * 64      # (for reference only: flops that are never executed are no roofline)
        whole(f"cfg4_knn_ssg_B{B4}_N2048", knn_run, 8, 40, roof4)
        del c4, P4, Bc4, idx4, blk4

        d5 = clouds(B, 4096, 999)
        p5 = d5.to(dev)
        whole(f"cfg5_share_geoa3_curvenet_B{B}_N4096", geo_runner(cnet, d5, labels_of(cnet, d5), 4096), 4, 12,
              kernel_roof("fps_kernel<16> (4096 -> 1024, the front of every CurveNet forward)", lambda: ops.fps(p5, 1024, None),
                          lane_ops=12.0 * B * 4096 * 1024, per=2, reps=4,
                          note="a chain of 1024 dependent arg-max steps on one workgroup per cloud (32 of 256 CUs): "
                               "latency-bound, the VALU fraction only says how far from compute-bound it is"))

        # ---- the targeted AOF loop (attack/AOF/TAOF_attack.py:137-224) on the headline victim, B=32, N=1024,
        # low_pass 100: two fused forward/backward pairs + two success-check forwards + the spectral re-projection per
        # iteration; the eigen-decomposition (torch.linalg.eigh, once per binary step) cancels in the slopes
        taof = M("3dpointcloudattack_amd.attack.AOF.TAOF_attack")
        with torch.no_grad():
            lp_all = model(data.transpose(1, 2).contiguous().to(dev))[0]
        tgt_aof = lp_all.topk(2)[1][:, 1].cpu()

        def aof_run(iters):
            atk = taof.CWTAOF(model, adv_utils.LogitsAdvLoss(kappa=0.), dist_utils.L2Dist(), attack_lr=LR, binary_step=1,
                              num_iter=iters, GAMMA=0.5, low_pass=100, clip_func=clip_utils.ClipPointsLinf(budget=BUDGET), device=dev)
            torch.manual_seed(9 + rank)
            atk.attack(data, tgt_aof, labels)
        Vq = torch.linalg.qr(torch.randn(B, NPTS, NPTS, device=dev))[0].contiguous()
        Vqt = Vq.transpose(1, 2).contiguous()
        adv_q = torch.randn(B, 3, NPTS, device=dev)
        bufs_q = [torch.empty_like(adv_q) for _ in range(3)]
        whole(f"aof_taof_pointnet_B{B}_N{NPTS}", aof_run, 100, 300,     # (long slopes: every call carries one eigh of ~0.12 s whose jitter must cancel)
              kernel_roof("rowdot3_kernel x2 (pc3d_spectral_reproject_f32: coeff = adv V, lfc / hfc = coeff_lo|hi V^T)",
                          lambda: ops.spectral_reproject(adv_q, Vq, Vqt, 100, *bufs_q), hbm_bytes=2.0 * B * NPTS * NPTS * 4,
                          note="both launches of one re-projection; algorithmic bytes = V and V^T read once each "
                               "(the three torch.bmm it replaces: 66 us in the same harness, tools/bench_spectral.py)"))
        del Vq, Vqt

        # ---- one WHOLE CW.attack at the headline shape, 2 binary steps x 500 iterations: wall time of the call —
        # upload, clean forward, graph capture, the two binary-search boundaries, final checks and the D2H of the result
        atk_full = CW(model, trans_model, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=KAPPA),
                      clip_func=clip_utils.ClipPointsLinf(budget=BUDGET), dist_func=dist_utils.ChamferDist(),
                      attack_lr=LR, binary_step=2, num_iter=500, device=dev)
        torch.manual_seed(1000 + rank)
        np.random.seed(1000 + rank)
        torch.cuda.synchronize()
        tw = time.perf_counter()
        bd_full, _, sn_full = atk_full.attack(data, labels)
        torch.cuda.synchronize()
        wall, per = across_ranks(time.perf_counter() - tw)
        sweep[f"whole_cw_attack_pointnet_chamfer_B{B}_N{NPTS}_2x500"] = {
            "wall_s": wall, "iters_per_s_incl_boundaries": world * 1000.0 / wall, "per_rank_s": per,
            "success_rank0": int(sn_full), "includes": "upload, clean forward, hipGraph capture, 2 binary-search boundaries, "
                                                        "final attack / shuffle / transfer checks, D2H of the result"}

    # ---- the one collective at the end of a job: gather every rank's results (unequal shards are padded inside)
    gather = None
    if dist_on:
        res_local = [st["o_bestattack"].detach(), st["o_bestdist"].detach(), st["pred"].detach()]
        if os.environ.get("PC3D_BENCH_REHEARSAL"):      # gloo has no GPU all_gather
            res_local = [t.cpu() for t in res_local]
        torch.cuda.synchronize()
        tg = time.perf_counter()
        res_all = sharding.gather_results(res_local)
        torch.cuda.synchronize()
        gms, _ = across_ranks((time.perf_counter() - tg) * 1e3)
        gather = {"ms": gms, "tensors": [list(t.shape) for t in res_all],
                  "bytes": int(sum(t.numel() * t.element_size() for t in res_all))}

    out = None
    if rank == 0:
        out = {
            "metric": "attack iters/s (B=32, N=1024, PointNet) + Chamfer HBM GB/s vs peak",
            "value": iters_per_s, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CW attack on PointNet(k=40, seeded random init), B=32 N=1024 fp32, "
                                   "UntargetedLogitsAdvLoss(kappa=30) + ChamferDist(adv2ori) + ClipPointsLinf(0.18), "
                                   "Adam lr 1e-2; one step = one hot-loop iteration for the whole batch",
                       "batch_per_gpu": B, "points": NPTS, "parallelism": f"independent attack instances x{world}",
                       "cloud_iters_per_s": iters_per_s * B},
            "roofline": roofline, "chamfer": chamfer,
        }
        out["sweep"] = sweep
        out["per_rank_ms_per_step"] = [e * 1e3 / args.steps for e in per_rank_elapsed]
        out["gather_results"] = gather
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pcs, labels)
            out["chamfer"]["cpu_baseline"] = chamfer_cpu_baseline()
            out["chamfer"]["cpu_baseline"]["gpu_us_per_pair"] = {
                "N4096": chamfer["launch_us"] / B, **{k: v["launch_us"] / B for k, v in chamfer["other_sizes"].items()}}
        elif world == 1:
            out["cpu_baseline"] = None
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
