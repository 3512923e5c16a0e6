#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (read-only at /root/reference) on seeded inputs.

Run in the build container only (the reference does not travel to the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [section ...]
Each fixture stores the inputs AND the reference's outputs, so the tests need nothing but the .npz.
Only data is written — no reference source is copied.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("PC3D_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def unit_cloud(rng, n):
    """Uniform-in-ball cloud, centred and scaled to the unit sphere like dataset/bosphorus_dataset.py:74-76."""
    g = rng.standard_normal((n, 3))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    p = g * rng.random((n, 1)) ** (1.0 / 3.0)
    p = p - p.mean(axis=0, keepdims=True)
    p = p / np.max(np.linalg.norm(p, axis=1))
    return p.astype(np.float32)


def perturbed(rng, p, sigma=0.01, budget=0.18):
    d = (sigma * rng.standard_normal(p.shape)).astype(np.float32)
    n = np.linalg.norm(d, axis=1, keepdims=True)
    d = d * np.minimum(1.0, budget / (n + 1e-9))
    return (p + d).astype(np.float32)


def load_real_clouds():
    """A few in-repo clouds (data files, not code): saved CW / AOF outputs and a raw scan."""
    out = {}
    p = os.path.join(REF, "attack/CW/AdvData/PointNet/0-88-63.txt")
    if os.path.exists(p):
        out["cw_adv_0_88_63"] = np.loadtxt(p, dtype=np.float32)[:, :3]
    for k in (0, 1):
        p = os.path.join(REF, f"attack/AOF/AdvData/PointNet/{k}.txt")
        if os.path.exists(p):
            out[f"aof_adv_{k}"] = np.loadtxt(p, dtype=np.float32)[:, :3]
    p = os.path.join(REF, "AddData/face0424.txt")
    if os.path.exists(p):
        try:
            arr = np.loadtxt(p, delimiter=",", dtype=np.float32)
        except ValueError:
            arr = np.loadtxt(p, dtype=np.float32)
        out["face0424"] = arr[:, :3]
    return out


# ---------------------------------------------------------------------------------------------------------
def gen_metrics():
    from utils import dis_utils_numpy as dun
    from utils import dis_utils_torch as dut
    from attack.CW.CW_utils.distance import chamfer as cw_chamfer, hausdorff as cw_hausdorff

    rng = np.random.default_rng(1234)
    fx = {}

    # --- numpy metrics: worked examples + hand case + random + real clouds
    pairs = {
        "kat_ones_twos": (np.ones((3, 3), np.float32), 2 * np.ones((3, 3), np.float32)),
        "hand_asym": (np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32),
                      np.array([[0, 0, 1], [3, 0, 0]], np.float32)),
        "single_point": (np.array([[0.5, -0.25, 1.0]], np.float32), np.array([[0.5, -0.25, 1.0]], np.float32)),
    }
    for n in (64, 257, 1024, 2048):
        p = unit_cloud(rng, n)
        pairs[f"rand_{n}"] = (perturbed(rng, p), p)
    pa, pb = unit_cloud(rng, 300), unit_cloud(rng, 1000)
    pairs["ragged_300_1000"] = (pa, pb)
    real = load_real_clouds()
    names = sorted(real)
    for nm in names:
        c = real[nm]
        sub = c[:: max(1, c.shape[0] // 1500)][:1500]
        pairs[f"real_{nm}"] = (perturbed(rng, sub, sigma=0.01 * float(np.abs(sub).max())), sub)
    fx["np_names"] = np.array(sorted(pairs))
    for nm in sorted(pairs):
        a, b = pairs[nm]
        fx[f"np_{nm}_a"], fx[f"np_{nm}_b"] = a, b
        fx[f"np_{nm}_out"] = np.array([dun.chamfer(a, b), dun.sgd_hausdorff_dis(a, b),
                                       dun.sgd_hausdorff_dis(b, a), dun.bid_hausdorff_dis(a, b)], np.float64)
    a, b = pairs["hand_asym"]
    fx["np_hand_asym_M"] = dun.pairwise_distances(a, b)
    a, b = pairs["rand_64"]
    fx["np_rand_64_M"] = dun.pairwise_distances(a, b)

    # --- torch metrics on [B,3,N]
    tpairs = {
        "kat": (np.array([[[1, 1, 1], [1, 1, 1], [1, 1, 1]]], np.float32),
                np.array([[[2, 2, 3], [2, 2, 2], [2, 2, 2]]], np.float32)),
    }
    for n, bsz in ((128, 2), (1024, 2)):
        p = np.stack([unit_cloud(rng, n) for _ in range(bsz)])
        q = np.stack([perturbed(rng, x) for x in p])
        tpairs[f"rand_{n}"] = (q.transpose(0, 2, 1).copy(), p.transpose(0, 2, 1).copy())
    fx["t_names"] = np.array(sorted(tpairs))
    for nm in sorted(tpairs):
        a, b = tpairs[nm]
        ta, tb = torch.from_numpy(a), torch.from_numpy(b)
        fx[f"t_{nm}_a"], fx[f"t_{nm}_b"] = a, b
        fx[f"t_{nm}_out"] = np.array([float(dut.euclidean_distances(ta, tb)), float(dut.chamfer(ta, tb)),
                                      float(dut.sgd_hausdorff_dis(ta, tb)), float(dut.bid_hausdorff_dis(ta, tb))],
                                     np.float64)
    a, b = tpairs["rand_128"]
    fx["t_rand_128_M"] = dut.pairwise_distances(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    # gradient of the (quirky) chamfer wrt a
    ta = torch.from_numpy(a).clone().requires_grad_()
    dut.chamfer(ta, torch.from_numpy(b)).backward()
    fx["t_rand_128_grad_a"] = ta.grad.numpy()

    # --- CW functors: (loss1, loss2) [B] for preds/gts [B,N,3] + gradients through the reference autograd
    cw = {}
    for nm, (n1, n2, bsz) in {"b4_256": (256, 256, 4), "b2_1024": (1024, 1024, 2), "ragged": (200, 333, 3)}.items():
        g = np.stack([unit_cloud(rng, n2) for _ in range(bsz)])
        if n1 == n2:
            p = np.stack([perturbed(rng, x) for x in g])
        else:
            p = np.stack([unit_cloud(rng, n1) for _ in range(bsz)])
        cw[nm] = (p, g)
    fx["cw_names"] = np.array(sorted(cw))
    for nm in sorted(cw):
        p, g = cw[nm]
        fx[f"cw_{nm}_preds"], fx[f"cw_{nm}_gts"] = p, g
        # values from a float64 run of the reference code (its fp32 expansion is itself ~1e-5 off, SURVEY A-3)
        tp64, tg64 = torch.from_numpy(p).double(), torch.from_numpy(g).double()
        l1, l2 = cw_chamfer(tp64, tg64)
        h1, h2 = cw_hausdorff(tp64, tg64)
        fx[f"cw_{nm}_chamfer"] = np.stack([l1.numpy(), l2.numpy()])
        fx[f"cw_{nm}_hausdorff"] = np.stack([h1.numpy(), h2.numpy()])
        # and the reference's own fp32 outputs, for the record (tolerance documented in the test)
        l1f, l2f = cw_chamfer(torch.from_numpy(p), torch.from_numpy(g))
        fx[f"cw_{nm}_chamfer_f32"] = np.stack([l1f.numpy(), l2f.numpy()])
        # gradients (float64 reference autograd): d(sum_b w1*l1 + w2*l2)/dpreds
        w1 = torch.linspace(0.5, 1.5, p.shape[0], dtype=torch.float64)
        w2 = torch.linspace(2.0, 1.0, p.shape[0], dtype=torch.float64)
        tp = tp64.clone().requires_grad_()
        tg = tg64.clone().requires_grad_()
        l1, l2 = cw_chamfer(tp, tg)
        ((l1 * w1).sum() + (l2 * w2).sum()).backward()
        fx[f"cw_{nm}_w"] = np.stack([w1.numpy(), w2.numpy()])
        fx[f"cw_{nm}_chamfer_gpreds"], fx[f"cw_{nm}_chamfer_ggts"] = tp.grad.numpy(), tg.grad.numpy()
        tp = tp64.clone().requires_grad_()
        h1, h2 = cw_hausdorff(tp, tg64)
        ((h1 * w1).sum() + (h2 * w2).sum()).backward()
        fx[f"cw_{nm}_hausdorff_gpreds"] = tp.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **fx)
    print("metrics.npz:", len(fx), "arrays")


def install_cpu_shim():
    """SURVEY §8(c) harness shim: lets the reference's hard-coded .cuda() calls run on this GPU-less box.
    Lives only in this generator; never edits the reference, never ships in the product."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    # On a GPU `.cpu()` COPIES device memory to the host. With everything already on the CPU it would return
    # `self`, and the reference's `input_val = adv_data.detach().cpu().numpy()` (CW_attack.py:133) would alias the
    # tensor Adam then updates in place — an artefact of the shim, not reference behaviour. Emulate the copy.
    _orig_cpu = torch.Tensor.cpu
    torch.Tensor.cpu = lambda self, *a, **k: _orig_cpu(self, *a, **k).clone()
    # model/dgcnn.py:209 and curvenet hard-code torch.device('cuda...') in torch.arange: drop the device kwarg
    _orig_arange = torch.arange
    torch.arange = lambda *a, **k: _orig_arange(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})


def canonical_unsorted_topk():
    """CurveNet picks its curve start points with ``torch.topk(..., sorted=False)`` (model/curvenet_util.py:457): the
    ORDER of the k indices is then unspecified and differs between torch's CPU and GPU kernels — and the walk's
    momentum step mixes values of different curves by their position (model/walk.py:104-105), so the reference's own
    output depends on that order. For fixtures that pin the walk inside the whole network, the unspecified order is
    fixed to the descending-score order (one valid instance of sorted=False); the mirror uses the same order."""
    _orig = torch.topk

    def topk(input, k, dim=-1, largest=True, sorted=True, **kw):
        return _orig(input, k, dim=dim, largest=largest, sorted=True, **kw)
    torch.topk = topk


def _seeded_pointnet(cls, k, seed):
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))  # repo root, for oracle.ref_torch helpers
    from oracle.ref_torch import seeded_state_dict, state_sha256
    m = cls(k=k, feature_transform=False)
    sd = seeded_state_dict(m, seed)
    m.load_state_dict(sd)
    m.eval()
    return m, state_sha256(sd)


def gen_pointnet():
    """Logits / STN transform / input-gradient of the reference PointNetCls on seeded weights."""
    from model.pointnet import PointNetCls
    rng = np.random.default_rng(4321)
    fx = {}
    model, sha = _seeded_pointnet(PointNetCls, 40, 0)
    fx["sha256"] = np.array(sha)
    for nm, (B, N) in {"b2_n1024": (2, 1024), "b3_n200": (3, 200)}.items():
        x = np.stack([unit_cloud(rng, N) for _ in range(B)]).transpose(0, 2, 1).copy()  # [B,3,N]
        tx = torch.from_numpy(x).requires_grad_()
        logp, trans, _ = model(tx)
        w = torch.from_numpy(rng.standard_normal(logp.shape).astype(np.float32))
        (logp * w).sum().backward()
        fx[f"{nm}_x"], fx[f"{nm}_logp"], fx[f"{nm}_trans"] = x, logp.detach().numpy(), trans.detach().numpy()
        fx[f"{nm}_w"], fx[f"{nm}_gx"] = w.numpy(), tx.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "pointnet.npz"), **fx)
    print("pointnet.npz:", len(fx), "arrays")


def gen_cw():
    """Short runs of the REAL reference CW.attack (B=1, as the reference requires) on seeded PointNet weights.
    Trajectories are captured through the dist_func hook (the loop hands it adv_data every iteration)."""
    install_cpu_shim()
    import contextlib
    import io
    from model.pointnet import PointNetCls
    from attack.CW.CW_attack import CW
    from attack.CW.CW_utils.adv_utils import UntargetedLogitsAdvLoss, LogitsAdvLoss
    from attack.CW.CW_utils.dist_utils import L2Dist, ChamferDist
    from attack.CW.CW_utils.clip_utils import ClipPointsLinf

    class Recorder(torch.nn.Module):
        def __init__(self, inner, transpose):
            super().__init__()
            self.inner, self.transpose, self.log = inner, transpose, []

        def forward(self, adv, ori, weights=None, batch_avg=True):
            self.log.append(adv.detach().numpy().copy())
            if self.transpose:  # point-set functor documented for [B,K,3] (additional_exp/CW_attack.py:151-153)
                return self.inner(adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous(), weights, batch_avg)
            return self.inner(adv, ori, weights, batch_avg)

    rng = np.random.default_rng(777)
    model, sha = _seeded_pointnet(PointNetCls, 40, 0)
    trans_model, _ = _seeded_pointnet(PointNetCls, 40, 1)
    fx = {"sha256": np.array(sha)}
    cases = {
        "l2_untarget": dict(dist="l2", method="untarget", kappa=5., N=256, steps=3, iters=15),
        "chamfer_untarget": dict(dist="chamfer", method="untarget", kappa=5., N=256, steps=3, iters=15),
        "l2_target": dict(dist="l2", method="target", kappa=0., N=200, steps=2, iters=25),
    }
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        c = cases[nm]
        pc = unit_cloud(rng, c["N"])[None]  # [1,N,3]
        with torch.no_grad():
            clean = int(torch.argmax(model(torch.from_numpy(pc).transpose(1, 2).contiguous())[0], dim=1))
        tgt = clean if c["method"] == "untarget" else int(torch.topk(model(torch.from_numpy(pc).transpose(1, 2).contiguous())[0], 2)[1][0, 1])
        adv_func = UntargetedLogitsAdvLoss(kappa=c["kappa"]) if c["method"] == "untarget" else LogitsAdvLoss(kappa=c["kappa"])
        rec = Recorder(L2Dist() if c["dist"] == "l2" else ChamferDist(), transpose=(c["dist"] != "l2"))
        atk = CW(model, trans_model, adv_func=adv_func, clip_func=ClipPointsLinf(budget=0.18), dist_func=rec,
                 attack_lr=1e-2, init_weight=10., max_weight=80., binary_step=c["steps"], num_iter=c["iters"],
                 attack_method=c["method"])
        torch.manual_seed(1000)
        np.random.seed(1000)
        with contextlib.redirect_stdout(io.StringIO()):
            bd, ba, sn = atk.attack(torch.from_numpy(pc), torch.tensor([tgt]))
        fx[f"{nm}_pc"], fx[f"{nm}_target"] = pc, np.array([tgt])
        fx[f"{nm}_cfg"] = np.array([c["steps"], c["iters"], c["kappa"]])
        fx[f"{nm}_bestdist"], fx[f"{nm}_bestattack"], fx[f"{nm}_success"] = bd, ba.astype(np.float32), np.array(sn)
        fx[f"{nm}_traj"] = np.stack(rec.log).astype(np.float32)[:, 0]  # [steps*iters, 3, K]
        fx[f"{nm}_fails"] = np.array([atk.attack_fail, atk.shuffle_fail, atk.trans_fail])
    np.savez_compressed(os.path.join(OUT, "cw.npz"), **fx)
    print("cw.npz:", len(fx), "arrays")


def gen_cw_full():
    """The FULL binary-search schedule at the headline size: the REAL reference's CW.attack on PointNet,
    B=1, N=1024, L2Dist (reproducible arithmetic), 10 binary steps x 100 iterations (attack/CW/CW_attack.py:93-200 with the
    Eval_CW.py:79-90 wiring), two untargeted seeds and one targeted. Stored: o_bestdist, o_bestattack, success, the fail
    counters, and the weight each binary step ran with (captured through the dist_func hook, which receives
    torch.from_numpy(current_weight) every iteration)."""
    install_cpu_shim()
    import contextlib
    import io
    from model.pointnet import PointNetCls
    from attack.CW.CW_attack import CW
    from attack.CW.CW_utils.adv_utils import UntargetedLogitsAdvLoss, LogitsAdvLoss
    from attack.CW.CW_utils.dist_utils import L2Dist
    from attack.CW.CW_utils.clip_utils import ClipPointsLinf

    class WeightRecorder(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner, self.weights = inner, []

        def forward(self, adv, ori, weights=None, batch_avg=True):
            self.weights.append(np.asarray(weights.detach().numpy(), dtype=np.float64).copy())
            return self.inner(adv, ori, weights, batch_avg)

    model, sha = _seeded_pointnet(PointNetCls, 40, 0)
    trans_model, _ = _seeded_pointnet(PointNetCls, 40, 1)
    fx = {"sha256": np.array(sha)}
    STEPS, ITERS, N = 10, 100, 1024
    cases = {"untarget_a": dict(method="untarget", kappa=2., cloud=4101, seed=1000),
             "untarget_b": dict(method="untarget", kappa=8., cloud=4102, seed=1001),
             "target_a": dict(method="target", kappa=0., cloud=4103, seed=1002)}
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        c = cases[nm]
        pc = unit_cloud(np.random.default_rng(c["cloud"]), N)[None]
        x = torch.from_numpy(pc).transpose(1, 2).contiguous()
        with torch.no_grad():
            logp = model(x)[0]
        clean = int(torch.argmax(logp, dim=1))
        tgt = clean if c["method"] == "untarget" else int(torch.topk(logp, 2)[1][0, 1])
        adv_func = UntargetedLogitsAdvLoss(kappa=c["kappa"]) if c["method"] == "untarget" else LogitsAdvLoss(kappa=c["kappa"])
        rec = WeightRecorder(L2Dist())
        atk = CW(model, trans_model, adv_func=adv_func, clip_func=ClipPointsLinf(budget=0.18), dist_func=rec,
                 attack_lr=1e-2, init_weight=10., max_weight=80., binary_step=STEPS, num_iter=ITERS,
                 attack_method=c["method"])
        torch.manual_seed(c["seed"])
        np.random.seed(c["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            bd, ba, sn = atk.attack(torch.from_numpy(pc), torch.tensor([tgt]))
        W = np.stack(rec.weights)                                  # [STEPS * ITERS, B]
        assert W.shape[0] == STEPS * ITERS
        with torch.no_grad():
            adv_lab = int(torch.argmax(model(torch.from_numpy(ba).float().transpose(1, 2).contiguous())[0], dim=1))
        fx[f"{nm}_pc"], fx[f"{nm}_target"], fx[f"{nm}_clean"] = pc, np.array([tgt]), np.array([clean])
        fx[f"{nm}_cfg"] = np.array([STEPS, ITERS, c["kappa"], c["seed"]])
        fx[f"{nm}_bestdist"], fx[f"{nm}_bestattack"], fx[f"{nm}_success"] = bd, ba.astype(np.float32), np.array(sn)
        fx[f"{nm}_weights"] = W[::ITERS].copy()                    # the weight of every binary step [STEPS, B]
        fx[f"{nm}_fails"] = np.array([atk.attack_fail, atk.shuffle_fail, atk.trans_fail])
        fx[f"{nm}_adv_label"] = np.array([adv_lab])
        print(nm, "bestdist", bd, "success", sn, "weights", W[::ITERS, 0], "fails", fx[f"{nm}_fails"], "adv label", adv_lab,
              "clean", clean, flush=True)
    np.savez_compressed(os.path.join(OUT, "cw_full.npz"), **fx)
    print("cw_full.npz:", len(fx), "arrays")


def gen_knn_full():
    """The REAL reference's CWKNN.attack on its PointNet++ SSG at N=1024 (B=1), ChamferDist + ProjectInnerClipLinf(0.18),
    lr 1e-2 (attack/KNN/Eval_KNN.py:139,243-244), 100 iterations. The victim draws an FPS start index from torch's global
    CPU generator in every set-abstraction layer of every forward (model/pointnet2_utils.py:72): the seed below fixes that
    stream, and the mirror consumes it in the same order."""
    install_cpu_shim()
    import contextlib
    import io
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict, state_sha256
    from model.pointnet import PointNetCls
    from model.pointnet2_SSG import PointNet_Ssg
    from attack.KNN.KNN_attack import CWKNN
    from attack.CW.CW_utils.adv_utils import UntargetedLogitsAdvLoss
    from attack.CW.CW_utils.dist_utils import ChamferDist
    from attack.CW.CW_utils.clip_utils import ProjectInnerClipLinf

    ssg = PointNet_Ssg(num_classes=40)
    sd = seeded_state_dict(ssg, 3)
    ssg.load_state_dict(sd)
    ssg.eval()
    pn = PointNetCls(k=40, feature_transform=False)
    pn.load_state_dict(seeded_state_dict(pn, 0))
    pn.eval()
    fx = {"sha256": np.array(state_sha256(sd))}
    # (On this seeded random-init victim the logit margin is a few hundredths and the Chamfer term, multiplied by K as in
    # the reference, dominates: the decisive outcome of a run is "not adversarial". The two cases below were picked among
    # several because their final margin stays > 0.025 under eight different FPS start draws — a success flag that does
    # not hinge on rounding. What they pin: the RNG stream order, the clean prediction, the fail counters, the cloud.)
    N, ITERS = 1024, 100
    cases = {"a": dict(cloud=5102, seed=2001, kappa=0.), "b": dict(cloud=5104, seed=2003, kappa=0.)}
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        c = cases[nm]
        pc = unit_cloud(np.random.default_rng(c["cloud"]), N)[None]
        torch.manual_seed(50)
        with torch.no_grad():
            clean = int(torch.argmax(ssg(torch.from_numpy(pc).transpose(1, 2).contiguous())[0], dim=1))
        atk = CWKNN(ssg, pn, pn, pn, pn, pn, adv_func=UntargetedLogitsAdvLoss(kappa=c["kappa"]), dist_func=ChamferDist(),
                    clip_func=ProjectInnerClipLinf(budget=0.18), attack_lr=1e-2, num_iter=ITERS, attack_method='untarget')
        torch.manual_seed(c["seed"])
        np.random.seed(c["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            adv, sn = atk.attack(torch.from_numpy(pc), torch.tensor([clean]))
        fx[f"{nm}_pc"], fx[f"{nm}_target"] = pc, np.array([clean])
        fx[f"{nm}_cfg"] = np.array([ITERS, 1e-2, c["kappa"], c["seed"]])
        fx[f"{nm}_adv"], fx[f"{nm}_success"] = adv.astype(np.float32), np.array(sn)
        fx[f"{nm}_fails"] = np.array([atk.attack_fail, atk.pt_fail])
        print(nm, "success", sn, "fails", fx[f"{nm}_fails"], "moved", float(np.abs(adv - pc).max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "knn_full.npz"), **fx)
    print("knn_full.npz:", len(fx), "arrays")


def gen_pointnet2():
    """PointNet++ ops (FPS / ball query / grouping) and the SSG / MSG classifiers of the reference on seeded inputs.
    The reference draws the FPS start index from the global CPU generator (pointnet2_utils.py:72): every call below is
    preceded by torch.manual_seed so the tests can replay the same stream."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict, state_sha256
    from model import pointnet2_utils as pu
    from model.pointnet2_SSG import PointNet_Ssg
    from model.pointnet2_MSG import PointNet_Msg
    rng = np.random.default_rng(2468)
    fx = {}
    real = load_real_clouds()
    face = real.get("face0424")
    clouds = {"rand_b3_n1024": np.stack([unit_cloud(rng, 1024) for _ in range(3)]),
              "rand_b2_n300": np.stack([unit_cloud(rng, 300) for _ in range(2)])}
    if face is not None:
        sub = face[:: max(1, face.shape[0] // 2048)][:2048].astype(np.float64)
        sub = sub - sub.mean(0, keepdims=True)
        sub = (sub / np.max(np.linalg.norm(sub, axis=1))).astype(np.float32)
        clouds["face_b1_n2048"] = sub[None]
    fx["names"] = np.array(sorted(clouds))
    for nm in sorted(clouds):
        xyz = torch.from_numpy(clouds[nm])
        B, N, _ = xyz.shape
        S, r, ns = min(128, N // 2), 0.25, 16
        fx[f"{nm}_xyz"] = clouds[nm]
        fx[f"{nm}_cfg"] = np.array([S, r, ns])
        torch.manual_seed(11)
        fps = pu.farthest_point_sample(xyz, S)
        fx[f"{nm}_fps"] = fps.numpy()
        new_xyz = pu.index_points(xyz, fps)
        idx = pu.query_ball_point(r, ns, xyz, new_xyz)
        fx[f"{nm}_ball"] = idx.numpy()
        # distance of every point to the ball surface, so tests can tolerate the expansion's rounding at the rim
        d = ((new_xyz[:, :, None, :].double() - xyz[:, None, :, :].double()) ** 2).sum(-1)
        fx[f"{nm}_rim"] = np.abs(d.numpy() - r * r).min(axis=2).astype(np.float32)
        feats = torch.from_numpy(rng.standard_normal((B, N, 5)).astype(np.float32))
        fx[f"{nm}_feats"] = feats.numpy()
        torch.manual_seed(11)
        nx, npts = pu.sample_and_group(S, r, ns, xyz, feats)
        fx[f"{nm}_sg_new_xyz"], fx[f"{nm}_sg_new_points"] = nx.numpy(), npts.numpy()
    # classifiers
    for cname, cls, kw in (("ssg", PointNet_Ssg, dict(num_classes=40)), ("msg", PointNet_Msg, dict(num_class=40, normal_channel=False))):
        m = cls(**kw)
        sd = seeded_state_dict(m, 3)
        m.load_state_dict(sd)
        m.eval()
        fx[f"{cname}_sha256"] = np.array(state_sha256(sd))
        x = np.stack([unit_cloud(rng, 1024) for _ in range(2)]).transpose(0, 2, 1).copy()
        tx = torch.from_numpy(x).requires_grad_()
        torch.manual_seed(21)
        logp = m(tx)[0]
        w = torch.from_numpy(rng.standard_normal(logp.shape).astype(np.float32))
        (logp * w).sum().backward()
        fx[f"{cname}_x"], fx[f"{cname}_logp"], fx[f"{cname}_w"], fx[f"{cname}_gx"] = x, logp.detach().numpy(), w.numpy(), tx.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "pointnet2.npz"), **fx)
    print("pointnet2.npz:", len(fx), "arrays")


def gen_knn():
    """Short runs of the REAL reference CWKNN.attack (B=1): PointNet victim (deterministic) with ChamferkNNDist +
    ProjectInnerClipLinf, and a PointNet++ SSG victim with ChamferDist (Eval_KNN.py:243-244 wiring)."""
    install_cpu_shim()
    import contextlib
    import io
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict
    from model.pointnet import PointNetCls
    from model.pointnet2_SSG import PointNet_Ssg
    from attack.KNN.KNN_attack import CWKNN
    from attack.CW.CW_utils.adv_utils import UntargetedLogitsAdvLoss
    from attack.CW.CW_utils.dist_utils import ChamferkNNDist, ChamferDist
    from attack.CW.CW_utils.clip_utils import ProjectInnerClipLinf

    def seeded(cls, seed, **kw):
        m = cls(**kw)
        m.load_state_dict(seeded_state_dict(m, seed))
        return m.eval()

    rng = np.random.default_rng(999)
    pn = seeded(PointNetCls, 0, k=40, feature_transform=False)
    ssg = seeded(PointNet_Ssg, 3, num_classes=40)
    fx = {}
    cases = {"pointnet_chamferknn": dict(victim=pn, dist=ChamferkNNDist(), N=256, iters=20, lr=1e-2, kappa=5.),
             "ssg_chamfer": dict(victim=ssg, dist=ChamferDist(), N=600, iters=6, lr=1e-2, kappa=5.)}
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        c = cases[nm]
        pc = unit_cloud(rng, c["N"])[None]
        torch.manual_seed(50)
        with torch.no_grad():
            clean = int(torch.argmax(c["victim"](torch.from_numpy(pc).transpose(1, 2).contiguous())[0], dim=1))
        atk = CWKNN(c["victim"], pn, pn, pn, pn, pn, adv_func=UntargetedLogitsAdvLoss(kappa=c["kappa"]),
                    dist_func=c["dist"], clip_func=ProjectInnerClipLinf(budget=0.18), attack_lr=c["lr"],
                    num_iter=c["iters"], attack_method='untarget')
        torch.manual_seed(1000)
        np.random.seed(1000)
        with contextlib.redirect_stdout(io.StringIO()):
            adv, sn = atk.attack(torch.from_numpy(pc), torch.tensor([clean]))
        fx[f"{nm}_pc"], fx[f"{nm}_target"] = pc, np.array([clean])
        fx[f"{nm}_cfg"] = np.array([c["iters"], c["lr"], c["kappa"]])
        fx[f"{nm}_adv"], fx[f"{nm}_success"] = adv.astype(np.float32), np.array(sn)
        fx[f"{nm}_fails"] = np.array([atk.attack_fail, atk.pt_fail])
    np.savez_compressed(os.path.join(OUT, "knn.npz"), **fx)
    print("knn.npz:", len(fx), "arrays")


def gen_dgcnn():
    """Reference DGCNN: kNN indices (xyz + feature space), dense graph feature, logits and input gradient."""
    install_cpu_shim()
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict, state_sha256
    from model import dgcnn as rd
    rng = np.random.default_rng(1357)
    fx = {}
    args = types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5)
    m = rd.DGCNN(args, output_channels=40)
    sd = seeded_state_dict(m, 5)
    m.load_state_dict(sd)
    m.eval()
    fx["sha256"] = np.array(state_sha256(sd))
    x = np.stack([unit_cloud(rng, 512) for _ in range(2)]).transpose(0, 2, 1).copy()
    tx = torch.from_numpy(x).requires_grad_()
    logp = m(tx)[0]
    w = torch.from_numpy(rng.standard_normal(logp.shape).astype(np.float32))
    (logp * w).sum().backward()
    fx["x"], fx["logp"], fx["w"], fx["gx"] = x, logp.detach().numpy(), w.numpy(), tx.grad.numpy()
    # primitives
    feat = rng.standard_normal((2, 64, 300)).astype(np.float32)
    fx["feat"] = feat
    fx["feat_knn"] = rd.knn(torch.from_numpy(feat), 20).numpy()
    fx["xyz_knn"] = rd.knn(torch.from_numpy(x), 20).numpy()
    fx["graph_feature"] = rd.get_graph_feature(torch.from_numpy(x[:, :, :64].copy()), k=8).numpy()
    np.savez_compressed(os.path.join(OUT, "dgcnn.npz"), **fx)
    print("dgcnn.npz:", len(fx), "arrays")


def gen_geoa3():
    """The REAL reference geoA3_attack on a seeded PointNet victim (B=1). Importing attack.GeoA3.* needs modules this
    container lacks (open3d, torchvision, seaborn) and two removed torch APIs; empty stand-in modules satisfy the
    imports — none of them is called on the attack path. The transfer models are dummy modules, which
    avoids the reference's post-loop crash on real ones (it feeds them [B,N,3], SURVEY A-6)."""
    install_cpu_shim()
    import contextlib
    import io
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict
    for name in ("open3d", "torchvision", "torchvision.transforms", "seaborn"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["seaborn"].set = lambda *a, **k: None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import importlib
    importlib.import_module("torch.autograd.gradcheck")
    sys.modules["torch.autograd.gradcheck"].zero_gradients = lambda *a, **k: None
    torch.symeig = lambda L, eigenvectors=True: torch.linalg.eigh(L)
    os.popen = lambda *a, **k: io.StringIO("24 80")
    sys.path.insert(0, os.path.join(REF, "attack", "GeoA3"))
    from model.pointnet import PointNetCls
    from attack.GeoA3 import GeoA3_attack as ga
    from attack.GeoA3 import loss_utils as lu
    from attack.GeoA3.utility import estimate_normal

    class Dummy(torch.nn.Module):
        def forward(self, x):
            z = torch.zeros(x.shape[0], 40)
            return z, z, z

    net = PointNetCls(k=40, feature_transform=False)
    net.load_state_dict(seeded_state_dict(net, 0))
    net.eval()
    rng = np.random.default_rng(8642)
    fx = {}
    base = dict(attack_method='untarget', curv_loss_weight=1.0, curv_loss_knn=16, initial_const=10, iter_max_steps=12,
                binary_max_steps=2, is_partial_var=False, optim='adam', lr=0.01, npoint=256, is_subsample_opt=False,
                eval_num=1, is_pre_jitter_input=False, cls_loss_type='CE', classes=40, confidence=0, dis_loss_type='CD',
                is_cd_single_side=False, dis_loss_weight=1.0, hd_loss_weight=0.1, uniform_loss_weight=0.0,
                is_use_lr_scheduler=False, is_debug=False, is_pro_grad=False, cc_linf=0.0, binary_step=2, num_iter=12,
                is_real_offset=False, knn_range=3)
    cases = {"ce_cd_hd_curv": {}, "margin_l2": dict(cls_loss_type='Margin', confidence=5., dis_loss_type='L2', hd_loss_weight=0,
                                                    curv_loss_weight=0)}
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        cfg = types.SimpleNamespace(**{**base, **cases[nm]})
        pc = unit_cloud(rng, 256)[None]
        with torch.no_grad():
            clean = int(torch.argmax(net(torch.from_numpy(pc).transpose(1, 2).contiguous())[0], dim=1))
        torch.manual_seed(77)
        np.random.seed(77)
        with contextlib.redirect_stdout(io.StringIO()):
            best, tgt, mask, steps, losses = ga.geoA3_attack(net, Dummy(), Dummy(), Dummy(), Dummy(), Dummy(),
                                                             torch.from_numpy(pc), torch.tensor([clean]), cfg, 0, 1)
        fx[f"{nm}_pc"], fx[f"{nm}_label"] = pc, np.array([clean])
        fx[f"{nm}_best"], fx[f"{nm}_mask"] = best.detach().numpy(), np.asarray(mask)
        fx[f"{nm}_steps"], fx[f"{nm}_losses"] = np.array(steps), np.array(losses, dtype=np.float64)
    # one _forward_step worth of loss terms + the normal estimate, for unit-level checks
    pc = torch.from_numpy(unit_cloud(rng, 300)[None]).transpose(1, 2).contiguous()
    adv = pc + 0.01 * torch.randn(pc.shape, generator=torch.Generator().manual_seed(5))
    normal = estimate_normal(pc, k=3)
    fx["unit_pc"], fx["unit_adv"], fx["unit_normal"] = pc.numpy(), adv.numpy(), normal.numpy()
    kappa_ori = lu._get_kappa_ori(pc, normal, 16)
    ak, nrm = lu._get_kappa_adv(adv, pc, normal, 16)
    fx["unit_kappa_ori"], fx["unit_kappa_adv"] = kappa_ori.numpy(), ak.numpy()
    fx["unit_terms"] = np.array([float(lu.chamfer_loss(adv, pc)), float(lu.pseudo_chamfer_loss(adv, pc)),
                                 float(lu.hausdorff_loss(adv, pc)), float(lu.curvature_loss(adv, pc, ak, kappa_ori)),
                                 float(lu.kNN_smoothing_loss(adv, 5, 1.1)), float(lu.norm_l2_loss(adv, pc))])
    np.savez_compressed(os.path.join(OUT, "geoa3.npz"), **fx)
    print("geoa3.npz:", len(fx), "arrays")


def gen_aof():
    """REAL reference AOF pieces: the graph Laplacian (+ its spectrum) and a short CWTAOF.attack run."""
    install_cpu_shim()
    import contextlib
    import io
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict
    torch.symeig = lambda L, eigenvectors=True: torch.linalg.eigh(L)
    from model.pointnet import PointNetCls
    from attack.AOF import TAOF_attack as ta
    from attack.CW.CW_utils.adv_utils import LogitsAdvLoss
    from attack.CW.CW_utils.dist_utils import L2Dist
    from attack.CW.CW_utils.clip_utils import ClipPointsLinf
    rng = np.random.default_rng(97531)
    fx = {}
    pc = np.stack([unit_cloud(rng, 200) for _ in range(2)]).transpose(0, 2, 1).copy()
    e, v = ta.get_Laplace_from_pc(torch.from_numpy(pc))
    fx["lap_pc"], fx["lap_eig"] = pc, e.numpy()
    fx["lap_knn"] = ta.knn(torch.from_numpy(pc), 30).numpy()
    net = PointNetCls(k=40, feature_transform=False)
    net.load_state_dict(seeded_state_dict(net, 0))
    net.eval()
    cloud = unit_cloud(rng, 256)[None]
    with torch.no_grad():
        lp = net(torch.from_numpy(cloud).transpose(1, 2).contiguous())[0]
    y_truth = int(lp.argmax(1))
    tgt = int(lp.topk(2)[1][0, 1])
    atk = ta.CWTAOF(net, LogitsAdvLoss(kappa=0.), L2Dist(), attack_lr=1e-2, binary_step=2, num_iter=10, GAMMA=0.5,
                    low_pass=40, clip_func=ClipPointsLinf(budget=0.18))
    torch.manual_seed(31)
    np.random.seed(31)
    with contextlib.redirect_stdout(io.StringIO()):
        bd, adv, sn = atk.attack(torch.from_numpy(cloud), torch.tensor([tgt]), torch.tensor([y_truth]))
    fx["atk_pc"], fx["atk_target"], fx["atk_ytruth"] = cloud, np.array([tgt]), np.array([y_truth])
    fx["atk_bestdist"], fx["atk_adv"], fx["atk_success"] = bd, adv.astype(np.float32), np.array(sn)
    np.savez_compressed(os.path.join(OUT, "aof.npz"), **fx)
    print("aof.npz:", len(fx), "arrays")


def gen_curvenet():
    """Reference CurveNet (model/curvenet.py) on seeded weights: logits and input gradient at N=1024 and N=2048 (the
    latter exercises the FPS + ball-query max-pool of the first block)."""
    install_cpu_shim()
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict, state_sha256
    from model.curvenet import CurveNet
    rng = np.random.default_rng(86420)
    m = CurveNet(num_classes=40)
    sd = seeded_state_dict(m, 9, gain=1.0)
    m.load_state_dict(sd)
    m.eval()
    fx = {"sha256": np.array(state_sha256(sd))}
    for nm, (B, N) in {"n1024": (2, 1024), "n2048": (1, 2048)}.items():
        x = np.stack([unit_cloud(rng, N) for _ in range(B)]).transpose(0, 2, 1).copy()
        tx = torch.from_numpy(x).requires_grad_()
        out = m(tx)[0]
        w = torch.from_numpy(rng.standard_normal(out.shape).astype(np.float32))
        (out * w).sum().backward()
        fx[f"{nm}_x"], fx[f"{nm}_logits"], fx[f"{nm}_w"], fx[f"{nm}_gx"] = x, out.detach().numpy(), w.numpy(), tx.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "curvenet.npz"), **fx)
    print("curvenet.npz:", len(fx), "arrays")


def gen_curvenet_blocks():
    """The reference's Walk (model/walk.py), CurveAggregation and LPFA (model/curvenet_util.py) run as stand-alone
    modules on NON-trivial weights and BatchNorm statistics (the whole-network fixture above uses a near-identity
    initialisation under which the walk's momentum path barely matters): inputs, state_dicts, outputs and input
    gradients. Pins the K16 / K17 / K18 kernels directly against the reference."""
    install_cpu_shim()
    from model.walk import Walk
    from model import curvenet_util as cu
    g = torch.Generator().manual_seed(97531)
    rng = np.random.default_rng(97531)

    def randomise(mod):
        with torch.no_grad():
            for name, p in mod.named_parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g) / float(p[0].numel()) ** 0.5)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.2 + (1.0 if name.endswith("weight") else 0.0))
            for name, b in mod.named_buffers():
                if name.endswith("running_mean"):
                    b.copy_(torch.randn(b.shape, generator=g) * 0.1)
                elif name.endswith("running_var"):
                    b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        return mod.eval()

    def save_sd(fx, prefix, mod):
        for k, v in mod.state_dict().items():
            fx[f"{prefix}.sd.{k}"] = v.numpy()

    fx = {}
    B, N, C, k, cn, cl = 2, 256, 16, 20, 100, 5
    xyz = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]).transpose(0, 2, 1).copy())     # [B,3,N]
    idx = cu.knn(xyz, k)                                                                                  # [B,N,k+1]
    fx["xyz"], fx["idx"] = xyz.numpy(), idx.numpy()
    # --- Walk
    walk = randomise(Walk(C, k, cn, cl))
    x = torch.randn(B, C, N, generator=g).requires_grad_()
    start = torch.stack([torch.randperm(N, generator=g)[:cn] for _ in range(B)]).unsqueeze(2)              # [B,cn,1]
    curves = walk(xyz, x, idx[:, :, 1:], start)
    w = torch.randn(curves.shape, generator=g)
    (curves * w).sum().backward()
    fx["walk.x"], fx["walk.start"], fx["walk.curves"], fx["walk.w"], fx["walk.gx"] = (
        x.detach().numpy(), start.numpy(), curves.detach().numpy(), w.numpy(), x.grad.numpy())
    save_sd(fx, "walk", walk)
    # --- CurveAggregation
    agg = randomise(cu.CurveAggregation(C))
    x = torch.randn(B, C, N, generator=g).requires_grad_()
    cv = torch.randn(B, C, cn, cl, generator=g).requires_grad_()
    y = agg(x, cv)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    fx["agg.x"], fx["agg.curves"], fx["agg.y"], fx["agg.w"], fx["agg.gx"], fx["agg.gcurves"] = (
        x.detach().numpy(), cv.detach().numpy(), y.detach().numpy(), w.numpy(), x.grad.numpy(), cv.grad.numpy())
    save_sd(fx, "agg", agg)
    # --- LPFA: first layer (initial=True, features = the points) and an inner layer
    for tag, mod, feat in (("lpfa0", randomise(cu.LPFA(9, 32, k, mlp_num=1, initial=True)), None),
                           ("lpfa1", randomise(cu.LPFA(C, C, k, mlp_num=1, initial=False)), torch.randn(B, C, N, generator=g))):
        pts = xyz.clone().requires_grad_()
        f = pts if feat is None else feat.clone().requires_grad_()
        y = mod(f, pts, idx=idx[:, :, :k])
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        fx[f"{tag}.y"], fx[f"{tag}.w"], fx[f"{tag}.gxyz"] = y.detach().numpy(), w.numpy(), pts.grad.numpy()
        if feat is not None:
            fx[f"{tag}.x"], fx[f"{tag}.gx"] = feat.numpy(), f.grad.numpy()
        save_sd(fx, tag, mod)
    np.savez_compressed(os.path.join(OUT, "curvenet_blocks.npz"), **fx)
    print("curvenet_blocks.npz:", len(fx), "arrays")


def gen_cw_additional():
    """Short runs of the REAL attack/additional_exp/CW_attack.py (B=1) on seeded PointNet weights. No functor of the
    reference accepts the `whether_target` keyword this attack passes (:245-259), so the adversarial functors are
    wrapped; the distance functors are the reference's own, called with [B,K,3] as that file does (:151-153)."""
    install_cpu_shim()
    import contextlib
    import io
    import random
    from model.pointnet import PointNetCls
    from attack.additional_exp.CW_attack import CW
    from attack.CW.CW_utils.adv_utils import UntargetedLogitsAdvLoss, LogitsAdvLoss
    from attack.CW.CW_utils.dist_utils import L2Dist, ChamferDist, ChamferkNNDist

    class Adv:
        def __init__(self, kappa):
            self.t, self.u = LogitsAdvLoss(kappa=kappa), UntargetedLogitsAdvLoss(kappa=kappa)

        def __call__(self, logits, label, whether_target=1):
            return (self.t if whether_target else self.u)(logits, label)

    class Recorder(torch.nn.Module):
        def __init__(self, inner, per_sample):
            super().__init__()
            self.inner, self.per_sample, self.log = inner, per_sample, []

        def forward(self, adv, ori, weights=None):
            self.log.append(adv.detach().numpy().copy())
            return self.inner(adv, ori, weights, batch_avg=not self.per_sample)

    rng = np.random.default_rng(2024)
    model, sha = _seeded_pointnet(PointNetCls, 40, 0)
    fx = {"sha256": np.array(sha)}
    cases = {
        "z_chamfer_target": dict(dist="chamfer", target=True, d1=True, renorm=False, eot=False, resample=False, N=200, steps=2, iters=12, kappa=0.),
        "eot_renorm_chamferknn_target": dict(dist="chamferknn", target=True, d1=True, renorm=True, eot=True, resample=False, N=160, steps=2, iters=6, kappa=0.),
        "free_l2_untarget": dict(dist="l2", target=False, d1=False, renorm=False, eot=False, resample=False, N=256, steps=3, iters=10, kappa=5.),
        "resample_chamfer_target": dict(dist="chamfer", target=True, d1=True, renorm=False, eot=True, resample=True, N=4000, steps=1, iters=2, kappa=0.),
    }
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        c = cases[nm]
        pc = unit_cloud(rng, c["N"])[None]
        with torch.no_grad():
            lg = model(torch.from_numpy(pc).transpose(1, 2).contiguous())[0]
        clean, runner_up = int(torch.argmax(lg, dim=1)), int(torch.topk(lg, 2)[1][0, 1])
        inner = {"chamfer": ChamferDist(), "chamferknn": ChamferkNNDist(), "l2": L2Dist()}[c["dist"]]
        # L2Dist takes channel-first clouds but is rotation-agnostic here (it only sums squares); the untargeted
        # branch iterates over dist_val (:172-174), so it needs the per-sample [B] form
        rec = Recorder(inner, per_sample=not c["target"])
        atk = CW(model, Adv(c["kappa"]), rec, attack_lr=1e-2, init_weight=10., max_weight=80., binary_step=c["steps"],
                 num_iter=c["iters"], whether_target=c["target"], whether_1d=c["d1"], whether_renormalization=c["renorm"],
                 whether_3Dtransform=c["eot"], whether_resample=c["resample"])
        torch.manual_seed(2000)
        random.seed(2000)
        np.random.seed(2000)
        with contextlib.redirect_stdout(io.StringIO()):
            bd, ba, sn = atk.attack(torch.from_numpy(pc), torch.tensor([runner_up]), torch.tensor([clean]))
        fx[f"{nm}_pc"], fx[f"{nm}_target"], fx[f"{nm}_origin"] = pc, np.array([runner_up]), np.array([clean])
        fx[f"{nm}_bestdist"], fx[f"{nm}_bestattack"], fx[f"{nm}_success"] = np.asarray(bd), ba.astype(np.float32), np.array(int(sn))
        traj = np.stack(rec.log).astype(np.float32)[:, 0]                      # [steps*iters, K, 3]
        fx[f"{nm}_traj"] = traj if c["N"] <= 512 else traj[:, ::16]
    np.savez_compressed(os.path.join(OUT, "cw_additional.npz"), **fx)
    print("cw_additional.npz:", len(fx), "arrays")


def gen_formats():
    """On-disk formats through the reference's own dataset code, on two data files the reference
    ships (copied to tests/golden/data/ as plain data): Bosphorus_Dataset.__getitem__'s csv-text branch
    (dataset/bosphorus_dataset.py:59-84) and AdvData_dataset.read_PC (:21-38). open3d (absent) is stubbed as an
    empty module and `np.float` (removed from numpy) restored for the import, as SURVEY §8(c) describes."""
    import types
    import pandas as pd
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    if not hasattr(np, "float"):
        np.float = float
    sys.path.insert(0, os.path.join(os.environ.get("PC3D_REFERENCE", "/root/reference"), "utils"))   # `from readbnt import ...`
    from dataset.bosphorus_dataset import Bosphorus_Dataset
    from dataset.AdvData_dataset import read_PC
    ref = os.environ.get("PC3D_REFERENCE", "/root/reference")
    fx = {}
    ds = object.__new__(Bosphorus_Dataset)                   # skip __init__ (needs the private csv index)
    ds.df = pd.DataFrame([[os.path.join(ref, "AddData", "face0424.txt"), 105]])
    np.random.seed(4242)
    pc, cls = ds[0]
    fx["bosphorus_pc"], fx["bosphorus_cls"] = pc.numpy(), cls.numpy()
    A, ori, tar = read_PC(0, os.path.join(ref, "attack", "CW", "AdvData", "PointNet"))
    fx["advdata_A"], fx["advdata_ori_tar"] = A, np.array([ori, tar])
    np.savez_compressed(os.path.join(OUT, "formats.npz"), **fx)
    print("formats.npz:", len(fx), "arrays")


def _geoa3_imports():
    """Stand-in modules / removed torch APIs the GeoA3 sources import but never call on the attack path (see
    gen_geoa3); returns (GeoA3_attack module, loss_utils, estimate_normal)."""
    import io
    import types
    import importlib
    for name in ("open3d", "torchvision", "torchvision.transforms", "seaborn"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["seaborn"].set = lambda *a, **k: None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    importlib.import_module("torch.autograd.gradcheck")
    sys.modules["torch.autograd.gradcheck"].zero_gradients = lambda *a, **k: None
    torch.symeig = lambda L, eigenvectors=True: torch.linalg.eigh(L)
    os.popen = lambda *a, **k: io.StringIO("24 80")
    sys.path.insert(0, os.path.join(REF, "attack", "GeoA3"))
    from attack.GeoA3 import GeoA3_attack as ga
    from attack.GeoA3 import loss_utils as lu
    from attack.GeoA3.utility import estimate_normal
    return ga, lu, estimate_normal


GEO_BASE = dict(attack_method='untarget', curv_loss_weight=1.0, curv_loss_knn=16, initial_const=10, iter_max_steps=12,
                binary_max_steps=2, is_partial_var=False, optim='adam', lr=0.01, npoint=256, is_subsample_opt=False,
                eval_num=1, is_pre_jitter_input=False, cls_loss_type='CE', classes=40, confidence=0, dis_loss_type='CD',
                is_cd_single_side=False, dis_loss_weight=1.0, hd_loss_weight=0.1, uniform_loss_weight=0.0,
                is_use_lr_scheduler=False, is_debug=False, is_pro_grad=False, cc_linf=0.0, binary_step=2, num_iter=12,
                is_real_offset=False, knn_range=3)


def gen_geoa3_dgcnn():
    """BASELINE configs[2] as a workload: the REAL reference geoA3_attack on the REAL reference DGCNN (B=1 as the
    reference requires, N=256, the Eval_GeoA3.py:154-164 default loss mix and a Margin/L2 variant). The loss curves
    [binary-step-2 iterations][B] and the result are stored; the offsets come from torch's CPU generator (seed 77)."""
    install_cpu_shim()
    import contextlib
    import io
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict, state_sha256
    ga, lu, estimate_normal = _geoa3_imports()
    from model import dgcnn as rd

    class Dummy(torch.nn.Module):
        def forward(self, x):
            z = torch.zeros(x.shape[0], 40)
            return z, z, z

    net = rd.DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
    sd = seeded_state_dict(net, 5)
    net.load_state_dict(sd)
    net.eval()
    rng = np.random.default_rng(24680)
    fx = {"sha256": np.array(state_sha256(sd))}
    cases = {"ce_cd_hd_curv": {}, "margin_l2": dict(cls_loss_type='Margin', confidence=5., dis_loss_type='L2', hd_loss_weight=0,
                                                    curv_loss_weight=0)}
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        cfg = types.SimpleNamespace(**{**GEO_BASE, **cases[nm]})
        pc = unit_cloud(rng, 256)[None]
        with torch.no_grad():
            clean = int(torch.argmax(net(torch.from_numpy(pc).transpose(1, 2).contiguous())[0], dim=1))
        torch.manual_seed(77)
        np.random.seed(77)
        with contextlib.redirect_stdout(io.StringIO()):
            best, tgt, mask, steps, losses = ga.geoA3_attack(net, Dummy(), Dummy(), Dummy(), Dummy(), Dummy(),
                                                             torch.from_numpy(pc), torch.tensor([clean]), cfg, 0, 1)
        fx[f"{nm}_pc"], fx[f"{nm}_label"] = pc, np.array([clean])
        fx[f"{nm}_best"], fx[f"{nm}_mask"] = best.detach().numpy(), np.asarray(mask)
        fx[f"{nm}_steps"], fx[f"{nm}_losses"] = np.array(steps), np.array(losses, dtype=np.float64)
        print(nm, "mask", mask, "steps", steps, "loss[0], loss[-1]", losses[0], losses[-1])
    np.savez_compressed(os.path.join(OUT, "geoa3_dgcnn.npz"), **fx)
    print("geoa3_dgcnn.npz:", len(fx), "arrays")


def gen_cw_curvenet():
    """BASELINE configs[4]'s victim under attack: the REAL reference CW.attack (B=1) on the REAL reference CurveNet
    (N=1024, seeded weights), L2 and Chamfer distance functors; trajectory through the dist_func hook."""
    install_cpu_shim()
    canonical_unsorted_topk()
    import contextlib
    import io
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict, state_sha256
    from model.curvenet import CurveNet
    from attack.CW.CW_attack import CW
    from attack.CW.CW_utils.adv_utils import UntargetedLogitsAdvLoss
    from attack.CW.CW_utils.dist_utils import L2Dist, ChamferDist
    from attack.CW.CW_utils.clip_utils import ClipPointsLinf

    class Recorder(torch.nn.Module):
        def __init__(self, inner, transpose):
            super().__init__()
            self.inner, self.transpose, self.log = inner, transpose, []

        def forward(self, adv, ori, weights=None, batch_avg=True):
            self.log.append(adv.detach().numpy().copy())
            if self.transpose:
                return self.inner(adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous(), weights, batch_avg)
            return self.inner(adv, ori, weights, batch_avg)

    # Seeded random CurveNet weights give a bias-dominated classifier (every cloud -> the same class, margin ~2): labels
    # would say nothing. The last layer's bias is therefore re-centred on the mean clean logits of six probe clouds —
    # margins become 0.002-0.23 and the clean labels differ per cloud. The bias vector is stored in the fixture; the
    # tests rebuild the weights as seeded_state_dict(seed 9, gain 2.45) + this bias.
    model = CurveNet(num_classes=40)
    sd = seeded_state_dict(model, 9)
    model.load_state_dict(sd)
    model.eval()
    rng = np.random.default_rng(13579)
    probes = [unit_cloud(rng, 1024)[None] for _ in range(6)]
    with torch.no_grad():
        L = torch.cat([model(torch.from_numpy(p).transpose(1, 2).contiguous())[0] for p in probes])
        model.conv2.bias -= L.mean(0)
    fx = {"sha256": np.array(state_sha256(model.state_dict())), "conv2_bias": model.conv2.bias.detach().numpy().copy()}
    cases = {"l2": dict(dist="l2", kappa=0., steps=2, iters=8, probe=0),
             "chamfer": dict(dist="chamfer", kappa=0., steps=2, iters=8, probe=5)}
    fx["names"] = np.array(sorted(cases))
    for nm in sorted(cases):
        c = cases[nm]
        pc = probes[c["probe"]]
        with torch.no_grad():
            logits = model(torch.from_numpy(pc).transpose(1, 2).contiguous())[0]
        clean = int(torch.argmax(logits, dim=1))
        rec = Recorder(L2Dist() if c["dist"] == "l2" else ChamferDist(), transpose=(c["dist"] != "l2"))
        atk = CW(model, model, adv_func=UntargetedLogitsAdvLoss(kappa=c["kappa"]), clip_func=ClipPointsLinf(budget=0.18),
                 dist_func=rec, attack_lr=1e-2, init_weight=10., max_weight=80., binary_step=c["steps"], num_iter=c["iters"],
                 attack_method="untarget")
        torch.manual_seed(2000)
        np.random.seed(2000)
        with contextlib.redirect_stdout(io.StringIO()):
            bd, ba, sn = atk.attack(torch.from_numpy(pc), torch.tensor([clean]))
        fx[f"{nm}_pc"], fx[f"{nm}_target"], fx[f"{nm}_clean_logits"] = pc, np.array([clean]), logits.numpy()
        fx[f"{nm}_cfg"] = np.array([c["steps"], c["iters"], c["kappa"]])
        fx[f"{nm}_bestdist"], fx[f"{nm}_bestattack"], fx[f"{nm}_success"] = bd, ba.astype(np.float32), np.array(sn)
        fx[f"{nm}_traj"] = np.stack(rec.log).astype(np.float32)[:, 0]
        fx[f"{nm}_fails"] = np.array([atk.attack_fail, atk.shuffle_fail, atk.trans_fail])
        with torch.no_grad():
            fx[f"{nm}_adv_label"] = model(torch.from_numpy(ba).float().transpose(1, 2).contiguous())[0].argmax(1).numpy()
        print(nm, "clean", clean, "bestdist", bd, "success", sn, "fails", fx[f"{nm}_fails"], "adv label", fx[f"{nm}_adv_label"])
    np.savez_compressed(os.path.join(OUT, "cw_curvenet.npz"), **fx)
    print("cw_curvenet.npz:", len(fx), "arrays")


def gen_f4():
    """the REAL reference's add-cluster / add-object distance functors
    (attack/CW/CW_utils/dist_utils.py:226-333) — per-sample values and the gradient of the batch mean. The Chamfer part
    is evaluated in float64 (the reference's fp32 expansion is itself ~1e-5 off, SURVEY A-3); FarthestDist also in the
    reference's own fp32."""
    install_cpu_shim()
    from attack.CW.CW_utils.dist_utils import FarthestDist, FarChamferDist, L2ChamferDist
    rng = np.random.default_rng(112233)
    fx = {}
    B, K, na, cp = 3, 300, 3, 32
    ori = np.stack([unit_cloud(rng, K) for _ in range(B)])
    centres = ori[:, rng.choice(K, na, replace=False)]                                   # clusters near the surface
    adv = (centres[:, :, None, :] + 0.05 * rng.standard_normal((B, na, cp, 3))).astype(np.float32)
    w = (0.5 + rng.random(B)).astype(np.float32)
    fx["ori"], fx["adv_clusters"], fx["weights"] = ori, adv, w
    tw = torch.from_numpy(w)
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        ta = torch.from_numpy(adv).to(dt).requires_grad_()
        per = FarthestDist()(ta, weights=tw, batch_avg=False)
        FarthestDist()(ta, weights=tw, batch_avg=True).backward()
        fx[f"far_{tag}"], fx[f"far_{tag}_grad"] = per.detach().numpy(), ta.grad.numpy()
    fx["far_noweights"] = FarthestDist()(torch.from_numpy(adv).double(), batch_avg=False).numpy()
    for method in ("adv2ori", "ori2adv", "both"):
        ta = torch.from_numpy(adv.reshape(B, na * cp, 3)).double().requires_grad_()
        to = torch.from_numpy(ori).double()
        f = FarChamferDist(num_add=na, chamfer_method=method, chamfer_weight=0.1)
        per = f(ta, to, weights=tw, batch_avg=False)
        f(ta, to, weights=tw, batch_avg=True).backward()
        fx[f"farchamfer_{method}"], fx[f"farchamfer_{method}_grad"] = per.detach().numpy(), ta.grad.numpy()
    # add-object functor: objects = perturbed copies of small clean objects, placed into the scene
    obj = (0.1 * rng.standard_normal((B, na, cp, 3))).astype(np.float32)
    adv_obj = (obj + 0.01 * rng.standard_normal(obj.shape)).astype(np.float32)
    placed = (adv_obj + centres[:, :, None, :]).reshape(B, na * cp, 3).astype(np.float32)
    fx["ori_obj"], fx["adv_obj"], fx["placed"] = obj, adv_obj, placed
    tp = torch.from_numpy(placed).double().requires_grad_()
    tao = torch.from_numpy(adv_obj).double().requires_grad_()
    f = L2ChamferDist(num_add=na, chamfer_method="adv2ori", chamfer_weight=0.2)
    per = f(tp, torch.from_numpy(ori).double(), tao, torch.from_numpy(obj).double(), weights=tw, batch_avg=False)
    f(tp, torch.from_numpy(ori).double(), tao, torch.from_numpy(obj).double(), weights=tw, batch_avg=True).backward()
    fx["l2chamfer"], fx["l2chamfer_grad_placed"], fx["l2chamfer_grad_obj"] = per.detach().numpy(), tp.grad.numpy(), tao.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "f4.npz"), **fx)
    print("f4.npz:", len(fx), "arrays", fx["far_f64"], fx["farchamfer_adv2ori"], fx["l2chamfer"])


def gen_metrics_n4096():
    """SURVEY §8(c)(1) asks for a2/a3 at N in {1024, 2048, 4096}: metrics.npz stops at 2048, this adds the N=4096 pair
    (and a ragged 4096 x 3000 one) as its own small fixture so that the older fixture's random stream is untouched."""
    from utils import dis_utils_numpy as dun
    rng = np.random.default_rng(4096)
    fx = {}
    p = unit_cloud(rng, 4096)
    pairs = {"rand_4096": (perturbed(rng, p), p), "ragged_4096_3000": (unit_cloud(rng, 4096), unit_cloud(rng, 3000))}
    fx["np_names"] = np.array(sorted(pairs))
    for nm in sorted(pairs):
        a, b = pairs[nm]
        fx[f"np_{nm}_a"], fx[f"np_{nm}_b"] = a, b
        fx[f"np_{nm}_out"] = np.array([dun.chamfer(a, b), dun.sgd_hausdorff_dis(a, b),
                                       dun.sgd_hausdorff_dis(b, a), dun.bid_hausdorff_dis(a, b)], np.float64)
    np.savez_compressed(os.path.join(OUT, "metrics_n4096.npz"), **fx)
    print("metrics_n4096.npz:", {k: fx[k] for k in fx if k.endswith("_out")})


def gen_curvenet_trace():
    """Stage-by-stage outputs of the REAL reference CurveNet on Kaiming-scale seeded weights (gain 2.45; the older
    curvenet.npz uses gain 1.0, under which everything past the first blocks is bias-dominated and deep mistakes
    cannot show): LPFA, the eight CIC blocks (positions + features), conv0 and the logits, B=1, N=1024. Features are
    stored as float16 summaries would hide flips, so full fp32 for the small late stages and a strided sample of the
    channels for the early ones."""
    install_cpu_shim()
    canonical_unsorted_topk()
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.ref_torch import seeded_state_dict, state_sha256
    from model.curvenet import CurveNet
    m = CurveNet(num_classes=40)
    sd = seeded_state_dict(m, 9)
    m.load_state_dict(sd)
    m.eval()
    rng = np.random.default_rng(13579)
    x = unit_cloud(rng, 1024)[None].transpose(0, 2, 1).copy()
    fx = {"sha256": np.array(state_sha256(sd)), "x": x}
    stages = ["lpfa", "cic11", "cic12", "cic21", "cic22", "cic31", "cic32", "cic41", "cic42", "conv0"]
    hooks = []

    def keep(name):
        def fn(mod, inp, out):
            feats = out[1] if isinstance(out, tuple) else out
            f = feats.detach().numpy()
            fx[f"{name}_feat"] = f[:, ::4].copy() if f.shape[1] * f.shape[2] > 70000 else f.copy()
            fx[f"{name}_cstride"] = np.array(4 if f.shape[1] * f.shape[2] > 70000 else 1)
            if isinstance(out, tuple):
                fx[f"{name}_xyz"] = out[0].detach().numpy().copy()
        return fn

    for nm in stages:
        hooks.append(getattr(m, nm).register_forward_hook(keep(nm)))
    # inside the first curve block: the walk's curves and the aggregation output
    hooks.append(m.cic11.curvegrouping.register_forward_hook(
        lambda mod, inp, out: fx.__setitem__("cic11_curves", out.detach().numpy().copy())))
    hooks.append(m.cic11.curveaggregation.register_forward_hook(
        lambda mod, inp, out: fx.__setitem__("cic11_agg", out.detach().numpy()[:, ::2].copy())))
    hooks.append(m.cic11.lpfa.register_forward_hook(
        lambda mod, inp, out: fx.__setitem__("cic11_lpfa", out.detach().numpy()[:, ::2].copy())))
    with torch.no_grad():
        fx["logits"] = m(torch.from_numpy(x))[0].numpy()
    for h in hooks:
        h.remove()
    fx["stages"] = np.array(stages)
    np.savez_compressed(os.path.join(OUT, "curvenet_trace.npz"), **fx)
    print("curvenet_trace.npz:", {k: v.shape for k, v in fx.items() if hasattr(v, "shape")})


def gen_config_sizes():
    """Reference runs at the BASELINE configs' POINT COUNTS (B = 1, as the reference requires), which the smaller fixtures
    above leave to self-consistency tests:
      * dgcnn_n1024_*: the REAL geoA3_attack on the REAL DGCNN at N = 1024 (configs[2]'s size), both loss mixes, 2 x 12
        iterations (attack/GeoA3/GeoA3_attack.py:185-473);
      * curvenet_n4096_*: the REAL CurveNet's logits and input gradient at N = 4096 (configs[4]'s size; model/curvenet.py:50-73);
      * cngeo_n1024_*: the REAL geoA3_attack on the REAL CurveNet (N = 1024), both loss mixes — with every iterate the
        loop handed to the victim and the victim's logits on it (`*_iter_inputs`, `*_iter_logits`): this seeded-random
        CurveNet changes its arg-max class at almost every step of the attack (its forward is piecewise: kNN graphs, FPS,
        arg-max walks), so two fp32 implementations of the LOOP part ways within an iteration or two whatever they do — the
        victim's forward on the reference's own iterates is what can be pinned tightly.
    For every geoA3 case the ORACLE's loop in the intended semantics (true squared distances in knn_points, SURVEY A-2)
    is run here too, on the same reference model, and its curve is stored (`*_olosses`, `*_omask`, `*_obest`): the GPU
    tests compare with these arrays instead of running the CPU oracle on the GPU box's host, whose BLAS picks other code
    paths. The oracle's as-written mode is checked against the reference's curve right here (`*_aw_dev`)."""
    install_cpu_shim()
    canonical_unsorted_topk()
    import contextlib
    import io
    import time
    import types
    # ONE thread: torch's multi-threaded CPU reductions change their summation order from run to run, and these loops
    # amplify that (two 8-thread runs of the same reference call ended 0.5 % (DGCNN) / 1.2 % (CurveNet) apart); on one
    # thread the reference's run is a pure function of its inputs and the fixture can be regenerated bit for bit
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import ref_torch as ort
    ga, lu, estimate_normal = _geoa3_imports()
    from model import dgcnn as rd
    from model.curvenet import CurveNet

    class Dummy(torch.nn.Module):
        def forward(self, x):
            z = torch.zeros(x.shape[0], 40)
            return z, z, z

    fx = {}
    cases = {"ce_cd_hd_curv": {}, "margin_l2": dict(cls_loss_type='Margin', confidence=5., dis_loss_type='L2', hd_loss_weight=0,
                                                    curv_loss_weight=0)}
    fx["names"] = np.array(sorted(cases))

    def geo_cases(prefix, net, N, rng, seed, keep_iterates=False):
        for nm in sorted(cases):
            t0 = time.time()
            seen_in, seen_out = [], []
            hooks = []
            if keep_iterates:       # every cloud the reference's loop hands to the victim, and what the victim answered
                hooks = [net.register_forward_pre_hook(lambda mod, inp: seen_in.append(inp[0].detach().numpy().copy())),
                         net.register_forward_hook(lambda mod, inp, out: seen_out.append(out[0].detach().numpy().copy()))]
            cfg = types.SimpleNamespace(**{**GEO_BASE, **cases[nm], "npoint": N})
            pc = unit_cloud(rng, N)[None]
            with torch.no_grad():
                clean = int(torch.argmax(net(torch.from_numpy(pc).transpose(1, 2).contiguous())[0], dim=1))
            torch.manual_seed(seed)
            np.random.seed(seed)
            with contextlib.redirect_stdout(io.StringIO()):
                best, tgt, mask, steps, losses = ga.geoA3_attack(net, Dummy(), Dummy(), Dummy(), Dummy(), Dummy(),
                                                                 torch.from_numpy(pc), torch.tensor([clean]), cfg, 0, 1)
            k = f"{prefix}_{nm}"
            for h in hooks:
                h.remove()
            if keep_iterates:       # the loop evaluates every iterate twice (:339 and the success check): keep one of each
                fx[f"{k}_iter_inputs"] = np.concatenate(seen_in[1::2]).astype(np.float32)      # ([0] is the clean prediction above)
                fx[f"{k}_iter_logits"] = np.concatenate(seen_out[1::2]).astype(np.float32)
            fx[f"{k}_pc"], fx[f"{k}_label"] = pc, np.array([clean])
            fx[f"{k}_best"], fx[f"{k}_mask"] = best.detach().numpy(), np.asarray(mask)
            fx[f"{k}_steps"], fx[f"{k}_losses"] = np.array(steps), np.array(losses, dtype=np.float64)
            for tag, aw in (("o", False), ("aw", True)):
                torch.manual_seed(seed)
                np.random.seed(seed)
                ob, _, om, osteps, ol = ort.GeoA3Oracle(as_written=aw).attack(net, torch.from_numpy(pc), torch.tensor([clean]), cfg,
                                                                             per_sample_label=True)
                ol = np.array(ol, dtype=np.float64)
                if aw:          # the oracle pinned against the reference at this size
                    dev_ = float(np.abs(ol - fx[f"{k}_losses"]).max() / max(1e-12, np.abs(fx[f"{k}_losses"]).max()))
                    fx[f"{k}_aw_dev"] = np.array(dev_)
                    assert np.array_equal(np.asarray(om), np.asarray(mask)), (k, om, mask)
                else:
                    fx[f"{k}_olosses"], fx[f"{k}_omask"], fx[f"{k}_obest"] = ol, np.asarray(om), ob.detach().numpy()
            print(k, "mask", mask, "steps", steps, "loss[0], loss[-1]", losses[0], losses[-1], "oracle as-written rel dev",
                  fx[f"{k}_aw_dev"], "intended loss[-1]", fx[f"{k}_olosses"][-1], f"{time.time() - t0:.0f} s", flush=True)

    # (1) DGCNN, N = 1024
    net = rd.DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
    sd = ort.seeded_state_dict(net, 5)
    net.load_state_dict(sd)
    net.eval()
    fx["dgcnn_sha256"] = np.array(ort.state_sha256(sd))
    # (0) the oracle's intended-semantics curves for the N = 256 cases of geoa3_dgcnn.npz (same clouds, labels, seed 77)
    g256 = np.load(os.path.join(OUT, "geoa3_dgcnn.npz"))
    for nm in sorted(cases):
        cfg = types.SimpleNamespace(**{**GEO_BASE, **cases[nm]})
        torch.manual_seed(77)
        np.random.seed(77)
        ob, _, om, _, ol = ort.GeoA3Oracle(as_written=False).attack(net, torch.from_numpy(g256[f"{nm}_pc"]),
                                                                    torch.from_numpy(g256[f"{nm}_label"]), cfg, per_sample_label=True)
        fx[f"dgcnn_n256_{nm}_olosses"], fx[f"dgcnn_n256_{nm}_omask"] = np.array(ol, dtype=np.float64), np.asarray(om)
        fx[f"dgcnn_n256_{nm}_obest"] = ob.detach().numpy()
    geo_cases("dgcnn_n1024", net, 1024, np.random.default_rng(1357911), 78)

    # (2) CurveNet forward / backward at N = 4096 (same weights as curvenet.npz: seed 9, gain 1)
    m = CurveNet(num_classes=40)
    sd = ort.seeded_state_dict(m, 9, gain=1.0)
    m.load_state_dict(sd)
    m.eval()
    fx["curvenet_sha256"] = np.array(ort.state_sha256(sd))
    rng = np.random.default_rng(975310)
    x = unit_cloud(rng, 4096)[None].transpose(0, 2, 1).copy()
    tx = torch.from_numpy(x).requires_grad_()
    t0 = time.time()
    out = m(tx)[0]
    w = torch.from_numpy(rng.standard_normal(out.shape).astype(np.float32))
    (out * w).sum().backward()
    fx["curvenet_n4096_x"], fx["curvenet_n4096_logits"] = x, out.detach().numpy()
    fx["curvenet_n4096_w"], fx["curvenet_n4096_gx"] = w.numpy(), tx.grad.numpy()
    print("curvenet n4096 forward + backward", f"{time.time() - t0:.0f} s", flush=True)

    # (3) geoA3_attack on CurveNet, N = 1024: the bias-re-centred classifier of cw_curvenet.npz (labels that mean something)
    cw = np.load(os.path.join(OUT, "cw_curvenet.npz"))
    m2 = CurveNet(num_classes=40)
    m2.load_state_dict(ort.seeded_state_dict(m2, 9))
    with torch.no_grad():
        m2.conv2.bias.copy_(torch.from_numpy(cw["conv2_bias"]))
    m2.eval()
    assert ort.state_sha256(m2.state_dict()) == str(cw["sha256"])
    fx["cngeo_sha256"] = np.array(str(cw["sha256"]))
    geo_cases("cngeo_n1024", m2, 1024, np.random.default_rng(86421), 79, keep_iterates=True)
    np.savez_compressed(os.path.join(OUT, "config_sizes.npz"), **fx)
    print("config_sizes.npz:", len(fx), "arrays")


SECTIONS = {"config_sizes": gen_config_sizes, "metrics_n4096": gen_metrics_n4096, "curvenet_trace": gen_curvenet_trace, "f4": gen_f4, "geoa3_dgcnn": gen_geoa3_dgcnn, "cw_curvenet": gen_cw_curvenet, "curvenet_blocks": gen_curvenet_blocks, "formats": gen_formats, "cw_additional": gen_cw_additional, "curvenet": gen_curvenet, "aof": gen_aof, "geoa3": gen_geoa3, "dgcnn": gen_dgcnn, "metrics": gen_metrics, "pointnet": gen_pointnet, "cw": gen_cw, "pointnet2": gen_pointnet2, "knn": gen_knn, "cw_full": gen_cw_full,
            "knn_full": gen_knn_full}

if __name__ == "__main__":
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
