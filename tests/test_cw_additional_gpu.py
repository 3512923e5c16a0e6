"""GPU parity: attack/additional_exp/CW_attack.py mirror vs the real reference's golden runs and
the oracle. The fp32 |x|^2+|y|^2-2xy Chamfer of the reference is rounding noise at adv ~ ori, so
iterates are compared with the oracle evaluating the SAME loop with float64 distances; results (success, labels,
constraints) are compared with the reference's own outputs."""
import importlib
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import hip_pointnet, oracle_pointnet, unit_cloud
from oracle import ref_torch as ort
from test_oracle_golden import ADDL_CASES, _AddlAdv

pytestmark = pytest.mark.gpu
M = importlib.import_module


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "cw_additional.npz"))


def _mods():
    return (M("3dpointcloudattack_amd.attack.additional_exp.CW_attack"), M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"),
            M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils"))


class _F64:
    """The oracle's distance functor evaluated in float64 (the reference's algorithm without its fp32 expansion noise)."""

    def __init__(self, inner, per_sample):
        self.inner, self.per_sample = inner, per_sample

    def __call__(self, adv, ori, w):
        return self.inner(adv.double(), ori.double(), w, batch_avg=not self.per_sample).float()


def _seed():
    torch.manual_seed(2000)
    random.seed(2000)
    np.random.seed(2000)


@pytest.mark.parametrize("nm", sorted(ADDL_CASES))
def test_additional_cw_vs_reference_and_oracle(dev, fx, nm):
    cwm, adv, dist = _mods()
    c = ADDL_CASES[nm]
    net, _ = hip_pointnet(0, dev)
    onet, _ = oracle_pointnet(0)
    pc, tgt, org = torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_target"]), torch.from_numpy(fx[f"{nm}_origin"])
    df = {"chamfer": dist.ChamferDist(), "chamferknn": dist.ChamferkNNDist(), "l2": dist.L2Dist()}[c["dist"]]
    per_sample = not c["target"]
    atk = cwm.CW(net, cwm.AdvLossAdapter(adv.LogitsAdvLoss(c["kappa"]), adv.UntargetedLogitsAdvLoss(c["kappa"])),
                 (lambda a, o, w: df(a, o, w, batch_avg=not per_sample)), attack_lr=1e-2, binary_step=c["steps"],
                 num_iter=c["iters"], whether_target=c["target"], whether_1d=c["d1"], whether_renormalization=c["renorm"],
                 whether_3Dtransform=c["eot"], whether_resample=c["resample"])
    _seed()
    bd, ba, sn = atk.attack(pc, tgt, org)
    assert bd.shape == (1,) and ba.shape == fx[f"{nm}_bestattack"].shape and ba.dtype == np.float64
    # same outcome as the real reference
    assert sn == int(fx[f"{nm}_success"]), nm
    # the oracle with float64 distances, same random streams
    oi = {"chamfer": ort.ChamferDist(), "chamferknn": ort.ChamferkNNDist(), "l2": ort.L2Dist()}[c["dist"]]
    _seed()
    obd, oba, osn = ort.cw_additional_attack(onet, pc, tgt, org, _AddlAdv(ort, c["kappa"]), _F64(oi, per_sample),
                                             attack_lr=1e-2, binary_step=c["steps"], num_iter=c["iters"],
                                             whether_target=c["target"], whether_1d=c["d1"],
                                             whether_renormalization=c["renorm"], whether_3Dtransform=c["eot"],
                                             whether_resample=c["resample"])
    assert sn == osn
    if sn:
        # best distance = a minimum over the iterations at which the (EOT-averaged) attack succeeds: WHICH iteration that
        # is flips with fp32 rounding of the victim (measured over two valid fp32 evaluations of the heads — library
        # GEMM and this package's small-batch kernel — 1.2 % and 3.2 % from the float64 oracle on the EOT case); the
        # clouds themselves are held to 2e-3 on 97 % of the coordinates below, the labels exactly
        np.testing.assert_allclose(bd, obd, rtol=5e-2, err_msg=nm)
        np.testing.assert_allclose(bd, fx[f"{nm}_bestdist"], rtol=0.15, err_msg=nm)       # the fp32-noisy reference
        # (EOT: ten random rotations per step feed the victim — the most chaotic configuration; measured 0.90-0.98
        # of the coordinates within 2e-3 over two valid fp32 evaluations of the victim's heads, two thirds of them being
        # the untouched x / y of a z-only attack when whether_1d is set)
        assert np.mean(np.abs(ba - oba) <= 2e-3) > (0.85 if c["eot"] else 0.97), nm
        with torch.no_grad():
            lab = onet(torch.from_numpy(ba).float().transpose(1, 2).contiguous())[0].argmax(1)
            lab_ref = onet(torch.from_numpy(fx[f"{nm}_bestattack"]).transpose(1, 2).contiguous())[0].argmax(1)
        assert torch.equal(lab, lab_ref), nm
    if c["d1"]:
        # z-only perturbation inside the +-0.4 box: x and y are bit-identical to the input (:266-275)
        assert np.array_equal(ba[..., :2].astype(np.float32), fx[f"{nm}_pc"][..., :2])
        assert np.all(np.abs(ba[..., 2] - fx[f"{nm}_pc"][..., 2]) <= 0.4 + 1e-6)


def test_additional_cw_batched(dev):
    """B > 1 (the reference is B = 1 only): per-sample bookkeeping, constraints on every sample."""
    cwm, adv, dist = _mods()
    net, _ = hip_pointnet(0, dev)
    rng = np.random.default_rng(9)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, 192) for _ in range(3)]))
    with torch.no_grad():
        lg = net(pcs.transpose(1, 2).contiguous().to(dev))[0]
    tgt = lg.topk(2, dim=1)[1][:, 1].cpu()
    df = dist.ChamferDist()
    atk = cwm.CW(net, cwm.AdvLossAdapter(adv.LogitsAdvLoss(0.), adv.UntargetedLogitsAdvLoss(0.)),
                 (lambda a, o, w: df(a, o, w, batch_avg=False)), binary_step=2, num_iter=10, whether_target=True,
                 whether_1d=True, whether_3Dtransform=True, whether_renormalization=True)
    _seed()
    bd, ba, sn = atk.attack(pcs, tgt, lg.argmax(1).cpu())
    assert bd.shape == (3,) and ba.shape == (3, 192, 3) and 0 <= sn <= 3
    assert np.array_equal(ba[..., :2].astype(np.float32), pcs.numpy()[..., :2])
    assert np.all(np.abs(ba[..., 2] - pcs.numpy()[..., 2]) <= 0.4 + 1e-6) and np.isfinite(ba).all()
    ok = bd < 1e9
    if ok.any():
        with torch.no_grad():
            # success was judged on the renormalised cloud the victim saw (:105-117)
            x = cwm._renormalize(torch.from_numpy(ba).float().transpose(1, 2).contiguous().to(dev))
            lab = net(x)[0].argmax(1).cpu()
        assert torch.equal(lab[torch.from_numpy(ok)], tgt[torch.from_numpy(ok)])
