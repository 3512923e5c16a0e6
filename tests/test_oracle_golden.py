"""CPU: the oracle (oracle/ref_numpy.py) against the golden vectors captured from the real reference."""
import numpy as np
import pytest

from oracle import ref_numpy as orc


def test_known_answers_from_reference_comments(metrics_fx):
    # utils/dis_utils_numpy.py:40-46 worked example: chamfer 2*sqrt(3), hausdorff sqrt(3)
    a, b = np.ones((3, 3)), 2 * np.ones((3, 3))
    assert orc.chamfer(a, b) == pytest.approx(2 * np.sqrt(3), rel=1e-12)
    assert orc.sgd_hausdorff_dis(a, b) == pytest.approx(np.sqrt(3), rel=1e-12)
    assert orc.bid_hausdorff_dis(a, b) == pytest.approx(np.sqrt(3), rel=1e-12)
    # SURVEY §8(c) hand case
    a = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], float)
    b = np.array([[0, 0, 1], [3, 0, 0]], float)
    assert orc.chamfer(a, b) == pytest.approx(3.0500938466, rel=1e-9)
    assert orc.sgd_hausdorff_dis(a, b) == pytest.approx(2.2360679775, rel=1e-9)
    assert orc.sgd_hausdorff_dis(b, a) == pytest.approx(2.0, rel=1e-12)


def test_numpy_metrics_match_reference(metrics_fx):
    fx = metrics_fx
    for nm in fx["np_names"]:
        a, b = fx[f"np_{nm}_a"], fx[f"np_{nm}_b"]
        ref = fx[f"np_{nm}_out"]
        got = [orc.chamfer(a, b), orc.sgd_hausdorff_dis(a, b), orc.sgd_hausdorff_dis(b, a), orc.bid_hausdorff_dis(a, b)]
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-14, err_msg=str(nm))
    np.testing.assert_allclose(orc.pairwise_distances(fx["np_hand_asym_a"], fx["np_hand_asym_b"]),
                               fx["np_hand_asym_M"], rtol=1e-13)
    np.testing.assert_allclose(orc.pairwise_distances(fx["np_rand_64_a"], fx["np_rand_64_b"]),
                               fx["np_rand_64_M"], rtol=1e-13)


def test_numpy_metrics_match_reference_n4096(metrics4096_fx):
    """the N=4096 pair (and a ragged 4096 x 3000 one) of the real reference."""
    fx = metrics4096_fx
    for nm in fx["np_names"]:
        a, b = fx[f"np_{nm}_a"], fx[f"np_{nm}_b"]
        got = [orc.chamfer(a, b), orc.sgd_hausdorff_dis(a, b), orc.sgd_hausdorff_dis(b, a), orc.bid_hausdorff_dis(a, b)]
        np.testing.assert_allclose(got, fx[f"np_{nm}_out"], rtol=1e-12, atol=1e-14, err_msg=str(nm))


def test_torch_twin_metrics_match_reference(metrics_fx):
    fx = metrics_fx
    for nm in fx["t_names"]:
        a, b = fx[f"t_{nm}_a"], fx[f"t_{nm}_b"]
        ref = fx[f"t_{nm}_out"]  # produced by the reference in fp32 through cdist's mm-expansion
        got = [orc.torch_euclidean_distances(a, b), orc.torch_chamfer(a, b),
               orc.torch_sgd_hausdorff_dis(a, b), orc.torch_bid_hausdorff_dis(a, b)]
        # fp32 cdist expansion is 1e-6..5e-5 off the exact value: tolerance 2e-4
        np.testing.assert_allclose(got, ref, rtol=2e-4, err_msg=str(nm))
    # the worked example commented at utils/dis_utils_torch.py:30-35
    a, b = fx["t_kat_a"], fx["t_kat_b"]
    assert orc.torch_euclidean_distances(a, b) == pytest.approx(5.9135914, rel=1e-6)
    assert orc.torch_chamfer(a, b) == pytest.approx(3.7032480, rel=1e-6)
    assert orc.torch_sgd_hausdorff_dis(a, b) == pytest.approx(1.7320508, rel=1e-6)
    assert orc.torch_bid_hausdorff_dis(a, b) == pytest.approx(2.4494898, rel=1e-6)
    np.testing.assert_allclose(orc.torch_pairwise_distances(fx["t_rand_128_a"], fx["t_rand_128_b"]),
                               fx["t_rand_128_M"], rtol=1e-3, atol=2e-4)


def test_cw_functors_match_reference(metrics_fx):
    fx = metrics_fx
    for nm in fx["cw_names"]:
        p, g = fx[f"cw_{nm}_preds"], fx[f"cw_{nm}_gts"]
        l1, l2 = orc.cw_chamfer(p, g)
        np.testing.assert_allclose(np.stack([l1, l2]), fx[f"cw_{nm}_chamfer"], rtol=1e-9, atol=1e-14)
        h1, h2 = orc.cw_hausdorff(p, g)
        np.testing.assert_allclose(np.stack([h1, h2]), fx[f"cw_{nm}_hausdorff"], rtol=1e-9, atol=1e-14)


def test_fp32_fma_emulation_is_close_to_f64(metrics_fx):
    fx = metrics_fx
    a, b = fx["np_rand_1024_a"], fx["np_rand_1024_b"]
    d64, i64 = orc.nn_sq(a, b)
    d32, i32 = orc.nn_sq_f32(a, b)
    np.testing.assert_allclose(d32, d64, rtol=1e-6)
    assert (i32 == i64).mean() > 0.999


# ---------------------------------------------------------------------------------------------------------
# torch oracle (victim model + CW loop) against the real reference's outputs
# ---------------------------------------------------------------------------------------------------------
import os

import torch

from conftest import GOLDEN
from oracle import ref_torch as ort


def _oracle_pointnet(seed, k=40):
    m = ort.PointNetCls(k=k)
    sd = ort.seeded_state_dict(m, seed)
    m.load_state_dict(sd)
    m.eval()
    return m, ort.state_sha256(sd)


def test_oracle_pointnet_matches_reference():
    fx = np.load(os.path.join(GOLDEN, "pointnet.npz"))
    model, sha = _oracle_pointnet(0)
    assert sha == str(fx["sha256"]), "seeded weights differ from the ones the reference was run with"
    for nm in ("b2_n1024", "b3_n200"):
        x = torch.from_numpy(fx[f"{nm}_x"]).requires_grad_()
        logp, trans, _ = model(x)
        np.testing.assert_allclose(logp.detach().numpy(), fx[f"{nm}_logp"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(trans.detach().numpy(), fx[f"{nm}_trans"], rtol=1e-5, atol=1e-6)
        (logp * torch.from_numpy(fx[f"{nm}_w"])).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), fx[f"{nm}_gx"], rtol=1e-4, atol=1e-6)


def test_oracle_cw_attack_matches_reference():
    fx = np.load(os.path.join(GOLDEN, "cw.npz"))
    model, sha = _oracle_pointnet(0)
    assert sha == str(fx["sha256"])
    for nm in fx["names"]:
        steps, iters, kappa = fx[f"{nm}_cfg"]
        method = "untarget" if "untarget" in str(nm) else "target"
        adv_func = ort.UntargetedLogitsAdvLoss(kappa) if method == "untarget" else ort.LogitsAdvLoss(kappa)
        dist = ort.L2Dist() if str(nm).startswith("l2") else ort.ChannelFirst(ort.ChamferDist())
        traj = []
        torch.manual_seed(1000)
        bd, ba, sn, _ = ort.cw_attack(model, torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_target"]),
                                      adv_func, dist, ort.ClipPointsLinf(0.18), attack_lr=1e-2,
                                      binary_step=int(steps), num_iter=int(iters), attack_method=method,
                                      record=lambda s, i, a: traj.append(a[0].copy()))
        # same ops on the same CPU: trajectories agree to fp32 rounding
        np.testing.assert_allclose(np.stack(traj), fx[f"{nm}_traj"], rtol=0, atol=2e-6, err_msg=str(nm))
        np.testing.assert_allclose(bd, fx[f"{nm}_bestdist"], rtol=1e-5, err_msg=str(nm))
        np.testing.assert_allclose(ba, fx[f"{nm}_bestattack"], atol=2e-6, err_msg=str(nm))
        assert sn == int(fx[f"{nm}_success"])


def test_oracle_pointnet2_matches_reference():
    fx = np.load(os.path.join(GOLDEN, "pointnet2.npz"))
    for nm in fx["names"]:
        xyz = torch.from_numpy(fx[f"{nm}_xyz"])
        S, r, ns = fx[f"{nm}_cfg"]
        S, ns = int(S), int(ns)
        torch.manual_seed(11)
        fps = ort.farthest_point_sample(xyz, S)
        assert np.array_equal(fps.numpy(), fx[f"{nm}_fps"]), nm
        new_xyz = ort.index_points(xyz, fps)
        assert np.array_equal(ort.query_ball_point(float(r), ns, xyz, new_xyz).numpy(), fx[f"{nm}_ball"]), nm
        torch.manual_seed(11)
        nx, npts = ort.sample_and_group(S, float(r), ns, xyz, torch.from_numpy(fx[f"{nm}_feats"]))
        np.testing.assert_array_equal(nx.numpy(), fx[f"{nm}_sg_new_xyz"])
        np.testing.assert_array_equal(npts.numpy(), fx[f"{nm}_sg_new_points"])
    for cname, cls in (("ssg", ort.PointNet_Ssg), ("msg", ort.PointNet_Msg)):
        m = cls(40)
        sd = ort.seeded_state_dict(m, 3)
        m.load_state_dict(sd)
        m.eval()
        assert ort.state_sha256(sd) == str(fx[f"{cname}_sha256"])
        x = torch.from_numpy(fx[f"{cname}_x"]).requires_grad_()
        torch.manual_seed(21)
        logp = m(x)[0]
        np.testing.assert_allclose(logp.detach().numpy(), fx[f"{cname}_logp"], rtol=1e-5, atol=1e-6)
        (logp * torch.from_numpy(fx[f"{cname}_w"])).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), fx[f"{cname}_gx"], rtol=1e-4, atol=1e-6)


def test_oracle_knn_attack_matches_reference():
    fx = np.load(os.path.join(GOLDEN, "knn.npz"))
    pn, _ = _oracle_pointnet(0)
    ssg = ort.PointNet_Ssg(40)
    ssg.load_state_dict(ort.seeded_state_dict(ssg, 3))
    ssg.eval()
    for nm in fx["names"]:
        iters, lr, kappa = fx[f"{nm}_cfg"]
        victim = pn if str(nm).startswith("pointnet") else ssg
        dist = ort.ChamferkNNDist() if "chamferknn" in str(nm) else ort.ChamferDist()
        torch.manual_seed(1000)
        adv, sn = ort.knn_attack(victim, torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_target"]),
                                 ort.UntargetedLogitsAdvLoss(kappa), dist, ort.ProjectInnerClipLinf(0.18),
                                 attack_lr=float(lr), num_iter=int(iters))
        np.testing.assert_allclose(adv, fx[f"{nm}_adv"], atol=2e-6, err_msg=str(nm))
        assert sn == int(fx[f"{nm}_success"])


def test_oracle_dgcnn_matches_reference():
    import types
    fx = np.load(os.path.join(GOLDEN, "dgcnn.npz"))
    m = ort.DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), 40)
    sd = ort.seeded_state_dict(m, 5)
    m.load_state_dict(sd)
    m.eval()
    assert ort.state_sha256(sd) == str(fx["sha256"])
    x = torch.from_numpy(fx["x"]).requires_grad_()
    logp = m(x)[0]
    np.testing.assert_allclose(logp.detach().numpy(), fx["logp"], rtol=1e-5, atol=1e-6)
    (logp * torch.from_numpy(fx["w"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), fx["gx"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(ort.dgcnn_knn(torch.from_numpy(fx["feat"]), 20).numpy(), fx["feat_knn"])
    np.testing.assert_array_equal(ort.get_graph_feature(torch.from_numpy(fx["x"][:, :, :64].copy()), k=8).numpy(),
                                  fx["graph_feature"])


def _geo_cfg(**over):
    import types
    base = dict(attack_method='untarget', curv_loss_weight=1.0, curv_loss_knn=16, initial_const=10, iter_max_steps=12,
                binary_max_steps=2, is_partial_var=False, optim='adam', lr=0.01, npoint=256, is_subsample_opt=False,
                eval_num=1, is_pre_jitter_input=False, cls_loss_type='CE', classes=40, confidence=0, dis_loss_type='CD',
                is_cd_single_side=False, dis_loss_weight=1.0, hd_loss_weight=0.1, uniform_loss_weight=0.0,
                is_use_lr_scheduler=False, is_debug=False, is_pro_grad=False, cc_linf=0.0, binary_step=2, num_iter=12,
                is_real_offset=False, knn_range=3, calculate_project_jitter_noise_iter=50, jitter_k=16,
                jitter_sigma=0.01, jitter_clip=0.05)
    base.update(over)
    return types.SimpleNamespace(**base)


GEO_CASES = {"ce_cd_hd_curv": {}, "margin_l2": dict(cls_loss_type='Margin', confidence=5., dis_loss_type='L2',
                                                    hd_loss_weight=0, curv_loss_weight=0)}


def test_oracle_geoa3_matches_reference():
    fx = np.load(os.path.join(GOLDEN, "geoa3.npz"))
    net, _ = _oracle_pointnet(0)
    orc_g = ort.GeoA3Oracle(as_written=True)
    # unit level: normals, kappas, loss terms
    pc, adv = torch.from_numpy(fx["unit_pc"]), torch.from_numpy(fx["unit_adv"])
    normal = orc_g.estimate_normal(pc, 3)
    agree = np.isclose(np.abs((normal.numpy() * fx["unit_normal"]).sum(1)), 1.0, atol=1e-3)
    assert agree.mean() > 0.97        # eigenvector of a (near-)degenerate 3-point covariance may differ
    nr = torch.from_numpy(fx["unit_normal"])
    ko = orc_g._kappa(pc, nr, 16)
    np.testing.assert_allclose(ko.numpy(), fx["unit_kappa_ori"], rtol=1e-4, atol=1e-6)
    ak, _ = orc_g.kappa_adv(adv, pc, nr, 16)
    np.testing.assert_allclose(ak.numpy(), fx["unit_kappa_adv"], rtol=1e-4, atol=1e-6)
    terms = [float(orc_g.chamfer_loss(adv, pc)), float(orc_g.pseudo_chamfer_loss(adv, pc)), float(orc_g.hausdorff_loss(adv, pc)),
             float(orc_g.curvature_loss(adv, pc, ak, ko))]
    np.testing.assert_allclose(terms, fx["unit_terms"][:4], rtol=1e-4)
    # full loop
    for nm in fx["names"]:
        cfg = _geo_cfg(**GEO_CASES[str(nm)])
        torch.manual_seed(77)
        np.random.seed(77)
        best, tgt, mask, steps, losses = orc_g.attack(net, torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_label"]), cfg)
        assert np.array_equal(mask, fx[f"{nm}_mask"]), nm
        np.testing.assert_allclose(np.array(losses), fx[f"{nm}_losses"], rtol=2e-3, atol=1e-4, err_msg=str(nm))
        if mask.any():
            assert steps == fx[f"{nm}_steps"].tolist()
            np.testing.assert_allclose(best.numpy(), fx[f"{nm}_best"], atol=1e-4)


def test_oracle_geoa3_on_dgcnn_matches_reference():
    """BASELINE configs[2] as a workload: the oracle's GeoA3 loop on the oracle's DGCNN against the real reference's
    geoA3_attack on the real reference DGCNN (tests/golden/geoa3_dgcnn.npz; same CPU-generator offsets)."""
    import types
    fx = np.load(os.path.join(GOLDEN, "geoa3_dgcnn.npz"))
    net = ort.DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
    sd = ort.seeded_state_dict(net, 5)
    net.load_state_dict(sd)
    net.eval()
    assert ort.state_sha256(sd) == str(fx["sha256"])
    orc_g = ort.GeoA3Oracle(as_written=True)
    for nm in fx["names"]:
        cfg = _geo_cfg(**GEO_CASES[str(nm)])
        torch.manual_seed(77)
        np.random.seed(77)
        best, tgt, mask, steps, losses = orc_g.attack(net, torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_label"]), cfg)
        assert np.array_equal(mask, fx[f"{nm}_mask"]), nm
        np.testing.assert_allclose(np.array(losses), fx[f"{nm}_losses"], rtol=2e-3, atol=1e-4, err_msg=str(nm))
        assert steps == fx[f"{nm}_steps"].tolist()
        np.testing.assert_allclose(best.numpy(), fx[f"{nm}_best"], atol=1e-4)


def test_oracle_geoa3_on_dgcnn_at_config_size_matches_reference():
    """The same pin at configs[2]'s point count: the oracle's GeoA3 loop (as written, SURVEY A-2) on the oracle's DGCNN
    against the real reference's geoA3_attack at N = 1024 (tests/golden/config_sizes.npz, generated on ONE thread — on one
    thread this run reproduces it to rounding; any thread count stays inside the band two reference runs differ by), and
    the stored intended-semantics curve the GPU tests compare with is what this oracle computes."""
    import types
    fx = np.load(os.path.join(GOLDEN, "config_sizes.npz"))
    net = ort.DGCNN(types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
    sd = ort.seeded_state_dict(net, 5)
    net.load_state_dict(sd)
    net.eval()
    assert ort.state_sha256(sd) == str(fx["dgcnn_sha256"])
    was = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        nm = "margin_l2"
        key = f"dgcnn_n1024_{nm}"
        cfg = _geo_cfg(npoint=1024, **GEO_CASES[nm])
        for as_written, ref in ((True, fx[f"{key}_losses"]), (False, fx[f"{key}_olosses"])):
            torch.manual_seed(78)
            np.random.seed(78)
            best, tgt, mask, steps, losses = ort.GeoA3Oracle(as_written=as_written).attack(
                net, torch.from_numpy(fx[f"{key}_pc"]), torch.from_numpy(fx[f"{key}_label"]), cfg, per_sample_label=True)
            assert np.array_equal(mask, fx[f"{key}_mask"])
            L = np.array(losses)
            np.testing.assert_allclose(L[:3], ref[:3], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(L, ref, rtol=3e-2, atol=1e-3)
    finally:
        torch.set_num_threads(was)


def test_oracle_f4_functors_match_reference():
    """FarthestDist / FarChamferDist / L2ChamferDist of the real reference (tests/golden/f4.npz)."""
    fx = np.load(os.path.join(GOLDEN, "f4.npz"))
    w = torch.from_numpy(fx["weights"])
    adv = torch.from_numpy(fx["adv_clusters"])
    B, na, cp, _ = adv.shape
    for dt, tag, tol in ((torch.float64, "f64", 1e-12), (torch.float32, "f32", 1e-6)):
        a = adv.to(dt).clone().requires_grad_()
        per = ort.farthest_dist(a, weights=w, batch_avg=False)
        ort.farthest_dist(a, weights=w).backward()
        np.testing.assert_allclose(per.detach().numpy(), fx[f"far_{tag}"], rtol=tol)
        np.testing.assert_allclose(a.grad.numpy(), fx[f"far_{tag}_grad"], rtol=max(tol, 1e-9), atol=1e-12)
    np.testing.assert_allclose(ort.farthest_dist(adv.double(), batch_avg=False).numpy(), fx["far_noweights"], rtol=1e-12)
    ori = torch.from_numpy(fx["ori"]).double()
    for method in ("adv2ori", "ori2adv", "both"):
        a = adv.reshape(B, na * cp, 3).double().requires_grad_()
        per = ort.far_chamfer_dist(a, ori, na, method=method, weights=w, batch_avg=False)
        ort.far_chamfer_dist(a, ori, na, method=method, weights=w).backward()
        np.testing.assert_allclose(per.detach().numpy(), fx[f"farchamfer_{method}"], rtol=1e-10)
        np.testing.assert_allclose(a.grad.numpy(), fx[f"farchamfer_{method}_grad"], rtol=1e-8, atol=1e-12)
    placed = torch.from_numpy(fx["placed"]).double().requires_grad_()
    ao = torch.from_numpy(fx["adv_obj"]).double().requires_grad_()
    oo = torch.from_numpy(fx["ori_obj"]).double()
    per = ort.l2_chamfer_dist(placed, ori, ao, oo, weights=w, batch_avg=False)
    ort.l2_chamfer_dist(placed, ori, ao, oo, weights=w).backward()
    np.testing.assert_allclose(per.detach().numpy(), fx["l2chamfer"], rtol=1e-10)
    np.testing.assert_allclose(placed.grad.numpy(), fx["l2chamfer_grad_placed"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(ao.grad.numpy(), fx["l2chamfer_grad_obj"], rtol=1e-8, atol=1e-12)


def test_oracle_aof_matches_reference():
    fx = np.load(os.path.join(GOLDEN, "aof.npz"))
    e, v, L = ort.get_Laplace_from_pc(torch.from_numpy(fx["lap_pc"]))
    np.testing.assert_allclose(e.numpy(), fx["lap_eig"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(ort.aof_knn(torch.from_numpy(fx["lap_pc"]), 30).numpy(), fx["lap_knn"])
    net, _ = _oracle_pointnet(0)
    torch.manual_seed(31)
    bd, adv, sn = ort.taof_attack(net, torch.from_numpy(fx["atk_pc"]), torch.from_numpy(fx["atk_target"]),
                                  torch.from_numpy(fx["atk_ytruth"]), ort.LogitsAdvLoss(0.), ort.ClipPointsLinf(0.18),
                                  binary_step=2, num_iter=10, low_pass=40)
    np.testing.assert_allclose(bd, fx["atk_bestdist"], rtol=1e-4)
    np.testing.assert_allclose(adv, fx["atk_adv"], atol=1e-5)
    assert sn == int(fx["atk_success"])


ADDL_CASES = {
    "z_chamfer_target": dict(dist="chamfer", target=True, d1=True, renorm=False, eot=False, resample=False, steps=2, iters=12, kappa=0.),
    "eot_renorm_chamferknn_target": dict(dist="chamferknn", target=True, d1=True, renorm=True, eot=True, resample=False, steps=2, iters=6, kappa=0.),
    "free_l2_untarget": dict(dist="l2", target=False, d1=False, renorm=False, eot=False, resample=False, steps=3, iters=10, kappa=5.),
    "resample_chamfer_target": dict(dist="chamfer", target=True, d1=True, renorm=False, eot=True, resample=True, steps=1, iters=2, kappa=0.),
}


class _AddlAdv:
    """adv_func(logits, label, whether_target=...) over the oracle's functors (additional_exp/CW_attack.py:245-259)."""

    def __init__(self, mod, kappa):
        self.t, self.u = mod.LogitsAdvLoss(kappa), mod.UntargetedLogitsAdvLoss(kappa)

    def __call__(self, logits, label, whether_target=1):
        return (self.t if whether_target else self.u)(logits, label)


def test_oracle_cw_additional_matches_reference():
    """oracle.cw_additional_attack vs the real attack/additional_exp/CW_attack.py (fixtures: make_golden.py
    cw_additional): same arithmetic on the same CPU backend, same random streams -> same iterates."""
    import random
    fx = np.load(os.path.join(GOLDEN, "cw_additional.npz"))
    net, _ = _oracle_pointnet(0)
    assert sorted(ADDL_CASES) == sorted(str(n) for n in fx["names"])
    for nm, c in ADDL_CASES.items():
        inner = {"chamfer": ort.ChamferDist(), "chamferknn": ort.ChamferkNNDist(), "l2": ort.L2Dist()}[c["dist"]]
        log = []

        def dist(adv, ori, w, inner=inner, per_sample=not c["target"], log=log):
            log.append(adv.detach().numpy().copy())
            return inner(adv, ori, w, batch_avg=not per_sample)
        torch.manual_seed(2000)
        random.seed(2000)
        np.random.seed(2000)
        bd, ba, sn = ort.cw_additional_attack(
            net, torch.from_numpy(fx[f"{nm}_pc"]), torch.from_numpy(fx[f"{nm}_target"]), torch.from_numpy(fx[f"{nm}_origin"]),
            _AddlAdv(ort, c["kappa"]), dist, attack_lr=1e-2, binary_step=c["steps"], num_iter=c["iters"],
            whether_target=c["target"], whether_1d=c["d1"], whether_renormalization=c["renorm"],
            whether_3Dtransform=c["eot"], whether_resample=c["resample"])
        traj = np.stack(log)[:, 0]
        if traj.shape[1] > 512:
            traj = traj[:, ::16]
        np.testing.assert_allclose(traj, fx[f"{nm}_traj"], atol=2e-5, err_msg=nm)
        assert sn == int(fx[f"{nm}_success"]), nm
        np.testing.assert_allclose(bd, fx[f"{nm}_bestdist"], rtol=1e-3, err_msg=nm)
        if sn:
            np.testing.assert_allclose(ba, fx[f"{nm}_bestattack"], atol=2e-5, err_msg=nm)
