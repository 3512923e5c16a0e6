"""Whole-attack wall time per iteration for BASELINE.json configs[2] (GeoA3 on DGCNN, B=32 N=1024), configs[3]
(KNN attack on PointNet++ SSG, B=64 N=2048) and one GPU's share of configs[4] (CW and GeoA3 on CurveNet, B=32
N=4096), short runs; prints JSON. PC3D_GRAPH_VICTIM=0 / 1 launches every victim eagerly / from hipGraphs (unset: each victim's default)."""
import importlib, sys, os, json, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import unit_cloud
seeded_state_dict = importlib.import_module("3dpointcloudattack_amd.seeding").seeded_state_dict
from test_oracle_golden import _geo_cfg
M = importlib.import_module
dev = torch.device("cuda:0")
def mk(modname, cls, seed=0, **kw):
    m = getattr(M(f"3dpointcloudattack_amd.model.{modname}"), cls)(**kw)
    m.load_state_dict(seeded_state_dict(m, seed)); return m.to(dev).eval()
rng = np.random.default_rng(0)
which = sys.argv[1:] or ["geoa3", "knn", "aof"]
_gv = os.environ.get("PC3D_GRAPH_VICTIM")      # unset: every victim's own default (graphed.wrap)
GRAPH = None if _gv is None else _gv != "0"
if os.environ.get("PC3D_FUSE12") == "0":      # A/B switch for the experiment recorded in DESIGN.md
    M("3dpointcloudattack_amd.model.pointnet2_utils").FUSE_LAYERS_1_2 = False
if os.environ.get("PC3D_L2_BITS") == "0":
    M("3dpointcloudattack_amd.ops").LAYER2_SIGN_BITS = False
if os.environ.get("PC3D_SA_TABLE") == "0":
    M("3dpointcloudattack_amd.ops").SA_BLOCK_TABLE = False
if os.environ.get("PC3D_SA_TABLE") == "32":      # unit tables for the streaming kernel only
    _ops = M("3dpointcloudattack_amd.ops")
    _orig_unit = _ops.sa_chain_table_unit
    _ops.sa_chain_table_unit = lambda *a: (_orig_unit(*a) if _orig_unit(*a) == 32 else 0)
if os.environ.get("PC3D_SA_UNIT8") in ("0", "res16"):   # A/B: the 16- / 32-row tables (all shapes, or the resident kernel only)
    _ops = M("3dpointcloudattack_amd.ops")
    _orig_unit8 = _ops.sa_chain_table_unit
    def _unit_old(S, ns, C1, C2, C3, _all=os.environ["PC3D_SA_UNIT8"] == "0"):
        u = _orig_unit8(S, ns, C1, C2, C3)
        if u != 8:
            return u
        if C1 <= 64 and C2 <= 64:
            return 16
        return (32 if ns >= 64 else 0) if _all else 8
    _ops.sa_chain_table_unit = _unit_old
if os.environ.get("PC3D_SA_SPARSE") == "0":
    M("3dpointcloudattack_amd.ops").SA_BWD_SPARSE = False
if os.environ.get("PC3D_SA_PACKED") == "0":
    M("3dpointcloudattack_amd.ops").SA_BWD_PACKED = False
if os.environ.get("PC3D_REV_INDEX") == "0":
    M("3dpointcloudattack_amd.model.pointnet2_utils").REVERSE_INDEX = False
if os.environ.get("PC3D_EXP_DELAY_STEPS"):
    # EXPERIMENT: a delay on the main stream every iteration — one workgroup running a sampling chain of that many steps (~0.57 us
    # each) after the update launch. If the loop is bound by the device the iteration grows by the delay; if the device waits for
    # the host somewhere, less. (How much of an iteration is host-bound cannot be read off a profiled trace: the profiler slows the host.)
    _ops = M("3dpointcloudattack_amd.ops")
    _dummy = torch.rand(1, 4096, 3, device=dev)
    _steps = int(os.environ["PC3D_EXP_DELAY_STEPS"])
    _adam_old = _ops.adam_clip_step
    def _adam_delayed(*a, **k):
        r = _adam_old(*a, **k)
        _ops.fps(_dummy, _steps)
        return r
    _ops.adam_clip_step = _adam_delayed
res = {}
if "cw_curvenet" in which:
    B, N, IT = 32, 4096, 30
    net = mk("curvenet", "CurveNet", 0, num_classes=40)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
    with torch.no_grad():
        lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
    cw = M("3dpointcloudattack_amd.attack.CW.CW_attack")
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    cw.CW.dist_stream = os.environ.get("PC3D_CW_DIST_STREAM", "1") != "0"
    ts = []
    for it in (6, 6, 6 + IT, 6 + 2 * IT, 6 + 3 * IT):      # three slopes, the median is reported (as for the KNN attack)
        atk = cw.CW(net, net, adv.UntargetedLogitsAdvLoss(kappa=0.), cu.ClipPointsLinf(budget=0.18), du.ChamferDist(method='adv2ori'),
                    attack_lr=1e-2, binary_step=1, num_iter=it, graph=True if GRAPH is None else GRAPH)
        torch.manual_seed(0); np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        atk.attack(pcs, lab)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    slopes = [(ts[i + 1] - ts[i]) / IT * 1e3 for i in (1, 2, 3)]
    res["cw_curvenet_B32_N4096_ms_per_iter"] = sorted(slopes)[1]
    res["cw_curvenet_slopes_ms"] = slopes
    print(res, flush=True)
if "geoa3_curvenet" in which:
    B, N, IT = 32, 4096, 30
    net = mk("curvenet", "CurveNet", 0, num_classes=40)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
    with torch.no_grad():
        lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
    ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
    ts = []
    for it in (6, 6, 6 + IT, 6 + 2 * IT, 6 + 3 * IT):
        cfg = _geo_cfg(iter_max_steps=it, binary_max_steps=1, npoint=N, cls_loss_type='CE', hd_loss_weight=0.1, curv_loss_weight=1.0)
        cfg.graph_victim = GRAPH
        cfg.search_stream = os.environ.get("PC3D_GEOA3_SEARCH_STREAM", "1") != "0"
        torch.manual_seed(0); np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ga.geoA3_attack(net, None, None, None, None, None, pcs, lab, cfg, 0, 1)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    slopes = [(ts[i + 1] - ts[i]) / IT * 1e3 for i in (1, 2, 3)]
    res["geoa3_curvenet_B32_N4096_ms_per_iter"] = sorted(slopes)[1]
    res["geoa3_curvenet_slopes_ms"] = slopes
    print(res, flush=True)
if "geoa3" in which:
    B, N, IT = 32, 1024, 600      # long slopes: every call carries an eigh of ~0.12 s whose jitter (+-10 ms) must cancel
    net = mk("dgcnn", "DGCNN", 0, args=types.SimpleNamespace(k=20, emb_dims=1024, dropout=0.5), output_channels=40)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
    with torch.no_grad():
        lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
    ga = M("3dpointcloudattack_amd.attack.GeoA3.GeoA3_attack")
    for it in (10, 10, 10 + IT):
        cfg = _geo_cfg(iter_max_steps=it, binary_max_steps=1, npoint=N, cls_loss_type='CE', hd_loss_weight=0.1, curv_loss_weight=1.0)
        cfg.graph_victim = GRAPH
        cfg.search_stream = os.environ.get("PC3D_GEOA3_SEARCH_STREAM", "1") != "0"
        torch.manual_seed(0); np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ga.geoA3_attack(net, None, None, None, None, None, pcs, lab, cfg, 0, 1)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        res.setdefault("geoa3_t", []).append(t)
    res["geoa3_dgcnn_B32_N1024_ms_per_iter"] = (res["geoa3_t"][2] - res["geoa3_t"][1]) / IT * 1e3
    print(res, flush=True)
if "knn" in which:
    B, N, IT = 64, 2048, 60
    net = mk("pointnet2_SSG", "PointNet_Ssg", 0, num_classes=40)
    net.geometry_stream = os.environ.get("PC3D_SA_GEO_STREAM", "1") != "0"          # A/B switches
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
    with torch.no_grad():
        lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
    ka = M("3dpointcloudattack_amd.attack.KNN.KNN_attack")
    ka.CWKNN.dist_stream = os.environ.get("PC3D_KNN_DIST_STREAM", "1") != "0"
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
    cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    ts = []
    for it in (10, 10, 10 + IT, 10 + 2 * IT, 10 + 3 * IT):     # three slopes; the median is reported (a one-off stall in a
                                                               # short run moves a single difference by +-20 %)
        atk = ka.CWKNN(net, None, None, None, None, None, adv.UntargetedLogitsAdvLoss(kappa=15.), du.ChamferkNNDist(chamfer_method='adv2ori', knn_k=5, knn_alpha=1.05, chamfer_weight=5., knn_weight=3.),
                       cu.ProjectInnerClipLinf(budget=0.18), attack_lr=1e-2, num_iter=it)
        torch.manual_seed(0); np.random.seed(0)
        data = torch.cat([pcs, torch.nn.functional.normalize(pcs, dim=2)], dim=2)    # [B,N,6]: points + normals
        torch.cuda.synchronize(); t0 = time.perf_counter()
        atk.attack(data, lab)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    slopes = [(ts[i + 1] - ts[i]) / IT * 1e3 for i in (1, 2, 3)]
    res["knn_ssg_B64_N2048_ms_per_iter"] = sorted(slopes)[1]
    res["knn_ssg_slopes_ms"] = slopes
    print(res, flush=True)
if "aof" in which:
    B, N, IT = 32, 1024, 600      # long slopes: every call carries an eigh of ~0.12 s whose jitter (+-10 ms) must cancel
    net = mk("pointnet", "PointNetCls", 0, k=40)
    pcs = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)]))
    with torch.no_grad():
        lg = net(pcs.transpose(1, 2).contiguous().to(dev))[0]
    lab, tgt = lg.argmax(1).cpu(), lg.topk(2, dim=1)[1][:, 1].cpu()
    ta = M("3dpointcloudattack_amd.attack.AOF.TAOF_attack")
    adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
    ts = []
    for it in (10, 10, 10 + IT):
        atk = ta.CWTAOF(net, adv.LogitsAdvLoss(0.), None, attack_lr=1e-2, binary_step=1, num_iter=it, GAMMA=0.5, low_pass=100,
                        clip_func=cu.ClipPointsLinf(budget=0.18))
        torch.manual_seed(0); np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        atk.attack(pcs, tgt, lab)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res["taof_pointnet_B32_N1024_ms_per_iter"] = (ts[2] - ts[1]) / IT * 1e3
    res["taof_setup_ms_per_binary_step"] = ts[1] * 1e3 - 10 * res["taof_pointnet_B32_N1024_ms_per_iter"]
    print(res, flush=True)
print(json.dumps(res))
