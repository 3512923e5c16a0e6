"""Host time of hipGraphLaunch for the captured CW/PointNet iteration graphs (1 and 4 iterations) vs their GPU time."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
M = importlib.import_module
dev = torch.device("cuda:0")
bench = M("bench")
PointNetCls = M("3dpointcloudattack_amd.model.pointnet").PointNetCls
CW = M("3dpointcloudattack_amd.attack.CW.CW_attack").CW
adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
model = PointNetCls(k=40, feature_transform=False); model.load_state_dict(bench.seeded_state(model, 0)); model = model.to(dev).eval()
rng = np.random.default_rng(1); data = torch.from_numpy(np.stack([bench.unit_cloud(rng, 1024) for _ in range(32)]))
with torch.no_grad(): labels = model(data.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
atk = CW(model, model, adv.UntargetedLogitsAdvLoss(kappa=30.), cu.ClipPointsLinf(budget=0.18), du.ChamferDist(), attack_lr=1e-2, binary_step=10, num_iter=500, device=dev)
st = atk._begin(data, labels); atk._begin_binary_step(st); run = atk._make_runner(st)
for g, name, its in [(run.graphs[n], "g%d" % n, n) for n in sorted(run.graphs)]:
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    hs = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 20
    print(name, "iterations", its, "host us per launch (GPU idle)", round(np.median(hs) * 1e6, 1), "steady us per launch", round(tot * 1e6, 1))

# where the fixed cost of a short timed region comes from: 20 iterations, host clock vs GPU events
for trial in range(3):
    for _ in range(5): run()
    run.flush(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20): run()
    run.flush()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("20 iterations: host enqueue ms", round((t1 - t0) * 1e3, 3), "host total ms", round((t2 - t0) * 1e3, 3), "GPU events ms", round(e0.elapsed_time(e1), 3))
