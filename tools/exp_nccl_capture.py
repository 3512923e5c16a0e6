"""Does hipGraph capture (CW runner, GraphedVictim) work while an RCCL process group (and its watchdog thread) is alive?
Single rank, nccl backend, world_size 1."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
from helpers import unit_cloud
M = importlib.import_module
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(1024, device=dev); dist.broadcast(t, src=0); dist.all_reduce(t); torch.cuda.synchronize()
seeded_state_dict = M("3dpointcloudattack_amd.seeding").seeded_state_dict
net = M("3dpointcloudattack_amd.model.pointnet").PointNetCls(k=40)
net.load_state_dict(seeded_state_dict(net, 0)); net = net.to(dev).eval()
cw = M("3dpointcloudattack_amd.attack.CW.CW_attack")
adv = M("3dpointcloudattack_amd.attack.CW.CW_utils.adv_utils"); du = M("3dpointcloudattack_amd.attack.CW.CW_utils.dist_utils")
cu = M("3dpointcloudattack_amd.attack.CW.CW_utils.clip_utils")
rng = np.random.default_rng(0)
pcs = torch.from_numpy(np.stack([unit_cloud(rng, 1024) for _ in range(32)]))
with torch.no_grad():
    lab = net(pcs.transpose(1, 2).contiguous().to(dev))[0].argmax(1).cpu()
for rep in range(3):
    atk = cw.CW(net, net, adv.UntargetedLogitsAdvLoss(kappa=30.), cu.ClipPointsLinf(budget=0.18), du.ChamferDist(method='adv2ori'),
                attack_lr=1e-2, binary_step=2, num_iter=50)
    t0 = time.perf_counter(); atk.attack(pcs, lab); torch.cuda.synchronize()
    dist.all_reduce(t)
    print("CW attack with graph capture under nccl ok", rep, round(time.perf_counter() - t0, 3), flush=True)
gv = M("3dpointcloudattack_amd.graphed").wrap(net)
x = pcs.transpose(1, 2).contiguous().to(dev).requires_grad_()
for i in range(3):
    gv(x)[0].sum().backward()
    with torch.no_grad(): gv(x)
dist.barrier(); torch.cuda.synchronize()
print("GraphedVictim under nccl ok", gv.stats, flush=True)
dist.destroy_process_group()
