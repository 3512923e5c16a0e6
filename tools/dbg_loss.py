import importlib, sys, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("3dpointcloudattack_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(3)
z = torch.randn(9, 40, device=dev) * 3
tgt = torch.randint(0, 40, (9,), device=dev)
for kind in ("untargeted_logits", "cross_entropy"):
    logp, pred, loss, g = ops.cls_loss(z, tgt, kind, 2.0, scale=1/9)
    torch.cuda.synchronize()
    print(kind, loss, -logp[torch.arange(9), tgt])
