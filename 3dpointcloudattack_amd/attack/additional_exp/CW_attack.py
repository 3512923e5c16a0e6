"""CW with the binary-searched point-set distance, z-only perturbation, renormalisation in the loop and expectation
over small random rotations / resampling — MI355X mirror of attack/additional_exp/CW_attack.py.

Same constructor and ``attack(data, target, origin_label)`` signature and return values as the reference (:14-18,
:47, :321). This is the caller that exercises the Chamfer / Chamfer+kNN regularisers as THE binary-searched distance
(:151-159): with this package's functors they are the fused nearest-neighbour / kNN HIP kernels, so no [B,K,K]
matrix is formed in any of the 1 + 10 forwards of an iteration.

Protocol kept from the reference:
  * ``adv_func(logits, label, whether_target=1|0)`` (:245-247,:254,:259) — note that no functor in the reference
    repository accepts that keyword; ``AdvLossAdapter`` below wraps the CW_utils.adv_utils functors;
  * ``dist_func(adv[B,K,3], ori[B,K,3], weights[B])`` (:151-153), scalar or per-sample result;
  * ``model(x[B,3,K]) -> (logits, _, _)`` (:75,:117);
  * random streams: the start noise and the rotation angles come from torch's global CPU generator (:88,:196), the
    rotation choice and the resampling indices from Python's ``random`` (:211,:238) — drawn here in the same order,
    so a seeded run consumes the same streams as the reference.
What changes: batches B > 1 work (the reference prints ``.item()`` and constrains only sample 0, :80-84,:268-275 —
here the z-only / box constraint applies to every sample); bookkeeping stays on the device (one host read per
binary-search step); the resampling draws ``K`` of ``2K`` indices (the reference hard-codes 4000 of 8000, :238).
"""
import random

import numpy as np
import torch
import torch.optim as optim

from ... import graphed as _graphed
from ... import ops


class AdvLossAdapter:
    """Gives a CW_utils.adv_utils-style functor pair the ``whether_target`` keyword this attack calls with (:245-259)."""

    def __init__(self, targeted_func, untargeted_func=None):
        self.targeted_func = targeted_func
        self.untargeted_func = untargeted_func if untargeted_func is not None else targeted_func

    def __call__(self, logits, label, whether_target=1):
        return (self.targeted_func if whether_target else self.untargeted_func)(logits, label)


def _renormalize(x):
    """:107-115 / :224-230 — centre and scale [B,3,K] to the unit sphere (differentiable, like the reference)."""
    p = x.permute(0, 2, 1)
    p = p - torch.mean(p, dim=1).unsqueeze(1)
    var = torch.max(torch.sqrt(torch.sum(p ** 2, dim=2)), dim=1, keepdim=True)[0]
    return (p / var.unsqueeze(1)).permute(0, 2, 1)


def _small_rotation(dev):
    """:195-219 — one draw of (theta ~ 1e-2 N(0,1), axis choice): z / x / y rotation with probability 0.2 each, else
    the identity. Consumes torch.randn(1) THEN random.random(), like the reference."""
    theta = torch.randn(1) * 1e-2
    c, s = float(torch.cos(theta)), float(torch.sin(theta))
    r = random.random()
    if r < 0.2:
        m = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    elif r < 0.4:
        m = [[1, 0, 0], [0, c, s], [0, -s, c]]
    elif r < 0.6:
        m = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    else:
        m = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    return ops.h2d(torch.tensor(m, dtype=torch.float32), dev).unsqueeze(0)


class CW:
    """Class for CW attack (additional experiments)."""

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2,
                 init_weight=10., max_weight=80., binary_step=10, num_iter=500, whether_target=True, whether_1d=True,
                 whether_renormalization=False,
                 whether_3Dtransform=False, whether_resample=False, device=None, verbose=False, graph=True):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.model = model.to(self.device)
        self.model.eval()
        self.model = _graphed.wrap(self.model, enable=graph)   # hipGraph replay for deterministic victims
        self.adv_func = adv_func
        self.dist_func = dist_func
        self.attack_lr = attack_lr
        self.init_weight = init_weight
        self.max_weight = max_weight
        self.binary_step = binary_step
        self.num_iter = num_iter
        self.whether_target = whether_target
        self.whether_1d = whether_1d
        self.whether_renormalization = whether_renormalization
        self.whether_3Dtransform = whether_3Dtransform
        self.box_constraint = 0.4
        self.whether_resample = whether_resample
        self.verbose = verbose

    def _forward(self, x):
        out = self.model(x)
        return out[0] if isinstance(out, tuple) else out

    def attack(self, data, target=torch.Tensor([0]), origin_label=torch.Tensor([105])):
        """data [B,K,3] (or [B,3,K]); target [B] (targeted) / origin_label [B] (untargeted).
        Returns (o_bestdist [B] float64, o_bestattack [B,K,3] float64, success_num) like the reference (:321)."""
        dev = self.device
        if data.shape[2] == 3:
            data = data.transpose(1, 2).contiguous()
        B, _, K = data.shape
        data = data.float().to(dev).detach()
        ori_data = data.clone().detach()
        target = target.long().to(dev).detach().view(-1)
        origin_label = origin_label.detach().to(dev).long().view(-1)
        if target.numel() == 1 and B > 1:
            target = target.expand(B)
        if origin_label.numel() == 1 and B > 1:
            origin_label = origin_label.expand(B)
        goal = target if self.whether_target else origin_label

        lower_bound = np.zeros((B,))
        upper_bound = np.ones((B,)) * self.max_weight
        current_weight = np.ones((B,)) * self.init_weight

        o_bestdist = torch.full((B,), 1e10, dtype=torch.float64, device=dev)
        o_bestscore = torch.full((B,), -1, dtype=torch.long, device=dev)
        o_bestattack = torch.zeros((B, 3, K), dtype=torch.float32, device=dev)

        with torch.no_grad():
            pred = torch.argmax(self._forward(ori_data), dim=1)
        if self.verbose:
            print('clean pred: {}  goal: {}'.format(pred.tolist(), goal.tolist()))

        # ONE start point for the whole search: later binary steps continue from the previous step's iterate (:88-93)
        adv_data = ori_data.clone().detach() + torch.randn((B, 3, K)).to(dev) * 1e-7
        input_val = adv_data.detach().clone()
        for binary_step in range(self.binary_step):
            adv_data.requires_grad_()
            bestdist = torch.full((B,), 1e10, dtype=torch.float64, device=dev)
            bestscore = torch.full((B,), -1, dtype=torch.long, device=dev)
            opt = optim.Adam([adv_data], lr=self.attack_lr, weight_decay=0.)
            weights = torch.from_numpy(current_weight)

            for iteration in range(self.num_iter):
                logits = self._forward(_renormalize(adv_data) if self.whether_renormalization else adv_data)
                pred = torch.argmax(logits, dim=1)
                dist_loss = self.dist_func(adv_data.permute(0, 2, 1), ori_data.permute(0, 2, 1), weights)

                # record values (device side; reference :155-182)
                with torch.no_grad():
                    dist_val = dist_loss.detach().double().reshape(-1).expand(B)
                    succ = (pred == goal) if self.whether_target else (pred != goal)
                    upd = succ & (dist_val < bestdist)
                    bestdist = torch.where(upd, dist_val, bestdist)
                    bestscore = torch.where(upd, pred, bestscore)
                    upd_o = succ & (dist_val < o_bestdist)
                    o_bestdist = torch.where(upd_o, dist_val, o_bestdist)
                    o_bestscore = torch.where(upd_o, pred, o_bestscore)
                    input_val = adv_data.detach().clone()
                    o_bestattack = torch.where(upd_o[:, None, None], input_val, o_bestattack)
                dist_loss = dist_loss.mean()

                wt = 1 if self.whether_target else 0
                if self.whether_3Dtransform:
                    # expectation over 10 random small rotations of the CLEAN cloud carrying the same perturbation
                    diff = adv_data - ori_data.detach()
                    losses = []
                    for _ in range(10):
                        Tr = _small_rotation(dev)
                        x = torch.bmm(Tr.expand(B, -1, -1), ori_data.detach()) + diff
                        if self.whether_renormalization:
                            x = _renormalize(x)
                        if self.whether_resample:
                            # the cloud twice, K of its 2K columns drawn without replacement (:236-239: index 0 excluded)
                            indices = random.sample(range(1, K * 2), K)
                            x = torch.index_select(torch.cat((x, x), 2), 2, ops.h2d(torch.tensor(indices, dtype=torch.long), dev))
                        losses.append(self.adv_func(self._forward(x), goal, whether_target=wt).mean())
                    adv_loss = torch.mean(torch.stack(losses))
                else:
                    adv_loss = self.adv_func(logits, goal, whether_target=wt).mean()
                loss = adv_loss + dist_loss
                if self.verbose and iteration % 10 == 0:
                    print('Step {}, iteration {}, success {}/{}  adv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, iteration, int(succ.sum()), B, float(adv_loss), float(dist_loss)))

                opt.zero_grad()
                loss.backward()
                opt.step()

                if self.whether_1d:
                    # attack on the z direction only, inside a +-0.4 box (:266-275)
                    with torch.no_grad():
                        adv_data[:, 0] = ori_data[:, 0]
                        adv_data[:, 1] = ori_data[:, 1]
                        adv_data[:, 2] = torch.max(torch.min(adv_data[:, 2], ori_data[:, 2] + self.box_constraint),
                                                   ori_data[:, 2] - self.box_constraint)

            # adjust weight factor (:283-303) — one host read per binary-search step
            bs, bd, obd = bestscore.cpu().numpy(), bestdist.cpu().numpy(), o_bestdist.cpu().numpy()
            goal_np = goal.cpu().numpy()
            for e in range(B):
                hit = (bs[e] == goal_np[e]) if self.whether_target else (bs[e] != goal_np[e])
                if hit and bs[e] != -1 and bd[e] <= obd[e]:
                    lower_bound[e] = max(lower_bound[e], current_weight[e])
                else:
                    upper_bound[e] = min(upper_bound[e], current_weight[e])
                current_weight[e] = (lower_bound[e] + upper_bound[e]) / 2.

        # fail to attack some examples: assign them the iterate the last pass started from (:309-310)
        fail_idx = torch.from_numpy(lower_bound == 0.).to(dev)
        o_bestattack = torch.where(fail_idx[:, None, None], input_val, o_bestattack)
        success_num = int((lower_bound > 0.).sum())
        if self.verbose:
            print('Successfully attack {}/{}   pred: {}'.format(success_num, B, pred.tolist()))
        return (o_bestdist.cpu().numpy(), o_bestattack.double().cpu().numpy().transpose((0, 2, 1)), success_num)
