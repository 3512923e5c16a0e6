"""kNN attack (AAAI'20) — MI355X mirror of attack/KNN/KNN_attack.py: one long Adam run (no binary search) with a
Chamfer (+ optional kNN-distance) regulariser and inner-point projection + per-point clip.

Same constructor / ``attack(data, target)`` signature and return values (attack/KNN/KNN_attack.py:19-20,56,244-246).
The loop (:99-141) runs on the device without host synchronisation: the per-iteration ``.item()`` success count
(:113-115) is dropped from the loop (it only fed a discarded local), Adam + ProjectInnerClipLinf are ONE HIP launch
when the clip functor is this package's own, and the regularisers are the fused NN / kNN kernels. B > 1 works; the
transfer / fail counters are sums over the batch. Transfer models may be ``None`` (skipped).
"""
import numpy as np
import torch
import torch.optim as optim

from ... import graphed as _graphed
from ... import ops
from ... import streams as _streams
from ..CW.CW_utils import clip_utils as _clip_utils
from ..CW.CW_utils import dist_utils as _dist_utils


def rand_row(array, dim_needed):
    """attack/KNN/KNN_attack.py:8-13."""
    row_total = array.shape[0]
    row_sequence = np.arange(row_total)
    np.random.shuffle(row_sequence)
    return array[row_sequence[0:dim_needed], :]


def _logits_of(out):
    return out[0] if isinstance(out, tuple) else out


class CWKNN:
    """Class for CW attack."""

    def __init__(self, model, pt_model, ptm_model, pts_model, dgcnn_model, cur_model, adv_func, dist_func, clip_func,
                 attack_lr=1e-3, num_iter=2500, attack_method='untarget', device=None, verbose=False, fused=True,
                 graph=True, sample_seeds=None, global_batch=None, deterministic=None):
        """Extra keywords (defaults = the reference's behaviour). For sharded runs: `sample_seeds` (one int
        per sample) draws every sample's start noise AND a PointNet++ victim's FPS start indices from that sample's own
        generator instead of the shared global stream, and `global_batch` is the size of the unsharded batch whose loss
        mean the shard is a part of — with both, a sample's trajectory does not depend on the batch or rank it runs in
        (bit for bit: the backward kernels sum in a fixed order, ops.DETERMINISTIC)."""
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

        def prep(m):
            if m is None:
                return None
            m = m.to(self.device)
            m.eval()
            return m

        # a victim with a deterministic forward replays its forward/backward from hipGraphs (graphed.py)
        self.model = _graphed.wrap(prep(model), enable=graph)
        self.pt_model = prep(pt_model)
        self.ptm_model = prep(ptm_model)
        self.pts_model = prep(pts_model)
        self.dgcnn_model = prep(dgcnn_model)
        self.cur_model = prep(cur_model)
        self.adv_func = adv_func
        self.dist_func = dist_func
        self.clip_func = clip_func
        self.attack_lr = attack_lr
        self.num_iter = num_iter
        self.attack_method = attack_method
        self.shuffle_fail = 0
        self.trans_fail = 0
        self.attack_fail = 0
        self.pt_fail = 0
        self.ptm_fail = 0
        self.pts_fail = 0
        self.dgcnn_fail = 0
        self.cur_fail = 0
        self.verbose = verbose
        self.fused = fused
        self.sample_seeds = sample_seeds
        self.global_batch = global_batch
        self.deterministic = deterministic     # None: ops.DETERMINISTIC; True / False: that mode during attack()

    def _success(self, pred, target):
        return (pred != target) if self.attack_method == 'untarget' else (pred == target)

    def _fused_clip(self):
        """(budget, use_normal) for a recognised clip functor, else None."""
        cf = self.clip_func
        if not self.fused or cf is None:
            return None
        if type(cf) is _clip_utils.ProjectInnerClipLinf:
            return float(cf.budget), True
        if type(cf) in (_clip_utils.ClipPointsLinf, _dist_utils.ClipPointsLinf):
            return float(cf.budget), False
        return None

    def attack(self, data, target):
        """Attack on given data to target.
        Args:
            data (torch.FloatTensor): victim data, [B, num_points, 3] (or 6 with normals)
            target (torch.LongTensor): target output, [B]
        Returns (adv [B,K,3] float32 numpy, success_num) like the reference (:244-246).
        """
        if self.deterministic is not None and self.deterministic != ops.DETERMINISTIC:
            with ops.deterministic(self.deterministic):
                return self.attack(data, target)
        if self.sample_seeds is None:
            return self._attack(data, target, None)
        from ...model import pointnet2_utils as _pn2
        assert len(self.sample_seeds) == data.shape[0], "sample_seeds needs one seed per sample"
        prev = _pn2.set_fps_start_source(_pn2.SeededFpsStarts([int(s) + 7919 for s in self.sample_seeds]))
        try:
            return self._attack(data, target, [torch.Generator().manual_seed(int(s)) for s in self.sample_seeds])
        finally:
            _pn2.set_fps_start_source(prev)

    def _attack(self, data, target, gens):
        dev = self.device
        B, K = data.shape[:2]
        ratio = (float(B) / float(self.global_batch)) if self.global_batch else 1.0
        data = data.float().to(dev).detach()
        data = data.transpose(1, 2).contiguous()
        ori_data = data.clone().detach()
        ori_data.requires_grad = False

        # points and normals (:68-73; with xyz only the positions double as "normals", SURVEY A-11)
        if ori_data.shape[1] == 3:
            normal = ori_data
        else:
            normal = ori_data[:, 3:, :]
            ori_data = ori_data[:, :3, :]

        # clean forward (:76-78). Kept even when nothing is printed: a PointNet++ victim draws its FPS start indices
        # from the global RNG on every forward, so skipping it would shift the stream the reference sees.
        with torch.no_grad():
            clean_logits = _logits_of(self.model(ori_data))
            clean_pred = torch.argmax(clean_logits, dim=1)
        if self.verbose:
            print("ori label:", clean_pred.tolist())
        target = target.long().to(dev).detach().view(-1)

        # init variables with small perturbation (CPU generator like the reference, :84-85)
        noise = torch.randn((B, 3, K)) if gens is None else torch.stack([torch.randn((3, K), generator=g) for g in gens])
        adv_data = ori_data.clone().detach() + noise.to(dev) * 1e-7
        adv_data.requires_grad_()
        fc = self._fused_clip()
        if fc is None:
            opt = optim.Adam([adv_data], lr=self.attack_lr, weight_decay=0.)
        else:
            exp_avg, exp_avg_sq = torch.zeros_like(adv_data), torch.zeros_like(adv_data)
        ori_t = ori_data.transpose(1, 2).contiguous()
        # Own functors on both sides: the loss is never assembled. Each functor hands back its per-sample terms with
        # d loss / d term built in (per_sample / per_sample_terms: the batch means, `* K`, the functor's own weights and the
        # shard ratio are constants), the backward starts from the terms, and the distance branch differentiates an ALIAS
        # of the iterate so that its gradient arrives in a buffer of its own, summed inside the Adam launch — none of the
        # ~25 mean / scale / add / accumulate launches of `loss = adv.mean() + dist.mean() * K; loss.backward()`.
        own_dist = (_dist_utils.ChamferDist, _dist_utils.HausdorffDist, _dist_utils.ChamferkNNDist)
        direct = (fc is not None and adv_data.is_cuda and type(self.dist_func) in own_dist
                  and hasattr(self.adv_func, "per_sample") and getattr(self, "direct_terms", True)
                  and clean_logits.dim() == 2 and 2 <= clean_logits.shape[1] <= 64 and clean_logits.dtype == torch.float32)
        if direct:
            adv_alias = adv_data.detach().requires_grad_()           # same storage: follows the in-place updates
            up_adv = np.float32(ratio)
            up_dist = np.float32(ratio) * np.float32(K) if ratio != 1.0 else np.float32(K)
            ones = ops.const_vec(dev, B, 1.0)

        # The distance term does not depend on the victim: its searches (kNN + Chamfer) run on a side stream beside the
        # victim's forward, whose first kernels (farthest-point sampling: one workgroup per cloud) leave most of the
        # chip idle. Autograd runs each node's backward on its forward's stream, so the two backwards overlap as well.
        cur = torch.cuda.current_stream(dev) if adv_data.is_cuda else None
        side = (_streams.side_stream(dev, _streams.TERMS)
                if cur is not None and getattr(self, "dist_stream", True) and getattr(self.model, "sampling_chain_front", False)
                else None)                      # ONE per process (see streams.py); only beside a sampling-chain victim
        # A PointNet++ victim draws an FPS start per sampling layer and forward from the CPU generator (SURVEY A-4): the
        # loop's draws are made here, in the same order, and uploaded once (pointnet2_utils.PredrawnFpsStarts)
        from ...model import pointnet2_utils as _pn2
        victim = getattr(self.model, "model", self.model)
        predrawn = None
        if adv_data.is_cuda and hasattr(victim, "sampling_input_sizes") and getattr(self, "predraw_starts", True):
            predrawn = _pn2.PredrawnFpsStarts(victim.sampling_input_sizes(K), B, self.num_iter, dev, inner=_pn2.fps_start_source())
            prev_source = _pn2.set_fps_start_source(predrawn)
        try:
            for iteration in range(self.num_iter):
                if direct:
                    if side is not None:
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            terms = self.dist_func.per_sample_terms(adv_alias, ori_data, up_dist)   # channel-first, zero-copy
                    logits = _logits_of(self.model(adv_data))
                    a_term = self.adv_func.per_sample(logits.contiguous(), target, up_adv)
                    if a_term is None:
                        raise RuntimeError("CWKNN: the victim's logits changed shape / dtype between forwards")
                    if side is not None:
                        cur.wait_stream(side)
                    else:
                        terms = self.dist_func.per_sample_terms(adv_alias, ori_data, up_dist)
                    adv_data.grad = None
                    adv_alias.grad = None
                    torch.autograd.backward([a_term] + terms, [ones] * (1 + len(terms)))
                    ops.adam_clip_step(adv_data.data, adv_data.grad, exp_avg, exp_avg_sq, iteration + 1, self.attack_lr,
                                       ori=ori_data, normal=normal if fc[1] else None, budget=fc[0], g2=adv_alias.grad)
                    continue
                if side is not None:
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        # in the official tensorflow code they use sum instead of mean, hence * K (:119-123)
                        dist_loss = self.dist_func(adv_data.transpose(1, 2).contiguous(), ori_t).mean() * K
                logits = _logits_of(self.model(adv_data))  # [B, num_classes]
                adv_loss = self.adv_func(logits, target).mean()
                if side is not None:
                    cur.wait_stream(side)
                else:
                    dist_loss = self.dist_func(adv_data.transpose(1, 2).contiguous(), ori_t).mean() * K
                loss = adv_loss + dist_loss
                if ratio != 1.0:
                    loss = loss * ratio            # batch means (:117-123) of a shard of a larger batch
                if fc is None:
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
                    adv_data.data = self.clip_func(adv_data.clone().detach(), ori_data, normal)
                else:
                    adv_data.grad = None
                    loss.backward()
                    ops.adam_clip_step(adv_data.data, adv_data.grad, exp_avg, exp_avg_sq, iteration + 1, self.attack_lr,
                                       ori=ori_data, normal=normal if fc[1] else None, budget=fc[0])
        finally:
            if predrawn is not None:
                _pn2.set_fps_start_source(prev_source)

        # end of CW attack
        with torch.no_grad():
            pred = torch.argmax(_logits_of(self.model(adv_data)), dim=-1)  # [B]
            success_num = int(self._success(pred, target).sum().item())
            if self.verbose:
                print('Successfully attack {}/{}'.format(success_num, B))
            result = adv_data.detach().float()
            # Test attack + transfer models (:160-240)
            self.attack_fail += int((~self._success(torch.argmax(_logits_of(self.model(result)), dim=1), target)).sum().item())
            for name, m in (("pt_fail", self.pt_model), ("ptm_fail", self.ptm_model), ("pts_fail", self.pts_model),
                            ("dgcnn_fail", self.dgcnn_model), ("cur_fail", self.cur_model)):
                if m is None:
                    continue
                p = torch.argmax(_logits_of(m(result)), dim=1)
                setattr(self, name, getattr(self, name) + int((~self._success(p, target)).sum().item()))

        adv_np = adv_data.detach().transpose(1, 2).contiguous().cpu().numpy()  # [B, K, 3]
        return adv_np, success_num
