"""MI355X mirror of attack/CW/CW_utils/adv_utils.py — adversarial (margin / CE) losses on the logits [B,k].
Tiny [B,k] tensors: expressed with device-side torch ops (no host sync, no per-call H2D like the reference's
torch.zeros(B,K).cuda() at adv_utils.py:27,74)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _kernel_loss(logits, targets, kind, kappa=0.):
    """The functor on the HIP kernel (pc3d_cls_loss_f32, raw mode: the loss on `logits` as given), or None when the
    inputs are not what it takes. One launch forward, one backward, instead of ~10 + ~10 ATen launches on [B,k]."""
    if not (torch.is_tensor(logits) and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2
            and logits.stride(1) == 1 and 2 <= logits.shape[1] <= 64 and torch.is_tensor(targets) and targets.is_cuda
            and targets.numel() == logits.shape[0]):
        return None
    from .... import ops
    return ops.adv_loss_raw(logits, targets.reshape(-1).long(), kind, kappa).mean()


def _kernel_term(logits, targets, kind, kappa, up):
    """Per-sample loss [B] whose gradient is fixed by the caller: the functor's value enters the attack's loss as
    `up * mean_b(term_b)`, so d loss / d term_b = up / B is multiplied into the gradient by the forward launch and the
    backward passes it on untouched (no mean / scale launches either way). None when the kernel does not take the inputs."""
    if not (torch.is_tensor(logits) and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2
            and logits.stride(1) == 1 and 2 <= logits.shape[1] <= 64 and torch.is_tensor(targets) and targets.is_cuda
            and targets.numel() == logits.shape[0]):
        return None
    import numpy as np
    from .... import ops
    return ops.adv_loss_raw(logits, targets.reshape(-1).long(), kind, kappa,
                            gscale=np.float32(up) / np.float32(logits.shape[0]))


def _real_other(logits, targets):
    B, K = logits.shape
    if len(targets.shape) == 1:
        targets = targets.view(-1, 1)
    targets = targets.long()
    one_hot = torch.zeros(B, K, device=logits.device).scatter_(1, targets, 1).float()
    real = torch.sum(one_hot * logits, dim=1)
    other = torch.max((1. - one_hot) * logits - one_hot * 10000., dim=1)[0]
    return real, other


class LogitsAdvLoss(nn.Module):
    """adv_utils.py:6-33 — targeted margin loss clamp(other - real + kappa, 0).mean()."""

    def __init__(self, kappa=0.):
        super(LogitsAdvLoss, self).__init__()
        self.kappa = kappa

    def per_sample(self, logits, targets, up=1.0):
        """[B] terms with d loss / d term = up / B built in (see _kernel_term), or None."""
        return _kernel_term(logits, targets, "logits", self.kappa, up)

    def forward(self, logits, targets):
        fast = _kernel_loss(logits, targets, "logits", self.kappa)
        if fast is not None:
            return fast
        real, other = _real_other(logits, targets)
        return torch.clamp(other - real + self.kappa, min=0.).mean()


class CrossEntropyAdvLoss(nn.Module):
    """adv_utils.py:36-51 — nll_loss on the model's log-probabilities."""

    def __init__(self):
        super(CrossEntropyAdvLoss, self).__init__()

    def per_sample(self, logits, targets, up=1.0):
        """[B] terms with d loss / d term = up / B built in (see _kernel_term), or None."""
        return _kernel_term(logits, targets, "cross_entropy", 0., up)

    def forward(self, logits, targets):
        fast = _kernel_loss(logits, targets, "cross_entropy")
        if fast is not None:
            return fast
        return F.nll_loss(logits, targets)


class UntargetedLogitsAdvLoss(nn.Module):
    """adv_utils.py:53-80 — untargeted margin loss clamp(real - other + kappa, 0).mean()."""

    def __init__(self, kappa=0.):
        super(UntargetedLogitsAdvLoss, self).__init__()
        self.kappa = kappa

    def per_sample(self, logits, targets, up=1.0):
        """[B] terms with d loss / d term = up / B built in (see _kernel_term), or None."""
        return _kernel_term(logits, targets, "untargeted_logits", self.kappa, up)

    def forward(self, logits, targets):
        fast = _kernel_loss(logits, targets, "untargeted_logits", self.kappa)
        if fast is not None:
            return fast
        real, other = _real_other(logits, targets)
        return torch.clamp(real - other + self.kappa, min=0.).mean()
