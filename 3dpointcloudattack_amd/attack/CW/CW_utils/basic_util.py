"""Mirror of the helpers of attack/CW/CW_utils/basic_util.py that the attack drivers import."""
import random

import numpy as np
import torch


def np2torch(tensor, device='cuda'):
    if isinstance(tensor, list):
        return [torch.from_numpy(t).to(device) for t in tensor]
    return torch.from_numpy(tensor).to(device)


def set_seed(seed=1):
    """basic_util.py:50-55."""
    print('Using random seed', seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def str2bool(v):
    return v.lower() in ("yes", "true", "t", "1")
