"""Stand-in for `pytorch_lightning.seed_everything` (test infrastructure only)."""
import random

import numpy as np
import torch


def seed_everything(seed=0, workers=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed
