"""Stub of timm.models.layers: DropPath is identity at inference, to_2tuple, trunc_normal_."""
import torch.nn as nn


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Identity):
    def __init__(self, drop_prob=0.0, *a, **k):
        super().__init__()
