import torch.nn as nn


class DropPath(nn.Identity):
    def __init__(self, drop_prob=0.0):
        super().__init__()


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)
