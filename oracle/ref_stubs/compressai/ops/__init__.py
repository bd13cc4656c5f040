import torch
import torch.nn as nn


class _LowerBoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        keep = (x >= bound) | (g < 0)
        return keep * g, None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)
