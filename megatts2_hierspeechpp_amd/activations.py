"""SnakeBeta parameters (reference: activations.py:79-119).  The arithmetic runs inside
the fused kernels; this module only owns the ``alpha`` / ``beta`` parameters so that the
checkpoint keys (``...activations.N.act.alpha|beta``) match."""
import torch
from torch import nn


class SnakeBeta(nn.Module):
    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        if not alpha_logscale:
            raise NotImplementedError("only SnakeBeta(alpha_logscale=True) is instantiated on the hot path")
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        self.alpha = nn.Parameter(torch.zeros(in_features), requires_grad=False)
        self.beta = nn.Parameter(torch.zeros(in_features), requires_grad=False)
        self.no_div_by_zero = 0.000000001
