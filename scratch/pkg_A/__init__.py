"""MI355X-native hot path of the variance-aware-masking progressive image codec.

Import as ``vampic`` (the directory name required by the build contract contains
hyphens, so the repo-root ``vampic`` package aliases it):

    from vampic import get_model, VarianceMaskingPIC, VarianceMaskingPICREM

Host code is Python on PyTorch-ROCm (memory, streams); every arithmetic op is a
hand-written gfx950 kernel in ``libvampic.so`` behind the C ABI of ``include/vampic.h``.
"""
from . import _lib                                                  # noqa: F401
from .models import (VarianceMaskingPIC, VarianceMaskingPICREM, get_model, models,   # noqa: F401
                     define_encoder, define_decoder, define_hyperprior)
from .entropy_models import EntropyBottleneck, GaussianConditional, get_scale_table  # noqa: F401
from .layers import (ChannelMask, GDN, Win_noShift_Attention, WinBasedAttention, WindowAttention,   # noqa: F401
                     ResidualUnit, ResidualBlock, LatentRateReduction, conv, deconv, conv1x1, conv3x3,
                     subpel_conv3x3)
from . import synth                                                 # noqa: F401

__all__ = ["get_model", "models", "VarianceMaskingPIC", "VarianceMaskingPICREM", "ChannelMask",
           "GaussianConditional", "EntropyBottleneck"]
