"""Host-side mirror of reference ``src/entropy_models/entropy_models.py``.

``GaussianConditional`` and ``EntropyBottleneck`` keep the reference's parameter /
buffer names; likelihood, quantisation and index building run as libvampic kernels.
The bitstream layer (``update`` CDF tables, ``compress`` / ``decompress``)
runs the rANS coder of ``csrc/rans.cpp`` on the host, where the reference runs compressai's.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import bitstream as bs
from . import ops
from .layers import _LowerBound, _no_autograd



class _EbAuxLoss(torch.autograd.Function):
    """loss and d loss / d quantiles in one launch of vam_eb_aux_loss."""

    @staticmethod
    def forward(ctx, quantiles, eb):
        import ctypes as C
        L.require_gpu()
        assert quantiles.is_cuda, "the entropy bottleneck's auxiliary loss runs on the GPU (no CPU fallback)"
        pp = eb.packed_params()
        loss = torch.zeros(1, dtype=torch.float64, device=quantiles.device)
        dq = torch.empty(quantiles.numel(), dtype=torch.float32, device=quantiles.device)
        tgt = (C.c_float * 3)(*[float(v) for v in eb.target.detach().cpu().tolist()])
        L.check(L.load().vam_eb_aux_loss(pp.data_ptr(), eb.channels, tgt, loss.data_ptr(), dq.data_ptr(), ops.stream_ptr()),
                "vam_eb_aux_loss")
        ctx.save_for_backward(dq)
        ctx.shape = quantiles.shape
        return loss.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return g * dq.view(ctx.shape), None


EB_PARAM_NAMES = ("_matrix0", "_bias0", "_factor0", "_matrix1", "_bias1", "_factor1", "_matrix2", "_bias2", "_factor2",
                  "_matrix3", "_bias3", "_factor3", "_matrix4", "_bias4", "quantiles")


class _EbTrainFn(torch.autograd.Function):
    """Training-mode entropy bottleneck as a module-level call (entropy_models.py:449-492 with ``training=True``):
    outputs = x + U(-1/2, 1/2), likelihood of the noisy value with compressai's LowerBound(1e-9).  Forward
    vam_eb_forward_noise, backward vam_eb_train_bwd (dx and the 15 tensors of the density network)."""

    @staticmethod
    def forward(ctx, x, noise, *params):
        L.require_gpu()
        z, nz = ops.from_nchw(x.detach()), ops.from_nchw(noise)
        pp = torch.cat([p.detach().reshape(-1).float() for p in params]).contiguous()
        lik = ops.new_view(z.B, z.H, z.W, z.C, x.device)
        zhat = ops.new_view(z.B, z.H, z.W, z.C, x.device)
        ops.eb_forward(z, pp, zhat, lik, None, noise=nz)
        ctx.z, ctx.nz, ctx.pp = z, nz, pp
        ctx.shapes = [p.shape for p in params]
        return x.detach() + noise, lik.torch_nchw().contiguous()

    @staticmethod
    def backward(ctx, g_out, g_lik):
        z = ctx.z
        gl = ops.from_nchw(g_lik.contiguous())
        dz = ops.new_view(z.B, z.H, z.W, z.C, z.buf.device)
        dpp = torch.zeros_like(ctx.pp)
        ops.eb_train_bwd(z, ctx.nz, ctx.pp, gl, dz, dpp)
        dx = dz.torch_nchw()
        if g_out is not None:
            dx = dx + g_out
        grads, off = [], 0
        for shp in ctx.shapes:
            n = int(np.prod(shp))
            grads.append(dpp[off:off + n].view(shp))
            off += n
        return (dx, None) + tuple(grads)


class _GaussTrainFn(torch.autograd.Function):
    """Training-mode Gaussian conditional as a module-level call (entropy_models.py:637-652 with ``training=True``):
    outputs = inputs + U(-1/2, 1/2) (quantize "noise" ignores the means, :132-138), likelihood of outputs - means under
    max(scales, 0.11) with LowerBound(1e-9); gradients w.r.t. inputs, scales and means incl. both LowerBound rules
    (vam_gauss_train forward / backward)."""

    @staticmethod
    def forward(ctx, inputs, scales, means, noise):
        L.require_gpu()
        y, sg, nz = ops.from_nchw(inputs.detach()), ops.from_nchw(scales.detach()), ops.from_nchw(noise)
        mu = ops.from_nchw(means.detach()) if means is not None else ops.new_view(y.B, y.H, y.W, y.C, inputs.device, zero=True)
        lik = ops.new_view(y.B, y.H, y.W, y.C, inputs.device)
        ops.gauss_train(y, mu, sg, nz, lik=lik)
        ctx.views = (y, mu, sg, nz)
        ctx.has_means = means is not None
        return inputs.detach() + noise, lik.torch_nchw().contiguous()

    @staticmethod
    def backward(ctx, g_out, g_lik):
        y, mu, sg, nz = ctx.views
        gl = ops.from_nchw(g_lik.contiguous())
        dmu = ops.new_view(y.B, y.H, y.W, y.C, y.buf.device)
        dsg = ops.new_view(y.B, y.H, y.W, y.C, y.buf.device)
        ops.gauss_train(y, mu, sg, nz, grad_lik=gl, dmu=dmu, dsigma=dsg)
        d_in = -dmu.torch_nchw()                      # the likelihood sees inputs + noise - means
        if g_out is not None:
            d_in = d_in + g_out
        return d_in, dsg.torch_nchw(), (dmu.torch_nchw() if ctx.has_means else None), None


class EntropyModel(nn.Module):
    """entropy_models.py:71-294 (buffers and the quantize/dequantize helpers)."""

    def __init__(self, likelihood_bound: float = 1e-9, entropy_coder: Optional[str] = None,
                 entropy_coder_precision: int = 16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        assert abs(likelihood_bound - 1e-9) < 1e-15, "kernels are built for the reference's 1e-9 likelihood bound"
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = _LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    @property
    def offset(self): return self._offset
    @property
    def quantized_cdf(self): return self._quantized_cdf
    @property
    def cdf_length(self): return self._cdf_length

    def quantize(self, inputs, mode, means=None, mask=None):
        """entropy_models.py:127-153.  "noise" uses torch's RNG exactly like the reference
        (uniform_(-1/2, 1/2)); the deterministic modes are one fused kernel."""
        if mode not in ("noise", "dequantize", "symbols"):
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        if mode == "noise":
            noise = torch.empty_like(inputs).uniform_(-0.5, 0.5)
            if mask is not None:
                noise = noise * mask
            return inputs + noise
        _no_autograd(inputs, means)
        y = ops.from_nchw(inputs)
        mu = ops.from_nchw(means.expand_as(inputs)) if means is not None else ops.new_view(y.B, y.H, y.W, y.C, inputs.device, zero=True)
        one = _ones_like(y)
        if mode == "dequantize":
            out = ops.new_view(y.B, y.H, y.W, y.C, inputs.device)
            ops.gauss_tail(y, mu, one, yhat=out)
            return out.torch_nchw()
        sym = torch.empty((y.B, y.H, y.W, y.C), dtype=torch.int32, device=inputs.device)
        ops.gauss_tail(y, mu, one, sym=sym)
        return sym.permute(0, 3, 1, 2)

    @staticmethod
    def dequantize(inputs, means=None):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs += means
        else:
            outputs = inputs.float()
        return outputs

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """entropy_models.py:175-183."""
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = bs.pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, : _cdf.size(0)] = _cdf
        return cdf

    def _check_tables(self):
        if self._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if len(self._quantized_cdf.size()) != 2:
            raise ValueError(f"Invalid CDF size {self._quantized_cdf.size()}")
        if self._offset.numel() == 0 or len(self._offset.size()) != 1:
            raise ValueError("Uninitialized offsets. Run update() first")
        if self._cdf_length.numel() == 0 or len(self._cdf_length.size()) != 1:
            raise ValueError("Uninitialized CDF lengths. Run update() first")

    def compress(self, inputs, indexes, means=None, flag=1, already_quantize=False):
        """entropy_models.py:206-242: one rANS stream per batch element."""
        if len(inputs.size()) < 2:
            raise ValueError("Invalid `inputs` size. Expected a tensor with at least 2 dimensions.")
        symbols = inputs if already_quantize else self.quantize(inputs, "symbols", means)
        if symbols.size() != indexes.size():
            raise ValueError("`inputs` and `indexes` should have the same size.")
        self._check_tables()
        t = bs.Tables.of(self)
        sym = symbols.detach().to(torch.int32).cpu().contiguous().numpy()
        idx = indexes.detach().to(torch.int32).cpu().contiguous().numpy()
        return [bs.encode(sym[i], idx[i], t) for i in range(sym.shape[0])]

    def decompress(self, strings, indexes, means=None, flag=1):
        """entropy_models.py:244-294."""
        if not isinstance(strings, (tuple, list)):
            raise ValueError("Invalid `strings` parameter type.")
        if not len(strings) == indexes.size(0):
            raise ValueError("Invalid strings or indexes parameters")
        if len(indexes.size()) < 2:
            raise ValueError("Invalid `indexes` size. Expected a tensor with at least 2 dimensions.")
        self._check_tables()
        if means is not None:
            if means.size()[:2] != indexes.size()[:2]:
                raise ValueError("Invalid means or indexes parameters")
            if means.size() != indexes.size():
                for i in range(2, len(indexes.size())):
                    if means.size(i) != 1:
                        raise ValueError("Invalid means parameters")
        t = bs.Tables.of(self)
        idx = indexes.detach().to(torch.int32).cpu().contiguous().numpy()
        out = np.stack([bs.decode(s, idx[i], t).reshape(idx[i].shape) for i, s in enumerate(strings)])
        outputs = torch.from_numpy(out).to(indexes.device)
        return self.dequantize(outputs, means)


def _ones_like(v: ops.View) -> ops.View:
    o = ops.new_view(v.B, v.H, v.W, v.C, v.buf.device)
    o.buf.fill_(1.0)
    return o


class EntropyBottleneck(EntropyModel):
    """entropy_models.py:297-525 (factorised prior; filters (3,3,3,3), init_scale 10)."""

    def __init__(self, channels: int, *args, tail_mass: float = 1e-9, init_scale: float = 10,
                 filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        assert self.filters == (3, 3, 3, 3), "vam_eb_forward is built for the reference's (3,3,3,3) filters"
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.Tensor([-self.init_scale, 0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def packed_params(self) -> torch.Tensor:
        """Parameter block in the order vam_eb_forward documents (include/vampic.h)."""
        names = ["_matrix0", "_bias0", "_factor0", "_matrix1", "_bias1", "_factor1", "_matrix2", "_bias2", "_factor2",
                 "_matrix3", "_bias3", "_factor3", "_matrix4", "_bias4", "quantiles"]
        ps = [getattr(self, n) for n in names]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_pp_key", None) != key:
            object.__setattr__(self, "_pp", torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous())
            object.__setattr__(self, "_pp_key", key)
        return self._pp

    def forward(self, x, training=None):
        """Eval forward (entropy_models.py:449-492): returns (round(x-med)+med, likelihood)."""
        if training is None:
            training = self.training
        if training:                        # additive-noise proxy, differentiable (reference :449-492; used by the harness)
            noise = torch.empty_like(x).uniform_(-0.5, 0.5)       # entropy_models.py:132-137
            return _EbTrainFn.apply(x, noise, *[getattr(self, n) for n in EB_PARAM_NAMES])
        _no_autograd(x)
        z = ops.from_nchw(x)
        zhat = ops.new_view(z.B, z.H, z.W, z.C, x.device)
        lik = ops.new_view(z.B, z.H, z.W, z.C, x.device)
        ops.eb_forward(z, self.packed_params(), zhat, lik)
        return zhat.torch_nchw(), lik.torch_nchw()

    def loss(self):
        """entropy_models.py:398-401: |logits_cumulative(quantiles) - target| summed, a function of the quantiles only
        (the density network is held constant); differentiable w.r.t. ``self.quantiles`` (models/base.py:22-29
        ``aux_loss`` + the aux optimiser of utility/functions.py:27-59)."""
        return _EbAuxLoss.apply(self.quantiles, self)

    def _logits_cumulative_host(self, v):
        """entropy_models.py:403-422 on the host (table building only; the per-pixel likelihood is
        vam_eb_forward)."""
        logits = v
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(getattr(self, f"_matrix{i:d}").detach().cpu()), logits)
            logits = logits + getattr(self, f"_bias{i:d}").detach().cpu()
            if i < len(self.filters):
                logits = logits + torch.tanh(getattr(self, f"_factor{i:d}").detach().cpu()) * torch.tanh(logits)
        return logits

    def update(self, force: bool = False) -> bool:
        """entropy_models.py:358-396: per-channel CDF tables from the learned quantiles."""
        q = self.quantiles.detach().cpu()
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        dev = self._offset.device
        self._offset = (-minima).to(dev)
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative_host(samples - 0.5)
        upper = self._logits_cumulative_host(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        object.__setattr__(self, "_tables_generation", getattr(self, "_tables_generation", 0) + 1)   # bitstream.Tables.of
        return True

    @staticmethod
    def _build_indexes(size):
        N, C_ = size[0], size[1]
        view = [1] * len(size)
        view[1] = -1
        return torch.arange(C_, dtype=torch.int32).view(*view).repeat(N, 1, *size[2:])

    def compress(self, x):
        """entropy_models.py:511-518."""
        indexes = self._build_indexes(x.size()).to(x.device)
        medians = self._get_medians().detach().reshape(1, -1, *([1] * (x.dim() - 2))).expand_as(x)
        return super().compress(x, indexes, medians)

    def decompress(self, strings, size):
        """entropy_models.py:520-525."""
        out_size = (len(strings), self._quantized_cdf.size(0), *size)
        dev = self.quantiles.device
        indexes = self._build_indexes(out_size).to(dev)
        medians = self._get_medians().detach().reshape(1, -1, *([1] * len(size))).expand(out_size)
        return super().decompress(strings, indexes, medians)


class GaussianConditional(EntropyModel):
    """entropy_models.py:528-673."""

    def __init__(self, scale_table, *args, scale_bound: float = 0.11, tail_mass: float = 1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.tail_mass = float(tail_mass)
        if scale_bound is None or scale_bound <= 0:
            raise ValueError("Invalid parameters")
        assert abs(scale_bound - 0.11) < 1e-12, "kernels are built for the reference's scale bound 0.11"
        self.lower_bound_scale = _LowerBound(scale_bound)
        self.register_buffer("scale_table", self._prepare_scale_table(scale_table) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    def update_scale_table(self, scale_table):
        """entropy_models.py:582-589."""
        self.update(scale_table)
        return True

    @staticmethod
    def _standardized_cumulative(inputs):
        return 0.5 * torch.erfc(float(-(2 ** -0.5)) * inputs)

    def update(self, scale_table):
        """entropy_models.py:591-618: one CDF table per entry of the scale table."""
        dev = self.scale_table.device
        table = self._prepare_scale_table([float(s) for s in scale_table])
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        scale = table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / scale)
        lower = self._standardized_cumulative((-0.5 - samples) / scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self.scale_table = table.to(dev)
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        object.__setattr__(self, "_tables_generation", getattr(self, "_tables_generation", 0) + 1)   # bitstream.Tables.of
        return True

    def forward(self, inputs, scales, means=None, training=None, mask=None):
        """Eval forward (entropy_models.py:637-652): (round(x-mu)+mu, likelihood)."""
        if training is None:
            training = self.training
        if training:                        # additive-noise proxy, differentiable (reference :637-652)
            noise = torch.empty_like(inputs).uniform_(-0.5, 0.5)  # entropy_models.py:132-137
            return _GaussTrainFn.apply(inputs, scales, means, noise)
        _no_autograd(inputs, scales, means)
        y = ops.from_nchw(inputs)
        sg = ops.from_nchw(scales)
        dev = inputs.device
        mu = ops.from_nchw(means) if means is not None else ops.new_view(y.B, y.H, y.W, y.C, dev, zero=True)
        out = ops.new_view(y.B, y.H, y.W, y.C, dev)
        lik = ops.new_view(y.B, y.H, y.W, y.C, dev)
        ops.gauss_tail(y, mu, sg, yhat=out, lik=lik)
        return out.torch_nchw(), lik.torch_nchw()

    def build_indexes(self, scales):
        """entropy_models.py:654-659."""
        if self.scale_table.numel() == 0:
            raise ValueError("empty scale table: call update_scale_table() / model.update() first")
        idx = ops.build_indexes(ops.from_nchw(scales), self.scale_table.to(scales.device).contiguous())
        return idx.permute(0, 3, 1, 2)


def get_scale_table(min=0.11, max=256, levels=64):
    """models/pic.py:17-18."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))
