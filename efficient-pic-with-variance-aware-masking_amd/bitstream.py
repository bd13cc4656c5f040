"""Host-side bitstream helpers over the C ABI (`vam_pmf_to_quantized_cdf`, `vam_rans_encode`,
`vam_rans_decode`): the role compressai's C++ extension plays for the reference
(entropy_models.py:61-64,175-183,231-239,280-290).  Streams are per (image, slice) byte strings."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L


def pmf_to_quantized_cdf(pmf: torch.Tensor, precision: int = 16) -> torch.Tensor:
    p = np.ascontiguousarray(pmf.detach().cpu().numpy(), dtype=np.float32)
    out = np.zeros(p.size + 1, dtype=np.int32)
    L.check(L.load().vam_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, out.ctypes.data), "vam_pmf_to_quantized_cdf")
    return torch.from_numpy(out)


@dataclass
class Tables:
    """CDF tables of one entropy model, host side, in the layout the coder wants."""
    cdf: np.ndarray       # [n, stride] int32
    sizes: np.ndarray     # [n] int32
    offsets: np.ndarray   # [n] int32

    @staticmethod
    def of(model) -> "Tables":
        if model._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")
        # _tables_generation is bumped by update() / load_state_dict(): a re-built table can land at the address (and
        # version) of the one it replaced, so pointers alone do not identify it
        key = (getattr(model, "_tables_generation", 0), model._quantized_cdf.data_ptr(), model._quantized_cdf._version,
               model._offset._version, tuple(model._quantized_cdf.shape))
        if getattr(model, "_tables_key", None) != key:
            t = Tables(np.ascontiguousarray(model._quantized_cdf.detach().cpu().numpy(), dtype=np.int32),
                       np.ascontiguousarray(model._cdf_length.detach().cpu().reshape(-1).numpy(), dtype=np.int32),
                       np.ascontiguousarray(model._offset.detach().cpu().reshape(-1).numpy(), dtype=np.int32))
            object.__setattr__(model, "_tables", t)
            object.__setattr__(model, "_tables_key", key)
        return model._tables


def encode(symbols: np.ndarray, indexes: np.ndarray, t: Tables) -> bytes:
    s = np.ascontiguousarray(symbols, dtype=np.int32).reshape(-1)
    i = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
    assert s.size == i.size
    buf = np.empty(8 * s.size + 64, dtype=np.uint8)
    n = L.load().vam_rans_encode(s.ctypes.data, i.ctypes.data, s.size, t.cdf.ctypes.data, t.cdf.shape[1],
                                 t.sizes.ctypes.data, t.offsets.ctypes.data, t.cdf.shape[0], buf.ctypes.data, buf.size)
    if n < 0:
        L.check(int(n), "vam_rans_encode")
    return buf[:n].tobytes()


def decode(stream: bytes, indexes: np.ndarray, t: Tables) -> np.ndarray:
    i = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
    out = np.empty(i.size, dtype=np.int32)
    src = np.frombuffer(stream, dtype=np.uint8)
    L.check(L.load().vam_rans_decode(src.ctypes.data, src.size, i.ctypes.data, i.size, t.cdf.ctypes.data, t.cdf.shape[1],
                                     t.sizes.ctypes.data, t.offsets.ctypes.data, t.cdf.shape[0], out.ctypes.data),
            "vam_rans_decode")
    return out
