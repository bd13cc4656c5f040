"""Bitstream layer on the host: the C++ coder behind the C ABI against the independent pure-Python
restatement (oracle/rans_oracle.py) of the same published algorithm, plus round-trip and
table-shape properties.  (compressai is absent offline: wire format unpinned, see DESIGN.md.)"""
import numpy as np
import pytest
import torch

import rans_oracle as R
import vampic
from vampic import bitstream as bs


def _tables(widths=(1, 3, 8, 20, 40)):
    cdfs, sizes, offs = [], [], []
    for w in widths:
        k = np.arange(-w, w + 1)
        pmf = np.exp(-0.5 * (k / (0.3 * w + 0.2)) ** 2).astype(np.float32)
        pmf /= pmf.sum()
        prob = torch.from_numpy(np.concatenate([pmf, [np.float32(1e-4)]]).astype(np.float32))
        c = bs.pmf_to_quantized_cdf(prob, 16).numpy()
        assert list(c) == R.pmf_to_quantized_cdf(prob.numpy(), 16)
        cdfs.append(c)
        sizes.append(len(c))
        offs.append(-w)
    tab = np.zeros((len(cdfs), max(sizes)), dtype=np.int32)
    for i, c in enumerate(cdfs):
        tab[i, :len(c)] = c
    return bs.Tables(tab, np.array(sizes, dtype=np.int32), np.array(offs, dtype=np.int32))


def test_cdf_properties():
    t = _tables()
    for i in range(t.cdf.shape[0]):
        c = t.cdf[i, :t.sizes[i]]
        assert c[0] == 0 and c[-1] == 65536 and (np.diff(c) >= 1).all()
    # a pmf with zeros still gets non-zero frequencies everywhere
    c = bs.pmf_to_quantized_cdf(torch.tensor([0.0, 0.0, 1.0, 0.0, 1e-9]), 16).numpy()
    assert (np.diff(c) >= 1).all() and c[-1] == 65536
    with pytest.raises(vampic._lib.VamError):
        bs.pmf_to_quantized_cdf(torch.tensor([0.5, float("nan")]), 16)


@pytest.mark.parametrize("n", [0, 1, 7, 4096])
def test_cpp_coder_equals_python_restatement_and_round_trips(n):
    t = _tables()
    rng = np.random.default_rng(n)
    idx = rng.integers(0, 5, n).astype(np.int32)
    width = np.array([1, 3, 8, 20, 40])[idx] if n else np.zeros(0)
    sym = np.round(rng.normal(0, 0.3 * width + 0.2)).astype(np.int32)
    if n > 100:
        sym[::97] = 500          # far out of range: bypass coding, several 4-bit chunks
        sym[::131] = -777
        sym[::53] = 41           # just outside the widest table
    stream = bs.encode(sym, idx, t)
    assert stream == R.encode(sym.tolist(), idx.tolist(), t.cdf.tolist(), t.sizes.tolist(), t.offsets.tolist())
    assert len(stream) % 4 == 0 and len(stream) >= 8
    assert np.array_equal(bs.decode(stream, idx, t), sym)
    assert R.decode(stream, idx.tolist(), t.cdf.tolist(), t.sizes.tolist(), t.offsets.tolist()) == sym.tolist()


def test_rate_is_close_to_entropy():
    t = _tables((8,))
    rng = np.random.default_rng(3)
    n = 20000
    pmf = np.diff(t.cdf[0, :t.sizes[0]]) / 65536.0
    sym = rng.choice(np.arange(-8, 8 + 1), size=n, p=pmf[:-1] / pmf[:-1].sum()).astype(np.int32)
    stream = bs.encode(sym, np.zeros(n, dtype=np.int32), t)
    ideal = -np.log2(pmf[sym + 8]).sum() / 8
    assert ideal <= len(stream) <= ideal * 1.001 + 16


def test_truncated_stream_is_an_error():
    t = _tables()
    idx = np.zeros(2000, dtype=np.int32) + 4
    sym = np.random.default_rng(0).integers(-30, 30, 2000).astype(np.int32)
    stream = bs.encode(sym, idx, t)
    with pytest.raises(vampic._lib.VamError):
        bs.decode(stream[:len(stream) // 2 // 4 * 4], idx, t)
    with pytest.raises(vampic._lib.VamError):
        bs.encode(sym, idx + 100, t)                 # table index out of range


def test_entropy_model_tables():
    gc = vampic.GaussianConditional(None)
    with pytest.raises(ValueError):
        bs.Tables.of(gc)                             # "Uninitialized CDFs. Run update() first"
    gc.update_scale_table(vampic.get_scale_table())
    assert tuple(gc._quantized_cdf.shape) == (64, 3133) and gc._cdf_length[0] == 5 and gc._offset[0] == -1
    lens = gc._cdf_length.numpy()
    assert (np.diff(lens) >= 0).all()                # wider tables for larger scales
    eb = vampic.EntropyBottleneck(16)
    eb.update()
    assert tuple(eb._quantized_cdf.shape) == (16, 23) and (eb._offset == -10).all()
