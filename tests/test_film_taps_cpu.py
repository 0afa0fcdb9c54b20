"""The short form of the filter weights in k_film_accum (kernels.hip, axis_taps_r2) against the general form (axis_taps, a restatement of
imageblock.cpp:117-147): the same bits for every tap of an interior pixel, for every table filter of radius 2.  Both forms are restated
here in float32 numpy, operation by operation; the GPU film tests check the kernels themselves against the oracle."""
import numpy as np
import pytest

f32 = np.float32


def make_table(kind):
    """32-entry discretisation of a radius-2 filter, last entry zero (rfilter.h: init_discretization)"""
    r = f32(2.0)
    x = (r * np.arange(31, dtype=np.float32)) / f32(31.0)
    if kind == "gaussian":
        alpha = f32(-1.0) / (f32(2.0) * f32(0.5) * f32(0.5))
        v = np.maximum(f32(0.0), np.exp(alpha * x * x).astype(np.float32) - f32(np.exp(alpha * r * r)))
    else:                                   # a kernel with negative lobes (mitchell-like shape; the values only have to be a table)
        v = (np.cos(x * f32(2.2)) * (f32(1.0) - x / r)).astype(np.float32)
    return np.concatenate([v.astype(np.float32), np.zeros(1, np.float32)])


def general(table, pos, size, t0, taps=4):
    """axis_taps: weights of the film pixels t0 .. t0 + 4 for a sample at block coordinate pos"""
    r, sf = f32(2.0), f32(31.0) / f32(2.0)
    lo = np.maximum(np.ceil(pos - r), f32(0.0)).astype(np.int64)
    hi = np.minimum(np.floor(pos + r).astype(np.int64), size - 1)
    base = lo.astype(np.float32) - pos
    out = np.zeros((pos.size, 5), np.float32)
    for k in range(5):
        t = t0 + k
        i = t - lo
        valid = (i >= 0) & (i < taps) & (t <= hi)
        xx = base + i.astype(np.float32)
        idx = np.minimum(np.abs(xx * sf).astype(np.int64), 31)
        out[:, k] = np.where(valid, table[idx], f32(0.0))
    return out


def short(table, pos, t0):
    """axis_taps_r2: no window tests, zeros from the extended table"""
    sf = f32(31.0) / f32(2.0)
    table64 = np.concatenate([table, np.zeros(32, np.float32)])
    lo = np.ceil(pos - f32(2.0)).astype(np.float32)
    base = lo - pos
    dk = lo - t0.astype(np.float32)
    out = np.zeros((pos.size, 5), np.float32)
    for k in range(5):
        xx = base + (f32(k) - dk)
        out[:, k] = table64[np.abs(xx * sf).astype(np.int64)]
    return out


@pytest.mark.parametrize("kind", ["gaussian", "lobes"])
def test_short_form_equals_general_form(kind):
    rng = np.random.default_rng(5)
    table = make_table(kind)
    size = 4096 + 4
    tq = rng.integers(2, size - 2, 400000)                       # interior pixels: taps t_q - 2 .. t_q + 2 on the film
    off = rng.random(400000, dtype=np.float32) - f32(0.5)          # a sample of pixel q sits at q + border + [-0.5, 0.5)
    # the cases rounding could split: exact integers, exact halves, one ulp either side of them
    special = np.array([-0.5, 0.0, 0.4999999, -0.4999999, 1e-7, -1e-7, 0.25, -0.25], np.float32)
    off[: special.size * 1000] = np.tile(special, 1000)
    pos = (tq.astype(np.float32) + off).astype(np.float32)
    for nudge in (0.0, 1.0, -1.0):                                 # and the neighbouring floats of every position
        p = np.nextafter(pos, pos + f32(nudge)) if nudge else pos
        keep = (p >= tq - 0.5) & (p < tq + 0.5 + 1e-3)
        a, b = general(table, p[keep], size, tq[keep] - 2), short(table, p[keep], tq[keep] - 2)
        assert np.array_equal(a, b), (kind, nudge, int((a != b).sum()))
