"""Tabulated spectra on the host: what the reference's XML loader does with `<spectrum value="400:0.3, 500:0.7, ..."/>` or
`<spectrum filename=.../>` in the RGB variants -- pre-integration against the CIE 1931 observer and conversion to linear sRGB
(``src/libcore/xml.cpp:1084-1146``, ``src/libcore/spectrum.cpp:9-86``, ``include/mitsuba/core/spectrum.h:127-237``)."""
import os
import re

import numpy as np

F32 = np.float32
MTS_WAVELENGTH_MIN, MTS_WAVELENGTH_MAX = F32(360.0), F32(830.0)
MTS_CIE_Y_NORMALIZATION = F32(1.0 / 106.7502593994140625)          # spectrum.h:133

_tables = None


def _cie():
    """the CIE 1931 tables shipped with the HIP sources (csrc/cie_data.h: generated data)"""
    global _tables
    if _tables is None:
        text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "cie_data.h")).read()
        out = {}
        for key in ("x", "y", "z", "d65"):
            body = re.search(r"kCie_%s\[95\]\s*=\s*\{(.*?)\};" % key, text, re.S).group(1)
            out[key] = np.array([float(v) for v in re.findall(r"[-+]?\d*\.\d+(?:[eE][-+]?\d+)?", body)], dtype=F32)
            assert out[key].size == 95
        _tables = out
    return _tables


def cie1931_xyz(wavelength):
    """spectrum.h:150-176: linear interpolation in the 95-sample table, 0 outside [360, 830] nm"""
    t = _cie()
    w = np.asarray(wavelength, dtype=F32)
    tt = (w - F32(360.0)) * F32(94.0 / 470.0)
    i0 = np.clip(tt.astype(np.int32), 0, 93)
    w1 = tt - i0.astype(F32)
    w0 = F32(1.0) - w1
    ok = (w >= F32(360.0)) & (w <= F32(830.0))
    return np.stack([np.where(ok, w0 * t[k][i0] + w1 * t[k][i0 + 1], F32(0)) for k in "xyz"], axis=-1).astype(F32)


def xyz_to_srgb(xyz):
    m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], dtype=F32)
    return (np.asarray(xyz, dtype=F32) @ m.T).astype(F32)


def spectrum_to_rgb(wavelengths, values, bounded=True):
    """spectrum.cpp:41-86: Riemann sum over 1000 steps of the piecewise-linear spectrum times the colour matching functions"""
    wl, val = np.asarray(wavelengths, dtype=F32), np.asarray(values, dtype=F32)
    if wl.size < 2 or wl.size != val.size:
        raise RuntimeError("spectrum: at least two wavelength:value pairs are required")
    if (np.diff(wl) < 0).any():
        raise RuntimeError("Wavelengths must be specified in increasing order!")
    steps = 1000
    x = MTS_WAVELENGTH_MIN + (np.arange(steps, dtype=F32) / F32(steps - 1)) * (MTS_WAVELENGTH_MAX - MTS_WAVELENGTH_MIN)
    keep = (x >= wl[0]) & (x <= wl[-1])
    x = x[keep]
    idx = np.clip(np.searchsorted(wl, x, side="right") - 1, 0, wl.size - 2)       # math::find_interval
    x0, x1, y0, y1 = wl[idx], wl[idx + 1], val[idx], val[idx + 1]
    y = (x * y0 - x1 * y0 - x * y1 + x0 * y1) / (x0 - x1)
    color = (cie1931_xyz(x) * y[:, None]).sum(axis=0, dtype=F32)
    color = color * ((MTS_WAVELENGTH_MAX - MTS_WAVELENGTH_MIN) / F32(steps))
    color = xyz_to_srgb(color)
    if bounded:
        color = np.clip(color, 0.0, 1.0)
    else:
        color = np.maximum(color, 0.0)
    return [float(c) for c in color]


def spectrum_from_file(path):
    """spectrum.cpp:9-38: `wavelength value` per line, '#' comments"""
    if not os.path.exists(path):
        raise RuntimeError('"%s": file does not exist!' % path)
    wl, val = [], []
    for line in open(path):
        line = line.strip()
        if not line or line[0] == "#":
            continue
        tok = line.split()
        if len(tok) != 2:
            raise RuntimeError('"%s": excess tokens after wavlengths-value pair in file:\n%s!' % (path, line))
        wl.append(float(tok[0])); val.append(float(tok[1]))
    return wl, val


def tabulated_to_rgb(wavelengths, values, within_emitter, name):
    """create_texture_from_spectrum, non-spectral branch (xml.cpp:1084-1140)"""
    unbounded = name in ("eta", "k", "int_ior", "ext_ior")                           # is_unbounded_spectrum (xml.cpp:83-85)
    scaled = [float(F32(v) * MTS_CIE_Y_NORMALIZATION) for v in values]
    return spectrum_to_rgb(wavelengths, scaled, bounded=not (within_emitter or unbounded))
