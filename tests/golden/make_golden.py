#!/usr/bin/env python3
"""Generates tests/golden/oracle_v2.npz with the CPU oracle (the reference itself cannot be built or imported in
the build container; its own known-answer values are asserted in tests/test_oracle_kat.py).
The file freezes the oracle's outputs on small seeded inputs so that (a) the oracle cannot drift silently and
(b) the GPU box can check the HIP path against committed data.  Inputs are regenerated from fixed seeds by
tests/test_golden.py; only outputs (and the random rays) are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import oracle_binding as ob          # noqa: E402
from mitsuba2_amd import scenes      # noqa: E402


def golden_inputs():
    rng = np.random.RandomState(20261004)
    sd = scenes.cornell_box()
    allp = np.concatenate([m["positions"] for m in sd["meshes"]])
    lo, hi = allp.min(0), allp.max(0)
    n = 1024
    o = (lo + (hi - lo) * rng.rand(n, 3)).astype(np.float32)
    d = rng.randn(n, 3)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    mint = np.full(n, 1e-4, np.float32)
    maxt = np.where(rng.rand(n) < 0.25, 300.0, np.inf).astype(np.float32)
    sensor = scenes.cornell_box_sensor(16, 16, 4, seed=7)
    return sd, (o, d, mint, maxt), sensor


def main():
    sd, (o, d, mint, maxt), sensor = golden_inputs()
    S = ob.OracleScene(sd, naive=True)
    out = {}
    out["ray_o"], out["ray_d"], out["ray_mint"], out["ray_maxt"] = o, d, mint, maxt
    t, prim, shape, u, v = S.ray_intersect(o, d, mint, maxt, naive=True)
    out.update(hit_t=t, hit_prim=prim, hit_shape=shape, hit_u=u, hit_v=v, any_hit=S.ray_test(o, d, mint, maxt, naive=True))
    out["si"] = S.fill_si(d[:64], prim[:64], u[:64], v[:64])
    desc = ob.make_desc(sensor)
    rgba, pos = S.sample_radiance(desc, 0, 256)
    out.update(sample_rgba=rgba, sample_pos=pos)
    out["film_wavefront"], _ = S.render(desc, mode=1)
    out["film_block"], _ = S.render(desc, mode=0, n_threads=1, block_size=32)
    box = dict(sensor); box["rfilter"], box["rfilter_param"] = "box", 0.5
    out["film_wavefront_box"], _ = S.render(ob.make_desc(box), mode=1)
    u32 = np.zeros(16, np.uint32); f32 = np.zeros(16, np.float32)
    ob.lib().mo_kat_pcg32(7, 11, 16, u32.ctypes.data, f32.ctypes.data)
    out.update(pcg_u32=u32, pcg_f32=f32)
    out["tea64"] = np.array([ob.lib().mo_kat_tea64_u64(a, b, 4) for a, b in ((0, 0), (1, 1), (12345678901234, 5), (5, 12345678901234))], np.uint64)
    np.savez_compressed(os.path.join(HERE, "oracle_v2.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_v2.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
