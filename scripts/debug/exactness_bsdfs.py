#!/usr/bin/env python3
"""Per-sample parity of the f-2 materials (tests/test_gpu_bsdfs.py MATERIALS) against the CPU oracle, measured: fraction of samples
with bit-identical radiance, and for the rest the size of the difference -- 'ulp' (relative difference below 1e-5: a last-place
difference somewhere that did not change any decision) or 'path' (the two sides took different branches)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mitsuba2_amd import render as R, scenes
import oracle_binding as ob
from test_gpu_bsdfs import MATERIALS, _scene


def main():
    sp = scenes.cornell_box_sensor(64, 64, spp=8, seed=21)
    sp["max_depth"] = 6
    n = 64 * 64 * 8
    for material in sorted(MATERIALS):
        cb = _scene(material)
        scene, sensor = R.Scene(cb), R.make_sensor(sp)
        rgb, mask, pos = R.PathIntegrator(max_depth=6, pipeline=4).sample(scene, sensor, 0, n)
        rgb = rgb.cpu().numpy()
        want, _ = ob.OracleScene(cb).sample_radiance(ob.make_desc(sp), 0, n)
        want = want[:, :3]
        exact = (rgb == want).all(1)
        rel = np.abs(rgb - want).max(1) / np.maximum(np.abs(want).max(1), 1e-6)
        bad = ~exact
        print("%-26s exact %.6f  differ %5d: ulp-level %5d, path-level %4d, worst rel %.3g  first %s" %
              (material, exact.mean(), bad.sum(), (bad & (rel < 1e-5)).sum(), (bad & (rel >= 1e-5)).sum(), rel.max(), np.nonzero(bad)[0][:5].tolist()))


if __name__ == "__main__":
    main()
