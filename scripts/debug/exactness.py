#!/usr/bin/env python3
"""Per-sample parity of the HIP path against the CPU oracle, measured (not asserted): fraction of samples whose radiance is
bit-identical, and the largest differences of the rest.  Cornell box, RGB and spectral, all three schedules."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from mitsuba2_amd import render as gpu, scenes
import oracle_binding as oracle


def main():
    sd = scenes.cornell_box()
    for max_depth in (-1, 3, 6):
        p = scenes.cornell_box_sensor(64, 48, 16, seed=3, max_depth=max_depth)
        first, count = 1000, 40000
        for variant in ("rgb", "spectral"):
            scene, sensor = gpu.Scene(sd, variant=variant), gpu.make_sensor(p)
            S = oracle.OracleScene(sd, naive=True, spectral_path=gpu.srgb_coeff_path() if variant == "spectral" else None)
            ref, ref_pos = S.sample_radiance(oracle.make_desc(p), first, count)
            for pipeline in (0, 1, 2):
                integ = gpu.PathIntegrator(max_depth=max_depth, rr_depth=5, pipeline=pipeline)
                rgb, mask, pos = integ.sample(scene, sensor, first, count)
                rgb = rgb.cpu().numpy()
                exact = (rgb == ref[:, :3]).all(axis=1)
                bad = np.nonzero(~exact)[0]
                rel = np.abs(rgb - ref[:, :3]).max(axis=1) / np.maximum(np.abs(ref[:, :3]).max(axis=1), 1e-6)
                print("depth %2d %-8s pipeline %d: exact %.6f (%d of %d differ), worst rel diff %.3g, differing samples %s"
                      % (max_depth, variant, pipeline, exact.mean(), len(bad), count, rel.max(), bad[:8].tolist()))


if __name__ == "__main__":
    main()
