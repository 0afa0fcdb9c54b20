#!/usr/bin/env python3
"""BASELINE config 4: one inverse-rendering iteration on the Cornell box (primal render + derivative render + adjoint +
Adam step), the setup of docs/examples/10_inverse_rendering/invert_cbox.py: path max_depth=3, box filter, spp=1,
unbiased=True, image writing disabled.  The reference quotes ~50 ms (unbiased) / ~27 ms (biased) per iteration on a
Titan RTX (docs/src/inverse_rendering/diff_render.rst:311-314), film resolution not stated; 256x256 is used here."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba2_amd import render, scenes, autodiff


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--texture", type=int, default=0, help="optimise a texture of this resolution on floor + back wall instead of the red wall colour")
    args = ap.parse_args()
    tex = None
    if args.texture:
        tex = np.full((args.texture, args.texture, 3), 0.5, np.float32)
    sd = scenes.cornell_box(texture=tex)
    for b, n in zip(sd["bsdfs"], ["white", "red", "green", "light", "textured"]):
        b["id"] = n
    p = scenes.cornell_box_sensor(args.res, args.res, args.spp, max_depth=3, rfilter="box")
    scene = render.Scene(sd, sensor=render.make_sensor(p), integrator=render.PathIntegrator(max_depth=3))
    params = autodiff.traverse(scene)
    key = "textured.reflectance.data" if args.texture else "red.reflectance.value"
    params.keep([key])
    ref = params[key].clone()
    image_ref = autodiff.render(scene, spp=8).detach()
    params[key] = torch.full_like(ref, 0.9)
    params.update()
    for unbiased in (True, False):
        opt = autodiff.Adam(params, lr=0.2 if not args.texture else 0.02)
        for it in range(5):          # warm-up
            img = autodiff.render(scene, optimizer=opt, unbiased=unbiased, spp=args.spp)
            (((img - image_ref) ** 2).sum() / img.numel()).backward()
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(args.iters):
            img = autodiff.render(scene, optimizer=opt, unbiased=unbiased, spp=args.spp)
            (((img - image_ref) ** 2).sum() / img.numel()).backward()
            opt.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.iters
        err = ((ref - params[key].detach()) ** 2).mean().item()
        print("cbox %dx%d spp=%d max_depth=3 box filter, %s, unbiased=%s: %.2f ms per iteration (fwd+adjoint+Adam), param mse %.3g"
              % (args.res, args.res, args.spp, key, unbiased, ms, err))


if __name__ == "__main__":
    main()
