#!/usr/bin/env python3
"""BSDF-parameter gradients (mtsamd_render_adjoint_param) against central finite differences of the primal render with common random
numbers: prints both for every component (tests/test_gpu_autodiff.py::test_bsdf_parameter_gradients asserts the agreement)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mitsuba2_amd import render as gpu, autodiff
from test_gpu_autodiff import _material_scene, _loss_and_grad, _loss_only


def main():
    mats = ({"type": "roughconductor", "alpha": 0.3, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14], "specular_reflectance": [0.9, 0.8, 0.7]},
            {"type": "plastic", "diffuse_reflectance": [0.2, 0.5, 0.3], "int_ior": 1.6})
    for spp in (64, 1024):
        sd, p, scene = _material_scene(gpu, mats, spp=spp)
        params = autodiff.traverse(scene)
        n_pix = 48 * 48
        weights = torch.ones(n_pix * 3, device="cuda") / n_pix
        for key, step in (("tall.alpha.value", 0.03), ("small.diffuse_reflectance.value", 0.03), ("tall.k.value", 0.15), ("tall.eta.value", 0.05),
                          ("tall.specular_reflectance.value", 0.03), ("small.specular_reflectance.value", 0.03), ("bsdf_0.reflectance.value", 0.03)):
            base = params[key].detach().clone()
            _, g = _loss_and_grad(autodiff, scene, params, key, weights, spp)
            for c in range(base.numel()):
                vp, vm = base.clone(), base.clone()
                vp.view(-1)[c] += step; vm.view(-1)[c] -= step
                fd = (_loss_only(autodiff, scene, params, key, vp, weights, spp) - _loss_only(autodiff, scene, params, key, vm, weights, spp)) / (2 * step)
                print("spp %5d %-36s [%d] adjoint %+.6f  finite difference %+.6f  ratio %.3f" % (spp, key, c, float(g.view(-1)[c]), fd, float(g.view(-1)[c]) / fd if fd else 0.0))
            params[key] = base
            params.update()


if __name__ == "__main__":
    main()
