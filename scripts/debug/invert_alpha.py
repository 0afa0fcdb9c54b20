#!/usr/bin/env python3
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mitsuba2_amd import render as gpu, autodiff
from test_gpu_autodiff import _material_scene

mats = ({"type": "roughconductor", "alpha": 0.15, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]}, {"type": "diffuse", "reflectance": [0.5, 0.5, 0.5]})
sd, p, scene = _material_scene(gpu, mats, w=64, h=64, spp=16, max_depth=4, seed=21)
params = autodiff.traverse(scene)
key = "tall.alpha.value"
params.keep([key])
image_ref = autodiff.render(scene, spp=512).detach()
def loss_of(alpha, spp):
    params[key] = [alpha]; params.update()
    with torch.no_grad():
        img = autodiff.render(scene, params=params, spp=spp)
    return float((((img - image_ref) ** 2).sum() / img.numel()).item())
for a in (0.1, 0.15, 0.2, 0.3, 0.45, 0.6, 0.9):
    print("alpha %.2f  L2 loss @1024spp %.6e" % (a, loss_of(a, 1024)))
params[key] = [0.45]; params.update()
opt = autodiff.Adam(params, lr=0.02)
for it in range(40):
    image = autodiff.render(scene, optimizer=opt, unbiased=True, spp=16)
    loss = ((image - image_ref) ** 2).sum() / image.numel()
    loss.backward()
    g = float(params[key].grad.item())
    opt.step()
    print("it %2d loss %.5e grad %+.4e alpha -> %.4f" % (it, float(loss.item()), g, float(params[key].item())))
print("---- with the clamp of the test")
params[key] = [0.45]; params.update()
opt = autodiff.Adam(params, lr=0.02)
for it in range(30):
    image = autodiff.render(scene, optimizer=opt, unbiased=True, spp=16)
    loss = ((image - image_ref) ** 2).sum() / image.numel()
    loss.backward()
    g = float(params[key].grad.item())
    opt.step()
    a_step = float(params[key].item())
    with torch.no_grad():
        params[key] = params[key].detach().clamp(0.02, 1.0)
        params[key].requires_grad_(True)
    print("it %2d grad %+.4e alpha after step %.4f after clamp %.4f requires_grad %s" % (it, g, a_step, float(params[key].item()), params[key].requires_grad))
