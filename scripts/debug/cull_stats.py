"""Diagnostic (libraries built with -DMTS_CULL_STATS=1): triangles really tested per closest-hit ray on the Cornell box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mitsuba2_amd import render as R, scenes
scene = R.Scene(scenes.cornell_box())
sensor = R.make_sensor(scenes.cornell_box_sensor(1024, 1024, 64))
integ = R.PathIntegrator()
integ.render(scene, sensor); torch.cuda.synchronize()
st = integ.stats
# any-hit loops still count nominally (36 per ray); closest-hit loops: nominal without culling, real with it
nominal = 36.0 * (st["closest_hit_rays"] + st["any_hit_rays"])
print("tri tests %.3e nominal %.3e  closest rays %.3e  -> %.2f tests per closest-hit ray (36 = no culling)" %
      (st["tri_tests"], nominal, st["closest_hit_rays"], (st["tri_tests"] - 36.0 * st["any_hit_rays"]) / st["closest_hit_rays"]))
