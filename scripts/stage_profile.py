import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from mitsuba2_amd import render, scenes
variant = sys.argv[1] if len(sys.argv) > 1 else "rgb"
sd = scenes.bumpy_sphere(256, 512)
scene = render.Scene(sd, variant=variant)
sensor = render.make_sensor(scenes.bumpy_sphere_sensor(1920, 1080, 64))
for prof in (False, True):
    integ = render.PathIntegrator(profile=prof)
    integ.render(scene, sensor); torch.cuda.synchronize()
    t0 = time.perf_counter(); integ.render(scene, sensor); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = integ.stats
    print(variant, "profile", prof, "%.1f ms" % (dt * 1e3), {k: v for k, v in st.items() if "ns" in k or "launch" in k or k in ("iterations", "segments", "samples")})
