import sys, time, torch
sys.path.insert(0, '/root/repo')
from mitsuba2_amd import render, scenes
scene = render.Scene(scenes.cornell_box())
integ = render.PathIntegrator()
N = 8
for tile in (32, 16, 8):
    ts = []
    sensor = render.make_sensor(scenes.cornell_box_sensor(1024, 1024, 256 * N))
    integ.render(scene, sensor, partition=(0, N, tile))
    for r in range(N):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        integ.render(scene, sensor, partition=(r, N, tile))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("tile_rows", tile, "per-rank ms", ["%.1f" % t for t in ts], "max %.1f mean %.1f -> efficiency %.3f, film ms %.2f" % (max(ts), sum(ts) / N, sum(ts) / N / max(ts), integ.stats["film_ns"] * 1e-6))
