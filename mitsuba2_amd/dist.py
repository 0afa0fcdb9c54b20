"""Film partition across the GPUs of one node + the one collective of the path.

The reference has no multi-device support at all; samples are independent and
``ImageBlock`` accumulation is a plain sum of weighted values and weights (``src/librender/imageblock.cpp:49-77``,
normalisation deferred to ``HDRFilm::bitmap``), so any partition of the camera samples followed by one sum is
exact up to fp32 addition order.

* one process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` in the CPU tests);
* the film is cut into tiles of ``tile_rows`` rows dealt round-robin to the ranks -- interleaving keeps the per-rank
  work balanced although path length varies over the image.  16 rows (half of ``MTS_BLOCK_SIZE``,
  ``include/mitsuba/render/spiral.h:10``, and the height of the film kernel's source tiles): on the 1024-row headline
  film cut over 8 ranks the slowest rank then takes 1.026 x the mean (32-row tiles: 1.055 x, 8-row tiles: 1.022 x;
  ``scripts/rank_balance.py``); RNG streams are seeded with the *global* sample index, so the image is independent of the
  number of ranks (up to the summation order of the final reduce);
* every rank splats its samples into a full-size XYZAW film (its tiles plus the filter apron), the scene is
  replicated, and the films are summed with ONE ``reduce`` (root 0) or ``all_reduce``.
"""
import torch
import torch.distributed as dist

TILE_ROWS = 16


def film_partition(rank, world_size, tile_rows=TILE_ROWS):
    """Partition descriptor for PathIntegrator.render(partition=...); None for a single process."""
    if world_size <= 1:
        return None
    return (int(rank), int(world_size), int(tile_rows))


def owned_rows(height, rank, world_size, tile_rows=TILE_ROWS):
    """Global film rows sampled by `rank` (for tests / bookkeeping)."""
    if world_size <= 1:
        return list(range(height))
    rows = []
    t = rank
    while t * tile_rows < height:
        rows.extend(range(t * tile_rows, min((t + 1) * tile_rows, height)))
        t += world_size
    return rows


def reduce_film(film, root=0, all_ranks=False):
    """Sum the per-rank XYZAW films.  ``film`` is modified in place on the receiving rank(s)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return film
    buf = film
    staged = film.is_cuda and dist.get_backend() == "gloo"      # rehearsal backend: stage through host memory
    if staged:
        buf = film.cpu()
    if all_ranks:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM)
    if staged:
        film.copy_(buf)
    return film


def render_distributed(integrator, scene, sensor, tile_rows=TILE_ROWS, all_ranks=False):
    """Integrator::render on this rank's share of the film followed by the film reduce.
    Returns the XYZAW film tensor (complete on rank 0, or on every rank with all_ranks=True)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    ok = integrator.render(scene, sensor, partition=film_partition(rank, world, tile_rows))
    if not ok:
        raise RuntimeError("render cancelled")
    return reduce_film(sensor.film().bitmap(raw=True), all_ranks=all_ranks)
