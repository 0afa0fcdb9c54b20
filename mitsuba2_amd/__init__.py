"""mitsuba2_amd -- MI355X (gfx950) wavefront path-tracing backend behind Mitsuba 2's operator API.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + host scheduler + C ABI, built into
``libmtsamd.so``), ``render.py`` (host-side mirror of the reference's Python surface), ``scenes.py``
(synthetic scene descriptions) and ``dist.py`` (film partition across GPUs, RCCL reduce).
"""
from . import scenes  # noqa: F401

__all__ = ["scenes", "render", "dist"]
