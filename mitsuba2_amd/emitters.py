"""Host-side emitter plugin parameters: plugin dictionaries of `area`, `constant`, `envmap` and the
delta emitters `point`, `spot`, `directional` -> the flat record of ``mtsamd_emitter_desc`` with the reference constructors'
defaults and error behaviour: ``src/emitters/point.cpp:52-65``, ``spot.cpp:68-91``, ``directional.cpp:43-63``.
"""
import numpy as np

TYPE_IDS = {"area": 0, "constant": 1, "envmap": 2, "point": 3, "spot": 4, "directional": 5}
_VALUE_KEY = {"area": "radiance", "constant": "radiance", "point": "intensity", "spot": "intensity", "directional": "irradiance"}


def _rgb(v):
    if isinstance(v, (int, float)):
        v = [float(v)] * 3
    a = np.asarray(v, dtype=np.float32).reshape(-1)
    if a.size == 1:
        a = np.repeat(a, 3)
    if a.size != 3:
        raise RuntimeError("expected a colour (3 values) or a constant")
    return [float(x) for x in a]


def normalize(e):
    """plugin dictionary -> dict(type (id), radiance, to_world (4x4 float32), cutoff_angle, beam_width [, data, scale])"""
    t = e.get("type", "area")
    if t not in TYPE_IDS:
        raise RuntimeError("Emitter plugin '%s' is not supported by this backend (%s)" % (t, ", ".join(TYPE_IDS)))
    out = dict(type=TYPE_IDS[t], radiance=[0.0, 0.0, 0.0], to_world=np.eye(4, dtype=np.float32), cutoff_angle=0.0, beam_width=0.0)
    if e.get("to_world") is not None:
        out["to_world"] = np.asarray(e["to_world"], dtype=np.float32).reshape(4, 4).copy()
    if t == "envmap":
        out["data"], out["scale"] = e["data"], float(e.get("scale", 1.0))
        return out
    key = _VALUE_KEY[t]
    value = e.get(key, e.get("radiance"))              # the synthetic scenes call every emitted quantity `radiance`
    out["radiance"] = _rgb(1.0 if value is None else value)     # Texture::D65(1.f) default
    if t == "point" and e.get("position") is not None:
        if e.get("to_world") is not None:                 # point.cpp:53-57
            raise RuntimeError("Only one of the parameters 'position' and 'to_world' can be specified at the same time!'")
        out["to_world"][:3, 3] = np.asarray(e["position"], dtype=np.float32)
    if t == "directional" and e.get("direction") is not None:
        if e.get("to_world") is not None:                 # directional.cpp:48-51
            raise RuntimeError("Only one of the parameters 'direction' and 'to_world' can be specified at the same time!'")
        d = np.asarray(e["direction"], dtype=np.float32)
        d = d / np.float32(np.sqrt(np.float32((d * d).sum())))
        # look_at(0, direction, up): only the image of the local +z axis (= direction) is used by the emitter
        a = np.float32([1, 0, 0]) if abs(d[0]) < 0.9 else np.float32([0, 1, 0])
        s = np.cross(a, d); s = s / np.linalg.norm(s)
        out["to_world"][:3, 0], out["to_world"][:3, 1], out["to_world"][:3, 2] = s, np.cross(d, s), d
    if t == "spot":                                       # spot.cpp:81-82
        if isinstance(e.get("texture"), dict):
            raise RuntimeError("spot: projection textures are not supported by this backend")
        out["cutoff_angle"] = float(e.get("cutoff_angle", 20.0))
        out["beam_width"] = float(e.get("beam_width", out["cutoff_angle"] * 3.0 / 4.0))
        if out["cutoff_angle"] < out["beam_width"]:
            raise RuntimeError("spot: cutoff_angle must not be smaller than beam_width")
    return out
