"""Host-side BSDF plugin parameters: turns the plugin dictionaries / XML properties of
`diffuse`, `conductor`, `roughconductor`, `dielectric`, `plastic` and the `twosided` adapter into the flat record the C ABI
takes (``mtsamd_bsdf_desc``; also `roughplastic`, ``roughplastic.cpp:146-178``, and `roughdielectric`, ``roughdielectric.cpp:141-196``), with the reference constructors' defaults and error behaviour:
``src/bsdfs/diffuse.cpp:72-76``, ``conductor.cpp:185-200``, ``roughconductor.cpp:143-192``, ``dielectric.cpp:175-198``,
``plastic.cpp:136-160``, ``twosided.cpp:63-92``, ``include/mitsuba/render/ior.h:20-101`` (named indices of refraction).
"""
import numpy as np

DIFFUSE, CONDUCTOR, ROUGHCONDUCTOR, DIELECTRIC, PLASTIC, ROUGHPLASTIC, ROUGHDIELECTRIC, THINDIELECTRIC, BLEND, MASK = range(10)
NESTING = {"blendbsdf": BLEND, "mask": MASK}      # src/bsdfs/blendbsdf.cpp, src/bsdfs/mask.cpp: over plain child BSDFs
TYPE_IDS = {"diffuse": DIFFUSE, "conductor": CONDUCTOR, "roughconductor": ROUGHCONDUCTOR, "dielectric": DIELECTRIC, "plastic": PLASTIC,
            "roughplastic": ROUGHPLASTIC, "roughdielectric": ROUGHDIELECTRIC, "thindielectric": THINDIELECTRIC}
SMOOTH = {DIFFUSE: True, CONDUCTOR: False, ROUGHCONDUCTOR: True, DIELECTRIC: False, PLASTIC: True, ROUGHPLASTIC: True, ROUGHDIELECTRIC: True,
          THINDIELECTRIC: False}      # BSDFFlags::Smooth
TRANSMISSIVE = {DIELECTRIC, ROUGHDIELECTRIC, THINDIELECTRIC}

# ior.h:23-50
IOR = {"vacuum": 1.0, "helium": 1.000036, "hydrogen": 1.000132, "air": 1.000277, "carbon dioxide": 1.00045, "water": 1.3330,
       "acetone": 1.36, "ethanol": 1.361, "carbon tetrachloride": 1.461, "glycerol": 1.4729, "benzene": 1.501, "silicone oil": 1.52045,
       "bromine": 1.661, "water ice": 1.31, "fused quartz": 1.458, "pyrex": 1.470, "acrylic glass": 1.49, "polypropylene": 1.49,
       "bk7": 1.5046, "sodium chloride": 1.544, "amber": 1.55, "pet": 1.5750, "diamond": 2.419}


def lookup_ior(value, default):
    """lookup_ior (ior.h:52-101): a float, or the name of a material"""
    if value is None:
        value = default
    if isinstance(value, (int, float)) and not isinstance(value, bool):
        return float(value)
    name = str(value).lower()
    if name not in IOR:
        raise RuntimeError('Unable to find an IOR value for "%s"! Valid choices are:%s' % (name, ", ".join(IOR)))
    return IOR[name]


def _is_uniform(v, default):
    """True if the parameter is a constant (`<spectrum value="c"/>` / a float: a `uniform` texture in the spectral variant,
    xml.cpp:1069-1083); False for an RGB colour (`srgb` texture, upsampled)"""
    if v is None:
        v = default
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _rgb(v, default):
    if v is None:
        v = default
    if isinstance(v, (int, float)):
        v = [float(v)] * 3
    a = np.asarray(v, dtype=np.float32).reshape(-1)
    if a.size == 1:
        a = np.repeat(a, 3)
    if a.size != 3:
        raise RuntimeError("expected a colour (3 values) or a constant")
    return [float(x) for x in a]


def is_transmissive(n):
    """a normalised record has a transmission component (what TwoSidedBRDF refuses, twosided.cpp:78-80)"""
    if n["type"] == MASK:
        return True
    if n["type"] == BLEND:
        return any(is_transmissive(c) for c in n["children"])
    return n["type"] in TRANSMISSIVE


def is_smooth(n):
    """BSDFFlags::Smooth of a normalised record (blend / mask: union of the nested flags)"""
    if n["type"] in (BLEND, MASK):
        return any(is_smooth(c) for c in n["children"])
    return SMOOTH[n["type"]]


def _normalize_nesting(b, t):
    """blendbsdf.cpp:57-79 / mask.cpp:67-91: child BSDFs in the order they are given, `weight` (required) / `opacity` (default 0.5) as
    a constant or a texture.  This backend nests one level: the children are plain BSDFs (their reflectance may be a `bitmap` /
    `checkerboard` texture in the RGB variant)."""
    pname = "weight" if t == "blendbsdf" else "opacity"
    bsdf_types = set(TYPE_IDS) | set(NESTING) | {"twosided"}
    children = [v for k, v in b.items() if k not in ("type", "id", pname) and isinstance(v, dict) and v.get("type", "diffuse") in bsdf_types]
    extra = [k for k, v in b.items() if k not in ("type", "id", pname) and not (isinstance(v, dict) and v.get("type", "diffuse") in bsdf_types)]
    if t == "blendbsdf":
        if len(children) > 2:
            raise RuntimeError("BlendBSDF: Cannot specify more than two child BSDFs")
        if pname not in b:
            raise RuntimeError('Property "weight" has not been specified!')
        if len(children) != 2:
            raise RuntimeError("BlendBSDF: Two child BSDFs must be specified!")
    else:
        if len(children) > 1:
            raise RuntimeError("Cannot specify more than one child BSDF")
        if not children:
            raise RuntimeError("Child BSDF not specified")
    if extra:
        raise RuntimeError('Error while loading: unreferenced property "%s" in bsdf plugin of type "%s"' % (extra[0], t))
    kids = [normalize(c) for c in children]
    for k in kids:
        if k["type"] in (BLEND, MASK):
            raise RuntimeError("%s: nested blendbsdf / mask children are not supported by this backend (one level of nesting)" % t)
    w = b.get(pname, 0.5)
    if isinstance(w, dict):
        if w.get("type") not in ("bitmap", "checkerboard"):
            raise RuntimeError("Texture plugin '%s' is not supported by this backend (bitmap, checkerboard)" % w.get("type"))
        weight = w
    else:
        a = np.asarray(w, dtype=np.float32).reshape(-1)
        if a.size != 1:
            raise RuntimeError("%s: '%s' is a scalar (Texture::eval_1): a constant or a texture" % (t, pname))
        weight = [float(a[0])] * 3
    out = dict(type=NESTING[t], twosided=False, reflectance=weight, specular_reflectance=[1.0] * 3, specular_transmittance=[1.0] * 3,
               eta=[0.0] * 3, k=[1.0] * 3, int_ior=1.0, ext_ior=1.0, alpha_u=0.1, alpha_v=0.1, distribution=0, sample_visible=True,
               nonlinear=False, uniform_mask=1, children=kids)
    if "id" in b:
        out["id"] = b["id"]
    return out


def flatten(records):
    """Top-level records followed by the children of blend / mask records; every nesting record gets `nested` = the table indices of
    its children (mtsamd_bsdf_desc::nested).  Shapes keep referring to the top-level indices."""
    flat = list(records)
    for r in records:
        if r["type"] in (BLEND, MASK):
            r["nested"] = [len(flat) + i for i in range(len(r["children"]))] + [-1] * (2 - len(r["children"]))
            flat.extend(r["children"])
    return flat


def normalize(b):
    """Plugin dictionary -> flat record: dict(type, twosided, reflectance (rgb list or bitmap dict), specular_reflectance,
    specular_transmittance, eta, k, int_ior, ext_ior, alpha_u, alpha_v, distribution, sample_visible, nonlinear)."""
    t = b.get("type", "diffuse")
    twosided = False
    if t == "twosided":                                      # twosided.cpp:63-92
        nested = [v for k, v in b.items() if k not in ("type", "id") and isinstance(v, dict)]
        if len(nested) == 0:
            raise RuntimeError("A nested one-sided material is required!")
        if len(nested) > 2:
            raise RuntimeError("At most two nested BSDFs can be specified!")
        if len(nested) == 2 and nested[0] != nested[1]:
            raise RuntimeError("twosided: two different nested BSDFs are not supported by this backend")
        out = normalize(nested[0])
        if is_transmissive(out):
            raise RuntimeError("Only materials without a transmission component can be nested!")
        out["twosided"] = True
        if "id" in b:
            out["id"] = b["id"]
        return out
    if t in NESTING:
        return _normalize_nesting(b, t)
    if t not in TYPE_IDS:
        raise RuntimeError("BSDF plugin '%s' is not supported by this backend (diffuse, conductor, roughconductor, dielectric, roughdielectric, thindielectric, plastic, roughplastic, twosided, blendbsdf, mask)" % t)
    tid = TYPE_IDS[t]
    out = dict(type=tid, twosided=twosided, reflectance=[0.5, 0.5, 0.5], specular_reflectance=[1.0] * 3, specular_transmittance=[1.0] * 3,
               eta=[0.0] * 3, k=[1.0] * 3, int_ior=1.0, ext_ior=1.0, alpha_u=0.1, alpha_v=0.1, distribution=0, sample_visible=True,
               nonlinear=False, uniform_mask=0)
    if "id" in b:
        out["id"] = b["id"]
    known = {"type", "id"}
    if tid == DIFFUSE:
        refl = b.get("reflectance", 0.5)                  # props.texture("reflectance", .5f): a uniform spectrum (diffuse.cpp:74)
        out["reflectance"] = refl if isinstance(refl, dict) else _rgb(refl, None)
        out["uniform_mask"] |= 1 if _is_uniform(refl, None) else 0
        known |= {"reflectance"}
    if tid in (CONDUCTOR, ROUGHCONDUCTOR):
        material = b.get("material", "none" if tid == CONDUCTOR else ("none" if "eta" in b else "Cu"))
        if "eta" in b or material == "none":
            if "material" in b and b["material"] != "none" and "eta" in b:
                raise RuntimeError("Should specify either (eta, k) or material, not both.")
            out["eta"], out["k"] = _rgb(b.get("eta"), 0.0), _rgb(b.get("k"), 1.0)
        else:
            raise RuntimeError('conductor material "%s": the measured IOR tables (data/ior/*.spd) are not shipped; specify eta and k' % material)
        out["specular_reflectance"] = _rgb(b.get("specular_reflectance"), 1.0)
        out["uniform_mask"] |= 2 if _is_uniform(b.get("specular_reflectance"), 1.0) else 0
        known |= {"material", "eta", "k", "specular_reflectance"}
    if tid in (ROUGHCONDUCTOR, ROUGHPLASTIC, ROUGHDIELECTRIC):
        distr = str(b.get("distribution", "beckmann")).lower()
        if distr not in ("beckmann", "ggx"):
            raise RuntimeError('Specified an invalid distribution "%s", must be "beckmann" or "ggx"!' % distr)
        out["distribution"] = 1 if distr == "ggx" else 0
        out["sample_visible"] = bool(b.get("sample_visible", True))
        if "alpha_u" in b or "alpha_v" in b:
            if "alpha_u" not in b or "alpha_v" not in b:
                raise RuntimeError("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.")
            if "alpha" in b:
                raise RuntimeError("Microfacet model: please specify either 'alpha' or 'alpha_u'/'alpha_v'.")
            out["alpha_u"], out["alpha_v"] = float(b["alpha_u"]), float(b["alpha_v"])
        else:
            out["alpha_u"] = out["alpha_v"] = float(b.get("alpha", 0.1))
        known |= {"distribution", "sample_visible", "alpha", "alpha_u", "alpha_v"}
        if tid == ROUGHPLASTIC and out["alpha_u"] != out["alpha_v"]:
            raise RuntimeError("The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!")
    if tid in (DIELECTRIC, PLASTIC, ROUGHPLASTIC, ROUGHDIELECTRIC, THINDIELECTRIC):
        out["int_ior"] = lookup_ior(b.get("int_ior"), "bk7" if tid in (DIELECTRIC, ROUGHDIELECTRIC, THINDIELECTRIC) else "polypropylene")
        out["ext_ior"] = lookup_ior(b.get("ext_ior"), "air")
        if out["int_ior"] < 0 or out["ext_ior"] < 0:
            raise RuntimeError("The interior and exterior indices of refraction must be positive!")
        if tid in (ROUGHPLASTIC, ROUGHDIELECTRIC) and out["int_ior"] == out["ext_ior"]:
            raise RuntimeError("The interior and exterior indices of refraction must be positive and differ!")
        out["specular_reflectance"] = _rgb(b.get("specular_reflectance"), 1.0)
        out["uniform_mask"] |= 2 if _is_uniform(b.get("specular_reflectance"), 1.0) else 0
        known |= {"int_ior", "ext_ior", "specular_reflectance"}
    if tid in (DIELECTRIC, ROUGHDIELECTRIC, THINDIELECTRIC):
        out["specular_transmittance"] = _rgb(b.get("specular_transmittance"), 1.0)
        out["uniform_mask"] |= 4 if _is_uniform(b.get("specular_transmittance"), 1.0) else 0
        known |= {"specular_transmittance"}
    if tid in (PLASTIC, ROUGHPLASTIC):
        dr = b.get("diffuse_reflectance")                  # a colour, a constant or a texture (bitmap / checkerboard)
        out["reflectance"] = dr if isinstance(dr, dict) else _rgb(dr, 0.5)
        out["uniform_mask"] |= 1 if _is_uniform(b.get("diffuse_reflectance"), 0.5) else 0
        out["nonlinear"] = bool(b.get("nonlinear", False))
        known |= {"diffuse_reflectance", "nonlinear"}
    extra = [k for k in b if k not in known]
    if extra:
        raise RuntimeError('Error while loading: unreferenced property "%s" in bsdf plugin of type "%s"' % (extra[0], t))
    return out
