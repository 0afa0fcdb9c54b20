"""Scene description loading: the subset of the reference's XML format
(``src/libcore/xml.cpp``) and ``load_dict`` needed to drive the hot path -- ``scene`` / ``integrator`` (path) /
``sensor`` (perspective, thinlens) / ``sampler`` (independent) / ``film`` (hdrfilm) / ``rfilter`` (gaussian, box, tent, catmullrom, mitchell, lanczos) / ``shape``
(obj, ply, rectangle) / ``bsdf`` (diffuse) / ``texture`` (bitmap) / ``emitter`` (area), with ``default`` + ``$param``
substitution, ``ref``/``id`` resolution, ``alias``, ``include`` and the ``transform`` operations.

Parsing is two-stage like the reference (``xml.cpp:327-933`` builds a property tree, ``xml.cpp:935-1060`` instantiates
it): :func:`parse_string` / :func:`parse_file` need no GPU and return a :class:`SceneDescription`; :func:`load_string` /
:func:`load_file` / :func:`load_dict` upload it (``render.Scene``).  Plugins outside the subset raise instead of being
approximated.  Error texts follow the reference's (``src/libcore/tests/test_xml.py``), without line/column positions.
"""
import math
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from . import loaders

F32 = np.float32

_OBJECT_TAGS = ("scene", "shape", "bsdf", "emitter", "sensor", "film", "sampler", "rfilter", "integrator", "texture", "spectrum_object", "medium", "phase", "volume")
_PROPERTY_TAGS = ("integer", "float", "string", "boolean", "rgb", "spectrum", "point", "vector", "transform", "ref")
_TRANSFORM_OPS = ("translate", "rotate", "scale", "lookat", "matrix")


class XMLError(RuntimeError):
    pass


# -------------------------------------------------------------------------------------------- transforms
def translate(v):
    m = np.eye(4, dtype=F32)
    m[:3, 3] = v
    return m


def scale(v):
    return np.diag(np.array([v[0], v[1], v[2], 1.0], dtype=F32))


def rotate(axis, angle_deg):
    """Transform4f::rotate (transform.h:175-178; enoki::rotate does not normalise the axis)."""
    a = np.asarray(axis, dtype=np.float64)
    t = math.radians(float(angle_deg))
    s, c = math.sin(t), math.cos(t)
    k = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    m = np.eye(4)
    m[:3, :3] = c * np.eye(3) + (1 - c) * np.outer(a, a) + s * k
    return m.astype(F32)


def look_at(origin, target, up):
    """Transform4f::look_at (transform.h:241-266)."""
    o, t, u = (np.asarray(x, dtype=F32) for x in (origin, target, up))
    with np.errstate(invalid="ignore", divide="ignore"):     # degenerate inputs produce NaNs, which the caller rejects
        d = t - o
        d = d / np.linalg.norm(d)
        left = np.cross(u, d)
        left = left / np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4, dtype=F32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


# -------------------------------------------------------------------------------------------- property tree
class Node:
    """One plugin instance of the property tree: tag (plugin class), type (plugin name), id, named properties."""

    def __init__(self, tag, type_=None, id_=None):
        self.tag, self.type, self.id = tag, type_, id_
        self.props = {}              # name -> python value | np.ndarray | Node | ("ref", id)
        self.children = []           # unnamed nested objects (Node or ("ref", id))
        self.queried = set()

    def set(self, name, value):
        if name in self.props:
            raise XMLError('Property "%s" was specified multiple times!' % name)
        self.props[name] = value

    def get(self, name, default=None, kind=None):
        """Properties::float_/int_/string/... (src/libcore/properties.cpp): typed lookup which marks the entry queried."""
        if name not in self.props:
            if default is None:
                raise XMLError('Property "%s" has not been specified!' % name)
            return default
        v = self.props[name]
        self.queried.add(name)
        if kind is not None:
            ok = {"float": isinstance(v, (float, int)) and not isinstance(v, bool), "int": isinstance(v, int) and not isinstance(v, bool),
                  "bool": isinstance(v, bool), "string": isinstance(v, str), "transform": isinstance(v, np.ndarray) and v.shape == (4, 4)}[kind]
            if not ok:
                raise XMLError('The property "%s" has the wrong type (expected <%s>).' % (name, {"int": "integer", "bool": "boolean"}.get(kind, kind)))
        return v

    def check_unqueried(self):
        """PluginManager::create_object's 'unreferenced property' check (xml.cpp:1015-1035)."""
        left = [k for k in self.props if k not in self.queried]
        if left:
            raise XMLError('Error while loading: unreferenced %s "%s" in %s plugin of type "%s"' %
                           ("property" if len(left) == 1 else "properties", '", "'.join(left), self.tag, self.type))


def _subst(value, params, where):
    """'$name' substitution (xml.cpp:336-358)."""
    if "$" not in value:
        return value
    for k in sorted(params, key=len, reverse=True):
        value = value.replace("$" + k, params[k])
    if "$" in value:
        raise XMLError('undefined parameter(s) in "%s" of element "%s"' % (value, where))
    return value


def _floats(text, n=None, what="floating point"):
    toks = [t for t in re.split(r"[\s,]+", text.strip()) if t]
    try:
        out = [float(t) for t in toks]
    except ValueError:
        raise XMLError('could not parse %s value "%s"' % (what, text))
    if n is not None and len(out) != n:
        raise XMLError('"%s": expected %d values' % (text, n))
    return out


def _check_attrs(el, allowed, required=True):
    for k in el.attrib:
        if k not in allowed:
            raise XMLError('unexpected attribute "%s" in element "%s".' % (k, el.tag))
    if required:
        for k in allowed:
            if k not in el.attrib and k not in ("id",):
                raise XMLError('missing attribute "%s" in element "%s".' % (k, el.tag))


def _xyz(el, default=0.0):
    """expand_value_to_xyz + parse_vector (xml.cpp:236-262,264-289)."""
    if "value" in el.attrib:
        v = _floats(el.attrib["value"])
        if len(v) == 1:
            v = v * 3
        if len(v) != 3:
            raise XMLError('"value" attribute must have exactly 1 or 3 elements')
        return v
    try:
        return [float(el.attrib.get(k, default)) for k in "xyz"]
    except ValueError:
        raise XMLError('could not parse floating point value in element "%s"' % el.tag)


class _Context:
    def __init__(self, params, base_dir):
        self.params = dict(params)
        self.base_dir = base_dir
        self.instances = {}          # id -> Node
        self.aliases = {}            # alias id -> target id
        self.tabulated = False       # a wavelength:value spectrum was seen (pre-integrated to RGB: RGB variants only)
        self.depth_includes = 0


def _parse_transform(el, ctx):
    _check_attrs(el, ("name",))
    m = np.eye(4, dtype=F32)
    for op in el:
        if op.tag not in _TRANSFORM_OPS:
            raise XMLError("transform nodes can only contain transform operations")
        a = {k: _subst(v, ctx.params, op.tag) for k, v in op.attrib.items()}
        op.attrib.update(a)
        if op.tag == "translate":
            _check_attrs(op, ("x", "y", "z", "value"), False)
            t = translate(_xyz(op))
        elif op.tag == "scale":
            _check_attrs(op, ("x", "y", "z", "value"), False)
            t = scale(_xyz(op, 1.0))
        elif op.tag == "rotate":
            _check_attrs(op, ("angle", "x", "y", "z", "value"), False)
            if "angle" not in op.attrib:
                raise XMLError('missing attribute "angle" in element "rotate".')
            t = rotate(_xyz(op), _floats(op.attrib["angle"], 1)[0])
        elif op.tag == "lookat":
            _check_attrs(op, ("origin", "target", "up"))
            t = look_at(_floats(op.attrib["origin"], 3), _floats(op.attrib["target"], 3), _floats(op.attrib["up"], 3))
            if np.isnan(t).any():
                raise XMLError("invalid lookat transformation")
        else:
            _check_attrs(op, ("value",))
            v = _floats(op.attrib["value"])
            if len(v) == 16:
                t = np.array(v, dtype=F32).reshape(4, 4)
            elif len(v) == 9:
                t = np.eye(4, dtype=F32)
                t[:3, :3] = np.array(v, dtype=F32).reshape(3, 3)
            else:
                raise XMLError("matrix: expected 16 or 9 values")
        m = (t.astype(np.float64) @ m.astype(np.float64)).astype(F32)       # ctx.transform = op * ctx.transform
    return m


def _parse_object(el, ctx, parent, is_root=False):
    tag = el.tag
    attrib = {k: _subst(v, ctx.params, tag) for k, v in el.attrib.items()}
    if tag == "scene":
        for k in attrib:
            if k != "version":
                raise XMLError('unexpected attribute "%s" in element "%s".' % (k, tag))
        node = Node("scene", "scene")
    else:
        for k in attrib:
            if k not in ("type", "id", "name"):
                raise XMLError('unexpected attribute "%s" in element "%s".' % (k, tag))
        if "type" not in attrib:
            raise XMLError('missing attribute "type" in element "%s".' % tag)
        node = Node(tag, attrib["type"], attrib.get("id"))
    if node.id is not None:
        if node.id.startswith("_"):
            raise XMLError('invalid id "%s" in element "%s": leading underscores are reserved for internal identifiers.' % (node.id, tag))
        if node.id in ctx.instances:
            raise XMLError('"%s" has duplicate id "%s"' % (tag, node.id))
        ctx.instances[node.id] = node
    for ch in el:
        _parse_child(ch, ctx, node)
    return node


def _parse_child(el, ctx, node):
    tag = el.tag
    if tag in _TRANSFORM_OPS:
        raise XMLError("transform operations can only occur in a transform node")
    if tag == "default":
        _check_attrs(el, ("name", "value"))
        ctx.params.setdefault(el.attrib["name"], _subst(el.attrib["value"], ctx.params, tag))
        return
    if tag == "alias":
        _check_attrs(el, ("id", "as"), False)
        if el.attrib.get("as") in ctx.instances or el.attrib.get("as") in ctx.aliases:
            raise XMLError('"alias" has duplicate id "%s"' % el.attrib.get("as"))
        ctx.aliases[el.attrib["as"]] = el.attrib["id"]
        return
    if tag == "include":
        _check_attrs(el, ("filename",))
        fn = _subst(el.attrib["filename"], ctx.params, tag)
        path = fn if os.path.isabs(fn) else os.path.join(ctx.base_dir, fn)
        if not os.path.exists(path):
            raise XMLError('included file "%s" not found!' % fn)
        ctx.depth_includes += 1
        if ctx.depth_includes > 15:
            raise XMLError("exceeded <include> recursion limit of 15")
        root = ET.parse(path).getroot()
        old = ctx.base_dir
        ctx.base_dir = os.path.dirname(os.path.abspath(path))
        if root.tag == "scene":          # a nested scene's children are spliced into the parent (xml.cpp:654-668)
            for ch in root:
                _parse_child(ch, ctx, node)
        else:
            _parse_child(root, ctx, node)
        ctx.base_dir = old
        ctx.depth_includes -= 1
        return
    if tag in _OBJECT_TAGS:
        child = _parse_object(el, ctx, node)
        name = el.attrib.get("name")
        if name is not None:
            node.set(name, child)
        else:
            node.children.append(child)
        return
    if tag not in _PROPERTY_TAGS:
        raise XMLError('unexpected tag "%s"' % tag)
    for ch in el:
        if tag == "transform":
            break
        raise XMLError('node "%s" cannot occur as child of a property' % ch.tag)
    a = {k: _subst(v, ctx.params, tag) for k, v in el.attrib.items()}
    name = a.get("name")
    if name is not None and name.startswith("_"):
        raise XMLError('invalid parameter name "%s" in element "%s": leading underscores are reserved for internal identifiers.' % (name, tag))
    if tag == "ref":
        for k in a:
            if k not in ("id", "name"):
                raise XMLError('unexpected attribute "%s" in element "%s".' % (k, tag))
        if "id" not in a:
            raise XMLError('missing attribute "id" in element "ref".')
        if name is not None:
            node.set(name, ("ref", a["id"]))
        else:
            node.children.append(("ref", a["id"]))
        return
    if tag == "transform":
        node.set(a.get("name", ""), _parse_transform(el, ctx))
        return
    if tag in ("point", "vector"):
        for k in a:
            if k not in ("name", "x", "y", "z", "value"):
                raise XMLError('unexpected attribute "%s" in element "%s".' % (k, tag))
        el.attrib.update(a)
        node.set(name, np.array(_xyz(el), dtype=F32))
        return
    for k in a:
        if k not in ("name", "value") and not (tag == "spectrum" and k == "filename"):
            raise XMLError('unexpected attribute "%s" in element "%s".' % (k, tag))
    for k in ("name", "value"):
        if k not in a and not (tag == "spectrum" and "filename" in a):
            raise XMLError('missing attribute "%s" in element "%s".' % (k, tag))
    v = a.get("value", "")
    if tag == "integer":
        if not re.fullmatch(r"\s*[+-]?\d+\s*", v):
            raise XMLError('could not parse integer value "%s".' % v)
        node.set(name, int(v))
    elif tag == "float":
        try:
            node.set(name, float(v))
        except ValueError:
            raise XMLError('could not parse floating point value "%s".' % v)
    elif tag == "boolean":
        if v.strip().lower() not in ("true", "false"):
            raise XMLError('could not parse boolean value "%s" -- must be "true" or "false".' % v)
        node.set(name, v.strip().lower() == "true")
    elif tag == "string":
        node.set(name, v)
    elif tag == "rgb":
        c = _floats(v)
        if len(c) == 1:
            c = c * 3
        if len(c) != 3:
            raise XMLError("'rgb' tag requires one or three values (got \"%s\")" % v)
        node.set(name, ("rgb", tuple(c)))
    elif tag == "spectrum":
        if ("filename" in a) == ("value" in a):
            raise XMLError("'spectrum' tag requires one of \"value\" or \"filename\" attributes")
        toks = v.replace(",", " ").split()
        if "filename" in a or len(toks) != 1:             # wavelength:value pairs, inline or from a file (xml.cpp:807-830)
            from . import spectrum as S
            if "filename" in a:
                fn = a["filename"]
                try:
                    wl, vals = S.spectrum_from_file(fn if os.path.isabs(fn) else os.path.join(ctx.base_dir, fn))
                except RuntimeError as e:
                    raise XMLError(str(e))
            else:
                wl, vals = [], []
                for tok in toks:
                    pair = tok.split(":")
                    if len(pair) != 2:
                        raise XMLError("invalid spectrum (expected wavelength:value pairs)")
                    try:
                        wl.append(float(pair[0])); vals.append(float(pair[1]))
                    except ValueError:
                        raise XMLError('could not parse wavelength:value pair: "%s"' % tok)
            node.set(name, ("tabulated", tuple(wl), tuple(vals)))
            ctx.tabulated = True
        else:
            try:
                node.set(name, ("spectrum", float(toks[0])))
            except ValueError:
                raise XMLError('could not parse constant spectrum "%s"' % toks[0])


def _resolve(ctx, item):
    if isinstance(item, tuple) and item and item[0] == "ref":
        rid = item[1]
        seen = set()
        while rid in ctx.aliases and rid not in seen:
            seen.add(rid)
            rid = ctx.aliases[rid]
        if rid not in ctx.instances:
            raise XMLError('reference to unknown object "%s"!' % rid)
        return ctx.instances[rid]
    return item


# -------------------------------------------------------------------------------------------- instantiation
class SceneDescription:
    """What the hot path needs from a scene file: the ``render.Scene`` dict plus sensor / integrator parameters."""

    def __init__(self):
        self.scene_dict = dict(meshes=[], bsdfs=[], emitters=[])
        self.sensors = []            # dicts of PerspectiveCamera / HDRFilm / IndependentSampler arguments
        self.integrator = None       # dict(max_depth, rr_depth)
        self.uses_tabulated_spectra = False


def _colour(value, what, emitter=False):
    """rgb / constant spectrum property -> linear RGB triple (create_texture_from_rgb / _from_spectrum, xml.cpp:1062-1110:
    in RGB mode a constant spectrum is a uniform value; for emitters it multiplies the D65 white point, i.e. RGB (1,1,1))."""
    if isinstance(value, tuple) and value[0] == "rgb":
        return [float(x) for x in value[1]]
    if isinstance(value, tuple) and value[0] == "spectrum":
        return [float(value[1])] * 3
    if isinstance(value, tuple) and value[0] == "tabulated":      # RGB variants: pre-integrated against the CIE observer
        from . import spectrum as S
        try:
            return S.tabulated_to_rgb(value[1], value[2], emitter, what.split(".")[-1])
        except RuntimeError as e:
            raise XMLError(str(e))
    if isinstance(value, (int, float)):
        return [float(value)] * 3
    raise XMLError("%s: expected an <rgb> or constant <spectrum> value" % what)


def _texture(ctx, node, base_dir):
    from . import bitmap
    to_uv = node.get("to_uv", np.eye(4, dtype=F32), "transform")
    if node.type == "checkerboard":                     # src/textures/checkerboard.cpp:40-44
        out = dict(type="checkerboard", to_uv=to_uv)
        for key, default in (("color0", 0.4), ("color1", 0.2)):
            v = _resolve(ctx, node.get(key, ("spectrum", default)))
            if isinstance(v, Node):
                raise XMLError("checkerboard: nested textures are not supported by this backend (constant colours only)")
            out[key] = _colour(v, "checkerboard." + key)
        node.check_unqueried()
        return out
    if node.type != "bitmap":
        raise XMLError('Texture plugin "%s" is not supported by this backend (bitmap, checkerboard)' % node.type)
    fn = node.get("filename", kind="string")
    path = fn if os.path.isabs(fn) else os.path.join(base_dir, fn)
    raw = node.get("raw", False, "bool")
    ft = node.get("filter_type", "bilinear", "string")
    wm = node.get("wrap_mode", "repeat", "string")
    if ft != "bilinear" or wm != "repeat":
        raise XMLError("bitmap texture: only filter_type=bilinear / wrap_mode=repeat are supported by this backend")
    node.check_unqueried()
    return dict(type="bitmap", data=bitmap.read_rgb(path, linearize=not raw), to_uv=to_uv)


def _bsdf_plugin_dict(ctx, node, base_dir):
    """property tree of a <bsdf> -> plugin dictionary understood by mitsuba2_amd.bsdfs.normalize"""
    if node.tag != "bsdf":
        raise XMLError('expected a bsdf, got "%s"' % node.tag)
    d = {"type": node.type}
    if node.id is not None:
        d["id"] = node.id
    nested = 0
    for k, v in list(node.props.items()):
        v = _resolve(ctx, v)
        node.queried.add(k)
        if isinstance(v, Node):
            if v.tag == "texture":
                d[k] = _texture(ctx, v, base_dir)
            elif v.tag == "bsdf":
                d["bsdf_%d" % nested] = _bsdf_plugin_dict(ctx, v, base_dir)
                nested += 1
            else:
                raise XMLError('bsdf: unexpected nested object "%s"' % v.tag)
        elif isinstance(v, tuple) and v and v[0] == "rgb":
            d[k] = _colour(v, k)
        elif isinstance(v, tuple) and v and v[0] == "spectrum":
            d[k] = float(v[1])                           # a constant: `uniform` spectrum in the spectral variant
        elif isinstance(v, tuple) and v and v[0] == "tabulated":
            d[k] = _colour(v, k)
        else:
            d[k] = v
    for c in node.children:
        c = _resolve(ctx, c)
        if c.tag != "bsdf":
            raise XMLError('bsdf: unexpected nested object "%s"' % c.tag)
        d["bsdf_%d" % nested] = _bsdf_plugin_dict(ctx, c, base_dir)
        nested += 1
    return d


def _bsdf(ctx, node, desc, cache, base_dir):
    if id(node) in cache:
        return cache[id(node)]
    from . import bsdfs
    entry = _bsdf_plugin_dict(ctx, node, base_dir)
    try:
        bsdfs.normalize(entry)                       # constructor-time validation (unknown plugin, bad parameters)
    except RuntimeError as e:
        raise XMLError(str(e))
    desc.scene_dict["bsdfs"].append(entry)
    cache[id(node)] = len(desc.scene_dict["bsdfs"]) - 1
    return cache[id(node)]


def _shape(ctx, node, desc, cache, base_dir):
    to_world = node.get("to_world", np.eye(4, dtype=F32), "transform")
    if node.type in ("obj", "ply", "serialized"):
        fn = node.get("filename", kind="string")
        path = fn if os.path.isabs(fn) else os.path.join(base_dir, fn)
        if not os.path.exists(path):
            raise XMLError('"%s": file does not exist!' % path)
        face_normals = node.get("face_normals", False, "bool")
        if node.type == "obj":
            mesh = loaders.load_obj(path, to_world, node.get("flip_tex_coords", True, "bool"), face_normals)
        elif node.type == "serialized":
            mesh = loaders.load_serialized(path, node.get("shape_index", 0, "int"), to_world, face_normals)
        else:
            mesh = loaders.load_ply(path, to_world, face_normals)
    elif node.type == "rectangle":
        mesh = loaders.rectangle(to_world, node.get("flip_normals", False, "bool"))
    else:
        raise XMLError('Shape plugin "%s" is not supported by this backend (obj, ply, serialized, rectangle)' % node.type)
    bsdf, emitter = None, None
    items = [_resolve(ctx, c) for c in node.children] + [_resolve(ctx, v) for k, v in node.props.items() if isinstance(v, (Node, tuple)) and (isinstance(v, Node) or v[0] == "ref")]
    for k, v in node.props.items():
        if isinstance(v, Node) or (isinstance(v, tuple) and v[0] == "ref"):
            node.queried.add(k)
    for it in items:
        if it.tag == "bsdf":
            if bsdf is not None:
                raise XMLError("Only a single BSDF child object can be specified per shape.")
            bsdf = _bsdf(ctx, it, desc, cache, base_dir)
        elif it.tag == "emitter":
            if emitter is not None:
                raise XMLError("Only a single Emitter child object can be specified per shape.")
            if it.type != "area":
                raise XMLError('Emitter plugin "%s" is not supported by this backend (area only)' % it.type)
            rad = _colour(_resolve(ctx, it.get("radiance", ("spectrum", 1.0))), "area.radiance", True)
            it.check_unqueried()
            desc.scene_dict["emitters"].append(dict(type="area", radiance=rad))
            emitter = len(desc.scene_dict["emitters"]) - 1
        else:
            raise XMLError('Tried to add an unsupported object of type "%s" to a shape' % it.tag)
    if bsdf is None:                 # Shape::Shape: default BSDF is `diffuse` (src/librender/shape.cpp:48-50)
        desc.scene_dict["bsdfs"].append(dict(type="diffuse", reflectance=[0.5, 0.5, 0.5]))
        bsdf = len(desc.scene_dict["bsdfs"]) - 1
    node.check_unqueried()
    mesh.update(bsdf=bsdf, emitter=-1 if emitter is None else emitter)
    if node.id is not None:
        mesh["id"] = node.id
    desc.scene_dict["meshes"].append(mesh)


def _sensor(ctx, node):
    if node.type not in ("perspective", "thinlens"):
        raise XMLError('Sensor plugin "%s" is not supported by this backend (perspective, thinlens)' % node.type)
    out = dict(type=node.type, to_world=node.get("to_world", np.eye(4, dtype=F32), "transform"), near_clip=node.get("near_clip", 1e-2, "float"),
               far_clip=node.get("far_clip", 1e4, "float"), fov_axis=node.get("fov_axis", "x", "string"), fov=None, focal_length=None)
    if "fov" in node.props:
        out["fov"] = node.get("fov", kind="float")
    if "focal_length" in node.props:
        out["focal_length"] = node.get("focal_length", kind="string")
    out["focus_distance"] = node.get("focus_distance", out["far_clip"], "float")       # ProjectiveCamera (sensor.cpp:104); unused by a pinhole camera
    if node.type == "thinlens":
        if "aperture_radius" not in node.props:
            raise XMLError('Property "aperture_radius" has not been specified!')          # thinlens.cpp:112
        out["aperture_radius"] = node.get("aperture_radius", kind="float")
    if node.get("shutter_close", 0.0, "float") != node.get("shutter_open", 0.0, "float"):
        raise XMLError("sensor: a non-zero shutter time (motion blur) is not supported by this backend")
    film = dict(width=768, height=576, crop_offset=None, crop_size=None, rfilter=("gaussian", 0.5))
    sampler = dict(sample_count=4, seed=0)
    for it in [_resolve(ctx, c) for c in node.children]:
        if it.tag == "film":
            if it.type != "hdrfilm":
                raise XMLError('Film plugin "%s" is not supported by this backend (hdrfilm only)' % it.type)
            film["width"], film["height"] = it.get("width", 768, "int"), it.get("height", 576, "int")
            cw, ch = it.get("crop_width", film["width"], "int"), it.get("crop_height", film["height"], "int")
            cx, cy = it.get("crop_offset_x", 0, "int"), it.get("crop_offset_y", 0, "int")
            film["crop_offset"], film["crop_size"] = (cx, cy), (cw, ch)
            for k, kind in (("file_format", "string"), ("pixel_format", "string"), ("component_format", "string"), ("high_quality_edges", "bool")):
                if k in it.props:
                    film[k] = it.get(k, kind=kind)
            for f in [_resolve(ctx, c) for c in it.children]:
                if f.tag != "rfilter":
                    raise XMLError('film: unexpected child "%s"' % f.tag)
                if f.type == "gaussian":
                    film["rfilter"] = ("gaussian", f.get("stddev", 0.5, "float"))
                elif f.type == "box":
                    film["rfilter"] = ("box", f.get("radius", 0.5, "float")) if "radius" in f.props else ("box", 0.5)
                elif f.type in ("tent", "catmullrom"):
                    film["rfilter"] = (f.type,)
                elif f.type == "mitchell":                       # mitchell.cpp:33-37
                    film["rfilter"] = ("mitchell", f.get("B", 1.0 / 3.0, "float"), f.get("C", 1.0 / 3.0, "float"))
                elif f.type == "lanczos":                        # lanczos.cpp:34
                    film["rfilter"] = ("lanczos", f.get("lobes", 3, "int"))
                else:
                    raise XMLError('Reconstruction filter "%s" is not supported by this backend (gaussian, box, tent, catmullrom, mitchell, lanczos)' % f.type)
                f.check_unqueried()
            it.check_unqueried()
        elif it.tag == "sampler":
            if it.type != "independent":
                raise XMLError('Sampler plugin "%s" is not supported by this backend (independent only)' % it.type)
            sampler = dict(sample_count=it.get("sample_count", 4, "int"), seed=it.get("seed", 0, "int"))
            it.check_unqueried()
        else:
            raise XMLError('sensor: unexpected child "%s"' % it.tag)
    node.check_unqueried()
    out.update(film=film, sampler=sampler)
    return out


def _integrator(ctx, it):
    if it.type == "path":
        out = dict(type="path", max_depth=it.get("max_depth", -1, "int"), rr_depth=it.get("rr_depth", 5, "int"))
    elif it.type == "direct":
        out = dict(type="direct", hide_emitters=it.get("hide_emitters", False, "bool"))
        for k in ("shading_samples", "emitter_samples", "bsdf_samples"):
            if k in it.props:
                out[k] = it.get(k, kind="int")
    elif it.type == "depth":
        out = dict(type="depth")
    elif it.type == "moment":                            # moment.cpp:36-55: nested sampling integrators, named by their property name
        nested = [(k, _resolve(ctx, v)) for k, v in it.props.items()] + [("integrator_%d" % i, _resolve(ctx, c)) for i, c in enumerate(it.children)]
        for k, _ in nested:
            it.queried.add(k)
        nested = [(k, v) for k, v in nested if isinstance(v, Node)]
        if len(nested) != 1 or nested[0][1].tag != "integrator":
            raise XMLError("moment: exactly one nested integrator is supported by this backend")
        out = dict(type="moment", name=nested[0][0], nested=_integrator(ctx, nested[0][1]))
    else:
        raise XMLError('Integrator plugin "%s" is not supported by this backend (path, direct, depth, moment)' % it.type)
    return out


def _make_integrator(spec):
    from . import render as R
    args = {k: v for k, v in spec.items() if k not in ("type", "nested", "name")}
    if spec.get("type") == "moment":
        return R.MomentIntegrator(_make_integrator(spec["nested"]), spec.get("name", "nested"))
    return {"path": R.PathIntegrator, "direct": R.DirectIntegrator, "depth": R.DepthIntegrator}[spec.get("type", "path")](**args)


def _instantiate(ctx, root, base_dir):
    if root.tag != "scene":
        raise XMLError('root element "%s" must be a scene in this backend' % root.tag)
    desc = SceneDescription()
    desc.uses_tabulated_spectra = ctx.tabulated
    cache = {}
    for k, v in root.props.items():
        if not isinstance(v, (Node, tuple)) or (isinstance(v, tuple) and v[0] != "ref"):
            raise XMLError('Error while loading: unreferenced property "%s" in scene' % k)
    items = [_resolve(ctx, c) for c in root.children] + [_resolve(ctx, v) for v in root.props.values()]
    for it in items:
        if it.tag == "shape":
            _shape(ctx, it, desc, cache, base_dir)
        elif it.tag == "bsdf":
            _bsdf(ctx, it, desc, cache, base_dir)
        elif it.tag == "sensor":
            desc.sensors.append(_sensor(ctx, it))
        elif it.tag == "integrator":
            desc.integrator = _integrator(ctx, it)
            it.check_unqueried()
        elif it.tag == "texture":
            continue                                 # instantiated where referenced
        elif it.tag == "emitter":
            if it.type == "constant":
                rad = _colour(_resolve(ctx, it.get("radiance", ("spectrum", 1.0))), "constant.radiance", True)
                it.check_unqueried()
                desc.scene_dict["emitters"].append(dict(type="constant", radiance=rad))
            elif it.type == "envmap":
                from . import bitmap
                fn = it.get("filename", kind="string")
                path = fn if os.path.isabs(fn) else os.path.join(base_dir, fn)
                if not os.path.exists(path):
                    raise XMLError('"%s": file does not exist!' % path)
                entry = dict(type="envmap", data=bitmap.read_rgb(path), scale=it.get("scale", 1.0, "float"),
                             to_world=it.get("to_world", np.eye(4, dtype=F32), "transform"))
                if it.id is not None:
                    entry["id"] = it.id                  # 'my_envmap.data' of mitsuba.python.util.traverse
                it.check_unqueried()
                desc.scene_dict["emitters"].append(entry)
            elif it.type in ("point", "spot", "directional"):       # point.cpp:52-65, spot.cpp:68-91, directional.cpp:43-63
                key = "irradiance" if it.type == "directional" else "intensity"
                entry = {"type": it.type, key: _colour(_resolve(ctx, it.get(key, ("spectrum", 1.0))), "%s.%s" % (it.type, key), True)}
                if "to_world" in it.props:
                    entry["to_world"] = it.get("to_world", kind="transform")
                if it.type == "point" and "position" in it.props:
                    entry["position"] = [float(x) for x in it.get("position")]
                if it.type == "directional" and "direction" in it.props:
                    entry["direction"] = [float(x) for x in it.get("direction")]
                if it.type == "spot":
                    for k in ("cutoff_angle", "beam_width"):
                        if k in it.props:
                            entry[k] = it.get(k, kind="float")
                    if any(c[0] == "texture" if isinstance(c, tuple) else getattr(c, "tag", "") == "texture" for c in it.children) or "texture" in it.props:
                        raise XMLError("spot: projection textures are not supported by this backend")
                it.check_unqueried()
                from . import emitters as E
                try:
                    E.normalize(entry)                              # the constructors' consistency checks
                except RuntimeError as err:
                    raise XMLError(str(err))
                desc.scene_dict["emitters"].append(entry)
            else:
                raise XMLError('Emitter plugin "%s" is not supported by this backend (area emitters attached to shapes, constant, envmap, point, spot, directional)' % it.type)
        else:
            raise XMLError('scene: unsupported child "%s"' % it.tag)
    return desc


def parse_string(string, base_dir=".", **params):
    """xml.load_string up to (not including) the device upload.  Keyword arguments are the ``-Dkey=value`` parameters."""
    try:
        root_el = ET.fromstring(string.strip() if isinstance(string, str) else string)
    except ET.ParseError as e:
        raise XMLError('Error while loading "<string>": %s' % e)
    if root_el.tag not in _OBJECT_TAGS:
        if root_el.tag in _PROPERTY_TAGS or root_el.tag in _TRANSFORM_OPS:
            raise XMLError('root element "%s" must be an object' % root_el.tag)
        raise XMLError('unexpected tag "%s"' % root_el.tag)
    if root_el.tag == "scene" and "version" not in root_el.attrib:
        raise XMLError('missing attribute "version" in element "scene".')
    ctx = _Context({k: str(v) for k, v in params.items()}, base_dir)
    root = _parse_object(root_el, ctx, None, True)
    return _instantiate(ctx, root, base_dir)


def parse_file(path, **params):
    if not os.path.exists(path):
        raise XMLError('"%s": file does not exist!' % path)
    with open(path, "r") as fh:
        return parse_string(fh.read(), os.path.dirname(os.path.abspath(path)), **params)


def instantiate(desc, device=0, variant="rgb"):
    """SceneDescription -> render.Scene with its sensors and integrator (needs the HIP library and a GPU)."""
    from . import render as R
    if variant == "spectral" and desc.uses_tabulated_spectra:
        raise XMLError("wavelength:value spectra are pre-integrated to RGB by this backend: RGB variant only")
    sensors = []
    for s in desc.sensors:
        f = s["film"]
        flt = R.make_filter(f["rfilter"][0], *f["rfilter"][1:])
        extra = {k: f[k] for k in ("file_format", "pixel_format", "component_format", "high_quality_edges") if k in f}
        film = R.HDRFilm(f["width"], f["height"], f["crop_offset"], f["crop_size"], flt, **extra)
        sampler = R.IndependentSampler(s["sampler"]["sample_count"], s["sampler"]["seed"])
        kw = dict(to_world=s["to_world"], fov=s["fov"], focal_length=s["focal_length"], fov_axis=s["fov_axis"],
                  near_clip=s["near_clip"], far_clip=s["far_clip"], film=film, sampler=sampler)
        if s.get("type", "perspective") == "thinlens":
            sensors.append(R.ThinLensCamera(aperture_radius=s["aperture_radius"], focus_distance=s.get("focus_distance"), **kw))
        else:
            sensors.append(R.PerspectiveCamera(**kw))
    integ = _make_integrator(desc.integrator) if desc.integrator is not None else None
    scene = R.Scene(desc.scene_dict, device=device, integrator=integ, variant=variant)
    scene._sensors = sensors
    return scene


def load_string(string, device=0, variant="rgb", base_dir=".", **params):
    """mitsuba.core.xml.load_string (src/libcore/python/xml.cpp) for the supported subset."""
    return instantiate(parse_string(string, base_dir, **params), device, variant)


def load_file(path, device=0, variant="rgb", **params):
    """mitsuba.core.xml.load_file"""
    return instantiate(parse_file(path, **params), device, variant)


# -------------------------------------------------------------------------------------------- load_dict
_PLUGIN_CLASS = {"twosided": "bsdf", "blendbsdf": "bsdf", "mask": "bsdf", "conductor": "bsdf", "roughconductor": "bsdf", "dielectric": "bsdf", "plastic": "bsdf", "roughplastic": "bsdf", "roughdielectric": "bsdf", "thindielectric": "bsdf", "path": "integrator", "perspective": "sensor", "thinlens": "sensor", "hdrfilm": "film", "independent": "sampler", "gaussian": "rfilter", "box": "rfilter", "tent": "rfilter", "catmullrom": "rfilter", "mitchell": "rfilter", "lanczos": "rfilter",
                 "direct": "integrator", "depth": "integrator", "moment": "integrator", "obj": "shape", "ply": "shape", "serialized": "shape", "rectangle": "shape", "diffuse": "bsdf", "area": "emitter", "constant": "emitter", "envmap": "emitter", "point": "emitter", "spot": "emitter", "directional": "emitter", "bitmap": "texture", "checkerboard": "texture", "scene": "scene"}


def _node_from_dict(d, ctx):
    if "type" not in d:
        raise XMLError("Missing key 'type'!")
    t = d["type"]
    if t not in _PLUGIN_CLASS:
        raise XMLError('Plugin "%s" is not supported by this backend' % t)
    node = Node(_PLUGIN_CLASS[t], t, d.get("id"))
    for k, v in d.items():
        if k in ("type", "id"):
            continue
        if isinstance(v, dict):
            t2 = v.get("type")
            if t2 == "rgb":
                if len(v) != 2:
                    raise XMLError("'rgb' dictionary should always contain 2 entries ('type' and 'value'), got %u." % len(v))
                node.set(k, ("rgb", tuple(float(x) for x in v["value"])))
            elif t2 == "spectrum":
                if len(v) != 2:
                    raise XMLError("'spectrum' dictionary should always contain 2 entries ('type' and 'value'), got %u." % len(v))
                if not isinstance(v.get("value"), (int, float)):
                    raise XMLError("tabulated spectra are not supported by this backend (constant values only)")
                node.set(k, ("spectrum", float(v["value"])))
            elif t2 == "ref":
                if node.tag == "scene":
                    raise XMLError("Reference found at the scene level: %s" % k)
                if v.get("id") not in ctx.instances:
                    raise XMLError('Referenced id "%s" not found: %s' % (v.get("id"), k))
                node.children.append(("ref", v["id"]))
            else:
                child = _node_from_dict(v, ctx)
                if node.tag == "scene":                    # referencable by key and by id (xml_v.cpp:228-243)
                    for rid in {k, child.id} - {None}:
                        if rid in ctx.instances:
                            raise XMLError("%s has duplicate id: %s" % (k, rid))
                        ctx.instances[rid] = child
                if child.tag == "texture":
                    node.set(k, child)
                else:
                    node.children.append(child)
        elif isinstance(v, np.ndarray) and v.shape == (4, 4):
            node.set(k, v.astype(F32))
        elif isinstance(v, (list, tuple, np.ndarray)):
            node.set(k, np.asarray(v, dtype=F32))
        else:
            node.set(k, v)
    return node


def parse_dict(d, base_dir="."):
    """mitsuba.core.xml.load_dict (src/libcore/python/xml_v.cpp:150-330) up to the device upload: nested dictionaries with
    a 'type' key per plugin; ``{'type': 'rgb', 'value': [...]}``, ``{'type': 'ref', 'id': ...}`` as in the reference."""
    ctx = _Context({}, base_dir)
    root = _node_from_dict(d, ctx)
    return _instantiate(ctx, root, base_dir)


def load_dict(d, device=0, variant="rgb", base_dir="."):
    return instantiate(parse_dict(d, base_dir), device, variant)
