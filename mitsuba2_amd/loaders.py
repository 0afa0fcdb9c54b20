"""Mesh ingestion: Wavefront OBJ and Stanford PLY loaders and vertex-normal
generation, producing the ``Mesh`` buffers the C ABI consumes (``include/mitsuba/render/mesh.h:80-90``).

Semantics follow ``src/shapes/obj.cpp:94-342`` (``to_world`` baked at load time, v/vt/vn triples de-duplicated,
``flip_tex_coords`` default true, polygons fan-triangulated, smooth normals generated when the file has none unless
``face_normals``), ``src/shapes/ply.cpp`` (ascii / binary PLY with x y z [nx ny nz] [u v | s t] and a face list) and
``Mesh::recompute_vertex_normals`` (``src/librender/mesh.cpp:199-253``: Thuermer & Wuethrich angle weighting).
"""
import struct

import numpy as np

F32 = np.float32


def transform_points(to_world, p):
    """Transform::transform_affine for points (transform.h:81-89)."""
    if to_world is None:
        return np.ascontiguousarray(p, dtype=F32)
    m = np.asarray(to_world, dtype=F32).reshape(4, 4)
    p = np.asarray(p, dtype=F32).reshape(-1, 3)
    return (p @ m[:3, :3].T + m[:3, 3]).astype(F32)


def transform_normals(to_world, n):
    """Normals transform with the inverse transpose and are re-normalised (obj.cpp:180-184)."""
    n = np.asarray(n, dtype=F32).reshape(-1, 3)
    if to_world is not None:
        m = np.asarray(to_world, dtype=np.float64).reshape(4, 4)
        n = (n.astype(np.float64) @ np.linalg.inv(m[:3, :3])).astype(F32)        # (M^-T n)^T = n^T M^-1
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return (n / np.where(ln > 0, ln, 1)).astype(F32)


def _unit_angle(a, b):
    """enoki::unit_angle: numerically robust angle between unit vectors."""
    d = np.sum(a * b, axis=-1)
    t = 2.0 * np.arcsin(np.clip(0.5 * np.linalg.norm(b - np.where(d[..., None] >= 0, a, -a), axis=-1), -1, 1))
    return np.where(d >= 0, t, np.pi - t)


def compute_vertex_normals(positions, faces):
    """Mesh::recompute_vertex_normals (mesh.cpp:199-253): face normals weighted by the face angle at each vertex;
    vertices without a valid normal get the 'bogus' value (1, 0, 0)."""
    p = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    v0, v1, v2 = p[f[:, 0]], p[f[:, 1]], p[f[:, 2]]
    n = np.cross(v1 - v0, v2 - v0)
    l2 = np.sum(n * n, axis=1)
    ok = l2 > 0
    n[ok] /= np.sqrt(l2[ok])[:, None]

    def nz(v):
        ln = np.linalg.norm(v, axis=1, keepdims=True)
        return v / np.where(ln > 0, ln, 1)

    ang = np.stack([_unit_angle(nz(v1 - v0), nz(v2 - v0)), _unit_angle(nz(v2 - v1), nz(v0 - v1)), _unit_angle(nz(v0 - v2), nz(v1 - v2))], axis=1)
    out = np.zeros_like(p)
    for j in range(3):
        np.add.at(out, f[ok, j], n[ok] * ang[ok, j:j + 1])
    ln = np.linalg.norm(out, axis=1)
    good = ln != 0
    out[good] /= ln[good, None]
    out[~good] = [1.0, 0.0, 0.0]
    return out.astype(F32)


def load_obj(path, to_world=None, flip_tex_coords=True, face_normals=False):
    """OBJMesh (obj.cpp:94-342) -> dict(positions, faces, normals, texcoords)."""
    verts, norms, uvs = [], [], []
    tris, key_to_id, keys = [], {}, []
    with open(path, "r") as fh:
        for line in fh:
            cur = line.strip()
            if len(cur) >= 1025:
                raise RuntimeError('Error while loading OBJ file "%s": file contains an excessively long line!' % path)
            if cur.startswith("v ") or cur.startswith("v\t"):
                verts.append([float(x) for x in cur.split()[1:4]])
            elif cur.startswith("vn"):
                norms.append([float(x) for x in cur.split()[1:4]])
            elif cur.startswith("vt"):
                uv = [float(x) for x in cur.split()[1:3]]
                if flip_tex_coords:
                    uv[1] = 1.0 - uv[1]
                uvs.append(uv)
            elif cur.startswith("f ") or cur.startswith("f\t"):
                ids = []
                for tok in cur.split()[1:]:
                    parts = tok.split("/")
                    if len(parts) > 3:
                        raise RuntimeError('Error while loading OBJ file "%s": could not parse line "%s"' % (path, cur))
                    key = tuple(int(x) if x else 0 for x in parts) + (0,) * (3 - len(parts))
                    if key[0] - 1 >= len(verts) or key[0] < 1:
                        raise RuntimeError('Error while loading OBJ file "%s": reference to invalid vertex %d!' % (path, key[0]))
                    if key not in key_to_id:
                        key_to_id[key] = len(keys)
                        keys.append(key)
                    ids.append(key_to_id[key])
                for k in range(2, len(ids)):                      # fan triangulation (obj.cpp:255-264)
                    tris.append([ids[0], ids[k - 1], ids[k]])
    if not tris:
        raise RuntimeError('Error while loading OBJ file "%s": no faces' % path)
    verts = transform_points(to_world, np.array(verts, dtype=F32).reshape(-1, 3))
    if not np.isfinite(verts).all():
        raise RuntimeError('Error while loading OBJ file "%s": mesh contains invalid vertex position data' % path)
    keys = np.array(keys, dtype=np.int64)
    positions = verts[keys[:, 0] - 1]
    faces = np.array(tris, dtype=np.uint32)
    texcoords = None
    if uvs:
        uv = np.array(uvs, dtype=F32)
        texcoords = np.zeros((len(keys), 2), F32)
        has = keys[:, 1] > 0
        if (keys[has, 1] - 1 >= len(uv)).any():
            raise RuntimeError('Error while loading OBJ file "%s": reference to invalid texture coordinate!' % path)
        texcoords[has] = uv[keys[has, 1] - 1]
    normals = None
    if not face_normals:
        if norms:
            nn = transform_normals(to_world, np.array(norms, dtype=F32))
            normals = np.zeros((len(keys), 3), F32)
            has = keys[:, 2] > 0
            if (keys[has, 2] - 1 >= len(nn)).any():
                raise RuntimeError('Error while loading OBJ file "%s": reference to invalid normal!' % path)
            normals[has] = nn[keys[has, 2] - 1]
        else:
            normals = compute_vertex_normals(positions, faces)
    return dict(positions=np.ascontiguousarray(positions, F32), faces=faces, normals=normals, texcoords=texcoords)


_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def load_ply(path, to_world=None, face_normals=False):
    """PLYMesh (src/shapes/ply.cpp): ascii and binary PLY -> dict(positions, faces, normals, texcoords)."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise RuntimeError('Error while loading PLY file "%s": invalid PLY header' % path)
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise RuntimeError('Error while loading PLY file "%s": unexpected end of header' % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append(dict(name=tok[1], count=int(tok[2]), props=[]))
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    elements[-1]["props"].append(("scalar", tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise RuntimeError('Error while loading PLY file "%s": unknown format "%s"' % (path, fmt))
        endian = "<" if fmt != "binary_big_endian" else ">"
        data = {}
        for el in elements:
            rows = []
            for _ in range(el["count"]):
                row = {}
                if fmt == "ascii":
                    vals = fh.readline().split()
                    pos = 0
                    for pr in el["props"]:
                        if pr[0] == "scalar":
                            row[pr[2]] = float(vals[pos]); pos += 1
                        else:
                            n = int(vals[pos]); pos += 1
                            row[pr[3]] = [int(float(v)) for v in vals[pos:pos + n]]; pos += n
                else:
                    for pr in el["props"]:
                        if pr[0] == "scalar":
                            c = _PLY_TYPES[pr[1]]
                            row[pr[2]] = struct.unpack(endian + c, fh.read(struct.calcsize(c)))[0]
                        else:
                            cc, ci = _PLY_TYPES[pr[1]], _PLY_TYPES[pr[2]]
                            n = struct.unpack(endian + cc, fh.read(struct.calcsize(cc)))[0]
                            row[pr[3]] = list(struct.unpack(endian + ci * n, fh.read(struct.calcsize(ci) * n)))
                rows.append(row)
            data[el["name"]] = rows
    if "vertex" not in data or "face" not in data:
        raise RuntimeError('Error while loading PLY file "%s": vertex / face elements missing' % path)
    v = data["vertex"]
    positions = transform_points(to_world, np.array([[r["x"], r["y"], r["z"]] for r in v], dtype=F32).reshape(-1, 3))
    tris = []
    for r in data["face"]:
        idx = r.get("vertex_index", r.get("vertex_indices"))
        if idx is None or len(idx) < 3:
            raise RuntimeError('Error while loading PLY file "%s": face without vertex indices' % path)
        for k in range(2, len(idx)):
            tris.append([idx[0], idx[k - 1], idx[k]])
    faces = np.array(tris, dtype=np.uint32).reshape(-1, 3)
    if faces.size and faces.max() >= len(positions):
        raise RuntimeError('Error while loading PLY file "%s": vertex index out of range' % path)
    normals = texcoords = None
    if v and "u" in v[0] and "v" in v[0]:
        texcoords = np.array([[r["u"], r["v"]] for r in v], dtype=F32)
    elif v and "s" in v[0] and "t" in v[0]:
        texcoords = np.array([[r["s"], r["t"]] for r in v], dtype=F32)
    if not face_normals:
        if v and "nx" in v[0]:
            normals = transform_normals(to_world, np.array([[r["nx"], r["ny"], r["nz"]] for r in v], dtype=F32))
        else:
            normals = compute_vertex_normals(positions, faces)
    return dict(positions=positions, faces=faces, normals=normals, texcoords=texcoords)


def load_serialized(path, shape_index=0, to_world=None, face_normals=False):
    """SerializedMesh (src/shapes/serialized.cpp:190-336): Mitsuba's compressed binary mesh format.  Header 0x041C + version 3 / 4,
    a zlib stream per sub-mesh (flags, [name], vertex / face counts, positions, [normals], [texcoords], [colours], indices) and an
    end-of-file dictionary with the offset of every sub-mesh."""
    import zlib
    def fail(msg):
        raise RuntimeError('Error while loading serialized file "%s": %s!' % (path, msg))
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 4:
        fail("encountered an invalid file format")
    fmt, version = struct.unpack_from("<hh", buf, 0)
    if fmt != 0x041C:
        fail("encountered an invalid file format")
    if version not in (3, 4):
        fail("encountered an incompatible file version")
    if shape_index < 0:
        fail("shape index must be nonnegative")
    offset = 0
    if shape_index != 0:
        count = struct.unpack_from("<I", buf, len(buf) - 4)[0]
        if shape_index >= count:
            fail("Unable to unserialize mesh, shape index is out of range! (requested %i out of 0..%i)" % (shape_index, count - 1))
        if version == 4:
            offset = struct.unpack_from("<Q", buf, len(buf) - 8 * (count - shape_index) - 4)[0]
        else:
            offset = struct.unpack_from("<I", buf, len(buf) - 4 * (count - shape_index + 1))[0]
    data = zlib.decompressobj().decompress(buf[offset + 4:])
    pos = 0
    flags = struct.unpack_from("<I", data, pos)[0]; pos += 4
    if version == 4:
        end = data.index(b"\0", pos)
        pos = end + 1
    n_verts, n_faces = struct.unpack_from("<QQ", data, pos); pos += 16
    dp = bool(flags & 0x2000)
    ft, fs = ("<f8", 8) if dp else ("<f4", 4)

    def block(dim):
        nonlocal pos
        a = np.frombuffer(data, dtype=ft, count=n_verts * dim, offset=pos).astype(F32).reshape(n_verts, dim)
        pos += n_verts * dim * fs
        return a
    positions = block(3)
    normals = block(3) if flags & 0x0001 else None
    texcoords = block(2) if flags & 0x0002 else None
    if flags & 0x0008:
        block(3)                                             # vertex colours: skipped (serialized.cpp:290-291)
    it = "<u8" if n_verts > 0xFFFFFFFF else "<u4"
    faces = np.frombuffer(data, dtype=it, count=n_faces * 3, offset=pos).astype(np.uint32).reshape(n_faces, 3)
    if faces.size and faces.max() >= n_verts:
        fail("vertex index out of range")
    positions = transform_points(to_world, positions)
    if face_normals:
        normals = None
    elif normals is not None:
        normals = transform_normals(to_world, normals)
    else:
        normals = compute_vertex_normals(positions, faces)
    return dict(positions=np.ascontiguousarray(positions, F32), faces=np.ascontiguousarray(faces), normals=normals,
                texcoords=None if texcoords is None else np.ascontiguousarray(texcoords, F32))


def write_serialized(path, meshes, version=4):
    """writes the same format (used by the tests and handy for exporting procedural scenes): `meshes` = list of dicts with
    positions, faces and optionally normals / texcoords"""
    import zlib
    offsets, out = [], bytearray()
    for m in meshes:
        offsets.append(len(out))
        p = np.asarray(m["positions"], F32).reshape(-1, 3)
        f = np.asarray(m["faces"], np.uint32).reshape(-1, 3)
        flags = 0x1000 | (0x0001 if m.get("normals") is not None else 0) | (0x0002 if m.get("texcoords") is not None else 0)
        body = struct.pack("<I", flags) + (b"mesh\0" if version == 4 else b"") + struct.pack("<QQ", p.shape[0], f.shape[0]) + p.astype("<f4").tobytes()
        if m.get("normals") is not None:
            body += np.asarray(m["normals"], "<f4").tobytes()
        if m.get("texcoords") is not None:
            body += np.asarray(m["texcoords"], "<f4").tobytes()
        body += f.astype("<u4").tobytes()
        out += struct.pack("<hh", 0x041C, version) + zlib.compress(body, 6)
    for o in offsets:
        out += struct.pack("<Q" if version == 4 else "<I", o)
    out += struct.pack("<I", len(meshes))
    with open(path, "wb") as fh:
        fh.write(bytes(out))


def rectangle(to_world=None, flip_normals=False):
    """`rectangle` shape (src/shapes/rectangle.cpp:73-90: [-1,1]^2 in the xy plane, normal +z, uv = (p.xy+1)/2)
    tessellated into two triangles.  The reference intersects and samples it analytically; the tessellation has the same
    geometry and the same uniform area density, but draws its emitter samples through the mesh path, so renders are
    statistically equivalent rather than sample-identical."""
    p = transform_points(to_world, np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=F32))
    f = [[0, 2, 1], [0, 3, 2]] if flip_normals else [[0, 1, 2], [0, 2, 3]]
    return dict(positions=p, faces=np.array(f, np.uint32), normals=None, texcoords=np.array([[0, 0], [1, 0], [1, 1], [0, 1]], F32))
