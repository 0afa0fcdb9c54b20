"""Image file input/output: the formats the film writes and the bitmap texture reads --
OpenEXR (scanline; none / ZIPS / ZIP compression; half, float, uint32 channels), PFM, Radiance RGBE, 8-bit PNG and .npy.
This is the host-side counterpart of ``Bitmap::write`` / ``Bitmap::read`` (``src/libcore/bitmap.cpp``) for those
formats; it is not on the hot path and runs on the CPU with numpy + zlib only.
"""
import struct
import zlib

import numpy as np


# -------------------------------------------------------------------------------------------- sRGB transfer
def srgb_to_linear(x):
    """enoki::srgb_to_linear as used by Bitmap::convert (bitmap.cpp:870-905)."""
    x = np.asarray(x, dtype=np.float32)
    return np.where(x <= 0.04045, x * np.float32(1 / 12.92), np.power((x + np.float32(0.055)) * np.float32(1 / 1.055), np.float32(2.4))).astype(np.float32)


def linear_to_srgb(x):
    x = np.asarray(x, dtype=np.float32)
    return np.where(x <= 0.0031308, x * np.float32(12.92), np.float32(1.055) * np.power(np.maximum(x, 0), np.float32(1 / 2.4)) - np.float32(0.055)).astype(np.float32)


# -------------------------------------------------------------------------------------------- PFM
def write_pfm(path, img):
    """Bitmap::write_pfm (bitmap.cpp:2308-2346): little-endian float32, rows bottom-up, 1 or 3 channels."""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 2:
        img = img[:, :, None]
    if img.shape[2] not in (1, 3):
        raise RuntimeError("write_pfm(): Unsupported number of channels! (must be 1 or 3)")
    with open(path, "wb") as fh:
        fh.write(b"P%s\n%d %d\n-1.000000\n" % (b"F" if img.shape[2] == 3 else b"f", img.shape[1], img.shape[0]))
        fh.write(np.ascontiguousarray(img[::-1]).astype("<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as fh:
        magic = fh.readline().strip()
        if magic not in (b"PF", b"Pf"):
            raise RuntimeError("read_pfm(): Invalid header!")
        dims = fh.readline().split()
        while len(dims) < 2:
            dims += fh.readline().split()
        w, h = int(dims[0]), int(dims[1])
        scale = float(fh.readline().strip())
        c = 3 if magic == b"PF" else 1
        data = np.frombuffer(fh.read(w * h * c * 4), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, c)
    return np.ascontiguousarray(data[::-1]).astype(np.float32)


# -------------------------------------------------------------------------------------------- RGBE
def write_rgbe(path, img):
    """Bitmap::write_rgbe (bitmap.cpp:2190-2306), flat (non run-length) scanlines."""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise RuntimeError("write_rgbe(): pixel format must be PixelFormat::RGB!")
    m = np.max(img, axis=2)
    mant, expo = np.frexp(m)
    scale = np.where(m > 1e-32, mant * 256.0 / np.where(m > 1e-32, m, 1), 0).astype(np.float32)
    out = np.zeros(img.shape[:2] + (4,), np.uint8)
    out[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, expo + 128, 0).astype(np.uint8)
    with open(path, "wb") as fh:
        fh.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (img.shape[0], img.shape[1]))
        fh.write(out.tobytes())


def read_rgbe(path):
    with open(path, "rb") as fh:
        if not fh.readline().startswith(b"#?"):
            raise RuntimeError("read_rgbe(): invalid header!")
        while True:
            line = fh.readline()
            if not line:
                raise RuntimeError("read_rgbe(): unexpected end of header")
            if line.strip() == b"":
                break
        tok = fh.readline().split()
        if len(tok) != 4 or tok[0] != b"-Y" or tok[2] != b"+X":
            raise RuntimeError("read_rgbe(): unsupported orientation")
        h, w = int(tok[1]), int(tok[3])
        raw = fh.read()
    out = np.zeros((h, w, 4), np.uint8)
    pos = 0
    for y in range(h):
        if w >= 8 and w < 32768 and raw[pos] == 2 and raw[pos + 1] == 2 and not (raw[pos + 2] & 0x80):       # new-style RLE
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = raw[pos]; pos += 1
                    if n > 128:
                        n -= 128
                        out[y, x:x + n, c] = raw[pos]; pos += 1
                    else:
                        out[y, x:x + n, c] = np.frombuffer(raw, np.uint8, n, pos); pos += n
                    x += n
        else:
            out[y] = np.frombuffer(raw, np.uint8, w * 4, pos).reshape(w, 4); pos += w * 4
    e = out[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(1.0, e - (128 + 8)), 0).astype(np.float32)
    return (out[..., :3].astype(np.float32) * f[..., None]).astype(np.float32)


# -------------------------------------------------------------------------------------------- PNG (8 bit)
def write_png(path, img):
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise RuntimeError("write_png(): expected uint8 data")
    if img.ndim == 2:
        img = img[:, :, None]
    ct = {1: 0, 2: 4, 3: 2, 4: 6}[img.shape[2]]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(img.shape[0]))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", img.shape[1], img.shape[0], 8, ct, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png(path):
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:8] != b"\x89PNG\r\n\x1a\n":
        raise RuntimeError("read_png(): invalid signature")
    pos, idat, hdr, plte = 8, [], None, None
    while pos < len(buf):
        n, tag = struct.unpack(">I4s", buf[pos:pos + 8])
        data = buf[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"PLTE":
            plte = np.frombuffer(data, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"IEND":
            break
    w, h, depth, ct, _, _, interlace = hdr
    if depth != 8 or interlace != 0:
        raise RuntimeError("read_png(): only 8-bit non-interlaced files are supported")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(w * ch, np.int32)
            for i in range(w * ch):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    out = out.reshape(h, w, ch)
    if ct == 3:
        out = plte[out[..., 0]]
    return out


# -------------------------------------------------------------------------------------------- OpenEXR (scanline)
_EXR_PIXEL = {"uint32": (0, np.uint32), "float16": (1, np.float16), "float32": (2, np.float32)}


def write_exr(path, channels, compression="zip"):
    """Bitmap::write_openexr (bitmap.cpp:1650-1790): ``channels`` is an ordered dict name -> (H, W) array of dtype
    float16 / float32 / uint32.  Scanline file, ZIP (16-line blocks) or uncompressed."""
    names = sorted(channels)                          # the file stores channels in alphabetical order
    h, w = np.asarray(channels[names[0]]).shape
    arrs = {}
    for n in names:
        a = np.ascontiguousarray(channels[n])
        if a.dtype.name not in _EXR_PIXEL or a.shape != (h, w):
            raise RuntimeError("write_exr(): channel '%s' has an unsupported type or shape" % n)
        arrs[n] = a
    comp = {"none": 0, "zips": 2, "zip": 3}[compression]
    lines = {0: 1, 2: 1, 3: 16}[comp]

    def attr(name, type_, data):
        return name.encode() + b"\0" + type_.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBxxxii", _EXR_PIXEL[arrs[n].dtype.name][0], 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp])) +
              attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
              attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    blocks = []
    for y0 in range(0, h, lines):
        y1 = min(y0 + lines, h)
        raw = b"".join(arrs[n][y].astype(arrs[n].dtype.newbyteorder("<")).tobytes() for y in range(y0, y1) for n in names)
        if comp:
            b = np.frombuffer(raw, np.uint8)
            half = (len(b) + 1) // 2
            re = np.empty_like(b)
            re[:half], re[half:] = b[0::2], b[1::2]                     # de-interleave, then delta-predict
            d = re.astype(np.int32)
            d[1:] = (d[1:] - d[:-1] + 128 + 256) & 255
            z = zlib.compress(d.astype(np.uint8).tobytes(), 6)
            if len(z) < len(raw):
                raw = z
        blocks.append(struct.pack("<ii", y0, len(raw)) + raw)
    off = len(header) + 8 * len(blocks)
    table = []
    for b in blocks:
        table.append(struct.pack("<Q", off))
        off += len(b)
    with open(path, "wb") as fh:
        fh.write(header + b"".join(table) + b"".join(blocks))


def read_exr(path):
    """Scanline OpenEXR with none / ZIPS / ZIP compression -> (dict name -> (H, W) array, channel order)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    magic, version = struct.unpack("<II", buf[:8])
    if magic != 20000630:
        raise RuntimeError("read_exr(): invalid magic number")
    if version & 0x1A00:
        raise RuntimeError("read_exr(): tiled / multi-part / deep files are not supported")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        e = buf.index(b"\0", pos); name = buf[pos:e].decode(); pos = e + 1
        e = buf.index(b"\0", pos); type_ = buf[pos:e].decode(); pos = e + 1
        n = struct.unpack("<i", buf[pos:pos + 4])[0]; pos += 4
        attrs[name] = (type_, buf[pos:pos + n]); pos += n
    pos += 1
    chans, cl, p = [], attrs["channels"][1], 0
    while cl[p] != 0:
        e = cl.index(b"\0", p); nm = cl[p:e].decode(); p = e + 1
        pt, _, xs, ys = struct.unpack("<iBxxxii", cl[p:p + 16]); p += 16
        if xs != 1 or ys != 1:
            raise RuntimeError("read_exr(): subsampled channels are not supported")
        chans.append((nm, [np.uint32, np.float16, np.float32][pt]))
    comp = attrs["compression"][1][0]
    if comp not in (0, 2, 3):
        raise RuntimeError("read_exr(): unsupported compression %d (none / ZIPS / ZIP only)" % comp)
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    lines = {0: 1, 2: 1, 3: 16}[comp]
    nblocks = (h + lines - 1) // lines
    offsets = struct.unpack("<%dQ" % nblocks, buf[pos:pos + 8 * nblocks])
    out = {nm: np.zeros((h, w), dt) for nm, dt in chans}
    row_bytes = sum(np.dtype(dt).itemsize for _, dt in chans) * w
    for off in offsets:
        y, n = struct.unpack("<ii", buf[off:off + 8])
        data = buf[off + 8:off + 8 + n]
        ny = min(lines, y1 + 1 - y)
        if comp and n < row_bytes * ny:
            d = np.frombuffer(zlib.decompress(data), np.uint8).astype(np.int32)
            d[1:] -= 128
            d = np.cumsum(d) & 255
            half = (len(d) + 1) // 2
            b = np.empty(len(d), np.uint8)
            b[0::2], b[1::2] = d[:half], d[half:]
            data = b.tobytes()
        p = 0
        for yy in range(y - y0, y - y0 + ny):
            for nm, dt in chans:
                k = np.dtype(dt).itemsize * w
                out[nm][yy] = np.frombuffer(data, np.dtype(dt).newbyteorder("<"), w, p)
                p += k
    return out, [nm for nm, _ in chans]


# -------------------------------------------------------------------------------------------- generic entry points
def read(path):
    """-> float32 (H, W, C) in the file's own encoding plus a flag telling whether the values are sRGB-encoded."""
    ext = path.lower().rsplit(".", 1)[-1]
    if ext == "png":
        return read_png(path).astype(np.float32) * np.float32(1 / 255.0), True
    if ext == "pfm":
        return read_pfm(path), False
    if ext in ("hdr", "rgbe"):
        return read_rgbe(path), False
    if ext == "npy":
        a = np.load(path, allow_pickle=False).astype(np.float32)
        return (a[:, :, None] if a.ndim == 2 else a), False
    if ext == "exr":
        ch, order = read_exr(path)
        pick = [n for n in ("R", "G", "B", "A") if n in ch] or [n for n in ("Y",) if n in ch] or order
        return np.stack([ch[n].astype(np.float32) for n in pick], axis=2), False
    raise RuntimeError('Bitmap: unsupported file format "%s" (png, pfm, hdr/rgbe, exr, npy)' % ext)


def read_rgb(path, linearize=True):
    """What BitmapTexture does with its file (src/textures/bitmap.cpp:60-110): drop alpha, expand luminance to RGB,
    convert to linear float32 (unless ``raw``)."""
    img, is_srgb = read(path)
    c = img.shape[2]
    if c in (2, 4):
        img = img[:, :, :c - 1]
    if img.shape[2] == 1:
        img = np.repeat(img, 3, axis=2)
    if is_srgb and linearize:
        img = srgb_to_linear(img)
    return np.ascontiguousarray(img, dtype=np.float32)
