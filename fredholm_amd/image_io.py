"""Image-file readers/writers of the host side (Python mirror of include/fredholm/image_io.h).

The reference decodes images with stb_image (fredholm/src/scene.cpp:7-66): 8-bit RGBA with a vertical flip for material
textures, float RGBA without a flip for the IBL.  PNG (non-interlaced), binary PPM/PGM and Radiance .hdr are read here from
their published specifications; JPEG is rejected.  Writers exist for tests and tools."""
import struct
import zlib

import numpy as np


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def decode_png(data):
    """bytes -> uint8 [h, w, 4] (row 0 = top row of the file)."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("png: bad signature")
    pos, idat, plte, trns, hdr = 8, [], b"", b"", None
    while pos + 12 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n:
            raise ValueError("png: truncated chunk")
        if typ == b"IHDR": hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE": plte = body
        elif typ == b"tRNS": trns = body
        elif typ == b"IDAT": idat.append(body)
        elif typ == b"IEND": break
        pos += 12 + n
    if hdr is None:
        raise ValueError("png: missing IHDR")
    w, h, depth, ctype, comp, flt, inter = hdr
    if comp or flt:
        raise ValueError("png: unknown compression/filter method")
    if inter:
        raise ValueError("png: interlaced images are not supported")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if channels is None:
        raise ValueError("png: bad colour type")
    if not (depth in (8, 16) or (ctype in (0, 3) and depth in (1, 2, 4))) or (ctype == 3 and depth == 16):
        raise ValueError("png: bad bit depth")
    bpp = (channels * depth + 7) // 8
    stride = (w * channels * depth + 7) // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    if raw.size < (stride + 1) * h:
        raise ValueError("png: not enough image data")
    raw = raw[:(stride + 1) * h].reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft, cur = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0: rec = cur
        elif ft == 2: rec = (cur + prev) & 255
        elif ft in (1, 3, 4):
            rec = np.zeros(stride, dtype=np.int32)
            for i in range(stride):  # left-neighbour dependence: sequential per filter unit
                a = rec[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1: pr = a
                elif ft == 3: pr = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                rec[i] = (cur[i] + pr) & 255
        else:
            raise ValueError("png: bad filter type")
        out[y] = rec
        prev = rec
    # samples -> [h, w, channels] raw integers
    if depth == 8:
        smp = out.reshape(h, w, channels).astype(np.uint32)
        hi = smp
    elif depth == 16:
        b = out.reshape(h, w, channels, 2).astype(np.uint32)
        smp = (b[..., 0] << 8) | b[..., 1]
        hi = b[..., 0]
    else:
        bits = np.unpackbits(out, axis=1)[:, :w * depth].reshape(h, w, depth)
        smp = np.zeros((h, w), dtype=np.uint32)
        for k in range(depth):
            smp = (smp << 1) | bits[..., k]
        smp = smp[..., None]
        hi = smp if ctype == 3 else smp * 255 // ((1 << depth) - 1)
    img = np.zeros((h, w, 4), dtype=np.uint8)
    img[..., 3] = 255
    if ctype == 3:
        pal = np.frombuffer(plte, dtype=np.uint8).reshape(-1, 3)
        idx = smp[..., 0]
        if idx.max(initial=0) >= len(pal):
            raise ValueError("png: palette index out of range")
        img[..., :3] = pal[idx]
        al = np.full(256, 255, dtype=np.uint8)
        al[:len(trns)] = np.frombuffer(trns, dtype=np.uint8)
        img[..., 3] = al[idx]
    elif ctype in (0, 4):
        img[..., :3] = hi[..., :1]
        if ctype == 4: img[..., 3] = hi[..., 1]
        elif len(trns) >= 2:
            img[..., 3] = np.where(smp[..., 0] == struct.unpack(">H", trns[:2])[0], 0, 255)
    else:
        img[..., :3] = hi[..., :3]
        if ctype == 6: img[..., 3] = hi[..., 3]
        elif len(trns) >= 6:
            key = np.array(struct.unpack(">HHH", trns[:6]), dtype=np.uint32)
            img[..., 3] = np.where((smp[..., :3] == key).all(axis=-1), 0, 255)
    return img


def _read(path):
    try:
        return open(path, "rb").read()
    except OSError as e:
        raise ValueError(f"failed to load {path}") from e  # scene.cpp:17-19


def decode_pnm(data):
    toks, pos = [], 0
    while len(toks) < 4:
        while data[pos:pos + 1].isspace(): pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] != b"\n": pos += 1
            continue
        s = pos
        while not data[pos:pos + 1].isspace(): pos += 1
        toks.append(data[s:pos])
    if toks[0] not in (b"P5", b"P6"):
        raise ValueError("pnm: only binary P5/P6 are supported")
    w, h, maxv = int(toks[1]), int(toks[2]), int(toks[3])
    if w <= 0 or h <= 0 or not 0 < maxv <= 255:
        raise ValueError("pnm: bad header")
    ch = 3 if toks[0] == b"P6" else 1
    px = np.frombuffer(data, dtype=np.uint8, count=w * h * ch, offset=pos + 1).reshape(h, w, ch).astype(np.uint32) * 255 // maxv
    img = np.full((h, w, 4), 255, dtype=np.uint8)
    img[..., :3] = px if ch == 3 else px[..., :1]
    return img


def load_rgba8(path, flip_vertically=True):
    """stbi_load(path, ..., STBI_rgb_alpha) with stbi_set_flip_vertically_on_load(flip) (scene.cpp:15-16)."""
    data = _read(path)
    if data[:2] == b"\x89P": img = decode_png(data)
    elif data[:2] in (b"P5", b"P6"): img = decode_pnm(data)
    else: raise ValueError(f"failed to load {path}: only PNG and binary PPM/PGM images are supported in this build")
    return np.ascontiguousarray(img[::-1] if flip_vertically else img)


def load_hdr(path):
    """stbi_loadf(path, ..., STBI_rgb_alpha) without flip (scene.cpp:44-45): float32 [h, w, 4], alpha 1."""
    data = _read(path)
    lines, pos = [], 0

    def line():
        nonlocal pos
        e = data.index(b"\n", pos)
        s = data[pos:e]
        pos = e + 1
        return s
    if line() not in (b"#?RADIANCE", b"#?RGBE"):
        raise ValueError("hdr: bad signature")
    fmt = False
    while True:
        s = line()
        if not s: break
        fmt |= s == b"FORMAT=32-bit_rle_rgbe"
    if not fmt:
        raise ValueError("hdr: unsupported format")
    t = line().split()
    if len(t) != 4 or t[0] != b"-Y" or t[2] != b"+X":
        raise ValueError("hdr: unsupported data layout")
    h, w = int(t[1]), int(t[3])
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and data[pos] == 2 and data[pos + 1] == 2 and not data[pos + 2] & 0x80:
            if (data[pos + 2] << 8 | data[pos + 3]) != w:
                raise ValueError("hdr: bad scanline width")
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = data[pos]; pos += 1
                    if n > 128:
                        n -= 128
                        if n == 0 or x + n > w: raise ValueError("hdr: bad run")
                        rgbe[y, x:x + n, c] = data[pos]; pos += 1
                    else:
                        if n == 0 or x + n > w: raise ValueError("hdr: bad run")
                        rgbe[y, x:x + n, c] = np.frombuffer(data, dtype=np.uint8, count=n, offset=pos); pos += n
                    x += n
        else:
            rgbe[y] = np.frombuffer(data, dtype=np.uint8, count=4 * w, offset=pos).reshape(w, 4)
            pos += 4 * w
    scale = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    out = np.ones((h, w, 4), dtype=np.float32)
    out[..., :3] = np.where(rgbe[..., 3:4] != 0, rgbe[..., :3].astype(np.float32) * scale[..., None], np.float32(0.0))
    return out


# ------------------------------------------------------------------------------------------------ writers (tests, tools)
def write_png(path, img, filter_type=None):
    """uint8 [h, w, 4|3|1] -> 8-bit PNG.  filter_type: None = none, or 0..4 applied to every row (exercises the decoder)."""
    img = np.asarray(img, dtype=np.uint8)
    if img.ndim == 2: img = img[..., None]
    h, w, ch = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    rows = img.reshape(h, w * ch).astype(np.int32)
    ft = 0 if filter_type is None else filter_type
    raw = bytearray()
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        cur = rows[y]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        pred = [0, a, prev, (a + prev) >> 1, _paeth(a, prev, c)][ft]
        raw.append(ft)
        raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def float_to_rgbe(rgb):
    rgb = np.asarray(rgb, dtype=np.float32)
    m = rgb.max(axis=-1)
    e = np.zeros(m.shape, dtype=np.int32)
    nz = m > 1e-32
    _, ex = np.frexp(m[nz])
    e[nz] = ex
    scale = np.zeros(m.shape, dtype=np.float32)
    scale[nz] = np.ldexp(np.float32(256.0), -ex).astype(np.float32)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(nz, e + 128, 0)
    return out


def write_hdr(path, rgb, rle=False):
    """float [h, w, 3] -> Radiance .hdr (flat scanlines, or new-style RLE with literal runs only / simple runs when rle)."""
    rgbe = float_to_rgbe(np.asarray(rgb)[..., :3])
    h, w = rgbe.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
        for y in range(h):
            if rle and 8 <= w < 32768:
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for c in range(4):
                    row, x = rgbe[y, :, c], 0
                    while x < w:
                        r = 1
                        while x + r < w and r < 127 and row[x + r] == row[x]: r += 1
                        if r >= 4:
                            f.write(bytes([128 + r, int(row[x])])); x += r
                        else:
                            n = min(128, w - x)
                            f.write(bytes([n]) + row[x:x + n].tobytes()); x += n
            else:
                f.write(rgbe[y].tobytes())
