"""Image-file readers/writers of the host side (Python mirror of include/fredholm/image_io.h).

The reference decodes images with stb_image (fredholm/src/scene.cpp:7-66): 8-bit RGBA with a vertical flip for material
textures, float RGBA without a flip for the IBL.  PNG (non-interlaced), binary PPM/PGM and Radiance .hdr are read here from
their published specifications (JPEG: sequential and progressive Huffman DCT).  Writers exist for tests and tools."""
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


def load_rgba8_native(path, flip_vertically=True):
    """the same image through the library's host-side decoder (fh_image_load_rgba8: include/fredholm/image_io.h compiled into
    libfredholm_hip.so) -- what the scene loaders use, since a pure-Python Huffman decoder is slow on real textures"""
    import ctypes as C
    from . import native as N
    L = N.lib()
    w, h, ptr = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    L.fh_image_load_rgba8.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8))]
    L.fh_image_free.argtypes = [C.POINTER(C.c_uint8)]
    L.fh_image_free.restype = None
    rc = L.fh_image_load_rgba8(str(path).encode(), int(bool(flip_vertically)), C.byref(w), C.byref(h), C.byref(ptr))
    if rc != 0:
        msg = L.fh_last_error(None)
        raise ValueError(msg.decode() if msg else f"failed to load {path}")
    try:
        return np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 4)).copy()
    finally:
        L.fh_image_free(ptr)


def load_texture(path, flip_vertically=True):
    """what the scene loaders call: the library's decoder when the library can be loaded, the pure-Python one otherwise (same bytes)"""
    try:
        from . import native as N
        N.lib()
    except Exception:
        return load_rgba8(path, flip_vertically)
    return load_rgba8_native(path, flip_vertically)


def load_rgba8(path, flip_vertically=True):
    """stbi_load(path, ..., STBI_rgb_alpha) with stbi_set_flip_vertically_on_load(flip) (scene.cpp:15-16); pure Python."""
    data = _read(path)
    try:
        if data[:2] == b"\x89P": img = decode_png(data)
        elif data[:2] in (b"P5", b"P6"): img = decode_pnm(data)
        elif data[:2] == b"\xff\xd8": img = decode_jpeg(data)
        else: raise ValueError(f"failed to load {path}: only PNG, JPEG (sequential or progressive Huffman) and binary PPM/PGM images are supported in this build")
    except ValueError:
        raise
    except Exception as e:  # ran off the end of a damaged file, undefined table, zlib error ...: one error type for callers
        raise ValueError(f"failed to load {path}: damaged or truncated file ({type(e).__name__}: {e})") from e
    return np.ascontiguousarray(img[::-1] if flip_vertically else img)


def load_hdr(path):
    """stbi_loadf(path, ..., STBI_rgb_alpha) without flip (scene.cpp:44-45): float32 [h, w, 4], alpha 1."""
    try:
        return _load_hdr(path)
    except ValueError:
        raise
    except Exception as e:
        raise ValueError(f"failed to load {path}: damaged or truncated file ({type(e).__name__}: {e})") from e


def _load_hdr(path):
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


# ------------------------------------------------------------------------------------------------ JPEG (ITU-T T.81): sequential and progressive Huffman DCT
_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


def _idct_block(coef):
    """8x8 integer IDCT of dequantised coefficients (natural order, int64) -> uint8 samples; the arithmetic of
    include/fredholm/image_io.h: JpegDecoder::idct, vectorised over the 8 lines of each pass."""
    F0298, F0390, F0541, F0765, F0899, F1175, F1501, F1847, F1961, F2053, F2562, F3072 = 2446, 3196, 4433, 6270, 7373, 9633, 12299, 15137, 16069, 16819, 20995, 25172
    CB, P1 = 13, 2

    def one_d(d, sh):  # d[k] = array of the k-th input over 8 lines
        z1 = (d[2] + d[6]) * F0541
        t2, t3 = z1 + d[6] * (-F1847), z1 + d[2] * F0765
        t0, t1 = (d[0] + d[4]) << CB, (d[0] - d[4]) << CB
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        a0, a1, a2, a3 = d[7], d[5], d[3], d[1]
        z1, z2, z3, z4 = a0 + a3, a1 + a2, a0 + a2, a1 + a3
        z5 = (z3 + z4) * F1175
        a0, a1, a2, a3 = a0 * F0298, a1 * F2053, a2 * F3072, a3 * F1501
        z1, z2, z3, z4 = z1 * -F0899, z2 * -F2562, z3 * -F1961 + z5, z4 * -F0390 + z5
        a0, a1, a2, a3 = a0 + z1 + z3, a1 + z2 + z4, a2 + z2 + z3, a3 + z1 + z4
        rnd = 1 << (sh - 1)
        return [(v + rnd) >> sh for v in (t10 + a3, t11 + a2, t12 + a1, t13 + a0, t13 - a0, t12 - a1, t11 - a2, t10 - a3)]
    c = np.asarray(coef, dtype=np.int64).reshape(8, 8)
    ws = np.stack(one_d([c[k, :] for k in range(8)], CB - P1), axis=0)          # pass 1: columns; ws[k, i]
    out = np.stack(one_d([ws[:, k] for k in range(8)], CB + P1 + 3), axis=1)    # pass 2: rows; out[i, k]
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def _jpeg_finish(comps, width, height, hmax, vmax, adobe):
    """component planes (uint8, MCU-padded) -> uint8 [h, w, 4]: triangle-filter upsampling and fixed-point YCbCr -> RGB, shared by the sequential and
    the progressive decoder"""
    def upsampled(c):
        fx, fy = hmax // c["h"], vmax // c["v"]
        p = c["plane"].astype(np.int32)
        if fx == 1 and fy == 1:
            return p
        H, W = p.shape

        def shifted(a, dy, dx):  # a[y + dy, x + dx] with edge replication
            ys = np.clip(np.arange(H) + dy, 0, H - 1)
            xs = np.clip(np.arange(W) + dx, 0, W - 1)
            return a[np.ix_(ys, xs)]
        if fx == 2 and fy == 1:
            out = np.zeros((H, 2 * W), np.int32)
            out[:, 0::2] = (3 * p + shifted(p, 0, -1) + 1) >> 2
            out[:, 1::2] = (3 * p + shifted(p, 0, 1) + 2) >> 2
            return out
        if fx == 1 and fy == 2:
            out = np.zeros((2 * H, W), np.int32)
            out[0::2] = (3 * p + shifted(p, -1, 0) + 1) >> 2
            out[1::2] = (3 * p + shifted(p, 1, 0) + 2) >> 2
            return out
        out = np.zeros((2 * H, 2 * W), np.int32)
        for oy, dy in ((0, -1), (1, 1)):
            near_col = 3 * p + shifted(p, dy, 0)
            for ox, dx, rnd in ((0, -1, 8), (1, 1, 7)):
                far_col = 3 * shifted(p, 0, dx) + shifted(p, dy, dx)
                out[oy::2, ox::2] = (3 * near_col + far_col + rnd) >> 4
        return out
    img = np.full((height, width, 4), 255, dtype=np.uint8)
    if len(comps) == 1:
        img[..., :3] = comps[0]["plane"][:height, :width, None]
        return img
    Y, cb, cr = (upsampled(c)[:height, :width].astype(np.int64) for c in comps)
    if adobe is None or adobe != 0:
        cb, cr = cb - 128, cr - 128
        img[..., 0] = np.clip(Y + ((91881 * cr + 32768) >> 16), 0, 255)
        img[..., 1] = np.clip(Y + ((-22554 * cb - 46802 * cr + 32768) >> 16), 0, 255)
        img[..., 2] = np.clip(Y + ((116130 * cb + 32768) >> 16), 0, 255)
    else:
        img[..., 0], img[..., 1], img[..., 2] = Y, cb, cr
    return img


def _decode_jpeg_progressive(data):
    """SOF2 files (annex G): every scan adds a band of coefficients (spectral selection) or one more bit of them (successive approximation) to a
    coefficient store that is dequantised and transformed once, after the last scan.  Same arithmetic after entropy decoding as the sequential
    path; bit-identical to include/fredholm/image_io.h: JpegDecoder (progressive part)."""
    pos = 2
    q, dc, ac = {}, {}, {}
    comps, width, height, restart, adobe = [], 0, 0, 0, None
    state = {"bits": 0, "n": 0, "pos": 0, "marker": False}

    def bit():
        if state["n"] == 0:
            b = 0
            if not state["marker"] and state["pos"] < len(data):
                b = data[state["pos"]]
                if b == 0xFF:
                    b2 = data[state["pos"] + 1] if state["pos"] + 1 < len(data) else 0xD9
                    if b2 == 0: state["pos"] += 2
                    else: state["marker"], b = True, 0
                else: state["pos"] += 1
            state["bits"], state["n"] = b, 8
        state["n"] -= 1
        return (state["bits"] >> state["n"]) & 1

    def receive(k):
        v = 0
        for _ in range(k): v = (v << 1) | bit()
        return v

    def extend(v, k):
        return 0 if k == 0 else (v - (1 << k) + 1 if v < (1 << (k - 1)) else v)

    def decode(tab):
        counts, syms = tab
        code = first = index = 0
        for ln in range(16):
            code |= bit()
            cnt = counts[ln]
            if code - cnt < first: return syms[index + (code - first)]
            index += cnt
            first = (first + cnt) << 1
            code <<= 1
        raise ValueError("jpeg: bad Huffman code")

    def scan(d):
        ns = d[0]
        if ns < 1 or ns > len(comps) or len(d) < 4 + 2 * ns: raise ValueError("jpeg: bad SOS")
        sc = []
        for i in range(ns):
            c = next((c for c in comps if c["id"] == d[1 + 2 * i]), None)
            if c is None: raise ValueError("jpeg: bad scan component")
            c["td"], c["ta"] = d[2 + 2 * i] >> 4, d[2 + 2 * i] & 15
            sc.append(c)
        Ss, Se, Ah, Al = d[1 + 2 * ns], d[2 + 2 * ns], d[3 + 2 * ns] >> 4, d[3 + 2 * ns] & 15
        if Ss > Se or Se > 63 or (Ss == 0 and Se != 0) or (Ss > 0 and ns != 1) or Al > 13 or (Ah and Ah != Al + 1): raise ValueError("jpeg: bad progressive scan parameters")
        for c in sc:
            if (Ss == 0 and Ah == 0 and c["td"] not in dc) or (Ss > 0 and c["ta"] not in ac): raise ValueError("jpeg: missing table")
        if ns > 1:
            units = [[(c, y * c["v"] + by, x * c["h"] + bx) for c in sc for by in range(c["v"]) for bx in range(c["h"])] for y in range(my) for x in range(mx)]
        else:
            c = sc[0]
            units = [[(c, y, x)] for y in range(c["nby"]) for x in range(c["nbx"])]
        state["n"], state["marker"] = 0, False
        for c in comps: c["pred"] = 0
        eobrun = 0
        p1, m1 = 1 << Al, -(1 << Al)
        for ui, unit in enumerate(units):
            if restart and ui and ui % restart == 0:
                state["n"], state["marker"] = 0, False
                while state["pos"] + 1 < len(data) and not (data[state["pos"]] == 0xFF and 0xD0 <= data[state["pos"] + 1] <= 0xD7): state["pos"] += 1
                state["pos"] += 2
                for c in comps: c["pred"] = 0
                eobrun = 0
            for c, by, bx in unit:
                blk = c["coef"][by, bx]
                if Ss == 0:
                    if Ah == 0:
                        t = decode(dc[c["td"]])
                        if t > 11: raise ValueError("jpeg: bad DC size")
                        c["pred"] += extend(receive(t), t)
                        blk[0] = c["pred"] * (1 << Al)
                    elif bit():
                        blk[0] |= p1
                    continue
                tab = ac[c["ta"]]
                if Ah == 0:
                    if eobrun:
                        eobrun -= 1
                        continue
                    k = Ss
                    while k <= Se:
                        rs = decode(tab)
                        r, sz = rs >> 4, rs & 15
                        if sz == 0:
                            if r < 15:
                                eobrun = (1 << r) - 1
                                if r: eobrun += receive(r)
                                break
                            k += 16
                            continue
                        k += r
                        if k > Se: raise ValueError("jpeg: bad AC run")
                        blk[_ZIGZAG[k]] = extend(receive(sz), sz) * (1 << Al)
                        k += 1
                    continue
                # refinement of an AC band (annex G.2.3)
                k = Ss
                if eobrun == 0:
                    while k <= Se:
                        rs = decode(tab)
                        r, sz = rs >> 4, rs & 15
                        value = 0
                        if sz:
                            if sz != 1: raise ValueError("jpeg: bad refinement code")
                            value = p1 if bit() else m1
                        elif r != 15:
                            eobrun = 1 << r
                            if r: eobrun += receive(r)
                            break
                        while k <= Se:
                            z = _ZIGZAG[k]
                            if blk[z] != 0:
                                if bit() and (blk[z] & p1) == 0: blk[z] += p1 if blk[z] >= 0 else m1
                            else:
                                r -= 1
                                if r < 0: break
                            k += 1
                        if value:
                            if k > Se: raise ValueError("jpeg: bad AC run")
                            blk[_ZIGZAG[k]] = value
                        k += 1
                if eobrun > 0:
                    while k <= Se:
                        z = _ZIGZAG[k]
                        if blk[z] != 0 and bit() and (blk[z] & p1) == 0: blk[z] += p1 if blk[z] >= 0 else m1
                        k += 1
                    eobrun -= 1

    def tables(d):
        p = 0
        while p < len(d):
            tc, th = d[p] >> 4, d[p] & 15
            counts = list(d[p + 1:p + 17])
            total = sum(counts)
            syms = d[p + 17:p + 17 + total]
            if tc > 1 or th > 3 or len(syms) != total: raise ValueError("jpeg: bad DHT")
            (ac if tc else dc)[th] = (counts, syms)
            p += 17 + total
    seen_scan = False
    while True:
        while pos + 1 < len(data) and not (data[pos] == 0xFF and data[pos + 1] not in (0x00, 0xFF)): pos += 1
        if pos + 1 >= len(data): raise ValueError("jpeg: truncated")
        marker = data[pos + 1]
        pos += 2
        if marker == 0xD9: break
        if 0xD0 <= marker <= 0xD7: continue
        n = (data[pos] << 8) | data[pos + 1]
        d = data[pos + 2:pos + n]
        if marker == 0xC2:
            if comps: raise ValueError("jpeg: more than one frame")
            if d[0] != 8: raise ValueError("jpeg: only 8-bit samples are supported")
            height, width, nc = (d[1] << 8) | d[2], (d[3] << 8) | d[4], d[5]
            if width <= 0 or height <= 0: raise ValueError("jpeg: bad dimensions")
            if nc not in (1, 3): raise ValueError("jpeg: only 1- and 3-component images are supported")
            comps = [{"id": d[6 + 3 * i], "h": d[7 + 3 * i] >> 4, "v": d[7 + 3 * i] & 15, "tq": d[8 + 3 * i], "pred": 0} for i in range(nc)]
            if any(not (1 <= c["h"] <= 2 and 1 <= c["v"] <= 2) for c in comps): raise ValueError("jpeg: unsupported sampling factors")
            if nc == 1: comps[0]["h"] = comps[0]["v"] = 1
            hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
            mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
            for c in comps:
                c["bw"], c["bh"] = mx * c["h"] * 8, my * c["v"] * 8
                c["nbx"], c["nby"] = -(-(-(-width * c["h"] // hmax)) // 8), -(-(-(-height * c["v"] // vmax)) // 8)  # blocks a non-interleaved scan covers
                c["coef"] = np.zeros((c["bh"] // 8, c["bw"] // 8, 64), dtype=np.int64)  # natural order
        elif marker == 0xC4: tables(d)
        elif marker == 0xDB:
            p = 0
            while p < len(d):
                pq, tq = d[p] >> 4, d[p] & 15
                p += 1
                tab = np.zeros(64, dtype=np.int64)
                for i in range(64):
                    tab[_ZIGZAG[i]] = ((d[p] << 8) | d[p + 1]) if pq else d[p]
                    p += pq + 1
                q[tq] = tab
        elif marker == 0xDD: restart = (d[0] << 8) | d[1]
        elif marker == 0xEE and d[:5] == b"Adobe": adobe = d[11]
        elif marker == 0xDA:
            if not comps: raise ValueError("jpeg: scan before frame")
            state["pos"] = pos + n
            scan(d)
            seen_scan = True
            pos = state["pos"]
            continue
        pos += n
    if not seen_scan: raise ValueError("jpeg: no image data")
    for c in comps:
        if c["tq"] not in q: raise ValueError("jpeg: missing table")
        c["plane"] = np.zeros((c["bh"], c["bw"]), dtype=np.uint8)
        for by in range(c["bh"] // 8):
            for bx in range(c["bw"] // 8):
                c["plane"][by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = _idct_block(c["coef"][by, bx] * q[c["tq"]])
    return _jpeg_finish(comps, width, height, hmax, vmax, adobe)


def decode_jpeg(data):
    """bytes -> uint8 [h, w, 4]; bit-identical to the C++ decoder (integer IDCT, triangle upsampling, fixed-point colour)."""
    if data[:2] != b"\xff\xd8":
        raise ValueError("jpeg: bad signature")
    pos = 2
    q, dc, ac = {}, {}, {}
    comps, width, height, restart, adobe = [], 0, 0, 0, None

    def tables(d, store):
        p = 0
        while p < len(d):
            tc, th = d[p] >> 4, d[p] & 15
            counts = list(d[p + 1:p + 17])
            total = sum(counts)
            syms = d[p + 17:p + 17 + total]
            if tc > 1 or th > 3 or len(syms) != total:
                raise ValueError("jpeg: bad DHT")
            store[tc][th] = (counts, syms)
            p += 17 + total
    while True:
        while not (data[pos] == 0xFF and data[pos + 1] not in (0x00, 0xFF)):
            pos += 1
        marker = data[pos + 1]
        pos += 2
        if marker == 0xD9:
            raise ValueError("jpeg: no image data")
        n = (data[pos] << 8) | data[pos + 1]
        d = data[pos + 2:pos + n]
        if marker == 0xDA:
            break
        if marker in (0xC0, 0xC1):
            if d[0] != 8: raise ValueError("jpeg: only 8-bit samples are supported")
            height, width, nc = (d[1] << 8) | d[2], (d[3] << 8) | d[4], d[5]
            if nc not in (1, 3): raise ValueError("jpeg: only 1- and 3-component images are supported")
            comps = [{"id": d[6 + 3 * i], "h": d[7 + 3 * i] >> 4, "v": d[7 + 3 * i] & 15, "tq": d[8 + 3 * i], "pred": 0} for i in range(nc)]
            if any(not (1 <= c["h"] <= 2 and 1 <= c["v"] <= 2) for c in comps): raise ValueError("jpeg: unsupported sampling factors")
            if nc == 1: comps[0]["h"] = comps[0]["v"] = 1
        elif marker == 0xC2: return _decode_jpeg_progressive(data)
        elif 0xC3 <= marker <= 0xCF and marker not in (0xC4, 0xC8, 0xCC): raise ValueError("jpeg: unsupported coding process")
        elif marker == 0xC4: tables(d, {0: dc, 1: ac})
        elif marker == 0xDB:
            p = 0
            while p < len(d):
                pq, tq = d[p] >> 4, d[p] & 15
                p += 1
                tab = np.zeros(64, dtype=np.int64)
                for i in range(64):
                    tab[_ZIGZAG[i]] = ((d[p] << 8) | d[p + 1]) if pq else d[p]
                    p += pq + 1
                q[tq] = tab
        elif marker == 0xDD: restart = (d[0] << 8) | d[1]
        elif marker == 0xEE and d[:5] == b"Adobe": adobe = d[11]
        pos += n
    ns = d[0]
    if ns != len(comps): raise ValueError("jpeg: non-interleaved scans are not supported")
    for i in range(ns):
        c = next(c for c in comps if c["id"] == d[1 + 2 * i])
        c["td"], c["ta"] = d[2 + 2 * i] >> 4, d[2 + 2 * i] & 15
    pos += n
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    for c in comps:
        c["bw"], c["bh"] = mx * c["h"] * 8, my * c["v"] * 8
        c["plane"] = np.zeros((c["bh"], c["bw"]), dtype=np.uint8)
    state = {"bits": 0, "n": 0, "pos": pos, "marker": False}

    def bit():
        if state["n"] == 0:
            b = 0
            if not state["marker"] and state["pos"] < len(data):
                b = data[state["pos"]]
                if b == 0xFF:
                    b2 = data[state["pos"] + 1] if state["pos"] + 1 < len(data) else 0xD9
                    if b2 == 0: state["pos"] += 2
                    else: state["marker"], b = True, 0
                else: state["pos"] += 1
            state["bits"], state["n"] = b, 8
        state["n"] -= 1
        return (state["bits"] >> state["n"]) & 1

    def receive(k):
        v = 0
        for _ in range(k): v = (v << 1) | bit()
        return v

    def extend(v, k):
        return 0 if k == 0 else (v - (1 << k) + 1 if v < (1 << (k - 1)) else v)

    def decode(tab):
        counts, syms = tab
        code = first = index = 0
        for ln in range(16):
            code |= bit()
            cnt = counts[ln]
            if code - cnt < first: return syms[index + (code - first)]
            index += cnt
            first = (first + cnt) << 1
            code <<= 1
        raise ValueError("jpeg: bad Huffman code")
    count = 0
    for y in range(my):
        for x in range(mx):
            if restart and count and count % restart == 0:
                state["n"], state["marker"] = 0, False
                while not (data[state["pos"]] == 0xFF and 0xD0 <= data[state["pos"] + 1] <= 0xD7): state["pos"] += 1
                state["pos"] += 2
                for c in comps: c["pred"] = 0
            count += 1
            for c in comps:
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        coef = np.zeros(64, dtype=np.int64)
                        t = decode(dc[c["td"]])
                        c["pred"] += extend(receive(t), t)
                        coef[0] = c["pred"] * q[c["tq"]][0]
                        k = 1
                        while k < 64:
                            rs = decode(ac[c["ta"]])
                            r, sz = rs >> 4, rs & 15
                            if sz == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                break
                            k += r
                            if k > 63: raise ValueError("jpeg: bad AC run")
                            coef[_ZIGZAG[k]] = extend(receive(sz), sz) * q[c["tq"]][_ZIGZAG[k]]
                            k += 1
                        y0, x0 = (y * c["v"] + by) * 8, (x * c["h"] + bx) * 8
                        c["plane"][y0:y0 + 8, x0:x0 + 8] = _idct_block(coef)

    return _jpeg_finish(comps, width, height, hmax, vmax, adobe)


_STD_LUMA_Q = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                        18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
_STD_CHROMA_Q = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)
# Annex K.3 typical Huffman tables: (BITS[1..16], HUFFVAL)
_STD_DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
_STD_DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
_STD_AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D],
             [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xA1, 0x08, 0x23, 0x42, 0xB1, 0xC1, 0x15, 0x52,
              0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0A, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3A, 0x43, 0x44, 0x45,
              0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x83,
              0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6,
              0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA, 0xE1, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8,
              0xE9, 0xEA, 0xF1, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA])
_STD_AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
             [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xA1, 0xB1, 0xC1, 0x09, 0x23, 0x33,
              0x52, 0xF0, 0x15, 0x62, 0x72, 0xD1, 0x0A, 0x16, 0x24, 0x34, 0xE1, 0x25, 0xF1, 0x17, 0x18, 0x19, 0x1A, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3A, 0x43, 0x44,
              0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A,
              0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4,
              0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7,
              0xE8, 0xE9, 0xEA, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA])


def write_jpeg(path, img, quality=90, subsampling=(1, 1), restart_interval=0):
    """Minimal baseline JPEG writer for tests and tools: uint8 [h, w, 3] (YCbCr 4:4:4, 4:2:2, 4:2:0 via subsampling=(h, v) of luma) or
    [h, w] (grey); Annex K quantisation tables scaled by `quality`, Annex K Huffman tables, optional restart markers."""
    img = np.asarray(img, dtype=np.uint8)
    grey = img.ndim == 2
    h, w = img.shape[:2]
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    qt = [np.clip((t * scale + 50) // 100, 1, 255).astype(np.int64) for t in (_STD_LUMA_Q, _STD_CHROMA_Q)]
    if grey:
        planes, samp = [img.astype(np.float64)], [(1, 1)]
    else:
        r, g, b = (img[..., k].astype(np.float64) for k in range(3))
        planes = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128.0, 0.5 * r - 0.418688 * g - 0.081312 * b + 128.0]
        samp = [tuple(subsampling), (1, 1), (1, 1)]
    hmax, vmax = samp[0]
    mx, my = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    padded = []
    for p, (sh, sv) in zip(planes, samp):
        fx, fy = hmax // sh, vmax // sv
        full = np.pad(p, ((0, my * 8 * vmax - h), (0, mx * 8 * hmax - w)), mode="edge")
        padded.append(full.reshape(full.shape[0] // fy, fy, full.shape[1] // fx, fx).mean(axis=(1, 3)))
    k = np.arange(8)
    C = np.sqrt(2.0 / 8) * np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16)
    C[0] /= np.sqrt(2.0)

    def codes(bits, vals):
        out, code, i = {}, 0, 0
        for ln in range(1, 17):
            for _ in range(bits[ln - 1]):
                out[vals[i]] = (code, ln)
                code += 1
                i += 1
            code <<= 1
        return out
    hd = [codes(*_STD_DC_L), codes(*_STD_DC_C)]
    ha = [codes(*_STD_AC_L), codes(*_STD_AC_C)]
    stream, acc, nacc = bytearray(), 0, 0

    def put(code, ln):
        nonlocal acc, nacc
        acc, nacc = (acc << ln) | code, nacc + ln
        while nacc >= 8:
            byte = (acc >> (nacc - 8)) & 0xFF
            stream.append(byte)
            if byte == 0xFF: stream.append(0)
            nacc -= 8
        acc &= (1 << nacc) - 1

    def flush():
        nonlocal acc, nacc
        if nacc: put((1 << (8 - nacc)) - 1, 8 - nacc)
        acc = nacc = 0

    def size_bits(v):
        a = abs(int(v))
        n = a.bit_length()
        return n, (v if v >= 0 else v + (1 << n) - 1) & ((1 << n) - 1)
    pred = [0] * len(planes)
    count = 0
    for y in range(my):
        for x in range(mx):
            if restart_interval and count and count % restart_interval == 0:
                flush()
                stream.extend([0xFF, 0xD0 + ((count // restart_interval - 1) & 7)])
                pred = [0] * len(planes)
            count += 1
            for ci, (p, (sh, sv)) in enumerate(zip(padded, samp)):
                t = 0 if ci == 0 else 1
                for by in range(sv):
                    for bx in range(sh):
                        blk = p[(y * sv + by) * 8:(y * sv + by) * 8 + 8, (x * sh + bx) * 8:(x * sh + bx) * 8 + 8] - 128.0
                        coef = np.rint((C @ blk @ C.T).reshape(64) / qt[t]).astype(np.int64)[_ZIGZAG]
                        n, bits = size_bits(coef[0] - pred[ci])
                        pred[ci] = int(coef[0])
                        put(*hd[t][n])
                        if n: put(bits, n)
                        run = 0
                        last = max([i for i in range(1, 64) if coef[i]] or [0])
                        for i in range(1, last + 1):
                            if coef[i] == 0:
                                run += 1
                                continue
                            while run > 15:
                                put(*ha[t][0xF0])
                                run -= 16
                            n, bits = size_bits(coef[i])
                            put(*ha[t][(run << 4) | n])
                            put(bits, n)
                            run = 0
                        if last < 63: put(*ha[t][0x00])
    flush()

    def seg(marker, body):
        return bytes([0xFF, marker]) + struct.pack(">H", len(body) + 2) + bytes(body)
    out = bytearray(b"\xff\xd8" + seg(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"))
    for t in range(1 if grey else 2):
        out += seg(0xDB, bytes([t]) + bytes(int(v) for v in qt[t][_ZIGZAG]))
    sof = struct.pack(">BHHB", 8, h, w, len(planes))
    for ci, (sh, sv) in enumerate(samp):
        sof += bytes([ci + 1, (sh << 4) | sv, 0 if ci == 0 else 1])
    out += seg(0xC0, sof)
    for cls, tid, (bits, vals) in ((0, 0, _STD_DC_L), (1, 0, _STD_AC_L)) + (() if grey else ((0, 1, _STD_DC_C), (1, 1, _STD_AC_C))):
        out += seg(0xC4, bytes([(cls << 4) | tid]) + bytes(bits) + bytes(vals))
    if restart_interval:
        out += seg(0xDD, struct.pack(">H", restart_interval))
    sos = bytes([len(planes)])
    for ci in range(len(planes)):
        sos += bytes([ci + 1, 0x00 if ci == 0 else 0x11])
    out += seg(0xDA, sos + b"\0\x3f\0") + bytes(stream) + b"\xff\xd9"
    open(path, "wb").write(bytes(out))


def write_jpeg_progressive(path, img, quality=90, subsampling=(1, 1), restart_interval=0, script=None):
    """Progressive-DCT JPEG writer (SOF2, ITU-T T.81 annex G) for tests: spectral selection and successive approximation with end-of-band
    runs.  Same colour transform, quantisation and sampling as write_jpeg.  `script` = list of scans (component indices, Ss, Se, Ah, Al);
    default: the usual 10-scan progression for colour (DC at 1 bit less, luma AC in two bands at 2 bits less, chroma AC, then the refinements)
    or its 6-scan grey counterpart.  The DC tables are annex K's; the AC table is a flat one that holds every symbol, EOBn included."""
    img = np.asarray(img, dtype=np.uint8)
    grey = img.ndim == 2
    h, w = img.shape[:2]
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    qt = [np.clip((t * scale + 50) // 100, 1, 255).astype(np.int64) for t in (_STD_LUMA_Q, _STD_CHROMA_Q)]
    if grey:
        planes, samp = [img.astype(np.float64)], [(1, 1)]
    else:
        r, g, b = (img[..., k].astype(np.float64) for k in range(3))
        planes = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128.0, 0.5 * r - 0.418688 * g - 0.081312 * b + 128.0]
        samp = [tuple(subsampling), (1, 1), (1, 1)]
    hmax, vmax = samp[0]
    mx, my = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    k = np.arange(8)
    Cm = np.sqrt(2.0 / 8) * np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16)
    Cm[0] /= np.sqrt(2.0)
    coefs, dims = [], []   # per component: int64 [by, bx, 64] (zigzag order) over the MCU-padded block grid; (blocks_x, blocks_y) that a non-interleaved scan covers
    for ci, (p, (sh, sv)) in enumerate(zip(planes, samp)):
        fx, fy = hmax // sh, vmax // sv
        full = np.pad(p, ((0, my * 8 * vmax - h), (0, mx * 8 * hmax - w)), mode="edge")
        sub = full.reshape(full.shape[0] // fy, fy, full.shape[1] // fx, fx).mean(axis=(1, 3)) - 128.0
        nby, nbx = sub.shape[0] // 8, sub.shape[1] // 8
        blocks = sub.reshape(nby, 8, nbx, 8).transpose(0, 2, 1, 3)
        dct = np.einsum("ij,yxjk,lk->yxil", Cm, blocks, Cm).reshape(nby, nbx, 64)
        coefs.append(np.rint(dct / qt[0 if ci == 0 else 1]).astype(np.int64)[..., _ZIGZAG])
        cw, ch = -(-w * sh // hmax), -(-h * sv // vmax)
        dims.append((-(-cw // 8), -(-ch // 8)))
    nc = len(planes)
    if script is None:
        script = ([(list(range(nc)), 0, 0, 0, 1), ([0], 1, 5, 0, 2)] + ([([2], 1, 63, 0, 1), ([1], 1, 63, 0, 1)] if nc == 3 else []) + [([0], 6, 63, 0, 2), ([0], 1, 63, 2, 1),
                  (list(range(nc)), 0, 0, 1, 0)] + ([([2], 1, 63, 1, 0), ([1], 1, 63, 1, 0)] if nc == 3 else []) + [([0], 1, 63, 1, 0)])

    def codes(bits, vals):
        out, code, i = {}, 0, 0
        for ln in range(1, 17):
            for _ in range(bits[ln - 1]):
                out[vals[i]] = (code, ln)
                code += 1
                i += 1
            code <<= 1
        return out
    flat_ac = ([0] * 7 + [128, 128] + [0] * 7, list(range(256)))  # every symbol (EOB0..EOB14, ZRL, all run/size pairs): 128 codes of 8 bits, 128 of 9
    hd = [codes(*_STD_DC_L), codes(*_STD_DC_C)]
    ha = codes(*flat_ac)

    def seg(marker, body):
        return bytes([0xFF, marker]) + struct.pack(">H", len(body) + 2) + bytes(body)
    out = bytearray(b"\xff\xd8" + seg(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"))
    for t in range(1 if grey else 2):
        out += seg(0xDB, bytes([t]) + bytes(int(v) for v in qt[t][_ZIGZAG]))
    sof = struct.pack(">BHHB", 8, h, w, nc)
    for ci, (sh, sv) in enumerate(samp):
        sof += bytes([ci + 1, (sh << 4) | sv, 0 if ci == 0 else 1])
    out += seg(0xC2, sof)
    for cls, tid, (bits, vals) in ((0, 0, _STD_DC_L), (1, 0, flat_ac)) + (() if grey else ((0, 1, _STD_DC_C),)):
        out += seg(0xC4, bytes([(cls << 4) | tid]) + bytes(bits) + bytes(vals))
    if restart_interval:
        out += seg(0xDD, struct.pack(">H", restart_interval))

    for comps_in_scan, Ss, Se, Ah, Al in script:
        stream, state = bytearray(), {"acc": 0, "n": 0}

        def put(code, ln):
            if ln == 0: return
            state["acc"], state["n"] = (state["acc"] << ln) | (code & ((1 << ln) - 1)), state["n"] + ln
            while state["n"] >= 8:
                byte = (state["acc"] >> (state["n"] - 8)) & 0xFF
                stream.append(byte)
                if byte == 0xFF: stream.append(0)
                state["n"] -= 8
            state["acc"] &= (1 << state["n"]) - 1

        def flush():
            if state["n"]: put((1 << (8 - state["n"])) - 1, 8 - state["n"])
            state["acc"] = state["n"] = 0

        def size_bits(v):
            a = abs(int(v))
            n = a.bit_length()
            return n, (v if v >= 0 else v + (1 << n) - 1) & ((1 << n) - 1)
        # the blocks of the scan, in coding order, with restart boundaries counted in MCUs
        if len(comps_in_scan) > 1:
            units = [[(ci, y * samp[ci][1] + by, x * samp[ci][0] + bx) for ci in comps_in_scan for by in range(samp[ci][1]) for bx in range(samp[ci][0])] for y in range(my) for x in range(mx)]
        else:
            ci = comps_in_scan[0]
            units = [[(ci, y, x)] for y in range(dims[ci][1]) for x in range(dims[ci][0])]
        pred = [0] * nc
        eobrun, pending = 0, []   # pending: correction bits that follow the EOB run / the next symbol (refinement scans)

        def emit_eobrun():
            nonlocal eobrun, pending
            if eobrun:
                nb = eobrun.bit_length() - 1
                put(*ha[nb << 4])
                if nb: put(eobrun & ((1 << nb) - 1), nb)
                eobrun = 0
            for bit_ in pending: put(bit_, 1)
            pending = []
        for ui, unit in enumerate(units):
            if restart_interval and ui and ui % restart_interval == 0:
                emit_eobrun()
                flush()
                stream.extend([0xFF, 0xD0 + ((ui // restart_interval - 1) & 7)])
                pred = [0] * nc
            for ci, by, bx in unit:
                c = coefs[ci][by, bx]
                if Ss == 0:  # DC
                    if Ah == 0:
                        v = int(c[0]) >> Al  # arithmetic shift: the point transform of annex G.1.2.1
                        n, bits = size_bits(v - pred[ci])
                        pred[ci] = v
                        put(*hd[0 if ci == 0 else 1][n])
                        put(bits, n)
                    else:
                        put((int(c[0]) >> Al) & 1, 1)
                    continue
                if Ah == 0:  # AC first pass: magnitudes divided by 2^Al towards zero
                    vals = [(abs(int(v)) >> Al) * (1 if v >= 0 else -1) for v in c[Ss:Se + 1]]
                    run = 0
                    last = max([i for i, v in enumerate(vals) if v] or [-1])
                    for i in range(last + 1):
                        if vals[i] == 0:
                            run += 1
                            continue
                        emit_eobrun()
                        while run > 15:
                            put(*ha[0xF0]); run -= 16
                        n, bits = size_bits(vals[i])
                        put(*ha[(run << 4) | n]); put(bits, n)
                        run = 0
                    if last < Se - Ss:
                        eobrun += 1
                        if eobrun == 0x7FFF: emit_eobrun()
                    continue
                # AC refinement (annex G.1.2.3): newly non-zero coefficients are coded with their sign, already non-zero ones get one correction bit
                absv = [abs(int(v)) >> Al for v in c[Ss:Se + 1]]
                eob = max([i for i, a in enumerate(absv) if a == 1] or [-1])  # last newly non-zero coefficient
                run, buffered = 0, []
                for i, a in enumerate(absv):
                    if a == 0:
                        run += 1
                        continue
                    while run > 15 and i <= eob:
                        emit_eobrun()
                        put(*ha[0xF0]); run -= 16
                        for bit_ in buffered: put(bit_, 1)
                        buffered = []
                    if a > 1:
                        buffered.append(a & 1)
                        continue
                    emit_eobrun()
                    put(*ha[(run << 4) | 1])
                    put(0 if c[Ss + i] < 0 else 1, 1)
                    for bit_ in buffered: put(bit_, 1)
                    buffered, run = [], 0
                if run > 0 or buffered:
                    eobrun += 1
                    pending += buffered
                    if eobrun == 0x7FFF or len(pending) > 900: emit_eobrun()
        emit_eobrun()
        flush()
        sos = bytes([len(comps_in_scan)])
        for ci in comps_in_scan:
            sos += bytes([ci + 1, ((0 if ci == 0 else 1) << 4) | 0])
        out += seg(0xDA, sos + bytes([Ss, Se, (Ah << 4) | Al])) + bytes(stream)
    out += b"\xff\xd9"
    open(path, "wb").write(bytes(out))
