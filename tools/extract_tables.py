#!/usr/bin/env python3
"""Extract the numeric constant tables the sampler / BSDF / sky need, as raw binary data.

Runs only in the dev container (needs /root/reference). The outputs are *data*, not code:

  fredholm_amd/data/sobol_1024x52.u32   Sobol' generator matrices (Joe & Kuo direction numbers
                                        as tabulated by L. Gruenschloss), reference:
                                        fredholm/modules/sobol.cu:9-10659
  fredholm_amd/data/lut_reflection.f32  16x16x2 GGX directional-albedo LUT, lut.cu:5-93
  fredholm_amd/data/lut_sheen.f32       16x16 sheen directional-albedo LUT, lut.cu:917-955
  fredholm_amd/data/hosek_rgb.f32       Hosek-Wilkie RGB sky model coefficients
                                        (3 x 1080 config + 3 x 120 radiance floats),
                                        arhosek_rgb_data.h:105-3841.  (c) 2012-2013 Lukas Hosek and
                                        Alexander Wilkie, 3-clause BSD, see fredholm_amd/data/NOTICE.

Bit-exact sampler parity with the reference is impossible without the very same direction
numbers, so the tables are carried as binary blobs; none of the reference's code is copied.
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/fredholm"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fredholm_amd", "data")


def braces_after(text, name):
    i = text.index(name)
    a = text.index("{", i)
    b = text.index("}", a)
    return text[a + 1:b]


def strip_comments(s):
    return re.sub(r"//[^\n]*", "", s)


def main():
    os.makedirs(OUT, exist_ok=True)

    sob = open(os.path.join(REF, "modules/sobol.cu")).read()
    body = braces_after(sob, "SOBOL_MATRICES[]")
    vals = [int(t, 16) for t in re.findall(r"0x([0-9a-fA-F]+)U", body)]
    assert len(vals) == 1024 * 52, len(vals)
    np.asarray(vals, dtype=np.uint32).tofile(os.path.join(OUT, "sobol_1024x52.u32"))

    lut = open(os.path.join(REF, "modules/lut.cu")).read()
    refl = np.asarray([float(t) for t in strip_comments(braces_after(lut, "REFLECTION_LUT[]")).replace("\n", " ").split(",") if t.strip()], dtype=np.float32)
    assert refl.size == 16 * 16 * 2, refl.size
    refl.tofile(os.path.join(OUT, "lut_reflection.f32"))
    sheen = np.asarray([float(t) for t in strip_comments(braces_after(lut, "SHEEN_LUT[]")).replace("\n", " ").split(",") if t.strip()], dtype=np.float32)
    assert sheen.size == 16 * 16, sheen.size
    sheen.tofile(os.path.join(OUT, "lut_sheen.f32"))

    hos = open(os.path.join(REF, "include/fredholm/arhosek_rgb_data.h")).read()
    parts = []
    for ch in (1, 2, 3):
        a = np.asarray([float(t) for t in strip_comments(braces_after(hos, f"datasetRGB{ch}[]")).split(",") if t.strip()], dtype=np.float32)
        assert a.size == 1080, a.size
        parts.append(a)
    for ch in (1, 2, 3):
        a = np.asarray([float(t) for t in strip_comments(braces_after(hos, f"datasetRGBRad{ch}[]")).split(",") if t.strip()], dtype=np.float32)
        assert a.size == 120, a.size
        parts.append(a)
    np.concatenate(parts).tofile(os.path.join(OUT, "hosek_rgb.f32"))
    print("tables written to", os.path.normpath(OUT))


if __name__ == "__main__":
    sys.exit(main())
