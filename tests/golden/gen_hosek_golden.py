#!/usr/bin/env python3
"""Generate tests/golden/hosek_reference_states.json by RUNNING the reference's own Hosek-Wilkie cook
(/root/reference/fredholm/include/fredholm/arhosek.h: arhosek_rgb_skymodelstate_alloc_init), built from its sources by
oracle/Makefile into oracle/_ref/libref_hosek.so.  Needs /root/reference; the fixture it writes is data (inputs and the
reference's outputs as float32 bit patterns) and travels with the repository."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

assert O.ref_hosek() is not None, "oracle/_ref/libref_hosek.so is missing: run `make -C oracle` where /root/reference exists"
cases = []
rng = np.random.default_rng(20261003)
grid = [(t, a, e) for t in (1.0, 2.0, 3.0, 4.5, 7.25, 10.0) for a in (0.0, 0.3, 1.0) for e in (0.02, 0.4, 1.2, float(np.float32(np.pi / 2)))]
grid += [(float(np.float32(rng.uniform(1, 10))), float(np.float32(rng.uniform(0, 1))), float(np.float32(rng.uniform(0.001, np.pi / 2)))) for _ in range(40)]
for t, a, e in grid:
    cfg, rad = O.ref_hosek_state(t, a, e)
    cases.append({"turbidity": t, "albedo": a, "elevation": e, "configs_bits": [int(x) for x in cfg.reshape(-1).view(np.uint32)], "radiances_bits": [int(x) for x in rad.view(np.uint32)]})
out = os.path.join(ROOT, "tests", "golden", "hosek_reference_states.json")
json.dump({"source": "reference arhosek.h cook run through oracle/_ref/libref_hosek.so (tests/golden/gen_hosek_golden.py)", "cases": cases}, open(out, "w"), indent=0)
print(len(cases), "cases ->", out)
