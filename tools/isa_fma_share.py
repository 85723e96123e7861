#!/usr/bin/env python3
"""Static share of FMA-class instructions (v_fma / v_fmac / v_mul / v_add / v_sub _f32 and their packed forms: the class a gfx950 SIMD issues in ~2.2 cycles, everything else
in ~4.1, profiles/r04_issue_peak.json) among the VALU instructions of each hot kernel's code object -> profiles/<tag>_isa_fma_share.json, keyed by kernel name, with the
source fingerprint of the device code it was counted on.  bench.py prices executed VALU instructions (counter files) with it: roofline.valu_busy_measured.
    python tools/isa_fma_share.py r05       (compiles fredholm_amd/csrc/render.hip to assembly with the Makefile's flags; no GPU needed)"""
import collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
import isa_stats

FMA = re.compile(r"^v_(fma|fmac|mul|add|sub|subrev|mad|mac)_f32|^v_pk_(fma|mul|add)_f32|^v_fma_mix")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
asm = "/tmp/render_fma_share.s"
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-unused-result".split()
os.path.exists(asm) and os.environ.get("REUSE_ASM") or subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, "fredholm_amd", "csrc", "render.hip")])
demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
out = {"source_fingerprint": bench.source_fingerprint(), "fma_class": FMA.pattern, "note": "static counts over the code object (every instruction once, whatever the loop it sits in)", "kernels": {}}
for sym, meta in isa_stats.kernels(asm).items():
    name = demangle(sym)
    short = re.sub(r"^void fh::\(anonymous namespace\)::", "", name).split("(")[0]
    if not short.startswith(("k_trace_closest_stream", "k_trace_secondary_stream", "k_trace_merged_stream", "k_shade", "k_generate", "k_sky_pixels", "k_tail", "k_trace_closest_coop", "k_trace_secondary_coop")):
        continue
    h = collections.Counter()
    for line in isa_stats.body(asm, sym):
        m = re.match(r"\s+([a-z_0-9]+)\s", line)
        if m and not line.lstrip().startswith((".", ";")):
            h[m.group(1)] += 1
    valu = sum(v for i, v in h.items() if i.startswith("v_"))
    fma = sum(v for i, v in h.items() if FMA.match(i))
    out["kernels"][short] = {"valu": valu, "fma_class": fma, "share": round(fma / max(valu, 1), 4), "vgprs": meta.get("vgpr_count", 0), "scratch_bytes": meta.get("private_segment_fixed_size", 0)}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_isa_fma_share.json"), "w"), indent=1)
for k, v in sorted(out["kernels"].items()):
    print(f"{v['share']:.3f}  {v['valu']:6d} VALU  {v['vgprs']:3d} vgpr  scratch {v['scratch_bytes']:3d}  {k}")
