#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on MI355X (BASELINE.json metric).

Default workload = BASELINE.json configs[2], the configuration the metric and target are quoted on: 1,000,000-triangle synthetic soup,
Hosek sky, 1920x1080, 1024 spp, max_depth 8, seed 1.  `--config {1,2,3,4}` selects the other GPU configurations:
  1  Cornell box + area light, 1080p, 256 spp, Standard Surface + MIS NEE, depth 8
  3  Sponza-class textured glTF (fredholm_amd/scenes_sponza.py, written to disk and read back through the glTF loader), 1080p, 4096 spp, depth 8
  4  rtcamp8 stand-in: soup + Cornell emitters, 3840x2160, 8192 spp, depth 16, + bloom / chromatic aberration / tone map on the whole frame
One step = one presented frame: fh_render of the configuration's spp (split by the library into passes of the path pool) with every input
resident in HBM, plus the post chain for config 4, plus -- for N > 1 -- the RCCL gather of the packed beauty tiles to rank 0 and their
un-permutation into the frame.  For N > 1 the frame is sharded by interleaved 32x32 pixel tiles (one process per GPU): total work is fixed
(strong scaling).

    python bench.py --gpus 1 --steps 16 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line on rank 0: the metric, `roofline` (what binds the dominant kernel: VALU issue, priced with measured issue ceilings; the SURVEY 8(d) bytes beside it),
`cpu_baseline` / `cpu_baseline_1t` (the CPU checker on the host cores), and -- N = 1 -- `parity` (a crop of the frame against the checker), `latency` (1-spp and 16-spp calls)
and `whole_frame`; DESIGN.md 5 describes every field.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
TRI_BYTES, RAY_BYTES, HIT_BYTES = 48, 32, 16  # algorithmic bytes per ray
SHADE_BYTES_PER_HIT = 32 + 16 + 112 + 16 + 16 + 3 * 48 + 48  # ray, hit, face record, throughput, radiance in; three secondary rays + next ray out
GENERATE_BYTES_PER_PATH, ACCUMULATE_BYTES_PER_SAMPLE = 116, 84 + 88
# ---- what binds the traversal kernels: VALU issue.  Ceilings from tools/micro/issue_peak.hip (profiles/r03_issue_peak.txt: every point one launch of >= 60 ms after
# 2 s of warm-up, shader clock measured in the kernel from s_memtime / s_memrealtime).  A gfx950 SIMD issues v_fma / v_mul / v_add_f32 in ~2.2 cycles per wave64
# instruction (0.94-1.04 G/s per SIMD at the 2.05-2.38 GHz the chip holds, i.e. the guide's 2-cycle figure, MI355X_MICROARCH.md:54,473) and EVERYTHING ELSE -- v_cvt_f32_ubyte,
# v_max3 / v_min3 / v_max / v_min, v_cmp, v_cndmask, shifts, logic, integer multiply -- in ~4.1 cycles (0.56-0.58 G/s), the two kinds side by side (a 1:1 mix of v_fma_f32
# and v_max3_f32: 2.3 cycles per instruction).  The cost of a piece of code is therefore ~4.1 cycles x its non-FMA instructions, and what the kernels are made of was
# measured directly, operands in registers, eight waves per SIMD, nothing but issue in the way:
NODE_TEST_SIMD_CYCLES = 510.0  # one wave-level 8-wide node test incl. the octant permutation (fh_trace.h: node8_test): 0.0047 G tests/s per SIMD at 2.384 GHz (510.1 and 512.4 in two runs)
TRI_TEST_SIMD_CYCLES = 175.0   # one wave-level watertight triangle test (fh_trace.h: tri_test): 0.0135 G tests/s per SIMD at 2.367 GHz
NOMINAL_CLOCK_GHZ = 2.4
N_SIMDS = 1024
VALU_FMA_PEAK_PER_CYCLE, VALU_OTHER_PEAK_PER_CYCLE = 1.0 / 2.2, 1.0 / 4.1  # wave64 instructions per cycle and SIMD, by class
# share of FMA-class instructions (v_fma / v_fmac / v_mul / v_add / v_sub_f32 and their packed forms) among the VALU instructions of the kernels that have no step counters of
# their own, counted over the code object (tools/isa_stats.py --hist): with the two classes issuing side by side an instruction of such a mix costs max(share x 2.2, (1 - share) x 4.1) cycles
STATIC_FMA_SHARE = {"k_shade": 0.39, "k_generate": 0.37}
ISSUE_MODEL_SOURCE = "tools/micro/issue_peak.hip -> profiles/r03_issue_peak.txt (operands in registers, 8 waves per SIMD, >= 60 ms per point, clock from s_memtime / s_memrealtime)"


def source_fingerprint():
    """hash of everything the device code is compiled from: ties a counter file (profiles/*_traffic_config*.json) and the issue-model constants to the code that ran"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "fredholm_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "fredholm_amd", "csrc", "*.h")) + [os.path.join(ROOT, "fredholm_amd", "csrc", "Makefile")] +
                   glob.glob(os.path.join(ROOT, "include", "fh_*.h")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def file_hash(rel):
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]


def issue_model():
    """cycles a wave-level node test / triangle test costs a SIMD when nothing but issue limits it: the newest profiles/r*_issue_peak.json (written by tools/micro/issue_peak.bin --json on the GPU box,
    which compiles fh_trace.h's own node8_test / tri_test) or, without one, the round-3 constants above.  `stale` = fh_trace.h has changed since the file was measured."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_peak.json")), reverse=True):
        try:
            j = json.load(open(f))
            return {"node": float(j["node8_test_simd_cycles"]), "tri": float(j["tri_test_simd_cycles"]), "source": os.path.relpath(f, ROOT) + " (tools/micro/issue_peak.hip, which includes fh_trace.h)",
                    "stale": j.get("fh_trace_h_sha256_16") != file_hash("fredholm_amd/csrc/fh_trace.h")}
        except Exception:
            continue
    return {"node": NODE_TEST_SIMD_CYCLES, "tri": TRI_TEST_SIMD_CYCLES, "source": ISSUE_MODEL_SOURCE, "stale": True}


def fma_share_of(kernel):
    """static share of FMA-class instructions among the VALU instructions of `kernel`'s code object: the newest profiles/r*_isa_fma_share.json (tools/isa_fma_share.py) that
    lists it; None without one.  `stale`: counted on other device sources than this library's."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_fma_share.json")), reverse=True):
        try:
            j = json.load(open(f))
            k = j["kernels"].get(kernel)
            if k:
                return {"share": float(k["share"]), "source": os.path.relpath(f, ROOT), "stale": j.get("source_fingerprint") != source_fingerprint()}
        except Exception:
            continue
    return None


def scaling_model(cfg, spp):
    """EMULATED strong scaling of the render phase: one GPU rendered each of the 8 tile shards of this configuration in turn (tools/shard_time.py -> profiles/r*_shard_times.jsonl);
    whole frame / slowest shard.  Not a measurement of eight GPUs: no gather, no second device.  The row of the nearest frame length is quoted."""
    import glob
    rows = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_shard_times.jsonl")), reverse=True):
        try:
            rows = [dict(json.loads(ln), file=os.path.relpath(f, ROOT)) for ln in open(f) if ln.startswith("{")]
        except Exception:
            continue
        rows = [x for x in rows if x.get("config") == cfg and x.get("tile") == 32]
        if rows:
            break
    if not rows:
        return None
    x = min(rows, key=lambda x: abs(x["spp"] - spp))
    return {"emulated": True, "world": x["world"], "spp": x["spp"], "whole_ms": x["whole_ms"], "slowest_shard_ms": x["max_ms"], "mean_shard_ms": x["mean_ms"], "max_over_mean": x["max_over_mean"],
            "render_speedup": x["render_speedup_whole_over_max"], "efficiency": x["efficiency"], "from": x["file"],
            **({"step_speedup": x["step_speedup"], "rank0_sink_ms": x["rank0_sink"], "sharded_step_ms": x["sharded_step_ms"]} if "step_speedup" in x else {}),
            "note": "one GPU rendered each rank's 32x32-tile shard in turn (tools/shard_time.py).  render_speedup: render phase only; step_speedup (round 6): the presented frame -- slowest shard + rank 0's "
                    "pack, the gather PRICED at one 4.15 MB shard per xGMI link, the one-launch un-permutation (fh_unpack_shards) and the post chain.  No second GPU was involved: not a measurement of scaling"}


def traversal_roofline(cnt, timed, key, steps, launches, avg_ms, avg_alone_ms, bytes_per_launch, kernel_name, where, bw, traffic):
    """bound: VALU issue.  `achieved` = SIMD issue cycles per second that went into the kernel's essential work -- its wave-level node tests and triangle tests (counted by the
    instrumented replay), each at the cycles it costs a SIMD when nothing but issue limits it -- `peak` = the issue cycles the chip has: 1024 SIMDs x the nominal 2.4 GHz.  Everything
    else the kernel spends cycles on (stack, queues, refill, waits; idle lanes are inside the wave-level counts) is below the line.  MODEL-DERIVED: the two per-test costs come from a
    microbenchmark, not from this run.  The HBM view SURVEY.md 8(d) asks for stays under `algorithmic_*`: those bytes come from L2 / Infinity Cache, not from HBM."""
    im = issue_model()
    wn, wt = cnt[f"wave_node_steps_{key}"], cnt[f"wave_tri_steps_{key}"]
    ck = "closest" if key == "closest" else "shadow"
    rays, nodes, tris = cnt[f"rays_{ck}"], cnt[f"nodes_{ck}"], cnt[f"tris_{ck}"]
    issue_cycles_per_launch = (wn * im["node"] + wt * im["tri"]) * steps / launches
    achieved = issue_cycles_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    peak = N_SIMDS * NOMINAL_CLOCK_GHZ
    alg_gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    clock = timed[f"clk_cycles_{ck}"] / timed[f"clk_ticks_{ck}"] * 0.1 if timed.get(f"clk_ticks_{ck}") else None
    bw_read, bw_copy = bw
    return {"bound": "valu_issue", "kernel": kernel_name, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "G SIMD issue cycles/s", "frac": round(achieved / peak, 5),
            "model_note": "frac is model-derived: counted wave-level tests x microbenchmarked issue cycles per test (issue_model) / launch time measured in this run",
            "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "launches": int(launches), "avg_launch_ms_alone": round(avg_alone_ms, 4),
            "frac_alone": round(issue_cycles_per_launch / (avg_alone_ms * 1e-3) / 1e9 / peak, 5) if avg_alone_ms > 0 else None,
            "clock_ghz_in_kernel": round(clock, 4) if clock else None,
            "frac_at_held_clock": round(achieved / (N_SIMDS * clock), 5) if clock else None,
            "issue_model": {"node_test_simd_cycles": im["node"], "tri_test_simd_cycles": im["tri"], "wave_node_tests_per_launch": int(wn * steps / launches),
                            "wave_tri_tests_per_launch": int(wt * steps / launches), "source": im["source"], "stale": im["stale"]},
            "lane_utilisation": {"node_tests": round(nodes / max(64 * wn, 1), 4), "triangle_tests": round(tris / max(64 * wt, 1), 4)},
            "per_ray": {"nodes": round(nodes / max(rays, 1), 2), "triangles": round(tris / max(rays, 1), 2), "bytes": round(bytes_per_launch * launches / steps / max(rays, 1), 1)},
            "algorithmic_bytes_per_launch": int(bytes_per_launch), "algorithmic_gbs": round(alg_gbs, 1), "algorithmic_gbs_over_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 5),
            "measured_hbm_gbs": {"read": round(bw_read, 1), "copy": round(bw_copy, 1)},
            "note": where + "; achieved = (wave-level node tests x %g + wave-level triangle tests x %g SIMD cycles)" % (im["node"], im["tri"]) + " per launch / launch time measured inside the timed region, where passes on the other "
                    "streams share the GPU with the launch; *_alone: the same launch with the GPU to itself (one untimed step with serial passes); algorithmic_* = SURVEY.md 8(d) bytes per ray x rays, "
                    "priced against HBM only for reference -- node and triangle arrays are served by L2 / Infinity Cache, see traffic"}


def shade_record(r, cfg, spp, timed, alone, cnt, steps):
    """the shade kernels' own record, whether or not they dominate: time alone, shaded hits per second, registers / waves per SIMD of every
    class's kernel (fh_kernel_info), and -- from the configuration's counter file when it carries a `shade` block of this pass size (tools/collect_profile6.py) -- VALU busy
    by the static-mix rule, lane utilisation, SQ_WAIT_ANY share, L2 hit rate, fabric bytes per hit"""
    rec = {"ms_per_step_alone": round(alone["shade_ms"], 3), "ms_per_step_in_flight": round(timed["shade_ms"] / max(steps, 1), 3), "launches_per_step": round(timed["n_shade_launches"] / max(steps, 1), 1),
           "shaded_hits_per_step": int(cnt["shaded_hits"]), "shaded_ghits_per_s_alone": round(cnt["shaded_hits"] / (alone["shade_ms"] * 1e-3) / 1e9, 3) if alone["shade_ms"] > 0 else None,
           "algorithmic_bytes_per_hit": SHADE_BYTES_PER_HIT, "kernels": r.shade_kernel_info()}
    for tj in counter_files(cfg):
        sh = tj.get("shade")
        if not sh:
            continue
        if pass_size_differs(tj, spp, timed["n_passes"], steps) is not False:
            rec["counters_unusable"] = {"file": tj["file"], "why": "samples per pass of the counter run differ from this run's"}
            break
        hits = max(sh.get("shaded_hits_in_counter_run") or 0, 1)
        rec.update({"valu_lane_utilisation": sh.get("valu_lane_utilisation"), "wait_any_frac_of_wave_cycles": sh.get("wait_any_frac_of_wave_cycles"), "l2_hit_rate": sh.get("tcc_hit_rate"),
                    "valu_wave_insts_per_hit": round(sh["valu_insts_total"] / hits, 1) if sh.get("valu_insts_total") and sh.get("shaded_hits_in_counter_run") else None,  # SQ_INSTS_VALU counts wave64 instructions
                    "valu_lane_insts_per_hit": round(sh["valu_insts_total"] / hits * 64.0 * sh["valu_lane_utilisation"], 0) if sh.get("valu_insts_total") and sh.get("shaded_hits_in_counter_run") and sh.get("valu_lane_utilisation") else None,
                    "fabric_bytes_per_hit": round(sh["traffic_bytes_total"] / hits, 1) if sh.get("traffic_bytes_total") and sh.get("shaded_hits_in_counter_run") else None,
                    "lds_bank_conflict_frac": sh.get("lds_bank_conflict_frac"), "counters_from": tj["file"], "counters_stale": tj.get("source_fingerprint") != source_fingerprint()})
        shares = [fma_share_of(f"k_shade<{k['compiled_for_lobes']}u, 3>") for k in rec["kernels"]]
        shares = [x["share"] for x in shares if x]
        fs = {"share": sum(shares) / len(shares)} if shares else None  # (the scene's shade kernels weigh alike here: their static mixes differ by a few per cent)
        if fs and rec.get("valu_wave_insts_per_hit") and alone["shade_ms"] > 0:
            rec["fma_class_share_static"] = round(fs["share"], 4)
            cyc_mix = max(fs["share"] / VALU_FMA_PEAK_PER_CYCLE, (1.0 - fs["share"]) / VALU_OTHER_PEAK_PER_CYCLE)
            rec["valu_busy"] = round(rec["valu_wave_insts_per_hit"] * cnt["shaded_hits"] * cyc_mix / (alone["shade_ms"] * 1e-3 * N_SIMDS * NOMINAL_CLOCK_GHZ * 1e9), 4)
        break
    return rec


def freeze_roofline_fields(roof):
    """Fixed-name fields next to `frac` (one definition that does not move again).  What each means:
      frac_valu_issue_model          counted wave-level node / triangle tests x their microbenchmarked issue cycles / launch time / (1024 SIMDs x 2.4 GHz): the model `frac` has carried since round 3
      frac_hbm_counters              (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of the kernel's own counter passes / launch time alone / 8 TB/s: what the fabric saw, against the HBM peak; null without a usable counter file
      frac_survey_8d_over_hbm_peak   SURVEY 8(d) bytes per ray x rays / launch time / 8 TB/s.  NOT a fraction of a roof that can bind: the node and triangle arrays are served by L2 and
                                     Infinity Cache, so this ratio may exceed 1 on small trees (it is exempt from the [0, 1] rule and says so in bound_verdict)
      ta_busy, valu_busy             TA_TA_BUSY / (256 CUs x cycles) and executed VALU instructions x the issue cycles of the static mix / SIMD cycles, from the counter file; null without one
      bound_verdict                  one line: which of these is the roof"""
    if roof.get("bound") == "valu_issue":
        roof["frac_valu_issue_model"] = roof.get("frac")
    roof["frac_hbm_counters"] = roof.get("frac_hbm_measured")
    roof["frac_survey_8d_over_hbm_peak"] = roof.get("algorithmic_gbs_over_hbm_peak") if roof.get("bound") != "hbm" else roof.get("frac")  # (bound "hbm": `frac` IS the algorithmic bytes over the HBM peak)
    roof["ta_busy"] = (roof.get("vl1d") or {}).get("ta_busy_frac")
    roof["valu_busy"] = roof.get("valu_busy_measured")
    s8, hc, ta, vb = roof["frac_survey_8d_over_hbm_peak"], roof["frac_hbm_counters"], roof["ta_busy"], roof["valu_busy"]
    parts = []
    if vb is not None and ta is not None:
        parts.append(f"bound by VALU issue ({vb:.2f} busy) and the vector L1's look-up rate (TA {ta:.2f} busy) together")
    elif roof.get("bound") == "valu_issue":
        parts.append(f"bound by VALU issue (model fraction {roof.get('frac')}; no counter file of this launch size for the measured busy shares)")
    elif roof.get("bound") == "hbm":
        parts.append("priced against HBM with its algorithmic bytes (DESIGN.md 4), a lower bound of what moves: the kernel waits on scattered 16-byte accesses and on VALU")
    if hc is not None:
        parts.append(f"HBM is not the roof: the counters see {hc:.2f} of the 8 TB/s peak")
    if s8 is not None:
        parts.append(f"the SURVEY 8(d) bytes per ray flow at {s8:.2f} x the HBM peak" + (", i.e. out of L2 / Infinity Cache" if (hc is not None and s8 > 2 * hc) or s8 > 0.66 else ""))
    roof["bound_verdict"] = "; ".join(parts) if parts else None
    return roof


FRAC_EXEMPT = ("frac_survey_8d_over_hbm_peak",)  # a cache-served byte rate over the HBM peak: see freeze_roofline_fields


def workload(cfg, tmpdir):
    """scene + environment + camera + frame parameters of BASELINE.json configs[cfg]"""
    import numpy as np
    from fredholm_amd import scenes
    if cfg == 1:
        sc = scenes.cornell_box()
        return dict(scene=sc, width=1920, height=1080, depth=8, spp=256, camera=scenes.CORNELL_CAMERA, bg=(0.0, 0.0, 0.0), sky=None, sun=None, dir_le=None, post=None, steps=4, warmup=1,
                    name="configs[1]: Cornell box + area light (36 triangles, 2 area lights), 1920x1080, 256 spp, Standard Surface defaults + MIS NEE, max_depth 8, seed 1")
    if cfg == 2:
        sc = scenes.triangle_soup(1_000_000)
        return dict(scene=sc, width=1920, height=1080, depth=8, spp=1024, camera=scenes.SOUP_CAMERA, bg=(0.0, 0.0, 0.0), sky=(3.0, 0.3), sun=scenes.SOUP_SUN, dir_le=None, post=None, steps=16, warmup=2,
                    name="configs[2]: 1M random-triangle soup (PCG32 seed 0x853c49e6748fea9b), Hosek sky turbidity 3 albedo 0.3, 1920x1080, max_depth 8, seed 1")
    if cfg == 3:
        from fredholm_amd import scenes_sponza as SS
        from fredholm_amd.scene import Scene
        path = os.path.join(tmpdir, "sponza_like.gltf")
        info = SS.write_sponza_gltf(path)
        S = Scene()
        S.load_model(path)  # the wire format in front of the hot path: .gltf + .bin + PNG / JPEG files -> flat arrays (scene.cpp:445-834)
        sc = S.as_dict()
        return dict(scene=sc, width=1920, height=1080, depth=8, spp=4096, camera=SS.SPONZA_CAMERA, bg=(0.0, 0.0, 0.0), sky=(3.0, 0.3), sun=SS.SPONZA_SUN, dir_le=(12.0, 11.0, 9.0), post=None, steps=2,
                    warmup=1, name=f"configs[3]: Sponza-class glTF ({info['triangles']} triangles after instancing, {info['textures']} PNG/JPEG textures, alpha cut-outs, metallic-roughness + normal "
                                   "maps, clearcoat; generated by fredholm_amd/scenes_sponza.py and loaded through the glTF reader), Hosek sky + sun, 1920x1080, 4096 spp, max_depth 8, seed 1")
    if cfg == 4:
        sc = scenes.soup_with_emitters(1_000_000)
        return dict(scene=sc, width=3840, height=2160, depth=16, spp=8192, camera=scenes.SOUP_CAMERA, bg=(0.0, 0.0, 0.0), sky=(3.0, 0.3), sun=scenes.SOUP_SUN, dir_le=None,
                    post=dict(use_bloom=True, bloom_threshold=2.0, bloom_sigma=5.0, ISO=80.0, chromatic_aberration=1.0), steps=2, warmup=1,
                    name="configs[4]: rtcamp8 stand-in (SURVEY.md 8(d) C5: asset absent -> 1M-triangle soup + Cornell emitter panel), Hosek sky, 3840x2160, 8192 spp, max_depth 16, "
                         "+ bloom(threshold 2, sigma 5) / chromatic aberration 1 / ISO 80 tone map on the whole frame (rtcamp8.cpp:57-60), seed 1")
    raise SystemExit(f"unknown --config {cfg}")


def apply_environment(x, w):
    """same calls on the HIP renderer and on the CPU checker"""
    if w["sun"] is not None:
        x.set_directional_light(w["dir_le"] or (0.0, 0.0, 0.0), w["sun"], 1.0 if w["dir_le"] else 0.0)  # sets the sun direction (renderer.h:563) ...
    if w["sky"] is not None:
        x.load_arhosek_sky(*w["sky"])


_CHECKER = {}


def checker_scene(w):
    """the CPU checker's scene of this workload (its own BVH build takes seconds on a million triangles: built once for the parity crop and the CPU baseline)"""
    import ctypes
    from oracle import pyoracle as O

    if w["name"] not in _CHECKER:
        S = O.Scene(w["scene"])
        apply_environment(S, w)
        if w["sun"] is not None and not w["dir_le"]:
            O.lib().orc_set_directional_light(S.h, 0, None, None, ctypes.c_float(0))  # ... without a directional light
        _CHECKER[w["name"]] = S
    return _CHECKER[w["name"]]


def cpu_baseline(w, seconds_target=12.0):
    """Time the CPU checker (oracle/, kind "port") on a bounded sample of the SAME workload: whole 1-spp passes of the frame, all host threads.
    Reported next to the GPU number; never the thing measured."""
    from fredholm_amd.renderer import Camera
    from oracle import pyoracle as O

    S = checker_scene(w)
    cam = Camera(**w["camera"]).params()
    threads = max(1, O.hardware_threads())
    W, H = w["width"], w["height"]
    L = S.new_layers(W, H)
    passes, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < seconds_target and passes < 64:
        S.render(cam, W, H, L, 1, w["depth"], bg=w["bg"], n_threads=threads)
        passes += 1
        dt = time.perf_counter() - t0
    return {"value": round(W * H * passes / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{passes} spp of the full {W}x{H} frame, max_depth {w['depth']}, same scene ({dt:.1f} s wall on {threads} threads)"}


def cpu_baseline_1t(seconds_target=10.0):
    """BASELINE.json configs[0] exactly as SURVEY.md 8(d)(i) specifies it -- Cornell box .obj scene, 512x512, one-sample launches (of 64), max_depth 4, diffuse-only
    materials, seed 1 -- on ONE thread of the CPU checker, bounded to ~10 s of launches."""
    from fredholm_amd import scenes
    from fredholm_amd.renderer import Camera
    from oracle import pyoracle as O

    S = O.Scene(scenes.cornell_box(diffuse_only=True))
    cam = Camera(**scenes.CORNELL_CAMERA).params()
    W = H = 512
    L = S.new_layers(W, H)
    launches, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < seconds_target and launches < 64:
        S.render(cam, W, H, L, 1, 4, bg=(0.0, 0.0, 0.0), n_threads=1)
        launches += 1
        dt = time.perf_counter() - t0
    return {"value": round(W * H * launches / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"configs[0]: Cornell box, 512x512, diffuse-only BSDF, max_depth 4, {launches} of the 64 one-sample launches ({dt:.1f} s wall on 1 thread)"}


def parity_block(r, w, cam, layers, bufs, rows, spp):
    """the second half of the metric ("per-pixel L2 vs OptiX ref"; the reference cannot run here, its CPU restatement stands in): `spp` samples of the full frame
    on the GPU from a cleared state against the same samples of the checker on a crop of rows.  Outside every timed region."""
    import numpy as np
    from oracle import pyoracle as O

    r.init_render_states()
    for t in bufs.values():
        t.zero_()
    for _ in range(spp):
        r.render(cam, w["bg"], layers, 1, w["depth"])
    r.wait_for_completion()
    y0, y1 = rows
    gpu = bufs["beauty"][y0:y1].cpu().numpy()
    S = checker_scene(w)
    L = S.new_layers(w["width"], w["height"])
    for _ in range(spp):
        S.render(cam.params(), w["width"], w["height"], L, 1, w["depth"], bg=w["bg"], n_threads=max(1, O.hardware_threads()), rows=rows)
    ref = L["beauty"][y0:y1]
    a, b = np.nan_to_num(gpu[..., :3].astype(np.float64)), np.nan_to_num(ref[..., :3].astype(np.float64))
    same = float((gpu.view(np.uint32) == ref.view(np.uint32)).all(axis=2).mean())
    return {"rows": [int(y0), int(y1)], "spp": int(spp), "pixels": int((y1 - y0) * w["width"]), "rmse": float(np.sqrt(((a - b) ** 2).mean())), "max_abs": float(np.abs(a - b).max()),
            "bit_identical_pixels": same, "crop_mean": float(b.mean()),
            "tolerance": "DESIGN.md 2: >= 99.9 % of pixels bit-identical and RMSE <= 1e-5 x image mean (the HIP path and the CPU checker share no floating-point freedom)",
            "against": "oracle/ (CPU restatement of pt.cu; the OptiX reference itself needs an NVIDIA GPU)"}


def latency_block(r, w, cam, layers, frames=(200, 100)):
    """the reference's own call pattern: Controller::render issues ONE sample per call and waits (app/controller.cpp:205-230), rtcamp8 sixteen
    (app/rtcamp8.cpp:183-189).  Wall time of fh_render + fh_sync per call at the configuration's resolution, outside the headline."""
    out = {"resolution": [w["width"], w["height"]], "max_depth": w["depth"], "note": "median / min wall ms of fh_render(n) + fh_sync over `frames` calls after 20 warm-up calls"}
    for spp, n in zip((1, 16), frames):
        for _ in range(20):
            r.render(cam, w["bg"], layers, spp, w["depth"])
            r.wait_for_completion()
        ts, sub = [], []
        for _ in range(n):
            t0 = time.perf_counter()
            r.render(cam, w["bg"], layers, spp, w["depth"])
            t1 = time.perf_counter()
            r.wait_for_completion()
            ts.append((time.perf_counter() - t0) * 1e3)
            sub.append((t1 - t0) * 1e3)
        ts.sort()
        sub.sort()
        # submit_ms: the host's part -- fh_render returns when every launch of the call is queued; what remains of median_ms is the GPU working through the chain
        out[f"spp{spp}"] = {"median_ms": round(ts[len(ts) // 2], 4), "min_ms": round(ts[0], 4), "submit_ms": round(sub[len(sub) // 2], 4), "frames": n,
                            "msamples_per_s": round(w["width"] * w["height"] * spp / ts[len(ts) // 2] / 1e3, 1)}
    return out


def pass_size_differs(pmc_k, spp, n_passes, steps):
    """the counters of a launch belong to one pass size: samples per pass as the library SUBMITTED them (a call that splits off its sky pixels cuts itself into other passes
    than the nominal pool size says) in the counter run against this run; None when the file does not say"""
    then = pmc_k.get("submitted_spp_per_pass")
    now = spp / max(n_passes / max(steps, 1), 1e-9)
    if not then:
        return None
    return {"then": then, "now": round(now, 2)} if abs(then - now) > 0.02 * now else False


def counter_files(cfg):
    """committed counter summaries of this configuration, newest first (separate rocprofv3 --pmc passes, tools/profile_round3.sh: counters cannot be read from inside this process)"""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            tj = json.load(open(f))
        except Exception:
            continue
        if tj.get("config", 2) != cfg:
            continue
        tj["file"] = os.path.relpath(f, ROOT)
        out.append(tj)
    return out


def usable_counters(cfg, kernel, spp, n_passes, steps):
    """Per-launch counters describe launches of ONE pass size.  A counter file is USED only when it is about `kernel` and the pass its counter run submitted
    (`submitted_spp_per_pass`) is within 2 % of the pass this run submitted (pass_size_differs(...) is False).  Anything else -- another pass size, a file that does not say
    what it submitted -- is refused: the line then carries NO counter-derived field (traffic, frac_hbm_measured, valu, vl1d, the shade / generate issue fraction) and names what
    it refused under `counters_unusable`.  Returns (counters or None, counters_unusable or None)."""
    refused = None
    for tj in counter_files(cfg):
        if not tj.get("kernel", "").startswith(kernel):
            continue
        differs = pass_size_differs(tj, spp, n_passes, steps)
        if differs is False:
            return tj, None
        if refused is None:
            now = round(spp / max(n_passes / max(steps, 1), 1e-9), 2)
            refused = {"file": tj["file"], "then": differs["then"] if differs else None, "now": now,
                       "why": "samples per pass of the counter run differ from this run's by more than 2 %" if differs else "the file does not say which pass size its counter run submitted"}
    return None, refused


def refuse_bad_fracs(obj, path=""):
    """A fraction of a roof is in [0, 1].  Any `frac*` field outside (a model priced against the wrong launch, a counter file of another launch size) is taken out of its place in
    the line and listed under `fractions_refused` at the top level with its value -- and the run FAILS: main() prints the line (so the record exists) and exits with code 3
    (round 6; before, the run went on).  Returns the list of (path, value) removed."""
    bad = []
    if isinstance(obj, dict):
        for k in list(obj.keys()):
            v = obj[k]
            if isinstance(v, (dict, list)):
                bad += refuse_bad_fracs(v, f"{path}.{k}" if path else k)
            elif k.startswith("frac") and k not in FRAC_EXEMPT and isinstance(v, (int, float)) and not isinstance(v, bool) and not (0.0 <= v <= 1.0):
                bad.append((f"{path}.{k}" if path else k, v))
                del obj[k]
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            bad += refuse_bad_fracs(v, f"{path}[{i}]")
    return bad


# device memory of all path pools together in the default run: a third of an MI355X's HBM.  profiles/r05_pool_sweep.json has the curve: configs[2] holds 88 GB at this budget (three
# passes of the pixels that can see the scene; the 204 GB round 4 printed was the nominal figure, never allocated) and loses 1.4 % at 44 GB (six passes), 3.5 % at 29 GB, 7 % at
# 16 GB; configs[3] loses 1 % against the 218 GB it took in round 4, 3 % at 64 GB, 9 % at 16 GB.
DEFAULT_POOL_GB = 96.0


def pass_size(r, torch, local_rank, n_owned, spp, pool_spp_arg=0, pool_gb=0.0):
    """samples per pixel per pass the path pools have room for (fh_set_path_pool takes owned pixels x this).  Big passes amortise what a pass pays once (its longest rays, the
    ends of the streaming launches), at 284-436 bytes per path in flight and three pools.  The budget is a memory figure: --pool-gb, by default DEFAULT_POOL_GB, never more than
    3/4 of the device memory still free (minus 2 GiB for what RCCL and the bandwidth probe allocate later).  The library cuts a step into equal passes of what fits (of the pixels
    that can see the scene: where most pixels cannot, a pass takes more samples than this figure says) and allocates a pool for the paths its passes really start, so the memory
    held afterwards (config.path_pools.gb) is at most the budget.  profiles/r05_pool_sweep.json has throughput against budget for configs[2] and [3]."""
    slot_bytes, n_pools = r.path_pool_bytes()
    if pool_spp_arg > 0:
        return pool_spp_arg, slot_bytes, n_pools
    budget = (pool_gb if pool_gb > 0 else DEFAULT_POOL_GB) * 1e9
    try:
        budget = min(budget, 0.75 * max(torch.cuda.mem_get_info(local_rank)[0] - (2 << 30), 1 << 30))
    except Exception:  # (no memory query: the budget as given)
        pass
    return max(budget / (n_pools * slot_bytes) / max(n_owned, 1), 1.0 / max(n_owned, 1)), slot_bytes, n_pools


def general_scene_block(local_rank, tmpdir, bw, spp=540, steps=2, warmup=1, pool_gb=0.0):
    """The general-scene leg of the default run (outside the headline's timed region): BASELINE.json configs[3] -- the Sponza-class textured interior, where every camera ray
    hits and a sample is 1.7 closest-hit + 4.6 secondary rays -- for `steps` frames of `spp` samples, plus the reference's own call pattern on it (1 and 16 samples per call,
    app/controller.cpp:205-230, app/rtcamp8.cpp:183-189), the dominant traversal kernel's roofline record and a parity crop against the CPU checker.  configs[2], the headline,
    sends 84 % of its camera rays past the scene; this is the number a user of the reference's GUI sees."""
    import torch

    import fredholm_amd as F
    from fredholm_amd import native as N

    w = workload(3, tmpdir)
    W, H, D = w["width"], w["height"], w["depth"]
    r = F.Renderer(local_rank)
    r.load_scene(w["scene"])
    r.build_ias()
    apply_environment(r, w)
    r.set_resolution(W, H)
    cam = F.Camera(**w["camera"])
    dev = torch.device("cuda", local_rank)
    bufs = {n: torch.zeros((H, W) if n == "depth" else (H, W, 4), dtype=torch.float32, device=dev) for n in F.RenderLayer.NAMES}
    layers = F.RenderLayer(r, W, H, pointers={n: t.data_ptr() for n, t in bufs.items()})
    n_owned = r.owned_pixel_count()
    pool_spp, slot_bytes, n_pools = pass_size(r, torch, local_rank, n_owned, spp, 0, pool_gb)
    r.set_path_pool(max(int(n_owned * pool_spp), 1))
    for _ in range(warmup):
        r.render(cam, w["bg"], layers, spp, D)
    r.wait_for_completion()
    r.set_flags(N.FLAG_TIME_KERNELS)
    r.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(cam, w["bg"], layers, spp, D)
        r.wait_for_completion()
    dt = time.perf_counter() - t0
    timed = r.stats()
    pool_bytes_timed, pool_paths_timed = r.path_pool_allocated()
    r.set_flags(N.FLAG_TIME_KERNELS | N.FLAG_SERIAL_PASSES)
    r.reset_stats()
    r.render(cam, w["bg"], layers, spp, D)
    r.wait_for_completion()
    alone = r.stats()
    r.set_flags(N.FLAG_COUNT_TRAVERSAL)
    r.reset_stats()
    r.render(cam, w["bg"], layers, spp, D)
    r.wait_for_completion()
    cnt = r.stats()
    r.set_flags(0)
    node_bytes = timed["bvh_node_bytes"] / max(timed["bvh_nodes"], 1)
    fam = {"closest": dict(kernel="k_trace_closest_stream", ms=timed["trace_closest_ms"], launches=timed["n_closest_launches"], alone=alone["trace_closest_ms"],
                           bytes=cnt["rays_closest"] * (RAY_BYTES + HIT_BYTES) + cnt["nodes_closest"] * node_bytes + cnt["tris_closest"] * TRI_BYTES),
           "shadow": dict(kernel="k_trace_secondary_stream", ms=timed["trace_shadow_ms"], launches=timed["n_shadow_launches"], alone=alone["trace_shadow_ms"],
                          bytes=cnt["rays_shadow"] * (RAY_BYTES + HIT_BYTES) + cnt["nodes_shadow"] * node_bytes + cnt["tris_shadow"] * TRI_BYTES)}
    key = max(fam, key=lambda k: fam[k]["alone"])
    f = fam[key]
    launches = max(f["launches"], 1)
    pmc_k, unusable = usable_counters(3, f["kernel"], spp, timed["n_passes"], steps)
    roof = traversal_roofline(cnt, timed, key, steps, launches, f["ms"] / launches, f["alone"] / max(launches / steps, 1), f["bytes"] * steps / launches, f["kernel"], "whole frame", bw,
                              pmc_k.get("traffic_bytes_per_launch") if pmc_k else None)
    roof["kernel_info"] = r.kernel_info(0 if key == "closest" else 1)
    if unusable:
        roof["counters_unusable"] = unusable
    if pmc_k:
        roof["counters_from"] = pmc_k.get("file")
        roof["counters_stale"] = not (pmc_k.get("source_fingerprint") == source_fingerprint())
        if pmc_k.get("traffic_bytes_per_launch") and f["alone"] > 0:
            roof["frac_hbm_measured"] = round(pmc_k["traffic_bytes_per_launch"] / (f["alone"] / max(launches / steps, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if pmc_k.get("vl1d"):
            roof["vl1d"] = dict(pmc_k["vl1d"], note="TCP_TOTAL_CACHE_ACCESSES_sum / 256 CUs / (GRBM_GUI_ACTIVE / 8 XCDs) and TA_TA_BUSY_sum likewise, kernels serialised")
        fs = fma_share_of(pmc_k.get("kernel", ""))
        if fs and pmc_k.get("valu_insts_per_launch") and f["alone"] > 0:  # (as in the headline's record: executed VALU instructions per cycle and SIMD x the issue cycles of the kernel's static mix)
            per_cycle = pmc_k["valu_insts_per_launch"] / N_SIMDS / (f["alone"] / max(launches / steps, 1) * 1e-3) / 1e9 / (roof.get("clock_ghz_in_kernel") or NOMINAL_CLOCK_GHZ)
            cyc_mix = max(fs["share"] / VALU_FMA_PEAK_PER_CYCLE, (1.0 - fs["share"]) / VALU_OTHER_PEAK_PER_CYCLE)
            roof["valu_busy_measured"] = round(per_cycle * cyc_mix, 4)
            roof["valu_busy_formula"] = {"insts_per_cycle_per_simd": round(per_cycle, 4), "fma_class_share_static": fs["share"], "share_from": fs["source"], "share_stale": fs["stale"]}
    freeze_roofline_fields(roof)
    shade = shade_record(r, 3, spp, timed, alone, cnt, steps)
    paths = max(cnt["paths"], 1)
    out = {"workload": w["name"], "msamples_per_s": round(W * H * spp * steps / dt / 1e6, 2), "ms_per_step": round(dt / steps * 1e3, 3), "spp_per_step": spp, "spp_per_pass": round(pool_spp, 3) if isinstance(pool_spp, float) else pool_spp,
           "passes_per_step": round(timed["n_passes"] / max(steps, 1), 2), "path_pools": {"gb": round(pool_bytes_timed / 1e9, 1), "paths": int(pool_paths_timed)},
           "steps": steps, "warmup": warmup, "triangles": int(w["scene"]["indices"].shape[0]),
           "note": "BASELINE.json configs[3] at a shorter frame than its 4096 spp (throughput does not depend on the frame length beyond three passes: 512 / 4096 spp measure within 1 %); 540 = twelve passes of 45 samples, the pass the 4096-spp run submits and its counter file was collected with",
           "roofline": roof, "shade": shade,
           "kernel_ms_per_step_alone": {"trace_closest": round(alone["trace_closest_ms"], 3), "trace_secondary": round(alone["trace_shadow_ms"], 3), "shade": round(alone["shade_ms"], 3),
                                        "generate": round(alone["generate_ms"], 3), "route_and_sort": round(alone["queue_ms"], 3), "accumulate": round(alone["accumulate_ms"], 3),
                                        "tail": round(alone["tail_ms"], 3), "render_total": round(alone["render_ms"], 3)},
           "rates": {"closest_hit_grays_per_s": round(cnt["rays_closest"] / (alone["trace_closest_ms"] * 1e-3) / 1e9, 3) if alone["trace_closest_ms"] > 0 else None,
                     "secondary_grays_per_s": round(cnt["rays_shadow"] / (alone["trace_shadow_ms"] * 1e-3) / 1e9, 3) if alone["trace_shadow_ms"] > 0 else None,
                     "shaded_ghits_per_s": round(cnt["shaded_hits"] / (alone["shade_ms"] * 1e-3) / 1e9, 3) if alone["shade_ms"] > 0 else None,
                     "per_sample": {"closest_rays": round(cnt["rays_closest"] / paths, 4), "secondary_rays": round(cnt["rays_shadow"] / paths, 4), "shaded_hits": round(cnt["shaded_hits"] / paths, 4)}},
           "bvh": {"build_ms": round(timed["bvh_build_ms"], 2), "nodes": timed["bvh_nodes"], "depth": timed["bvh_depth"]}}
    out["parity"] = parity_block(r, w, cam, layers, bufs, (H // 2 - 2, H // 2 + 2), 1)
    out["latency"] = latency_block(r, w, cam, layers, frames=(100, 40))
    r.close()
    return out


def self_launch(n_gpus, argv=None):
    """start `n_gpus` ranks of this script on this node (one process per GPU, rendezvous on 127.0.0.1, a free port) and wait: returns the launcher's exit code, which is
    non-zero when any rank failed"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    print("bench.py: no WORLD_SIZE in the environment, starting the ranks myself: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4), help="BASELINE.json configs[] index (default 2: the configuration the metric is quoted on)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step; 0 = the configuration's (one fh_render call = one presented frame)")
    ap.add_argument("--pool-spp", type=int, default=0, help="samples per pixel per pass of the path pool; 0 = equal passes of at most ~128 spp of a 1080p frame and 3/4 of the free device memory, at least three per step")
    ap.add_argument("--pool-gb", type=float, default=0.0, help="device memory of all path pools together in GB (the library then cuts a step into as many equal passes as that needs); 0 = see --pool-spp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras of the N = 1 line (parity crop, small-launch latency)")
    ap.add_argument("--no-general-scene", action="store_true", help="skip the configs[3] leg the default (configs[2], N = 1) run appends as `general_scene`")
    ap.add_argument("--check-frame", action="store_true", help="N > 1: rank 0 re-renders the whole frame unsharded and compares it bit for bit with the gathered one")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts the N ranks through torch.distributed.run as CHILD processes -- it has made no GPU call (torch is
        # not even imported yet) and never replaces itself -- relays their output (rank 0 prints the JSON line) and exits with the launcher's code
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch

    import fredholm_amd as F
    from fredholm_amd import distributed as D
    from fredholm_amd import native as N
    from fredholm_amd.renderer import PostProcessParams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and args.gpus > 1:
        raise SystemExit(f"--gpus {args.gpus} under a launcher that started {world} rank(s): start it with --nproc-per-node {args.gpus}, or without a launcher (bench.py then starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    backend = os.environ.get("FH_BENCH_BACKEND", "nccl")  # "gloo": functional test of the N > 1 path on a box with fewer GPUs
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    failed_rc = 0
    tmp = tempfile.TemporaryDirectory(prefix="fh_bench_")
    w = workload(args.config, tmp.name)
    WIDTH, HEIGHT, MAX_DEPTH = w["width"], w["height"], w["depth"]
    spp = args.spp or w["spp"]
    steps = args.steps if args.steps is not None else w["steps"]
    warmup = args.warmup if args.warmup is not None else w["warmup"]

    # ---- scene, BVH, environment: everything resident in HBM before the timed region
    sc = w["scene"]
    r = F.Renderer(local_rank)
    r.load_scene(sc)
    r.build_ias()
    apply_environment(r, w)
    if w["sun"] is not None and not w["dir_le"]:
        r.clear_directional_light()  # ... without a directional light
    r.set_resolution(WIDTH, HEIGHT)
    if world > 1:
        r.set_tile_shard(rank, world, 32, 32)
    cam = F.Camera(**w["camera"])
    dev = torch.device("cuda", local_rank)
    # every device buffer of the run is allocated BEFORE the path pools are sized by what is left
    bufs = {n: torch.zeros((HEIGHT, WIDTH) if n == "depth" else (HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev) for n in F.RenderLayer.NAMES}
    layers = F.RenderLayer(r, WIDTH, HEIGHT, pointers={n: t.data_ptr() for n, t in bufs.items()})
    extras = world == 1 and not args.no_extras
    bufs2 = {n: torch.zeros((HEIGHT, WIDTH) if n == "depth" else (HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev) for n in F.RenderLayer.NAMES} if extras else None
    layers2 = F.RenderLayer(r, WIDTH, HEIGHT, pointers={n: t.data_ptr() for n, t in bufs2.items()}) if extras else None
    n_owned = r.owned_pixel_count()
    # N > 1: equal-size packed shards gathered to rank 0, which un-permutes them into the frame with the library's own tile map
    pad = D.max_owned(WIDTH, HEIGHT, world) if world > 1 else n_owned
    packed = torch.zeros((pad, 4), dtype=torch.float32, device=dev) if world > 1 else None
    gathered = [torch.zeros_like(packed) for _ in range(world)] if (world > 1 and rank == 0) else None
    frame = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
    post = PostProcessParams(**w["post"]) if w["post"] else None
    pp_bufs = [torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev) for _ in range(3)] if post else None
    if world > 1:  # wrong device binding, unequal shard shapes, a rendezvous the environment does not describe: fail here, not in step 1 (tools/rccl_gather_probe.py runs the same check alone)
        D.preflight(dist, dev, pad)
    # path-pool slots = owned pixels x samples per pass, one pool per pass in flight (pass_size above)
    pool_spp, slot_bytes, n_pools = pass_size(r, torch, local_rank, n_owned, spp, args.pool_spp, args.pool_gb)
    r.set_path_pool(max(int(n_owned * pool_spp), 1))
    bw_read, bw_copy = r.measure_bandwidth(1 << 30, 6) if rank == 0 else (0.0, 0.0)  # measured HBM roofline of this GPU
    torch.cuda.synchronize()

    # N > 1: the collective runs on torch's stream, the renderer on the library's.  The two are ordered by events, never through the host: torch's stream waits for the
    # pack, the library's stream waits for the gather (on every rank: the next pack must not overwrite `packed` under the collective), and fh_unpack_shard is one
    # asynchronous launch -- a step has no host synchronisation between fh_render and the presented frame
    lib_stream = torch.cuda.ExternalStream(r.stream(), device=dev) if world > 1 else None
    ev_packed = torch.cuda.Event() if world > 1 else None
    ev_gathered = torch.cuda.Event() if world > 1 else None

    def step():
        r.render(cam, w["bg"], layers, spp, MAX_DEPTH)
        presented = bufs["beauty"]
        if world > 1:
            r.pack_owned(bufs["beauty"].data_ptr(), 4, packed.data_ptr())
            ev_packed.record(lib_stream)
            torch.cuda.current_stream().wait_event(ev_packed)
            if backend == "nccl":
                dist.gather(packed, gathered, dst=0)
            else:                            # test path: stage through host memory (the .cpu() copy waits for torch's stream, which waits for the pack)
                host = [torch.empty(packed.shape, dtype=packed.dtype) for _ in range(world)] if rank == 0 else None
                dist.gather(packed.cpu(), host, dst=0)
                if rank == 0:
                    for k in range(world):
                        gathered[k].copy_(host[k], non_blocking=False)
            ev_gathered.record(torch.cuda.current_stream())
            lib_stream.wait_event(ev_gathered)
            if rank == 0:                    # present the assembled frame: every rank's shard back into place in ONE launch (fh_unpack_shards, the inverse of the packs)
                r.unpack_shards([g.data_ptr() for g in gathered], 4, frame.data_ptr())
                presented = frame
        if post and rank == 0:               # the post chain runs on the assembled frame (bloom has a 16-pixel halo), on the library's stream
            r.post_process(presented.data_ptr(), pp_bufs[0].data_ptr(), pp_bufs[1].data_ptr(), WIDTH, HEIGHT, post, pp_bufs[2].data_ptr())

    def fence():
        r.wait_for_completion()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    r.set_flags(N.FLAG_TIME_KERNELS)  # HIP events around the kernel launches on the library's own streams
    r.reset_stats()
    fence()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        step()
        r.wait_for_completion()          # a presented frame ends with ONE host synchronisation on the library's stream (which, N > 1, has waited for the gather): per-step times for the min / median below
        step_ms.append((time.perf_counter() - ts) * 1e3)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timed = r.stats()
    pool_bytes_timed, pool_paths_timed = r.path_pool_allocated()

    # ---- which kernel dominates: with two passes in flight a HIP-event span also contains the other stream's work (the register-heavy shade
    # kernels wait for CUs the traversal kernels hold), so the ranking comes from one more, untimed step whose passes run one after the other
    # (FH_FLAG_SERIAL_PASSES): every kernel alone on the GPU.  The roofline's launch time stays the one measured over the timed region.
    r.set_flags(N.FLAG_TIME_KERNELS | N.FLAG_SERIAL_PASSES)
    r.reset_stats()
    r.render(cam, w["bg"], layers, spp, MAX_DEPTH)
    r.wait_for_completion()
    alone = r.stats()

    # ---- algorithmic bytes of the traversal kernels: one instrumented (counting) replay of a step, untimed.
    # Deterministic sampling makes every step trace the same number of rays/nodes/triangles up to the sample index.
    r.set_flags(N.FLAG_COUNT_TRAVERSAL)
    r.reset_stats()
    r.render(cam, w["bg"], layers, spp, MAX_DEPTH)
    r.wait_for_completion()
    cnt = r.stats()
    # ... and once more with every ray started at the root (FH_FLAG_ROOT_START): what the same hits cost in tests when no ray starts at the node of its face
    r.set_flags(N.FLAG_COUNT_TRAVERSAL | N.FLAG_ROOT_START)
    r.reset_stats()
    r.render(cam, w["bg"], layers, spp, MAX_DEPTH)
    r.wait_for_completion()
    cnt_root = r.stats()
    r.set_flags(0)

    if args.check_frame and world > 1 and rank == 0:
        total_spp = spp * (warmup + steps + 3)  # + the serial step + the two counting replays, which only the local layers saw
        r2 = F.Renderer(local_rank)
        r2.load_scene(sc)
        r2.build_ias()
        apply_environment(r2, w)
        if w["sun"] is not None and not w["dir_le"]:
            r2.clear_directional_light()
        r2.set_resolution(WIDTH, HEIGHT)
        full = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
        others = {n: torch.zeros((HEIGHT, WIDTH) if n == "depth" else (HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev) for n in F.RenderLayer.NAMES}
        others["beauty"] = full
        layers2 = F.RenderLayer(r2, WIDTH, HEIGHT, pointers={n: t.data_ptr() for n, t in others.items()})
        # 1. the frame assembled from every rank's tiles after the last timed step against an unsharded render of as many samples
        r2.render(cam, w["bg"], layers2, total_spp - 3 * spp, MAX_DEPTH)
        r2.wait_for_completion()
        whole = bool((full.view(torch.int32) == frame.view(torch.int32)).all().item())
        print(f"check-frame: frame gathered from {world} ranks bit-identical to the unsharded render: {whole}", file=sys.stderr, flush=True)
        if not whole:
            raise SystemExit("gathered frame differs from the unsharded one")
        # 2. this rank's tiles after the serial step and the counting replays (three more steps that only the local layers saw)
        r2.render(cam, w["bg"], layers2, 3 * spp, MAX_DEPTH)
        r2.wait_for_completion()
        own0 = torch.from_numpy(D.tile_ownership(WIDTH, HEIGHT, 0, world).astype(np.int64)).to(dev)
        r.pack_owned(bufs["beauty"].data_ptr(), 4, packed.data_ptr())
        r.wait_for_completion()
        ok = bool((full.reshape(-1, 4).view(torch.int32)[own0] == packed[: own0.numel()].view(torch.int32)).all().item())
        print(f"check-frame: rank-0 shard bit-identical to the unsharded render: {ok}", file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit("sharded render differs from the unsharded one")
        r2.close()
    if rank == 0:
        samples = WIDTH * HEIGHT * spp * steps
        value = samples / dt / 1e6
        node_bytes = timed["bvh_node_bytes"] / max(timed["bvh_nodes"], 1)  # 64 B wide nodes (64 B for the binary fallback too)
        paths_per_step = cnt["paths"]
        small_tree = timed["bvh_nodes"] < 4096  # traced in fixed 64-ray batches (render.hip: render_submit), everything else by the streaming kernels
        # every timed kernel family: summed HIP-event ms over the timed steps, launches, algorithmic bytes per step from the counted replay, ms of one step alone
        fam = {
            "k_trace_closest_stream": dict(ms=timed["trace_closest_ms"], launches=timed["n_closest_launches"], alone=alone["trace_closest_ms"], key="closest",
                                           bytes=cnt["rays_closest"] * (RAY_BYTES + HIT_BYTES) + cnt["nodes_closest"] * node_bytes + cnt["tris_closest"] * TRI_BYTES),
            "k_trace_secondary_stream": dict(ms=timed["trace_shadow_ms"], launches=timed["n_shadow_launches"], alone=alone["trace_shadow_ms"], key="shadow",
                                             bytes=cnt["rays_shadow"] * (RAY_BYTES + HIT_BYTES) + cnt["nodes_shadow"] * node_bytes + cnt["tris_shadow"] * TRI_BYTES),
            "k_shade": dict(ms=timed["shade_ms"], launches=timed["n_shade_launches"], alone=alone["shade_ms"], bytes=cnt["shaded_hits"] * SHADE_BYTES_PER_HIT),
            "k_tail": dict(ms=timed["tail_ms"], launches=timed["n_tail_launches"], alone=alone["tail_ms"], bytes=None),
            "k_generate": dict(ms=timed["generate_ms"], launches=timed["n_generate_launches"], alone=alone["generate_ms"], bytes=paths_per_step * GENERATE_BYTES_PER_PATH),
            "k_accumulate": dict(ms=timed["accumulate_ms"], launches=timed["n_accumulate_launches"], alone=alone["accumulate_ms"], bytes=paths_per_step * ACCUMULATE_BYTES_PER_SAMPLE),
        }
        dom = max((k for k in fam if fam[k]["bytes"] is not None), key=lambda k: fam[k]["alone"])  # (the fused tail has no counted bytes of its own)
        f = fam[dom]
        launches = max(f["launches"], 1)
        bytes_per_launch = f["bytes"] * steps / launches
        avg_ms = f["ms"] / launches
        avg_alone_ms = f["alone"] / max(launches / steps, 1)
        alg_gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # the dominant kernel's own counters, from a counter run of THIS pass size or not at all (a shard's launches are not the launches the counters saw either)
        pmc_k, unusable = usable_counters(args.config, dom, spp, timed["n_passes"], steps) if world == 1 else (None, None)
        pmc = pmc_k
        traffic = pmc_k.get("traffic_bytes_per_launch") if pmc_k else None
        kernel_name = dom.replace("_stream", "_coop") if dom.startswith("k_trace") and small_tree else dom
        where = ("rank 0 shard" if world > 1 else "whole frame")
        if dom.startswith("k_trace"):
            roof = traversal_roofline(cnt, timed, f["key"], steps, launches, avg_ms, avg_alone_ms, bytes_per_launch, kernel_name, where, (bw_read, bw_copy), traffic)
        elif pmc_k and pmc_k.get("valu_insts_per_launch") and dom in STATIC_FMA_SHARE and avg_ms > 0:
            # ---- the shade / generate kernels are VALU-issue-bound as well (software transcendentals, hashing, divisions): executed VALU instructions per launch (counter
            # run of the same launch size) x the issue cycles an instruction of the kernel's mix costs, against the issue cycles the chip has
            share = STATIC_FMA_SHARE[dom]
            cyc = max(share / VALU_FMA_PEAK_PER_CYCLE, (1.0 - share) / VALU_OTHER_PEAK_PER_CYCLE)
            issue_cycles_per_launch = pmc_k["valu_insts_per_launch"] * cyc
            achieved = issue_cycles_per_launch / (avg_ms * 1e-3) / 1e9
            peak = N_SIMDS * NOMINAL_CLOCK_GHZ
            roof = {"bound": "valu_issue", "kernel": kernel_name, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "G SIMD issue cycles/s", "frac": round(achieved / peak, 5), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches": int(f["launches"]), "avg_launch_ms_alone": round(avg_alone_ms, 4),
                    "frac_alone": round(issue_cycles_per_launch / (avg_alone_ms * 1e-3) / 1e9 / peak, 5) if avg_alone_ms > 0 else None,
                    "issue_model": {"valu_insts_per_launch": pmc_k["valu_insts_per_launch"], "fma_class_share": share, "cycles_per_instruction_of_this_mix": round(cyc, 3),
                                    "source": "SQ_INSTS_VALU of the kernel (" + str(pmc_k.get("file")) + "), class rates from profiles/r03_issue_peak.txt, class share from the code object (tools/isa_stats.py)"},
                    "algorithmic_bytes_per_launch": int(bytes_per_launch), "algorithmic_gbs": round(alg_gbs, 1), "algorithmic_gbs_over_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 5),
                    "measured_hbm_gbs": {"read": round(bw_read, 1), "copy": round(bw_copy, 1)},
                    "note": where + "; achieved = executed VALU instructions per launch x issue cycles per instruction of the kernel's mix / launch time inside the timed region (passes on the other streams share the "
                            "GPU with the launch: *_alone is the same launch with the GPU to itself); algorithmic_* = DESIGN.md 4 bytes per hit, for reference"}
        else:
            roof = {"bound": "hbm", "kernel": kernel_name, "achieved": round(alg_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches": int(f["launches"]), "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms_alone": round(avg_alone_ms, 4),
                    "frac_alone": round(bytes_per_launch / (avg_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if avg_alone_ms > 0 else None,
                    "measured_hbm_gbs": {"read": round(bw_read, 1), "copy": round(bw_copy, 1)}, "algorithmic_gbs_over_measured_copy": round(alg_gbs / bw_copy, 5) if bw_copy > 0 else None,  # (a ratio, not a fraction of a roof: cache-served bytes can exceed the copy rate)
                    "note": where + "; achieved = algorithmic bytes (DESIGN.md 4) / kernel time inside the timed region; the kernel waits on scattered 16-byte accesses to path and face records and on VALU "
                            "(BSDF, software transcendentals), so the algorithmic figure is a lower bound of what moves"}
        if dom.startswith("k_trace"):
            roof["kernel_info"] = r.kernel_info(0 if f.get("key") == "closest" else 1)  # registers / LDS / scratch / workgroups per CU of the variant that ran
            # `frac` counts the tests that were DONE.  Where the library starts first-hit rays at the node of the face they leave, the same hits need fewer
            # tests than a walk from the root: the replay with FH_FLAG_ROOT_START counts those, and the same launch time priced with them is what `frac` would be if the
            # avoided tests still counted as work
            im_, key_ = issue_model(), f["key"]
            ck_ = "closest" if key_ == "closest" else "shadow"
            wn0, wt0 = cnt_root[f"wave_node_steps_{key_}"], cnt_root[f"wave_tri_steps_{key_}"]
            cyc0 = (wn0 * im_["node"] + wt0 * im_["tri"]) * steps / launches
            roof["from_root"] = {"nodes_per_ray": round(cnt_root[f"nodes_{ck_}"] / max(cnt_root[f"rays_{ck_}"], 1), 2), "triangles_per_ray": round(cnt_root[f"tris_{ck_}"] / max(cnt_root[f"rays_{ck_}"], 1), 2),
                                 "wave_node_tests_per_launch": int(wn0 * steps / launches), "wave_tri_tests_per_launch": int(wt0 * steps / launches),
                                 "share_of_those_tests_done": round((cnt[f"wave_node_steps_{key_}"] * im_["node"] + cnt[f"wave_tri_steps_{key_}"] * im_["tri"]) / max(wn0 * im_["node"] + wt0 * im_["tri"], 1.0), 4),
                                 "frac_priced_with_these_tests": round(cyc0 / (avg_ms * 1e-3) / 1e9 / (N_SIMDS * NOMINAL_CLOCK_GHZ), 5) if avg_ms > 0 else None,
                                 "note": "the counting replay with every ray started at the root (FH_FLAG_ROOT_START): the tests a walk from the root needs for the same hits; frac counts the tests done, "
                                         "so it falls when tests are avoided -- frac_priced_with_these_tests is this run's launch time priced with the root-start test counts"}
        if pmc_k and avg_alone_ms > 0:  # the counters are collected with the kernels serialised, so they are priced against the kernel's time alone
            if traffic:
                # what the fabric-side counters saw of this kernel against the HBM peak (2 x FETCH_SIZE + WRITE_SIZE of the counter run over this run's launch time alone)
                roof["frac_hbm_measured"] = roof["hbm_traffic_frac"] = round(traffic / (avg_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            if pmc_k.get("valu_insts_per_launch"):
                per_cycle = pmc_k["valu_insts_per_launch"] / N_SIMDS / (avg_alone_ms * 1e-3) / 1e9 / (roof.get("clock_ghz_in_kernel") or NOMINAL_CLOCK_GHZ)
                fs = fma_share_of(pmc_k.get("kernel", ""))
                if fs:
                    # MEASURED VALU busy share: executed VALU instructions per cycle and SIMD (SQ_INSTS_VALU of the counter run / this run's launch time alone x 1024 SIMDs x the clock held)
                    # x the issue cycles an instruction of the kernel's static mix costs (FMA-class 2.2, everything else 4.1 cycles, the two classes side by side)
                    cyc_mix = max(fs["share"] / VALU_FMA_PEAK_PER_CYCLE, (1.0 - fs["share"]) / VALU_OTHER_PEAK_PER_CYCLE)
                    roof["valu_busy_measured"] = round(per_cycle * cyc_mix, 4)
                    roof["valu_busy_formula"] = {"fma_class_share_static": fs["share"], "share_from": fs["source"], "share_stale": fs["stale"], "cycles_per_instruction_of_this_mix": round(cyc_mix, 3),
                                                 "is": "SQ_INSTS_VALU per launch / (avg_launch_ms_alone x 1024 SIMDs x clock_ghz_in_kernel) x max(share x 2.2, (1 - share) x 4.1)"}
                roof["valu"] = {"insts_per_launch": pmc_k["valu_insts_per_launch"], "insts_per_cycle_per_simd": round(per_cycle, 4),
                                "peak_per_cycle_per_simd": {"fma_mul_add_f32": round(VALU_FMA_PEAK_PER_CYCLE, 3), "everything_else": round(VALU_OTHER_PEAK_PER_CYCLE, 3)},
                                "lane_utilisation": pmc_k.get("valu_lane_utilisation"), "wait_any_frac_of_wave_cycles": pmc_k.get("wait_any_frac_of_wave_cycles")}
            if pmc_k.get("vl1d"):
                # the second limit of the traversal kernels: every lane of a node or triangle load asks the vector L1 for its own line (profiles/r03_issue_peak.txt,
                # "global_load_dwordx4, 64 lanes in 64 L1-resident lines", holds the rate the L1 serves such loads at)
                roof["vl1d"] = dict(pmc_k["vl1d"], note="TCP_TOTAL_CACHE_ACCESSES_sum / 256 CUs / (GRBM_GUI_ACTIVE / 8 XCDs) and TA_TA_BUSY_sum likewise, kernels serialised")
            roof["counters_from"] = pmc_k.get("file")
            # the counters were collected by another process, possibly from another build: they describe THIS library only if the device sources and the kernel's
            # registers / LDS / scratch are what they were then (tools/collect_profile4.py stores both in the file)
            now = {"source_fingerprint": source_fingerprint(), "kernel_info": r.kernel_info(0 if f.get("key") == "closest" else 1) if dom.startswith("k_trace") else None}
            then = {"source_fingerprint": pmc_k.get("source_fingerprint"), "kernel_info": pmc_k.get("kernel_info")}
            same_kernel = then["kernel_info"] is None or now["kernel_info"] is None or all(then["kernel_info"].get(k) == now["kernel_info"].get(k) for k in ("vgprs", "static_lds_bytes", "scratch_bytes"))
            roof["counters_stale"] = not (then["source_fingerprint"] == now["source_fingerprint"] and same_kernel)
            roof["counters_built_from"] = {"git_head": pmc_k.get("git_head"), "source_fingerprint": then["source_fingerprint"], "this_run": now["source_fingerprint"]}
        if unusable:
            roof["counters_unusable"] = unusable
        freeze_roofline_fields(roof)
        shade = shade_record(r, args.config, spp, timed, alone, cnt, steps) if world == 1 else None
        sec = lambda k: fam[k]["alone"] * 1e-3  # seconds per step with the kernel alone on the GPU
        rates = {"closest_hit_grays_per_s": round(cnt["rays_closest"] / sec("k_trace_closest_stream") / 1e9, 3) if sec("k_trace_closest_stream") > 0 else None,
                 "secondary_grays_per_s": round(cnt["rays_shadow"] / sec("k_trace_secondary_stream") / 1e9, 3) if sec("k_trace_secondary_stream") > 0 else None,
                 "shaded_ghits_per_s": round(cnt["shaded_hits"] / sec("k_shade") / 1e9, 3) if sec("k_shade") > 0 else None,
                 "per_sample": {"closest_rays": round(cnt["rays_closest"] / max(paths_per_step, 1), 4), "secondary_rays": round(cnt["rays_shadow"] / max(paths_per_step, 1), 4),
                                "shaded_hits": round(cnt["shaded_hits"] / max(paths_per_step, 1), 4)},
                 "note": "each kernel alone on the GPU; per_sample: what one camera sample of this workload costs -- on configs[2] 84 % of the camera rays miss the scene bounds (the thin-lens model "
                         "puts the lens a focal length behind the camera origin) and end in k_generate, so Msamples/s of this workload is not a general-scene number: configs[3] (an interior, every ray hits) is"}
        # ---- the whole frame: algorithmic bytes of every counted kernel family over the step, and the fabric-side bytes the counters saw over a step
        alg_step = sum(v["bytes"] for v in fam.values() if v["bytes"] is not None)
        whole = {"algorithmic_bytes_per_step": int(alg_step), "algorithmic_gbs": round(alg_step / (dt / steps) / 1e9, 1), "algorithmic_gbs_over_hbm_peak": round(alg_step / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5),
                 "note": "traversal + shade + generate + accumulate bytes (DESIGN.md 4) over the wall time of a step; the fused tail, route and sort kernels are not counted"}
        if pmc and pmc.get("frame_traffic_bytes_per_spp"):
            fb = pmc["frame_traffic_bytes_per_spp"] * spp
            whole.update({"fabric_bytes_per_step": int(fb), "fabric_gbs": round(fb / (dt / steps) / 1e9, 1), "fabric_frac_of_hbm_peak": round(fb / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5),
                          "fabric_from": pmc.get("file")})
        sm = sorted(step_ms)
        out = {
            "metric": "Msamples/s at 1920x1080, max_depth=8" if args.config in (1, 2, 3) else f"Msamples/s at {WIDTH}x{HEIGHT}, max_depth={MAX_DEPTH}", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": w["name"], "spp_per_step": spp, "spp_per_pass": round(pool_spp, 3) if isinstance(pool_spp, float) else pool_spp, "passes_per_step": round(timed["n_passes"] / max(steps, 1), 2),
                       "sky_pixel_sample_share": round(timed["sky_pixel_samples"] / max(timed["paths"], 1), 4),  # samples of pixels no ray of which can reach the scene bounds: rendered by k_sky_pixels, outside the passes (the path pools hold the other pixels only, so a pass takes more samples of them than spp_per_pass)
                       "path_pools": {"pools": n_pools, "bytes_per_path": slot_bytes, "gb": round(pool_bytes_timed / 1e9, 1), "paths": int(pool_paths_timed), "budget_gb": round(n_pools * slot_bytes * n_owned * pool_spp / 1e9, 1),
                                      "note": "gb = device memory the pools held after the timed region (fh_path_pool_allocated): a pool is allocated for the paths its passes start, pixels that can see the scene x samples per pass; budget_gb = what the caller allowed (fh_set_path_pool)"}, "triangles": int(sc["indices"].shape[0]), "parallelism": f"pixel-tile x{world}" if world > 1 else "single GPU",
                       "gather": (("RCCL" if backend == "nccl" else backend + " (functional test: staged through host memory)") + " gather of packed float4 beauty tiles to rank 0 + fh_unpack_shards (one launch), inside the timed region") if world > 1 else "none",
                       "post": "bloom + chromatic aberration + tone map on the whole frame, inside the timed region" if post else "none"},
            "step_ms": {"min": round(sm[0], 3), "median": round(sm[len(sm) // 2], 3), "max": round(sm[-1], 3)},
            "source_fingerprint": source_fingerprint(),  # of the device sources this library was built from (what profiles/*_traffic_config*.json are checked against)
            "roofline": roof, **({"shade": shade} if shade else {}), "rates": rates, "whole_frame": whole,
            "kernel_ms_per_step": {"trace_closest": round(timed["trace_closest_ms"] / steps, 3), "trace_secondary": round(timed["trace_shadow_ms"] / steps, 3), "shade": round(timed["shade_ms"] / steps, 3),
                                   "tail": round(timed["tail_ms"] / steps, 3), "generate": round(timed["generate_ms"] / steps, 3), "accumulate": round(timed["accumulate_ms"] / steps, 3),
                                   "route_and_sort": round(timed["queue_ms"] / steps, 3), "render_total": round(timed["render_ms"] / steps, 3),
                                   "note": "HIP-event spans on the library's three streams; passes overlap, so the spans do not add up to render_total"},
            "kernel_ms_per_step_alone": {"trace_closest": round(alone["trace_closest_ms"], 3), "trace_secondary": round(alone["trace_shadow_ms"], 3), "shade": round(alone["shade_ms"], 3),
                                         "tail": round(alone["tail_ms"], 3), "generate": round(alone["generate_ms"], 3), "accumulate": round(alone["accumulate_ms"], 3),
                                         "route_and_sort": round(alone["queue_ms"], 3), "render_total": round(alone["render_ms"], 3),
                                         "note": "one untimed step with FH_FLAG_SERIAL_PASSES: every kernel alone on the GPU (these add up)"},
            **({"post_chain": {"ms_per_frame": round(timed["post_ms"] / max(timed["n_post_launches"], 1), 3), "frames": int(timed["n_post_launches"]),
                               "algorithmic_gbs": round(WIDTH * HEIGHT * (16 + 16 + 16 + 16 + 16 + 16 + 16) / (timed["post_ms"] / max(timed["n_post_launches"], 1) * 1e-3) / 1e9, 1) if timed["post_ms"] > 0 else None,
                               "note": "bloom threshold (16 B in, 16 out per pixel) + 33x33 blur through a 48x48 LDS tile (16 + 16 in, 16 out) + tone map (16 in, 16 out); the blur is LDS-bound: 1089 LDS taps per pixel"}}
               if post else {}),
            "bvh": {"build_ms": round(timed["bvh_build_ms"], 2), "nodes": timed["bvh_nodes"], "node_bytes": timed["bvh_node_bytes"], "tri_bytes": timed["bvh_tri_bytes"], "depth": timed["bvh_depth"]},
        }
        if world == 1:
            sm_ = scaling_model(args.config, spp)
            if sm_:
                out["scaling_model"] = sm_
        if extras:
            rows = (HEIGHT // 2 - 4, HEIGHT // 2 + 4)  # eight rows through the middle of the frame (every configuration has geometry there)
            out["parity"] = parity_block(r, w, cam, layers2, bufs2, rows, 2)
            out["latency"] = latency_block(r, w, cam, layers2)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
            out["cpu_baseline_1t"] = cpu_baseline_1t()
        if extras and args.config == 2 and not args.no_general_scene:
            r.close()  # (the headline's path pools go back to the device first)
            r = None
            out["general_scene"] = general_scene_block(local_rank, tmp.name, (bw_read, bw_copy), pool_gb=args.pool_gb)
        refused = refuse_bad_fracs(out)
        if refused:
            out["fractions_refused"] = [{"field": k, "value": v} for k, v in refused]
            print("bench.py: refused to print fractions outside [0, 1]: " + ", ".join(f"{k} = {v}" for k, v in refused), file=sys.stderr, flush=True)
            out["failed"] = "a fraction of a roof outside [0, 1]: the line is printed for the record, the run exits with code 3"
        print(json.dumps(out), flush=True)
        if refused:
            failed_rc = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if r is not None:
        r.close()
    tmp.cleanup()
    if failed_rc:
        sys.exit(failed_rc)


if __name__ == "__main__":
    main()
