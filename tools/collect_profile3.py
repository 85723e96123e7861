"""after tools/profile_round3.sh <tag> <config> [pmc spp]: copy the summaries into profiles/ and write profiles/<tag>_traffic_config<N>.json from the PMC summary -- the
counters of the configuration's dominant kernel (what bench.py quotes as roofline.traffic / roofline.valu) and the fabric-side bytes of a whole frame per sample per pixel-frame
(whole_frame.fabric_*).  python tools/collect_profile3.py <tag> <config> [pmc spp]"""
import json, re, shutil, sys
tag, cfg = sys.argv[1], int(sys.argv[2])
pspp = int(sys.argv[3]) if len(sys.argv) > 3 else 384
for f in ("bench.json", "kernel_stats.csv", "serial_kernel_stats.csv", "pmc_summary.txt"):
    shutil.copy(f"gpurun_out/{tag}_{f}", f"profiles/{tag}_{f}")
b = json.load(open(f"profiles/{tag}_bench.json"))
dom = b["roofline"]["kernel"]
s = open(f"gpurun_out/{tag}_pmc_summary.txt").read()
blocks = re.split(r"\n(?=k_)", s)
def parse(blk):
    head = blk.splitlines()[0]
    d = {"name": head.split("  dispatches=")[0], "dispatches": int(re.search(r"dispatches=(\d+)", head).group(1))}
    for m in re.finditer(r"^\s+(\w+)\s+mean\s+([\d.]+)\s+total\s+(\d+)", blk, re.M):
        d[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    return d
ks = [parse(x) for x in blocks if x.startswith("k_")]
# the dominant kernel's un-instrumented instantiation with the most dispatches
cand = [k for k in ks if k["name"].startswith(dom) and not k["name"].startswith(dom + "<true")]
k = max(cand, key=lambda k: k["dispatches"] * k.get("SQ_INSTS_VALU", (0, 0))[0])
g = lambda n: k[n][0]
# the vector-memory side, from its own passes (gpurun_out/<tag>_l1_summary.txt -> profiles/): GRBM_GUI_ACTIVE sums the eight XCDs' cycles, the TA / TCP counters the 256 CUs' units
vl1d = None
try:
    shutil.copy(f"gpurun_out/{tag}_l1_summary.txt", f"profiles/{tag}_l1_summary.txt")
    kl = [parse(x) for x in re.split(r"\n(?=k_)", open(f"gpurun_out/{tag}_l1_summary.txt").read()) if x.startswith("k_")]
    kl = [x for x in kl if x["name"] == k["name"]][0]
    cyc = kl["GRBM_GUI_ACTIVE"][0] / 8.0
    vl1d = {"accesses_per_launch": int(kl["TCP_TOTAL_CACHE_ACCESSES_sum"][0]), "l2_read_requests_per_launch": int(kl["TCP_TCC_READ_REQ_sum"][0]), "cycles_per_launch": int(cyc),
            "accesses_per_cycle_per_cu": round(kl["TCP_TOTAL_CACHE_ACCESSES_sum"][0] / 256.0 / cyc, 4), "ta_busy_frac": round(kl["TA_TA_BUSY_sum"][0] / 256.0 / cyc, 4),
            "accesses_per_load_instruction": round(kl["TCP_TOTAL_CACHE_ACCESSES_sum"][0] / max(kl["TA_FLAT_READ_WAVEFRONTS_sum"][0], 1.0), 2),
            "peak_accesses_per_cycle_per_cu": 1.0, "source": f"profiles/{tag}_l1_summary.txt; peak: profiles/r03_issue_peak.txt (64 lanes in 64 L1-resident lines: 64 cycles per load instruction and CU)"}
except Exception as e:
    print("no vector-memory passes:", e)
renders = 5  # --warmup 1 --steps 1 + the serial step + the two counting replays (the second with every ray started at the root)
pmc_line = [ln for ln in open(f"gpurun_out/{tag}_sqa.log") if ln.startswith("{")][-1]
pmc_cfg = json.loads(pmc_line)["config"]
pmc_pass_spp = pmc_cfg["spp_per_pass"]  # the nominal launch size the counters belong to: bench.py quotes them only for runs with the same samples per pass
# ... and the pass the library really submitted in the counter run (a call that splits off its sky pixels cuts itself differently): bench.py flags a run whose passes differ
pmc_submitted = round(pmc_cfg["spp_per_step"] / max(pmc_cfg.get("passes_per_step", 0) or 1, 1e-9), 2) if pmc_cfg.get("passes_per_step") else None
run_spp = pmc_cfg["spp_per_step"]  # samples per pixel of ONE render of the counter run (round 6, r6-15: two passes of `pspp`)
frame = sum((2 * x.get("FETCH_SIZE", (0, 0))[1] + x.get("WRITE_SIZE", (0, 0))[1]) * 1024 for x in ks)
d = {"kernel": k["name"], "config": cfg, "spp_per_pass": pmc_pass_spp, "submitted_spp_per_pass": pmc_submitted,
     "source": f"profiles/{tag}_pmc_summary.txt (tools/profile_round3.sh {tag} {cfg} {pspp}: separate rocprofv3 --pmc passes -- SQ group a, SQ group b, FETCH_SIZE, WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- of "
               f"bench.py --config {cfg} --steps 1 --warmup 1 --spp {run_spp} --no-cpu-baseline --no-extras, FH_PIPELINE=0, FH_TAIL_DEPTH pinned to the steady depth: calls of {pmc_cfg.get('passes_per_step')} passes; mean over the {k['dispatches']} dispatches of the kernel)",
     "FETCH_SIZE_KB_per_launch": g("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": g("WRITE_SIZE"),
     "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; KB = 1024 B.  The guide calibrates the doubling on wide streaming reads only; for scattered 16-B loads it is an upper bound.",
     "traffic_bytes_per_launch": int((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024),
     "tcc_hit_rate": round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4),
     "valu_insts_per_launch": int(g("SQ_INSTS_VALU")), "vmem_rd_insts_per_launch": int(g("SQ_INSTS_VMEM_RD")), "vmem_wr_insts_per_launch": int(g("SQ_INSTS_VMEM_WR")),
     "lds_insts_per_launch": int(g("SQ_INSTS_LDS")), "valu_lane_utilisation": round(g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU")), 4),
     "wait_any_frac_of_wave_cycles": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4), "waves_per_launch": int(g("SQ_WAVES")),
     "grbm_gui_active_per_launch": g("GRBM_GUI_ACTIVE"),
     "vl1d": vl1d,
     "frame_traffic_bytes_per_spp": frame / (renders * run_spp),
     "frame_note": f"(2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over every kernel of the run / ({renders} renders x {run_spp} spp): fabric-side bytes one sample per pixel of the whole frame costs",
     "note": "fabric-side bytes (L2 misses; Infinity-Cache hits are counted)."}
# ---- round 6: the cooperative triangle test's LDS bank conflicts of the dominant kernel, and the shade kernels' own block (bench.py: `shade` record) -- every un-instrumented k_shade
# instantiation of the run summed; the counter run renders `renders` times and the shade kernels are the same in all of them
if "SQ_LDS_BANK_CONFLICT" in k and "SQ_LDS_IDX_ACTIVE" in k:
    d["lds_bank_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0), 4)
sh = [x for x in ks if x["name"].startswith("k_shade<")]
if sh:
    tot = lambda n: sum(x.get(n, (0, 0))[1] for x in sh)
    pmc_full = json.loads(pmc_line)
    per_sample = (pmc_full.get("rates") or {}).get("per_sample", {}).get("shaded_hits")
    res = (1920 * 1080) if cfg != 4 else (3840 * 2160)
    d["shade"] = {"kernels": sorted(x["name"] for x in sh), "dispatches": int(sum(x["dispatches"] for x in sh)),
                  "valu_insts_total": int(tot("SQ_INSTS_VALU")), "valu_lane_utilisation": round(tot("SQ_THREAD_CYCLES_VALU") / max(64 * tot("SQ_INSTS_VALU"), 1), 4),
                  "wait_any_frac_of_wave_cycles": round(tot("SQ_WAIT_ANY") / max(tot("SQ_WAVE_CYCLES"), 1), 4),
                  "tcc_hit_rate": round(tot("TCC_HIT_sum") / max(tot("TCC_HIT_sum") + tot("TCC_MISS_sum"), 1), 4),
                  "traffic_bytes_total": int((2 * tot("FETCH_SIZE") + tot("WRITE_SIZE")) * 1024),
                  "lds_bank_conflict_frac": round(tot("SQ_LDS_BANK_CONFLICT") / max(tot("SQ_LDS_IDX_ACTIVE"), 1), 4) if tot("SQ_LDS_IDX_ACTIVE") else None,
                  "shaded_hits_in_counter_run": int(per_sample * res * run_spp * renders) if per_sample else None,
                  "note": f"summed over the {renders} renders of the counter run (timed step, warm-up, serial step, two counting replays: the shade kernels are the same in all of them)"}
# what the counters belong to (bench.py: roofline.counters_stale): the device sources and the kernel's registers / LDS / scratch of the run that was profiled, and the commit it was collected at
import subprocess
d["source_fingerprint"] = b.get("source_fingerprint")
d["kernel_info"] = b["roofline"].get("kernel_info")
try:
    d["git_head"] = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True).strip() + (" + uncommitted changes" if subprocess.check_output(["git", "status", "--porcelain", "--", "fredholm_amd/csrc", "include"], text=True).strip() else "")
except Exception:
    d["git_head"] = None
json.dump(d, open(f"profiles/{tag}_traffic_config{cfg}.json", "w"), indent=1)
print(b["value"], b["step_ms"], {x: b["roofline"][x] for x in ("bound", "kernel", "achieved", "peak", "frac") if x in b["roofline"]}, b["kernel_ms_per_step_alone"], b["rates"], b.get("cpu_baseline", {}).get("value"))
print({x: d[x] for x in ("kernel", "traffic_bytes_per_launch", "tcc_hit_rate", "valu_insts_per_launch", "valu_lane_utilisation", "wait_any_frac_of_wave_cycles", "waves_per_launch", "frame_traffic_bytes_per_spp")})
