"""after tools/profile_round2.sh <tag>: copy the summaries into profiles/ and write profiles/<tag>_traffic.json from the PMC summary.  python tools/collect_profile.py <tag>"""
import json, re, shutil, sys
tag = sys.argv[1]
for f in ("bench.json", "kernel_stats.csv", "serial_bench.json", "serial_kernel_stats.csv", "pmc_summary.txt"):
    shutil.copy(f"gpurun_out/{tag}_{f}", f"profiles/{tag}_{f}")
s = open(f"gpurun_out/{tag}_pmc_summary.txt").read()
name = "k_trace_secondary_stream<false, false, false>"
blk = s[s.index(name):]
blk = blk[:blk.index("WRITE_SIZE") + 200]
g = lambda n: float(re.search(n + r"\s+mean\s+([\d.]+)", blk).group(1))
disp = int(re.search(r"dispatches=(\d+)", blk).group(1))
fetch, write = g("FETCH_SIZE"), g("WRITE_SIZE")
d = {"kernel": name, "config": 2,
     "source": f"profiles/{tag}_pmc_summary.txt (tools/profile_round2.sh {tag}: separate rocprofv3 --pmc passes -- SQ group a, SQ group b, FETCH_SIZE, WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- of bench.py --steps 1 --warmup 1 --spp 384 --no-cpu-baseline; mean over the {disp} dispatches of the kernel; same launch size as the timed runs: 128 spp of the 1080p frame per pass)",
     "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
     "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; KB = 1024 B.  The guide calibrates the doubling on wide streaming reads only; for the scattered 16-B node/triangle loads of this kernel it is an upper bound.",
     "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
     "tcc_hit_rate": round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4),
     "valu_insts_per_launch": int(g("SQ_INSTS_VALU")), "vmem_rd_insts_per_launch": int(g("SQ_INSTS_VMEM_RD")), "vmem_wr_insts_per_launch": int(g("SQ_INSTS_VMEM_WR")),
     "lds_insts_per_launch": int(g("SQ_INSTS_LDS")), "valu_lane_utilisation": round(g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU")), 4),
     "wait_any_frac_of_wave_cycles": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4), "waves_per_launch": int(g("SQ_WAVES")),
     "note": "fabric-side bytes (L2 misses; Infinity-Cache hits are counted)."}
json.dump(d, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
b = json.load(open(f"profiles/{tag}_bench.json"))
print(b["value"], b["step_ms"], b["roofline"], b["kernel_ms_per_step_alone"], b["rates"], b["cpu_baseline"]["value"], b["bvh"])
print({k: d[k] for k in ("traffic_bytes_per_launch", "tcc_hit_rate", "valu_insts_per_launch", "valu_lane_utilisation", "wait_any_frac_of_wave_cycles", "waves_per_launch")})
