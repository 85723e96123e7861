"""Cross-check of roofline.avg_launch_ms (HIP events on the library's streams inside bench.py's timed region) against rocprofv3's kernel trace of the SAME command:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras --no-general-scene > line.json
    python3 tools/timed_region_avg.py <kernel_trace.csv> line.json
The trace holds the dominant kernel's launches of the warm-up steps, the timed steps and the untimed serial step (the two counting replays run instrumented instantiations, other
names).  The launches of the timed region are those in the middle: the first warmup / steps x N and the last N / steps are dropped (N = the line's roofline.launches), and the
mean duration of the rest is printed next to the line's figure."""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = line["roofline"]
name = r["kernel"]
ks = sorted(((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in rows if name + "<false" in x["Kernel_Name"]), key=lambda k: k[0])
n_timed, steps, warmup = int(r["launches"]), line["steps"], line["warmup"]
per_step = n_timed / steps
lead = len(ks) - n_timed - round(per_step)          # launches before the timed region (warm-up; adaptive depth makes its steps a launch shorter or longer than the timed ones)
mid = ks[lead:lead + n_timed]
avg = sum(e - s for s, e in mid) / len(mid) / 1e6
serial = ks[lead + n_timed:]
print(json.dumps({"kernel": name, "launches_in_trace": len(ks), "taken_as_warm_up": lead, "timed_region": len(mid), "serial_step": len(serial),
                  "rocprof_avg_ms_timed_region": round(avg, 4), "rocprof_avg_ms_serial_step": round(sum(e - s for s, e in serial) / max(len(serial), 1) / 1e6, 4),
                  "rocprof_avg_ms_all_launches": round(sum(e - s for s, e in ks) / len(ks) / 1e6, 4),
                  "bench_avg_launch_ms": r["avg_launch_ms"], "bench_avg_launch_ms_alone": r["avg_launch_ms_alone"], "ratio_timed_region": round(avg / r["avg_launch_ms"], 4)}))
