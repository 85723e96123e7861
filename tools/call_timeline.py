"""Kernel-trace TIMELINE of the reference's own call pattern (1 sample per call in the GUI, controller.cpp:224; 16 in rtcamp8, rtcamp8.cpp:183-189).

  run (GPU box, under the profiler):   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 tools/call_timeline.py run <config> <spp> [calls]
  reduce (anywhere):                   python3 tools/call_timeline.py reduce <kernel_trace.csv> <spp> [calls] > profiles/r06_timeline_config3_1spp.txt

`run` warms up, then issues `calls` times fh_render(spp) + fh_sync.  `reduce` cuts the trace into calls (a one-pass call is the launches from its k_generate to its
k_accumulate), takes the call of MEDIAN length and prints every
launch of it (start and end relative to the call's first kernel, duration, gap to the end of whatever finished last before it) and the sums: busy time (union of the kernel intervals),
idle time inside the call (no kernel of the process running), the time per kernel family, the launches per family."""
import csv
import os
import sys


def run(cfg, spp, calls):
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import fredholm_amd as F
    with tempfile.TemporaryDirectory() as td:
        w = bench.workload(cfg, td)
        r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
    bench.apply_environment(r, w)
    if w["sun"] is not None and not w["dir_le"]:
        r.clear_directional_light()
    W, H = 1920, 1080
    r.set_resolution(W, H)
    L = F.RenderLayer(r, W, H)
    cam = F.Camera(**w["camera"])
    for _ in range(30):
        r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
    import time
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"configs[{cfg}] {W}x{H} fh_render({spp}) + fh_sync under the profiler: median {ts[len(ts) // 2]:.3f} ms, min {ts[0]:.3f} ms over {calls} calls", flush=True)
    r.close()


def short(name):
    n = name.replace("void ", "").replace("fh::(anonymous namespace)::", "").replace("fh::", "")
    return n.split("(")[0]


FAMILY = (("k_trace_merged", "trace (merged)"), ("k_trace_closest", "trace closest"), ("k_trace_secondary", "trace secondary"), ("k_shade", "shade"), ("k_miss_primary", "shade"), ("k_tail", "tail"),
          ("k_route", "route + sort"), ("k_sort", "route + sort"), ("k_cell", "route + sort"), ("k_count", "route + sort"), ("k_scatter", "route + sort"), ("k_scan", "route + sort"), ("k_generate", "generate"), ("k_bump", "generate"), ("k_accumulate", "accumulate"),
          ("k_sky", "sky pixels"), ("k_split", "sky pixels"))


def family(n):
    for key, fam in FAMILY:
        if n.startswith(key):
            return fam
    return "other (" + n[:24] + ")"


def reduce(path, spp, calls):
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows), key=lambda k: k[0])
    # a call = the launches from one k_generate up to and including the next k_accumulate; one-pass calls have exactly one of each
    gens = [i for i, k in enumerate(ks) if k[2].startswith("k_generate")]
    gens = gens[-calls:]
    spans = []
    for gi, g in enumerate(gens):
        end = gens[gi + 1] if gi + 1 < len(gens) else len(ks)
        acc = [i for i in range(g, end) if ks[i][2].startswith("k_accumulate")]
        if not acc:
            continue
        last = acc[-1]
        spans.append((ks[last][1] - ks[g][0], g, last))
    spans.sort()
    length, g, last = spans[len(spans) // 2]
    call = ks[g:last + 1]
    t0 = call[0][0]
    print(f"# {os.path.basename(path)}: {len(spans)} calls of fh_render({spp}); the call of median length: first kernel start to last kernel end {length / 1000:.1f} us, {len(call)} launches")
    print(f"# {'kernel':44s} {'queue':>6s} {'start us':>10s} {'end us':>10s} {'dur us':>9s} {'gap us':>8s}   (gap = start minus the latest end of any earlier launch; negative: overlaps)")
    busy_end = t0
    idle = 0
    fam_time, fam_n = {}, {}
    for s, e, n, q in call:
        gap = s - busy_end
        if gap > 0:
            idle += gap
        print(f"{n[:46]:46s} {q:>6s} {(s - t0) / 1000:10.1f} {(e - t0) / 1000:10.1f} {(e - s) / 1000:9.1f} {gap / 1000:8.1f}")
        busy_end = max(busy_end, e)
        f = family(n)
        fam_time[f] = fam_time.get(f, 0) + (e - s)
        fam_n[f] = fam_n.get(f, 0) + 1
    total = busy_end - t0
    print(f"# span {total / 1000:.1f} us = busy (union of kernel intervals) {(total - idle) / 1000:.1f} + idle between launches {idle / 1000:.1f} ({100.0 * idle / total:.1f} %)")
    print("# by family (kernel time summed, launches): " + "; ".join(f"{f} {t / 1000:.1f} us / {fam_n[f]}" for f, t in sorted(fam_time.items(), key=lambda x: -x[1])))
    short_l = [(e - s) for s, e, n, q in call if (e - s) < 20000]
    print(f"# launches under 20 us: {len(short_l)} of {len(call)}, {sum(short_l) / 1000:.1f} us in all")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    else:
        reduce(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 40)
