#!/usr/bin/env python3
"""isa_stats.py -- register / scratch / LDS numbers and an instruction histogram per kernel from a gfx950 assembly listing.

    hipcc <flags of csrc/Makefile> -S --cuda-device-only -o /tmp/render.s fredholm_amd/csrc/render.hip
    python tools/isa_stats.py /tmp/render.s k_trace_secondary_stream            # table of every instantiation
    python tools/isa_stats.py /tmp/render.s k_trace_secondary_stream --hist 'ILb0ELb0ELb0E'   # instruction classes of one of them

The histogram is static (instructions in the code object, not executed ones); it is what the share of FMA-class instructions in the
traversal loop and the per-node instruction counts quoted in DESIGN.md come from.
"""
import collections
import re
import sys


def kernels(path):
    meta = {}
    cur = None
    for line in open(path, errors="replace"):
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            cur = m.group(1)
            meta[cur] = {}
            continue
        if cur:
            m = re.match(r"\s+\.(vgpr_count|sgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s+(\d+)", line)
            if m:
                meta[cur][m.group(1)] = int(m.group(2))
    return meta


def body(path, symbol):
    out, on = [], False
    for line in open(path, errors="replace"):
        if line.startswith(symbol + ":"):
            on = True
            continue
        if on:
            if line.startswith(".Lfunc_end"):
                break
            out.append(line)
    return out


def main():
    path, pat = sys.argv[1], sys.argv[2]
    meta = kernels(path)
    names = [k for k in meta if pat in k]
    if "--hist" in sys.argv:
        sel = sys.argv[sys.argv.index("--hist") + 1]
        names = [k for k in names if sel in k]
        for k in names:
            h = collections.Counter()
            for line in body(path, k):
                m = re.match(r"\s+([a-z_0-9]+)\s", line)
                if m and not line.lstrip().startswith((".", ";")):
                    h[m.group(1)] += 1
            tot = sum(h.values())
            valu = sum(v for i, v in h.items() if i.startswith("v_"))
            print(f"{k}\n  {tot} instructions, {valu} VALU, {sum(v for i, v in h.items() if i.startswith('s_'))} SALU, "
                  f"{sum(v for i, v in h.items() if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))} VMEM, {sum(v for i, v in h.items() if i.startswith('ds_'))} LDS")
            for i, v in h.most_common(45):
                print(f"    {v:6d}  {i}")
        return
    for k in names:
        m = meta[k]
        print(f"{m.get('vgpr_count', 0):4d} vgpr {m.get('agpr_count', 0):3d} agpr {m.get('sgpr_count', 0):4d} sgpr  scratch {m.get('private_segment_fixed_size', 0):5d} B  lds {m.get('group_segment_fixed_size', 0):6d} B  "
              f"spills v{m.get('vgpr_spill_count', 0)} s{m.get('sgpr_spill_count', 0)}  {k}")


if __name__ == "__main__":
    main()
