#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import collections
import csv
import glob
import sys


def main(patterns):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for pat in patterns:
        for f in glob.glob(pat, recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                short = k.replace("void ", "").replace("fh::(anonymous namespace)::", "").split("(")[0]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
        if not k.startswith("k_"):
            continue
        cs = acc[k]
        n = max(len(v) for v in cs.values())
        print(f"{k}  dispatches={n}")
        for c in sorted(cs):
            v = cs[c]
            print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  total {sum(v):18.0f}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["gpurun_out/pmc_*/**/*counter_collection.csv"])
