"""The gate of a round's end (CPU; sorts last so that a red gate hides nothing else).

tools/round_end.sh runs, on a CLEAN tree and in one gpurun call, `pytest -m gpu -x -q`, `__graft_entry__.smoke()` and the default `python bench.py`, and commits the log as
profiles/rNN_final_gpu_tests.log.  The log begins with HEAD, a hash over HEAD's tree entries of every path that run depends on (tools/covered_tree_hash.py: fredholm_amd,
include, oracle, bench.py, __graft_entry__.py, tests, profiles/*traffic*.json, profiles/*issue_peak.json) and bench.py's source fingerprint.  This test fails when the newest such log is not green or is
about other contents than HEAD's: nothing under those paths may be committed after the round's final GPU run (round 4 committed a counter file after it, which changed what a
GPU test asserted, and the driver's run stopped at test 5 of 201).
"""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_log():
    logs = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_final_gpu_tests.log")):
        m = re.match(r"r(\d+)_final_gpu_tests\.log$", os.path.basename(f))
        if m and open(f).readline().startswith("git_head: "):
            logs.append((int(m.group(1)), f))
    return max(logs)[1] if logs else None


def _header(path):
    h = {}
    for line in open(path):
        if line.startswith("----"):
            break
        k, _, v = line.partition(": ")
        h[k.strip()] = v.strip()
    return h


def test_final_gpu_run_of_the_round_is_green_and_is_about_this_tree():
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no .git here (a gpurun snapshot): the gate is checked where the history is")
    log = _newest_log()
    assert log, "no profiles/rNN_final_gpu_tests.log written by tools/round_end.sh"
    text = open(log).read()
    h = _header(log)
    assert "round_end: GREEN" in text and "pytest_rc: 0" in text and "smoke_rc: 0" in text and "bench_rc: 0" in text, f"{log} is not a green run"
    m = re.search(r"(\d+) passed", text)
    assert m and int(m.group(1)) >= 190 and " failed" not in text.split("---- __graft_entry__")[0], "the GPU suite did not run to its end"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ROOT)
    from covered_tree_hash import covered_tree_hash
    import bench
    assert h["source_fingerprint"] == h["source_fingerprint_on_box"] == bench.source_fingerprint(), "device sources changed after the round's final GPU run: run tools/round_end.sh again"
    assert h["covered_tree_sha256"] == covered_tree_hash("HEAD"), (
        f"{os.path.relpath(log, ROOT)} saw other contents of fredholm_amd, include, oracle, bench.py, __graft_entry__.py, tests or the counter files than HEAD holds: run tools/round_end.sh again")
    # the logged commit is in this history
    assert subprocess.run(["git", "-C", ROOT, "cat-file", "-e", h["git_head"] + "^{commit}"]).returncode == 0
