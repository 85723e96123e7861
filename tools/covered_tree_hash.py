#!/usr/bin/env python3
"""sha256 over HEAD's tree entries (mode, blob id, path) of everything the GPU run of a round's end depends on: fredholm_amd, include, oracle, bench.py, __graft_entry__.py, tests,
profiles/*traffic*.json, profiles/*issue_peak.json.  tools/round_end.sh logs it; tests/test_zz_round_end.py compares."""
import fnmatch
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIRS = ("fredholm_amd/", "include/", "tests/", "oracle/")  # (round 6: the whole package -- the GPU tests and bench.py import its Python half too -- and the checker)
FILES = ("bench.py", "__graft_entry__.py")
GLOBS = ("profiles/*traffic*.json", "profiles/*issue_peak.json")


def covered(path):
    return path.startswith(DIRS) or path in FILES or any(fnmatch.fnmatch(path, g) for g in GLOBS)


def covered_tree_hash(rev="HEAD"):
    out = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "-r", rev], text=True)
    h = hashlib.sha256()
    n = 0
    for line in sorted(out.splitlines()):
        meta, path = line.split("\t", 1)
        if covered(path):
            h.update(line.encode() + b"\n")
            n += 1
    if n == 0:
        raise SystemExit("no covered path in " + rev)
    return h.hexdigest()[:32]


if __name__ == "__main__":
    print(covered_tree_hash(sys.argv[1] if len(sys.argv) > 1 else "HEAD"))
