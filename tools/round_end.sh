#!/bin/bash
# tools/round_end.sh [rNN] -- the gate at the end of a round ("Next" 1c): ONE gpurun call on a CLEAN tree
#   1. `git status --porcelain` must be empty: what runs on the GPU box is exactly HEAD
#   2. on the box (tools/round_end_gpu.sh): pytest -m gpu -x -q, __graft_entry__.smoke(), python bench.py (the driver's default line)
#   3. the log comes back as profiles/rNN_final_gpu_tests.log and begins with HEAD, the hash of the paths the GPU run depends on (tools/covered_tree_hash.py) and bench.py's
#      source_fingerprint; tests/test_zz_round_end.py (CPU) fails when HEAD's covered paths are no longer what the log saw.
# Nothing under the covered paths may be committed after this; the log itself (profiles/rNN_final_*) is not covered.
set -euo pipefail
cd "$(dirname "$0")/.."
tag=${1:-r06}
if [ -n "$(git status --porcelain)" ]; then
  echo "round_end: the tree is not clean -- commit first:" >&2
  git status --porcelain >&2
  exit 1
fi
python __graft_entry__.py > /dev/null   # the in-tree .so files the box will load are built from HEAD's sources
if [ -n "$(git status --porcelain)" ]; then echo "round_end: the build changed tracked files" >&2; exit 1; fi
head=$(git rev-parse HEAD)
tree=$(python tools/covered_tree_hash.py)
fp=$(python -c "import bench; print(bench.source_fingerprint())")
mkdir -p gpurun_out
rm -f gpurun_out/${tag}_final_gpu_tests.log gpurun_out/${tag}_final_bench.json
/usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/round_end_gpu.sh $tag $head $tree $fp" || true
test -s gpurun_out/${tag}_final_gpu_tests.log || { echo "round_end: no log came back" >&2; exit 1; }
cp gpurun_out/${tag}_final_gpu_tests.log profiles/${tag}_final_gpu_tests.log
test -s gpurun_out/${tag}_final_bench.json && cp gpurun_out/${tag}_final_bench.json profiles/${tag}_final_bench.json
grep -q "^round_end: GREEN" profiles/${tag}_final_gpu_tests.log || { echo "round_end: NOT GREEN -- see profiles/${tag}_final_gpu_tests.log" >&2; tail -30 profiles/${tag}_final_gpu_tests.log >&2; exit 1; }
echo "round_end: green at $head; commit profiles/${tag}_final_gpu_tests.log (and nothing under the covered paths after it)"
