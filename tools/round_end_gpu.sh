#!/bin/bash
# the GPU-box half of tools/round_end.sh: bash tools/round_end_gpu.sh <tag> <git head> <covered tree hash> <source fingerprint>
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
tag=$1; head=$2; tree=$3; fp=$4
log=gpurun_out/${tag}_final_gpu_tests.log
mkdir -p gpurun_out
{
  echo "git_head: $head"
  echo "covered_tree_sha256: $tree"
  echo "source_fingerprint: $fp"
  echo "date: $(date -u +%Y-%m-%dT%H:%M:%SZ)"
} > $log
# the box has no .git: what it CAN check is that the device sources it holds hash to the fingerprint the clean tree had
here=$(python -c "import bench; print(bench.source_fingerprint())")
echo "source_fingerprint_on_box: $here" >> $log
[ "$here" = "$fp" ] || { echo "round_end: RED (the box's sources are not HEAD's)" >> $log; exit 1; }
echo "---- python -m pytest tests -m gpu -x -q" >> $log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider >> $log 2>&1
rc_tests=$?
echo "pytest_rc: $rc_tests" >> $log
[ $rc_tests -eq 0 ] || { echo "round_end: RED (pytest -m gpu)" >> $log; exit 1; }
echo "---- __graft_entry__.smoke()" >> $log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" >> $log 2>&1
rc_smoke=$?
echo "smoke_rc: $rc_smoke" >> $log
[ $rc_smoke -eq 0 ] || { echo "round_end: RED (smoke)" >> $log; exit 1; }
echo "---- python bench.py" >> $log
timeout -k 10 900 python bench.py > gpurun_out/${tag}_final_bench.json 2>> $log
rc_bench=$?
echo "bench_rc: $rc_bench" >> $log
python - >> $log 2>&1 <<PY
import json
j = json.loads(open("gpurun_out/${tag}_final_bench.json").read().strip().splitlines()[-1])
g = j.get("general_scene", {})
print("bench: value %.1f %s, ms_per_step %.2f, roofline.frac %.4f (%s), general_scene %.1f, fractions_refused %s" % (
    j["value"], j["unit"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel"], g.get("msamples_per_s", float("nan")), j.get("fractions_refused")))
assert "fractions_refused" not in j
PY
rc_line=$?
[ $rc_bench -eq 0 ] && [ $rc_line -eq 0 ] || { echo "round_end: RED (bench.py)" >> $log; exit 1; }
echo "round_end: GREEN at $head" >> $log
