#!/bin/bash
# round 4, call 7: single-pass calls with secondary rays + next closest-hit rays in ONE launch (FH_MERGE=0: two launches on two streams): parity suite, then 1-spp / 16-spp frame times off / on twice
cd $GRAFT_REPO_ROOT
echo "== full GPU suite"; timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== latency"; bash tools/gpu_latency_ab.sh "FH_MERGE=0" "1 2 3"
