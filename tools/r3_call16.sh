#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 560 bash tools/profile_round3.sh r03_b3 3 324 | tail -2
timeout -k 10 560 bash tools/profile_round3.sh r03_b4 4 78 | tail -2
