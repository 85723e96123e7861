#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 bash tools/profile_round3.sh r03_b 2 384 | tail -3
timeout -k 10 500 bash tools/profile_round3.sh r03_b1 1 258 | tail -3
