#!/bin/bash
# round 4, profile call A: issue-model constants of this fh_trace.h, then the round's profile set of configs[2] (bench line, kernel traces in flight and serial, SQ x2 / FETCH / TCC
# passes, the two vector-memory passes)
cd $GRAFT_REPO_ROOT
h=$(sha256sum fredholm_amd/csrc/fh_trace.h | cut -c1-16)
timeout -k 10 200 tools/micro/issue_peak.bin --json gpurun_out/r04_issue_peak.json $h 60 > gpurun_out/r04_issue_peak.txt 2>&1 || { tail -5 gpurun_out/r04_issue_peak.txt; exit 1; }
cp gpurun_out/r04_issue_peak.json profiles/r04_issue_peak.json; cat gpurun_out/r04_issue_peak.json
bash tools/profile_round3.sh r04_f 2 384
