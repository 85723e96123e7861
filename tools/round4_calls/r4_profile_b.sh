#!/bin/bash
# round 4, profile call B: the round's profile set of configs[3] (pass size of its default run: 103 samples per pixel -> 309 in the counter passes)
cd $GRAFT_REPO_ROOT
bash tools/profile_round3.sh r04_g3 3 309
