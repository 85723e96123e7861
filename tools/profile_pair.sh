#!/bin/bash
# two configurations' round-3 profile sets in one gpurun call: tools/profile_pair.sh <tagA> <cfgA> <sppA> <tagB> <cfgB> <sppB>
bash tools/profile_round3.sh $1 $2 $3 $7 > gpurun_out/$1_profile.log 2>&1; tail -3 gpurun_out/$1_profile.log | cut -c1-300
bash tools/profile_round3.sh $4 $5 $6 $7 > gpurun_out/$4_profile.log 2>&1; tail -3 gpurun_out/$4_profile.log | cut -c1-300
