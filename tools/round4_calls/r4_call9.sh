#!/bin/bash
# round 4, call 9: re-tune the streaming parameters at seven workgroups per CU (chunk sizes, refill threshold, flush threshold), configs[3] (512 spp) and configs[2]
cd $GRAFT_REPO_ROOT
for cfg in 3 2; do
  S="--spp 512 --steps 2 --warmup 1 --no-extras"; [ $cfg = 2 ] && S="--steps 6 --warmup 2 --no-extras"
  echo "== configs[$cfg]"
  bash tools/gpu_env_ab.sh "FH_X=0 FH_STREAM_CHUNK=32 FH_STREAM_CHUNK=128 FH_STREAM_CHUNK_CLOSEST=64 FH_STREAM_CHUNK_CLOSEST=256 FH_STREAM_REFILL=16 FH_STREAM_REFILL=32 FH_COOP_T=24 FH_COOP_T=48 FH_X=0" "$cfg" "$S"
done
