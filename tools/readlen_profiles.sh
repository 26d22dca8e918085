#!/bin/bash
# Round evidence per read length (GPU box, repo root):  bash tools/readlen_profiles.sh <tag>
# k_fast time per block for 150 / 100 / 60 bp (geometry chosen from the probe), SQ counters and FETCH_SIZE / WRITE_SIZE per length
# (passes of their own), phase shares of the stamp build per length.
set -e
TAG=${1:-readlen}; R=$PWD; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 300 python3 tools/geom_bench.py --tiles auto > $O/geom.txt 2>&1
timeout -k 10 200 python3 tools/geom_bench.py --tiles auto --sam >> $O/geom.txt 2>&1
echo "geom done"
timeout -k 10 900 bash tools/pmc_geom.sh $O/pmc "150:auto 100:auto 60:auto" > $O/pmc.txt 2>&1
echo "pmc done"
for rl in 150 100 60; do
  echo "== read_len $rl" >> $O/shares.txt
  MKT_TILES=auto MKT_LIB=$R/microcket_amd/libmkt_hip_stamps.so timeout -k 10 200 python3 tools/phase_shares.py 4000000 no $rl >> $O/shares.txt 2>&1
done
cat $O/geom.txt $O/pmc.txt $O/shares.txt
