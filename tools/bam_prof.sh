#!/bin/bash
# rocprofv3 kernel statistics of bin/sam2bam on a synthetic .sam (made by tools/bam_bench.py in $TMPDIR).  Usage: bash tools/bam_prof.sh <tag> [level]
set -e
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
LEVEL=${2:-1}
OUT=$REPO/gpurun_out/bamprof_$TAG
mkdir -p $OUT
python $REPO/tools/bam_bench.py 2000000 $LEVEL > $OUT/bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bam -- $REPO/microcket_amd/bin/sam2bam -l $LEVEL -o /tmp/prof.bam /tmp/bench.sam > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
cp $f $OUT/kernel_stats.csv
cat $OUT/bench.txt
cut -c1-170 $OUT/kernel_stats.csv
