#!/bin/bash
# Diagnostic: instruction-cache and issue counters of k_fast (GPU box, from the repo root): bash tools/pmc_icache.sh gpurun_out/pmc_icache
set -e
OUT=$1; R=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/pi_run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import microcket_amd as m
ctx = m.Context("unc", 0.5, 10, False, 8, device=0, tiles=m.TILES_FAST)
ds = ctx.dataset(20260105, 0, 4000000, 1 << 21)
for _ in range(2):
    for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
    ctx.sync()
PY
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$OUT/a -- python3 /tmp/pi_run.py > /dev/null 2> $R/$OUT/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/$OUT/b -- python3 /tmp/pi_run.py > /dev/null 2> $R/$OUT/b.err
cd $R
python3 - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            if "k_fast" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(agg): print(f"{k:32s} {agg[k] / n[k]:16.0f} per launch ({n[k]} launches)")
PY
