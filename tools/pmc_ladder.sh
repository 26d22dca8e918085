#!/bin/bash
# Diagnostic: SQ instruction counters of k_fast when every tile stops after phase k (stamp build).
# usage (on the GPU box, from the repo root): bash tools/pmc_ladder.sh gpurun_out/pmc_ladder
set -e
OUT=$1; R=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/pl_run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import microcket_amd as m
ctx = m.Context("unc", 0.5, 10, False, 8, device=0, tiles=m.TILES_FAST)
ds = ctx.dataset(20260105, 0, 2000000, 1 << 20)
for _ in range(2):
    for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
    try: ctx.sync()
    except Exception: pass
PY
cd /tmp
for k in 1 2 9 3 4 6 7 0; do
  MKT_DEBUG_STOP=$k MKT_NO_STAMPS=1 MKT_LIB=$R/microcket_amd/libmkt_hip_stamps.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/stop$k -- python3 /tmp/pl_run.py > /dev/null 2> $R/$OUT/stop$k.err
done
