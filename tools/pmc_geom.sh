#!/bin/bash
# Diagnostic: SQ counters and HBM traffic (FETCH_SIZE / WRITE_SIZE, passes of their own) of k_fast per (read length, tile geometry).  GPU box, repo root:  bash tools/pmc_geom.sh gpurun_out/pmc_geom "150:fast 100:auto"
set -e
OUT=$1; R=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/pg_run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import microcket_amd as m
from microcket_amd import capi
rl, tn = int(sys.argv[1]), sys.argv[2]
ctx = m.Context("unc", 0.5, 10, False, 8, device=0, tiles={"fast": capi.TILES_FAST, "auto": capi.TILES_AUTO}[tn])
ds = ctx.dataset(20260105, 0, 4000000, 1 << 21, read_len=rl)
for _ in range(3):
    for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
    ctx.sync()
PY
cd /tmp
for spec in $2; do
  rl=${spec%%:*}; tn=${spec##*:}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/a_${rl}_$tn -- python3 /tmp/pg_run.py $rl $tn > /dev/null 2> $R/$OUT/a_${rl}_$tn.err
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/b_${rl}_$tn -- python3 /tmp/pg_run.py $rl $tn > /dev/null 2> $R/$OUT/b_${rl}_$tn.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/c_${rl}_$tn -- python3 /tmp/pg_run.py $rl $tn > /dev/null 2> $R/$OUT/c_${rl}_$tn.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/d_${rl}_$tn -- python3 /tmp/pg_run.py $rl $tn > /dev/null 2> $R/$OUT/d_${rl}_$tn.err
done
cd $R
python3 tools/pmc_geom_summary.py $OUT
