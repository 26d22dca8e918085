#!/bin/bash
# Calibration: FETCH_SIZE of k_fast when every tile stops after phase k (stamp build): k=1 is the pure newline-scan
# stream whose byte count is known (window bytes = (TILE+HB+HF)/TILE x block bytes).
# usage (GPU box, repo root): bash tools/pmc_fetch_ladder.sh gpurun_out/fetch_ladder
set -e
OUT=$1; R=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/pl_run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import microcket_amd as m
ctx = m.Context("unc", 0.5, 10, False, 8, device=0, tiles=m.TILES_FAST)
ds = ctx.dataset(20260105, 0, 2000000, 1 << 20)
print("block bytes", [n for (p, n, g) in ds.blocks], file=sys.stderr)
for _ in range(2):
    for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
    try: ctx.sync()
    except Exception: pass
PY
cd /tmp
for k in 1 3 0; do
  MKT_DEBUG_STOP=$k MKT_NO_STAMPS=1 MKT_LIB=$R/microcket_amd/libmkt_hip_stamps.so rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/stop$k -- python3 /tmp/pl_run.py > /dev/null 2> $R/$OUT/stop$k.err
  python3 - <<PY
import csv, glob
for f in glob.glob("$R/$OUT/stop$k/**/*counter_collection.csv", recursive=True):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_fast" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    print("stop $k: k_fast launches", len(v), "FETCH_SIZE KiB/launch", [round(x) for x in v])
PY
  grep "block bytes" $R/$OUT/stop$k.err || true
done
