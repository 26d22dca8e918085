#!/bin/bash
# Round-end evidence (GPU box, from the repo root):  bash tools/round_profiles.sh <tag>
# Writes gpurun_out/<tag>/: the default bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
# passes on the SAME 100 M-pair workload (separate --pmc runs, as MI355X_MICROARCH.md prescribes), the phase ladder and the
# end-to-end CLI timings.
set -e
TAG=${1:-round}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 python3 $R/bench.py > $O/bench_100M.json 2> $O/bench_100M.err
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "stats done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs > $O/pmc_write.json 2> $O/pmc_write.err
echo "pmc done"
cd $R
python3 tools/summarize_pmc.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/pmc_fetch.json > $O/pmc_summary.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
timeout -k 10 200 python3 tools/phase_ladder.py libmkt_hip_stamps.so > $O/phase_ladder.txt 2>&1 || true
timeout -k 10 300 python3 tools/cli_bench.py 8000000 > $O/cli.txt 2>&1 || true
cat $O/pmc_summary.txt; head -12 $O/kernel_stats.csv; cat $O/phase_ladder.txt; cat $O/cli.txt
