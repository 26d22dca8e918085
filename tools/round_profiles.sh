set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $O/bench_100M.json 2> $O/bench_100M.err
echo bench done; cut -c1-200 $O/bench_100M.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
echo stats done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --pairs 16000000 --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --pairs 16000000 --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
echo pmc done
cd $R
python3 tools/summarize_pmc.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) > $O/pmc_summary.txt
cat $O/pmc_summary.txt
cat $(ls $O/stats/*/*kernel_stats.csv | head -1)
