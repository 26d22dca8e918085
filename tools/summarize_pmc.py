"""Summarises rocprofv3 --pmc CSVs (FETCH_SIZE / WRITE_SIZE passes) per kernel.

    python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> [bench_line.json]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports one half of the bytes of a wide coalesced streaming read (16 B per lane), so the read side is doubled
(tools/pmc_fetch_ladder.sh re-checks that factor on this kernel's own newline-scan stream, whose byte count is known);
WRITE_SIZE is exact for 16-byte streaming stores (other widths uncalibrated).  With the bench line of the same run the
last line relates k_fast's traffic to its algorithmic bytes; `--json` style output goes to profiles/*_traffic.json.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
print(f"{'kernel':60s} {'launches':>8s} {'FETCH_SIZE KiB/launch':>22s} {'x2 corrected MB':>16s} {'WRITE_SIZE KiB/launch':>22s} {'MB':>10s} {'HBM traffic MB/launch':>22s}")
fast = None
for k in sorted(set(f) | set(w)):
    fa = sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1)
    wa = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    rd = 2 * fa * 1024 / 1e6
    wr = wa * 1024 / 1e6
    print(f"{k[:60]:60s} {len(f.get(k, [])):8d} {fa:22.1f} {rd:16.1f} {wa:22.1f} {wr:10.1f} {rd + wr:22.1f}")
    if "k_fast" in k:
        fast = (rd + wr) * 1e6
if fast and len(sys.argv) > 3:
    line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    algo = line["roofline"]["algorithmic_bytes_per_launch"]
    print(json.dumps({"kernel": "k_fast", "traffic_bytes_per_launch": fast, "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": fast / algo,
                      "workload": line["config"]["workload"], "block_groups": line["config"]["block_groups"]}))
