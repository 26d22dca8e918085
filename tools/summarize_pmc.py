"""Summarises rocprofv3 --pmc CSVs (FETCH_SIZE / WRITE_SIZE passes) per kernel.

    python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv>

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are
in KiB; on gfx950 FETCH_SIZE reports one half of the bytes of a wide coalesced streaming read, so
the read side is doubled; WRITE_SIZE is exact for 16-byte streaming stores (other widths uncalibrated).
"""
import collections
import csv
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
for k in sorted(set(f) | set(w)):
    fa = sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1)
    wa = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    rd = 2 * fa * 1024 / 1e6
    wr = wa * 1024 / 1e6
    print(f"{k[:60]:60s} {len(f.get(k, [])):8d} {fa:22.1f} {rd:16.1f} {wa:22.1f} {wr:10.1f} {rd + wr:22.1f}")
