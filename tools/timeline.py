"""Per-block timeline from a rocprofv3 --kernel-trace CSV: python tools/timeline.py <kernel_trace.csv> [n_blocks_to_show]
Prints, for the last blocks of the run, each kernel's start offset and duration relative to the block's k_fast start."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = [i for i, r in enumerate(rows) if "k_fast" in r["Kernel_Name"]]
for a, b in list(zip(idx, idx[1:]))[-show:]:
    t0 = int(rows[a]["Start_Timestamp"])
    print("block: next k_fast starts at +%.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("   +%8.1f us  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
