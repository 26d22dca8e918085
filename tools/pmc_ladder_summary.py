"""Per-stop SQ counters of k_fast from tools/pmc_ladder.sh output: python tools/pmc_ladder_summary.py <dir>"""
import collections, csv, glob, os, sys
d = sys.argv[1]
names = None
for stop in (1, 2, 9, 3, 4, 6, 7, 0):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(os.path.join(d, f"stop{stop}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fast" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if not agg: continue
    if names is None:
        names = sorted(agg); print("stop " + " ".join(f"{x[3:]:>16s}" for x in names))
    print(f"{stop:4d} " + " ".join(f"{agg[x] / n[x]:16.0f}" for x in names), f" launches {n[names[0]]}")
