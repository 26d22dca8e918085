"""Sums the SQ counters of tools/pmc_geom.sh per (pass, read length, geometry) over the k_fast launches."""
import collections, csv, glob, os, sys
out = sys.argv[1]
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "[abcd]_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[2:]
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        n = collections.Counter()
        launches = set()
        for r in csv.DictReader(open(f)):
            if "k_fast" in r["Kernel_Name"]:
                n[r["Counter_Name"]] += float(r["Counter_Value"])
                launches.add(r["Dispatch_Id"])
        for k, v in n.items():
            rows[tag][k] = v
        rows[tag]["launches"] = len(launches)
names = sorted({k for r in rows.values() for k in r})
print("%-12s " % "len_geom" + " ".join("%18s" % k[-18:] for k in names))
for tag, r in rows.items():
    print("%-12s " % tag + " ".join("%18.0f" % r.get(k, 0) for k in names))
