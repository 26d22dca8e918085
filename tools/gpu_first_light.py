"""First-light script for a GPU box: streams synthetic sets through the C ABI and compares with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
import microcket_amd as m

bad = 0
print("devices", m.device_count(), flush=True)
for prof, seed, n in (("stress", 3, 3000), ("unc", 1, 3000), ("flash", 2, 3000)):
    text = util.synth(prof, seed, n)
    for mode in ("unc", "flash"):
        for (T, ratio, mapq, sam) in ((4, 0.5, 10, True), (2, 0.8, 30, True), (3, 0.5, 10, False)):
            po, so, lo, st = util.oracle_run(text, mode, T, ratio, mapq, sam)
            for tiles, ordered in ((m.TILES_SMALL, True), (m.TILES_FAST, True), (m.TILES_FAST, False)):
                for block in (0, 1 << 16):
                    t0 = time.time()
                    try:
                        with m.Context(mode, ratio, mapq, sam, T, 0, block, tiles, ordered) as c:
                            p, s, stats, log = c.run_bytes(text, chunk=100000)
                    except Exception as e:
                        print("EXC", prof, mode, T, ratio, mapq, sam, tiles, block, e, flush=True)
                        bad += 1
                        continue
                    if ordered:
                        ok = (p == po and s == so and log == lo and stats.groups == st.groups)
                    else:
                        ok = (util.canon(p) == util.canon(po) and util.canon(s) == util.canon(so) and log == lo and stats.groups == st.groups)
                    if not ok:
                        bad += 1
                        print("MISMATCH", prof, mode, T, ratio, mapq, sam, "tiles", tiles, "block", block, len(p), len(po), len(s), len(so), log == lo, stats.groups, st.groups,
                              util.canon(p) == util.canon(po), flush=True)
                    else:
                        print("ok", prof, mode, T, ratio, mapq, sam, tiles, ordered, block, "%.2fs" % (time.time() - t0), flush=True)
print("BAD", bad)
sys.exit(1 if bad else 0)
