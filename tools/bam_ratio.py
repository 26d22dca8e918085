"""How the GPU deflate compares with zlib on the same BGZF payloads (synthetic .sam; raw blocks taken from the level-0 file)."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import util, bamio
import test_gpu_bam as T
import microcket_amd as m
body = util.synth("unc", 31, 60000)
hdr, order = T.header_for(body)
b0, _, n = m.sam_to_bam(hdr + body, sorted=True, level=0)
raws = [r for _, _, r in bamio.bgzf_blocks(b0) if r]
raw = sum(len(r) for r in raws)
print("records", n, "raw BAM bytes", raw, "blocks", len(raws))
for lv in (1, 2):
    try:
        b1, _, _ = m.sam_to_bam(hdr + body, sorted=True, level=lv)
        print(f"GPU level {lv}: {len(b1)} bytes = {len(b1) / raw:.3f} of raw")
    except Exception as e:
        print("level", lv, e)
def z(level, strategy):
    t = 0
    for r in raws:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        t += len(c.compress(r) + c.flush()) + 26
    return t
for name, lv, st in (("zlib 1", 1, zlib.Z_DEFAULT_STRATEGY), ("zlib 6 (samtools default)", 6, zlib.Z_DEFAULT_STRATEGY), ("zlib 1 fixed codes", 1, zlib.Z_FIXED), ("zlib 6 fixed codes", 6, zlib.Z_FIXED), ("zlib huffman only", 6, zlib.Z_HUFFMAN_ONLY)):
    s = z(lv, st)
    print(f"{name}: {s} bytes = {s / raw:.3f} of raw")
