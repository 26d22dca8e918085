"""Hand-written edge-case SAM inputs (one record group per branch of the reference's classifiers).

Writes edge_unc.sam, edge_flash.sam and edge_aba.sam next to this script.  Deterministic, no randomness.
Branches follow SURVEY.md 3.4 / 3.5; comments name the reference lines each group exercises.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SEQ = "ACGT" * 40


def rec(q, flag, chrom, pos, mapq, cigar, extra=""):
    n = 0
    num = ""
    for c in cigar:
        if c.isdigit():
            num += c
        else:
            if c in "MIS":
                n += int(num)
            num = ""
    seq = (SEQ * 4)[:max(n, 1)]
    return f"{q}\t{flag}\t{chrom}\t{pos}\t{mapq}\t{cigar}\t=\t{pos}\t0\t{seq}\t{'F' * len(seq)}\tNM:i:0{extra}\n"


def unc():
    L = []
    a = L.append
    a("@HD\tVN:1.6\tSO:queryname\n")
    a("@SQ\tSN:chr1\tLN:248956422\n")
    a("@SQ\tSN:chr10\tLN:133797422\n")
    a("@PG\tID:bwa\tPN:bwa\tVN:0.7.17-r1188\tCL:bwa mem -5SP ref.fa r1.fq r2.fq\n")
    g = 0

    def grp(*recs):
        nonlocal g
        g += 1
        for r in recs:
            a(rec(f"read{g:04d}", *r))

    # ---- category 0 (1+1), 5'-end rule (unc2pairs.h:133-145), ordering + strands (:315-348)
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 50000, 60, "150M"))            # + +  cis10K
    grp((81, "chr1", 1000, 60, "150M"), (145, "chr1", 3000, 60, "150M"))             # - -  cis1K, right ends
    grp((65, "chr1", 90000, 60, "150M"), (145, "chr1", 1000, 60, "150M"))            # swapped order
    grp((65, "chr10", 500, 60, "150M"), (129, "chr2", 500, 60, "150M"))              # trans, chr10 < chr2 bytewise
    grp((65, "chr2", 500, 60, "150M"), (129, "chr10", 500, 60, "150M"))              # trans, swapped
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 1010, 60, "150M"))             # dist 10: selfCircle
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 1011, 60, "150M"))             # dist 11: cis0
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 1000, 60, "150M"))             # equal positions: else-branch, dist 0
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 1999, 60, "150M"))             # dist 999 cis0
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 2000, 60, "150M"))             # dist 1000 cis1K
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 10999, 60, "150M"))            # dist 9999 cis1K
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 11000, 60, "150M"))            # dist 10000 cis10K
    # ---- integrity (pairutil.h:180-188): clip > 20 counts, <= 20 does not
    grp((65, "chr1", 1000, 60, "75M75S"), (129, "chr1", 9000, 60, "150M"))           # 75 >= 150*0.5 passes
    grp((65, "chr1", 1000, 60, "74M76S"), (129, "chr1", 9000, 60, "150M"))           # 74 < 75 lowMap
    grp((65, "chr1", 1000, 60, "20S130M"), (129, "chr1", 9000, 60, "130M20S"))       # clips of 20 ignored
    grp((65, "chr1", 1000, 60, "21S30M99S"), (129, "chr1", 9000, 60, "150M"))        # 30 < 150*.5 lowMap
    grp((65, "chr1", 1000, 60, "120M30S"), (129, "chr1", 9000, 60, "30S120M"))       # float32 edge at ratio 0.8
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 9000, 60, "60M90H"))           # second read lowMap (H clip)
    # ---- indels (pairutil.h:96-104)
    grp((65, "chr1", 1000, 60, "70M2I78M"), (145, "chr1", 7000, 60, "70M5D80M"))
    # ---- N-spliced reads: pairable test (unc2pairs.h:146-190)
    grp((65, "chr1", 1000, 60, "50M500N100M"), (145, "chr1", 2200, 60, "150M"))      # s1 '+' spliced, pairable (2349-1550 <= 1000)
    grp((65, "chr1", 1000, 60, "50M500N100M"), (145, "chr1", 2402, 60, "150M"))      # 2551-1550 = 1001 unpaired
    grp((65, "chr1", 1000, 60, "50M500N100M"), (145, "chr1", 2401, 60, "150M"))      # exactly 1000
    grp((65, "chr1", 1000, 60, "50M500N100M"), (129, "chr1", 2200, 60, "150M"))      # wrong strand: unpaired
    grp((81, "chr1", 5000, 60, "50M500N100M"), (129, "chr1", 4200, 60, "150M"))      # s1 '-' spliced
    grp((65, "chr1", 1000, 60, "150M"), (145, "chr1", 1500, 60, "50M500N100M"))      # s2 spliced, s1 '+'
    grp((81, "chr1", 5000, 60, "150M"), (129, "chr1", 3900, 60, "50M500N100M"))      # s2 spliced, s1 '-'
    grp((65, "chr1", 1000, 60, "50M500N100M"), (145, "chr2", 2200, 60, "150M"))      # other chr: unpaired
    grp((65, "chr1", 1000, 60, "50M50N50M50N50M"), (129, "chr1", 9000, 60, "150M"))  # 3+1 segments: manyHits
    grp((65, "chr1", 1000, 60, "50M500N100M"), (129, "chr1", 9000, 60, "50M500N100M"))  # 2+2 segments: manyHits
    # ---- category 1 (1+2) and 2 (2+1) (unc2pairs.h:84-121,191-308)
    grp((65, "chr1", 1000, 60, "150M"), (145, "chr1", 1300, 60, "100M50S"), (2177, "chr5", 777, 60, "100H50M"))
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr5", 777, 60, "100M50S"), (2193, "chr1", 1300, 60, "100H50M"))   # pairs with the 2nd
    grp((81, "chr1", 5000, 60, "150M"), (129, "chr1", 4300, 60, "100M50S"), (2177, "chr5", 777, 60, "100H50M"))
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 1300, 60, "100M50S"), (2177, "chr5", 777, 60, "100H50M"))   # nobody pairs: unpaired
    grp((65, "chr1", 1000, 60, "100M50S"), (2113, "chr7", 4242, 60, "100H50M"), (145, "chr1", 1300, 60, "150M"))  # 2+1
    grp((65, "chr7", 4242, 60, "100M50S"), (2129, "chr1", 5000, 60, "100H50M"), (129, "chr1", 4300, 60, "150M"))  # 2+1, pairs with 2nd
    grp((65, "chr1", 1000, 60, "60S90M"), (2113, "chr7", 4242, 60, "60M90H"), (145, "chr1", 1300, 60, "150M"))    # leftClip > rightClip: right end used
    grp((65, "chr1", 1000, 60, "150M"), (145, "chr1", 1300, 60, "50M100S"), (2177, "chr5", 777, 60, "150H20M"))   # lowMap in 2-seg check
    # quirk Q3 (pairutil.h:200): the second record's right clip is tested through the FIRST record's
    grp((65, "chr1", 1000, 60, "150M"), (145, "chr1", 1300, 60, "100M10S"), (2177, "chr5", 777, 60, "40M110S"))
    grp((65, "chr1", 1000, 60, "150M"), (145, "chr1", 1300, 60, "100M30S"), (2177, "chr5", 777, 60, "40M110S"))
    grp((65, "chr1", 1000, 60, "150M"), (145, "chr1", 1300, 60, "100M50S"), (2177, "chr5", 777, 60, "50M10N50M"))  # segCnt != 1: manyHits
    # ---- silent drops (unc2pairs.h:52-59)
    grp((65, "chr1", 1000, 60, "150M"))                                               # R1 only
    grp((129, "chr1", 1000, 60, "150M"), (129, "chr1", 3000, 60, "150M"))             # R2 only
    grp((65, "chr1", 1000, 60, "100M50S"), (2113, "chr2", 1, 60, "100H50M"), (129, "chr1", 3000, 60, "100M50S"), (2177, "chr3", 9, 60, "100H50M"))  # 2+2
    # ---- records without the 64/128 bits are ignored but travel to the .sam (unc2pairs.h:45-48,351-356)
    grp((65, "chr1", 1000, 60, "150M"), (1, "chr9", 5, 60, "150M"), (129, "chr1", 30000, 60, "150M"))
    grp((193, "chr1", 1000, 60, "150M"), (129, "chr1", 30000, 60, "150M"))            # 64|128: counts as R1
    # ---- per-line filter (pairutil.h:157-161): low MAPQ, secondary, QC-fail, duplicate
    grp((65, "chr1", 1000, 60, "150M"), (321, "chr3", 5, 60, "150M"), (129, "chr1", 30000, 60, "150M"))     # 0x100 inside the group
    grp((65, "chr1", 1000, 60, "150M"), (577, "chr3", 5, 60, "150M"), (129, "chr1", 30000, 60, "150M"))     # 0x200
    grp((65, "chr1", 1000, 60, "150M"), (1089, "chr3", 5, 60, "150M"), (129, "chr1", 30000, 60, "150M"))    # 0x400
    grp((65, "chr1", 1000, 60, "100M50S"), (2113, "chr3", 5, 5, "100H50M"), (129, "chr1", 30000, 60, "150M"))  # supplementary filtered by MAPQ
    grp((65, "chr1", 1000, 9, "150M"), (129, "chr1", 30000, 60, "150M"))              # R1 filtered: R2 alone -> silent drop
    grp((65, "chr1", 1000, 10, "150M"), (129, "chr1", 30000, 10, "150M"))             # MAPQ == min passes
    grp((65, "chr1", 1000, 3, "150M"), (129, "chr1", 30000, 3, "150M"))               # whole group filtered
    grp((65, "chr1", 1000, 255, "150M"), (129, "chr1", 30000, 255, "150M"))           # STAR-style MAPQ
    # ---- long names (first six fields beyond 128 bytes: generic parser path)
    long_q = "Q" * 150
    a(rec(long_q, 65, "chr1", 1000, 60, "150M"))
    a(rec(long_q, 129, "chr1_KI270706v1_random_with_a_very_long_contig_name_" + "x" * 80, 30000, 60, "150M"))
    # ---- large coordinates (POS >= 2^31 is out of contract: the reference reads POS into an int, unc2pairs.h:36)
    grp((65, "chr1", 2147483000, 60, "150M"), (129, "chr1", 1, 60, "150M"))
    # ---- the last group of the input is never classified (quirk Q1)
    grp((65, "chr1", 1000, 60, "150M"), (129, "chr1", 50000, 60, "150M"))
    return "".join(L)


def aba():
    """Input that is NOT name-grouped line by line: a filtered line with another name sits inside a group (A kept, B filtered,
    A kept).  The reference filters before it compares names (pairutil.h:157-163), so A's two lines are ONE group; a block or
    shard cut between A and B must not split it (the host cut of mkt_host.h works on surviving lines)."""
    L = []
    a = L.append
    for g in range(40):
        q = f"aba{g:03d}"
        a(rec(q, 65, "chr1", 1000 + 17 * g, 60, "150M"))
        for k in range(g % 3 + 1):                                                    # 1..3 filtered strangers in between
            a(rec(f"stranger{g:03d}_{k}", 129 if k % 2 else 65, "chr2", 5000 + g, 0 if k != 1 else 60, "150M") if k != 1
              else rec(f"stranger{g:03d}_{k}", 129 + 256, "chr2", 5000 + g, 60, "150M"))    # MAPQ 0 / secondary (0x100): both filtered
        a(rec(q, 129, "chr1", 90000 + 31 * g, 60, "150M"))
        if g % 5 == 0:                                                                # a plain group in between
            a(rec(f"plain{g:03d}", 65, "chr3", 100 + g, 60, "150M"))
            a(rec(f"plain{g:03d}", 129, "chr3", 70000 + g, 60, "150M"))
    return "".join(L)


def flash():
    L = []
    a = L.append
    a("@HD\tVN:1.6\tSO:queryname\n")
    g = 0

    def grp(*recs):
        nonlocal g
        g += 1
        for r in recs:
            a(rec(f"frag{g:04d}", *r))

    # size 1 (flash2pairs.h:26-68)
    grp((0, "chr1", 1000, 60, "250M"))                    # dist 249: cis0, strands printed + -
    grp((16, "chr1", 1000, 60, "100M900N100M"))           # 2 segments: right[1], dist 1099 cis1K
    grp((0, "chr1", 1000, 60, "100M9800N100M"))           # cis10K (dist 9999 -> cis1K? 1000+100+9800+100-1-1000 = 9999)
    grp((0, "chr1", 1000, 60, "100M9801N100M"))           # dist 10000
    grp((0, "chr1", 1000, 60, "50M50N50M50N50M"))         # 3 segments: manyHits (before the integrity test)
    grp((0, "chr1", 1000, 60, "100M150S"))                # 100 < 250*0.5: lowMap
    grp((0, "chr1", 1000, 60, "125M125S"))                # passes at 0.5
    grp((0, "chr1", 1000, 60, "20S210M20S"))              # small clips ignored
    # size 2 (flash2pairs.h:69-149)
    grp((0, "chr1", 1000, 60, "120M130S"), (2048, "chr1", 90000, 60, "120H130M"))
    grp((16, "chr1", 1000, 60, "130S120M"), (2064, "chr1", 90000, 60, "130M120H"))   # leftClip > rightClip -> right end
    grp((0, "chr2", 1000, 60, "120M130S"), (2048, "chr10", 90000, 60, "120H130M"))   # trans swapped
    grp((0, "chr1", 1000, 60, "120M130S"), (2048, "chr1", 1005, 60, "120H130M"))     # selfCircle? positions: 1000 vs right end
    grp((0, "chr1", 1000, 60, "120M130S"), (2048, "chr1", 1000, 60, "130M120H"))     # equal positions
    grp((0, "chr1", 1000, 60, "120M130S"), (2048, "chr1", 1010, 60, "130M120H"))     # dist 10 selfCircle
    grp((0, "chr1", 1000, 60, "120M130S"), (2048, "chr1", 1011, 60, "130M120H"))     # dist 11
    grp((0, "chr1", 1000, 60, "60M10N60M130S"), (2048, "chr1", 90000, 60, "120H130M"))  # segCnt != 1: manyHits
    grp((0, "chr1", 1000, 60, "40M210S"), (2048, "chr1", 90000, 60, "210H40M"))      # 80 < 250*.5: lowMap
    # quirk Q3
    grp((0, "chr1", 1000, 60, "100M10S"), (2048, "chr1", 90000, 60, "40M110S"))
    grp((0, "chr1", 1000, 60, "100M30S"), (2048, "chr1", 90000, 60, "40M110S"))
    # size >= 3 (flash2pairs.h:150-153)
    grp((0, "chr1", 1000, 60, "80M170S"), (2048, "chr2", 5, 60, "80H80M90H"), (2048, "chr3", 7, 60, "160H90M"))
    # filtered lines change the group size
    grp((0, "chr1", 1000, 60, "80M170S"), (2048, "chr2", 5, 3, "80H80M90H"), (2048, "chr3", 7, 60, "160H90M"))
    grp((0, "chr1", 1000, 60, "250M"), (256, "chr9", 5, 60, "250M"))
    grp((0, "chr1", 1000, 0, "250M"))
    # last group (quirk Q1)
    grp((0, "chr1", 1000, 60, "250M"))
    return "".join(L)


if __name__ == "__main__":
    open(os.path.join(HERE, "edge_unc.sam"), "w").write(unc())
    open(os.path.join(HERE, "edge_flash.sam"), "w").write(flash())
    open(os.path.join(HERE, "edge_aba.sam"), "w").write(aba())
    print("written")
