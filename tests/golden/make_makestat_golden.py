"""Generates tests/golden/makestat_golden.json: seeded log sets (the inputs of the reference's bin/make.stat.pl) and the text that
script prints for them, run here with perl on /root/reference/bin/make.stat.pl (nothing of it is copied).  The .log files of the
`*2pairs.log` kind are exactly what this package's sam2pairs writes.   python tests/golden/make_makestat_golden.py"""
import json
import os
import random
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = "/root/reference/bin/make.stat.pl"


def case(seed, concat, old_flash):
    r = random.Random(seed)
    total = r.randrange(10**6, 2 * 10**8)
    ktrim = int(total * r.uniform(0.8, 0.99))
    uniq = int(ktrim * r.uniform(0.5, 0.95))
    files = {"trim.log": f"Total\t{total}\nPass\t{ktrim}\n" + (f"Total\t{total // 3}\nPass\t{ktrim // 3}\n" if seed % 2 else ""),
             "rmdup.log": f"Total\t{ktrim}\nUniq\t{uniq}\nDup\t{ktrim - uniq - 17}\nDiscard\t17\n"}
    cat = int(uniq * r.uniform(0.2, 0.6))
    unc = uniq - cat
    cut = int(unc * r.uniform(0.9, 1.0))

    def plog(n):
        ks = ["lowMap", "manyHits", "unpaired", "selfCircle", "trans", "cis10K", "cis1K", "cis0"]
        w = [r.random() for _ in ks]
        vals = [int(n * x / sum(w)) for x in w]
        return "".join(f"{k}\t{v}\n" for k, v in zip(ks, vals))
    if concat == "yes":
        if old_flash:
            files["flash.log"] = f"[FLASH] Read combination statistics:\n[FLASH]     Total pairs:      {uniq}\n[FLASH]     Combined pairs:   {cat}\n[FLASH]     Uncombined pairs: {unc}\n"
            if seed % 3:
                files["cut.log"] = f"Total\t{unc}\nPass\t{cut}\n"
        else:
            files["stitch.stat"] = f"Stitched\t{cat}\tUnstitched\t{unc}\tPass\t{cut}\n"
        files["flash2pairs.log"] = plog(int(cat * 0.9))
    files["unc2pairs.log"] = plog(int((cut if concat == "yes" else uniq) * 0.85))
    with tempfile.TemporaryDirectory() as d:
        for fn, txt in files.items():
            open(os.path.join(d, "s." + fn), "w").write(txt)
        out = subprocess.run(["perl", SCRIPT, "s", concat], cwd=d, stdout=subprocess.PIPE, check=True).stdout.decode()
    return {"seed": seed, "concat": concat, "files": files, "expected": out}


def main():
    cases = [case(s, c, o) for s, (c, o) in enumerate([("yes", False), ("no", False), ("yes", True), ("yes", True), ("yes", False), ("no", False)], 1)]
    json.dump({"generator": "tests/golden/make_makestat_golden.py", "cases": cases}, open(os.path.join(HERE, "makestat_golden.json"), "w"), indent=1, sort_keys=True)
    print("cases:", len(cases))


if __name__ == "__main__":
    main()
