"""Developer tool: aggregate a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch of each counter).
usage: python tools/pmc_summary.py <..._counter_collection.csv> [name-substring]"""
import csv
import re
import sys
from collections import defaultdict


def main(path, filt=""):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("mmvae::", "").replace("bf16_t", "bf")[:60]
        if filt and filt not in k:
            continue
        key = (k, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key].add(r["Dispatch_Id"])
    names = sorted({c for v in acc.values() for c in v})
    print("kernel | grid | n | " + " | ".join(names))
    for key, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        n = len(cnt[key])
        print(f"{key[0]} | {key[1]} | {n} | " + " | ".join(f"{v.get(c, 0) / n:.3g}" for c in names))


if __name__ == "__main__":
    main(*sys.argv[1:])
