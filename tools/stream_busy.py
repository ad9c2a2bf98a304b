"""Developer tool: per-queue (stream) busy time and per-kernel-family totals of the LAST traced train step of a rocprofv3 kernel trace.
usage: python tools/stream_busy.py <..._kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict


def main(path, marker="adam"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    step = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in step)
    main_q = step[-1]["Queue_Id"]
    fam = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    busy = defaultdict(float)
    for r in step:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = re.sub(r"<.*$", "", re.sub(r"\(.*$", "", r["Kernel_Name"])).replace("void ", "").replace("mmvae::", "")
        q = "main" if r["Queue_Id"] == main_q else "side"
        fam[q][k][0] += 1
        fam[q][k][1] += d
        busy[q] += d
    print(f"span {(t1 - t0) / 1e3:.1f} us; busy main {busy['main']:.1f} us, side {busy['side']:.1f} us; {len(step)} launches")
    for q in ("main", "side"):
        print(f"--- {q}")
        for k, (n, d) in sorted(fam[q].items(), key=lambda kv: -kv[1][1])[:18]:
            print(f"{d:9.1f} us {n:4d}x  {k}")


if __name__ == "__main__":
    main(*sys.argv[1:])
