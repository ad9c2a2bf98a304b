"""Developer tool: print the kernel launch sequence of the LAST traced train step from a rocprofv3 kernel-trace CSV
(kernel, grid, LDS bytes, duration in us) so each launch can be mapped to a layer, plus per-kernel totals.
usage: python tools/trace_step.py <..._kernel_trace.csv> [marker-kernel-substring]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("bf16_t", "bf")[:70]


def main(path, marker="adam"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    main_q = step[-1]["Queue_Id"]              # the Adam kernel runs on the caller's stream
    tot = defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for i, r in enumerate(step):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy += d
        k = short(r["Kernel_Name"])
        tot[k][0] += 1
        tot[k][1] += d
        grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"]))) for a in "XYZ")
        q = "M" if r["Queue_Id"] == main_q else "s"
        print(f"{i:4d} {q} {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}us  grid {grid:14s} lds {r.get('LDS_Block_Size', '?'):>6s}  {k}")
    wall = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"\nstep: {len(step)} launches, busy {busy / 1e3:.3f} ms, span {wall / 1e3:.3f} ms")
    for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{d / 1e3:8.3f} ms {n:4d}x {d / n:8.1f}us  {k}")


if __name__ == "__main__":
    main(*sys.argv[1:])
