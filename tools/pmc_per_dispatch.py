"""Developer tool: per-dispatch HBM traffic of the LAST train step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).
  fetched MB = 2 * FETCH_SIZE * 1024 / 1e6   (gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes; unit KB)
  written MB = WRITE_SIZE * 1024 / 1e6
Prints one line per dispatch (index, fetched MB, written MB, kernel) and per-family totals; the two passes must see the same dispatch sequence.
usage: python tools/pmc_per_dispatch.py <fetch counter_collection.csv> <write counter_collection.csv>"""
import csv
import re
import sys
from collections import OrderedDict, defaultdict


def per_dispatch(path, counter):
    d = OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])
        name, val = d.get(k, (r["Kernel_Name"], 0.0))
        d[k] = (name, val + float(r["Counter_Value"]))
    return [d[k] for k in sorted(d)]


def last_step(rows):
    ends = [i for i, (n, _) in enumerate(rows) if "adam" in n]
    return rows[ends[-2] + 1:ends[-1] + 1]


def family(name):
    return re.sub(r"<.*", "", re.sub(r"\(.*$", "", name).replace("void ", "").replace("mmvae::", ""))


def main(fetch_csv, write_csv):
    f = last_step(per_dispatch(fetch_csv, "FETCH_SIZE"))
    w = last_step(per_dispatch(write_csv, "WRITE_SIZE"))
    if len(f) != len(w) or any(family(a[0]) != family(b[0]) for a, b in zip(f, w)):
        raise SystemExit(f"the two passes saw different dispatch sequences ({len(f)} vs {len(w)})")
    fam = defaultdict(lambda: [0, 0.0, 0.0])
    tf = tw = 0.0
    print("# idx  fetched_MB  written_MB  kernel")
    for i, ((n, fv), (_, wv)) in enumerate(zip(f, w)):
        fb, wb = 2 * fv * 1024 / 1e6, wv * 1024 / 1e6
        tf += fb; tw += wb
        e = fam[family(n)]
        e[0] += 1; e[1] += fb; e[2] += wb
        print(f"{i:4d} {fb:10.1f} {wb:10.1f}  {re.sub(r'[(].*$', '', n).replace('void ', '')[:110]}")
    print("# family totals: launches  fetched_MB  written_MB")
    for k, (c, fb, wb) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"# {c:4d} {fb:10.1f} {wb:10.1f}  {k}")
    print(f"# step: fetched {tf / 1e3:.2f} GB + written {tw / 1e3:.2f} GB = {(tf + tw) / 1e3:.2f} GB over {len(f)} dispatches")


if __name__ == "__main__":
    main(*sys.argv[1:])
