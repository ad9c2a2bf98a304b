"""Developer tool: side-by-side per-launch durations of sweep files written by tools/sweep_env.sh.
usage: python tools/sweep_diff.py VAR v1 v2 ... [--filter substr] [--min us]"""
import re
import sys


def main(argv):
    var, vals, filt, minus = argv[0], [], "", 40.0
    i = 1
    while i < len(argv):
        if argv[i] == "--filter":
            filt = argv[i + 1]; i += 2
        elif argv[i] == "--min":
            minus = float(argv[i + 1]); i += 2
        else:
            vals.append(argv[i]); i += 1
    rows = {}
    for v in vals:
        for line in open(f"gpurun_out/sweep_{var}_{v}.txt"):
            m = re.match(r"\s*(\d+)\s+[\d.]+\s+([\d.]+)us\s+grid (\S+)\s+lds\s+\S+\s+(.*)", line)
            if m and filt in m.group(4):
                rows.setdefault(int(m.group(1)), {})[v] = (float(m.group(2)), m.group(3), m.group(4).replace("mmvae::", ""))
    tot = {v: 0.0 for v in vals}
    for i, d in sorted(rows.items()):
        for v in vals:
            tot[v] += d.get(v, (0,))[0]
        if max(x[0] for x in d.values()) < minus:
            continue
        any_ = next(iter(d.values()))
        print(f"{i:4d} " + " ".join(f"{d[v][0]:8.1f}" if v in d else "    -   " for v in vals) + f"  {any_[1]:12s} {any_[2][:70]}")
    print("sum  " + " ".join(f"{tot[v]:8.1f}" for v in vals))


if __name__ == "__main__":
    main(sys.argv[1:])
