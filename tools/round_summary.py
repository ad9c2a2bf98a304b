"""Developer tool: profiles/rNN_summary.md tables (kernel table per step, per-layer table of the channel-heavy convs) from the files
tools/profile_round.sh produced and that were copied into profiles/.   usage: python tools/round_summary.py r02 7 > tables.md"""
import csv
import re
import sys


def strip(t):
    return t.split("\n", 1)[1] if t.startswith("/opt") else t


def cols(txt):
    d = {}
    for l in txt.splitlines():
        p = l.split()
        if len(p) >= 8 and p[1] in ("fwd", "dgrad"):
            d[(p[0], p[1])] = (float(p[2]), float(p[3]))
    return d


def main(tag="r02", steps="7"):
    steps = int(steps)
    rows = list(csv.DictReader(open(f"profiles/{tag}_kernel_stats_c2_bf16_N5120.csv")))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    a = cols(strip(open(f"profiles/{tag}_deep_layers.txt").read()))
    b = cols(strip(open(f"profiles/{tag}_deep_layers_before.txt").read()))
    print("| layer | op | GFLOP | round-1 kernels us | deep2_conv us | TFLOP/s | % of 2.5 PF |\n|---|---|---|---|---|---|---|")
    for k, (us, gf) in a.items():
        print(f"| {k[0]} | {k[1]} | {gf:.2f} | {b.get(k, (0, 0))[0]:.1f} | {us:.1f} | {gf / us * 1e3:.0f} | {gf / us * 1e3 / 2500 * 100:.1f} |")
    print(f"| sum | | | {sum(v[0] for v in b.values()):.0f} | {sum(v[0] for v in a.values()):.0f} | | |\n")
    print("| ms/step | calls/step | avg us | % | kernel |\n|---|---|---|---|---|")
    for r in rows[:40]:
        n = re.sub(r"\(.*$", "", r["Name"]).replace("void ", "").replace("mmvae::", "")
        calls, t = int(r["Calls"]), int(r["TotalDurationNs"])
        print(f"| {t / steps / 1e6:.3f} | {calls / steps:.0f} | {t / calls / 1e3:.1f} | {100 * t / tot:.1f} | `{n[:90]}` |")


if __name__ == "__main__":
    main(*sys.argv[1:])
