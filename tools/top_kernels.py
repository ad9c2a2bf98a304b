"""Developer tool: one table of the train step's kernel families for profiles/ and for bench.py's `top_kernels`.
Joins, for the LAST traced step of a `bench.py` run:
  * a rocprofv3 --kernel-trace CSV                          -> launches, ms per step, stream (caller's / side)
  * the launch census sequence (MMVAE_LAUNCH_STATS + MMVAE_LAUNCH_SEQ=1, runtime.cpp)  -> ALGORITHMIC bytes per launch (what its launcher declared:
    every operand touched once); matched to the trace per family in launch order
  * optionally the per-dispatch PMC table of tools/pmc_per_dispatch.py (separate --pmc passes)  -> measured HBM bytes
and prints JSON: {"build", "frames", "step_ms_traced", "families": [{family, launches, ms, stream, alg_MB, hbm_MB, alg_TBs, traffic_ratio}], "largest": [...]}
usage: python tools/top_kernels.py <kernel_trace.csv> <census.seq> [per_dispatch.txt] > profiles/rNN_top_kernels.json"""
import csv
import importlib
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.getcwd())

# launcher name (census) -> kernel family name (trace)
ALIASES = {"deep2_conv": ["deep2_conv_kernel", "deep2_conv_lite_kernel"], "deep2_conv_f8": ["deep2_conv_f8_kernel"], "wgrad": ["wgrad_kernel"],
           "tail_join_bwd_reduce": ["tail_join_bwd_kernel"], "tail_join_bwd_apply": ["tail_join_bwd_kernel"], "gather_gemm": ["gather_gemm_kernel"],
           "bn_bwd_finalize2": ["bn_bwd_finalize_kernel"], "adam": ["adam_kernel"], "pack_multi": ["pack_multi_kernel"], "convert": ["convert_kernel"],
           "bn_bwd_reduce_nchw": ["plane_reduce_nchw_kernel"], "chan_stats_nchw": ["plane_reduce_nchw_kernel"]}


def family(name):
    return re.sub(r"<.*", "", re.sub(r"\(.*$", "", name).replace("void ", "").replace("mmvae::", ""))


def last_step(rows, key):
    ends = [i for i, r in enumerate(rows) if "adam" in key(r)]
    return rows[ends[-2] + 1:ends[-1] + 1]


def main(trace_csv, seq_path, pmc_path=None, frames=5120):
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
    step = last_step(rows, lambda r: r["Kernel_Name"])
    main_q = step[-1]["Queue_Id"]
    t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
    # census sequence of the last step
    seq = [l.split() for l in open(seq_path) if l.strip()]
    ends = [i for i, (n, _) in enumerate(seq) if n == "adam"]
    seq = seq[ends[-2] + 1:ends[-1] + 1]
    per_fam_bytes = defaultdict(list)
    for n, b in seq:
        for f in ALIASES.get(n, [n + "_kernel"]):
            per_fam_bytes[f].append(float(b))
    # several trace families may share one launcher (deep2 / deep2_lite): hand the launcher's entries out in trace order
    shared = {}
    for n, fams in ALIASES.items():
        if len(fams) > 1:
            for f in fams:
                shared[f] = fams
    pmc = None
    pmc_fam = defaultdict(list)              # per family, in dispatch order (the PMC passes are separate runs of the same step)
    if pmc_path:
        pmc = []
        for l in open(pmc_path):
            if l.startswith("#") or not l.strip():
                continue
            p = l.rstrip().split(None, 3)
            pmc.append((family(p[3]), float(p[1]), float(p[2])))
            pmc_fam[family(p[3])].append((float(p[1]) + float(p[2])) * 1e6)
    pmc_taken = defaultdict(int)
    fams = defaultdict(lambda: dict(launches=0, ms=0.0, side=0, alg=0.0, hbm=0.0, alg_known=0))
    taken = defaultdict(int)
    single = []
    for i, r in enumerate(step):
        f = family(r["Kernel_Name"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        e = fams[f]
        e["launches"] += 1; e["ms"] += d; e["side"] += r["Queue_Id"] != main_q
        pool = shared.get(f, [f])[0]
        lst = per_fam_bytes.get(f) if f not in shared else per_fam_bytes.get(pool)
        key = pool if f in shared else f
        alg = None
        if lst is not None and taken[key] < len(lst):
            alg = lst[taken[key]]; taken[key] += 1
        hb = None
        if pmc and pmc_taken[f] < len(pmc_fam[f]):
            hb = pmc_fam[f][pmc_taken[f]]; pmc_taken[f] += 1
        if alg:
            e["alg"] += alg; e["alg_known"] += 1
        if hb is not None:
            e["hbm"] += hb
        single.append(dict(index=i, kernel=re.sub(r"[(].*$", "", r["Kernel_Name"]).replace("void ", "")[:120], ms=d, stream="side" if r["Queue_Id"] != main_q else "main",
                           alg_MB=None if not alg else alg / 1e6, hbm_MB=None if hb is None else hb / 1e6,
                           alg_TBs=None if not alg else alg / d / 1e9))
    out = []
    for f, e in sorted(fams.items(), key=lambda kv: -kv[1]["ms"]):
        out.append(dict(family=f, launches=e["launches"], ms=round(e["ms"], 4), stream="side" if e["side"] * 2 > e["launches"] else "main",
                        alg_MB=round(e["alg"] / 1e6, 1) if e["alg_known"] == e["launches"] else None,
                        hbm_MB=round(e["hbm"] / 1e6, 1) if pmc else None,
                        alg_TBs=round(e["alg"] / e["ms"] / 1e9, 3) if e["alg_known"] == e["launches"] and e["ms"] > 0 else None,
                        traffic_ratio=round(e["hbm"] / e["alg"], 3) if (pmc and e["alg_known"] == e["launches"] and e["alg"] > 0) else None))
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    res = dict(build=L.build_hash(), frames=int(frames), launches=len(step), step_ms_traced=round((t1 - t0) / 1e6, 4),
               kernel_ms_sum=round(sum(e["ms"] for e in fams.values()), 4),
               hbm_GB=round(sum(a + b for _, a, b in pmc) / 1e3, 3) if pmc else None,
               families=out[:24], largest=sorted(single, key=lambda s: -s["ms"])[:12])
    for s_ in res["largest"]:
        for k in ("ms", "alg_MB", "hbm_MB", "alg_TBs"):
            if s_[k] is not None:
                s_[k] = round(s_[k], 4)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
