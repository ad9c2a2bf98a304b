"""Developer tool: HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as the
MI355X guide prescribes) of `bench.py --steps 1 --warmup 1` -> rows of profiles/rNN_pmc_hbm.csv, keyed by the build hash so that
bench.py only quotes them for the build they were measured on.
  fetch_bytes = 2 * FETCH_SIZE * 1024   (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes; unit KB)
  write_bytes = WRITE_SIZE * 1024
keys: uplayer5.conv2.fwd   the last convT4_stream_kernel (else patch_conv_kernel) dispatch before the tail forward kernel
      (decoder.uplayer5.0.conv2 forward)
      uplayer5.join_bwd   the last up-block's backward in one pass (join_bwd_stream_kernel); builds without it:
      uplayer5.join_bwd_apply = the apply pass of the join backward (tail_apply_mfma_kernel, else the second tail_join_bwd_kernel dispatch)
      wgrad2.<layer>   a wgrad2_kernel dispatch + its wgrad_reduce_kernel, for the five layers named in pick()
      wgrad_stream.encoder.layer1.pair   the last wgrad_stream_kernel dispatch of the step + its two reduces
      __step__   every dispatch of the last train step (between two Adam kernels)
usage: python tools/pmc_hbm_csv.py <fetch counter_collection.csv> <write counter_collection.csv> <frames> > profiles/rNN_pmc_hbm.csv"""
import csv
import importlib
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.getcwd())


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


def pick(step):
    names = [n for n, _ in step]
    out = {"__step__": sum(v for _, v in step)}
    j = next(i for i, n in enumerate(names) if "tail_join_fwd_kernel" in n or "tail_fwd_stream_kernel" in n or "up5_tail_fwd_kernel" in n)
    conv = "convT4_stream_kernel" if any("convT4_stream_kernel" in n for n in names[:j]) else "patch_conv_kernel"
    i = max(k for k in range(j) if conv in names[k])
    out["uplayer5.conv2.fwd"] = step[i][1]
    jb = [k for k, n in enumerate(names) if "join_bwd_stream_kernel" in n]
    ta = [k for k, n in enumerate(names) if "tail_apply_mfma_kernel" in n]
    tb = [k for k, n in enumerate(names) if "tail_join_bwd_kernel" in n]
    # the wgrad2 family (bench.py's weight-gradient table): the step issues its 12 wgrad2 launches in a fixed order that starts with
    # decoder.uplayer3 (conv2, upsample, conv1) and ends with encoder.layer2 (conv2, downsample, conv1) -- the candidates for the family's
    # slowest instance are the first two and the last three.  Each launch's partial-image reduce is the next wgrad_reduce dispatch.
    w2 = [k for k, n in enumerate(names) if "wgrad2_kernel" in n]
    if len(w2) in (11, 12):     # (12 since encoder.layer3's 1x1 stride-2 shortcut runs on it too)
        for idx, layer in ((0, "decoder.uplayer3.0.conv2"), (1, "decoder.uplayer3.0.upsample.0"), (-3, "encoder.layer2.0.conv2"),
                           (-2, "encoder.layer2.0.downsample.0"), (-1, "encoder.layer2.0.conv1")):
            k = w2[idx]
            red = next((q for q in range(k + 1, len(names)) if "wgrad_reduce_kernel" in names[q]), None)
            out["wgrad2." + layer] = step[k][1] + (step[red][1] if red is not None else 0.0)
    # the wgrad_stream family's largest instance: encoder.layer1's conv1 + shortcut pair, the last wgrad_stream_kernel dispatch of the step
    # (+ the two wgrad_reduce dispatches that follow it)
    ws = [k for k, n in enumerate(names) if "wgrad_stream_kernel" in n]
    if ws:
        k = ws[-1]
        reds = [q for q in range(k + 1, len(names)) if "wgrad_reduce_kernel" in names[q]][:2]
        out["wgrad_stream.encoder.layer1.pair"] = step[k][1] + sum(step[q][1] for q in reds)
    if jb:
        out["uplayer5.join_bwd"] = step[jb[0]][1]
    elif ta or len(tb) > 1:
        out["uplayer5.join_bwd_apply"] = step[ta[0] if ta else tb[1]][1]
    return out


def main(fetch_csv, write_csv, frames):
    build = importlib.import_module("moving-mnist-vae_amd._lib").build_hash()
    f = pick(last_step(per_dispatch(fetch_csv, "FETCH_SIZE")))
    w = pick(last_step(per_dispatch(write_csv, "WRITE_SIZE")))
    print("build,frames,key,fetch_bytes,write_bytes")
    for k in f:
        if k in w:
            print(f"{build},{int(frames)},{k},{2 * f[k] * 1024:.0f},{w[k] * 1024:.0f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
