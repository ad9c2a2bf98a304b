"""Developer tool: the isolated roofline launches of `python bench.py` in a rocprofv3 kernel trace of that very command.
bench.py times 20 isolated launches (after warm-up ones) of the roofline kernel (convT4_stream_kernel<32, ..> at decoder.uplayer5.0.conv2's
shape) and 10 of the largest launch (tail_apply_mfma_kernel) with events on the launch stream AFTER its timed steps; this prints the
profiler's durations of exactly those dispatches (the last 20 / 10 of each kernel in the trace) next to the in-step ones.
usage: python tools/roofline_trace.py s_kernel_trace.csv"""
import csv
import sys


def main(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    for key, reps in (("convT4_stream_kernel<32", 20), ("tail_apply_mfma_kernel", 10)):
        # the isolated launches end the longest run of back-to-back dispatches of the kernel (warm-up launches first, nothing else in between)
        i, best = 0, None
        while i < len(rows):
            if key in names[i]:
                j = i
                while j < len(rows) and names[j] == names[i]:
                    j += 1
                if j - i >= reps + 2 and (best is None or j - i > best[1] - best[0]):
                    best = (i, j)
                i = j
            else:
                i += 1
        if best is None:
            print(f"{key}: no run of >= {reps + 2} back-to-back dispatches found")
            continue
        iso = dur[best[1] - reps:best[1]]
        longest = max(d for n, d in zip(names, dur) if key in n)
        step = [d for k, (n, d) in enumerate(zip(names, dur)) if key in n and not best[0] <= k < best[1] and d > 0.5 * longest]
        print(f"{key}: isolated (the {reps} timed launches): avg {sum(iso) / len(iso):.1f} us, min {min(iso):.1f}, max {max(iso):.1f}; "
              f"full-size launches inside the train steps ({len(step)}): avg {sum(step) / max(len(step), 1):.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
