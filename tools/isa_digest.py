"""Developer tool: digest of one kernel's ISA around its MFMA loop (loads, waits, LDS reads, branches; runs of MFMAs collapsed).
usage: python tools/isa_digest.py file.s mangled-name-substring"""
import sys


def main(path, name):
    s = open(path).read()
    i = s.index(name + ":")
    j = s.index("s_endpgm", i)
    body = s[i:j].split("\n")
    mf = [k for k, l in enumerate(body) if "v_mfma" in l]
    print(len(body), "lines; mfma", len(mf), mf[0], mf[-1])
    out = []
    for k in range(max(0, mf[0] - 40), mf[-1] + 8):
        l = body[k]
        if any(x in l for x in ("global_load", "buffer_load", "s_waitcnt", "ds_read", "ds_write", "s_cbranch", "LBB", "sched", "s_barrier", "scratch_")):
            out.append(f"{k} {l.strip()}")
        elif "v_mfma" in l:
            if out and out[-1].startswith("   mfma"):
                out[-1] += "+"
            else:
                out.append("   mfma")
    print("\n".join(out))


if __name__ == "__main__":
    main(*sys.argv[1:])
