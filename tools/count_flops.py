"""Developer tool: algorithmic FLOPs per 64x64 frame of the conv-VAE (2 x conv / convT MACs; train = 3 x forward minus the
stem's data gradient, which is never needed), for the reference depth and the deeper build-defined variant.  Reproduces SURVEY.md
section 8(d) for blocks = 1 (76 349 440 / 227 409 920 at z = 128)."""
import sys


def count(z, blocks, S=64):
    H = S // 2
    macs_stem = 32 * 1 * 25 * H * H
    macs = macs_stem
    inpl = 32
    for planes in (32, 64, 128, 256):
        for b in range(blocks):
            if b == 0:
                H //= 2
            macs += planes * inpl * 9 * H * H + planes * planes * 9 * H * H + (planes * inpl * H * H if b == 0 else 0)
            inpl = planes
    macs += 2 * z * 256 + z * 128 * 4
    cin, H = 128, 2
    for planes in (128, 64, 32, 16, 16):
        for b in range(blocks - 1):
            macs += cin * cin * H * H + cin * cin * 9 * H * H
        macs += planes * cin * H * H + planes * planes * 16 * H * H + cin * planes * 16 * H * H
        cin, H = planes, H * 2
    macs += 16 * 1 * 9 * H * H
    return 2 * macs, 3 * 2 * macs - 2 * macs_stem


if __name__ == "__main__":
    for z, b in ((128, 1), (512, 2)) if len(sys.argv) < 3 else ((int(sys.argv[1]), int(sys.argv[2])),):
        print(f"z={z} blocks={b}: forward {count(z, b)[0]} train {count(z, b)[1]} FLOP/frame")
