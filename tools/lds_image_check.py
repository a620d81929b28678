"""Bank-conflict check of the row-major K / V LDS images of the dense twin of the 32x32-MFMA prefill body
(prefill_mfma32.h, DENSE): the 16-byte row reads of the K fragments (ds_read_b128), the transposed reads of the V^T
fragments (ds_read_b64_tr_b16) and the copy's 16-byte writes (ds_write_b128), with the banking rules of
MI355X_MICROARCH.md 'LDS'.  Prints the worst multiplicity per instruction kind; 1 = conflict-free.  Host-only."""
import sys

B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def swz(D, row, c):
    r1 = row >> 1
    if D == 64:   # 128-byte rows: two rows per bank row
        return c ^ (((r1 & 1) << 2) | ((r1 >> 1) & 3))
    # 256-byte rows (the guide's image (b))
    return c ^ (((row & 3) << 2) | ((row >> 2) & 3))


def off(D, row, c):
    return row * D * 2 + swz(D, row, c) * 16


def worst(addrs_by_lane, groups, unit, nbanks_units):
    w = 1
    for g in groups:
        slots = {}
        for l in g:
            a = addrs_by_lane[l]
            slots.setdefault((a // unit) % nbanks_units, set()).add(a // unit)
        w = max(w, max(len(v) for v in slots.values()))
    return w


def main():
    for D in (64, 128):
        krow = lambda col: (col & ~0xc) | ((col & 4) << 1) | ((col & 8) >> 1)
        wk = 1
        for kh in range(2):
            for ks in range(D // 16):
                addrs = []
                for lane in range(64):
                    col, hi = lane & 31, lane >> 5
                    addrs.append(off(D, 32 * kh + krow(col), 2 * ks + hi))
                wk = max(wk, worst(addrs, B128_GROUPS, 16, 16))
        wv = 1
        for db in range(D // 32):
            for ks in range(4):
                for t in range(2):
                    addrs = []
                    for lane in range(64):
                        col, hi = lane & 31, lane >> 5
                        q, pp = (lane & 15) >> 2, lane & 3
                        key = 16 * ks + 8 * hi + 4 * t + q
                        c = 4 * db + 2 * (col >> 4) + (pp >> 1)
                        addrs.append(off(D, key, c) + 8 * (pp & 1))
                    wv = max(wv, worst(addrs, [list(range(32)), list(range(32, 64))], 8, 32))
        ww = 1
        cpr = D // 8
        for piece in range(64 * cpr // 64):
            addrs = []
            for lane in range(64):
                row = piece * (64 // cpr) + lane // cpr
                addrs.append(off(D, row, lane % cpr))
            ww = max(ww, worst(addrs, [list(range(8 * i, 8 * i + 8)) for i in range(8)], 16, 8))
        print(f"D {D}: K row reads {wk}-way, V transposed reads {wv}-way, copy writes {ww}-way")
    return 0


if __name__ == "__main__":
    sys.exit(main())
