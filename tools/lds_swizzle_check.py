#!/usr/bin/env python3
"""Exhaustive bank-conflict count of the direct C_in = 64 kernels' fragment reads (csrc/conv_c64.hip) under the real
ds_read_b128 lane groups of gfx950 (MI355X_MICROARCH.md, LDS table): rows of 128 bytes, lane (fr, fh) reads 16-byte chunk
(4 s + fh) ^ swz(row) of row start + fr.  Prints the extra LDS cycles (sum over groups of max multiplicity - 1) for the former
and the current key over all start rows (the current one must print 0), and those of the fused kernels' 8-byte patch stores."""
G = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G += [[lane + 32 for lane in g] for g in G]


def extra_cycles(swz, starts):
    total = 0
    for b in starts:
        for s in (0, 1):
            for g in G:
                seen = {}
                for lane in g:
                    fr, fh = lane & 15, lane >> 4
                    row = b + fr
                    slot = ((row & 1) << 3) | ((4 * s + fh) ^ swz(row))     # 16-byte slot within the 256-byte bank row
                    seen[slot] = seen.get(slot, 0) + 1
                total += max(seen.values()) - 1
    return total


def patch_store_extra_cycles(swz):
    """The fused conv1 kernels' conv1_1 patch stores: lane (fr, fh) writes 8 bytes at row (b + fr), chunk (2 j + (fh >> 1)) ^ swz(row),
    half fh & 1.  ds_write_b64: four contiguous 16-lane groups, 32 banks of 4 bytes."""
    extra = n = 0
    for b in range(64):
        for j in range(4):
            for fh in range(4):
                banks = {}
                for fr in range(16):
                    row = b + fr
                    a = row * 128 + (((2 * j + (fh >> 1)) ^ swz(row)) << 4) + ((fh & 1) << 3)
                    for d in (0, 4):
                        banks.setdefault(((a + d) // 4) % 32, set()).add(a + d)
                extra += max(len(v) for v in banks.values()) - 1
                n += 1
    return extra, n


def ps_extra_cycles(TC, HALO, RM):
    """csrc/conv_ps.hip pixel-fragment reads: lane (fr, fh) of pixel fragment i (wave row wm) reads, for tap (r, s) and K substep
    ks, the 16-byte chunk (4 ks + fh) ^ ((tc + RM R0 + s + RM r) & 6) of patch pixel (R0 + r) PW + tc + s, where pixel
    ml = 112 wm + 16 i + fr sits at tile row R0 = ml // TC, column tc = ml % TC and PW = TC + 2 HALO (pixels are 128-byte rows).
    Returns (extra cycles, group accesses) over every wm, i, tap and substep."""
    PW, KW = TC + 2 * HALO, 2 * HALO + 1
    extra = n = 0
    for wm in (0, 1):
        for i in range(7):
            for r in range(KW):
                for s_ in range(KW):
                    for ks in (0, 1):
                        for g in G:
                            seen = {}
                            for lane in g:
                                fr, fh = lane & 15, lane >> 4
                                ml = 112 * wm + 16 * i + fr
                                R0, tc = divmod(ml, TC)
                                px = (R0 + r) * PW + tc + s_
                                chunk = (4 * ks + fh) ^ ((tc + RM * R0 + s_ + RM * r) & 6)
                                slot = ((px & 1) << 3) | chunk
                                seen[slot] = seen.get(slot, 0) + 1
                            extra += max(seen.values()) - 1
                            n += 1
    return extra, n


if __name__ == "__main__":
    for TC, HALO, RM in ((28, 1, 4), (14, 1, 6), (14, 2, 6), (28, 2, 4)):
        best = min(range(8), key=lambda m: ps_extra_cycles(TC, HALO, m)[0])
        e, n_ = ps_extra_cycles(TC, HALO, RM)
        print("conv_ps TC=%d %dx%d key (col + %d row) & 6: %d extra cycles over %d group accesses (best multiplier %d: %d)"
              % (TC, 2 * HALO + 1, 2 * HALO + 1, RM, e, n_, best, ps_extra_cycles(TC, HALO, best)[0]))
    n = 64 * 2 * 4
    print("round 1 (row >> 1) & 7: %d extra cycles over %d group accesses" % (extra_cycles(lambda r: (r >> 1) & 7, range(64)), n))
    print("round 2 row & 6       : %d extra cycles over %d group accesses" % (extra_cycles(lambda r: r & 6, range(64)), n))
    new = extra_cycles(lambda r: r & 7, range(64))
    print("row & 7               : %d extra cycles over %d group accesses" % (new, n))
    w6, nw = patch_store_extra_cycles(lambda r: r & 6)
    w7, _ = patch_store_extra_cycles(lambda r: r & 7)
    print("conv1_1 patch stores (ds_write_b64): row & 6 %d, row & 7 %d extra cycles over %d group accesses" % (w6, w7, nw))
    raise SystemExit(0 if new == 0 and w7 <= nw else 1)
