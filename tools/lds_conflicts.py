#!/usr/bin/env python3
"""Static check of the LDS bank behaviour of the FFT transposes (tfhe_kernels.h Geom::idx1/idx2)
against the gfx950 banking rules of MI355X_MICROARCH.md (LDS table):
  ds_write_b64: 4 groups of 16 contiguous lanes, bank = (addr/4) mod 32
  ds_read_b64 : 2 groups of 32 lanes,            bank = (addr/4) mod 64
  ds_read2_b64: each of its two accesses 4 groups of 16 lanes, bank = (addr/4) mod 32 -- the form hipcc
                emits when it pairs adjacent reads
An access is conflict-free when no two lanes of a group touch the same bank with different
addresses.  Prints the worst multiplicity per (N, transpose, direction, access)."""


def geom(logn):
    N = 1 << logn
    NC, LB = N // 2, logn - 1
    PPL = NC // 64
    R = 3 if PPL == 8 else 4
    CB = LB - 2 * R
    jA = lambda t, m: t + 64 * m
    jB = lambda t, m: ((t >> CB) << 6) + (m << CB) + (t & ((1 << CB) - 1))
    jC = lambda t, m: PPL * t + m
    idx1 = lambda j: j + ((j >> 6) << CB)
    idx2 = lambda j: j + (j >> R)
    return PPL, jA, jB, jC, idx1, idx2


def worst(addr_of_lane, group_size, nbanks):
    w = 1
    for g0 in range(0, 64, group_size):
        banks = {}
        for lane in range(g0, g0 + group_size):
            a = addr_of_lane(lane)  # byte address of an 8-byte access
            for half in (0, 4):
                banks.setdefault(((a + half) // 4) % nbanks, set()).add(a)
        w = max(w, max(len(s) for s in banks.values()))
    return w


def main():
    ok = True
    for logn in (10, 11):
        PPL, jA, jB, jC, idx1, idx2 = geom(logn)
        cases = [("A->B (ifft)", idx1, jA, jB), ("B->C (ifft)", idx2, jB, jC),
                 ("C->B (fft)", idx2, jC, jB), ("B->A (fft)", idx1, jB, jA)]
        for name, idx, jw, jr in cases:
            ww = max(worst(lambda t: 8 * idx(jw(t, m)), 16, 32) for m in range(PPL))
            rr = max(worst(lambda t: 8 * idx(jr(t, m)), 32, 64) for m in range(PPL))
            rr2 = max(worst(lambda t: 8 * idx(jr(t, m)), 16, 32) for m in range(PPL))
            # the transposes must also be permutations: every index written exactly once and read once
            wset = sorted(idx(jw(t, m)) for t in range(64) for m in range(PPL))
            rset = sorted(idx(jr(t, m)) for t in range(64) for m in range(PPL))
            perm = wset == rset and len(set(wset)) == 64 * PPL and max(wset) < (1 << (logn - 1)) + 64
            ok &= (ww == 1 and rr == 1 and rr2 == 1 and perm)
            print(f"N={1 << logn:5d} {name:12s} write x{ww} read x{rr} read2 x{rr2} permutation={'ok' if perm else 'BAD'}")
    # complex points (16-byte elements) of the N=1024 blind-rotation kernel: ds_write_b128 is served in 8 groups
    # of 8 contiguous lanes over 32 banks, ds_read_b128 in the 4 groups of 16 lanes listed in
    # MI355X_MICROARCH.md (LDS table) over 64 banks
    R128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
            list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    W128 = [list(range(g, g + 8)) for g in range(0, 64, 8)]

    def worst128(addr_of_lane, groups, nbanks):
        w = 1
        for lanes in groups:
            banks = {}
            for lane in lanes:
                a = addr_of_lane(lane)
                for q in range(0, 16, 4):
                    banks.setdefault(((a + q) // 4) % nbanks, set()).add(a)
            w = max(w, max(len(v) for v in banks.values()))
        return w

    for logn in (10, 11):
        PPL, jA, jB, jC, idx1, idx2 = geom(logn)
        for name, idx, jw, jr in [("A->B (ifft)", idx1, jA, jB), ("B->C (ifft)", idx2, jB, jC), ("C->B (fft)", idx2, jC, jB),
                                  ("B->A (fft)", idx1, jB, jA)]:
            ww = max(worst128(lambda t: 16 * idx(jw(t, m)), W128, 32) for m in range(PPL))
            rr = max(worst128(lambda t: 16 * idx(jr(t, m)), R128, 64) for m in range(PPL))
            if logn == 10:  # the layout the N=1024 kernel uses
                ok &= (ww == 1 and rr == 1)
            print(f"N={1 << logn:5d} {name:12s} complex points: ds_write_b128 x{ww} ds_read_b128 x{rr}")
    ok &= generic_part(worst128_cost, R128, W128)
    print("all conflict-free" if ok else "CONFLICTS / LAYOUT ERROR")
    return 0 if ok else 1


def gen_sw(j):
    """csrc/tfhe_kernels_generic.h gen_sw: where point j of a polynomial sits in its work buffer"""
    return j ^ (((j >> 3) & 7) ^ ((j >> 2) & 8) ^ ((j >> 6) & 1))


def worst128_cost(addr_of_lane, groups, nbanks):
    """LDS cycles of one wave instruction on 16-byte elements: per lane group the largest number of distinct addresses on a bank"""
    tot = 0
    for lanes in groups:
        banks = {}
        for lane in lanes:
            a = addr_of_lane(lane)
            for q in range(0, 16, 4):
                banks.setdefault(((a + q) // 4) % nbanks, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot


def generic_part(cost, R128, W128):
    """The ring-degree-generic kernels (tfhe_kernels_generic.h): every pass of gen_ifft / gen_fft reads and writes the same points in
    place; item t of a pass (consecutive lanes = consecutive items) touches
      first / last / single layer of half-size h : runs of h consecutive points (two per item, h apart)
      two layers (h, h/2)                        : base(t) + {0, 1, 2, 3} x h/2, base = 4 (t - t % (h/2)) + t % (h/2)
      last three layers                          : 8 t + {0 .. 7}
    Checked for every N the generic kernels serve in LDS, every wavefront of a team, with and without the swizzle."""
    ok = True
    for NC in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        if gen_sw(NC - 1) >= NC or sorted(gen_sw(j) for j in range(NC)) != list(range(NC)):
            print(f"generic N={2 * NC}: gen_sw is not a permutation of the buffer")
            ok = False
        assert all(gen_sw(a + b) == gen_sw(a) ^ gen_sw(b) for a in range(0, NC, 8) for b in range(8)), "linearity the passes rely on"
        worst = {False: 1.0, True: 1.0}
        for swz in (False, True):
            sw = gen_sw if swz else (lambda j: j)
            # (count, points of item i): a pass shorter than the team is cut into sub-teams that take further polynomials (gen_split):
            # lane t works on item t % count of polynomial t / count, NC points further on
            pats = [("layer", NC // 2, [lambda i: i, lambda i: i + NC // 2])]
            hh = NC // 8
            while hh >= 8:
                def mk(k, hh=hh):
                    return lambda i: (((i - (i & (hh - 1))) << 2) + (i & (hh - 1))) + k * hh
                pats.append((f"pair {hh}", NC // 4, [mk(k) for k in range(4)]))
                hh //= 2
            if NC >= 16:
                pats.append(("eight", NC // 8, [(lambda i, k=k: 8 * i + k) for k in range(8)]))
            for name, count, fns in pats:
                for wave0 in range(0, max(64, count), 64):
                    for f in fns:
                        addr = lambda t: 16 * (((wave0 + t) // count) * NC + sw(f((wave0 + t) % count)))
                        r = cost(addr, R128, 64) / len(R128)
                        w = cost(addr, W128, 32) / len(W128)
                        worst[swz] = max(worst[swz], r, w)
        print(f"generic N={2 * NC:5d}: worst conflict multiplicity of a pass: x{worst[False]:.0f} unswizzled, x{worst[True]:.0f} with gen_sw")
        # below N = 256 several polynomials share a wavefront's pass (their buffers are a multiple of 256 bytes apart): not pursued
        ok &= worst[True] == 1.0 or NC < 128
    return ok


if __name__ == "__main__":
    raise SystemExit(main())
