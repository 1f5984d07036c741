#!/usr/bin/env python3
"""LDS bank-conflict model of ntt_tile_kernel (csrc/ntt.hip), access class by access class ("nobody has located which access
conflicts").  The banking rules are the ones of /opt/skills/guides/MI355X_MICROARCH.md §LDS: a wave64 LDS instruction is served in fixed lane
groups, one LDS cycle per group; within a group every further distinct address on a busy bank costs one more cycle.

    ds_read_b128   4 groups of 16 lanes {0-3,12-15,20-27} {4-11,16-19,28-31} (+32), bank = (a/4) % 64, 4 banks per lane
    ds_read_b32    2 groups of 32 lanes, bank = (a/4) % 32     (ds_read2_b32 = two of them)
    ds_write_b128  8 groups of 8 contiguous lanes, bank = (a/4) % 32
    ds_write_b32   2 groups of 32 lanes, bank = (a/4) % 32

An LDS element is 48 B (9 limbs + 3 pad words): the kernel moves it as 2 x b128 + 1 x b32 (hipcc -S); a stage twiddle is a packed 36-byte
element read as 4 x ds_read2_b32 + 1 x ds_read_b32.  The model walks one tile of a pass for all four waves of the workgroup and reports, per
access class, the LDS-array cycles and how many of them are conflict cycles — the ratio the PMC pair SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
measures for the whole kernel.

    python tools/ntt_lds_model.py [coset]         # the 2^22 transform's three passes (or a k = 19 coset transform's): r05 layout, planes only, the r06 layout
"""
import sys

G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G32 = [list(range(32)), list(range(32, 64))]
GW128 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def cycles(addrs, groups, banks, width_dw):
    """addrs: byte address per lane (None = lane inactive) -> (cycles, conflict cycles)"""
    tot = conf = 0
    for g in groups:
        per_bank = {}
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            for w in range(width_dw):
                dw = a // 4 + w
                per_bank.setdefault(dw % banks, set()).add(dw)
        c = max((len(s) for s in per_bank.values()), default=0)
        if c:
            tot += c
            conf += c - 1
    return tot, conf


class Acc:
    def __init__(self):
        self.d = {}

    def add(self, cls, tc):
        t, c = self.d.get(cls, (0, 0))
        self.d[cls] = (t + tc[0], c + tc[1])


def bitrev(x, m):
    return int(format(x, "0%db" % m)[::-1], 2) if m else 0


def model(m, cb, kind, layout, quarter=False):
    """one tile; kind 0 = first pass, 1 = middle, 2 = last.  layout: dict(swz=fn(e)->physical element index, tw=fn(k, m)->physical twiddle index)"""
    acc = Acc()
    C, R, T = 1 << cb, 1 << m, 256
    swz, twp = layout["swz"], layout["tw"]
    if layout.get("planes"):   # r06: limbs 0-3, limbs 4-7 and limb 8 in three planes of 16 / 16 / 4 bytes per element
        parts = lambda e: (16 * swz(e), 16384 + 16 * swz(e), 32768 + 4 * swz(e))
        TW0 = 36 * 1024
    else:                      # r05: 48-byte elements
        parts = lambda e: (48 * swz(e), 48 * swz(e) + 16, 48 * swz(e) + 32)
        TW0 = 48 * 1024

    def elem_read(cls, es):     # es: element index per lane
        acc.add(cls, cycles([None if e is None else parts(e)[0] for e in es], G128, 64, 4))
        acc.add(cls, cycles([None if e is None else parts(e)[1] for e in es], G128, 64, 4))
        acc.add(cls, cycles([None if e is None else parts(e)[2] for e in es], G32, 32, 1))

    def elem_write(cls, es):
        acc.add(cls, cycles([None if e is None else parts(e)[0] for e in es], GW128, 32, 4))
        acc.add(cls, cycles([None if e is None else parts(e)[1] for e in es], GW128, 32, 4))
        acc.add(cls, cycles([None if e is None else parts(e)[2] for e in es], G32, 32, 1))

    def tw_read(cls, ks):
        if layout.get("planes"):   # the twiddle planes: 16 / 16 / 4 bytes per entry behind the data planes
            ps = [twp(k, m) for k in ks]
            acc.add(cls, cycles([TW0 + 16 * q for q in ps], G128, 64, 4))
            acc.add(cls, cycles([TW0 + 4096 + 16 * q for q in ps], G128, 64, 4))
            acc.add(cls, cycles([TW0 + 8192 + 4 * q for q in ps], G32, 32, 1))
            return
        for w in range(9):      # read2_b32 = two b32 accesses; 9 dword accesses in all
            acc.add(cls, cycles([TW0 + 36 * twp(k, m) + 4 * w for k in ks], G32, 32, 1))

    for wave in range(4):
        tids = [wave * 64 + l for l in range(64)]
        # fill
        for k in range(4):
            if kind == 0 and quarter and k and not (m & 1):
                continue
            es = []
            for tid in tids:
                e = tid + T * k
                t, c = e >> cb, e & (C - 1)
                es.append((bitrev(t, m) << cb) + c)
            elem_write("fill (write)", es)
        st = 0
        if m & 1:
            for k in range(2):
                e0s = []
                for tid in tids:
                    b = tid + T * k
                    c, p = b & (C - 1), b >> cb
                    e0s.append(((p << 1) << cb) + c)
                for off in (0, C):
                    elem_read("radix-2 stage (read)", [e + off for e in e0s])
                    elem_write("radix-2 stage (write)", [e + off for e in e0s])
            st = 1
        while st < m:
            h = 1 << st
            e0s, iis = [], []
            for tid in tids:
                gc, gp = tid & (C - 1), tid >> cb
                i, blk = gp & (h - 1), gp >> st
                e0s.append((((blk << (st + 2)) + i) << cb) + gc)
                iis.append(i)
            stride = h << cb
            zeros = (1 if st == 0 else 2 if st == 1 else 0) if (kind == 0 and quarter and st < 2) else 0
            name = "round st=%d" % st
            if zeros == 1:
                elem_read(name + " data (read)", e0s)
                for j in (1, 2, 3):
                    elem_write(name + " data (write)", [e + j * stride for e in e0s])
            else:
                for j in ((0, 2) if zeros == 2 else (0, 1, 2, 3)):
                    elem_read(name + " data (read)", [e + j * stride for e in e0s])
                if st and zeros == 0:
                    tw_read(name + " twiddles", [i << (m - 1 - st) for i in iis])
                tw_read(name + " twiddles", [i << (m - 2 - st) for i in iis])
                tw_read(name + " twiddles", [(i + h) << (m - 2 - st) for i in iis])
                for j in (0, 1, 2, 3):
                    elem_write(name + " data (write)", [e + j * stride for e in e0s])
            st += 2
        for k in range(4):
            es = []
            for tid in tids:
                e = tid + T * k
                if kind == 0:
                    c, u = e >> m, e & (R - 1)
                else:
                    u, c = e >> cb, e & (C - 1)
                es.append((u << cb) + c)
            elem_read("read-out (read)", es)
    return acc.d


def report(title, d):
    tot = sum(t for t, _ in d.values())
    conf = sum(c for _, c in d.values())
    print("%s: LDS-array cycles per tile %d, conflict cycles %d = %.2f" % (title, tot, conf, conf / tot))
    for k, (t, c) in d.items():
        if c:
            print("    %-28s %6d cycles, %6d conflict (%.2f of the class, %.3f of the tile)" % (k, t, c, c / t, c / tot))
    return tot, conf


R05 = dict(swz=lambda e: e, tw=lambda k, m: k)


ROWS_R06 = (1, 6, 18, 15, 27)   # TileLayoutPlanes::swz in csrc/ntt.hip: bit b of X[e >> 5] = parity((e >> 5) & ROWS_R06[b])


def swz_r06(e):
    up, x = e >> 5, 0
    for b, r in enumerate(ROWS_R06):
        x |= (bin(up & r).count("1") & 1) << b
    return e ^ x


# r06 (shipped): three limb planes at the swizzled index, stage twiddles in the same planes at the skewed index k + (k >> 4)
R06 = dict(swz=swz_r06, tw=lambda k, m: k + (k >> 4), planes=True)
R06_NO_SWZ = dict(swz=lambda e: e, tw=lambda k, m: k + (k >> 4), planes=True)

if __name__ == "__main__":
    passes = [(8, 2, 0, False), (7, 3, 1, False), (7, 3, 2, False)]
    if len(sys.argv) > 1 and sys.argv[1] == "coset":
        passes = [(7, 3, 0, True), (7, 3, 1, False), (7, 3, 2, False)]
    for name, lay in (("r05 layout", R05), ("r06 planes, no swizzle", R06_NO_SWZ), ("r06 layout", R06)):
        T = Cc = 0
        for m, cb, kind, q in passes:
            t, c = report("%s  m=%d cb=%d kind=%d%s" % (name, m, cb, kind, " quarter" if q else ""), model(m, cb, kind, lay, q))
            T += t
            Cc += c
        print("== %s, whole transform: conflict / all LDS cycles = %.3f\n" % (name, Cc / T))
