"""Generates the instruction order of GEMM variant 12's k-tile body: moviigen1.1_amd/csrc/gemm_bf16_v12_{body,last}.inc.

Variant 11 (tools/gen_gemm_v11_schedule.py) has ONE barrier per k-tile behind `vmcnt(0)`: the 16 LDS-DMA loads a wave issues in k-tile t
must land before the top of k-tile t+1, so they are packed into the first half of the k-tile, where they collide with the fragment reads,
and the barrier still waits for the stragglers (s_memtime: 2528 cycles per k-tile for 2048 cycles of MFMAs; PMC, profiles/r05a_pmc_lib_gemm.txt:
23-26 shader cycles per MFMA and SIMD against 19.2 for the vendor library's kernel of the same tile, which therefore runs the same work at
1.72 instead of 2.05 GHz under the same package-power limit and finishes 13 % earlier).

Variant 12 is the pipeline that gives loads MORE THAN A WHOLE K-TILE to land with the same two 64 KiB stages: the stage a k-tile is
read from is refilled, region by region, WHILE it is being consumed — for the k-tile two ahead.

  body of stream position g (k-tile g of the workgroup's k-tile stream, stage s = g & 1; its k-step-0 fragments are in registers):
    k-step 0 MFMAs (64) ...... in their gaps: the 8 A-fragment reads of k-step 1 (stage s)
        BAR1 = lgkmcnt(0) + s_barrier: every wave has read ALL of stage s's A rows -> the A region of stage s is free
                               ... the wave's 8 A loads of k-tile g+2 -> stage s, interleaved with the 8 W-fragment reads of k-step 1
        BAR2 = lgkmcnt(0) + s_barrier: the W region of stage s is free -> the wave's 8 W loads of k-tile g+2
    k-step 1 MFMAs (64) ...... more loads;
        BAR3 = vmcnt(n) + s_barrier (n = loads of THIS body issued so far): k-tile g+1 — issued one body ago — has landed in stage s^1
                               ... the 16 fragment reads of k-step 0 of k-tile g+1, the last loads
  (no wait at the end: the next body's counted lgkmcnt waits follow the order in which those 16 reads were issued)
Three barriers instead of one, but none of them behind a drained memory pipe: a load has ~1.4 k-tiles (~3000 cycles) to land.
`last` = the body of an output tile's LAST k-tile: no loads (the stage becomes the epilogue's transposition buffer), no reads of a next
k-tile (their registers are the epilogue's), no barriers; the .hip closes it with vmcnt(0) + s_barrier.

MFMA order as variants 7 / 11: per k-step, token half h outer, feature block i inner (group g = 8 h + i: acc[i][4h .. 4h+3]) — every
accumulator sees its k-steps in the same order as in every other variant: identical bits.
usage: python tools/gen_gemm_v12_schedule.py [key=value ...]     keys: b1 b2 b3 (barrier gaps), g3 (first gap of the third load group)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = {'b1': 19, 'b2': 46, 'b3': 91, 'g3': 86, 'name': ''}
for a in sys.argv[1:]:
    k, v = a.split('=')
    P[k] = v if k == 'name' else int(v)

NEXT_ORDER = [('w', 0), ('a', 0), ('a', 1), ('a', 2), ('a', 3)] + [('w', b) for b in range(1, 8)] + [('a', b) for b in range(4, 8)]


class Sched:
    def __init__(self, carried, builtin=False):
        self.out, self.builtin = [], builtin
        self.issued = [f'f0{op}[{b}]' for op, b in NEXT_ORDER] if carried else []   # reads in flight at the top: the previous body's last 16
        self.returned_upto = -1

    def emit(self, s):
        self.out.append('    ' + s + ' V12_SB;')

    def read(self, ks, op, blk, nxt=False):
        name = f'f{ks}{op}[{blk}]'
        base = ('wbn' if op == 'w' else 'abn') if nxt else (('wb' if op == 'w' else 'ab') + str(ks))
        self.emit(f'v11_rd<{blk * 2048}>({name}, {base});')
        self.issued.append(name)

    def wait_for(self, names):
        idx = [len(self.issued) - 1 - self.issued[::-1].index(n) for n in names if n in self.issued]
        if not idx or max(idx) <= self.returned_upto:
            return
        n = min(len(self.issued) - 1 - max(idx), 15)        # lgkmcnt is a 4-bit field: a smaller count only waits for more
        self.emit(f'v11_wait<{n}>();')
        self.returned_upto = len(self.issued) - 1 - n

    def all_returned(self):
        self.returned_upto = len(self.issued) - 1

    def mfma(self, ks, i, j):
        if self.builtin:        # the compiler sees these: it orders the epilogue's accumulator reads (and its own register moves) behind them
            self.emit(f'acc[{i}][{j}] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f{ks}w[{i}], f{ks}a[{j}], acc[{i}][{j}], 0, 0, 0);')
        else:
            self.emit(f'v11_mfma(acc[{i}][{j}], f{ks}w[{i}], f{ks}a[{j}]);')


def needs(ks, g):
    h, i = g >> 3, g & 7
    return [f'f{ks}w[{i}]'] + [f'f{ks}a[{4 * h + jj}]' for jj in range(4)]


def body(full):
    s = Sched(carried=True, builtin=not full)
    gaps = {}

    def put(k, ev):
        assert 0 <= k < 128, k
        gaps.setdefault(k, []).append(ev)
    for n in range(8):                                   # A fragments of k-step 1
        put(1 + 2 * n, ('R1', 'a', n))
    if full:
        b1, b2, b3, g3 = P['b1'], P['b2'], P['b3'], P['g3']
        assert b1 > 15
        put(b1 - 2, ('M0', 0))
        put(b1, ('BAR_LGKM', 1))
        k = b1 + 2
        for n in range(5):                               # A loads 0-4 with the W fragments 0-4 of k-step 1
            put(k, ('G', n)); put(k + 1, ('M0', n + 1)); put(k + 2, ('R1', 'w', n))
            k += 3
        for n in range(5, 8):
            put(k + 1, ('R1', 'w', n))
            k += 2
        assert k + 2 < b2, (k, b2)
        put(b2, ('BAR_LGKM', 2))
        k = b2 + 2
        for n in range(5, 10):                           # A loads 5-7, W loads 8-9
            put(k, ('G', n)); put(k + 1, ('M0', n + 1))
            k += 3
        assert k <= 64 + 8
        k = g3
        for n in range(10, 13):                          # W loads 10-12
            put(k, ('G', n))
            if n < 12:
                put(k + 1, ('M0', n + 1))
            k += 2
        assert k - 1 <= b3, (k, b3)
        put(b3, ('BAR_VM', 13))
        # behind BAR3: the 16 reads of the next k-tile's k-step 0, the last three loads between them
        k = b3 + 1
        slots = []
        for n in range(16):
            slots.append(k)
            k += 1 if n % 2 == 0 else 2
        for (op, b), kk in zip(NEXT_ORDER, slots):
            put(kk, ('RN', op, b))
        free = [g for g in range(b3 + 1, 126) if g not in slots]
        for n, kk in zip(range(13, 16), free[0::2]):
            put(kk, ('M0', n))
            put(kk + 3 if (kk + 3) not in slots else kk + 4, ('G', n))
    else:
        for n in range(8):
            put(17 + 2 * n, ('R1', 'w', n))
    order = [(0, g) for g in range(16)] + [(1, g) for g in range(16)]
    loads_issued = 0
    for k4, (ks, g) in enumerate(order):
        s.wait_for(needs(ks, g))
        for jj in range(4):
            s.mfma(ks, g & 7, 4 * (g >> 3) + jj)
            for e in gaps.get(4 * k4 + jj, []):
                if e[0] == 'M0':
                    s.emit(f'V12_M0({e[1]});')
                elif e[0] == 'G':
                    s.emit(f'V12_G({e[1]});')
                    loads_issued += 1
                elif e[0] == 'BAR_LGKM':
                    s.emit(f'V12_BAR_LGKM({e[1]});')
                    s.all_returned()
                elif e[0] == 'BAR_VM':
                    assert loads_issued == e[1], (loads_issued, e)
                    s.emit(f'V12_BAR_VM({e[1]});')
                elif e[0] == 'R1':
                    s.read(1, e[1], e[2])
                elif e[0] == 'RN':
                    s.read(0, e[1], e[2], nxt=True)
    if full:
        assert loads_issued == 16 and s.issued[-16:] == [f'f0{op}[{b}]' for op, b in NEXT_ORDER]
    return s


for name, full in (('body', True), ('last', False)):
    sch = body(full)
    path = os.path.join(os.environ.get('MG_V12_GEN_DIR', os.path.join(ROOT, 'moviigen1.1_amd', 'csrc')), f'gemm_bf16_v12_{name}{P["name"]}.inc')
    with open(path, 'w') as f:
        f.write(f'// GENERATED by tools/gen_gemm_v12_schedule.py (b1 = {P["b1"]}, b2 = {P["b2"]}, b3 = {P["b3"]}, g3 = {P["g3"]}) — do not edit.\n'
                f'// GEMM variant 12, section `{name}`: one k-tile of 128 MFMAs, every other instruction in the gap behind an MFMA.\n')
        f.write('\n'.join(sch.out) + '\n')
    o = sch.out
    print(name, len(o), 'statements;', sum('mfma' in x for x in o), 'MFMAs,', sum('v11_rd' in x for x in o), 'reads,', sum('V12_G' in x for x in o), 'loads,',
          sum('v11_wait' in x for x in o), 'counted waits,', sum('V12_BAR' in x for x in o), 'barriers')
