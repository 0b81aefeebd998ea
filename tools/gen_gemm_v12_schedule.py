"""Generates the instruction order of GEMM variant 12's k-tile body: moviigen1.1_amd/csrc/gemm_bf16_v12_body_s<N>.inc and gemm_bf16_v12_last.inc.

Variant 11 (tools/gen_gemm_v11_schedule.py) has ONE barrier per k-tile behind `vmcnt(0)`: the 16 LDS-DMA loads a wave issues in k-tile t
must land before the top of k-tile t+1, so they are packed into the first half of the k-tile, where they collide with the fragment reads,
and the barrier still waits for the stragglers (s_memtime: 2528 cycles per k-tile for 2048 cycles of MFMAs; PMC, profiles/r05a_pmc_lib_gemm.txt:
23-26 shader cycles per MFMA and SIMD against 19.2 for the vendor library's kernel of the same tile, which therefore runs the same work at
1.72 instead of 2.05 GHz under the same package-power limit and finishes 13 % earlier).

Variant 12 is the pipeline that gives loads MORE THAN A WHOLE K-TILE to land with the same two 64 KiB stages: the stage a k-tile is
read from is refilled, region by region, WHILE it is being consumed — for the k-tile two ahead.

  body of stream position g (k-tile g of the workgroup's k-tile stream, stage s = g & 1; its k-step-0 fragments are in registers):
    k-step 0 MFMAs (64) ...... in their gaps: the 8 A-fragment reads of k-step 1 (stage s)
        BAR_LGKM (lgkmcnt(0) + s_barrier): every wave has read ALL of stage s's A rows -> the A region of stage s is free
                               ... the wave's 8 A loads of k-tile g+2 -> stage s, interleaved with the 8 W-fragment reads of k-step 1
        BAR_LGKM: the W region of stage s is free -> the wave's 8 W loads of k-tile g+2            ("2bar" schedules: ONE barrier behind all 16 reads)
    k-step 1 MFMAs (64) ...... more loads;
        BAR_VM (vmcnt(n) + s_barrier, n = loads of THIS body issued so far): k-tile g+1 — issued one body ago — has landed in stage s^1
                               ... the 16 fragment reads of k-step 0 of k-tile g+1, the last loads
  (no wait at the end: the next body's counted lgkmcnt waits follow the order in which those 16 reads were issued)
No barrier stands behind a drained memory pipe: a load has ~1.4 k-tiles (~3000 cycles) to land.
`last` = the body of an output tile's LAST k-tile: no loads (the stage becomes the epilogue's transposition buffer), no reads of a next
k-tile (their registers are the epilogue's), no barriers, BUILTIN MFMAs (the compiler then orders the epilogue's accumulator reads behind them).

The generator CHECKS what makes a schedule legal: an A (W) load only behind a barrier that follows the last A (W) fragment read of the
stage; next-k-tile reads only behind BAR_VM; vmcnt's count = the loads issued before it; 16 loads, 32 reads, 128 MFMAs.
MFMA order as variants 7 / 11: per k-step, token half h outer, feature block i inner (group g = 8 h + i: acc[i][4h .. 4h+3]) — every
accumulator sees its k-steps in the same order as in every other variant: identical bits.
usage: python tools/gen_gemm_v12_schedule.py          writes every schedule of SCHEDULES (the .hip instantiates the ones its launcher names)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEXT_ORDER = [('w', 0), ('a', 0), ('a', 1), ('a', 2), ('a', 3)] + [('w', b) for b in range(1, 8)] + [('a', b) for b in range(4, 8)]


def tail(b3, stride=(1, 2), load_every=6):
    """behind BAR_VM: the 16 reads of the next k-tile's k-step 0, the last three loads between them"""
    ev, k, slots = [], b3 + 1, []
    for n in range(16):
        slots.append(k)
        k += stride[n % 2]
    assert slots[-1] <= 125, slots
    for (op, b), kk in zip(NEXT_ORDER, slots):
        ev.append((kk, ('RN', op, b)))
    k = b3 + 5
    for n in range(13, 16):
        while k in slots:
            k += 1
        ev.append((k, ('G', n)))
        k += load_every
    assert k - load_every <= 126
    return ev


def three_bar(b1=19, b2=46, b3=91, g3=86, tail_stride=(1, 2), load_every=6):
    """the first schedule (profiles/r05b_gemm_v12.log), modelled on where the vendor kernel puts its barriers"""
    ev = []
    for n in range(8):
        ev.append((1 + 2 * n, ('R1', 'a', n)))
    ev.append((b1, ('BAR_LGKM',)))
    k = b1 + 2
    for n in range(5):                               # A loads 0-4 with the W fragments 0-4 of k-step 1
        ev.append((k, ('G', n)))
        ev.append((k + 2, ('R1', 'w', n)))
        k += 3
    for n in range(5, 8):
        ev.append((k + 1, ('R1', 'w', n)))
        k += 2
    assert k + 2 < b2, (k, b2)
    ev.append((b2, ('BAR_LGKM',)))
    k = b2 + 2
    for n in range(5, 10):                           # A loads 5-7, W loads 8-9
        ev.append((k, ('G', n)))
        k += 3
    k = g3
    for n in range(10, 13):                          # W loads 10-12
        ev.append((k, ('G', n)))
        k += 2
    assert k - 1 <= b3, (k, b3)
    ev.append((b3, ('BAR_VM',)))
    return ev + tail(b3, tail_stride, load_every)


def three_bar_spread(b1=19, b2=46, b3=96, stride=5):
    """three barriers, the loads behind BAR2 spread evenly up to BAR_VM instead of two clusters"""
    ev = []
    for n in range(8):
        ev.append((1 + 2 * n, ('R1', 'a', n)))
    ev.append((b1, ('BAR_LGKM',)))
    k = b1 + 2
    for n in range(5):
        ev.append((k, ('G', n)))
        ev.append((k + 2, ('R1', 'w', n)))
        k += 3
    for n in range(5, 8):
        ev.append((k + 1, ('R1', 'w', n)))
        k += 2
    ev.append((b2, ('BAR_LGKM',)))
    k = b2 + 2
    for n in range(5, 13):
        ev.append((k, ('G', n)))
        k += stride
    assert k - stride < b3, (k, b3)
    ev.append((b3, ('BAR_VM',)))
    return ev + tail(b3, (1, 1), load_every=5)


def spread(r1w_first=17, r1w_stride=2, bars=(34,), first=36, stride=6.0, b3=92, rn_stride=(1, 2)):
    """generic: A reads at gaps 1, 3 .. 15; W reads from r1w_first at r1w_stride; lgkmcnt barriers at `bars`; the 16 loads at
    first + n * stride (whatever side of BAR_VM they fall on); behind BAR_VM (gap b3) the 16 next-k-tile reads in the gaps without a load"""
    ev = [(1 + 2 * n, ('R1', 'a', n)) for n in range(8)]
    ev += [(r1w_first + r1w_stride * n, ('R1', 'w', n)) for n in range(8)]
    ev += [(b, ('BAR_LGKM',)) for b in bars]
    lg = [int(round(first + n * stride)) for n in range(16)]
    assert lg[-1] <= 126 and len(set(lg)) == 16, lg
    ev += [(g, ('G', n)) for n, g in enumerate(lg)]
    ev.append((b3, ('BAR_VM',)))
    k, n = b3 + 1, 0
    while n < 16:
        assert k <= 126, ('next reads do not fit', k)
        if k not in lg:
            ev.append((k, ('RN',) + NEXT_ORDER[n]))
            n += 1
            k += rn_stride[n % 2]
        else:
            k += 1
    ev.sort(key=lambda t: (t[0], 0 if t[1][0] == 'BAR_LGKM' else 1))
    return ev


SCHEDULES = {
    0: ('3 barriers at gaps 19 / 46 / 91 (the first version)', three_bar()),
    1: ('3 barriers, BAR_VM late (gap 99)', three_bar(b3=99, g3=94, tail_stride=(1, 1), load_every=4)),
    2: ('1 lgkmcnt barrier at gap 34, loads every 6th gap from 36, BAR_VM at 92', spread()),
    3: ('as 0 with the lgkmcnt barriers 4 gaps later (23 / 50): the reads in front of them get 8 MFMAs to return', three_bar(b1=23, b2=50)),
    4: ('as 0 with the lgkmcnt barriers 8 gaps later (27 / 54)', three_bar(b1=27, b2=54)),
    5: ('3 barriers, loads behind BAR2 every 5th gap, BAR_VM at gap 96', three_bar_spread()),
}


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


def body(events):
    """events = [(gap, event)] of a full body, or None for the last k-tile of a tile; gap k = behind the k-th MFMA"""
    full = events is not None
    s = Sched(carried=True, builtin=not full)
    if not full:
        events = [(1 + 2 * n, ('R1', 'a', n)) for n in range(8)] + [(17 + 2 * n, ('R1', 'w', n)) for n in range(8)]
    gaps = {}
    for k, e in events:
        assert 0 <= k < 128, (k, e)
        if e[0] == 'G':                       # M0 is written one gap before its load (no s_nop between them)
            gaps.setdefault(k - 1, []).append(('M0', e[1]))
    for k, e in events:
        gaps.setdefault(k, []).append(e)
    if full:
        # Loop-carried state, advanced INSIDE the body by one pinned instruction each (V12_X), in a free gap behind its last use: the
        # k-step-1 read addresses of this stage (ab1, wb1) behind the last R1 read, the k-step-0 addresses of the other stage (abn, wbn)
        # behind the last RN read, the M0 base (lload) behind the last M0 write, the k offset (kb) behind the last load.  Left to the
        # compiler they were ~10 SALU / VALU instructions at the top of every body, with the matrix pipe empty.
        last = {'ab1': max(k for k, e in events if e[0] == 'R1' and e[1] == 'a'), 'wb1': max(k for k, e in events if e[0] == 'R1' and e[1] == 'w'),
                'abn': max(k for k, e in events if e[0] == 'RN' and e[1] == 'a'), 'wbn': max(k for k, e in events if e[0] == 'RN' and e[1] == 'w'),
                'lload': max(k for k, e in events if e[0] == 'G') - 1, 'kb': max(k for k, e in events if e[0] == 'G')}
        for name in ('ab1', 'wb1', 'abn', 'wbn', 'lload', 'kb'):
            k = last[name] + 1
            while k in gaps and k < 127:
                k += 1
            assert k <= 127, (name, k)
            gaps.setdefault(k, []).append(('X', name))
    order = [(0, g) for g in range(16)] + [(1, g) for g in range(16)]
    loads, reads1 = 0, {'a': 0, 'w': 0}
    free = {'a': False, 'w': False}           # region of the stage released by a barrier behind its last fragment read
    landed, n_next = False, 0
    for k4, (ks, g) in enumerate(order):
        s.wait_for(needs(ks, g))
        for jj in range(4):
            s.mfma(ks, g & 7, 4 * (g >> 3) + jj)
            for e in gaps.get(4 * k4 + jj, []):
                if e[0] == 'M0':
                    s.emit(f'V12_M0({e[1]});')
                elif e[0] == 'G':
                    assert e[1] == loads, 'loads in piece order'
                    assert free['a' if e[1] < 8 else 'w'], ('load into a region still being read', e, 4 * k4 + jj)
                    s.emit(f'V12_G({e[1]});')
                    loads += 1
                elif e[0] == 'BAR_LGKM':
                    s.emit('V12_BAR_LGKM;')
                    s.all_returned()
                    free = {op: reads1[op] == 8 for op in 'aw'}
                elif e[0] == 'BAR_VM':
                    assert not landed and loads <= 15
                    s.emit(f'V12_BAR_VM({loads});')
                    landed = True
                elif e[0] == 'X':
                    s.emit(f'V12_X_{e[1]};')
                elif e[0] == 'R1':
                    assert not free[e[1]]
                    s.read(1, e[1], e[2])
                    reads1[e[1]] += 1
                elif e[0] == 'RN':
                    assert landed and (e[1], e[2]) == NEXT_ORDER[n_next]
                    s.read(0, e[1], e[2], nxt=True)
                    n_next += 1
    assert reads1 == {'a': 8, 'w': 8}
    if full:
        assert loads == 16 and n_next == 16, (loads, n_next)
    return s


def write(name, sch, note):
    path = os.path.join(os.environ.get('MG_V12_GEN_DIR', os.path.join(ROOT, 'moviigen1.1_amd', 'csrc')), f'gemm_bf16_v12_{name}.inc')
    with open(path, 'w') as f:
        f.write(f'// GENERATED by tools/gen_gemm_v12_schedule.py — do not edit.  {note}\n'
                f'// GEMM variant 12, section `{name}`: one k-tile of 128 MFMAs, every other instruction in the gap behind an MFMA.\n')
        f.write('\n'.join(sch.out) + '\n')
    o = sch.out
    print(name, len(o), 'statements;', sum('mfma' in x for x in o), 'MFMAs,', sum('v11_rd' in x for x in o), 'reads,', sum('V12_G' in x for x in o), 'loads,',
          sum('v11_wait' in x for x in o), 'counted waits,', sum('V12_BAR' in x for x in o), 'barriers')


if __name__ == '__main__':
    # without arguments: the two bodies the library compiles (0 = shipped, 2 = its A/B partner); `... 1 3 4 5` regenerates the bodies
    # that were measured and archived (experiments/gemm_v12_bodies/, set MG_V12_GEN_DIR to write there)
    only = [int(a) for a in sys.argv[1:]] or [0, 2]
    for n, (note, ev) in SCHEDULES.items():
        if n in only:
            write(f'body_s{n}', body(ev), note)
    write('last', body(None), 'last k-tile of an output tile')
