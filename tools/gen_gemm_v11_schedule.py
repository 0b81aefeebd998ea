"""Generates the instruction order of GEMM variant 11's k-tile (64 deep): moviigen1.1_amd/csrc/gemm_bf16_v11_{ktile_s6,ktile_s4,tail}.inc.

Why a generator: variant 7 (one wave per SIMD, 128 x 128 per wave) issues its non-MFMA work in clumps — after every four MFMAs
`s_add m0 / v_lshl_add_u64 / s_nop / ds_read / global_load_lds / s_waitcnt` — and a clump takes longer than the 16 cycles the fourth
MFMA keeps the matrix pipe busy.  The vendor library's assembly kernel of the same structure (256 x 256 x 64, four waves, two 64 KiB
stages, direct-to-LDS loads) never puts more than one or two cheap instructions between two MFMAs and reaches 85 % matrix-pipe
occupancy against 64 % (profiles/r04k_pmc_lib_gemm.txt).  Variant 11 takes that discipline: every ds_read / LDS-DMA load sits ALONE
in the gap behind an MFMA, the loads are buffer loads with scalar offsets, and the lgkmcnt waits are the minimal counted ones —
computed here from the order in which the reads are issued.

A k-tile t as the wave executes it (s_memtime of the first version: 85 barrier + 255 from the barrier to the first MFMA + 1234 + 1052):
  top (in the .hip)   vmcnt(0) + barrier: k-tile t has landed, everyone is done READING k-tile t-1
  six fragment reads  of k-step 0 of t (W0 A0 A1 A2 A3 W1)
  TAIL of k-tile t-1  the last TAIL_GROUPS x 4 MFMAs of its k-step 1 — they need registers only, so they run in the shadow of those
                      reads' latency (the bubble between the barrier and the first MFMA of a k-tile was 255 cycles = 10 %);
                      more reads of k-step 0 ride in their gaps
  k-step 0 of t       64 MFMAs; in their gaps: the remaining reads of k-step 0 (just in time), the wave's 16 LDS-DMA loads of k-tile
                      t+1 (M0 written one gap before each), the 16 reads of k-step 1
  k-step 1 of t       its first 64 - 4 TAIL_GROUPS MFMAs
`tail` = the tail alone (after the last k-tile of an output tile); the FIRST k-tile of an output tile runs the same code with the
tail's fragments zeroed (16 MFMAs that add nothing: no second code path, no branch).
MFMA order as variant 7: token half h outer, feature block i inner (group g = 8 h + i: acc[i][4h .. 4h+3]).
usage: python tools/gen_gemm_v11_schedule.py                                       (the two shipped schedules + the tail)
       python tools/gen_gemm_v11_schedule.py TAIL_GROUPS G_STRIDE G_START            (one experimental schedule: ktile_s<G_STRIDE>)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAIL_GROUPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G_STRIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 4
G_START = int(sys.argv[3]) if len(sys.argv) > 3 else 4


class Sched:
    def __init__(self):
        self.out, self.issued, self.returned_upto = [], [], -1

    def emit(self, s):
        self.out.append('    ' + s + ' V11_SB;')

    def read(self, ks, op, blk):
        name = f'f{ks}{op}[{blk}]'
        self.emit(f'v11_rd<{blk * 2048}>({name}, {("wb" if op == "w" else "ab") + str(ks)});')
        self.issued.append(name)

    def wait_for(self, names):
        """s_waitcnt lgkmcnt(n): every read in `names` has returned (n = reads issued after the youngest of them) — only when an
        earlier wait does not already imply it.  Names never read in this section are in registers since the previous k-tile."""
        idx = [len(self.issued) - 1 - self.issued[::-1].index(n) for n in names if n in self.issued]
        if not idx or max(idx) <= self.returned_upto:
            return
        n = len(self.issued) - 1 - max(idx)
        assert n <= 15, 'lgkmcnt is a 4-bit field'
        self.emit(f'v11_wait<{n}>();')
        self.returned_upto = max(idx)

    def mfma(self, ks, i, j):
        self.emit(f'v11_mfma(acc[{i}][{j}], f{ks}w[{i}], f{ks}a[{j}]);')


def needs(ks, g):
    h, i = g >> 3, g & 7
    return [f'f{ks}w[{i}]'] + [f'f{ks}a[{4 * h + jj}]' for jj in range(4)]


def tail_only():
    """the stand-alone tail uses the BUILTIN MFMA: the compiler then knows which accumulators are still in the matrix pipe when
    the epilogue starts reading them (it cannot see inside the inline-asm MFMAs of the loop)."""
    s = Sched()
    for g in range(16 - TAIL_GROUPS, 16):
        for jj in range(4):
            i, j = g & 7, 4 * (g >> 3) + jj
            s.emit(f'acc[{i}][{j}] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1w[{i}], f1a[{j}], acc[{i}][{j}], 0, 0, 0);')
    return s


def ktile(with_tail=True):
    """the body of a k-tile in execution order: tail of the previous k-tile (4 TAIL_GROUPS MFMAs), k-step 0 (64), the head of k-step 1;
    every other instruction goes into a numbered gap (gap k = behind the k-th MFMA of the body)."""
    s = Sched()
    T = 4 * TAIL_GROUPS
    body = [(1, g) for g in range(16 - TAIL_GROUPS, 16)] + [(0, g) for g in range(16)] + [(1, g) for g in range(16 - TAIL_GROUPS)]
    gaps = {}

    def put(k, ev, front=False):
        gaps.setdefault(k, [])
        gaps[k].insert(0, ev) if front else gaps[k].append(ev)
    # the first six reads of this k-tile: in the tail's gaps, every other one.  (Issued in a row right behind the barrier, the four
    # waves' 24 reads queue at the LDS pipe — 8 cycles each — and a wave cannot issue its tail MFMAs until its own six are accepted.)
    first0 = [('w', 0), ('a', 0), ('a', 1), ('a', 2), ('a', 3), ('w', 1)]
    for n, item in enumerate(first0):
        put(1 + 2 * n, ('R0',) + item)
    # the other ten reads of k-step 0, two groups ahead of their first use (W2..W7 at groups 2..7 of k-step 0, A4..A7 at group 8)
    late0 = [('w', b) for b in range(2, 8)] + [('a', b) for b in range(4, 8)]
    for n, item in enumerate(late0):
        need = T + 4 * (2 + n) if item[0] == 'w' else T + 32
        put(max(13 + 2 * n, min(need - 10, T + 2 * n)), ('R0',) + item)
    # the 16 LDS-DMA loads of the NEXT k-tile, as early as the barrier allows (its stage is free from the top of this k-tile on):
    # the later they are issued the longer the next barrier waits for them (412 cycles at K = 13 824 when the last one went out 48
    # MFMAs before it); M0 is written one gap before each load
    for n in range(16):
        k = G_START + G_STRIDE * n
        put(k - 1, ('M0', n), front=True)
        put(k, ('G', n))
    # the 16 reads of k-step 1: behind the tail (they overwrite its operands), in the free gaps of k-step 0
    reads1 = [('w', 0), ('a', 0), ('a', 1), ('a', 2), ('a', 3)] + [('w', b) for b in range(1, 8)] + [('a', b) for b in range(4, 8)]
    k = T + 3
    for item in reads1:
        while any(e[0] in ('R0', 'R1', 'G') for e in gaps.get(k, [])):
            k += 1
        put(k, ('R1',) + item)
        k += 3
    assert k <= T + 64 + 3, k
    marks = {T: 'V11_T(1);', T + 64: 'V11_T(2);'}
    for k, (ks, g) in enumerate(body):
        if 4 * k in marks:
            pass
        s.wait_for(needs(ks, g)) if not (ks == 1 and k < TAIL_GROUPS) else None
        if 4 * k in marks:
            s.emit(marks[4 * k])
        for jj in range(4):
            s.mfma(ks, g & 7, 4 * (g >> 3) + jj)
            for e in gaps.get(4 * k + jj, []):
                if e[0] == 'M0':
                    s.emit(f'V11_M0({e[1]});')
                elif e[0] == 'G':
                    s.emit(f'V11_G({e[1]});')
                else:
                    s.read(0 if e[0] == 'R0' else 1, e[1], e[2])
    # whatever the tail (next k-tile, or the stand-alone one) needs must have returned: it waits for nothing
    s.wait_for([n for g in range(16 - TAIL_GROUPS, 16) for n in needs(1, g)])
    return s


if len(sys.argv) == 1:
    # the two schedules the library ships: loads every 6th gap (K <= 8192: 2528 cycles per k-tile at N = K = 5120 against 2611 with
    # every 4th) and every 4th gap (K > 8192, ffn.2: its A rows are 27 KiB apart and the loads need the extra time to land —
    # +3.1 % over variant 8 against +1.3 % with every 6th; profiles/r04n_gemm_v11.log)
    sections = []
    for G_STRIDE in (6, 4):
        sections.append((f'ktile_s{G_STRIDE}', ktile(), G_STRIDE))
    sections.append(('tail', tail_only(), 0))
else:
    sections = [('ktile_s%d' % G_STRIDE, ktile(), G_STRIDE), ('tail', tail_only(), 0)]
for name, sch, G_STRIDE in sections:
    path = os.path.join(os.environ.get('MG_V11_GEN_DIR', os.path.join(ROOT, 'moviigen1.1_amd', 'csrc')), f'gemm_bf16_v11_{name}.inc')
    with open(path, 'w') as f:
        f.write(f'// GENERATED by tools/gen_gemm_v11_schedule.py (TAIL_GROUPS = {TAIL_GROUPS}, G_STRIDE = {G_STRIDE}) — do not edit.\n'
                f'// GEMM variant 11, section `{name}`: every non-MFMA instruction alone in the gap behind an MFMA, counted lgkmcnt waits.\n')
        f.write('\n'.join(sch.out) + '\n')
    o = sch.out
    print(name, len(o), 'statements;', sum('mfma' in x for x in o), 'MFMAs,', sum('v11_rd' in x for x in o), 'reads,', sum('V11_G' in x for x in o), 'loads,',
          sum('v11_wait' in x for x in o), 'waits')
