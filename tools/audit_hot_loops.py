"""Static audit of the MFMA hot loops: compile a kernel source to gfx950 assembly and check, for every instantiation of
the named kernel, the basic block that holds its steady-state loop (the block with the most MFMAs):

  * no scratch access and no v_cndmask / v_readfirstlane chains in it (a runtime-indexed register array — e.g. a
    pointer array indexed by a value hipcc did not fold — turns into hundreds of selects per iteration: the round-3
    GEMM ran 7x slower in ONE epilogue instantiation that way, with every parity test green),
  * the expected MFMA count, and at most `max_other` other instructions.

usage: python tools/audit_hot_loops.py            (exit code 1 on a violation; prints one line per kernel)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'moviigen1.1_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

# source file, kernel name fragment, regex of instantiations to skip (the s_memtime profiling builds), MFMAs per
# iteration, max other instructions per iteration
KERNELS = [
    ('gemm_bf16_v8.hip', 'gemm_bf16_v8_kernel', r'kernelILi\dELb1E', 64, 125),
    ('gemm_bf16_v7.hip', 'gemm_bf16_v7_kernel', r'kernelILi\dELb1E', 128, 180),
    # variant 11: the k-tile is generated inline assembly (128 MFMAs, 32 LDS reads, 16 buffer loads to LDS + 16 M0 writes, 18
    # waits, 17 s_nop, one barrier = 116) plus the loop's own bookkeeping blocks; the limit checks that the compiler adds no more
    ('gemm_bf16_v11.hip', 'gemm_bf16_v11_kernel', r'kernelILi\dELi\dELb1E', 128, 165),
    # variant 12: ONE self-looping block — 128 MFMAs + 32 LDS reads + 16 loads + 16 M0 writes + 6 state toggles + 5 counted waits + 3 x
    # (wait + barrier) + ~17 s_nop 0 in front of the M0 writes + the loop's add / compare / branch = 101
    ('gemm_bf16_v12.hip', 'gemm_bf16_v12_kernel', r'kernelILi\dELi\dELb1E', 128, 108),
    # attention: 128 MFMAs + 64 exp + 64 adds + 32 cvt + 32 LDS reads + 8 buffer loads to LDS + waits / state rotation (235; the variant that
    # scales q in the loop: 299)
    ('attn_hd128_m16.hip', 'attn_hd128_m16_kernel', r'kernelILb1E', 128, 305),
]
FORBIDDEN = ('scratch_', 'v_cndmask', 'v_readfirstlane')
MAX_ACC_MOVES = 8      # v_accvgpr_read / _write per iteration (register-file shuffles of a kernel at the 512-register limit)


def device_asm(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', os.path.join(CSRC, src),
                        '-o', out], check=True, capture_output=True)
        return open(out).read()


def audit(src, frag, skip, n_mfma, max_other):
    text = device_asm(src)
    problems, report = [], []
    # split into functions
    funcs = re.split(r'^(_Z\w+):', text, flags=re.M)
    for name, body in zip(funcs[1::2], funcs[2::2]):
        if frag not in name:
            continue
        if re.search(skip, name):
            continue
        # the steady-state loop body = the basic block(s) inside a loop (LLVM marks their label lines "Loop") holding
        # exactly the iteration's MFMA count; the fewest other instructions wins when a peeled copy exists
        parts = re.split(r'^(\.LBB\d+_\d+:.*)$', body.split('.Lfunc_end')[0], flags=re.M)
        def loop_body(lab, blk):
            # a block's text runs on to the next label: cut it at its own back edge (what follows is the loop's exit path)
            me = lab.split(':')[0]
            lines = blk.splitlines()
            back = [i for i, ln in enumerate(lines) if re.search(r'\bs_cbranch\w*\s+' + re.escape(me) + r'\b', ln)]
            return '\n'.join(lines[:back[-1] + 1]) if back else blk
        # group the blocks by the INNERMOST loop they belong to (LLVM's label comments: "in Loop: Header=BBx_y Depth=d" on a
        # member, "This (Inner) Loop Header" on the header itself): a steady-state loop with a rarely taken branch inside
        # (e.g. the residual warm-up of the gated GEMM) spans several blocks
        loops = {}
        for lab, blk in zip(parts[1::2], parts[2::2]):
            me = lab.split(':')[0].lstrip('.L')
            head_txt = lab + '\n' + '\n'.join(blk.splitlines()[:3])
            m = re.search(r'in Loop: Header=(BB\d+_\d+)', lab)
            if 'Loop Header' in head_txt:
                hdr = me
            elif m:
                hdr = m.group(1)
            else:
                continue
            loops.setdefault(hdr, []).append(loop_body(lab, blk) if hdr == me else blk)
        cands = ['\n'.join(blks) for blks in loops.values()]
        cands = [blk for blk in cands if len(re.findall(r'\bv_mfma_', blk)) == n_mfma]
        if not cands:
            problems.append(f'{name}: no loop block with {n_mfma} MFMAs')
            continue
        best = min(cands, key=lambda b: sum(1 for ln in b.splitlines() if ln.startswith('\t') and not ln.strip().startswith((';', '.'))))
        ins = [ln.split()[0] for ln in best.splitlines() if ln.startswith('\t') and not ln.strip().startswith((';', '.'))]
        mf = sum(1 for i in ins if i.startswith('v_mfma_'))
        other = len(ins) - mf
        bad = sorted({i for i in ins if i.startswith(FORBIDDEN)})
        report.append(f'{name[:70]:70s} hot block: {mf} MFMA, {other} other instructions' + (f', FORBIDDEN {bad}' if bad else ''))
        if other > max_other:
            problems.append(f'{name}: {other} non-MFMA instructions in the hot block (limit {max_other})')
        if bad:
            problems.append(f'{name}: {bad} in the hot block')
        # a vmcnt wait the COMPILER put into a loop whose loads are opaque asm = a preheader reload carried around the back edge; it waits for the
        # loop's own refill loads (attention, round 5: step A 1275 -> 2150 cycles).  The hand-written waits are vmcnt(0) at a fence or counted
        # waits inside the generated GEMM bodies — those files are exempt.
        if src.startswith('attn_'):
            vm = [ln.strip() for ln in best.splitlines() if re.search(r'\bs_waitcnt\b.*vmcnt\((?!0\))', ln)]
            if vm:
                problems.append(f'{name}: partial vmcnt wait(s) in the hot block: {vm}')
        acc_moves = sum(1 for i in ins if i.startswith('v_accvgpr_'))
        if acc_moves > MAX_ACC_MOVES:
            problems.append(f'{name}: {acc_moves} v_accvgpr moves in the hot block (limit {MAX_ACC_MOVES})')
    if not report:
        problems.append(f'{src}: no kernel matching {frag}')
    return report, problems


def regs(op):
    """the VGPR numbers an assembly operand names: v7 -> {7}, v[4:7] -> {4, 5, 6, 7}"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', op)
    return {int(m.group(1))} if m else set()


def store_hazards(src):
    """A wide store directly followed by a PACKED-fp32 VALU instruction that overwrites its data registers: hipcc leaves no wait
    state there for ds_write* and for buffer stores with a scalar offset register, and on gfx950 the store picks up part of the new
    value in four lanes of every 16 (found in the row-wise epilogue of GEMM variant 11; csrc/gemm_bf16_v11.hip).  Scans EVERY kernel
    of a source file; returns a list of 'kernel: store / next instruction' strings."""
    out = []
    text = device_asm(src)
    funcs = re.split(r'^(_Z\w+|\w+_kernel\w*):', text, flags=re.M)
    for name, body in zip(funcs[1::2], funcs[2::2]):
        lines = [ln.strip() for ln in body.split('.Lfunc_end')[0].splitlines()
                 if ln.startswith('\t') and not ln.strip().startswith((';', '.'))]
        for a, b in zip(lines, lines[1:]):
            op = a.split()[0]
            if not op.startswith(('buffer_store_', 'global_store_', 'flat_store_', 'scratch_store_', 'ds_write')):
                continue
            if not b.startswith('v_pk_'):
                continue
            ops = [o.strip() for o in a[len(op):].split(',')]
            if op.startswith('buffer_store_'):
                data = regs(ops[0])
            elif op.startswith('ds_write'):
                data = set().union(*[regs(o.split()[0]) for o in ops[1:] if o.split()])
            else:
                data = regs(ops[1]) if len(ops) > 1 else set()
            dst = regs(b[len(b.split()[0]):].split(',')[0].strip())
            if len(data) >= 3 and data & dst:          # more than 64 bits of store data (64-bit stores: no failure observed)
                out.append(f'{name[:60]}: `{a}` directly followed by `{b}`')
    return out


def sgprs(op):
    m = re.fullmatch(r's\[(\d+):(\d+)\]', op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r's(\d+)', op)
    return {int(m.group(1))} if m else set()


def lane_read_hazards(src):
    """A VALU write of an SGPR (v_readfirstlane / v_readlane) followed within 5 wait states by a VMEM instruction that reads it — as a
    buffer resource or a scalar offset.  hipcc keeps that distance for the VMEM instructions it emits itself; an INLINE-ASSEMBLY load
    (the LDS-DMA loads of GEMM variants 11 / 12) is opaque to it, and a spill reload (v_readlane) placed directly in front of one would be
    5 wait states short.  Scans every kernel of a source file."""
    out = []
    text = device_asm(src)
    funcs = re.split(r'^(_Z\w+|\w+_kernel\w*):', text, flags=re.M)
    for name, body in zip(funcs[1::2], funcs[2::2]):
        lines = [ln.strip().split(';')[0].strip() for ln in body.split('.Lfunc_end')[0].splitlines()
                 if ln.startswith('\t') and not ln.strip().startswith((';', '.'))]
        for i, a in enumerate(lines):
            if not a.startswith(('v_readfirstlane_b32', 'v_readlane_b32')):
                continue
            dst = sgprs(a.split()[1].rstrip(','))
            states = 0
            for b in lines[i + 1:i + 8]:
                op = b.split()[0]
                if op.startswith(('buffer_', 'global_load_lds', 's_buffer')):
                    used = set()
                    for o in b[len(op):].replace(' offen', '').replace(' lds', '').split(','):
                        o = o.strip().split()[0] if o.strip() else ''
                        used |= sgprs(o)
                    if dst & used and states < 5:
                        out.append(f'{name[:60]}: `{a}` {states} wait states in front of `{b}`')
                    break
                states += (int(b.split()[1]) + 1) if op == 's_nop' else 1
                if states >= 5 or op.startswith(('s_cbranch', 's_branch', 's_endpgm')):
                    break
    return out


def exit_pad(src, frag, skip, n_mfma):
    """GEMM variant 12: the k-loop's MFMAs are inline assembly, so the compiler does not know their results are in flight when the loop
    ends; the source pads with `s_nop 15` x 2 right behind the loop.  Checks that no accumulator instruction stands between the loop's
    back edge and that pad, in every instantiation."""
    out = []
    text = device_asm(src)
    funcs = re.split(r'^(_Z\w+):', text, flags=re.M)
    for name, body in zip(funcs[1::2], funcs[2::2]):
        if frag not in name or re.search(skip, name):
            continue
        lines = [ln.strip().split(';')[0].strip() for ln in body.split('.Lfunc_end')[0].splitlines() if ln.strip() and not ln.strip().startswith(';')]
        found = False
        for i, ln in enumerate(lines):
            m = re.match(r'(\.LBB\d+_\d+):', ln)
            if not m:
                continue
            j = i + 1
            blk = []
            while j < len(lines) and not re.match(r'\.LBB\d+_\d+:', lines[j]):
                blk.append(lines[j])
                j += 1
            back = [k for k, x in enumerate(blk) if re.search(r's_cbranch\w*\s+' + re.escape(m.group(1)) + r'\b', x)]
            if not back or sum(1 for x in blk[:back[-1]] if x.startswith('v_mfma_')) != n_mfma:
                continue
            found = True
            # walk every control-flow path from the back edge's fall-through until it meets the pad
            label_at = {mm.group(1): k for k, x in enumerate(lines) for mm in [re.match(r'(\.LBB\d+_\d+):', x)] if mm}
            start = i + 1 + back[-1] + 1
            todo, seen = [start], set()
            while todo:
                k = todo.pop()
                while k < len(lines) and k not in seen:
                    seen.add(k)
                    x = lines[k]
                    if k == i:                  # back in the loop (the second pass through the same body): MFMA behind MFMA, as on the back edge
                        break
                    if x.startswith('s_nop 15'):
                        break
                    if x.startswith(('v_accvgpr', 'v_mfma')):
                        out.append(f'{name[:60]}: `{x}` between the k-loop and its exit pad')
                        break
                    mb = re.match(r's_(c?)branch\w*\s+(\.LBB\d+_\d+)', x)
                    if mb:
                        if mb.group(2) != m.group(1):
                            todo.append(label_at[mb.group(2)])
                        if not mb.group(1):
                            break
                    if x.startswith('s_endpgm'):
                        out.append(f'{name[:60]}: a path from the k-loop reaches s_endpgm without the pad')
                        break
                    k += 1
        if not found:
            out.append(f'{name[:60]}: no self-looping block with {n_mfma} MFMAs')
    return out


def accumulator_spills(src, frag, skip):
    """no accumulator register goes to scratch: `scratch_store ... a[..]` anywhere in a kernel whose MFMAs are (partly) inline assembly.  Round 6: with 48 registers freed
    in gemm_bf16_v12_kernel<fp32 store> the allocator parked a[128:131] in scratch between the MFMAs of the last k-tile that reuse them, and the tile came out wrong with
    the hot block untouched (test_gemm_epilogues[3-*])."""
    text = device_asm(src)
    out = []
    funcs = re.split(r'^(_Z\w+):', text, flags=re.M)
    for name, body in zip(funcs[1::2], funcs[2::2]):
        if frag not in name or re.search(skip, name):
            continue
        n = len(re.findall(r'scratch_store_dword\w*\s+off,\s*a\[?\d', body.split('.Lfunc_end')[0]))
        if n:
            out.append(f'{name[:60]}: {n} accumulator spill(s) to scratch')
    return out


def main():
    allp = []
    for k in KERNELS:
        rep, prob = audit(*k)
        print('\n'.join(rep))
        allp += prob
    n_src = 0
    for src in sorted(os.listdir(CSRC)):
        if src.endswith('.hip'):
            n_src += 1
            allp += ['store data hazard: ' + h for h in store_hazards(src)]
            allp += ['lane-read hazard: ' + h for h in lane_read_hazards(src)]
    print(f'store-data / packed-fp32 and lane-read / VMEM hazard scans: {n_src} source files')
    pads = exit_pad('gemm_bf16_v12.hip', 'gemm_bf16_v12_kernel', r'kernelILi\dELi\dELb1E', 128)
    print(f'variant 12 exit pad: {"ok" if not pads else "VIOLATED"}')
    allp += pads
    spills = accumulator_spills('gemm_bf16_v12.hip', 'gemm_bf16_v12_kernel', r'kernelILi\dELi\dELb1E')
    print(f'variant 12 accumulator spills: {"none" if not spills else "FOUND"}')
    allp += spills
    for p in allp:
        print('VIOLATION:', p)
    return 1 if allp else 0


if __name__ == '__main__':
    sys.exit(main())
