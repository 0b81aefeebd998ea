"""Summarise a rocprofv3 rocpd sqlite result (…_results.db) into the text table committed under
profiles/.   python tools/rocprof_summary.py <db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)           # drop the argument list
    name = name.replace('void ', '')
    return name if len(name) <= 70 else name[:67] + '...'


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    lines = ['# rocprofv3 --kernel-trace --stats summary (durations in microseconds)',
             f'# source: {sys.argv[1]}',
             f'{"kernel":70s} {"calls":>7s} {"total_us":>14s} {"avg_us":>12s} {"pct":>7s}']
    for name, calls, tot, avg, pct in rows[:30]:
        lines.append(f'{short(name):70s} {calls:7d} {tot:14.1f} {avg:12.1f} {pct:7.2f}')
    # split the attention kernel into self-attention (long) and cross-attention (512 keys) launches
    try:
        q = ("select (end-start)/1000.0 from kernels where name like '%attn_hd128%'")
        d = [r[0] for r in cur.execute(q)]
        big = [x for x in d if x > 10000]
        small = [x for x in d if x <= 10000]
        if big and max(big) > 2 * min(big):
            # two kinds of long launches in one trace: the workload's (all heads) and the box calibration's (8 heads, bench.py without --no-calibration)
            cut = (max(big) + min(big)) / 2
            cal = [x for x in big if x <= cut]
            big = [x for x in big if x > cut]
            lines.append(f'# (box-calibration launches of the same kernel, fewer heads: n={len(cal)} avg_us={sum(cal)/len(cal):.1f} — not part of a step)')
        if big:
            lines.append(f'# mg_attn_fwd_bf16_hd128 self-attention launches (all video keys): n={len(big)} avg_us={sum(big)/len(big):.1f}')
        if small:
            lines.append(f'# mg_attn_fwd_bf16_hd128 cross-attention launches (512 keys)   : n={len(small)} avg_us={sum(small)/len(small):.1f}')
    except sqlite3.Error as e:  # schema differences between rocprofv3 versions
        lines.append(f'# (per-launch split unavailable: {e})')
    text = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text)
    print(text)


if __name__ == '__main__':
    main()
