#!/usr/bin/env python3
"""Summaries out of rocprofv3's rocpd sqlite output (ROCm 7.2 writes <pid>_results.db unless --output-format csv).
  rocpd_summary.py stats <results.db> <out.csv>      per-kernel calls / total / average / min / max (ns) / percentage,
                                                     the table `rocprofv3 --kernel-trace --stats` prints
  rocpd_summary.py traffic <out.json> <fetch.db> <write.db> [<fetch.db> <write.db> ...]
                                                     HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per kernel
                                                     (gfx950 correction of MI355X_MICROARCH.md, HBM section)
  rocpd_summary.py mfma <out.json> <mfma_busy.db> <gui_active.db> [...]
                                                     MfmaUtil per kernel = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 * 1024
                                                     SIMDs); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs"""
import csv, json, sqlite3, sys
from make_traffic import mangled_like


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels '
                     'group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), r[4], r[5], round(100.0 * r[2] / total, 3)])
    for r in rows[:8]:
        print(f'{100.0 * r[2] / total:6.2f}%  calls {r[1]:6d}  avg {r[3] / 1e3:9.2f} us  {r[0][:110]}')


def counter_means(db, counter):
    c = sqlite3.connect(db)
    return dict(c.execute('select kernel_name, avg(value) from counters_collection where counter_name = ? group by kernel_name',
                          (counter,)).fetchall())


def traffic(out, dbs):
    res = {}
    for i in range(0, len(dbs), 2):
        fetch, write = counter_means(dbs[i], 'FETCH_SIZE'), counter_means(dbs[i + 1], 'WRITE_SIZE')
        for k in fetch:
            if k in write and any(t in k for t in ('conv_', 'wgrad', 'rdb_')):
                res[mangled_like(k)] = int((2 * fetch[k] + write[k]) * 1024)
    add_aliases(res)
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


def add_aliases(res):
    """bench.py's kernel names (sr_kernel_name) omit trailing default template arguments: conv_f32_kernel<.., ABL = 0, NW = 4,
    WT = 32> and conv_bf16_kernel<.., S2 = false>."""
    for k in list(res):
        for tail in ('ELi0ELi4ELi32E', 'ELi0ELi4E'):
            if k.startswith('conv_f32_kernel') and k.endswith(tail):
                res[k[:-len(tail) + 1]] = res[k]
        if k.startswith('conv_bf16_kernel') and k.count('EL') == 4 and k.endswith('ELb0E'):
            res[k[:-4]] = res[k]


def per_dispatch(db, counter, how):
    c = sqlite3.connect(db)
    rows = c.execute(f'select kernel_name, dispatch_id, {how}(value) from counters_collection where counter_name = ? '
                     'group by kernel_name, dispatch_id', (counter,)).fetchall()
    agg = {}
    for k, _, v in rows:
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += v
    return {k: a[1] / a[0] for k, a in agg.items()}


def mfma(out, dbs):
    res = {}
    for i in range(0, len(dbs), 2):
        busy, gui = per_dispatch(dbs[i], 'SQ_VALU_MFMA_BUSY_CYCLES', 'sum'), per_dispatch(dbs[i + 1], 'GRBM_GUI_ACTIVE', 'max')
        for k in busy:
            if k in gui and any(t in k for t in ('conv_', 'wgrad', 'rdb_')):
                res[mangled_like(k)] = {'mfma_util': round(busy[k] / (gui[k] / 8 * 1024), 4), 'cycles_per_launch': int(gui[k] / 8)}
    add_aliases(res)
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == '__main__':
    if sys.argv[1] == 'mfma':
        mfma(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'timeline':
        pass   # defined below
    else:
        traffic(sys.argv[2], sys.argv[3:])


def timeline(db, out_txt, marker='adam_kernel'):
    """Per-queue timeline of the LAST complete training step in a kernel trace (a step = the dispatches between two consecutive
    last-Adam launches): wall time, busy time per hardware queue (the caller's stream and the library's second lane land on
    different queues), time during which two queues run kernels at once, and the average duration of the kernels that ran on the
    second queue — the numbers that say whether work moved under other work or merely beside it."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)').fetchall()]
    qcol = next((k for k in ('queue_id', 'queue', 'stream_id', 'stream') if k in cols), None)
    lines = [f'columns of `kernels`: {cols}', f'queue column: {qcol}']
    if not qcol or 'start' not in cols or 'end' not in cols:
        open(out_txt, 'w').write('\n'.join(lines) + '\n')
        print('\n'.join(lines))
        return
    rows = c.execute(f'select name, start, end, {qcol} from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    # two Adam launches per GAN step (G and D): a step ends at every second one
    ends = marks[1::2] if len(marks) >= 4 else marks
    if len(ends) < 2:
        lines.append('fewer than two steps in the trace')
        open(out_txt, 'w').write('\n'.join(lines) + '\n')
        return
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0, t1 = min(r[1] for r in step), max(r[2] for r in step)
    lines.append(f'last step: {len(step)} dispatches, wall {(t1 - t0) / 1e6:.3f} ms, kernel time summed {sum(r[2] - r[1] for r in step) / 1e6:.3f} ms')
    queues = {}
    for r in step:
        queues.setdefault(r[3], []).append(r)
    for q, rs in sorted(queues.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        busy = sum(r[2] - r[1] for r in rs)
        names = {}
        for r in rs:
            names[r[0][:60]] = names.get(r[0][:60], 0) + (r[2] - r[1])
        top = ', '.join(f'{k.split("(")[0][-44:]} {v / 1e6:.2f} ms' for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:3])
        lines.append(f'queue {q}: {len(rs)} dispatches, busy {busy / 1e6:.3f} ms, first {(rs[0][1] - t0) / 1e6:.3f} ms, last end {(max(r[2] for r in rs) - t0) / 1e6:.3f} ms; {top}')
    # overlap: sweep
    ev = []
    for r in step:
        ev.append((r[1], 1, r[3]))
        ev.append((r[2], -1, r[3]))
    ev.sort()
    active, last, both, any_ = {}, t0, 0, 0
    for t, d, q in ev:
        n_active = sum(1 for v in active.values() if v > 0)
        if n_active >= 1:
            any_ += t - last
        if n_active >= 2:
            both += t - last
        last = t
        active[q] = active.get(q, 0) + d
    lines.append(f'some queue busy {any_ / 1e6:.3f} ms, two or more queues busy at once {both / 1e6:.3f} ms, idle {(t1 - t0 - any_) / 1e6:.3f} ms')
    for key in ('wgrad_rdb_bf16', 'rdb_fused'):
        ds = [r[2] - r[1] for r in step if key in r[0]]
        if ds:
            lines.append(f'{key}: {len(ds)} launches, average {sum(ds) / len(ds) / 1e3:.1f} us, total {sum(ds) / 1e6:.2f} ms')
    open(out_txt, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'timeline':
    timeline(sys.argv[2], sys.argv[3])
