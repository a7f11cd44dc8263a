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
    else:
        traffic(sys.argv[2], sys.argv[3:])
