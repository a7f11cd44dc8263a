#!/usr/bin/env python3
"""Summaries out of rocprofv3's rocpd sqlite output (ROCm 7.2 writes <pid>_results.db unless --output-format csv).
  rocpd_summary.py stats <results.db> <out.csv>      per-kernel calls / total / average / min / max (ns) / percentage,
                                                     the table `rocprofv3 --kernel-trace --stats` prints
  rocpd_summary.py traffic <out.json> <fetch.db> <write.db> [<fetch.db> <write.db> ...]
                                                     HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per kernel
                                                     (gfx950 correction of MI355X_MICROARCH.md, HBM section)"""
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
            if k in write and any(t in k for t in ('conv_', 'wgrad')):
                res[mangled_like(k)] = int((2 * fetch[k] + write[k]) * 1024)
    for k in list(res):  # bench.py's names omit the trailing default template arguments <.., ABL = 0, NW = 4, WT = 32>
        for tail in ('ELi0ELi4ELi32E', 'ELi0ELi4E'):
            if k.startswith('conv_f32_kernel') and k.endswith(tail):
                res[k[:-len(tail) + 1]] = res[k]
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        traffic(sys.argv[2], sys.argv[3:])
