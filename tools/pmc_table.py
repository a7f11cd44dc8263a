"""Per-kernel means of every counter found in rocprofv3 --pmc output directories (rocpd sqlite or counter_collection csv)."""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def read(d):
    acc = defaultdict(lambda: defaultdict(list))
    for db in glob.glob(os.path.join(d, '**', '*.db'), recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = c.execute('select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by kernel_name, counter_name, dispatch_id').fetchall()
        except sqlite3.Error:
            continue
        for k, n, _, v in rows:
            acc[k][n].append(v)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        per = defaultdict(float)
        for row in csv.DictReader(open(f)):
            per[(row['Kernel_Name'], row['Counter_Name'], row['Dispatch_Id'])] += float(row['Counter_Value'])
        for (k, n, _), v in per.items():
            acc[k][n].append(v)
    return acc


def main():
    allk = defaultdict(dict)
    for d in sys.argv[1:]:
        for k, cs in read(d).items():
            for n, vals in cs.items():
                allk[k][n] = (sum(vals) / len(vals), len(vals))
    for k in sorted(allk, key=lambda k: -allk[k].get('SQ_BUSY_CYCLES', allk[k].get('GRBM_GUI_ACTIVE', (0, 0)))[0] * allk[k].get('SQ_BUSY_CYCLES', allk[k].get('GRBM_GUI_ACTIVE', (0, 1)))[1]):
        if not any(t in k for t in ('conv', 'wgrad', 'rdb_')):
            continue
        print(k[:120])
        for n in sorted(allk[k]):
            print(f'    {n:32s} {allk[k][n][0]:16.0f}   (dispatches {allk[k][n][1]})')


if __name__ == '__main__':
    main()
