#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files per kernel name (mean per dispatch)."""
import csv
import collections
import sys


def summarise(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row['Kernel_Name']
            agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
    return agg


if __name__ == '__main__':
    for p in sys.argv[1:]:
        print('#', p)
        for k, cs in summarise(p).items():
            if not any(t in k for t in ('conv_', 'conv3x3', 'wgrad', 'pack_table', 'reduce')):
                continue
            short = k.split('(')[0][-60:] if 'anonymous' not in k else k[k.index('::') + 2:k.index('>(') + 1]
            print(short, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
