#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE counter_collection.csv files):
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, with the gfx950 correction of MI355X_MICROARCH.md (HBM
section: FETCH_SIZE tallies 128-byte requests at 64 B -> double it; both counters are in KB), keyed by the kernel
names bench.py / sr_kernel_name use (e.g. conv_f32_kernelILi1ELi4ELi3ELb0E).
usage: make_traffic.py out.json fetch.csv write.csv [fetch2.csv write2.csv ...]"""
import csv, collections, json, re, sys


def mean_per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] == counter:
                acc[row['Kernel_Name']].append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def mangled_like(name):
    """'void (anonymous namespace)::conv_f32_kernel<1, 4, 3, false, 0, 4>(...)' -> 'conv_f32_kernelILi1ELi4ELi3ELb0ELi0ELi4E'"""
    m = re.search(r'(\w+)<([^>]*)>', name)
    if not m:
        m2 = re.search(r'(\w+)\(', name)
        return m2.group(1) if m2 else name
    parts = []
    for a in m.group(2).split(','):
        a = a.strip()
        parts.append('Lb1E' if a == 'true' else 'Lb0E' if a == 'false' else f'Li{a}E')
    return m.group(1) + 'I' + ''.join(parts)


if __name__ == '__main__':
    out = {}
    args = sys.argv[2:]
    for i in range(0, len(args), 2):
        fetch, write = mean_per_kernel(args[i], 'FETCH_SIZE'), mean_per_kernel(args[i + 1], 'WRITE_SIZE')
        for k in fetch:
            if k in write and any(t in k for t in ('conv_', 'wgrad')):
                out[mangled_like(k)] = int((2 * fetch[k] + write[k]) * 1024)
    # bench.py looks kernels up by sr_kernel_name(), which omits trailing default template arguments: add prefix aliases
    for k in list(out):
        for cut in ('ELi0ELi4E',):
            if k.endswith(cut):
                out[k[:-len(cut) + 1]] = out[k]
    json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))
