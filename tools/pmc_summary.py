#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each collected in its own run with
--kernel-trace only) into per-kernel HBM traffic per launch.

Corrections (MI355X_MICROARCH.md, "HBM"): counter unit = KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide coalesced streaming read, so reads are doubled; WRITE_SIZE is exact for 16-byte stores.

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').strip()
        name = re.sub(r'<.*>', '', name)
        d[name][0] += 1
        d[name][1] += float(r['Counter_Value'])
    return d


def main():
    f = agg(sys.argv[1], 'FETCH_SIZE')
    w = agg(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(f) | set(w)):
        nf, nw = max(1, f[k][0]), max(1, w[k][0])
        rd = 2.0 * f[k][1] * 1024.0 / nf          # gfx950: FETCH_SIZE x2
        wr = w[k][1] * 1024.0 / nw
        out[k] = {'launches': max(f[k][0], w[k][0]), 'read_bytes_per_launch': round(rd), 'write_bytes_per_launch': round(wr),
                  'hbm_bytes_per_launch': round(rd + wr)}
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of `python3 bench.py --steps 3 --warmup 1 '
                       '--no_cpu_baseline --no_roofline`; KiB units; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B)',
               'kernels': out}, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
        print('%-34s %5d launches  %8.1f MB read  %8.1f MB written per launch' % (k, v['launches'], v['read_bytes_per_launch'] / 1e6, v['write_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
