#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each collected in its own run with --kernel-trace only) into
per-kernel HBM traffic per launch -- one pair of passes over TRAINING steps only and, optionally, one pair over EVALUATION
steps only -- plus the per-step totals.

Corrections (MI355X_MICROARCH.md, "HBM"): counter unit = KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide coalesced streaming read, so reads are doubled; WRITE_SIZE is exact for 16-byte stores.

usage: pmc_summary.py <train fetch csv> <train write csv> <train steps> <out.json> [<eval fetch csv> <eval write csv> <eval steps>]
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


def summarise(fetch_csv, write_csv, steps):
    f = agg(fetch_csv, 'FETCH_SIZE')
    w = agg(write_csv, 'WRITE_SIZE')
    out = {}
    total = 0.0
    for k in sorted(set(f) | set(w)):
        nf, nw = max(1, f[k][0]), max(1, w[k][0])
        rd = 2.0 * f[k][1] * 1024.0 / nf          # gfx950: FETCH_SIZE x2
        wr = w[k][1] * 1024.0 / nw
        n = max(f[k][0], w[k][0])
        out[k] = {'launches': n, 'launches_per_step': round(n / steps, 2), 'read_bytes_per_launch': round(rd),
                  'write_bytes_per_launch': round(wr), 'hbm_bytes_per_launch': round(rd + wr)}
        # torch's own kernels (at::native / elementwise_kernel / rocprim sorts) and the runtime's buffer copies / fills in these runs come from the
        # synthetic batch generation and the model initialisation BEFORE the steps (the engine's step launches none: their launch counts are the
        # same in a 4-step and a 13-step run, profiles/r04_bench_tmall_kernel_stats.csv): listed, not counted into the per-step total
        if not (k.startswith('at::native') or 'elementwise_kernel' in k or k.startswith('at::') or k.startswith('rocprim') or k.startswith('__amd_rocclr')):
            total += (rd + wr) * n
    return out, total / steps


def main():
    steps = int(sys.argv[3])
    out, per_step = summarise(sys.argv[1], sys.argv[2], steps)
    res = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each in its own run (--kernel-trace only) of `python3 bench.py [--dtype bf16] --steps 3 --warmup 1 '
                   '--eval_steps 0 --no_cpu_baseline --no_roofline --no_feed --no_bf16_line` (%d training steps, no evaluation steps); KiB units; '
                   'FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B)' % steps,
           'train_steps': steps, 'hbm_GB_per_train_step': round(per_step / 1e9, 3), 'kernels': out}
    if len(sys.argv) > 7:
        esteps = int(sys.argv[7])
        eout, eper = summarise(sys.argv[5], sys.argv[6], esteps)
        res.update({'eval_steps': esteps, 'hbm_GB_per_eval_step': round(eper / 1e9, 3), 'kernels_eval': eout})
    # what was measured: the commit the caller names (INTEL_COMMIT: the GPU box has no .git) and the hash of the kernel sources of the build that ran
    import os
    res['commit'] = os.environ.get('INTEL_COMMIT')
    if not res['commit']:
        try:
            import subprocess
            res['commit'] = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
        except Exception:
            res['commit'] = 'unknown'
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
        from intel_sigir2023_amd import build as _b
        res['csrc_stamp'] = _b._stamp()
    except Exception:
        res['csrc_stamp'] = 'unknown'
    json.dump(res, open(sys.argv[4], 'w'), indent=1, sort_keys=True)
    print('HBM traffic per training step: %.2f GB' % (per_step / 1e9))
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:14]:
        print('%-34s %6.1f launches/step  %8.1f MB read  %8.1f MB written per launch' % (k, v['launches_per_step'], v['read_bytes_per_launch'] / 1e6, v['write_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
