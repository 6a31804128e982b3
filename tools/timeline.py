"""Timeline of one training step from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`).

usage: python tools/timeline.py <dir with *_kernel_trace.csv> [first_step_kernel=bpr_loss_kernel]

Splits the trace into steps at every launch of the loss kernel, then for the middle steps reports the wall time, the
union of the kernel intervals (GPU busy), the summed kernel time (> union when branches overlap on several streams), the
idle gaps and the kernels that run alone for longest -- what bounds the step when the sum is larger than the wall.
"""
import csv
import glob
import os
import re
import sys
from collections import Counter


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return n.split('(')[0].split('<')[0]


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'bpr_loss_kernel'
    f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '0')))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2] == marker]
    if len(marks) < 4:
        raise SystemExit('fewer than 4 steps in the trace')
    # one step = from a loss kernel to the next one (same phase of consecutive steps)
    steps = [(marks[i], marks[i + 1]) for i in range(1, len(marks) - 1)]
    wall, busy, total = [], [], []
    alone, gaps_after = Counter(), Counter()
    conc = Counter()
    for lo, hi in steps:
        seg = rows[lo:hi]
        t0, t1 = seg[0][0], rows[hi][0]
        wall.append(t1 - t0)
        total.append(sum(min(e, t1) - s for s, e, _, _ in seg))
        ev = []
        for s, e, n, q in seg:
            ev.append((s, 1, n))
            ev.append((min(e, t1), -1, n))
        ev.sort(key=lambda x: (x[0], x[1]))
        live, last, b = [], t0, 0
        prev_name = seg[0][2]
        for t, k, n in ev:
            if live:
                b += t - last
                conc[min(len(live), 4)] += t - last
                if len(live) == 1:
                    alone[live[0]] += t - last
            elif t > last:
                gaps_after[prev_name] += t - last
            if k == 1:
                live.append(n)
            else:
                live.remove(n)
                prev_name = n
            last = t
        busy.append(b)
    n = len(steps)
    print('%d steps: wall %.3f ms, GPU busy (union) %.3f ms, summed kernel time %.3f ms, idle %.3f ms' %
          (n, sum(wall) / n / 1e6, sum(busy) / n / 1e6, sum(total) / n / 1e6, (sum(wall) - sum(busy)) / n / 1e6))
    print('time with k kernels in flight (ms/step):', {k: round(v / n / 1e6, 3) for k, v in sorted(conc.items())})
    print('running alone (ms/step):')
    for k, v in alone.most_common(25):
        print('   %-36s %.3f' % (k, v / n / 1e6))
    print('idle gaps after (ms/step):')
    for k, v in gaps_after.most_common(12):
        print('   %-36s %.3f' % (k, v / n / 1e6))


if __name__ == '__main__':
    main()
