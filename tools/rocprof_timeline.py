"""One training step as the GPU really ran it, from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py (kernel start / end timestamps
per hardware queue; no HIP events in the stream, so small-kernel chains are not inflated the way tools/step_timeline.py's are).
usage: python tools/rocprof_timeline.py <dir with *_kernel_trace.csv> [step index from the end = 2] [marker = bpr_loss_kernel]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marker = sys.argv[3] if len(sys.argv) > 3 else 'bpr_loss_kernel'
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']), r['Kernel_Name']))
rows.sort()


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', re.sub(r'<.*?>', '', n)).strip()


marks = [i for i, r in enumerate(rows) if short(r[3]) == marker]
lo, hi = marks[-back - 1], marks[-back]
seg = rows[lo:hi]
t0 = seg[0][0]
wall = rows[hi][0] - t0
# union / concurrency
ev = sorted([(s, 1) for s, e, q, n in seg] + [(min(e, rows[hi][0]), -1) for s, e, q, n in seg])
busy, depth, last, conc = 0, 0, t0, {}
for t, dlt in ev:
    if depth > 0:
        busy += t - last
    conc[depth] = conc.get(depth, 0) + (t - last)
    depth += dlt
    last = t
print('step wall %.1f us, busy (union) %.1f us, idle %.1f us, summed %.1f us' % (wall / 1e3, busy / 1e3, (wall - busy) / 1e3, sum(e - s for s, e, q, n in seg) / 1e3))
print('us with k kernels in flight:', {k: round(v / 1e3, 1) for k, v in sorted(conc.items())})
qs = sorted(set(q for s, e, q, n in seg))
for s, e, q, n in seg:
    print('%8.1f %7.1f  q%d %s' % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), short(n)))
