"""Per kernel of a `rocprofv3 --kernel-trace --output-format csv` run: LDS per workgroup, registers per lane, workgroup size, grid -- what decides which kernels can share a CU.
usage: python tools/kernel_resources.py <dir with *_kernel_trace.csv>"""
import csv, glob, re, sys
seen = {}
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'\(.*', '', n)
        k = (n, r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['Workgroup_Size_X'])
        e = seen.setdefault(k, [0, 0, 0])
        e[0] += 1
        e[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        e[2] = max(e[2], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
print('%-62s %8s %5s %5s %5s %7s %9s' % ('kernel', 'LDS B', 'VGPR', 'AGPR', 'WG', 'max WGs', 'total us'))
for k, e in sorted(seen.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%-62s %8s %5s %5s %5s %7d %9.0f' % (k[0][:62], k[1], k[2], k[3], k[4], e[2], e[1] / 1e3))
