#!/bin/bash
# SQ counter passes over one small program (default tools/pair_bench.py): where a kernel's wave cycles go.
# usage (on the GPU box): tools/pmc_kernel.sh <tag> <kernel-name-substring> -- python3 tools/pair_bench.py ...
tag=$1; pat=$2; shift 3
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p1 -- "$@" > $out/p1.log 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $out/p2 -- "$@" > $out/p2.log 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $out/p3 -- "$@" > $out/p3.log 2>&1 < /dev/null
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
for p in ('p1', 'p2', 'p3'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, p), recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-60:]
                e = d[k][r['Counter_Name']]
                e[0] += 1; e[1] += float(r['Counter_Value'])
        for k, cs in d.items():
            print(p, k)
            for c, (n, v) in sorted(cs.items()):
                print('   %-28s %14.0f per launch (%d launches)' % (c, v / n, n))
PY
find $out -name "*.db" -delete 2>/dev/null
