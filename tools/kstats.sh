#!/bin/bash
# usage: tools/kstats.sh <tag> <filter> -- <python script args...>
# Runs `rocprofv3 --kernel-trace --stats` on a python command (program directly after `--`), prints the
# kernel_stats.csv lines matching <filter>.  Bounded by timeout; never reads stdin.
tag=$1; filt=$2; shift 3
out=gpurun_out/ks_$tag
mkdir -p $out
( cd /tmp; export TMPDIR=/tmp )
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/log.txt 2>&1 < /dev/null
f=$(find $out -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no stats file"; tail -5 $out/log.txt; exit 0; fi
grep -E "$filt" "$f" < /dev/null | awk -F, '{printf "%-70s calls %s avg_us %.1f\n", substr($1,1,70), $2, $4/1000}'
