"""Order-preserving summary of one kernel's ISA: runs of global / flat loads, stores, LDS reads / writes and MFMAs between the waits, barriers and branches, so
that one can SEE where the compiler put a request relative to its use (round 6: prefetch requests sunk to right in front of their use, weight images fetched
with FLAT loads -- DESIGN.md section 6).  No GPU needed.
usage: python tools/isa_trace.py <file stem, e.g. pair> <substring of the mangled kernel name> [first line] [last line]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from intel_sigir2023_amd import build as B      # noqa: E402

stem, pat = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
src = os.path.join(ROOT, 'intel_sigir2023_amd', 'csrc', stem + '.hip')
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, 'k.s')
    subprocess.run([B._hipcc()] + B.FLAGS + ['-x', 'hip', '--cuda-device-only', '-S', src, '-o', out], check=True, capture_output=True)
    lines = open(out).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(pat) + r'\S*:', l))
kinds = [('gload', re.compile(r'\b(global_load|buffer_load)')), ('FLAT-load', re.compile(r'\bflat_load')), ('gstore', re.compile(r'\b(global_store|buffer_store)')),
         ('FLAT-store', re.compile(r'\bflat_store')), ('atomic', re.compile(r'_atomic_')), ('lds-read', re.compile(r'\bds_read')), ('lds-write', re.compile(r'\bds_write')),
         ('mfma-bf16', re.compile(r'v_mfma_f32_16x16x(16|32)_?bf16')), ('mfma-f32', re.compile(r'v_mfma_f32_\d+x\d+x\d+_?f32')), ('scratch', re.compile(r'\bscratch_'))]
marks = re.compile(r'\b(s_waitcnt|s_barrier|s_cbranch\w*|s_endpgm)\b|^\.LBB')
print(lines[start][:120])
run, n, first = None, 0, 0


def flush():
    global run, n
    if run:
        print('%6d   %s x %d' % (first, run, n))
    run, n = None, 0


for i, l in enumerate(lines[start + 1:], 1):
    k = next((name for name, rx in kinds if rx.search(l)), None)
    if k:
        if k != run:
            flush()
            run, first = k, i
        n += 1
    elif marks.search(l):
        flush()
        if lo <= i <= hi:
            print('%6d %s' % (i, l.strip().split(';')[0][:70]))
        if 's_endpgm' in l:
            break
