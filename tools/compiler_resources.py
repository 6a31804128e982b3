"""VGPR / SGPR / scratch (spill) counts of the large kernels as the COMPILER reports them (no GPU needed): each .hip file compiled with the library's own flags
plus -Rpass-analysis=kernel-resource-usage, one line per template instantiation.
usage: python tools/compiler_resources.py [file stems = pair tower_bwd tower enc enc_bwd attn_seq] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from intel_sigir2023_amd import build as B      # noqa: E402

stems = sys.argv[1:] or ['pair', 'tower_bwd', 'tower', 'enc', 'enc_bwd', 'attn_seq']
print('# compiler-reported resources (hipcc %s -Rpass-analysis=kernel-resource-usage; LDS is dynamic in these kernels: DESIGN.md section 3 gives the sizes)' % ' '.join(B.FLAGS))
print('# file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | occupancy waves/SIMD')
for f in stems:
    src = os.path.join(ROOT, 'intel_sigir2023_amd', 'csrc', f + '.hip')
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([B._hipcc()] + B.FLAGS + ['-x', 'hip', '-c', src, '-o', os.path.join(d, 'o.o'), '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
    seen = set()
    for b in re.split(r'remark: Function Name: ', r.stderr)[1:]:
        name = b.split(' ')[0].strip()
        g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]      # noqa: E731
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(.*', '', re.sub(r'\(anonymous namespace\)::|void ', '', dem))
        if dem in seen:
            continue
        seen.add(dem)
        print('%s.hip | %s | %s | %s | %s | %s | %s | %s' % (f, dem, g('    VGPRs'), g('AGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'), g('VGPRs Spill'), g(r'Occupancy \[waves/SIMD\]')))
