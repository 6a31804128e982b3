"""Builds libintel_hip.so (the C-ABI library of include/intel_hip.h) in-tree with hipcc for gfx950.

No torch headers are involved: the library is plain HIP + extern "C", bound from Python with ctypes
(intel_sigir2023_amd/_lib.py).  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libintel_hip.so')
OBJ = os.path.join(HERE, 'build')
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-Wall', '-Wno-unused-function']
if os.environ.get('INTEL_DEBUG_BUILD') == '1':      # probe / ablation hooks (csrc/common.h: INTEL_DEBUG_ENV); never set for the product library
    FLAGS.append('-DINTEL_DEBUG')


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (need ROCm with gfx950 support)')


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _stamp():
    h = hashlib.sha256()
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    files.append(os.path.join(HERE, '..', 'include', 'intel_hip.h'))
    for f in files:
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode())
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def is_current():
    """True when libintel_hip.so exists and was built from the sources as they are now."""
    stamp_file = os.path.join(OBJ, 'stamp')
    try:
        return os.path.exists(LIB) and open(stamp_file).read() == _stamp()
    except OSError:
        return False


def build_library(force=False, verbose=False):
    """Compile every csrc/*.hip|*.cpp and link libintel_hip.so.  Returns the library path.
    Safe under torchrun: one process builds at a time (fcntl lock on build/lock; the others find the finished library when
    they get the lock), and the library is linked to a temporary file and renamed into place, so no process can map a
    half-written file."""
    import fcntl
    if not force and is_current():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, 'lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_current():      # another rank built it while this one waited
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    stamp_file = os.path.join(OBJ, 'stamp')
    stamp = _stamp()
    hipcc = _hipcc()

    hdr = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [os.path.join('..', '..', 'include', 'intel_hip.h')]:
        if f.endswith('.h'):
            with open(os.path.join(CSRC, f), 'rb') as fh:
                hdr.update(f.encode())
                hdr.update(fh.read())
    hdr.update(' '.join(FLAGS).encode())

    def compile_one(src):
        # per-object stamp (source + every header + flags): an edit to one .hip recompiles that file only
        obj = os.path.join(OBJ, os.path.basename(src) + '.o')
        h = hdr.copy()
        with open(src, 'rb') as fh:
            h.update(fh.read())
        want = h.hexdigest()
        try:
            if os.path.exists(obj) and open(obj + '.stamp').read() == want:
                return obj
        except OSError:
            pass
        cmd = [hipcc] + FLAGS + ['-x', 'hip', '-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        with open(obj + '.stamp', 'w') as fh:
            fh.write(want)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    tmp = LIB + '.tmp%d' % os.getpid()
    r = subprocess.run([hipcc, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', tmp] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    os.replace(tmp, LIB)
    with open(stamp_file, 'w') as fh:
        fh.write(stamp)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
