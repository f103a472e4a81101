"""Build libevoke_hip.so (gfx950) in-tree with hipcc.  `python -m evoke_amd.build`.

The GPU box receives the built .so with the repo snapshot; `evoke_amd.hip` refuses to run without it."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'obj')
LIB = os.path.join(HERE, 'libevoke_hip.so')               # fp16 storage (the default: 1e-3 loss parity, dynamic loss scaling)
LIB_BF16 = os.path.join(HERE, 'libevoke_hip_bf16.so')     # same sources, -DEVK_STORE_BF16 (EVK_STORE=bf16)
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-ffp-contract=fast']


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _build_one(lib, obj_dir, extra, verbose, force):
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'evoke_hip.h'))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(obj_dir, s[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [os.path.join(CSRC, s)] + hdrs):
            jobs.append([HIPCC] + FLAGS + extra + ['-c', os.path.join(CSRC, s), '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if jobs or force or _stale(lib, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib


def build(verbose=False, force=False):
    """Both libraries: fp16 storage (default) and bf16 storage (EVK_STORE=bf16)."""
    with ThreadPoolExecutor(max_workers=2) as ex:       # the two libraries compile side by side (gemm.hip dominates both)
        alt = ex.submit(_build_one, LIB_BF16, os.path.join(CSRC, 'obj_bf16'), ['-DEVK_STORE_BF16'], verbose, force)
        lib = ex.submit(_build_one, LIB, OBJ, [], verbose, force)
        alt.result()
        return lib.result()


def source_fingerprint():
    """sha256 (16 hex digits) over the kernel sources, the C-ABI header and the host modules: profiles/*traffic.json is stamped
    with it, and bench.py reports `roofline.traffic` only while the stamp matches the code that is running."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'evoke_hip.h'))
    files += [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith('.py')]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(build(verbose=True, force='--force' in sys.argv))
