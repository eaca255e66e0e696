"""Build libfdsr_hip.so in-tree with hipcc for gfx950 (no torch extension machinery:
the library is a plain C-ABI shared object, see include/fdsr.h)."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libfdsr_hip.so')
# -fno-slp-vectorize: packed f32 VALU (v_pk_fma_f32 ...) next to MFMAs is slower than the scalar forms on
# gfx950 (MI355X_MICROARCH.md, cycle constants); +0.8 % end to end in a same-box A/B
SOURCES = [('fdsr_kernels.hip', ['-O3', '-munsafe-fp-atomics']), ('fdsr_conv_h.hip', ['-O3', '-fno-slp-vectorize']),
           ('fdsr_conv_up2.hip', ['-O3', '-fno-slp-vectorize']),
           ('fdsr_engine.cpp', ['-O2'])]
COMMON = ['--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp', '.h'))]
    deps.append(os.path.join(os.path.dirname(CSRC), '..', 'include', 'fdsr.h'))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    for src, flags in SOURCES:
        obj = os.path.join(CSRC, src.rsplit('.', 1)[0] + '.o')
        cmd = [hipcc] + COMMON + flags + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


DEMO_SRC = os.path.join(os.path.dirname(CSRC), '..', 'examples', 'fdsr_demo.c')
DEMO = os.path.join(os.path.dirname(CSRC), '..', 'examples', 'fdsr_demo')


def build_demo(force=False, verbose=True):
    """examples/fdsr_demo: the C ABI driven from plain C99 (gcc, no torch), linked against the in-tree library."""
    src, out = os.path.normpath(DEMO_SRC), os.path.normpath(DEMO)
    if not os.path.exists(src):
        return None
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return out
    inc = os.path.normpath(os.path.join(os.path.dirname(CSRC), '..', 'include'))
    cmd = ['gcc', '-std=c99', '-O2', '-Wall', '-D__HIP_PLATFORM_AMD__', '-I' + inc, '-I/opt/rocm/include', src,
           '-L' + CSRC, '-lfdsr_hip', '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,$ORIGIN/../fastdiffsr_amd/csrc',
           '-Wl,-rpath,/opt/rocm/lib', '-o', out]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    build_demo(force='--force' in sys.argv)
