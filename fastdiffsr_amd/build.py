"""Build libfdsr_hip.so in-tree with hipcc for gfx950 (no torch extension machinery:
the library is a plain C-ABI shared object, see include/fdsr.h)."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libfdsr_hip.so')
# -fno-slp-vectorize: packed f32 VALU (v_pk_fma_f32 ...) next to MFMAs is slower than the scalar forms on
# gfx950 (MI355X_MICROARCH.md, cycle constants); +0.8 % end to end in a same-box A/B
# fdsr_conv_strip.hip compiles three times (-DSTRIP_PART: the host side + f16x3 kernels, the bf16 kernels, the f16 kernels): objects of their own, in parallel
SOURCES = [('fdsr_kernels.hip', ['-O3']), ('fdsr_conv_h.hip', ['-O3', '-fno-slp-vectorize']),
           ('fdsr_conv_up2.hip', ['-O3', '-fno-slp-vectorize']),
           ('fdsr_conv_k32.hip', ['-O3', '-fno-slp-vectorize']), ('fdsr_conv_strip.hip', ['-O3', '-fno-slp-vectorize']),
           ('fdsr_conv_strip.hip', ['-O3', '-fno-slp-vectorize', '-DSTRIP_PART=2']), ('fdsr_conv_strip.hip', ['-O3', '-fno-slp-vectorize', '-DSTRIP_PART=3']), ('fdsr_conv_tail.hip', ['-O3', '-fno-slp-vectorize']),
           ('fdsr_val.hip', ['-O3']), ('fdsr_train.hip', ['-O3']), ('fdsr_engine.cpp', ['-O2']), ('fdsr_train.cpp', ['-O2'])]
COMMON = ['--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


STAMP_MARK = b'FDSR_SRC_SHA256='


def source_hash():
    """SHA-256 over every file the library is built from (csrc/*.hip|cpp|h, include/fdsr.h) and the compile
    flags.  build() compiles it into the library (fdsr_version() ends with it), so a binary can be matched to
    the tree it came from -- the gitignored .so travels with the snapshot to the GPU box."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp', '.h')))
    paths = [os.path.join(CSRC, f) for f in files] + [os.path.normpath(os.path.join(CSRC, '..', '..', 'include', 'fdsr.h'))]
    for pth in paths:
        h.update(os.path.basename(pth).encode() + b'\0')
        h.update(open(pth, 'rb').read())
        h.update(b'\0')
    h.update(repr((SOURCES, COMMON)).encode())
    return h.hexdigest()


def library_stamp(lib=None):
    """The source hash compiled into an existing library, read from the file (no dlopen); None if absent."""
    lib = lib or LIB
    if not os.path.exists(lib):
        return None
    blob = open(lib, 'rb').read()
    i = blob.find(STAMP_MARK)
    if i < 0:
        return None
    stamp = blob[i + len(STAMP_MARK):i + len(STAMP_MARK) + 64]
    try:
        return stamp.decode('ascii')
    except UnicodeDecodeError:
        return None


def needs_build():
    return library_stamp() != source_hash()


def _object_key(src, flags):
    """What one object file depends on: its source, every internal header, its flags."""
    import hashlib
    h = hashlib.sha256()
    deps = [src] + sorted(f for f in os.listdir(CSRC) if f.endswith('.h'))
    for f in deps:
        h.update(f.encode() + b'\0' + open(os.path.join(CSRC, f), 'rb').read() + b'\0')
    h.update(open(os.path.normpath(os.path.join(CSRC, '..', '..', 'include', 'fdsr.h')), 'rb').read())
    h.update(repr((flags, COMMON)).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    stamp = source_hash()

    def compile_one(item):
        src, flags = item
        part = ''.join('_part' + f.split('=')[1] for f in flags if f.startswith('-DSTRIP_PART='))
        obj = os.path.join(CSRC, src.replace('.', '_') + part + '.o')   # fdsr_train.hip and fdsr_train.cpp both exist; fdsr_conv_strip.hip compiles in three parts
        extra = ['-DFDSR_SRC_SHA256="%s"' % stamp] if src == 'fdsr_engine.cpp' else []
        key = _object_key(src, flags + extra)
        keyfile = obj + '.key'       # objects whose inputs did not change are reused (kernel files take ~30 s each)
        if not force and os.path.exists(obj) and os.path.exists(keyfile) and open(keyfile).read() == key:
            return obj
        cmd = [hipcc] + COMMON + flags + extra + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(keyfile, 'w').write(key)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,-z,defs'] + objs + ['-o', LIB]   # a missing object fails HERE, not at dlopen
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    assert library_stamp() == stamp, 'built library does not carry its source hash'
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
