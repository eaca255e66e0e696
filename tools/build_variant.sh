#!/bin/bash
# Build an A/B variant of the WHOLE library with extra hipcc flags on every source: tools/build_variant.sh <tag> [extra hipcc flags...]
# -> fastdiffsr_amd/csrc/ab/libfdsr_hip_<tag>.so  (run with FDSR_LIB=<that path>).  The source list and the per-file flags are
# fastdiffsr_amd/build.py's SOURCES / COMMON, a failed compile fails the script, the link refuses undefined symbols.
set -e
TAG=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
cd $R
python - "$TAG" "$@" <<'PY'
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
from fastdiffsr_amd import build as B
tag, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(B.CSRC, 'ab', tag)
os.makedirs(out, exist_ok=True)
def one(item):
    src, flags = item
    obj = os.path.join(out, src.replace('.', '_') + '.o')
    stamp = ['-DFDSR_SRC_SHA256="variant-%s"' % tag] if src == 'fdsr_engine.cpp' else []
    subprocess.check_call([B._hipcc()] + B.COMMON + flags + stamp + extra + ['-c', os.path.join(B.CSRC, src), '-o', obj])
    return obj
with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(one, B.SOURCES))
lib = os.path.join(B.CSRC, 'ab', 'libfdsr_hip_%s.so' % tag)
subprocess.check_call([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,-z,defs'] + objs + ['-o', lib])
for o in objs:
    os.remove(o)
os.rmdir(out)
print('built', lib)
PY
