"""The C-ABI library loads without a GPU and exports every symbol include/fdsr.h declares
(no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'fdsr.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(fdsr_[a-z_0-9]+)\s*\(', txt)))


def test_header_symbols_exported_and_bound():
    from fastdiffsr_amd import _lib, build
    build.build(force=False, verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f'{n} declared in fdsr.h but not exported'
    assert set(names) == set(_lib.SYMBOLS), 'ctypes binding table and header disagree'
    assert b'gfx950' in lib.fdsr_version()
    # provenance: the library carries the hash of the tree it was built from, and that is THIS tree
    stamp = lib.fdsr_version().decode().rsplit('FDSR_SRC_SHA256=', 1)[1]
    assert stamp == build.source_hash() == build.library_stamp()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from fastdiffsr_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB', str(tmp_path / 'nope.so'))
    with pytest.raises(ImportError, match='no CPU/PyTorch fallback'):
        _lib.load()


def test_product_never_imports_oracle():
    """The shipped package must not reference oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, 'fastdiffsr_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_header_is_plain_c99_and_demo_links(tmp_path):
    """include/fdsr.h must compile as C (no C++ or torch types at the boundary), and the plain-C host in
    examples/ must link against the in-tree library with nothing but the HIP runtime."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which('gcc')
    if gcc is None:
        import pytest
        pytest.skip('no gcc')
    src = tmp_path / 'hdr.c'
    src.write_text('#include "fdsr.h"\nint main(void) { fdsr_config c; (void)c; return sizeof(fdsr_schedule) ? 0 : 1; }\n')
    subprocess.check_call([gcc, '-std=c99', '-pedantic', '-Wall', '-Werror', '-I' + os.path.join(root, 'include'),
                           '-fsyntax-only', str(src)])
    from fastdiffsr_amd import build as b
    demo = b.build_demo(force=True, verbose=False)
    out = subprocess.run(['ldd', demo], capture_output=True, text=True).stdout
    assert 'libfdsr_hip.so' in out and 'not found' not in out.split('libfdsr_hip.so')[1].split('\n')[0]
    assert 'torch' not in out and 'python' not in out


def test_debug_options_are_an_abi_call_not_environment():
    """Launcher A/B options go through fdsr_debug_option (host-only, no GPU needed): known names are accepted, unknown ones are
    FDSR_E_INVALID, and no kernel / engine source reads the environment."""
    from fastdiffsr_amd import _lib
    lib = _lib.load()
    defaults = {'rider': 2, 'up2': 1, 'splitk': 1, 'sk_target': 256, 'th_min_wgs': 256, 'strip': 91, 'strip_min_wgs': 512,
                'wgrad_form': 0, 'wgrad_colsum': 1, 'wgrad_f32': 0, 'gnb_fuse': 1, 'drop_stage': 1, 'wgrad_big_bytes': 1 << 32, 'sat_guard': 1, 'gn_consumer': 1, 'tail': 1, 'k32': 1275, 'k32_sb_min_wgs': 1024, 'k32_stagger': 0, 'drop_image_offset': 0}
    for name, value in defaults.items():          # every documented name is accepted (set to its default)
        assert lib.fdsr_debug_option(name.encode(), value) == 0, name
    assert lib.fdsr_debug_option(b'no_such_option', 1) == -1          # FDSR_E_INVALID
    with pytest.raises(_lib.FdsrError):
        _lib.debug_option('no_such_option', 1)
    csrc = os.path.join(ROOT, 'fastdiffsr_amd', 'csrc')
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.cpp', '.h')):
            assert 'getenv' not in open(os.path.join(csrc, f)).read(), f


def test_kernel_form_bits_match_the_header():
    """enum fdsr_k32_bits / fdsr_strip_bits of include/fdsr.h and their mirror in _lib.py (names without the prefix)."""
    from fastdiffsr_amd import _lib
    txt = open(os.path.join(ROOT, 'include', 'fdsr.h')).read()
    for prefix, table, default in (('FDSR_K32_', _lib.K32, _lib.K32_DEFAULT), ('FDSR_STRIP_', _lib.STRIP, _lib.STRIP_DEFAULT)):
        found = {m.group(1): int(m.group(2)) for m in re.finditer(prefix + r'([A-Z0-9_]+) = (\d+),', txt)}
        assert found == table, (prefix, found)
        expr = re.search(prefix + r'DEFAULT = ([0-9| ]+)', txt).group(1)
        assert eval(expr) == default
    assert _lib.bit_names(_lib.STRIP, 91) == 'BF16_64|F16X3_64|BF16_CAT64|BF16_RIDER|BF16_COUT128'


def test_precision_codes_mirror_the_header():
    """`_lib.PRECISIONS` (what Engine.set_precision passes) against the FDSR_PREC_* defines of include/fdsr.h, the f16 mode included."""
    import re
    from fastdiffsr_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'fdsr.h')).read()
    defs = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r'#define\s+FDSR_PREC_(\w+)\s+(\d+)', text)}
    assert defs == _lib.PRECISIONS, (defs, _lib.PRECISIONS)
    assert defs['f16'] == 3
