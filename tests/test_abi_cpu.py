"""CPU: the C-ABI library loads and exports every symbol include/ieee_amd.h declares, and the
ctypes table in ieee_amd/_lib.py covers exactly that set.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ieee_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ieee_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "ieee_sqeuclid_distmat" in syms and "ieee_rank_market1501" in syms and "ieee_last_error" in syms


def test_library_exports_every_declared_symbol():
    from ieee_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), "libieee_amd.so does not export %s" % s


def test_ctypes_table_matches_header():
    from ieee_amd import _lib
    assert sorted(_lib.exported_symbols()) == declared_symbols()


def test_version_and_error_text():
    from ieee_amd import _lib
    lib = _lib.load()
    assert lib.ieee_version() == 1
    assert isinstance(lib.ieee_last_error(), bytes)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ieee_amd import _lib
    from ieee_amd.metrics import compute_distance_matrix
    with pytest.raises(_lib.IeeeAmdError):
        compute_distance_matrix(torch.zeros(2, 8), torch.zeros(3, 8))


def test_loading_the_library_brings_torch_in_first():
    """torch's wheel carries its own HIP runtime; loaded before torch, libieee_amd.so would bind the system copy and the process
    would hold two runtimes (the one the library talks to then does not see torch's device: ieee_device_is_gfx950() = 0).
    _lib.load() therefore imports torch before it opens the library -- checked in a fresh interpreter."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import ieee_amd; assert 'torch' not in sys.modules; "
            "from ieee_amd import _lib; _lib.load(); assert 'torch' in sys.modules; print('ok')" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_struct_layouts_match_the_header(tmp_path):
    """the ctypes mirrors of the header's structs (ieee_conv_extras of the explicit-argument conv calls, ieee_wgrad_reduce_desc
    of the deferred / chained weight gradients) against what a C compiler makes of include/ieee_amd.h: size and every offset"""
    import subprocess
    from ieee_amd import _lib
    fields = {"ieee_conv_extras": [f[0] for f in _lib.ConvExtras._fields_],
              "ieee_wgrad_reduce_desc": [f[0] for f in _lib.WgradReduceDesc._fields_],
              "ieee_sgemm_set": [f[0] for f in _lib.SgemmSet._fields_]}
    lines = []
    for st, names in fields.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (st, st))
        lines += ['printf(" %%zu", offsetof(%s, %s));' % (st, f) for f in names]
        lines.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ieee_amd.h"\nint main(void) {\n%s\nreturn 0; }\n' % "\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    mirrors = {"ieee_conv_extras": _lib.ConvExtras, "ieee_wgrad_reduce_desc": _lib.WgradReduceDesc, "ieee_sgemm_set": _lib.SgemmSet}
    assert len(out) == 3
    for line in out:
        name, size, *offs = line.split()
        cls = mirrors[name]
        assert ctypes.sizeof(cls) == int(size), name
        assert [getattr(cls, f).offset for f in fields[name]] == [int(o) for o in offs], name
