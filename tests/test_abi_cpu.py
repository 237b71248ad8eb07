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
