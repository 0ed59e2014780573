"""The C-ABI library loads (no GPU needed) and exports every function include/s2vt.h declares;
the ctypes table in _lib.py covers exactly that set."""
import ctypes
import os
import re

import s2vt_amd
from s2vt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "s2vt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(s2vt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 10
    L = ctypes.CDLL(_lib.lib_path())
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/s2vt.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()
    L = s2vt_amd.lib()
    assert L.s2vt_version() >= 100
    assert L.s2vt_error_string(-1) == b"bad argument"


def test_argument_validation_without_gpu():
    L = s2vt_amd.lib()
    assert L.s2vt_math_eval(0, None, None, 4, None) == -1
    assert L.s2vt_sample_workspace_bytes(None, 4, 2, 1) == 0
    d = _lib.Dims(16, 11, 3, 4, 2, 3, 0, 0)
    assert L.s2vt_sample_workspace_bytes(ctypes.byref(d), 4, 2, 1) > 0
    assert L.s2vt_sample(ctypes.byref(d), None, None, 4, 2, 1, 0, 0, None, None, 0, None) == -1
