"""CPU-side checks of the C-ABI boundary: the in-tree library loads, exports EVERY symbol that
include/glnn_hip.h declares, the Python binding table matches the header, and ops refuse CPU tensors
(no silent fallback).  No compute call is made here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "glnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"GLNN_API\s+[\w\s\*]+?\b(glnn_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    lib_path = ge.build()
    h = ctypes.CDLL(lib_path)
    fns = header_functions()
    assert len(fns) >= 15
    for name in fns:
        assert hasattr(h, name), f"{name} declared in include/glnn_hip.h but not exported"
    h.glnn_abi_version.restype = ctypes.c_int
    assert h.glnn_abi_version() == 4
    h.glnn_last_error.restype = ctypes.c_char_p
    assert h.glnn_last_error() is not None


def test_binding_table_matches_header():
    import glnn_amd
    from glnn_amd import _lib
    fns = header_functions()
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in fns, name
        assert len(argtypes) == fns[name], f"{name}: binding has {len(argtypes)} args, header {fns[name]}"
    assert set(fns) - set(_lib.SIGNATURES) == {"glnn_last_error"}
    glnn_amd.lib()


def test_invalid_arguments_are_reported_not_crashed():
    from glnn_amd import _lib
    h = _lib.lib()
    rc = h.glnn_spmm_csr_f32(None, None, 4, 4, None, 4, 4, 0, None, None, None, 0, None, None, None, 0, None, 4, None)
    assert rc == -1 and b"null pointer" in h.glnn_last_error()
    assert h.glnn_spmm_csr_f32(None, None, 0, 0, None, 4, 4, 0, None, None, None, 0, None, None, None, 0, None, 4, None) == 0   # empty: no-op
    rc = h.glnn_gemm_f32(None, 4, None, None, None, 0.0, 0, 4, 4, None, 4, 0, 4, None, None, None, 0, None, 4, None, 0, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    from glnn_amd import GlnnError, ops
    x = torch.zeros(4, 4)
    with pytest.raises(GlnnError):
        ops.gemm(x, x)
    with pytest.raises(GlnnError):
        ops.log_softmax(x)
    with pytest.raises(GlnnError):
        ops.spmm(torch.zeros(5, dtype=torch.int64), torch.zeros(1, dtype=torch.int32), x, 4, ops.AGG_SUM)
