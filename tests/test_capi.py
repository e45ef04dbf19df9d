"""CPU: the C-ABI shared library builds/loads and exports every symbol include/hoisdf.h declares,
and the ctypes table in hoisdf_amd/_lib.py agrees with the header (argument counts).
No compute calls here (no GPU)."""
import os
import re

import pytest

from hoisdf_amd import _lib

HEADER = open(_lib.HEADER_PATH).read()


def header_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(?:int|long|void|const char\*)\s+(hoisdf_\w+)\s*\(([^;]*?)\)\s*;", body, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len(args.split(","))
        fns[m.group(1)] = n
    return fns


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


def test_header_declares_a_reasonable_surface():
    fns = header_functions()
    assert len(fns) >= 25
    for must in ("hoisdf_project_gather_fwd", "hoisdf_linear_fwd", "hoisdf_attention_fwd", "hoisdf_attention_bwd",
                 "hoisdf_select_smallest_abs", "hoisdf_vote_fwd", "hoisdf_last_error", "hoisdf_version"):
        assert must in fns


def test_library_exports_every_declared_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), f"libhoisdf_hip.so does not export {name}"


def test_ctypes_table_matches_header(lib):
    fns = header_functions()
    for name, args in _lib.SIGNATURES.items():
        assert name in fns, f"{name} bound in _lib.py but not declared in hoisdf.h"
        assert len(args) == fns[name], f"{name}: {len(args)} ctypes args vs {fns[name]} in the header"
    for name in fns:
        assert name in _lib.SIGNATURES or name in _lib._RET or name in _lib._OTHER, name
    for name, (args, _) in _lib._OTHER.items():
        assert len(args) == fns[name], name


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.hoisdf_version()
    assert isinstance(lib.hoisdf_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    """bad arguments are rejected before any HIP call"""
    rc = lib.hoisdf_linear_fwd(None, 4, None, 4, None, None, 4, 8, 4, 4, 0, 0.0, 0, None, None)
    assert rc == -1 and b"null" in lib.hoisdf_last_error()
    with pytest.raises(_lib.HoisdfError):
        _lib.call("hoisdf_select_smallest_abs", None, None, None, 1, 1, None, None)


def test_no_cpu_fallback():
    import torch
    from hoisdf_amd import ops
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))


def test_collective_library_exports_its_header():
    """include/hoisdf_collective.h <-> libhoisdf_rccl.so (symbol table only: loading it would pull RCCL in)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "hoisdf_amd", "libhoisdf_rccl.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "hoisdf_collective.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(hoisdf_\w+)\s*\(", hdr))
    assert {"hoisdf_allreduce", "hoisdf_coll_init", "hoisdf_coll_unique_id", "hoisdf_coll_destroy"} <= declared
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (hoisdf_\w+)", syms))
    assert declared <= exported, declared - exported
