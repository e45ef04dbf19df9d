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
    for m in re.finditer(r"\b(?:int|long|void|const char\*|const uint32_t\*)\s+(hoisdf_\w+)\s*\(([^;]*?)\)\s*;", body, flags=re.S):
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


def test_coarse_entries_size_queries_and_validation_need_no_gpu(lib):
    """the module-granularity entries: size queries are pure host arithmetic (the same carving code as the real call, dry), and
    malformed descriptors / null buffers are refused before any launch"""
    import ctypes as C
    d = _lib.EncoderLayerDesc(B=32, S=2048, E=256, F=1024, H=4, n_query=1536, n_inter=1536, eps=1e-5, drop_p=0.1, attention=2,
                              attention_bwd_emulated=1, training=1)
    saved = lib.hoisdf_encoder_layer_saved_bytes(C.addressof(d))
    wf, wb = lib.hoisdf_encoder_layer_workspace_bytes(C.addressof(d), 0), lib.hoisdf_encoder_layer_workspace_bytes(C.addressof(d), 1)
    assert saved > 2048 * 32 * 256 * 4 * 3 and wf > 0 and wb > wf                 # at least the q | k | v projections are kept
    d.E = 250                                                                    # not a multiple of the head count / of 4
    assert lib.hoisdf_encoder_layer_saved_bytes(C.addressof(d)) == -1
    assert lib.hoisdf_encoder_layer_fwd(None, None, C.addressof(d), None, None, None, 0, None, 0, None) == -1
    dd = _lib.DecoderLayerDesc(B=32, Q=17, S=2048, E=256, F=1024, H=4, kv_len=1536, eps=1e-5, drop_p=0.1, training=1)
    assert lib.hoisdf_decoder_layer_saved_bytes(C.addressof(dd)) > 0 and lib.hoisdf_decoder_layer_workspace_bytes(C.addressof(dd), 1) > 0
    dd.Q = 65                                                                    # the masked self-attention kernel holds <= 64 queries
    assert lib.hoisdf_decoder_layer_saved_bytes(C.addressof(dd)) == -1
    assert lib.hoisdf_sdf_query_train_saved_bytes(49152, 992) > 49152 * 992 * 4 and lib.hoisdf_sdf_query_train_workspace_bytes(49152, 992, 1) > 0
    assert lib.hoisdf_sdf_infer_workspace(100000, 32, 992) > 0 and lib.hoisdf_sdf_infer_workspace(-1, 32, 992) == -1
    assert lib.hoisdf_mano_dirs_image_floats() == 145 * 2334 + 16 * 778
    assert lib.hoisdf_mano_head_fwd(None, 96, 0, None, 10, 4, None, None, None, None, None, None, None, None, None, 0, 0, None, None, None,
                                    None, None) == -1 and b"null" in lib.hoisdf_last_error()
    assert lib.hoisdf_mano_head_fwd(None, 96, 0, None, 10, 0, None, None, None, None, None, None, None, None, None, 0, 0, None, None, None,
                                    None, None) == 0                                # zero hands: nothing to do
