"""The C-ABI shared library loads on a CPU-only host and exports every symbol include/*.h declares.
No compute call is made here (no GPU): only planning (workspace sizing) and argument validation."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from ptq4vit_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ptq4vit_hip.h")).read()
    dbg = open(os.path.join(ROOT, "include", "ptq4vit_hip_debug.h")).read()
    assert "p4v_debug_" not in hdr.split("#ifndef PTQ4VIT_HIP_H")[1].replace("p4v_debug_*", ""), "debug entry points belong in ptq4vit_hip_debug.h"
    declared = set(re.findall(r"\b(p4v_[A-Za-z0-9_]+)\s*\(", hdr)) | set(re.findall(r"\b(p4v_[A-Za-z0-9_]+)\s*\(", dbg))
    assert {"p4v_linear_calibrate", "p4v_matmul_calibrate", "p4v_conv_calibrate", "p4v_version"} <= declared
    from ptq4vit_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} is declared in the header but not exported by libptq4vit_hip.so"


def test_version(lib):
    hdr = open(os.path.join(ROOT, "include", "ptq4vit_hip.h")).read()
    assert lib.p4v_version() == int(re.search(r"#define P4V_VERSION (\d+)", hdr).group(1)) >= 130


def test_workspace_planning_matches_shapes(lib):
    from ptq4vit_amd import _lib
    d = _lib.LinearDesc(32, 197, 768, 2304, 3, 1, 1, 8, 8, 4, 100, 3, 0, 0, 1, 0)
    need = lib.p4v_linear_workspace_bytes(C.byref(d))
    # candidate-expanded int8 plane of the activation search dominates: 100 x 6400(pad) x 768
    assert 100 * 6304 * 768 < need < 4 * 100 * 6656 * 768
    m = _lib.MatMulDesc()
    m.batch, m.heads, m.M, m.K, m.N = 32, 12, 197, 64, 197
    m.A_bit = m.B_bit = 8
    m.metric, m.eq_n, m.search_round = 4, 100, 3
    assert lib.p4v_matmul_workspace_bytes(C.byref(m)) > 100 * 384 * 256 * 64
    c = _lib.ConvDesc(32, 3, 224, 224, 768, 16, 16, 16, 16, 0, 0, 1, 1, 8, 32, 4, 100, 3, 1, 0, 1, 0)
    assert lib.p4v_conv_workspace_bytes(C.byref(c)) > 100 * 768 * 768 * 4


def test_invalid_arguments_are_reported_not_crashed(lib):
    from ptq4vit_amd import _lib
    d = _lib.LinearDesc(32, 197, 768, 2304, 3, 1, 1, 8, 8, 4, 100, 3, 0, 0, 1, 0)
    null = C.c_void_p(0)
    rc = lib.p4v_linear_calibrate(C.byref(d), null, null, null, null, null, null, null, null, null, null, null, 0, null)
    assert rc == -1 and b"null pointer" in lib.p4v_last_error()
    bad = _lib.LinearDesc(32, 197, 768, 2304, 5, 1, 1, 8, 8, 4, 100, 3, 0, 0, 1, 0)   # 2304 % 5 != 0
    assert lib.p4v_linear_workspace_bytes(C.byref(bad)) == 0
    assert b"must divide" in lib.p4v_last_error()


def test_granular_entry_points_validate_arguments(lib):
    """the per-pass entry points exist and refuse bad arguments without touching a GPU."""
    from ptq4vit_amd import _lib
    null = C.c_void_p(0)
    d = _lib.LinearDesc(32, 197, 768, 2304, 3, 1, 1, 8, 8, 4, 100, 3, 0, 0, 1, 0)
    assert lib.p4v_amax_init_linear(C.byref(d), null, null, null, null, null, 0, null) == -1
    assert b"p4v_amax_init_linear" in lib.p4v_last_error()
    for fn in (lib.p4v_linear_search_w, lib.p4v_linear_search_a):
        assert fn(C.byref(d), *([null] * 10), null, 0, null) == -1
    m = _lib.MatMulDesc()
    m.batch, m.heads, m.M, m.K, m.N = 2, 3, 8, 4, 8
    m.A_bit = m.B_bit = 8
    m.metric, m.eq_n, m.search_round, m.sos = 4, 20, 1, 1
    one = C.c_void_p(64)   # never dereferenced: the sos / non-sos mismatch is refused first
    assert lib.p4v_matmul_search_A(C.byref(m), *([one] * 9), one, 64, null) == -1
    assert b"p4v_sos_search_split" in lib.p4v_last_error()
    m.sos = 0
    assert lib.p4v_sos_search_split(C.byref(m), *([one] * 8), one, 64, null) == -1
    assert b"not a split-of-softmax" in lib.p4v_last_error()
    c = _lib.ConvDesc(2, 3, 32, 32, 8, 16, 16, 16, 16, 0, 0, 1, 1, 8, 32, 4, 100, 3, 1, 0, 1, 0)
    assert lib.p4v_conv_search_w_layerwise(C.byref(c), *([one] * 10), one, 64, null) == -1
    assert b"channelwise does not match" in lib.p4v_last_error()
    assert lib.p4v_conv_search_a(C.byref(c), *([one] * 10), one, 64, null) == -1
    assert b"a_bit >= 32" in lib.p4v_last_error()
    assert lib.p4v_score_argmax_gather(null, 0, 0, null, null, null, null) == -1
    # the plane / export entry points of version 1.3 (rows a9, f-3)
    pd = _lib.PlaneDesc(16, 16, 64, 1, 7, -128, 127, 128, 0.0, 0)                       # unknown mode
    assert lib.p4v_pack_plane_i8(C.byref(pd), one, one, one, null) == -1 and b"unknown mode" in lib.p4v_last_error()
    pd = _lib.PlaneDesc(16, 16, 64, 1, _lib.PLANE_SOS_HI, 0, 127, 128, 0.0, 0)
    assert lib.p4v_pack_plane_i8(C.byref(pd), one, null, one, null) == -1 and b"split" in lib.p4v_last_error()
    ed = _lib.ExportDesc()
    ed.mode = 9
    assert lib.p4v_export_quantize(C.byref(ed), one, one, null, one, null) == -1 and b"unknown mode" in lib.p4v_last_error()
    assert lib.p4v_debug_set_variant(-1, 0) == -1 and lib.p4v_debug_set_tuning(99, 0) == -1
    assert lib.p4v_debug_set_variant(0, 0) == 0 and lib.p4v_stats_enable(0) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ptq4vit_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
