"""The C-ABI library loads without a GPU and exports exactly what include/mbar_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mbar_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:int|void|const char\*)\s+(mbar_\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["mbar_ctx_create", "mbar_ctx_upload_u", "mbar_ctx_set_Nk", "mbar_eval", "mbar_lognum", "mbar_logw",
                 "mbar_gram_w", "mbar_solve_adaptive", "mbar_solve_sci", "mbar_ctx_comm_init", "mbar_last_error",
                 "mbar_ctx_destroy", "mbar_ctx_timing"]:
        assert must in names
    assert len(names) >= 25


def test_library_exports_every_declared_symbol():
    from pymbar_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/mbar_hip.h but not exported"
    # and the ctypes table binds exactly the declared surface
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_no_compute_symbols_leak_torch():
    """The boundary is plain C: no torch / c10 / at:: symbols are linked into the library."""
    import subprocess
    from pymbar_amd import _lib

    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "c10" not in out and "torch" not in out and "at::" not in out
    assert "hipLaunchKernel" in out or "hipModuleLaunchKernel" in out or "__hipPushCallConfiguration" in out


def test_version_and_error_paths_without_gpu():
    from pymbar_amd import _lib

    lib = _lib.load_library()
    assert lib.mbar_version() >= 100
    if _lib.device_count() == 0:
        ctx = ctypes.c_void_p()
        rc = lib.mbar_ctx_create(ctypes.byref(ctx), 0, 4, 16)
        assert rc != 0 and not ctx.value
        assert "device" in _lib.last_error(None).lower()
    rc = lib.mbar_ctx_create(None, 0, 4, 16)
    assert rc == -1
