"""The C-ABI library loads and exports exactly what include/diffstereo.h declares.
Runs without a GPU: only argument-validation paths are called (they return before
any HIP call)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from __graft_entry__ import ROOT, load_package

load_package()
from stereo_vision_amd import diffstereo  # noqa: E402

HEADER = os.path.join(ROOT, "include", "diffstereo.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(diffstereo.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = diffstereo.lib()  # raises if the .so is missing or a symbol is not exported
    out = subprocess.check_output(["nm", "-D", "--defined-only", diffstereo.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (ds_[a-z0-9_]+)\b", out))
    assert exported == set(declared_symbols())
    assert lib.ds_create is not None


def test_static_library_holds_the_same_objects():
    a = os.path.join(os.path.dirname(diffstereo.LIB_PATH), "libdiffstereo.a")
    out = subprocess.check_output(["nm", "--defined-only", a], text=True)
    assert set(re.findall(r"\bT (ds_[a-z0-9_]+)\b", out)) == set(declared_symbols())


def test_argument_validation_needs_no_gpu():
    lib = diffstereo.lib()
    img = np.zeros((4, 4), np.uint8)
    p = img.ctypes.data_as(C.c_void_p)
    h = C.c_void_p()
    assert lib.ds_create(p, p, 4, 4, 0, 1.2, 0, C.byref(h)) == -1      # D < 1
    assert b"D must be" in lib.ds_last_error(None)
    assert lib.ds_create(p, p, 4, 4, 257, 1.2, 0, C.byref(h)) == -1    # D > 256
    assert lib.ds_create(p, p, 0, 4, 4, 1.2, 0, C.byref(h)) == -1      # empty image
    assert lib.ds_create(p, p, 4, 4, 4, 1.2, 7, C.byref(h)) == -1      # dtype
    assert lib.ds_create(None, p, 4, 4, 4, 1.2, 0, C.byref(h)) == -1   # NULL image
    assert lib.ds_create(p, p, 4, 4, 4, float("inf"), 0, C.byref(h)) == -1
    assert lib.ds_create_band(p, p, 4, 4, 4, 1.2, 0, 2, 2, -1, C.byref(h)) == -1  # empty band
    assert lib.ds_create_band(p, p, 4, 4, 4, 1.2, 0, 0, 5, -1, C.byref(h)) == -1  # band > H
    assert h.value is None
    assert lib.ds_iterate(None, 1) == -1
    assert lib.ds_energy(None, None, None) == -1
    lib.ds_destroy(None)  # no-op


def test_python_binding_has_no_cpu_fallback(monkeypatch):
    """The product path must fail loudly when the HIP library is missing."""
    import pytest
    monkeypatch.setattr(diffstereo, "_LIB", None)
    monkeypatch.setattr(diffstereo, "LIB_PATH", "/nonexistent/libdiffstereo.so")
    with pytest.raises(ImportError):
        diffstereo.lib()
    src = open(os.path.join(ROOT, "stereo-vision_amd", "diffstereo.py")).read()
    src += open(os.path.join(ROOT, "stereo-vision_amd", "bands.py")).read()
    assert "oracle" not in src.replace("CPU oracle", "")


def test_missing_librccl_is_an_error_code_not_a_crash():
    """ds_comm_* on a host where librccl cannot be loaded: DS_ERR_STATE with a text, twice in a
    row (the loader's message can only be read once).  A child process: the lookup is cached."""
    code = (
        "import ctypes as C, sys\n"
        "lib = C.CDLL(sys.argv[1])\n"
        "lib.ds_comm_last_error.restype = C.c_char_p\n"
        "ident = (C.c_uint8 * 128)()\n"
        "for _ in range(2):\n"
        "    rc = lib.ds_comm_unique_id(ident)\n"
        "    text = lib.ds_comm_last_error(None)\n"
        "    assert rc == -3, rc\n"
        "    assert b'librccl not found' in text, text\n"
        "comm = C.c_void_p()\n"
        "assert lib.ds_comm_create(ident, 0, 1, 0, C.byref(comm)) == -3\n"
        "assert comm.value is None\n"
        "print('ok')\n"
    )
    env = dict(os.environ, DS_RCCL_LIBRARY="/nonexistent/librccl-not-here.so")
    out = subprocess.run(["python3", "-c", code, diffstereo.LIB_PATH], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
