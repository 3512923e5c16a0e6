"""CPU: the C-ABI shared library loads and exports every symbol include/pc3d.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "pc3d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pc3d_\w+)\s*\(", txt)))


def test_header_symbols_exported(pc3d):
    lib = pc3d.load()
    names = _declared()
    assert "pc3d_nn_bidir_f32" in names and "pc3d_version" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pc3d.h but not exported by libpc3d_hip.so"


def test_python_signature_table_covers_header(pc3d):
    from importlib import import_module
    _lib = import_module("3dpointcloudattack_amd._lib")
    declared = set(_declared()) - {"pc3d_version", "pc3d_last_error"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_version_and_error_channel(pc3d):
    lib = pc3d.load()
    assert lib.pc3d_version() >= 100
    assert isinstance(lib.pc3d_last_error(), bytes)


def test_argument_errors_do_not_touch_the_gpu(pc3d):
    from importlib import import_module
    _lib = import_module("3dpointcloudattack_amd._lib")
    lib = pc3d.load()
    # M = 0 is rejected before any HIP call
    rc = lib.pc3d_nn_f32(None, 0, 0, 0, None, 0, 0, 0, 1, 4, 0, None, None, None)
    assert rc == -22
    assert b"M must be" in lib.pc3d_last_error()
    # empty batch is a no-op
    assert lib.pc3d_nn_f32(None, 0, 0, 0, None, 0, 0, 0, 0, 4, 4, None, None, None) == 0


def test_ops_refuse_cpu_tensors(ops):
    import pytest
    import torch
    with pytest.raises(Exception) as ei:
        ops.nn_raw(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
    assert "GPU only" in str(ei.value)
