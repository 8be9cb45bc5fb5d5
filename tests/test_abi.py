"""The C-ABI library loads without a GPU and exports every symbol include/sphx.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

import yasph2d_amd as y
from yasph2d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "sphx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sphx_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(sphx_lib):
    names = header_functions()
    assert len(names) >= 55
    for n in names:
        assert hasattr(sphx_lib, n), f"{n} declared in include/sphx.h but not exported by libsphx.so"


def test_binding_table_covers_header(sphx_lib):
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_abi_version(sphx_lib):
    assert sphx_lib.sphx_abi_version() == 5


def test_struct_layouts_match_header():
    # sizes the C compiler produces for the PODs of sphx.h (natural alignment, no packing)
    assert C.sizeof(_lib.SphxParams) == 4 * 9 + 4 * 4 + 4 * 3 + 4 * 4 == 80
    assert C.sizeof(_lib.SphxStepStats) == 56
    assert C.sizeof(_lib.SphxKernelTime) == 48 + 8 + 8 + 8


def test_default_params():
    p = y.default_params()
    assert abs(p.smoothing_length - 0.02) < 1e-9 and abs(p.particle_mass - 0.01) < 1e-9
    assert p.max_density_iterations == 200 and p.max_divergence_iterations == 400  # dfsph.rs:50,54
    assert tuple(p.gravity) == (0.0, pytest.approx(-9.81)) and tuple(p.grid_min) == (-100.0, -100.0)


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device sphx_create must return an error, never a working context."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(y.SphxError) as e:
        y.SphxContext()
    assert e.value.code == _lib.ERR_NO_DEVICE


def test_invalid_arguments(sphx_lib):
    assert sphx_lib.sphx_default_params(2.0, 0.0, 100.0, None) == _lib.ERR_INVALID_ARGUMENT
    assert sphx_lib.sphx_create(None, None) == _lib.ERR_INVALID_ARGUMENT
    assert sphx_lib.sphx_step_begin(None, 0.1, None) == _lib.ERR_INVALID_ARGUMENT
