"""The C-ABI shared library loads and exports every symbol include/hipnlp.h declares; without a HIP
device the product fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from hippopt_amd import _abi, hipnlp
from hippopt_amd.kinodyn_settings import periodic_step_settings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "hipnlp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hipnlp_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = hipnlp.load_library()
    names = declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hipnlp.EXPORTS) == names


def test_struct_sizes_match_header():
    # the ctypes mirrors must have the C layout: robot model = 23 ints + 3 ints (+pad) + doubles
    assert C.sizeof(_abi.RobotModelC) % 8 == 0
    n_double = 23 * 9 + 23 * 3 + 23 * 3 + 24 + 24 * 3 + 24 * 9 + 3 * 9 + 3 * 3
    assert C.sizeof(_abi.RobotModelC) == 8 * n_double + 4 * 24 + 4 * 4  # parent[23]+pad, frame_link[3]+pad
    assert C.sizeof(_abi.DimsC) == 44


def test_no_device_is_a_loud_error(model):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(hipnlp.HipNlpError) as e:
        hipnlp.HipNlp(periodic_step_settings(4, model), model)
    assert e.value.code == _abi.E_NODEVICE
    assert "no CPU fallback" in str(e.value)


def test_invalid_descriptor_codes(model):
    st = periodic_step_settings(1, model)  # horizon < 2
    with pytest.raises(hipnlp.HipNlpError) as e:
        hipnlp.HipNlp(st, model)
    assert e.value.code == _abi.E_INVALID
