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
    src = open(os.path.join(ROOT, "include", "hipnlp.h")).read() + open(os.path.join(ROOT, "include", "hipnlp_ipopt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hipnlp_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = hipnlp.load_library()
    names = declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hipnlp.EXPORTS) == names


def test_struct_layouts_match_the_c_header(tmp_path):
    """sizes and member offsets as a C compiler lays include/hipnlp.h out, against the ctypes mirrors (a caller built against the
    header and this package must agree byte for byte)"""
    import subprocess
    probes = [("hipnlp_robot_model", _abi.RobotModelC, ["parent", "R_fix", "mass", "frame_link", "frame_o"]),
              ("hipnlp_terrain_step", _abi.TerrainStepC, ["position", "edge_sharpness"]),
              ("hipnlp_settings", _abi.SettingsC, ["yaw_corner", "final_state_weight", "joint_regularization_cost_weights", "n_terrain_steps", "terrain_steps"]),
              ("hipnlp_desc", _abi.DescC, ["model", "batch", "device", "abi_version", "flags"]),
              ("hipnlp_dims", _abi.DimsC, ["nnz_knot", "shard_grad_off", "m_full", "n_lifted"]),
              ("hipnlp_pose_settings", _abi.PoseSettingsC, ["com_position_type", "base_quaternion_cost_multiplier", "hand_type", "hand_regularization_cost_multiplier"]),
              ("hipnlp_pose_desc", _abi.PoseDescC, ["model", "batch", "abi_version", "flags"]),
              ("hipnlp_pose_dims", _abi.PoseDimsC, ["np"])]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hipnlp.h"', 'int main(void) {', '  printf("%d\\n", HIPNLP_ABI_VERSION);']
    for cname, _, members in probes:
        lines.append('  printf("%%zu\\n", sizeof(%s));' % cname)
        lines += ['  printf("%%zu\\n", offsetof(%s, %s));' % (cname, m) for m in members]
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi_probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi_probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = iter(int(v) for v in subprocess.check_output([str(exe)]).split())
    assert next(out) == _abi.ABI_VERSION == hipnlp.load_library().hipnlp_abi_version()
    for cname, mirror, members in probes:
        assert next(out) == C.sizeof(mirror), cname
        for m in members:
            assert next(out) == getattr(mirror, m).offset, (cname, m)


def test_descriptor_of_another_abi_version_is_refused(model):
    """hipnlp_create / hipnlp_pose_create check desc.abi_version before anything else (no device needed to be told)"""
    from hippopt_amd.pose_settings import pose_finder_settings
    lib = hipnlp.load_library()
    desc = _abi.DescC()
    desc.settings, desc.model, desc.batch = periodic_step_settings(4, model).to_c(), model.to_c(), 1
    pose = _abi.PoseDescC()
    pose.settings, pose.model, pose.batch = pose_finder_settings(model).to_c(), model.to_c(), 1
    for d, create, last in ((desc, lib.hipnlp_create, lib.hipnlp_last_error), (pose, lib.hipnlp_pose_create, lib.hipnlp_pose_last_error)):
        for version in (0, 1, _abi.ABI_VERSION + 1):
            d.abi_version = version
            h = C.c_void_p()
            assert create(C.byref(d), C.byref(h)) == _abi.E_INVALID and not h.value
            assert b"abi_version" in last(None)
    assert "gfx950" in hipnlp.build_info()


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


def test_cpulist_of_a_numa_node_is_parsed():
    from hippopt_amd.hipnlp import parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert parse_cpulist("64-127,192-255") == set(range(64, 128)) | set(range(192, 256))
    assert parse_cpulist("") == set()
