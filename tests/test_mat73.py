"""MAT v7.3 container (hippopt_amd/mat73.py): the format the reference's main scripts write their results in
(hdf5storage.savemat, main_periodic_step.py:509-513).  CPU only; skipped where no HDF5 C library is installed."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from hippopt_amd import mat73

pytestmark = pytest.mark.skipif(not mat73.available(), reason="no HDF5 C library in this image")


def _sample():
    rng = np.random.default_rng(3)
    return {
        "output": {
            "cost_value": 12.5,
            "values": {"system": [{"kinematics": {"joints": {"positions": np.linspace(-1, 1, 23)}}, "dt": 0.1},
                                  {"kinematics": {"joints": {"positions": np.zeros(23)}}, "dt": 0.1}]},
            "matrix": np.arange(6.0).reshape(2, 3),
            "name": "périodic step",
            "converged": True,
            "iterations": np.int64(57),
            "unset": None,
            "left[0]": {"f": np.array([1.0, 2.0, 3.0])},
            "mixed": [{"a": np.arange(3.0)}, "text", [1.0, 2.0], []],
        },
        "big": rng.standard_normal((100, 50)),      # 40 KB: the compressed branch
        "i32": np.arange(12, dtype=np.int32).reshape(3, 4),
    }


def test_round_trip(tmp_path):
    d = _sample()
    f = mat73.savemat(str(tmp_path / "res"), d)
    assert f.endswith("res.mat") and mat73.is_v73(f)
    b = mat73.loadmat(f)
    assert set(b) == {"output", "big", "i32"}
    o = b["output"]
    assert float(o["cost_value"]) == 12.5
    assert np.array_equal(o["matrix"], d["output"]["matrix"])
    assert o["name"] == "périodic step"
    assert o["converged"].dtype == bool and bool(o["converged"])
    assert o["iterations"].dtype == np.int64 and int(o["iterations"]) == 57
    assert o["unset"].size == 0
    assert np.array_equal(o["left[0]"]["f"], [1.0, 2.0, 3.0])
    assert len(o["values"]["system"]) == 2
    assert np.array_equal(o["values"]["system"][0]["kinematics"]["joints"]["positions"], np.linspace(-1, 1, 23))
    m = o["mixed"]
    assert np.array_equal(m[0]["a"], np.arange(3.0)) and m[1] == "text" and [float(x) for x in m[2]] == [1.0, 2.0] and m[3] == []
    assert np.array_equal(b["big"], d["big"])            # bit exact through gzip + shuffle + fletcher32
    assert b["i32"].dtype == np.int32 and np.array_equal(b["i32"], d["i32"])
    # without squeezing: MATLAB's dimensions (row vectors 1 x n, scalars 1 x 1)
    raw = mat73.loadmat(f, squeeze=False)
    assert raw["output"]["left[0]"]["f"].shape == (1, 3) and raw["output"]["cost_value"].shape == (1, 1)


def test_header_and_field_order(tmp_path):
    f = mat73.savemat(str(tmp_path / "h.mat"), {"s": {"zeta": 1.0, "alpha": 2.0, "mid": 3.0}})
    head = open(f, "rb").read(520)
    assert head[:19] == b"MATLAB 7.3 MAT-file" and b"HDF5 schema 1.00 ." in head[:116]
    assert head[116:124] == bytes(8) and head[124:128] == b"\x00\x02IM"
    assert head[128:512] == bytes(384) and head[512:520] == b"\x89HDF\r\n\x1a\n"
    # MATLAB_fields keeps the insertion order of a struct whose names are valid MATLAB field names (HDF5 lists links by name)
    assert list(mat73.loadmat(f)["s"]) == ["zeta", "alpha", "mid"]
    # scipy recognises the file as v7.3 (which it does not read)
    from scipy.io import loadmat
    with pytest.raises(NotImplementedError, match="7.3"):
        loadmat(f)


@pytest.mark.skipif(shutil.which("h5dump") is None and not os.path.exists("/opt/conda/bin/h5dump"), reason="no h5dump")
def test_layout_as_seen_by_the_hdf5_tools(tmp_path):
    """the on-disk conventions, read with the HDF5 project's own tool rather than with this package's reader"""
    f = mat73.savemat(str(tmp_path / "l.mat"), {"out": {"m": np.arange(6.0).reshape(2, 3), "txt": "ab", "c": [1.0, "x"], "e": None, "b": False}})
    tool = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    dump = subprocess.run([tool, f], capture_output=True, text=True, check=True).stdout
    flat = " ".join(dump.split())
    assert 'GROUP "#refs#"' in flat and '"canonical empty"' in flat
    assert 'GROUP "out" { ATTRIBUTE "MATLAB_class"' in flat and '(0): "struct"' in flat
    # 2 x 3 in MATLAB = 3 x 2 on disk, column by column
    assert 'DATASET "m" { DATATYPE H5T_IEEE_F64LE DATASPACE SIMPLE { ( 3, 2 ) / ( 3, 2 ) } DATA { (0,0): 0, 3, (1,0): 1, 4, (2,0): 2, 5 }' in flat
    assert 'DATASET "txt" { DATATYPE H5T_STD_U16LE DATASPACE SIMPLE { ( 2, 1 ) / ( 2, 1 ) } DATA { (0,0): 97, (1,0): 98 }' in flat
    assert 'ATTRIBUTE "MATLAB_int_decode"' in flat and '(0): "char"' in flat and '(0): "logical"' in flat
    assert 'DATASET "c" { DATATYPE H5T_REFERENCE { H5T_STD_REF_OBJECT } DATASPACE SIMPLE { ( 2, 1 ) / ( 2, 1 ) }' in flat and '(0): "cell"' in flat
    assert 'DATASET "e" { DATATYPE H5T_STD_U64LE DATASPACE SIMPLE { ( 2 ) / ( 2 ) } DATA { (0): 0, 0 }' in flat and 'ATTRIBUTE "MATLAB_empty"' in flat
    info = subprocess.run([tool, "-B", "-H", f], capture_output=True, text=True).stdout
    assert "USERBLOCK_SIZE 512" in " ".join(info.split())


def test_planner_output_and_guess_through_v73(tmp_path):
    """{"output": output.to_dict(), "guess": guess.to_dict(flatten=False)} (main_periodic_step.py:503-513), v7.3 and v5 give the same dictionary"""
    from hippopt_amd.base import Output
    from hippopt_amd.robot_model import synthetic_ergocub
    from hippopt_amd.serialization import load_mat, save_mat
    from test_host_api import _planner
    model = synthetic_ergocub()
    pl, st = _planner(model)
    guess = pl.get_initial_guess()
    guess.system[1].kinematics.joints.positions = np.linspace(-1, 1, 23)
    out = Output(values=guess, cost_value=12.5, cost_values={"com_velocity_error": 1.5, "system.contact_points.left[0].f_regularization": 2.0},
                 constraint_multipliers={"joint_position_dynamics": np.arange(46.0).reshape(2, 23)})
    f73, f5 = str(tmp_path / "a73.mat"), str(tmp_path / "a5.mat")
    save_mat(f73, output=out, guess=guess)                 # default: what the reference writes
    save_mat(f5, output=out, guess=guess, format="5")
    assert mat73.is_v73(f73) and not mat73.is_v73(f5)
    a, b = load_mat(f73), load_mat(f5)

    def same(x, y, path=""):
        if isinstance(x, dict):
            assert isinstance(y, dict) and set(x) == set(y), path
            for k in x:
                same(x[k], y[k], path + "/" + k)
        elif isinstance(x, list):
            assert len(x) == len(y), path
            for i, (u, v) in enumerate(zip(x, y)):
                same(u, v, "%s[%d]" % (path, i))
        else:
            assert np.array_equal(np.asarray(x, dtype=float).reshape(-1), np.asarray(y, dtype=float).reshape(-1)), path
    same(a, b)
    assert float(a["output"]["cost_values"]["system"]["contact_points"]["left[0]"]["f_regularization"]) == 2.0
    assert np.allclose(a["output"]["constraint_multipliers"]["joint_position_dynamics"], np.arange(46.0).reshape(2, 23))


def test_refuses_what_the_format_cannot_hold(tmp_path):
    with pytest.raises(mat73.Mat73Error):
        mat73.savemat(str(tmp_path / "x.mat"), {"a": {"b/c": 1.0}})
    with pytest.raises(mat73.Mat73Error):
        mat73.savemat(str(tmp_path / "y.mat"), {"z": np.array([1 + 2j])})
    open(tmp_path / "n.mat", "wb").write(b"not a mat file")
    with pytest.raises(mat73.Mat73Error):
        mat73.loadmat(str(tmp_path / "n.mat"))


def _random_tree(rng, depth=0):
    kind = rng.randint(9 if depth < 3 else 6)
    if kind == 0:
        return float(rng.standard_normal())
    if kind == 1:
        shape = tuple(int(v) for v in rng.randint(0, 4, size=rng.randint(1, 4)))
        return rng.standard_normal(shape)
    if kind == 2:
        return "".join(chr(int(c)) for c in rng.choice([65, 97, 0x00e9, 0x4e2d, 0x1f600, 32, 48], size=rng.randint(0, 6)))
    if kind == 3:
        return bool(rng.randint(2))
    if kind == 4:
        return rng.randint(-5, 5, size=(2, 3)).astype([np.int8, np.uint16, np.int32, np.int64, np.uint8][rng.randint(5)])
    if kind == 5:
        return None
    if kind in (6, 7):
        return {("k%d" % i if rng.randint(2) else "odd name[%d]" % i): _random_tree(rng, depth + 1) for i in range(rng.randint(0, 4))}
    return [_random_tree(rng, depth + 1) for _ in range(rng.randint(0, 4))]


def _same(a, b, path="/"):
    """what loadmat (squeezing) must give back for what savemat was handed"""
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), path
        for k in a:
            _same(a[k], b[k], path + k + "/")
    elif isinstance(a, list):
        assert isinstance(b, list) and len(a) == len(b), (path, a, b)
        for i, (u, v) in enumerate(zip(a, b)):
            _same(u, v, "%s[%d]/" % (path, i))
    elif isinstance(a, str):
        assert b == a, (path, a, b)
    elif a is None:
        assert isinstance(b, np.ndarray) and b.size == 0, path
    else:
        x = np.asarray(a)
        assert isinstance(b, np.ndarray), (path, type(b))
        if x.size == 0:
            assert b.size == 0, path
        else:
            assert b.dtype == (np.bool_ if x.dtype == np.bool_ else x.dtype if x.dtype.kind in "iu" else np.float64), (path, b.dtype, x.dtype)
            assert np.array_equal(np.squeeze(x), b), path


@pytest.mark.parametrize("seed", range(12))
def test_random_nested_structures_round_trip(tmp_path, seed):
    """seeded random trees of structs / cells / arrays of 0-3 dimensions (empty ones included) / integers / logicals / strings with
    characters outside the BMP / unset leaves, through a file and back"""
    rng = np.random.RandomState(100 + seed)
    d = {"v%d" % i: _random_tree(rng) for i in range(5)}
    f = mat73.savemat(str(tmp_path / ("r%d.mat" % seed)), d, compress=bool(seed % 2))
    _same(d, mat73.loadmat(f))
