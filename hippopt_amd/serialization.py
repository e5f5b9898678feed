"""Result / guess serialisation with the dictionary layout of the reference's main scripts
(`{"output": output.to_dict(), "guess": guess.to_dict(flatten=False)}`, main_periodic_step.py:503-513; Output.to_dict:
base/problem.py:58-79).

The reference stores that dictionary with `hdf5storage.savemat` (MAT v7.3 = HDF5).  Neither hdf5storage nor h5py is in this
image, but the HDF5 C library is: `mat73.py` binds it with ctypes and writes / reads MAT v7.3 in hdf5storage's MATLAB-compatible
layout (structs as groups, cells as object references into "/#refs#", reversed dimensions, MATLAB_class attributes, the MAT
header in the user block) — the default container where the library is found.  Without it (or with `format="5"`) the container
is MAT v5 through `scipy.io.savemat` — same nested struct / cell layout, loadable in MATLAB and with `scipy.io.loadmat`
(scipy truncates names > 31 chars unless long_field_names is set, which it is).  `load_mat` recognises either."""
import numpy as np


def _plain(obj):
    """OptimizationObject dictionaries -> nested dict / list / float ndarray (None -> empty array, like DM conversion of an unset leaf)."""
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        items = [_plain(v) for v in obj]
        arr = np.empty(len(items), dtype=object)
        for i, v in enumerate(items):
            arr[i] = v
        return arr
    if obj is None:
        return np.zeros((0, 0))
    if isinstance(obj, (str, bytes)):
        return obj
    return np.asarray(obj, dtype=float)


def to_mat_dict(output=None, guess=None, **extra) -> dict:
    d = {}
    if output is not None:
        d["output"] = _plain(output.to_dict())
    if guess is not None:
        if isinstance(guess, list):
            d["guess"] = _plain([g.to_dict(flatten=False) for g in guess])
        else:
            d["guess"] = _plain(guess.to_dict(flatten=False))
    for k, v in extra.items():
        d[k] = _plain(v)
    return d


def save_mat(file_name: str, output=None, guess=None, format: str = None, **extra) -> dict:
    """format: "7.3" (what the reference writes; needs the HDF5 C library), "5", or None = "7.3" where possible."""
    from . import mat73
    d = to_mat_dict(output, guess, **extra)
    if format is None:
        format = "7.3" if mat73.available() else "5"
    if format == "7.3":
        mat73.savemat(file_name, d, appendmat=False)
    elif format == "5":
        from scipy.io import savemat
        savemat(file_name, d, long_field_names=True, oned_as="column")
    else:
        raise ValueError("format is '7.3' or '5'")
    return d


def load_mat(file_name: str) -> dict:
    """Back to nested dicts / lists / ndarrays (structs -> dict, cell arrays -> list)."""
    from . import mat73
    if mat73.is_v73(file_name):
        return mat73.loadmat(file_name)
    from scipy.io import loadmat
    raw = loadmat(file_name, struct_as_record=False, squeeze_me=True)

    def conv(o):
        if hasattr(o, "_fieldnames"):
            return {k: conv(getattr(o, k)) for k in o._fieldnames}
        if isinstance(o, np.ndarray) and o.dtype == object:
            return [conv(v) for v in o.reshape(-1)]
        return o
    return {k: conv(v) for k, v in raw.items() if not k.startswith("__")}
