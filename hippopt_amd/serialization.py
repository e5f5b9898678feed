"""Result / guess serialisation with the dictionary layout of the reference's main scripts
(`{"output": output.to_dict(), "guess": guess.to_dict(flatten=False)}`, main_periodic_step.py:503-513; Output.to_dict:
base/problem.py:58-79).

The reference stores that dictionary with `hdf5storage.savemat` (MAT v7.3 = HDF5).  Neither hdf5storage nor h5py is in this
image, so the container here is MAT v5 through `scipy.io.savemat` — same nested struct / cell layout, loadable in MATLAB and with
`scipy.io.loadmat`; a v7.3 writer can be swapped in behind `save_mat` where h5py exists.  Keys that are not valid MATLAB
field names are kept as they are on the reference side too (hdf5storage escapes them; scipy truncates names > 31 chars unless
long_field_names is set, which it is)."""
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


def save_mat(file_name: str, output=None, guess=None, **extra) -> dict:
    from scipy.io import savemat
    d = to_mat_dict(output, guess, **extra)
    savemat(file_name, d, long_field_names=True, oned_as="column")
    return d


def load_mat(file_name: str) -> dict:
    """Back to nested dicts / lists / ndarrays (structs -> dict, cell arrays -> list)."""
    from scipy.io import loadmat
    raw = loadmat(file_name, struct_as_record=False, squeeze_me=True)

    def conv(o):
        if hasattr(o, "_fieldnames"):
            return {k: conv(getattr(o, k)) for k in o._fieldnames}
        if isinstance(o, np.ndarray) and o.dtype == object:
            return [conv(v) for v in o.reshape(-1)]
        return o
    return {k: conv(v) for k, v in raw.items() if not k.startswith("__")}
