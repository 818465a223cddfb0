"""ANN weight containers.

The reference stores its emulators in HDF5 files (keys: YST1
Payne/predict/ystpred.py:22-38; LinNet/SMLP Payne/train/NNmodels.py:44-89 +
Payne/predict/predictspec.py:45-49; photometric nets Payne/predict/photANN.py:60-80).
h5py is not part of this image's main interpreter, so the native container here is an
``.npz`` with the SAME key names ('/' in HDF5 group paths kept verbatim); ``.h5`` files are
read when h5py is importable.  ``convert_h5_to_npz`` turns one into the other; it needs only
numpy + h5py, so it runs under any interpreter that has them:

    /opt/conda/bin/python3.9 -m thepayne_amd.nnio NN.h5 [NN.npz]        (from the repository root)

(tests/test_h5_route.py does exactly that with files in the reference's three layouts.)
"""
import os

import numpy as np

from . import _lib

__all__ = ["load_arrays", "save_npz", "convert_h5_to_npz", "normalize_spec_net", "load_spec_net",
           "load_phot_nets", "stack_phot_nets"]


def _read_h5(path):
    try:
        import h5py
    except ImportError:
        raise IOError("%s is HDF5 but h5py is not installed; convert it to .npz "
                      "(thepayne_amd.nnio.convert_h5_to_npz) where h5py exists" % path)
    out = {}

    def visit(name, obj):
        if hasattr(obj, "shape"):
            a = np.array(obj)
            if a.dtype == object:              # variable-length strings (label names): fixed-length bytes, loadable without pickle
                a = np.array([x if isinstance(x, bytes) else str(x).encode("utf-8") for x in a.ravel()]).reshape(a.shape)
            out[name] = a
    with h5py.File(path, "r") as f:
        f.visititems(visit)
    return out


def load_arrays(path):
    """{key: ndarray} from .npz or .h5/.hdf5."""
    if not os.path.exists(path):
        raise IOError("cannot find ANN file %s" % path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    return _read_h5(path)


def save_npz(path, arrays):
    np.savez(path, **{k: np.asarray(v) for k, v in arrays.items() if not isinstance(v, (str, list))})


def convert_h5_to_npz(h5path, npzpath=None):
    arrs = _read_h5(h5path)
    npzpath = npzpath or os.path.splitext(h5path)[0] + ".npz"
    np.savez(npzpath, **arrs)
    return npzpath


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def normalize_spec_net(arrs, NNtype="YST1", rescale_teff=True):
    """Arrays of one spectral emulator -> dict(layers=[(W,b,act)...], xmin, xmax,
    wavelength, resolution, kind).  Accepts the file key names and the in-memory
    names used by thepayne_amd.synth.  ``rescale_teff``: the Teff/1000 -> Teff fix of x_min/x_max that
    PayneSpecPredict applies to the SPECTRAL network only (ystpred.py:76-79); the continuum network
    (``Canns``, ystpred.py:81-85) is used as stored, so its loader passes False."""
    g = lambda *names: next((np.asarray(arrs[n]) for n in names if n in arrs), None)
    if NNtype in ("YST1", "YST2"):
        layers = [(_f32(arrs["w_array_%d" % i]), _f32(arrs["b_array_%d" % i]),
                   _lib.ACT_LRELU if i < 2 else _lib.ACT_NONE) for i in range(3)]
        xmin = np.array(g("x_min"), dtype=np.float64).copy()
        xmax = np.array(g("x_max"), dtype=np.float64).copy()
        if rescale_teff and xmin[0] < 1000.0:     # Teff/1000 convention, ystpred.py:76-79
            xmin[0] *= 1000.0
            xmax[0] *= 1000.0
        wave = g("wavelength")
        res = np.atleast_1d(g("resolution"))[0]    # ystpred.py:36
    elif NNtype == "LinNet":
        p = "model/" if "model/lin1.weight" in arrs else ""
        layers = [(_f32(arrs[p + "lin%d.weight" % i]), _f32(arrs[p + "lin%d.bias" % i]),
                   _lib.ACT_SIGMOID if i < 6 else _lib.ACT_NONE) for i in range(1, 7)]
        xmin, xmax = np.array(g("xmin"), dtype=np.float64), np.array(g("xmax"), dtype=np.float64)
        wave = g("wavelengths", "wavelength")
        res = float(np.atleast_1d(g("resolution"))[0])
    elif NNtype == "SMLP":
        p = "model/" if "model/features.0.weight" in arrs else ""
        layers = [(_f32(arrs[p + "features.%d.weight" % i]), _f32(arrs[p + "features.%d.bias" % i]),
                   _lib.ACT_LRELU if i < 6 else _lib.ACT_NONE) for i in (0, 2, 4, 6)]
        xmin, xmax = np.array(g("xmin"), dtype=np.float64), np.array(g("xmax"), dtype=np.float64)
        wave = g("wavelengths", "wavelength")
        res = float(np.atleast_1d(g("resolution"))[0])
    else:
        raise IOError("NNtype %r is not supported (the reference's ResNet cannot be constructed either: "
                      "NNmodels.py:38 vs :172)" % (NNtype,))
    return dict(kind=NNtype, layers=layers, xmin=xmin, xmax=xmax,
                wavelength=np.ascontiguousarray(wave, dtype=np.float64), resolution=float(res))


def load_spec_net(nnpath, NNtype="YST1", rescale_teff=True):
    """nnpath: file path, or an already-loaded {key: array} dict."""
    arrs = nnpath if isinstance(nnpath, dict) else load_arrays(nnpath)
    return normalize_spec_net(arrs, NNtype, rescale_teff=rescale_teff)


def stack_phot_nets(per_filter, filters):
    """[{lin1.weight,...,xmin,xmax}] -> the stacked layout of photANN.fastANN
    (Payne/predict/photANN.py:97-116)."""
    def key(d, k):
        return np.asarray(d["model/" + k] if ("model/" + k) in d else d[k])
    F = len(per_filter)
    H = key(per_filter[0], "lin1.weight").shape[0]
    out = {
        "filters": list(filters),
        "w1": _f32(np.stack([key(d, "lin1.weight") for d in per_filter])),
        "b1": _f32(np.stack([key(d, "lin1.bias") for d in per_filter])).reshape(F, H, 1),
        "w2": _f32(np.stack([key(d, "lin2.weight") for d in per_filter])),
        "b2": _f32(np.stack([key(d, "lin2.bias") for d in per_filter])).reshape(F, H, 1),
        "w3": _f32(np.stack([key(d, "lin3.weight") for d in per_filter])).reshape(F, 1, H),
        "b3": _f32(np.stack([key(d, "lin3.bias") for d in per_filter])).reshape(F, 1, 1),
        "xmin": np.array(per_filter[0]["xmin"], dtype=np.float64),     # set_minmax uses the first net
        "xmax": np.array(per_filter[0]["xmax"], dtype=np.float64),
    }
    return out


def load_phot_nets(filters, nnpath):
    """Read nnpath + 'nnMIST_<filter>.{npz,h5}' for each filter (photANN.py:60) and stack them.
    nnpath may also be an already-stacked dict (w1..b3, xmin, xmax, filters)."""
    if isinstance(nnpath, dict):
        if list(nnpath["filters"]) == list(filters):
            return nnpath
        idx = [list(nnpath["filters"]).index(f) for f in filters]
        out = {k: (v[idx] if k in ("w1", "b1", "w2", "b2", "w3", "b3") else v) for k, v in nnpath.items()}
        out["filters"] = list(filters)
        return out
    per = []
    for f in filters:
        base = nnpath + "nnMIST_{0}".format(f)
        path = next((base + ext for ext in (".npz", ".h5") if os.path.exists(base + ext)), None)
        if path is None:
            raise IOError("Cannot find NN file for {0} under {1}".format(f, nnpath))
        per.append(load_arrays(path))
    return stack_phot_nets(per, filters)


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m thepayne_amd.nnio FILE.h5 [FILE.npz]")
    print(convert_h5_to_npz(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))
