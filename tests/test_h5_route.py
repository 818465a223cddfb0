"""The HDF5 route: the only way a real ThePayne network reaches this build.

h5py is absent from the main interpreter, but another interpreter of this image may have it
(/opt/conda/bin/python3.9; override with PAYNE_H5_PYTHON).  That interpreter writes files in the
reference's three layouts --
  * YST1: flat keys w_array_{0,1,2}, b_array_{0,1,2}, x_min, x_max, wavelength, resolution[1]
    (Payne/predict/ystpred.py:22-38),
  * LinNet / SMLP: model/lin{1..6}.{weight,bias} / model/features.{0,2,4,6}.{weight,bias}, xmin, xmax, label_i,
    wavelengths, resolution (Payne/train/NNmodels.py:44-89, Payne/train/trainspec.py:214-232,
    Payne/predict/predictspec.py:43-49),
  * photometric nets: one nnMIST_<filter>.h5 per filter with model/lin{1,2,3}.{weight,bias}, xmin, xmax
    (Payne/predict/photANN.py:60-80)
-- runs `python -m thepayne_amd.nnio` (convert_h5_to_npz) on them, and the main interpreter's loaders must
return the arrays that went in."""
import os
import subprocess
import sys

import numpy as np
import pytest

from thepayne_amd import nnio, synth, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5PY = os.environ.get("PAYNE_H5_PYTHON", "/opt/conda/bin/python3.9")


def _has_h5py():
    if not os.path.exists(H5PY):
        return False
    r = subprocess.run([H5PY, "-c", "import h5py, numpy"], capture_output=True)
    return r.returncode == 0


pytestmark = pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py in this image")

WRITER = r"""
import sys, numpy as np, h5py
src, dst = sys.argv[1], sys.argv[2]
z = np.load(src)
with h5py.File(dst, 'w') as f:
    for k in z.files:
        a = z[k]
        if k in ('testpred',):
            f.create_dataset(k, data=a, compression='gzip')     # trainspec.py:216-219 compresses these
        elif a.ndim == 0:
            f.create_dataset(k, data=a[()])
        else:
            f.create_dataset(k, data=a)
"""


def _write_h5(tmp_path, name, arrays):
    """arrays -> <name>.h5 through the h5py interpreter ('/' in a key makes HDF5 groups, as torch state dicts are stored)."""
    stage = str(tmp_path / (name + "_stage.npz"))
    np.savez(stage, **arrays)
    dst = str(tmp_path / (name + ".h5"))
    subprocess.run([H5PY, "-c", WRITER, stage, dst], check=True, capture_output=True)
    return dst


def _convert(h5, npz=None):
    cmd = [H5PY, "-m", "thepayne_amd.nnio", h5] + ([npz] if npz else [])
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout.strip().splitlines()[-1]
    assert os.path.exists(out)
    return out


@pytest.mark.parametrize("kk", [False, True])
def test_yst1_file_round_trip(tmp_path, kk):
    raw = synth.make_yst_net(npix=256, H=24, seed=3)
    arrs = {k: v for k, v in raw.items() if k != "kind"}
    arrs["resolution"] = np.array([raw["resolution"]])                 # stored as a 1-vector (ystpred.py:36 takes [0])
    if kk:                                                               # a network trained on Teff/1000 (ystpred.py:76-79)
        arrs["x_min"] = raw["x_min"].copy(); arrs["x_max"] = raw["x_max"].copy()
        arrs["x_min"][0] /= 1000.0; arrs["x_max"][0] /= 1000.0
    h5 = _write_h5(tmp_path, "yst", arrs)
    npz = _convert(h5)
    assert npz == str(tmp_path / "yst.npz")
    got = nnio.load_spec_net(npz, "YST1")
    want = nnio.normalize_spec_net(raw, "YST1")
    for (w, b, a), (w0, b0, a0) in zip(got["layers"], want["layers"]):
        assert w.dtype == np.float32 and np.array_equal(w, w0) and np.array_equal(b, b0) and a == a0
    np.testing.assert_allclose(got["xmin"], want["xmin"], rtol=1e-15)   # the kK file comes back in K for the SPECTRAL network ...
    np.testing.assert_allclose(got["xmax"], want["xmax"], rtol=1e-15)
    assert np.array_equal(got["wavelength"], want["wavelength"]) and got["resolution"] == want["resolution"]
    if kk:                                                               # ... and stays as stored for a continuum network (ystpred.py:81-85)
        asis = nnio.load_spec_net(npz, "YST1", rescale_teff=False)
        assert asis["xmin"][0] == arrs["x_min"][0] and asis["xmax"][0] == arrs["x_max"][0]


@pytest.mark.parametrize("kind", ["LinNet", "SMLP"])
def test_torch_state_dict_file_round_trip(tmp_path, kind):
    raw = synth.make_torch_net(kind, npix=128, H=(16, 12, 8), seed=5)
    arrs = {}
    for k, v in raw.items():
        if k == "kind":
            continue
        if k.endswith(".weight") or k.endswith(".bias"):
            arrs["model/" + k] = v                                       # NNmodels.py:58-63: the state dict lives in group 'model'
        elif k == "wavelength":
            arrs["wavelengths"] = v                                      # trainspec.py:222, predictspec.py:47
        elif k == "resolution":
            arrs[k] = np.array(v)                                        # 0-d dataset (trainspec.py:224)
        else:
            arrs[k] = v
    arrs["label_i"] = np.array([x.encode("ascii") for x in ("teff", "logg", "feh", "afe")])
    arrs["testpred"] = np.zeros((3, 128), dtype=np.float32)              # present in real files; ignored by the loaders
    h5 = _write_h5(tmp_path, "nn", arrs)
    npz = _convert(h5, str(tmp_path / "converted.npz"))
    loaded = nnio.load_arrays(npz)
    assert "model/lin1.weight" in loaded or "model/features.0.weight" in loaded
    assert [x.decode() for x in loaded["label_i"]] == ["teff", "logg", "feh", "afe"]
    got = nnio.load_spec_net(npz, kind)
    want = nnio.normalize_spec_net(raw, kind)
    assert len(got["layers"]) == len(want["layers"]) == (6 if kind == "LinNet" else 4)
    for (w, b, a), (w0, b0, a0) in zip(got["layers"], want["layers"]):
        assert np.array_equal(w, w0) and np.array_equal(b, b0) and a == a0
    assert got["layers"][-1][2] == _lib.ACT_NONE
    assert np.array_equal(got["xmin"], want["xmin"]) and np.array_equal(got["wavelength"], want["wavelength"])
    assert got["resolution"] == want["resolution"]


def test_photometric_nets_per_filter_files(tmp_path):
    filters = synth.PHOT_FILTERS[:4]
    stacked = synth.make_phot_nets(filters, H=16, seed=2)
    d = tmp_path / "photANN"
    d.mkdir()
    for i, f in enumerate(filters):
        arrs = {"model/lin1.weight": stacked["w1"][i], "model/lin1.bias": stacked["b1"][i, :, 0],
                "model/lin2.weight": stacked["w2"][i], "model/lin2.bias": stacked["b2"][i, :, 0],
                "model/lin3.weight": stacked["w3"][i], "model/lin3.bias": stacked["b3"][i, :, 0],
                "xmin": stacked["xmin"], "xmax": stacked["xmax"]}
        _convert(_write_h5(d, "nnMIST_" + f, arrs))
        os.remove(str(d / ("nnMIST_%s.h5" % f)))                          # only the converted container remains
    got = nnio.load_phot_nets(filters, str(d) + os.sep)
    for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
        assert got[k].shape == stacked[k].shape and np.array_equal(got[k], stacked[k]), k
    assert np.array_equal(got["xmin"], stacked["xmin"]) and np.array_equal(got["xmax"], stacked["xmax"])
    # a subset in another order, from the stacked dict (FitPayne passes the filters of the observed photometry)
    sub = nnio.load_phot_nets([filters[2], filters[0]], got)
    assert np.array_equal(sub["w2"][0], stacked["w2"][2]) and np.array_equal(sub["w2"][1], stacked["w2"][0])


def test_h5_without_h5py_says_what_to_do(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"\x89HDF\r\n\x1a\n")
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py importable here")
    except ImportError:
        pass
    with pytest.raises(IOError, match="convert it to .npz"):
        nnio.load_arrays(str(p))
