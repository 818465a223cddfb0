"""Import the UNMODIFIED reference (/root/reference) in the survey container.

Only ``oracle/gen_golden.py`` (and ``tests/test_reference_live.py``, which
skips when /root/reference is absent) use this.  The reference needs dynesty,
astropy and h5py, none of which are installed; they are stubbed at import time
only: dynesty is touched only inside the sampler loop,
astropy only to parse one embedded text table, h5py only to read ANN files,
which we serve from an in-memory registry ``REG[path] = {key: ndarray}``.
Nothing here ever travels to the GPU box as a dependency.
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
REG = {}


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "Payne"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _Dataset(object):
    def __init__(self, a):
        self.a = np.asarray(a)

    shape = property(lambda self: self.a.shape)

    def __getitem__(self, k):
        return self.a[k]

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __iter__(self):
        return iter(self.a)

    def __len__(self):
        return len(self.a)


class _Group(object):
    def __init__(self, d, prefix=""):
        self.d, self.p = d, prefix

    def __getitem__(self, k):
        full = self.p + k
        if full in self.d:
            return _Dataset(self.d[full])
        if any(x.startswith(full + "/") for x in self.d):
            return _Group(self.d, full + "/")
        raise KeyError(full)

    def keys(self):
        return sorted({x[len(self.p):].split("/")[0] for x in self.d if x.startswith(self.p)})

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass


def _h5file(path, mode="r"):
    if path not in REG:
        raise IOError(path)
    return _Group(REG[path])


def _ascii_read(s):
    rows = [l.split() for l in s.strip().splitlines() if l.strip()]
    hdr, rows = rows[0], rows[1:]
    dt = [(hdr[0], "U32")] + [(h, "f8") for h in hdr[1:]]
    return np.array([tuple([r[0]] + [float(x) for x in r[1:]]) for r in rows], dtype=dt)


_installed = False


def install():
    """Install the stubs and put the reference on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    _stub("dynesty")
    _stub("h5py", File=_h5file)
    a = _stub("astropy")
    _stub("astropy.utils")
    _stub("astropy.utils.exceptions", AstropyWarning=type("AW", (Warning,), {}),
          AstropyDeprecationWarning=type("ADW", (Warning,), {}))
    a.units = _stub("astropy.units")
    _stub("astropy.coordinates", SkyCoord=None, CylindricalRepresentation=None)
    _stub("astropy.table", Table=None, vstack=None, join=None)
    asc = types.SimpleNamespace(read=staticmethod(_ascii_read))
    _stub("astropy.io", ascii=asc)
    _stub("astropy.io.ascii", read=_ascii_read)
    _stub("astropy.io.fits")
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def register_yst(path, net):
    """Expose a synthetic YST1 net as the HDF5 file ystpred.Net.readNN reads."""
    REG[path] = {
        "w_array_0": net["w_array_0"], "w_array_1": net["w_array_1"], "w_array_2": net["w_array_2"],
        "b_array_0": net["b_array_0"], "b_array_1": net["b_array_1"], "b_array_2": net["b_array_2"],
        "x_min": net["x_min"].copy(), "x_max": net["x_max"].copy(),
        "wavelength": net["wavelength"], "resolution": np.array([net["resolution"]]),
    }


def register_torchnet(path, net):
    """LinNet / SMLP file layout (predictspec.py:45-49, NNmodels.py:44-89)."""
    d = {"xmin": net["xmin"], "xmax": net["xmax"], "wavelengths": net["wavelength"],
         "resolution": np.array(net["resolution"]),
         "label_i": np.array([b"teff", b"logg", b"feh", b"afe"])}
    for k, v in net.items():
        if k.endswith(".weight") or k.endswith(".bias"):
            d["model/" + k] = v
    REG[path] = d


def register_phot(dirpath, phot):
    """One nnMIST_<filter>.h5 per filter (photANN.py:60-80)."""
    for i, f in enumerate(phot["filters"]):
        REG[dirpath + "nnMIST_%s.h5" % f] = {
            "model/lin1.weight": phot["w1"][i], "model/lin1.bias": phot["b1"][i, :, 0],
            "model/lin2.weight": phot["w2"][i], "model/lin2.bias": phot["b2"][i, :, 0],
            "model/lin3.weight": phot["w3"][i], "model/lin3.bias": phot["b3"][i, :, 0],
            "xmin": phot["xmin"], "xmax": phot["xmax"],
        }
