"""SED predictor: mirrors Payne/predict/predictsed.py."""
import json
import os

import numpy as np

from .. import nnio
from ..engine import PayneEngine
from .photANN import fastANN
from .highred import highAv

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "highav_table.json")) as _fh:
    _ALLFILTERS = list(json.load(_fh)["filters"].keys())      # predictsed.py:8-23


class FastPayneSEDPredict(object):
    """``sed(logt, logg, feh, afe, logl, av, rv, dist, logA, band_indices)`` ->
    magnitudes (predictsed.py:64-103).  New: ``sed_batch(pars[B,9])``."""

    def __init__(self, usebands=None, nnpath=None, device=None, b_max=256):
        if usebands is None:
            usebands = _ALLFILTERS
        self.filternames = list(usebands)
        self.stack = nnio.load_phot_nets(self.filternames, nnpath)
        self.anns = fastANN(self.stack, self.filternames, device=device, b_max=b_max)
        self.engine = self.anns.engine
        self.HiAv = highAv(self.filternames)

    def sed(self, logt=None, logg=None, feh=None, afe=None, logl=None, av=0.0, rv=3.1,
            dist=None, logA=None, band_indices=slice(None)):
        if not ((logl is not None and dist is not None) or (logA is not None)):
            raise IOError('cannot understand input pars into sed function')
        nan = np.nan
        row = [logt, logg, feh, afe, av, rv,
               nan if logl is None else logl, nan if dist is None else dist, nan if logA is None else logA]
        m = self.engine.sed_batch(np.array([row], dtype=np.float64)).cpu().numpy()[0]
        try:
            return m[band_indices]
        except IndexError:
            return [m]

    def sed_batch(self, pars):
        """pars[B, 9] = logt,logg,feh,afe,av,rv,logl,dist,logA (NaN = absent) -> [B, F] device tensor."""
        return self.engine.sed_batch(pars)


class PayneSEDPredict(FastPayneSEDPredict):
    """The reference's per-filter loop version (predictsed.py:25-62); same numbers."""

    def __init__(self, usebands=None, nnpath=None, **kw):
        super(PayneSEDPredict, self).__init__(usebands=usebands, nnpath=nnpath, **kw)

    def sed(self, logt=None, logg=None, feh=None, afe=None, logl=None, av=0.0, rv=None,
            dist=None, logA=None, filters=None):
        m = super(PayneSEDPredict, self).sed(logt=logt, logg=logg, feh=feh, afe=afe, logl=logl, av=av,
                                             rv=3.1 if rv is None else rv, dist=dist, logA=logA)
        if filters is None:
            return m
        return np.array([m[self.filternames.index(f)] for f in filters])
