"""Photometric bolometric-correction nets: mirrors Payne/predict/photANN.py.

``ANN(ff, nnpath)`` = one filter's 6->H->H->1 sigmoid net; ``fastANN(nnlist, bandlist)``
= the stacked evaluator.  Both evaluate on the GPU (payne_sed_kernel, mode 2)."""
import numpy as np

from .. import nnio
from ..engine import PayneEngine


class fastANN(object):
    def __init__(self, nnlist, bandlist, device=None, b_max=256):
        self.filternames = list(bandlist)
        if isinstance(nnlist, dict):
            self.stack = nnlist
        else:
            self.stack = nnio.stack_phot_nets([n.arrays for n in nnlist], self.filternames)
        for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
            setattr(self, k, self.stack[k])
        self.xmin, self.xmax = self.stack["xmin"], self.stack["xmax"]
        self.range = self.xmax - self.xmin
        self.engine = PayneEngine(phot=self.stack, b_max=b_max, device=device)

    def eval(self, x):
        """x = [Teff, logg, feh, afe, av, rv] (or [B,6]) -> BC per filter (photANN.py:125-131)."""
        x = np.asarray(x, dtype=np.float64)
        out = self.engine.bc_batch(np.atleast_2d(x)).cpu().numpy()
        return np.squeeze(out[0] if x.ndim == 1 else out)


class ANN(object):
    def __init__(self, ff, nnpath=None, **kwargs):
        self.verbose = kwargs.get('verbose', True)
        if nnpath is None:
            raise IOError("no default photometric ANNs ship with this build; pass nnpath=")
        self.nnpath = nnpath
        stack = nnio.load_phot_nets([ff], nnpath)
        self.arrays = {"lin1.weight": stack["w1"][0], "lin1.bias": stack["b1"][0, :, 0],
                       "lin2.weight": stack["w2"][0], "lin2.bias": stack["b2"][0, :, 0],
                       "lin3.weight": stack["w3"][0], "lin3.bias": stack["b3"][0, :, 0],
                       "xmin": stack["xmin"], "xmax": stack["xmax"]}
        self.D_in, self.H, self.D_out = 6, stack["w1"].shape[1], 1
        self._fast = None
        self._ff = ff
        self._kw = {k: v for k, v in kwargs.items() if k in ('device', 'b_max')}

    def eval(self, x):
        if self._fast is None:
            self._fast = fastANN([self], [self._ff], **self._kw)
        return self._fast.eval(x)
