"""``Payne`` -- the reference's import names, answered by this build.

The reference's callers do ``from Payne.fitting import fitstar`` (demo/runPayne.py:1),
``from Payne.predict.ystpred import PayneSpecPredict`` and so on.  This package holds no code of its
own: every ``Payne.<x>`` module of the likelihood hot path is the
``thepayne_amd.<x>`` module of the same name, registered under both names, so existing scripts
run on the MI355X path without edits.  Modules of the reference outside that path (training, grid
readers, the JAX mirror) do not exist here and raise ImportError as any missing module would.
"""
import importlib
import sys

import thepayne_amd as _impl

__version__ = getattr(_impl, "__version__", "0")
__abspath__ = _impl.__path__[0] + "/"          # the reference's install-time constant (setup.py:38-52)

_MODULES = ["fitting", "fitting.fitstar", "fitting.likelihood", "fitting.prior", "fitting.genmod", "fitting.fitutils",
            "fitting.advancedpriors", "predict", "predict.ystpred", "predict.predictspec", "predict.predictsed",
            "predict.photANN", "predict.highred", "utils", "utils.smoothing"]
for _m in _MODULES:
    try:
        _mod = importlib.import_module("thepayne_amd." + _m)
    except ImportError:
        continue
    sys.modules[__name__ + "." + _m] = _mod
    if "." not in _m:
        globals()[_m] = _mod
del _m, _mod
