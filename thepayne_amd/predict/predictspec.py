"""LinNet / SMLP predictor: mirrors Payne/predict/predictspec.py (ANN, PayneSpecPredict)."""
from ._spec import PayneSpecPredict as _Base, SpecANN, speedoflight  # noqa: F401


class ANN(SpecANN):
    """``ANN(nnpath, NNtype=...)`` of predictspec.py:29-74."""

    def __init__(self, nnpath=None, **kwargs):
        super(ANN, self).__init__(nnpath, kwargs.get('NNtype', 'LinNet'),
                                  **{k: v for k, v in kwargs.items() if k in ('b_max', 'device')})
        self.inlabels = ['teff', 'logg', 'feh', 'afe'][:self.n_labels]


class PayneSpecPredict(_Base):
    default_NNtype = "LinNet"
