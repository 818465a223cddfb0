"""YST1 predictor: mirrors Payne/predict/ystpred.py (Net, PayneSpecPredict)."""
from ._spec import PayneSpecPredict as _Base, SpecANN, speedoflight  # noqa: F401


class Net(SpecANN):
    """``Net(NNpath)`` of ystpred.py:18-58 (2-hidden-layer leaky-ReLU MLP)."""

    def __init__(self, NNpath, **kw):
        super(Net, self).__init__(NNpath, "YST1", **kw)


class PayneSpecPredict(_Base):
    default_NNtype = "YST1"
