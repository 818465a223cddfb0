from .nested import NestedSampler, Results  # noqa: F401
from .dynamic import DynamicNestedSampler  # noqa: F401
