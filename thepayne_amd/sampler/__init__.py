from .nested import NestedSampler, Results  # noqa: F401
