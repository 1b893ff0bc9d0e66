from . import datasets  # noqa: F401
