from . import utils, folder  # noqa: F401
