from . import transforms, quadrature, broadcasting, cholesky, errors, warnings  # noqa: F401
