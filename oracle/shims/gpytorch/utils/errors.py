class NanError(RuntimeError):
    pass


class NotPSDError(RuntimeError):
    pass
