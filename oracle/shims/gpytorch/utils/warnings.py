class NumericalWarning(RuntimeWarning):
    pass
