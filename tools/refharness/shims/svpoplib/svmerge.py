def __getattr__(name):
    def _missing(*a, **k):
        raise NotImplementedError(f'svpoplib.{__name__.split(".")[-1]}.{name} is not available in the oracle harness')
    return _missing
