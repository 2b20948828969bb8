def dotplot(*a, **k):
    raise NotImplementedError('kanapy.plot.dotplot is not available in the oracle harness')
