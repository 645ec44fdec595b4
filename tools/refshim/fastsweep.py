"""python/redistancing.py imports fastsweep at module level; redistancing is not part of what the stand-in runs."""


def redistance(phi):
    raise NotImplementedError('fastsweep is not available in the stand-in (tools/refshim)')
