"""`redistance(phi)` (python/redistancing.py:4-13).  The reference forwards to the native
`fastsweep` package; here the HIP solver of libdsdf.so does the work."""
import dsdf


def redistance(phi, method='fastsweep'):
    if method != 'fastsweep':
        raise ValueError("Invalid re-distancing method")      # skfmm ('fmm') is not available
    return dsdf.redistance(phi)
