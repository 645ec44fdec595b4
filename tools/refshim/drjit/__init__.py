"""`import drjit as dr` of the reference, resolved to the torch-backed stand-in (tools/refshim/_core.py: what it is and is not)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _core import *                                                                   # noqa: F401,F403,E402
from _core import (abs_ as abs, all_ as all, any_ as any, eval_ as eval, max_ as max, min_ as min,   # noqa: F401,E402,A004
                   inf, pi, JitFlag, ADMode, Loop)
