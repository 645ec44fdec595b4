#!/usr/bin/env python3
"""Builds the kernel A/B candidates (lib/variants/libdsdf_<tag>.so) with hipcc -- here, on the CPU-only container; the
.so files travel to the GPU box with the gpurun snapshot.  tools/ab_variants.sh times them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

AB = dict(sys.argv[i].split('=', 1) for i in range(1, len(sys.argv)) if '=' in sys.argv[i] and not sys.argv[i].startswith('-'))
for tag, defs in AB.items():
    print(tag, g.build_variant(tag, defs.split(), force=True))
