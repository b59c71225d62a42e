#!/usr/bin/env python3
"""A/B of skinny-GEMM variants in one process (in-graph time per launch): LayerNorm-folded vs plain on the same shape, etc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "default"]
src = open(os.path.join(ROOT, "tools", "skinny_bench.py")).read().split("SHAPES =")[0]
exec(src)
for rep in range(2):
    run("fc1 ln=1", 2, 2048, 512, True, 0)
    run("fc1 ln=1 no stat loads", 2, 2048, 512, True, 1)
    run("fc1 ln=0", 2, 2048, 512, False, 0)
    run("qkv-like ln=1", 2, 1536, 512, True, 0)
    run("qkv-like ln=1 no stat loads", 2, 1536, 512, True, 1)
    run("qkv-like ln=0", 2, 1536, 512, False, 0)
