"""Root cause of the fuzz cases of round 1 that failed the `s` comparison (gpurun_out/fuzz1.log of round 1: cases 2, 3, 9, 12,
19, 22, 27, 33 of `scripts/fuzz_parity.py 40 1`; the draws of seed 1 reproduce the logged configurations).  Prints, per
case, the figures tests/test_gpu_fuzz.py asserts on: which singular values differ and how large they are, whether the
tile decisions agree, and the singular values of the data projected on the span of the PASSING tile components only.
    python scripts/fuzz_explain.py [case ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import draw_cases, run_case
from localmd_amd._lib import Context

if __name__ == "__main__":
    want = [int(a) for a in sys.argv[1:]] or [2, 3, 9, 12, 19, 22, 27, 33]
    ctx = Context(0)
    for c in draw_cases(max(want) + 1):
        if c[0] in want:
            run_case(ctx, *c, out=lambda s: print(s, flush=True))
