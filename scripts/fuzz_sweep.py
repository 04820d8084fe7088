"""Fresh seeded draws through the assertions of tests/test_gpu_fuzz.py (beyond the ten cases of the graded suite).
    python scripts/fuzz_sweep.py FAMILY SEED N [case ...]    FAMILY = base (the draws of round 1) | wide (many tiles, R > frames) | options (the rest of the option space)"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import draw_cases, draw_wide_cases, draw_option_cases, run_case, check_case
from localmd_amd._lib import Context

if __name__ == "__main__":
    family, seed, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    only = [int(a) for a in sys.argv[4:]]
    cases = {"base": draw_cases, "wide": draw_wide_cases, "options": draw_option_cases}[family](n, seed)
    if only:
        cases = [c for c in cases if c[0] in only]
    ctx = Context(0)
    failed = []
    for c in cases:
        t0 = time.time()
        lines = []
        try:
            fig = run_case(ctx, *c, out=lines.append)
            d = fig["diag"]
            lines.append(f"   route: R = {d.get('n_tile_cols')} tile columns, components {len(fig['pmd'].s)}, eig order {d.get('eig_order')}")
            check_case(fig)
            print("\n".join(lines) + f"\n   OK ({time.time() - t0:.1f} s)", flush=True)
        except Exception:
            failed.append(c[0])
            print("\n".join(lines) + "\n   FAILED:\n" + traceback.format_exc(), flush=True)
    print(f"{family} seed {seed}: {len(failed)} of {len(cases)} cases failed {failed}", flush=True)
