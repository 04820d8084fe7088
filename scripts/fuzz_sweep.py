"""Fresh seeded draws through the assertions of tests/test_gpu_fuzz.py (beyond the ten cases of the graded suite).
    python scripts/fuzz_sweep.py FAMILY SEED N [case ...]    FAMILY = base (the draws of round 1) | wide (many tiles, R > frames) | options (the rest of the option space) | edge (extremes; both sides must refuse the same inputs) | medium (larger movies, minutes of oracle time) | tall (config-2-like: R several times the frames) | widecomp (max_components 55-110, background_rank up to 64: the generic-width path)"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import draw_cases, draw_wide_cases, draw_option_cases, draw_edge_cases, draw_medium_cases, draw_tall_cases, draw_widecomp_cases, run_case, check_case
from localmd_amd._lib import Context

if __name__ == "__main__":
    family, seed, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    only = [int(a) for a in sys.argv[4:]]
    cases = {"base": draw_cases, "wide": draw_wide_cases, "options": draw_option_cases, "edge": draw_edge_cases, "medium": draw_medium_cases, "tall": draw_tall_cases, "widecomp": draw_widecomp_cases}[family](n, seed)
    if only:
        cases = [c for c in cases if c[0] in only]
    ctx = Context(0)
    failed = []
    for c in cases:
        t0 = time.time()
        lines = []
        try:
            if family == "edge":
                # the oracle first: what it refuses (ValueError, as the reference), the HIP path has to refuse too
                import numpy as np
                import localmd_amd
                from localmd_amd import decomposition as Dm
                from localmd_amd.synthetic import make_movie
                from oracle import pmd_oracle as O
                from tests.util import DeviceSource
                _, T, d1, d2, b1, b2, frames, kw, extra = c
                mov = make_movie(T, d1, d2, seed=1000 + c[0])
                Dm.QUIET = True
                errs = []
                for side in ("oracle", "hip"):
                    try:
                        np.random.seed(7)
                        if side == "oracle":
                            O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=(1.0, 1.7), **kw)
                        else:
                            localmd_amd.localmd_decomposition(mov, (b1, b2), frames, seed=123, ctx=ctx, thresholds=(1.0, 1.7), **kw)
                        errs.append(None)
                    except Exception as e:     # noqa: BLE001 - the class of the exception is what is compared
                        errs.append(e)
                kinds = [None if e is None else type(e).__name__ for e in errs]
                if kinds[0] is not None or kinds[1] is not None:
                    lines.append(f"case {c[0]}: T={T} fov={d1}x{d2} block={b1}x{b2} frames={frames} {kw}")
                    lines.append(f"   oracle: {kinds[0]} {str(errs[0])[:120] if errs[0] else ''}")
                    lines.append(f"   HIP:    {kinds[1]} {str(errs[1])[:120] if errs[1] else ''}")
                    assert kinds[0] == kinds[1], kinds
                    print("\n".join(lines) + "\n   OK (refused on both sides)", flush=True)
                    continue
            fig = run_case(ctx, *c, out=lines.append)
            d = fig["diag"]
            lines.append(f"   route: R = {d.get('n_tile_cols')} tile columns, components {len(fig['pmd'].s)}, eig order {d.get('eig_order')}")
            check_case(fig)
            print("\n".join(lines) + f"\n   OK ({time.time() - t0:.1f} s)", flush=True)
        except Exception:
            failed.append(c[0])
            print("\n".join(lines) + "\n   FAILED:\n" + traceback.format_exc(), flush=True)
    print(f"{family} seed {seed}: {len(failed)} of {len(cases)} cases failed {failed}", flush=True)
