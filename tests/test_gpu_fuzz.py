"""
Bounded, seeded fuzz of the whole decomposition against the oracle (GPU): the random draws of round 1's
`scripts/fuzz_parity.py 40 1`, restricted to a fixed list of cases that includes every case that run reported as failing its
`s` comparison (2, 3, 9, 12, 19, 22, 27, 33; gpurun_out/fuzz1.log of round 1, analysed in profiles/r02_fuzz_explain.txt).

What those failures were (profiles/r02_fuzz_explain.txt): every offending singular value is a noise-level one (sigma ~
30-120 against sigma_1 ~ 550-960; a lone exception in case 19 comes with a knife-edge tile decision).  These draws fit
tiles on only 30-60 temporal bins, so the weak components - those every tile KEEPS although they fail the roughness tests
(evaluation.py:195-222), and the weakest passing ones next to the truncation of the randomised sketch at max_components -
are ill-conditioned functions of the data: fp32 rounding moves them by per cents in ANY fp32 implementation.  The
decisive measurement, encoded here: the same draws through the oracle's float64 arbiter form.  The HIP result's
distance to the arbiter (singular values on the span of the passing tile components, U_data on the stable columns) must
not exceed a small multiple of the distance of the reference's own fp32 arithmetic (the oracle with double- and with
single-precision LAPACK) to the arbiter; tile decisions, CSR structure, shapes and the fit to the data must agree.
"""
import numpy as np
import pytest

from oracle import pmd_oracle as O
from tests import parity_metrics as PM
from tests.util import DeviceSource
import tests.test_gpu_parity as tp

pytestmark = pytest.mark.gpu

FUZZ_SEED = 1
FUZZ_CASES = [0, 2, 3, 5, 9, 12, 19, 22, 27, 33]


def draw_cases(n_cases, seed=FUZZ_SEED):
    """The exact draws of scripts/fuzz_parity.py (round 1)."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        b1, b2 = (int(2 * rng.integers(5, 17)) for _ in range(2))
        d1 = int(rng.integers(b1, 3 * b1 + 8))
        d2 = int(rng.integers(b2, 3 * b2 + 8))
        taf = int(rng.choice([1, 2, 4, 5, 10]))
        T = int(rng.integers(300, 900))
        frames = T if rng.random() < 0.6 else int(rng.integers(260, T))
        kw = dict(max_components=int(rng.integers(2, 11)), background_rank=int(rng.integers(0, 6)), temporal_avg_factor=taf,
                  spatial_avg_factor=int(rng.choice([1, 2, 3])), order=str(rng.choice(["F", "C"])),
                  compute_normalizer=bool(rng.random() < 0.8), max_consecutive_failures=int(rng.choice([1, 1, 2])))
        if rng.random() < 0.25 and frames >= 400:
            wc = int(frames // 2 // taf * taf)
            if wc >= 100:
                kw["window_chunks"] = wc
        out.append((case, T, d1, d2, b1, b2, frames, kw))
    return out


def draw_wide_cases(n_cases, seed):
    """Draws aimed at the R > frames routes of the global stage: many small tiles, few frames."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        b1, b2 = (int(2 * rng.integers(5, 9)) for _ in range(2))
        d1 = int(rng.integers(4 * b1, 7 * b1))
        d2 = int(rng.integers(4 * b2, 7 * b2))
        taf = int(rng.choice([1, 2, 4, 5]))
        T = int(rng.integers(260, 520))
        frames = T if rng.random() < 0.6 else int(rng.integers(256, T))
        kw = dict(max_components=int(rng.integers(4, 13)), background_rank=int(rng.integers(0, 4)), temporal_avg_factor=taf,
                  spatial_avg_factor=int(rng.choice([1, 2])), order=str(rng.choice(["F", "C"])),
                  compute_normalizer=bool(rng.random() < 0.8), max_consecutive_failures=int(rng.choice([1, 1, 2])))
        out.append((case, T, d1, d2, b1, b2, frames, kw))
    return out


def draw_option_cases(n_cases, seed):
    """Draws over the option space the first two families leave out: large max_components / background_rank, blocks up to
    40 x 40, rank_prune, pixel_weighting, window_chunks, integer and float64 input arrays, other noise levels."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        b1, b2 = (int(2 * rng.integers(5, 21)) for _ in range(2))
        d1 = int(rng.integers(b1, int(2.5 * b1) + 8))
        d2 = int(rng.integers(b2, int(2.5 * b2) + 8))
        taf = int(rng.choice([1, 2, 5, 10]))
        saf = int(rng.choice([1, 2, 4]))
        T = int(rng.integers(300, 700))
        frames = T if rng.random() < 0.6 else int(rng.integers(260, T))
        pooled = (-(-b1 // saf)) * (-(-b2 // saf))
        r_max = min(54, frames // taf, pooled)
        kw = dict(max_components=int(rng.integers(min(8, r_max), r_max + 1)), background_rank=int(rng.integers(0, 21)),
                  temporal_avg_factor=taf, spatial_avg_factor=saf, order=str(rng.choice(["F", "C"])),
                  compute_normalizer=bool(rng.random() < 0.8), max_consecutive_failures=int(rng.choice([1, 2, 3])))
        if rng.random() < 0.3:
            kw["rank_prune"] = True
            kw["rank_prune_factor"] = float(rng.choice([0.3, 0.5, 0.7]))
        if rng.random() < 0.3:
            kw["pixel_weighting"] = (0.5 + np.random.default_rng(seed * 1000 + case).random((d1, d2))).astype(np.float32)
        if rng.random() < 0.25 and frames >= 400:
            wc = int(frames // 2 // taf * taf)
            if wc >= 100:
                kw["window_chunks"] = wc
        extra = {"noise": float(rng.choice([0.5, 1.0, 2.0])), "dtype": str(rng.choice(["float32", "float32", "uint16", "float64"]))}
        out.append((case, T, d1, d2, b1, b2, frames, kw, extra))
    return out


def draw_widecomp_cases(n_cases, seed):
    """Draws on the generic-width tile path (round 3): max_components 55 ... 110 (128-row tiles), background_rank up to 90
    (generic-width background rSVD above 54, blocks of 64 background columns above 64), with the other options of draw_option_cases - residual windows, rank_prune,
    pixel weights, both pixel orders, several consecutive failures (which is what lets a tile keep more than 64 components)."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        b1, b2 = (int(2 * rng.integers(7, 17)) for _ in range(2))
        d1 = int(rng.integers(b1, int(2.2 * b1) + 6))
        d2 = int(rng.integers(b2, int(2.2 * b2) + 6))
        taf = int(rng.choice([2, 5, 10]))
        saf = int(rng.choice([1, 2]))
        T = int(rng.integers(700, 1300))
        frames = T if rng.random() < 0.6 else int(rng.integers(600, T))
        pooled = (-(-b1 // saf)) * (-(-b2 // saf))
        r_max = min(110, frames // taf - 1, max(56, pooled))
        r = int(rng.integers(55, max(56, r_max) + 1))
        kw = dict(max_components=r, background_rank=int(rng.choice([0, 3, 15, 40, 58, 90])),
                  temporal_avg_factor=taf, spatial_avg_factor=saf, order=str(rng.choice(["F", "C"])),
                  max_consecutive_failures=int(rng.choice([1, 1, 3, 100])))
        if rng.random() < 0.25:
            kw["rank_prune"] = True
            kw["rank_prune_factor"] = float(rng.choice([0.3, 0.6]))
        if rng.random() < 0.25:
            kw["pixel_weighting"] = (0.5 + np.random.default_rng(seed * 1000 + case).random((d1, d2))).astype(np.float32)
        if rng.random() < 0.3:
            wc = int(frames // 2 // taf * taf)
            if wc // taf > r:
                kw["window_chunks"] = wc
        extra = {"noise": float(rng.choice([0.5, 1.0])), "dtype": "float32"}
        out.append((case, T, d1, d2, b1, b2, frames, kw, extra))
    return out


def draw_edge_cases(n_cases, seed):
    """Draws at the edges of the argument space: fewer than 256 frames (one Welch segment), FOV equal to or barely larger
    than a block, frame counts that are no multiple of the temporal factor, one or two components, background ranks
    around the number of bins, odd and too-small blocks (both sides must refuse the same inputs)."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        b1, b2 = (int(rng.choice([8, 10, 10, 11, 12, 12, 14, 16, 20, 24])) for _ in range(2))
        d1 = int(rng.choice([b1, b1 + 1, b1 + b1 // 2, 2 * b1 - 1, 2 * b1, 2 * b1 + 3]))
        d2 = int(rng.choice([b2, b2 + 1, b2 + b2 // 2, 2 * b2 - 1, 2 * b2, 2 * b2 + 3]))
        taf = int(rng.choice([1, 2, 3, 4, 7, 10]))
        T = int(rng.integers(40, 320))
        frames = int(rng.choice([T, T, T, T + 5, max(8, T // 2), max(8, T - 1), max(8, T - 1), taf - 1 if taf > 1 else 1]))
        kw = dict(max_components=int(rng.choice([1, 2, 3, 5, 20])), background_rank=int(rng.choice([0, 1, 2, 8, 30])),
                  temporal_avg_factor=taf, spatial_avg_factor=int(rng.choice([1, 2, 3, 5])), order=str(rng.choice(["F", "C"])),
                  compute_normalizer=bool(rng.random() < 0.7), max_consecutive_failures=int(rng.choice([1, 2])))
        if rng.random() < 0.2:
            kw["window_chunks"] = int(rng.choice([taf * 5, taf * 13, frames + 1, max(1, frames // 2)]))
        out.append((case, T, d1, d2, b1, b2, frames, kw, {"noise": 1.0, "dtype": "float32"}))
    return out


def draw_medium_cases(n_cases, seed):
    """Fewer, larger draws (FOV around 100-160 pixels a side, 1000-2500 frames): beyond the thresholds of the small families - the
    library's own eigensolver (orders above 512), the blocked Cholesky, split-K products, batches of tiles."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        b1, b2 = (int(2 * rng.integers(8, 17)) for _ in range(2))
        d1 = int(rng.integers(3 * b1, 5 * b1))
        d2 = int(rng.integers(3 * b2, 5 * b2))
        taf = int(rng.choice([2, 5, 10]))
        T = int(rng.integers(1000, 2500))
        frames = T if rng.random() < 0.6 else int(rng.integers(700, T))
        kw = dict(max_components=int(rng.integers(8, 21)), background_rank=int(rng.integers(0, 9)), temporal_avg_factor=taf,
                  spatial_avg_factor=int(rng.choice([1, 2])), order=str(rng.choice(["F", "C"])),
                  compute_normalizer=True, max_consecutive_failures=int(rng.choice([1, 1, 2])))
        out.append((case, T, d1, d2, b1, b2, frames, kw, {"noise": float(rng.choice([0.5, 1.0])), "dtype": "float32"}))
    return out


def draw_tall_cases(n_cases, seed):
    """Config-2-like draws: around a thousand small tiles and 600-1500 frames, R several times the number of frames (the
    row-sharded Cholesky route with the library's own eigensolver at orders 600-1500)."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        b1, b2 = (int(2 * rng.integers(5, 8)) for _ in range(2))
        d1 = int(rng.integers(150, 220))
        d2 = int(rng.integers(150, 220))
        taf = int(rng.choice([2, 5, 10]))
        T = int(rng.integers(600, 1500))
        frames = T if rng.random() < 0.7 else int(rng.integers(500, T))
        kw = dict(max_components=int(rng.integers(6, 11)), background_rank=int(rng.integers(0, 6)), temporal_avg_factor=taf,
                  spatial_avg_factor=int(rng.choice([1, 2])), order=str(rng.choice(["F", "C"])),
                  compute_normalizer=True, max_consecutive_failures=1)
        out.append((case, T, d1, d2, b1, b2, frames, kw, {"noise": 1.0, "dtype": "float32"}))
    return out


def passing_span_singular_values(res, passed, n_tile_cols, mov, mean_img, std_img, order):
    """Singular values (float64) of the standardised movie projected on span(U[:, passing tile columns + background])."""
    T = mov.shape[0]
    u = res.u.tocsc()
    keep = np.ones(u.shape[1], bool)
    keep[:n_tile_cols] = passed
    # An all-zero column (the reference's placeholder for background_rank = 0) must not enter the QR: its Householder
    # step is the identity and its column of Q a vector of the orthogonal complement that depends on the signs of
    # rounding-level pivots - on sparse columns those differ between two runs that agree to 1e-7.
    keep &= np.asarray(np.abs(u).sum(axis=0)).reshape(-1) > 0
    if not keep.any():
        return np.zeros(0), 0
    q, _ = np.linalg.qr(np.asarray(u[:, keep].todense(), dtype=np.float64))
    y = ((mov.astype(np.float64) - mean_img[None]) / std_img[None]).reshape(T, -1, order=order).T
    return np.linalg.svd(q.T @ y, compute_uv=False), int(keep.sum())


def probe_fit(res, mov, mean_img, std_img, order, n_probes=600):
    """Mean squared residual of the reconstruction against the standardised data on fixed random probes."""
    T, d1, d2 = mov.shape
    rng_ = np.random.default_rng(0)
    pi, pt = rng_.integers(0, d1 * d2, n_probes), rng_.integers(0, T, n_probes)
    y = ((mov - mean_img[None]) / std_img[None]).reshape(T, -1, order=order)[pt, pi]
    rec = np.einsum("pk,k,kp->p", np.asarray(res.u.tocsr()[pi] @ res.r, dtype=np.float64), np.asarray(res.s, dtype=np.float64),
                    np.asarray(res.v[:, pt], dtype=np.float64))
    return float(np.mean((rec - y) ** 2))


def run_case(ctx, case, T, d1, d2, b1, b2, frames, kw, extra=None, out=None):
    """Returns a dict of the measured figures of one draw (and reports them through `out`)."""
    say = out or (lambda s: None)
    if extra:
        from localmd_amd.synthetic import make_movie

        mov = make_movie(T, d1, d2, seed=1000 + case, noise=extra["noise"])
        if extra["dtype"] == "uint16":
            mov = np.clip(np.round(mov * 20.0), 0, 65535).astype(np.uint16)
        else:
            mov = mov.astype(extra["dtype"])
    else:
        mov = tp._movie(T, d1, d2, seed=1000 + case)
    # injected thresholds between the statistics of signal components and of noise (as in scripts/fuzz_parity.py)
    pmd, diag, ref = tp._compare_full(ctx, mov, (b1, b2), frames, thresholds=(1.0, 1.7), **kw)
    shown = {k: (f"array{v.shape}" if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    say(f"case {case}: T={T} fov={d1}x{d2} block={b1}x{b2} frames={frames} {shown} {extra or ''}")
    fig = {"pmd": pmd, "diag": diag, "ref": ref}
    dr = diag["tile_ranks"].astype(int) - ref.diag["tile_ranks"].astype(int)
    thr = diag["thresholds"]
    knife = np.zeros(len(dr), dtype=bool)
    for t, dl in enumerate(ref.diag["tile_diag"]):
        for w in dl:
            mrg = np.minimum(np.abs(w["spatial"] - thr[0]) / thr[0], np.abs(w["temporal"] - thr[1]) / thr[1])
            knife[t] |= bool(np.any(mrg < 1e-2))
    fig["rank_diff_tiles"] = np.nonzero(dr)[0]
    fig["knife"] = knife
    say(f"   tiles {len(dr)}, tile ranks differ in {int((dr != 0).sum())} tiles (knife-edge tiles: {int(knife.sum())}); components {len(pmd.s)} vs {len(ref.s)}")
    n = min(len(pmd.s), len(ref.s))
    k = max(1, n // 4)
    rel = np.abs(pmd.s[:n] - ref.s[:n]) / ref.s[:n]
    bad = np.nonzero(rel[:k] > 2e-3)[0]
    fig["s_bad"] = bad
    say(f"   top-quarter ({k}) singular values off by > 2e-3: indices {bad.tolist()}, values {np.round(ref.s[bad], 2).tolist()} "
        f"(s1 = {ref.s[0]:.1f}), rel {np.round(rel[bad], 4).tolist()}")
    if np.all(dr == 0):
        fig.update(_arbiter_distances(ctx, mov, (b1, b2), frames, kw, pmd, diag, ref, say))
    if np.all(dr == 0):
        hp, hg = PM.hip_cols(diag)
        op, og = PM.oracle_cols(ref)
        ntc = diag["n_tile_cols"]
        both = hp & op
        fig["decisions_equal"] = bool(np.array_equal(hp, op))
        say(f"   tile columns {ntc}: passing on both sides {int(both.sum())}, kept failing {int((~hp).sum())} (HIP) / {int((~op).sum())} (oracle), "
            f"decisions equal: {fig['decisions_equal']}")
        mean64, std64 = ref.mean_img.astype(np.float64), ref.std_img.astype(np.float64)
        sa, na = passing_span_singular_values(pmd, both, ntc, mov, mean64, std64, kw.get("order", "F"))
        sb, _ = passing_span_singular_values(ref, both, ntc, mov, mean64, std64, kw.get("order", "F"))
        relp = np.abs(sa - sb) / sb
        kq = max(1, len(sa) // 4)
        fig["span_s_rel_all"], fig["span_s_rel_top"] = float(relp.max(initial=0.0)), float(relp[:kq].max(initial=0.0))
        say(f"   projected on the {na} passing columns (+ background): s rel diff max {fig['span_s_rel_all']:.2e} over all {len(sa)}, top-quarter {fig['span_s_rel_top']:.2e}")
        if len(bad):
            # where the offending values sit: next to the smallest singular values of the passing span = noise level
            say(f"   offending values {np.round(ref.s[bad], 1).tolist()} vs the passing span's smallest value {sb[-1]:.1f} and median {np.median(sb):.1f}")
        m = PM.measure(pmd, ref, (hp, hg), (op, og), ntc)
        fig["measure"] = m
        say(f"   U_data |diff| stable columns {m['u_data_err_stable']:.2e} ({m['n_stable_cols'][0]} of {m['n_stable_cols'][1]}), all columns "
            f"{m['u_data_err_all']:.2e} (max |U_data| {m['u_data_max_abs']:.3f})")
    else:
        for t in np.nonzero(dr)[0][:6]:
            say(f"   tile {t}: ranks {diag['tile_ranks'][t]} vs {ref.diag['tile_ranks'][t]}, knife-edge: {bool(knife[t])}")
    # fit to the data on random probes (the quantity a user cares about)
    order = kw.get("order", "F")
    fig["fit"] = (probe_fit(pmd, mov, ref.mean_img, ref.std_img, order), probe_fit(ref, mov, ref.mean_img, ref.std_img, order))
    say(f"   mean squared residual on 600 probes: {fig['fit'][0]:.4f} (HIP) / {fig['fit'][1]:.4f} (oracle)")
    if "fits" in fig:
        say("   the same for the other arithmetic forms: " + ", ".join(f"{k} {v:.4f}" for k, v in fig["fits"].items()))
    return fig


def _arbiter_distances(ctx, mov, block, frames, kw, pmd, diag, ref, say):
    """Distances of the three fp32 results (HIP, oracle with double-precision LAPACK, oracle with single-precision LAPACK)
    to the float64 arbiter: singular values on the span of the passing components, U_data on the stable columns."""
    okw = {k: v for k, v in kw.items()}
    seed = 123   # tp._compare_full's device seed

    def run(lapack="double", fp64=False):
        O.LAPACK_PRECISION = lapack
        np.random.seed(7)
        try:
            if fp64:
                with O.arbiter_precision():
                    return O.localmd_decomposition(mov, block, frames, rng=DeviceSource(ctx, seed), thresholds=diag["thresholds"], dtype="float64", **okw)
            return O.localmd_decomposition(mov, block, frames, rng=DeviceSource(ctx, seed), thresholds=diag["thresholds"], **okw)
        finally:
            O.LAPACK_PRECISION = "double"

    arb, ref1 = run(fp64=True), run(lapack="single")
    out = {}
    # Conditioning of the matrix the orthogonalisation diagonalises when R > frames (decomposition.py:974-999):
    # C = right^T (U^T U) right.  Eigenvalues below ~eps32 lambda_max are not resolvable by ANY fp32 eigensolver, and the
    # reference scales those directions by 1 / sqrt(|lambda|).
    ua = arb.u.tocsr().astype(np.float64)
    right = np.asarray(arb.diag["v_cropped"], dtype=np.float64)
    cond = None
    if ua.shape[1] > right.shape[1]:
        lam_c = np.linalg.eigvalsh(right.T @ np.asarray((ua.T @ ua) @ right))
        cond = float(lam_c[0] / lam_c[-1])
        say(f"   R > frames: lambda_min / lambda_max of right^T U^T U right = {cond:.2e} (eps32 = {np.finfo(np.float32).eps:.2e})")
    fits = {"arbiter fp64": probe_fit(arb, mov, ref.mean_img, ref.std_img, kw.get("order", "F")),
            "oracle fp32 single-LAPACK": probe_fit(ref1, mov, ref.mean_img, ref.std_img, kw.get("order", "F"))}
    sides = {"HIP": (pmd, diag["tile_ranks"], PM.hip_cols(diag)), "oracle fp32": (ref, ref.diag["tile_ranks"], PM.oracle_cols(ref)),
             "oracle fp32 single-LAPACK": (ref1, ref1.diag["tile_ranks"], PM.oracle_cols(ref1))}
    ac = PM.oracle_cols(arb)
    ntc = diag["n_tile_cols"]
    order = kw.get("order", "F")
    mean64, std64 = arb.mean_img.astype(np.float64), arb.std_img.astype(np.float64)
    for name, (res, ranks, cols) in sides.items():
        if not np.array_equal(ranks, arb.diag["tile_ranks"]):
            say(f"   {name} vs arbiter fp64: tile ranks differ (knife-edge decision), not compared")
            continue
        both = cols[0] & ac[0]
        sa, _ = passing_span_singular_values(res, both, ntc, mov, mean64, std64, order)
        sb, _ = passing_span_singular_values(arb, both, ntc, mov, mean64, std64, order)
        relp = np.abs(sa - sb) / sb
        m = PM.measure(res, arb, cols, ac, ntc)
        out[name] = {"span_top": float(relp[:max(1, len(sa) // 4)].max(initial=0.0)), "span_all": float(relp.max(initial=0.0)), "u_stable": m["u_data_err_stable"],
                     "s_signal": float(m["s_rel"][m["signal"]].max(initial=0.0)), "vt_signal": float(m["vt_row_err"][m["signal"]].max(initial=0.0))}
        say(f"   {name} vs arbiter fp64: passing-span s top-quarter {out[name]['span_top']:.2e} / all {out[name]['span_all']:.2e}, "
            f"U_data stable {m['u_data_err_stable']:.2e}, final s signal {out[name]['s_signal']:.2e}, Vt signal {out[name]['vt_signal']:.2e}")
    return {"arbiter": out, "fits": fits, "gram_cond": cond}


_CASES = {c[0]: c for c in draw_cases(max(FUZZ_CASES) + 1)}
# the second family (many tiles, R > frames): a well-conditioned draw, three with Gram eigenvalues at the rounding level
# (5: the constant null direction + a failed Cholesky step - the fallback that once blew up; 20: where the reference's
# own arithmetic ends at 12.6 / 55; 21: resolvable except for one direction)
WIDE_SEED = 11
WIDE_CASES = [4, 5, 20, 21]
_WIDE = {c[0]: c for c in draw_wide_cases(max(WIDE_CASES) + 1, WIDE_SEED)}
# the third family (the rest of the option space; 3 seeds x 48 draws swept with scripts/fuzz_sweep.py).  Seed 22: draws 3 and 10 had Vt errors of 7e-4 / 1e-3 on signal components
# while the global stage still ran its small eigenproblems in fp32 divide and conquer; draws 13 and 45 (30 x 40 and 36 x 40
# pixel tiles, spatial_avg_factor = 1, 58 / 57 sketch columns) need the CholeskyQR2 form of the sketch basis.  Seed 21:
# rank_prune + float64 input at max_components = 47; rank_prune + pixel_weighting on uint16 input.
# Seed 23, draw 11: max_components = 30 with 27 time bins per window (the reference's slicing returns 27 components).
OPTION_CASES = [(22, 3), (22, 10), (22, 13), (22, 45), (21, 7), (21, 16), (23, 11)]
_OPTION = {(sd, c[0]): c for sd in (21, 22, 23) for c in draw_option_cases(48, sd)}


# the generic-width family (round 3; 28 draws swept over two seeds).  Seed 1: draw 0 = max_components 108 with residual windows
# and background_rank 90; draw 1 = every component of 30 x 28-pixel tiles kept (max_consecutive_failures 100, 88 components,
# two blocks of 64 rows per tile in the global stage), R > frames in the unresolvable regime - the draw that showed the
# double-precision eigenvalue paths amplifying null directions beyond anything the reference's fp32 arithmetic does;
# draw 4 = rank_prune with 84 components and 58 background columns, C order; draw 10 = pixel weights + windows at 109.
WIDECOMP_CASES = [(1, 0), (1, 1), (1, 4), (1, 10)]
_WIDECOMP = {(1, c[0]): c for c in draw_widecomp_cases(16, 1)}


def check_case(fig):
    """The assertions of one draw (shared with scripts/fuzz_sweep.py)."""
    pmd, ref = fig["pmd"], fig["ref"]
    assert np.all(np.isfinite(pmd.s)) and np.all(np.isfinite(pmd.v)) and np.all(np.isfinite(pmd.u.data)) and np.all(np.isfinite(pmd.r))
    np.testing.assert_allclose(pmd.mean_img, ref.mean_img, rtol=1e-5)
    np.testing.assert_allclose(pmd.var_img, ref.std_img, rtol=2e-4)
    # tile ranks may differ only where a decision statistic sits within 1 % of its threshold
    assert set(fig["rank_diff_tiles"].tolist()) <= set(np.nonzero(fig["knife"])[0].tolist()), fig["rank_diff_tiles"]
    # fit to the data: as good as the oracle's, or - where fp32 loses the small half of the spectrum in EVERY implementation
    # (R > frames with a few hundred frames) - not farther from the float64 result than the
    # reference's own fp32 arithmetic
    e1, e0 = fig["fit"]
    unresolvable = fig.get("gram_cond") is not None and fig["gram_cond"] < 16 * np.finfo(np.float32).eps
    # the same regime reached through rank_prune (the Gram matrix of the randomly mixed traces): recognised by the
    # reference's own arithmetic - in BOTH LAPACK precisions - missing the float64 signal singular values by more than 1 %
    arb_ = fig.get("arbiter", {})
    if "oracle fp32" in arb_ and "oracle fp32 single-LAPACK" in arb_:
        unresolvable = unresolvable or min(arb_["oracle fp32"]["s_signal"], arb_["oracle fp32 single-LAPACK"]["s_signal"]) > 1e-2
    if abs(e1 - e0) >= 0.05 * e0 + 1e-6:
        fits = fig.get("fits")
        assert fits is not None, fig["fit"]
        ea, es = fits["arbiter fp64"], fits["oracle fp32 single-LAPACK"]
        if unresolvable:
            # no fp32 form is reliable here (profiles/r02_fuzz_wide.txt: the reference's arithmetic ends between 0.9 and 55
            # where float64 gives 0.9); what is asserted is the absence of a blow-up (a fallback bug once gave 1400)
            assert e1 <= 5.0 * ea, (fig["fit"], fits)
        else:
            assert abs(e1 - ea) <= 3.0 * max(abs(e0 - ea), abs(es - ea)) + 0.05 * ea, (fig["fit"], fits)
    if len(fig["rank_diff_tiles"]) == 0:
        assert pmd.r.shape == ref.r.shape and pmd.s.shape == ref.s.shape and pmd.v.shape == ref.v.shape
        assert fig["measure"]["csr_equal"]
        # distance to the float64 arbiter: HIP within a small multiple of the reference's own fp32 arithmetic
        arb = fig["arbiter"]
        # R > frames with eigenvalues of the orthogonalisation's Gram matrix at the fp32 rounding level: the global stage is
        # not determined by fp32 arithmetic (every fp32 eigensolver returns different vectors for that part of the spectrum,
        # the 1 / sqrt(|lambda|) scaling amplifies them: DESIGN section 2, profiles/r02_fuzz_wide.txt); the tile stage (U) and
        # the fit to the data are still compared
        keys = (("span_top", 2e-4), ("span_all", 5e-4), ("u_stable", 2e-5)) + (() if unresolvable else (("s_signal", 1e-4), ("vt_signal", 3e-4)))
        if "HIP" in arb and len(arb) == 3:
            for key, floor in keys:
                worst_ref = max(arb["oracle fp32"][key], arb["oracle fp32 single-LAPACK"][key])
                assert arb["HIP"][key] <= 3.0 * worst_ref + floor, (key, arb["HIP"][key], arb["oracle fp32"][key], arb["oracle fp32 single-LAPACK"][key])


@pytest.mark.parametrize("case", FUZZ_CASES)
def test_fuzz_case(gpu_ctx, case):
    lines = []
    fig = run_case(gpu_ctx, *_CASES[case], out=lines.append)
    print("\n".join(lines))
    check_case(fig)


@pytest.mark.parametrize("case", WIDE_CASES)
def test_fuzz_wide_case(gpu_ctx, case):
    lines = []
    fig = run_case(gpu_ctx, *_WIDE[case], out=lines.append)
    print("\n".join(lines))
    check_case(fig)


@pytest.mark.parametrize("seed,case", OPTION_CASES)
def test_fuzz_option_case(gpu_ctx, seed, case):
    lines = []
    fig = run_case(gpu_ctx, *_OPTION[(seed, case)], out=lines.append)
    print("\n".join(lines))
    check_case(fig)


@pytest.mark.parametrize("seed,case", WIDECOMP_CASES)
def test_fuzz_widecomp_case(gpu_ctx, seed, case):
    lines = []
    fig = run_case(gpu_ctx, *_WIDECOMP[(seed, case)], out=lines.append)
    print("\n".join(lines))
    check_case(fig)
