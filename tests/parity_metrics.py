"""
Measured parity figures between two results of the decomposition on the same inputs (test infrastructure).

``a`` and ``b`` are PMDArray-like objects (``.u`` CSR, ``.r``, ``.s``, ``.v``); ``b`` is the referee (the CPU oracle in
fp32, or its float64 arbiter form).  Every figure is a plain number so the callers can assert on it or print a table.

What is well posed and what is not: every tile keeps the first component(s) that FAIL the roughness
tests (evaluation.py:195-222) - vectors of the tile's noise subspace, whose neighbouring singular values differ by a
few per cent, so two correct fp32 implementations return different vectors of that subspace.  Hence the classes:
  * stable tile columns: passed both roughness tests on both sides and have a relative singular-value gap > GAP to
    both neighbours inside their tile -> compared element-wise (U_data, rows of R);
  * the other tile columns -> compared through what they span together with the stable ones (U R, reconstruction);
  * separated final components: relative gap of s to both neighbours > GAP on both sides -> Vt rows, R columns and
    (U R) columns compared after sign alignment.
"""
import numpy as np

GAP = 2e-2
# A component with (s / s_1)^2 below the fp32 unit roundoff cannot be resolved by an fp32 Gram-matrix SVD, which is what
# the reference computes (decomposition.py:1089-1099): its eigenvalue s^2 is smaller than the rounding error of the
# largest one.  Such components (e.g. the numerically null direction the reference keeps through |lambda|,
# decomposition.py:984-988) are arbitrary on every fp32 side and excluded from the comparisons.
RESOLVABLE = float(np.sqrt(np.finfo(np.float32).eps))


def rel_gaps(sv):
    sv = np.asarray(sv, dtype=np.float64)
    return np.minimum(np.abs(np.diff(sv, prepend=np.inf)), np.abs(np.diff(sv, append=0))) / np.maximum(sv, 1e-300)


def tile_column_table(tile_ranks, tile_good, v_cropped, max_rank=None):
    """Per tile column of U: (tile, index in tile, passed, sigma = norm of its trace, relative gap inside the tile).
    The last column of a tile that was cut at max_rank has an unknown gap to the next (discarded) one: gap 0."""
    tile_ranks = np.asarray(tile_ranks, dtype=np.int64)
    n = int(tile_ranks.sum())
    tile = np.repeat(np.arange(len(tile_ranks)), tile_ranks)
    off = np.concatenate([[0], np.cumsum(tile_ranks)])
    idx = np.arange(n) - off[tile]
    sigma = np.linalg.norm(np.asarray(v_cropped[:n], dtype=np.float64), axis=1)
    passed = np.zeros(n, dtype=bool)
    gap = np.zeros(n)
    for t, rk in enumerate(tile_ranks):
        if rk == 0:
            continue
        sl = slice(off[t], off[t] + rk)
        passed[sl] = np.asarray(tile_good[t]).reshape(-1)[:rk] > 0
        gap[sl] = rel_gaps(sigma[sl])
        if max_rank is not None and rk >= max_rank:
            gap[off[t] + rk - 1] = 0.0
    return tile, idx, passed, sigma, gap


def oracle_cols(res):
    """(passed, gap) per tile column of an oracle result (its diagnostics: per-window decisions, v_cropped)."""
    good = [d[0]["good"] if len(d) == 1 else np.concatenate([w["good"][w["kept"]] for w in d]) for d in res.diag["tile_diag"]]
    good = [np.concatenate([g, np.zeros(64, bool)]) for g in good]
    t = tile_column_table(res.diag["tile_ranks"], good, res.diag["v_cropped"], res.diag.get("max_components"))
    return t[2], t[4]


def hip_cols(diag):
    """(passed, gap) per tile column of a HIP result (return_diagnostics=True)."""
    ranks = np.asarray(diag["tile_ranks"], dtype=np.int64)
    n = int(ranks.sum())
    off = np.concatenate([[0], np.cumsum(ranks)])
    passed, gap = np.zeros(n, bool), np.zeros(n)
    sig = np.asarray(diag["col_sigma"][:n], dtype=np.float64)
    for t, rk in enumerate(ranks):
        sl = slice(off[t], off[t] + rk)
        kept = np.nonzero(diag["tile_keep"][t] > 0)[0][:rk]   # first window; columns of later windows count as unstable
        passed[off[t]:off[t] + len(kept)] = diag["tile_good"][t][kept] > 0
        gap[sl] = rel_gaps(sig[sl])
        if rk >= diag["max_components"] and rk > 0:
            gap[off[t] + rk - 1] = 0.0
    return passed, gap


def measure(a, b, a_cols=None, b_cols=None, n_tile_cols=None):
    """a_cols / b_cols: (passed, gap) per tile column of each side (tile_column_table); n_tile_cols: number of tile
    columns (the background columns behind them are always 'stable').  Returns a dict of measured figures."""
    out = {}
    out["shape_equal"] = (a.u.shape == b.u.shape and a.r.shape == b.r.shape and a.s.shape == b.s.shape and a.v.shape == b.v.shape)
    au, bu = a.u.tocsr(), b.u.tocsr()
    au.sort_indices(); bu.sort_indices()
    out["csr_equal"] = bool(au.shape == bu.shape and np.array_equal(au.indptr, bu.indptr) and np.array_equal(au.indices, bu.indices))
    n = min(len(a.s), len(b.s))
    out["n_components"] = (len(a.s), len(b.s))
    s_a, s_b = np.asarray(a.s[:n], np.float64), np.asarray(b.s[:n], np.float64)
    out["s_rel"] = np.abs(s_a - s_b) / s_b
    gaps = np.minimum(rel_gaps(b.s)[:n], rel_gaps(a.s)[:n])
    valid = (s_b > RESOLVABLE * s_b[0]) & (s_a > RESOLVABLE * s_a[0])
    sep = (gaps > GAP) & valid
    if len(a.s) != len(b.s):
        sep[n - 1] = False
    signal = sep & (s_b > 5e-2 * s_b[0])
    out["gaps"], out["sep"], out["valid"], out["signal"] = gaps, sep, valid, signal
    # Vt rows, sign aligned
    va, vb = np.asarray(a.v[:n], np.float64), np.asarray(b.v[:n], np.float64)
    sgn = np.where(np.sum(va * vb, axis=1) < 0, -1.0, 1.0)
    out["vt_row_err"] = np.linalg.norm(va * sgn[:, None] - vb, axis=1) / np.linalg.norm(vb, axis=1)
    # U R columns (orthonormal spatial components), same signs as the Vt rows
    ur_a = np.asarray(a.u @ a.r[:, :n], np.float64) * sgn[None, :]
    ur_b = np.asarray(b.u @ b.r[:, :n], np.float64)
    out["ur_col_err"] = np.linalg.norm(ur_a - ur_b, axis=0) / np.linalg.norm(ur_b, axis=0)
    # Orthonormality.  Both sides obtain Vt = W^T V / s from an fp32 eigendecomposition of V V^T, whose residual is
    # ~eps lambda_1: (Vt Vt^T - I)_cc' ~ eps s_1^2 / (s_c s_c'), and [U R] inherits the same growth from the Gram matrix
    # behind P when R > frames.  Reported: the plain deviation on the strong components (s > 5 % of s_1) and the
    # deviation weighted by s_c s_c' / s_1^2 (= the residual of the Gram eigendecomposition relative to lambda_1)
    # over all resolvable ones.
    strong = valid & (s_b > 5e-2 * s_b[0])
    out["strong"] = strong

    def orth(mat_cols, sv, mask):
        e = mat_cols[:, mask].T @ mat_cols[:, mask] - np.eye(int(mask.sum()))
        w = sv[mask] / sv[0]
        return np.abs(e), np.abs(e) * np.outer(w, w)

    ea, eaw = orth(ur_a, s_a, valid)
    eb, ebw = orth(ur_b, s_b, valid)
    st = strong[valid]
    out["orth_ur"] = (float(ea[np.ix_(st, st)].max()), float(eb[np.ix_(st, st)].max()))
    out["orth_ur_weighted"] = (float(eaw.max()), float(ebw.max()))
    ea, eaw = orth(va.T, s_a, valid)
    eb, ebw = orth(vb.T, s_b, valid)
    out["orth_vt"] = (float(ea[np.ix_(st, st)].max()), float(eb[np.ix_(st, st)].max()))
    out["orth_vt_weighted"] = (float(eaw.max()), float(ebw.max()))
    if out["csr_equal"] and a_cols is not None:
        R = au.shape[1]
        ntc = R if n_tile_cols is None else n_tile_cols
        stable = np.ones(R, dtype=bool)
        stable[:ntc] = a_cols[0] & b_cols[0] & (np.minimum(a_cols[1], b_cols[1]) > GAP)
        ac, bc = au.tocsc(), bu.tocsc()
        # column signs of U (each tile column is an SVD vector: its sign is arbitrary on both sides)
        dots = np.asarray(ac.multiply(bc).sum(axis=0)).reshape(-1)
        csgn = np.where(dots < 0, -1.0, 1.0)
        diff = np.abs(ac.data * np.repeat(csgn, np.diff(ac.indptr)) - bc.data)
        col_of = np.repeat(np.arange(R), np.diff(ac.indptr))
        out["u_data_max_abs"] = float(np.abs(bc.data).max())
        out["u_data_err_stable"] = float(diff[stable[col_of]].max(initial=0.0))
        out["u_data_err_all"] = float(diff.max(initial=0.0))
        out["n_stable_cols"] = (int(stable.sum()), R)
        # R: rows of stable columns x separated components (R = P W is unique given U R because U has full column rank)
        ra = np.asarray(a.r[:R, :n], np.float64) * csgn[:, None] * sgn[None, :]
        rb = np.asarray(b.r[:R, :n], np.float64)
        sub = np.ix_(np.nonzero(stable)[0], np.nonzero(signal)[0])
        out["r_max_abs"] = float(np.abs(rb[:, valid]).max())
        out["r_err_stable_signal"] = float(np.abs(ra[sub] - rb[sub]).max(initial=0.0))
    return out


def probes(a, b, shape, n=1000, seed=0):
    """Reconstruction U R diag(s) Vt at random (pixel, frame) probes: max |a - b| relative to the peak of b."""
    T, d1, d2 = shape
    rng = np.random.default_rng(seed)
    pi, pt = rng.integers(0, d1 * d2, n), rng.integers(0, T, n)
    ra = np.einsum("pk,k,kp->p", np.asarray((a.u.tocsr()[pi]) @ a.r), a.s, a.v[:, pt])
    rb = np.einsum("pk,k,kp->p", np.asarray((b.u.tocsr()[pi]) @ b.r), b.s, b.v[:, pt])
    return float(np.abs(ra - rb).max() / np.abs(rb).max())


def summary_lines(name, m, s_ref):
    """Human-readable lines of a measure() dict (used for the records under profiles/)."""
    n = len(m["s_rel"])
    sep, sig, valid = m["sep"], m["signal"], m["valid"]
    lines = [f"[{name}] components {m['n_components'][0]} vs {m['n_components'][1]}, shapes equal {m['shape_equal']}, CSR structure equal {m['csr_equal']}"]
    lines.append(f"[{name}] resolvable components (s > sqrt(eps32) s1): {int(valid.sum())} of {n}")
    lines.append(f"[{name}] s rel err: max over resolvable {m['s_rel'][valid].max():.2e}, top-100 {m['s_rel'][:100].max():.2e}, signal (s > 5% s1, separated; {int(sig.sum())}) "
                 f"{m['s_rel'][sig].max(initial=0):.2e}")
    for key, label in (("vt_row_err", "Vt row"), ("ur_col_err", "(U R) column")):
        e = m[key]
        lines.append(f"[{name}] {label} err: signal max {e[sig].max(initial=0):.2e}, separated ({int(sep.sum())}) median {np.median(e[sep]) if sep.any() else 0:.2e} "
                     f"max {e[sep].max(initial=0):.2e}")
    if "u_data_err_stable" in m:
        lines.append(f"[{name}] U_data |diff| (max |U_data| {m['u_data_max_abs']:.3f}): stable columns {m['n_stable_cols'][0]}/{m['n_stable_cols'][1]} "
                     f"{m['u_data_err_stable']:.2e}, all columns {m['u_data_err_all']:.2e}")
        lines.append(f"[{name}] R |diff| on stable rows x signal columns (max |R| {m['r_max_abs']:.3f}): {m['r_err_stable_signal']:.2e}")
    lines.append(f"[{name}] strong components ({int(m['strong'].sum())}): |(UR)^T(UR) - I| {m['orth_ur'][0]:.2e} (referee {m['orth_ur'][1]:.2e}); |Vt Vt^T - I| {m['orth_vt'][0]:.2e} (referee {m['orth_vt'][1]:.2e})")
    lines.append(f"[{name}] resolvable components, weighted by s_c s_c' / s_1^2: |(UR)^T(UR) - I| {m['orth_ur_weighted'][0]:.2e} (referee {m['orth_ur_weighted'][1]:.2e}); "
                 f"|Vt Vt^T - I| {m['orth_vt_weighted'][0]:.2e} (referee {m['orth_vt_weighted'][1]:.2e})")
    return lines


# ------------------------------------------------------------------------------------------------------------
# BASELINE configurations: HIP against the fp32 oracle, and every fp32 result against the float64 arbiter
# ------------------------------------------------------------------------------------------------------------
CONFIGS = {
    # BASELINE.json configs[0] (demoMovie stand-in: the real file is absent), configs[1], and the headline regime of
    # configs[2] (T = 10^4, r = 50, R > frames, order-9999 eigenproblem) on a 256 x 256 sub-field of view
    "config1": dict(T=2000, d1=60, d2=80, block=(20, 20), frames=100, kw=dict(), sim_iters=250),
    "config2": dict(T=2000, d1=256, d2=256, block=(20, 20), frames=2000, kw=dict(max_components=8), sim_iters=50),
    "headline": dict(T=10000, d1=256, d2=256, block=(20, 20), frames=10000, kw=dict(), sim_iters=50),
}


def run_config(ctx, name, arbiter=False, single=False, out=print, movie_seed=0, fp32_oracle=True):
    """Runs the HIP path and the oracle(s) on one configuration; returns {"hip": (pmd, diag), "results": {label: oracle
    result}, "measures": {"A vs B": measure dict}, "probes": {...}}.  `out` receives the report lines."""
    import time
    import localmd_amd
    from localmd_amd import decomposition as Dm
    from localmd_amd.synthetic import make_movie
    from oracle import pmd_oracle as O
    from tests.util import DeviceSource

    if name in CONFIGS:
        c = CONFIGS[name]
    else:  # "<T>x<d1>x<d2>xb<block>xr<max_components>"
        T, d1, d2, b, r = name.split("x")
        c = dict(T=int(T), d1=int(d1), d2=int(d2), block=(int(b[1:]),) * 2, frames=int(T), kw=dict(max_components=int(r[1:])), sim_iters=50)
    Dm.QUIET = True
    mov = make_movie(c["T"], c["d1"], c["d2"], seed=movie_seed)
    t0 = time.perf_counter()
    np.random.seed(7)
    pmd, diag = localmd_amd.localmd_decomposition(mov, c["block"], c["frames"], seed=123, sim_iters=c["sim_iters"],
                                                  return_diagnostics=True, ctx=ctx, **c["kw"])
    out(f"# {name}: {c['T']} frames of {c['d1']} x {c['d2']}, block {c['block']}, frames_to_init {c['frames']}, {c['kw']}")
    out(f"HIP: {time.perf_counter() - t0:.1f} s (incl. diagnostics), tiles {len(diag['tile_ranks'])}, rank before {diag['rank_before']} -> after "
        f"{diag['rank_after']}, orthogonalizer {diag['orthogonalizer']}, thresholds {diag['thresholds']}")
    results, measures, prb = {}, {}, {}

    def oracle(label, precision=None, lapack="double"):
        import sys
        import threading

        t1 = time.perf_counter()
        O.LAPACK_PRECISION = lapack
        np.random.seed(7)
        # a long oracle run prints nothing for minutes: a heartbeat on the real stdout keeps job watchdogs informed
        stop = threading.Event()

        def beat():
            while not stop.wait(60.0):
                sys.__stdout__.write(f"[{label}: {time.perf_counter() - t1:.0f} s ...]\n")
                sys.__stdout__.flush()

        hb = threading.Thread(target=beat, daemon=True)
        hb.start()
        try:
            if precision == "fp64":
                with O.arbiter_precision():
                    res = O.localmd_decomposition(mov, c["block"], c["frames"], rng=DeviceSource(ctx, 123), thresholds=diag["thresholds"],
                                                  dtype="float64", **c["kw"])
            else:
                res = O.localmd_decomposition(mov, c["block"], c["frames"], rng=DeviceSource(ctx, 123), thresholds=diag["thresholds"], **c["kw"])
        finally:
            O.LAPACK_PRECISION = "double"
            stop.set()
        out(f"{label}: {time.perf_counter() - t1:.1f} s, rank before {res.diag['rank_before']} -> after {len(res.s)}")
        results[label] = res
        return res

    def compare(la, a, a_ranks, a_cols, lb, b):
        same = np.array_equal(a_ranks, b.diag["tile_ranks"])
        m = measure(a, b, a_cols, oracle_cols(b), diag["n_tile_cols"]) if same else measure(a, b)
        key = f"{la} vs {lb}"
        measures[key] = m
        prb[key] = probes(a, b, mov.shape)
        for ln in summary_lines(key, m, b.s):
            out(ln)
        out(f"[{key}] reconstruction probes: {prb[key]:.2e} of the peak")

    hc = hip_cols(diag)
    mism = np.zeros(0, dtype=np.int64)
    if fp32_oracle:
        ref = oracle("oracle fp32")
        mism = np.nonzero(diag["tile_ranks"] != ref.diag["tile_ranks"])[0]
        out(f"tile ranks differing (HIP vs oracle fp32): {len(mism)} of {len(diag['tile_ranks'])} {list(mism[:10])}")
        out(f"mean_img rel {np.abs(pmd.mean_img / ref.mean_img - 1).max():.2e}, std_img rel {np.abs(pmd.var_img / ref.std_img - 1).max():.2e}")
        compare("HIP", pmd, diag["tile_ranks"], hc, "oracle fp32", ref)
    if single:
        ref1 = oracle("oracle fp32 single-LAPACK", lapack="single")
        compare("oracle fp32 single-LAPACK", ref1, ref1.diag["tile_ranks"], oracle_cols(ref1), "oracle fp32", ref)
    if arbiter:
        arb = oracle("arbiter fp64", precision="fp64")
        compare("HIP", pmd, diag["tile_ranks"], hc, "arbiter fp64", arb)
        for k, v in list(results.items()):
            if k != "arbiter fp64":
                compare(k, v, v.diag["tile_ranks"], oracle_cols(v), "arbiter fp64", arb)
    return {"hip": (pmd, diag), "results": results, "measures": measures, "probes": prb, "movie": mov, "rank_mismatch": mism}


# ------------------------------------------------------------------------------------------------------------
# Committed referee fixtures (tests/golden/make_parity_fixtures.py): the oracle runs ONCE in the build container with
# the host Philox source; the GPU box compares the HIP path with the stored vectors without running the oracle.
# The device generator restates the same counter-based streams (rng.hip; equal up to the rounding of logf / sincosf),
# so the same seed gives the same Gaussian test matrices to ~1 ulp.
# ------------------------------------------------------------------------------------------------------------
N_PIX_SAMPLE = 2048
N_STABLE_COLS = 64
N_PROBES = 2000


def fixture_from_result(res, shape, seed=0):
    """Arrays of one referee result (OracleResult) that the fixture keeps: everything small in full, the large factors on
    the compared components / a fixed sample of pixels, columns and probes."""
    import hashlib

    T, d1, d2 = shape
    rng = np.random.default_rng(seed)
    u = res.u.tocsr()
    u.sort_indices()
    s = np.asarray(res.s, np.float64)
    gaps = rel_gaps(s)
    sig = np.nonzero((gaps > GAP) & (s > 5e-2 * s[0]))[0]
    pix = np.sort(rng.choice(d1 * d2, size=min(N_PIX_SAMPLE, d1 * d2), replace=False))
    ur = np.asarray(u[pix] @ np.asarray(res.r[:, sig], np.float64))
    passed, gap = oracle_cols(res)
    ntc = len(passed)
    stable = np.nonzero(passed & (gap > GAP))[0]
    cols = np.sort(rng.choice(stable, size=min(N_STABLE_COLS, len(stable)), replace=False)) if len(stable) else stable
    uc = u.tocsc()
    col_data = [np.asarray(uc.data[uc.indptr[c]:uc.indptr[c + 1]], np.float64) for c in cols]
    col_rows = [np.asarray(uc.indices[uc.indptr[c]:uc.indptr[c + 1]], np.int32) for c in cols]
    pi, pt = rng.integers(0, d1 * d2, N_PROBES), rng.integers(0, T, N_PROBES)
    rec = np.einsum("pk,k,kp->p", np.asarray(u[pi] @ res.r), res.s, res.v[:, pt])
    return {
        "tile_ranks": np.asarray(res.diag["tile_ranks"], np.int32), "thresholds": np.asarray(res.diag["thresholds"], np.float64),
        "rank_before": np.int64(res.diag["rank_before"]), "max_components": np.int64(res.diag["max_components"]),
        "U_shape": np.asarray(u.shape, np.int64), "U_indptr": np.asarray(u.indptr, np.int32), "U_nnz": np.int64(u.nnz),
        "U_indices_sha256": np.frombuffer(hashlib.sha256(np.ascontiguousarray(u.indices, np.int32).tobytes()).digest(), np.uint8),
        "s": np.asarray(res.s, np.float32), "signal": sig.astype(np.int32), "Vt_signal": np.asarray(res.v[sig], np.float32),
        "pix_sample": pix.astype(np.int32), "UR_signal_sample": ur.astype(np.float32),
        "col_passed": passed, "col_gap": gap.astype(np.float32), "stable_cols": cols.astype(np.int32),
        "stable_col_rows": np.concatenate(col_rows) if col_rows else np.zeros(0, np.int32),
        "stable_col_data": np.concatenate(col_data) if col_data else np.zeros(0),
        "stable_col_ptr": np.concatenate([[0], np.cumsum([len(c) for c in col_data])]).astype(np.int64),
        "R_stable_signal": np.asarray(res.r[np.ix_(cols, sig)], np.float32), "R_max_abs": np.float32(np.abs(res.r[:ntc]).max()),
        "probe_pix": pi.astype(np.int32), "probe_frame": pt.astype(np.int32), "probe_rec": rec.astype(np.float32),
        "mean_sample": np.asarray(res.mean_img, np.float32).reshape(-1)[pix], "std_sample": np.asarray(res.std_img, np.float32).reshape(-1)[pix],
    }


def measure_fixture(pmd, diag, fx):
    """The HIP result against one referee of a committed fixture (dict of the arrays above).  Returns plain numbers."""
    import hashlib

    out = {}
    u = pmd.u.tocsr()
    u.sort_indices()
    out["ranks_equal"] = bool(np.array_equal(diag["tile_ranks"], fx["tile_ranks"]))
    out["n_rank_mismatch"] = int(np.sum(np.asarray(diag["tile_ranks"]) != fx["tile_ranks"])) if len(diag["tile_ranks"]) == len(fx["tile_ranks"]) else -1
    out["shape_equal"] = bool(tuple(u.shape) == tuple(fx["U_shape"]) and len(pmd.s) == len(fx["s"]))
    out["indptr_equal"] = bool(np.array_equal(u.indptr, fx["U_indptr"]))
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(u.indices, np.int32).tobytes()).digest(), np.uint8)
    out["indices_equal"] = bool(u.nnz == int(fx["U_nnz"]) and np.array_equal(sha, fx["U_indices_sha256"]))
    n = min(len(pmd.s), len(fx["s"]))
    s_a, s_b = np.asarray(pmd.s[:n], np.float64), np.asarray(fx["s"][:n], np.float64)
    sig = np.asarray(fx["signal"], np.int64)
    keep = (sig < n)
    keep &= rel_gaps(s_a)[np.minimum(sig, n - 1)] > GAP      # separated on the HIP side too
    out["n_signal"] = int(keep.sum())
    sg = sig[keep]
    out["s_rel_signal"] = float((np.abs(s_a[sg] - s_b[sg]) / s_b[sg]).max(initial=0.0))
    valid = s_b > RESOLVABLE * s_b[0]
    out["s_rel_resolvable"] = float((np.abs(s_a - s_b) / s_b)[valid].max(initial=0.0))
    va, vb = np.asarray(pmd.v[sg], np.float64), np.asarray(fx["Vt_signal"][keep], np.float64)
    sgn = np.where(np.sum(va * vb, axis=1) < 0, -1.0, 1.0)
    out["vt_row_err"] = np.linalg.norm(va * sgn[:, None] - vb, axis=1) / np.linalg.norm(vb, axis=1)
    # Frobenius error of the compared block of Vt (the north star's "Vt Frobenius error"), relative
    out["vt_fro_err"] = float(np.linalg.norm(va * sgn[:, None] - vb) / max(np.linalg.norm(vb), 1e-300))
    pix = np.asarray(fx["pix_sample"], np.int64)
    ur_a = np.asarray(u[pix] @ np.asarray(pmd.r[:, sg], np.float64)) * sgn[None, :]
    ur_b = np.asarray(fx["UR_signal_sample"], np.float64)[:, keep]
    out["ur_col_err"] = np.linalg.norm(ur_a - ur_b, axis=0) / np.maximum(np.linalg.norm(ur_b, axis=0), 1e-300)
    if out["indptr_equal"] and out["indices_equal"]:
        a_passed, a_gap = hip_cols(diag)
        uc = u.tocsc()
        errs, r_errs, n_cmp = [0.0], [0.0], 0
        ptr_ = fx["stable_col_ptr"]
        for k, c in enumerate(np.asarray(fx["stable_cols"], np.int64)):
            if not (a_passed[c] and a_gap[c] > GAP):
                continue
            da = np.asarray(uc.data[uc.indptr[c]:uc.indptr[c + 1]], np.float64)
            db = fx["stable_col_data"][ptr_[k]:ptr_[k + 1]]
            if len(da) != len(db) or not np.array_equal(uc.indices[uc.indptr[c]:uc.indptr[c + 1]], fx["stable_col_rows"][ptr_[k]:ptr_[k + 1]]):
                errs.append(np.inf)
                continue
            cs = -1.0 if float(da @ db) < 0 else 1.0
            errs.append(float(np.abs(cs * da - db).max()))
            ra = np.asarray(pmd.r[c, sg], np.float64) * cs * sgn
            r_errs.append(float(np.abs(ra - np.asarray(fx["R_stable_signal"][k], np.float64)[keep]).max(initial=0.0)))
            n_cmp += 1
        out["n_stable_cols_compared"] = n_cmp
        out["u_data_err_stable"] = float(max(errs))
        out["u_data_max_abs"] = float(np.abs(fx["stable_col_data"]).max(initial=0.0))
        out["r_err_stable_signal"] = float(max(r_errs))
        out["r_max_abs"] = float(fx["R_max_abs"])
    pi, pt = np.asarray(fx["probe_pix"], np.int64), np.asarray(fx["probe_frame"], np.int64)
    rec = np.einsum("pk,k,kp->p", np.asarray(u[pi] @ pmd.r), pmd.s, pmd.v[:, pt])
    out["probes"] = float(np.abs(rec - fx["probe_rec"]).max() / np.abs(fx["probe_rec"]).max())
    out["mean_rel"] = float(np.abs(np.asarray(pmd.mean_img).reshape(-1)[pix] / fx["mean_sample"] - 1).max())
    out["std_rel"] = float(np.abs(np.asarray(pmd.var_img).reshape(-1)[pix] / fx["std_sample"] - 1).max())
    return out


def fixture_summary(name, m):
    ln = [f"[{name}] tile ranks equal {m['ranks_equal']} ({m['n_rank_mismatch']} differ), shapes equal {m['shape_equal']}, U_indptr equal {m['indptr_equal']}, "
          f"U_indices equal {m['indices_equal']}"]
    ln.append(f"[{name}] signal components compared: {m['n_signal']}; s rel {m['s_rel_signal']:.2e} (all resolvable: {m['s_rel_resolvable']:.2e}); "
              f"Vt rows max {m['vt_row_err'].max(initial=0):.2e} median {np.median(m['vt_row_err']) if len(m['vt_row_err']) else 0:.2e}, Frobenius {m['vt_fro_err']:.2e}; "
              f"(U R) columns (pixel sample) max {m['ur_col_err'].max(initial=0):.2e}")
    if "u_data_err_stable" in m:
        ln.append(f"[{name}] U_data on {m['n_stable_cols_compared']} stable columns: {m['u_data_err_stable']:.2e} (max |U_data| {m['u_data_max_abs']:.3f}); "
                  f"R on those rows x signal columns: {m['r_err_stable_signal']:.2e} (max |R| {m['r_max_abs']:.3f})")
    ln.append(f"[{name}] reconstruction probes {m['probes']:.2e} of the peak; mean_img rel {m['mean_rel']:.2e}, std_img rel {m['std_rel']:.2e}")
    return ln
