"""
Parity at the BASELINE configurations (GPU): BASELINE.json configs[0] (60 x 80 x 2000 stand-in for demoMovie,
frames_to_init = 100), configs[1] (256 x 256 x 2000, <= 8 components per tile) in full, and the regime of the headline
configs[2] (T = 10^4, 50 components per tile, R > frames, the own eigensolver at order 10^4) on a 256 x 256 sub-field
of view and the R <= frames route at scale (128 x 128 x 10^4), both against committed referee fixtures
(tests/golden/parity_*.npz, tests/golden/make_parity_fixtures.py) so that no test of this file is skipped.

Referees: the fp32 oracle (NumPy arrays in fp32, LAPACK through numpy = computed in double), the same oracle with true
single-precision LAPACK (the arithmetic jaxlib's CPU kernels run the reference in), and the oracle's float64 "arbiter"
form = the exact-arithmetic limit of the reference's algorithm on the same inputs.  In the R > frames regime the
reference's algorithm squares a condition number of ~10^3 in fp32 twice (C = M^T G M, then V V^T), and two fp32
implementations agree only to a few 10^-4 on Vt: the tests therefore assert (i) bit-exact structure, (ii) absolute
tolerances on every quantity north_star names (U_data, R, s, Vt) per component class, and (iii) that the HIP result is
not farther from the float64 arbiter than the reference's own arithmetic (the single-LAPACK oracle) is.
"""
import os

import numpy as np
import pytest

from tests import parity_metrics as PM

pytestmark = pytest.mark.gpu


def _assert_common(res, u_tol, s_sig, vt_sig, ur_sig, r_sig, probe_tol, orth_tol):
    pmd, diag = res["hip"]
    ref = res["results"]["oracle fp32"]
    assert len(res["rank_mismatch"]) == 0, res["rank_mismatch"]
    m = res["measures"]["HIP vs oracle fp32"]
    assert m["csr_equal"] and m["shape_equal"], (m["csr_equal"], m["shape_equal"], m["n_components"])
    np.testing.assert_allclose(pmd.mean_img, ref.mean_img, rtol=1e-5)
    np.testing.assert_allclose(pmd.var_img, ref.std_img, rtol=2e-4)
    sig = m["signal"]
    assert sig.sum() >= 5
    assert m["u_data_err_stable"] < u_tol * m["u_data_max_abs"], m["u_data_err_stable"]
    assert m["s_rel"][sig].max() < s_sig, m["s_rel"][sig].max()
    assert m["vt_row_err"][sig].max() < vt_sig, m["vt_row_err"][sig].max()
    assert m["ur_col_err"][sig].max() < ur_sig, m["ur_col_err"][sig].max()
    assert m["r_err_stable_signal"] < r_sig * m["r_max_abs"], (m["r_err_stable_signal"], m["r_max_abs"])
    assert res["probes"]["HIP vs oracle fp32"] < probe_tol
    assert m["orth_ur"][0] < orth_tol and m["orth_vt"][0] < orth_tol, (m["orth_ur"], m["orth_vt"])
    assert m["orth_ur_weighted"][0] < 1e-4 and m["orth_vt_weighted"][0] < 1e-5, (m["orth_ur_weighted"], m["orth_vt_weighted"])
    return m


def _assert_not_farther_than_reference_arithmetic(res, slack=1.5, floor=1e-4):
    """Distance to the float64 arbiter on the signal components: HIP <= slack x (single-precision-LAPACK oracle) + floor."""
    mh = res["measures"]["HIP vs arbiter fp64"]
    ms = res["measures"]["oracle fp32 single-LAPACK vs arbiter fp64"]
    sig = mh["signal"] & ms["signal"]
    assert sig.sum() >= 5
    for key in ("s_rel", "vt_row_err", "ur_col_err"):
        assert mh[key][sig].max() <= slack * ms[key][sig].max() + floor, (key, mh[key][sig].max(), ms[key][sig].max())
    assert mh["u_data_err_stable"] <= slack * ms["u_data_err_stable"] + 1e-6


def test_config1_demo_standin_parity(gpu_ctx):
    """60 x 80 x 2000, 20 x 20 blocks, frames_to_init = 100 (so max_components is capped to 10 and the 236 tile + background
    columns exceed the 100 fitted frames: right-matrix route).  Measured round 2: Vt signal 8.9e-4, s 3.5e-5, U_data 1.5e-5."""
    lines = []
    res = PM.run_config(gpu_ctx, "config1", arbiter=True, single=True, out=lines.append)
    print("\n".join(lines))
    pmd, diag = res["hip"]
    assert diag["max_components"] == 10 and diag["rank_before"] > diag["crop"] == 100
    _assert_common(res, u_tol=5e-4, s_sig=5e-4, vt_sig=3e-3, ur_sig=3e-3, r_sig=5e-3, probe_tol=2e-3, orth_tol=5e-3)
    _assert_not_farther_than_reference_arithmetic(res)


def test_config2_full_parity(gpu_ctx):
    """256 x 256 x 2000, 625 tiles, R = 5015 > 2000 frames, 2000 components on both sides (the reference keeps the
    numerically null direction).  Measured round 2: Vt signal 5.4e-4 (HIP vs arbiter 2.5e-4; single-LAPACK oracle vs
    arbiter 6.6e-4), s signal 4.8e-4 (HIP vs arbiter 1.2e-5), U_data stable 6.7e-5 of 0.245 (vs arbiter 3.1e-6)."""
    lines = []
    res = PM.run_config(gpu_ctx, "config2", arbiter=True, single=True, out=lines.append)
    print("\n".join(lines))
    pmd, diag = res["hip"]
    assert pmd.s.shape == (2000,) and diag["rank_before"] == res["results"]["oracle fp32"].diag["rank_before"]
    _assert_common(res, u_tol=1e-3, s_sig=2e-3, vt_sig=2e-3, ur_sig=5e-2, r_sig=5e-1, probe_tol=1e-2, orth_tol=1e-2)
    _assert_not_farther_than_reference_arithmetic(res)
    # against the arbiter itself the HIP path holds tighter figures than against the fp32 oracle
    mh = res["measures"]["HIP vs arbiter fp64"]
    sig = mh["signal"]
    # (round 3, chunked fp32 accumulation + fp64 eigenvector refinement: s 2.5e-6, Vt 4.0e-5, U_data 2.8e-6 - the NORTH STAR's
    # Vt < 1e-4, against the exact result of the reference's algorithm, at a BASELINE configuration with R > frames)
    assert mh["s_rel"][sig].max() < 2e-5 and mh["vt_row_err"][sig].max() < 1e-4 and mh["u_data_err_stable"] < 1e-5, \
        (mh["s_rel"][sig].max(), mh["vt_row_err"][sig].max(), mh["u_data_err_stable"])


def _run_fixture_case(gpu_ctx, name):
    """HIP path on the inputs of a committed referee fixture (tests/golden/parity_<name>.npz, written in the build container
    by tests/golden/make_parity_fixtures.py from the CPU oracle with the host Philox source).  The device generator restates
    the same counter-based streams, so the same seed gives the same Gaussian test matrices to ~1 ulp; thresholds are the
    fixture's.  Returns {"f32": measures against the fp32 oracle, "f64": against the float64 arbiter}, diag."""
    import localmd_amd
    from localmd_amd import decomposition as Dm
    from localmd_amd.synthetic import make_movie

    Dm.QUIET = True
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"parity_{name}.npz"), allow_pickle=False)
    c = {k[5:]: g[k].item() for k in g.files if k.startswith("case_")}
    mov = make_movie(c["T"], c["d1"], c["d2"], seed=c["movie_seed"], ladder=c["ladder"], ladder_top=c["ladder_top"],
                     ladder_ratio=c["ladder_ratio"], ladder_smooth=float(c.get("ladder_smooth", 0.0)))
    np.random.seed(c["np_seed"])
    pmd, diag = localmd_amd.localmd_decomposition(mov, (c["block"], c["block"]), c["T"], max_components=c["max_components"], seed=c["seed"],
                                                  thresholds=tuple(g["f32_thresholds"]), return_diagnostics=True, ctx=gpu_ctx)
    out = {}
    for ref in ("f32", "f64"):
        fx = {k[len(ref) + 1:]: g[k] for k in g.files if k.startswith(ref + "_")}
        out[ref] = PM.measure_fixture(pmd, diag, fx)
        print("\n".join(PM.fixture_summary(f"{name}: HIP vs {'oracle fp32' if ref == 'f32' else 'arbiter fp64'}", out[ref])))
    gpu_ctx.release_workspace()
    return out, diag


def _assert_structure(m):
    assert m["ranks_equal"], m["n_rank_mismatch"]
    assert m["shape_equal"] and m["indptr_equal"] and m["indices_equal"]
    assert m["mean_rel"] < 1e-5 and m["std_rel"] < 2e-4, (m["mean_rel"], m["std_rel"])


def test_r_le_frames_at_scale_meets_the_north_star_vt_tolerance(gpu_ctx):
    """128 x 128 x 10000, 20 x 20 blocks, 50 components per tile (round-2 verdict, item 1d): R = 2027 <= frames, i.e.
    right_mat = I (decomposition.py:978-979), the eigenvector route pmd_gram_u + pmd_orthogonalize with the library's own
    eigensolver at order ~ 2000 and pmd_projected_svd at the same order.  Against the committed fp32-oracle fixture:
    structure bit-exact, and the NORTH STAR's tolerance - Vt error < 1e-4 - on the signal components (>= 20 of them:
    the movie carries a ladder of bright sources with separated singular values)."""
    out, diag = _run_fixture_case(gpu_ctx, "rle")
    assert diag["rank_before"] <= diag["crop"] == 10000 and diag["orthogonalizer"] == "eigh"
    for ref in ("f32", "f64"):
        m = out[ref]
        _assert_structure(m)
        assert m["n_signal"] >= 20, m["n_signal"]
        assert m["vt_fro_err"] < 1e-4 and m["vt_row_err"].max() < 1e-4, (ref, m["vt_fro_err"], m["vt_row_err"].max())
        assert m["s_rel_signal"] < 2e-5, m["s_rel_signal"]
        assert m["ur_col_err"].max() < 2e-4, m["ur_col_err"].max()
        assert m["u_data_err_stable"] < 1e-4 * m["u_data_max_abs"] and m["n_stable_cols_compared"] >= 32
        assert m["r_err_stable_signal"] < 2e-4 * m["r_max_abs"]
        assert m["probes"] < 5e-4, m["probes"]
    # the reference's own arithmetic (oracle with single-precision LAPACK) against the arbiter, from the fixture: the HIP path
    # is not farther from the exact result than that (measured: Vt 2.8e-5 against 4.4e-5)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "parity_rle.npz"), allow_pickle=False)
    if "ref32s_vt_row_err" in g.files:
        ma = out["f64"]
        assert ma["vt_row_err"].max() <= 1.5 * g["ref32s_vt_row_err"].max() + 1e-5
        assert ma["s_rel_signal"] <= 1.5 * g["ref32s_s_rel_signal"].max() + 1e-5


def test_headline_regime_parity(gpu_ctx):
    """T = 10^4 frames, 50 components per tile, 1156 tiles of a 352 x 352 field of view: R = 17 327 > frames, the Cholesky route
    and the library's own eigensolver at order 10^4 - the regime that dominates bench.py - against the committed fixture
    (fp32 oracle and float64 arbiter, generated once in the build container: 15 + 22 minutes of CPU; plus the distances of
    the single-precision-LAPACK oracle to the arbiter, the reference's own arithmetic).  The movie carries a ladder of 28
    bright band-limited sources: 25 leading components with singular values separated by more than 2 % are compared one by
    one.  Measured (round 3): vs the arbiter s 1.1e-4, Vt rows 1.1e-3 (median 2.2e-4), (U R) 7.4e-2, R 3.7e-3 of 0.86; the
    NumPy oracle (double-precision LAPACK on fp32 data) sits at 2.8e-5 / 2.9e-4 / 1.4e-2 from the arbiter."""
    out, diag = _run_fixture_case(gpu_ctx, "headline")
    assert diag["rank_before"] > diag["crop"] == 10000 and diag["orthogonalizer"] == "cholesky"
    assert diag["null_direction"]["split_off"]
    for ref in ("f32", "f64"):
        m = out[ref]
        _assert_structure(m)
        assert m["n_signal"] >= 20, m["n_signal"]
        assert m["u_data_err_stable"] < 1e-4 * m["u_data_max_abs"] and m["n_stable_cols_compared"] >= 32
    mo, ma = out["f32"], out["f64"]
    # against the exact result of the reference's algorithm (float64 arbiter)
    assert ma["s_rel_signal"] < 3e-4 and ma["vt_row_err"].max() < 3e-3 and np.median(ma["vt_row_err"]) < 6e-4, \
        (ma["s_rel_signal"], ma["vt_row_err"].max(), np.median(ma["vt_row_err"]))
    assert ma["ur_col_err"].max() < 1.5e-1 and ma["r_err_stable_signal"] < 1e-2 * ma["r_max_abs"], (ma["ur_col_err"].max(), ma["r_err_stable_signal"])
    assert ma["probes"] < 5e-2, ma["probes"]
    # against the fp32 oracle (whose own distance to the arbiter is part of the figure in this regime)
    assert mo["s_rel_signal"] < 3e-4 and mo["vt_row_err"].max() < 3e-3 and mo["ur_col_err"].max() < 1.5e-1, \
        (mo["s_rel_signal"], mo["vt_row_err"].max(), mo["ur_col_err"].max())
    assert mo["probes"] < 5e-2, mo["probes"]
    # against the reference's own arithmetic (single-precision LAPACK), when its distances are in the fixture: the HIP path is
    # not farther from the arbiter than 1.5 x that
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "parity_headline.npz"), allow_pickle=False)
    if "ref32s_vt_row_err" in g.files:
        print("single-precision-LAPACK oracle vs arbiter: s %.2e, Vt rows %.2e, (U R) %.2e, probes %.2e" % (
            g["ref32s_s_rel_signal"].max(), g["ref32s_vt_row_err"].max(), g["ref32s_ur_col_err"].max(), float(g["ref32s_probes"])))
        assert ma["vt_row_err"].max() <= 1.5 * g["ref32s_vt_row_err"].max() + 1e-4
        assert ma["s_rel_signal"] <= 1.5 * g["ref32s_s_rel_signal"].max() + 1e-4
        assert ma["ur_col_err"].max() <= 1.5 * g["ref32s_ur_col_err"].max() + 1e-4


def test_hip_path_reproduces_committed_golden_fixture(gpu_ctx):
    """tests/golden/oracle_small.npz (written by tests/golden/make_golden.py from the oracle with the HOST Philox source,
    seed 5) against the HIP path with the same seed: the device generator restates the same counter-based streams
    (rng.hip; equal up to the rounding of logf / sincosf), so the committed vectors are reproduced without running the
    oracle - tile ranks and CSR structure bit for bit, everything else to fp32 accuracy."""
    import os
    import scipy.sparse
    import localmd_amd
    from localmd_amd import decomposition as Dm
    from localmd_amd.synthetic import make_movie
    from tests.util import sign_align

    Dm.QUIET = True
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_small.npz"), allow_pickle=False)
    T, d1, d2 = (int(v) for v in g["movie_shape"])
    mov = make_movie(T, d1, d2, seed=int(g["movie_seed"]))
    np.random.seed(3)
    pmd, diag = localmd_amd.localmd_decomposition(mov, (20, 16), 400, max_components=5, background_rank=2, seed=5,
                                                  thresholds=tuple(g["thresholds"]), return_diagnostics=True, ctx=gpu_ctx)
    np.testing.assert_array_equal(diag["tile_ranks"], g["tile_ranks"])
    assert tuple(pmd.u.shape) == tuple(g["U_shape"])
    np.testing.assert_array_equal(pmd.u.indptr, g["U_indptr"])
    np.testing.assert_array_equal(pmd.u.indices, g["U_indices"])
    np.testing.assert_allclose(pmd.mean_img, g["mean_img"], rtol=1e-5)
    np.testing.assert_allclose(pmd.var_img, g["std_img"], rtol=2e-4)
    assert pmd.s.shape == g["s"].shape and pmd.r.shape == g["R"].shape and pmd.v.shape == g["Vt"].shape
    strong = g["s"] > 5e-2 * g["s"][0]
    np.testing.assert_allclose(pmd.s[strong], g["s"][strong], rtol=2e-4)
    # column signs of U are arbitrary per tile component: compare |U_data| where it is not tiny, and the reconstruction
    big = np.abs(g["U_data"]) > 1e-2
    np.testing.assert_allclose(np.abs(pmd.u.data[big]), np.abs(g["U_data"][big]), rtol=5e-3)
    u_ref = scipy.sparse.csr_matrix((g["U_data"], g["U_indices"], g["U_indptr"]), shape=tuple(g["U_shape"]))
    rng = np.random.default_rng(0)
    pi, pt = rng.integers(0, d1 * d2, 500), rng.integers(0, T, 500)
    rec = np.einsum("pk,k,kp->p", np.asarray(pmd.u[pi] @ pmd.r), pmd.s, pmd.v[:, pt])
    rec0 = np.einsum("pk,k,kp->p", np.asarray(u_ref[pi] @ g["R"]), g["s"], g["Vt"][:, pt])
    assert np.abs(rec - rec0).max() < 2e-3 * np.abs(rec0).max()
    va = sign_align(pmd.v, g["Vt"], axis=1)
    gaps = PM.rel_gaps(g["s"])
    sig = strong & (gaps > PM.GAP)
    assert np.linalg.norm(va[sig] - g["Vt"][sig]) / np.linalg.norm(g["Vt"][sig]) < 5e-4
