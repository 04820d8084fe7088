import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as tp
from tests import parity_metrics as PM
from localmd_amd._lib import Context
ctx = Context(0)
case = dict(T=303, d1=50, d2=50, block=(10, 10), frames=303, kw=dict(max_components=8, background_rank=2))
mov = tp._movie(case["T"], case["d1"], case["d2"], seed=case["T"])
for mode in ("auto", "eigh"):
    pmd, diag, ref = tp._compare_full(ctx, mov, case["block"], case["frames"], sim_iters=8, orthogonalizer=mode, **case["kw"])
    print(mode, "route", diag["orthogonalizer"], "rank_before", diag["rank_before"], "crop", diag["crop"], "rank_after", diag["rank_after"], len(ref.s))
    m = PM.measure(pmd, ref)
    n = len(m["s_rel"])
    ur = np.asarray(pmd.u @ pmd.r, np.float64)
    g = ur.T @ ur - np.eye(ur.shape[1])
    i, j = np.unravel_index(np.abs(g).argmax(), g.shape)
    print("  worst orth entry", g[i, j], "at", i, j, "s there", pmd.s[i], pmd.s[j], "s1", pmd.s[0], "valid", m["valid"].sum(), "of", n)
    print("  diag errs of last 10:", np.round(np.diag(g)[-10:], 4))
    print("  s tail hip", pmd.s[-6:], "ref", ref.s[-6:])
    print("  orth_ur", m["orth_ur"], "orth_vt", m["orth_vt"], "probe", PM.probes(pmd, ref, mov.shape))
