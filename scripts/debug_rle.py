"""Where does the HIP path lose accuracy on the R <= frames fixture case?  (diagnostic; GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie

Dm.QUIET = True
g = np.load(os.path.join(ROOT, "tests", "golden", "parity_rle.npz"))
c = {k[5:]: g[k].item() for k in g.files if k.startswith("case_")}
mov = make_movie(c["T"], c["d1"], c["d2"], seed=c["movie_seed"], ladder=c["ladder"], ladder_top=c["ladder_top"], ladder_ratio=c["ladder_ratio"])
ctx = Context(0)
np.random.seed(c["np_seed"])
pmd, diag = localmd_amd.localmd_decomposition(mov, (c["block"],) * 2, c["T"], max_components=c["max_components"], seed=c["seed"],
                                              thresholds=tuple(g["f32_thresholds"]), return_diagnostics=True, ctx=ctx)
sig = g["f64_signal"]
s_ref = g["f64_s"].astype(np.float64)
s_hip = pmd.s.astype(np.float64)
print("s rel (HIP vs arbiter) on signal:", np.abs(s_hip[sig] / s_ref[sig] - 1).max())
vp = diag["v_proj"].astype(np.float64)
sv = np.linalg.svd(vp, compute_uv=False)
print("svd64(Vp_hip) vs HIP s   :", np.abs(sv[sig] / s_hip[sig] - 1).max())
print("svd64(Vp_hip) vs arbiter :", np.abs(sv[sig] / s_ref[sig] - 1).max())
# orthonormality of U R on the signal components, float64
ur = np.asarray(pmd.u @ pmd.r[:, sig].astype(np.float64))
print("|(UR)^T(UR) - I| on signal comps:", np.abs(ur.T @ ur - np.eye(len(sig))).max())
# exact projection of the standardised movie on span(U): singular values via QR of U (float64)
import scipy.sparse.linalg
x = (mov.reshape(c["T"], -1).astype(np.float64) - pmd.mean_img.reshape(1, -1)) / pmd.var_img.reshape(1, -1)
if pmd.order == "F":
    x = x.reshape(c["T"], c["d1"], c["d2"]).transpose(0, 2, 1).reshape(c["T"], -1)
ud = np.asarray(pmd.u.todense(), dtype=np.float64)
q, _ = np.linalg.qr(ud)
se = np.linalg.svd(q.T @ x.T, compute_uv=False)
print("exact singular values of the projection on span(U_hip) vs arbiter:", np.abs(se[sig] / s_ref[sig] - 1).max())
print("                                                       vs HIP s  :", np.abs(se[sig] / s_hip[sig] - 1).max())
