"""HIP path against the oracle (fp32) and against the oracle's float64 arbiter form at the BASELINE configurations:
prints the measured U_data / R / s / Vt figures of tests/parity_metrics.py (tests/test_gpu_baseline_parity.py asserts
on the same figures).
    python scripts/parity_table.py config1|config2|headline|<T>x<d1>x<d2>xb<block>xr<max_components> [arbiter] [single]
"arbiter": also run the float64 form and report every fp32 result's distance to it; "single": also the oracle with
true single-precision LAPACK (scipy s-routines), the arithmetic jaxlib's CPU kernels run the reference in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity_metrics as PM
from localmd_amd._lib import Context

if __name__ == "__main__":
    ctx = Context(0)
    only = "arbiter-only" in sys.argv[2:]   # HIP against the float64 arbiter alone (the headline regime: each oracle run takes minutes)
    PM.run_config(ctx, sys.argv[1], arbiter=only or "arbiter" in sys.argv[2:], single="single" in sys.argv[2:],
                  out=lambda ln: print(ln, flush=True), fp32_oracle=not only)
    ctx.close()
