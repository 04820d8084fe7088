"""Shared helpers for the parity tests (test infrastructure only)."""
import numpy as np


class DeviceSource:
    """Random source for the oracle that returns the matrices the HIP library generates
    (pmd_rng_normal), so both sides consume bit-identical Gaussian inputs."""

    def __init__(self, ctx, seed):
        self.ctx = ctx
        self.seed = int(seed)

    def _fill(self, stream, index, rows, cols):
        import torch
        from localmd_amd._lib import ptr

        out = torch.empty((rows, cols), dtype=torch.float32, device=self.ctx.device)
        self.ctx.call("pmd_rng_normal", self.seed, stream, index, 0, 1, rows, cols, 0, ptr(out), cols, 0)
        self.ctx.sync()
        return out.cpu().numpy()

    def omega(self, stream, index, rows, cols):
        return self._fill(stream, index, rows, cols)

    def noise(self, index, d1, d2, t):
        z = self._fill(2, index, d1 * d2, t)
        return z.reshape(d2, d1, t).transpose(1, 0, 2)


def sign_align(a, b, axis=0):
    """Flip the sign of each column (axis=0) / row (axis=1) of ``a`` to best match ``b``."""
    dots = np.sum(a * b, axis=axis, keepdims=True)
    return a * np.where(dots < 0, -1.0, 1.0)


def rel_err(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) /
                 max(np.linalg.norm(np.asarray(b, dtype=np.float64)), 1e-300))
