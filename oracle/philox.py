"""
TEST INFRASTRUCTURE (oracle/) -- not part of the product path.

NumPy restatement of the counter-based random-number scheme the HIP library uses
(localmd_amd/csrc/rng.hip): Philox4x32-10 + Box-Muller on 24-bit uniforms.

The reference draws its Gaussian test matrices with ``jax.random.normal``
(/root/reference/localmd/decomposition.py:62, :127, :870; pmd_loader.py:56), whose
threefry bit layout depends on the (unpinned) JAX version, so the reference's stream is
not a parity target.  What IS pinned here is our own stream:
integer Philox words are bit-exact between this file and the device; the float
Box-Muller transform agrees to ~1 ulp (libm vs. device math), which is why GPU parity
tests inject the *device-generated* matrices into the oracle rather than regenerate.

Element ``e`` of logical array (stream, index) comes from Philox block ``q = e // 4``,
lane ``e % 4``, counter = (q_lo, q_hi, index, stream), key = (seed_lo, seed_hi).
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

# stream ids (shared with localmd_amd/_streams.py and csrc/pmd_common.h)
STREAM_BG_OMEGA = 1
STREAM_SIM_NOISE = 2
STREAM_SIM_OMEGA = 3
STREAM_TILE_OMEGA = 4
STREAM_PRUNE = 5


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """counter: (n, 4) uint32, key: (2,) uint32 -> (n, 4) uint32."""
    c = counter.astype(np.uint32).copy()
    k0 = np.uint32(key[0])
    k1 = np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = PHILOX_M0 * c[:, 0].astype(np.uint64)
            p1 = PHILOX_M1 * c[:, 2].astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & mask).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & mask).astype(np.uint32)
            n0 = hi1 ^ c[:, 1] ^ k0
            n1 = lo1
            n2 = hi0 ^ c[:, 3] ^ k1
            n3 = lo0
            c = np.stack([n0, n1, n2, n3], axis=1)
            k0 = np.uint32((int(k0) + int(PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(PHILOX_W1)) & 0xFFFFFFFF)
    return c


def philox_words(seed: int, stream: int, index: int, n_blocks: int, first_block: int = 0):
    q = np.arange(first_block, first_block + n_blocks, dtype=np.uint64)
    ctr = np.empty((n_blocks, 4), dtype=np.uint32)
    ctr[:, 0] = (q & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(index & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32(stream & 0xFFFFFFFF)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key)


def normals(seed: int, stream: int, index: int, n: int) -> np.ndarray:
    """First ``n`` elements of logical normal array (stream, index), float32."""
    n_blocks = (n + 3) // 4
    w = philox_words(seed, stream, index, n_blocks)
    two_m24 = np.float32(2.0 ** -24)
    u = ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * two_m24
    r0 = np.sqrt(np.float32(-2.0) * np.log(u[:, 0]))
    r1 = np.sqrt(np.float32(-2.0) * np.log(u[:, 2]))
    th0 = np.float32(2.0 * np.pi) * u[:, 1]
    th1 = np.float32(2.0 * np.pi) * u[:, 3]
    z = np.stack(
        [r0 * np.cos(th0), r0 * np.sin(th0), r1 * np.cos(th1), r1 * np.sin(th1)], axis=1
    ).astype(np.float32)
    return z.reshape(-1)[:n]


class PhiloxSource:
    """Random source handed to the oracle when no device stream is injected."""

    def __init__(self, seed: int):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    def omega(self, stream: int, index: int, rows: int, cols: int) -> np.ndarray:
        """(rows, cols) float32; element (t, c) is logical element e = t*cols + c."""
        return normals(self.seed, stream, index, rows * cols).reshape(rows, cols)

    def noise(self, index: int, d1: int, d2: int, t: int) -> np.ndarray:
        """(d1, d2, t) float32; element (i, j, tau) is e = (i + d1*j)*t + tau."""
        z = normals(self.seed, STREAM_SIM_NOISE, index, d1 * d2 * t).reshape(d1 * d2, t)
        return z.reshape(d2, d1, t).transpose(1, 0, 2)
