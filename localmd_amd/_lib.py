"""
ctypes binding of libpmd_hip.so (C ABI: include/pmd_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, the product
raises.  Build the library with ``localmd_amd/csrc/build.sh`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMD_HIP_LIB") or os.path.join(_HERE, "libpmd_hip.so")

c_i, c_l, c_f, c_d, c_p = C.c_int, C.c_long, C.c_float, C.c_double, C.c_void_p
c_u64, c_u32, c_sz = C.c_uint64, C.c_uint32, C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/pmd_hip.h
SIGNATURES = {
    "pmd_version": (c_i, []),
    "pmd_comm_unique_id": (c_i, [c_p]),
    "pmd_ctx_create": (c_i, [c_i, c_p, C.POINTER(c_p)]),
    "pmd_ctx_destroy": (c_i, [c_p]),
    "pmd_ctx_set_stream": (c_i, [c_p, c_p]),
    "pmd_ctx_sync": (c_i, [c_p]),
    "pmd_last_error": (C.c_char_p, [c_p]),
    "pmd_profile_enable": (c_i, [c_p, c_i]),
    "pmd_profile_query": (c_i, [c_p, C.c_char_p, C.POINTER(c_d), C.POINTER(c_i)]),
    "pmd_profile_names": (c_i, [c_p, C.c_char_p, c_i]),
    "pmd_tile_dpad": (c_i, [c_i]),
    "pmd_tile_rpad": (c_i, [c_i]),
    "pmd_time_ld": (c_l, [c_l]),
    "pmd_rng_normal": (c_i, [c_p, c_u64, c_u32, c_u32, c_u32, c_i, c_l, c_i, c_i, c_p, c_l, c_l]),
    "pmd_stats_workspace_bytes": (c_sz, [c_i, c_l, c_i]),
    "pmd_stats": (c_i, [c_p, c_p, c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_sz]),
    "pmd_standardize_transpose": (c_i, [c_p, c_p, c_l, c_p, c_i, c_p, c_p, c_p, c_l]),
    "pmd_background_rsvd_workspace_bytes": (c_sz, [c_l, c_i, c_i]),
    "pmd_background_rsvd": (c_i, [c_p, c_p, c_l, c_i, c_l, c_i, c_u64, c_p, c_p, c_sz]),
    "pmd_bg_project_workspace_bytes": (c_sz, [c_l, c_i]),
    "pmd_bg_project": (c_i, [c_p, c_p, c_l, c_i, c_l, c_p, c_i, c_p, c_l, c_p, c_sz]),
    "pmd_bg_filter": (c_i, [c_p, c_p, c_p, c_l, c_i, c_l, c_p, c_i, c_p, c_l]),
    "pmd_scale_rows": (c_i, [c_p, c_p, c_l, c_i, c_l, c_p]),
    "pmd_threshold_sim_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "pmd_threshold_sim": (c_i, [c_p, c_i, c_i, c_i, c_i, c_u64, c_p, c_p, c_sz]),
    "pmd_tiles_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_l]),
    "pmd_tiles_decompose": (c_i, [c_p, c_p, c_l, c_l, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_f, c_f,
                                  c_i, c_u64, c_u32, c_u32, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_sz]),
    "pmd_tiles_decompose_staged": (c_i, [c_p, c_p, c_l, c_l, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_f,
                                         c_f, c_i, c_u64, c_u32, c_u32, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_i]),
    "pmd_csr_rows_spmm": (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_p, c_l]),
    "pmd_diag_workspace_bytes": (c_sz, [c_l, c_l]),
    "pmd_neighbour_moments": (c_i, [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_sz]),
    "pmd_lag_moments": (c_i, [c_p, c_p, c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_sz]),
    "pmd_neighbour_image": (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p]),
    "pmd_lag_image": (c_i, [c_p, c_p, c_l, c_l, c_p]),
    "pmd_transpose_affine": (c_i, [c_p, c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_l]),
    "pmd_tiles_hook_offsets": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_p, c_p]),
    "pmd_tiles_residual_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_l]),
    "pmd_tiles_residual": (c_i, [c_p, c_p, c_l, c_l, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_u64, c_u32, c_u32,
                                 c_p, c_p, c_p, c_p, c_p, c_p, c_sz]),
    "pmd_tiles_truncate": (c_i, [c_p, c_p, c_i, c_p, c_i, c_i]),
    "pmd_weight_tiles": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i]),
    "pmd_tiles_project": (c_i, [c_p, c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_l, c_i]),
    "pmd_tiles_project_ranked": (c_i, [c_p, c_p, c_l, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_l, c_i, c_p]),
    "pmd_compact_rows": (c_i, [c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_l, c_i]),
    "pmd_gram_u": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_l, c_i, c_p, c_l]),
    "pmd_orthogonalize_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "pmd_orthogonalize": (c_i, [c_p, c_p, c_i, c_p, c_i, c_l, c_p, c_l, C.POINTER(c_i), c_p, c_sz]),
    "pmd_projected_svd_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "pmd_projected_svd": (c_i, [c_p, c_p, c_i, c_l, c_p, c_i, c_i, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_sz]),
    "pmd_gram_blocks": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_l, c_i, c_p, c_p, c_p, c_l]),
    "pmd_gram_apply": (c_i, [c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_l]),
    "pmd_orthogonalize_factored_workspace_bytes": (c_sz, [c_i]),
    "pmd_orthogonalize_factored": (c_i, [c_p, c_p, c_i, c_i, c_l, c_p, c_l, c_p, c_l, C.POINTER(c_i), c_p, c_sz]),
    "pmd_orthogonalize_chol": (c_i, [c_p, c_p, c_i, c_i, c_l, c_p, c_l, c_p, c_l, C.POINTER(c_i), c_p, c_sz]),
    "pmd_orthogonalize_chol_workspace_bytes": (c_sz, [c_i, c_i]),
    "pmd_projected_svd_factored_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "pmd_projected_svd_factored": (c_i, [c_p, c_p, c_i, c_i, c_l, c_p, c_i, c_l, c_p, c_i, c_l, c_p, c_l, c_p, c_p, c_l,
                                         c_p, c_l, c_p, c_p, c_i, c_p, c_sz]),
    "pmd_gram_mtgm_workspace_bytes": (c_sz, [c_i, c_i]),
    "pmd_gram_mtgm_ld": (c_l, [c_i]),
    "pmd_gram_mtgm": (c_i, [c_p, c_p, c_i, c_i, c_l, c_p, c_l, c_p, c_l, c_p, c_sz]),
    "pmd_chol_inverse_workspace_bytes": (c_sz, [c_i]),
    "pmd_chol_inverse": (c_i, [c_p, c_p, c_i, c_l, c_i, C.POINTER(c_i), c_p, c_sz]),
    "pmd_ctx_set_null_cutoff": (c_i, [c_p, C.c_float]),
    "pmd_comm_init": (c_i, [c_p, c_p, c_i, c_i]),
    "pmd_comm_destroy": (c_i, [c_p]),
    "pmd_comm_all_reduce_f32": (c_i, [c_p, c_p, c_sz]),
    "pmd_comm_all_gather": (c_i, [c_p, c_p, c_p, c_sz]),
    "pmd_transpose": (c_i, [c_p, c_p, c_l, c_i, c_i, c_p, c_l]),
    "pmd_psvd_vp_gram": (c_i, [c_p, c_p, c_i, c_i, c_l, c_p, c_i, c_l, c_i, c_p, c_l, c_p, c_l]),
    "pmd_psvd_finish_workspace_bytes": (c_sz, [c_i]),
    "pmd_psvd_finish": (c_i, [c_p, c_p, c_l, c_i, c_p, c_i, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_sz]),
    "pmd_csr_count": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p]),
    "pmd_csr_fill": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i,
                           c_p, c_p, c_p, c_p, c_i]),
    "pmd_scratch_trim": (c_i, [c_p, C.c_size_t]),
    "pmd_gemm_split_active": (c_i, [c_p, c_i, c_i, c_i]),
    "pmd_gemm": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p, c_l, c_f, c_p, c_l]),
    "pmdk_tile_atx": (c_i, [c_p, c_p, c_l, c_p, c_i, c_l, c_i, c_p, c_l, c_i, c_p, c_l, c_l, c_i, c_i, c_i]),
    "pmdk_tile_atx_rows": (c_i, [c_p, c_p, c_l, c_p, c_i, c_l, c_i, c_p, c_l, c_i, c_p, c_l, c_l, c_i, c_i, c_i, c_i]),
    "pmdk_tile_xbt": (c_i, [c_p, c_p, c_l, c_p, c_i, c_l, c_i, c_p, c_l, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_i]),
    "pmdk_tile_gram": (c_i, [c_p, c_p, c_l, c_l, c_i, c_i, c_i, c_p]),
    "pmdk_tile_rowmix": (c_i, [c_p, c_p, c_l, c_l, c_p, c_l, c_i, c_i, c_p, c_l, c_l, c_i, c_i]),
    "pmdk_small_qr": (c_i, [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_i, c_i]),
    "pmdk_small_eig": (c_i, [c_p, c_p, c_i, c_i, c_i, c_d, c_p, c_p, c_i]),
    "pmdk_tile_pool_bin": (c_i, [c_p, c_p, c_l, c_l, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_l]),
    "pmdk_roughness": (c_i, [c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_l, c_i, c_i, c_p, c_i]),
    "pmdk_syevd": (c_i, [c_p, c_i, c_p, c_l, c_p, c_p, c_p]),
    "pmdk_sytrd": (c_i, [c_p, c_i, c_p, c_l, c_p, c_p, c_p, c_i]),
    "pmdk_sy2sb": (c_i, [c_p, c_i, c_p, c_l, c_p, C.POINTER(c_i)]),
    "pmdk_sytrd2": (c_i, [c_p, c_i, c_p, c_l, c_p, c_p, c_p, C.POINTER(c_i)]),
}

_lib = None


class PMDLibraryError(RuntimeError):
    pass


def load():
    """Load libpmd_hip.so (once) and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: device pointers are shared with torch, so both must bind the same HIP runtime
    # instance (torch bundles its own libamdhip64 / librocblas; whichever loads first wins).
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise PMDLibraryError(
            f"{LIB_PATH} not found: build it with localmd_amd/csrc/build.sh (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device (or host) pointer of a torch tensor / None -> c_void_p."""
    if t is None:
        return c_p(0)
    return c_p(t.data_ptr())


class Context:
    """One pmd_ctx per (thread, device), bound to torch's current stream on that device."""

    def __init__(self, device_index=0):
        import torch

        if not torch.cuda.is_available():
            raise PMDLibraryError("no HIP device visible: the PMD hot path has no CPU fallback")
        self.lib = load()
        self.device_index = int(device_index)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        handle = c_p(0)
        rc = self.lib.pmd_ctx_create(self.device_index, c_p(stream), C.byref(handle))
        if rc != 0:
            raise PMDLibraryError(f"pmd_ctx_create failed with code {rc}")
        self.handle = handle
        self._ws = None
        self._side = None
        self._pin_free = {}     # page-locked staging buffers for host -> device table uploads, by size class
        self._pin_used = []

    def pinned(self, nbytes):
        """A page-locked uint8 staging buffer of at least nbytes, owned by the context until release_pinned() (an asynchronous
        copy out of pageable memory is a synchronous staged copy in disguise: 10+ ms for the 16 MB pixel lists of a
        16 129-tile grid).  Buffers are kept between calls by power-of-two size class."""
        import torch

        size = 1 << max(12, int(nbytes - 1).bit_length())
        pool = self._pin_free.setdefault(size, [])
        buf = pool.pop() if pool else torch.empty(size, dtype=torch.uint8, pin_memory=True)
        self._pin_used.append((size, buf))
        return buf

    def release_pinned(self):
        """Return the staging buffers to the pool (call after a synchronisation: their transfers have completed)."""
        for size, buf in self._pin_used:
            self._pin_free.setdefault(size, []).append(buf)
        self._pin_used = []

    def side(self):
        """A second context on its own HIP stream (``.stream``) for work that may overlap the main stream's
        (e.g. the latency-bound Cholesky step next to a large GEMM).  Created on first use, closed with self."""
        import torch

        if self._side is None:
            # high priority: its short kernels should slip in between the workgroups of a large GEMM on the main stream
            st = torch.cuda.Stream(device=self.device, priority=-1)
            with torch.cuda.stream(st):
                sc = Context(self.device_index)
            sc.stream = st
            self._side = sc
        return self._side

    def close(self):
        if getattr(self, "_side", None) is not None:
            self._side.close()
            self._side = None
        if getattr(self, "handle", None):
            self.lib.pmd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.handle, *args)
        if rc != 0:
            msg = self.lib.pmd_last_error(self.handle)
            raise PMDLibraryError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")

    def sync(self):
        self.call("pmd_ctx_sync")

    def profile_enable(self, on=True):
        self.call("pmd_profile_enable", 1 if on else 0)

    def profile_summary(self):
        """{kernel group: (total ms, launches)} measured with HIP events since profile_enable."""
        buf = C.create_string_buffer(4096)
        self.call("pmd_profile_names", buf, 4096)
        out = {}
        for name in buf.value.decode().split("\n"):
            if not name:
                continue
            ms, cnt = c_d(0), c_i(0)
            self.call("pmd_profile_query", name.encode(), C.byref(ms), C.byref(cnt))
            out[name] = (ms.value, cnt.value)
        return out

    def workspace(self, nbytes):
        """A reusable device scratch buffer of at least nbytes (uint8 tensor)."""
        import torch

        nbytes = int(nbytes)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def release_workspace(self):
        self._ws = None
        if getattr(self, "_side", None) is not None:
            self._side._ws = None
