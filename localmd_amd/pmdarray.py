"""
PMDArray: array-like view of a decomposition ``Y ~ mean + std * ([U R] diag(s) Vt)``.

Mirrors /root/reference/localmd/pmdarray.py:7-171 (constructor signature, ``u/r/s/v``,
``mean_img``/``var_img``, ``order``, ``shape``/``dtype``/``ndim``, ``spatial_crop``,
``temporal_crop``, ``__getitem__``).  One documented deviation: the reference's two-key
branch (``arr[frames, rows]``) raises TypeError (pmdarray.py:145-149 passes two positional
arguments to ``spatial_crop``); here it selects rows and all columns.

``save_npz``/``load_npz`` write/read the reference's on-disk layout (README.md:31-62,
demos/official_demo.ipynb cells 8 and 10).
"""
from typing import Union

import numpy as np
import scipy.sparse


class PMDArray:
    def __init__(self, u, r, s, v, data_shape, data_order, mean_img, std_img):
        self.order = data_order
        self.num_frames, self.fov_dim1, self.fov_dim2 = (int(x) for x in data_shape)
        self._u = u.tocsr()
        self._r = r
        self._s = s
        self._v = v
        self._combined = None
        self._dev = None
        self.mean_img = mean_img
        self.var_img = std_img  # NB: a noise *std* estimate, stored under the reference's name
        self.row_indices = np.arange(self.fov_dim1 * self.fov_dim2).reshape(
            (self.fov_dim1, self.fov_dim2), order=self.order
        )

    # ---- optional device residency: __getitem__ then expands on the GPU (libpmd_hip.so: pmd_csr_rows_spmm,
    # pmd_gemm, pmd_transpose_affine) and returns the same NumPy arrays
    def to_device(self, device=None, ctx=None):
        """Upload U (CSR, values as float32), R diag(s), Vt once.  Afterwards ``self[...]`` runs on the GPU: no
        (R x T) combined temporal matrix is formed on the host.  Fails loudly without the HIP library / a device."""
        import torch
        from ._lib import Context

        own = ctx is None
        if own:
            ctx = Context(0 if device is None else int(device))
        dev = ctx.device
        u = self._u
        self._dev = {
            "ctx": ctx, "own_ctx": own,
            "indptr": torch.from_numpy(u.indptr.astype(np.int64)).to(dev),
            "indices": torch.from_numpy(u.indices.astype(np.int32)).to(dev),
            "data": torch.from_numpy(u.data.astype(np.float32)).to(dev),
            "rs": torch.from_numpy(np.ascontiguousarray(self._r * self._s[None, :], dtype=np.float32)).to(dev),
            "v": torch.from_numpy(np.ascontiguousarray(self._v, dtype=np.float32)).to(dev),
            "row_nnz": np.diff(u.indptr).astype(np.int64),
        }
        return self

    def to_host(self):
        """Drop the device copies; ``self[...]`` goes back to the reference's SciPy / NumPy expansion."""
        dev = getattr(self, "_dev", None)
        self._dev = None
        if dev is not None and dev["own_ctx"]:
            dev["ctx"].close()
        return self

    def _getitem_device(self, skey, tkey):
        import torch
        from ._lib import ptr

        dv = self._dev
        ctx = dv["ctx"]
        dev = ctx.device
        if skey[0] is None or skey[1] is None or tkey is None:
            raise ValueError("Cannot pass in None for indexing")
        k0, k1 = self._as_list(skey[0]), self._as_list(skey[1])
        used_rows = self.row_indices[k0, k1]
        fov = used_rows.shape
        sel = np.ascontiguousarray(used_rows.reshape(-1), dtype=np.int32)  # C order: out[f].reshape(fov) is the frame
        fidx = np.arange(self.num_frames)[self._as_list(tkey)].reshape(-1)
        n_sel, nf = int(sel.size), int(fidx.size)
        out = np.empty((nf, n_sel), dtype=np.float32)
        if n_sel == 0:
            # an empty spatial selection: the reference's reshape cannot infer the frame axis and raises (pmdarray.py:165);
            # the same statement, so the same ValueError, here
            np.empty((0, nf), dtype=np.float32).reshape(fov + (-1,), order=self.order)
        if n_sel == 0 or nf == 0:
            return out.reshape((nf,) + fov).squeeze()
        rcols, rp = dv["rs"].shape
        T = dv["v"].shape[1]
        sel_dev = torch.from_numpy(sel).to(dev)
        scale = torch.from_numpy(np.ascontiguousarray(self.var_img[k0, k1], dtype=np.float32).reshape(-1)).to(dev)
        shift = torch.from_numpy(np.ascontiguousarray(self.mean_img[k0, k1], dtype=np.float32).reshape(-1)).to(dev)
        nnz_sel = int(dv["row_nnz"][sel].sum())
        # dense product first (temporal = (R s) Vt[:, frames], then the sparse rows) or last ((U_sel R s) first)
        temporal_first = float(rcols) * rp * nf + float(nnz_sel) * nf < float(nnz_sel) * rp + float(n_sel) * rp * nf
        chunk = max(1, min(nf, (1 << 28) // max(n_sel, rcols if temporal_first else 1)))
        if chunk >= 4:
            chunk -= chunk % 4
        ldc = (chunk + 3) // 4 * 4
        acc = torch.zeros((n_sel, ldc), dtype=torch.float32, device=dev)
        outc = torch.empty((chunk, n_sel), dtype=torch.float32, device=dev)
        w = None
        if temporal_first:
            ct = torch.zeros((rcols, ldc), dtype=torch.float32, device=dev)
        else:
            w = torch.empty((n_sel, rp), dtype=torch.float32, device=dev)
            ctx.call("pmd_csr_rows_spmm", ptr(dv["indptr"]), ptr(dv["indices"]), ptr(dv["data"]), ptr(sel_dev), n_sel,
                     ptr(dv["rs"]), rp, rp, ptr(w), rp)
        contiguous = nf == 1 or bool(np.all(np.diff(fidx) == 1))
        fidx_dev = None if contiguous else torch.from_numpy(fidx.astype(np.int64)).to(dev)
        for f0 in range(0, nf, chunk):
            fn = min(chunk, nf - f0)
            if contiguous:
                vsel, ldv = dv["v"][:, int(fidx[f0]):], T
            else:
                vsel = dv["v"].index_select(1, fidx_dev[f0:f0 + fn]).contiguous()
                ldv = fn
            fn4 = min((fn + 3) // 4 * 4, ldc)
            if temporal_first:
                ctx.call("pmd_gemm", 0, 0, rcols, fn, rp, 1.0, ptr(dv["rs"]), rp, ptr(vsel), ldv, 0.0, ptr(ct), ldc)
                ctx.call("pmd_csr_rows_spmm", ptr(dv["indptr"]), ptr(dv["indices"]), ptr(dv["data"]), ptr(sel_dev), n_sel,
                         ptr(ct), ldc, fn4, ptr(acc), ldc)
            else:
                ctx.call("pmd_gemm", 0, 0, n_sel, fn, rp, 1.0, ptr(w), rp, ptr(vsel), ldv, 0.0, ptr(acc), ldc)
            ctx.call("pmd_transpose_affine", ptr(acc), ldc, n_sel, fn, ptr(scale), ptr(shift), ptr(outc), n_sel)
            ctx.sync()
            out[f0:f0 + fn] = outc[:fn].cpu().numpy()
        return out.reshape((nf,) + fov).squeeze()

    @property
    def _combined_temporal(self):
        """(R * s) V, built on first use and cached: __getitem__ is then one sparse-dense product
        (the reference forms it eagerly in __init__, pmdarray.py:50-52)."""
        if self._combined is None:
            self._combined = (self._r * self._s[None, :]).dot(self._v)
        return self._combined

    u = property(lambda self: self._u)
    r = property(lambda self: self._r)
    s = property(lambda self: self._s)
    v = property(lambda self: self._v)

    @property
    def dtype(self):
        return np.float32

    @property
    def shape(self):
        return (self.num_frames, self.fov_dim1, self.fov_dim2)

    @property
    def ndim(self):
        return 3

    @staticmethod
    def _as_list(key):
        return [key] if isinstance(key, (int, np.integer)) else key

    def spatial_crop(self, key):
        """key: 2-tuple of row / column selectors -> (u rows (csr), mean, std, implied fov shape)."""
        if key[0] is None or key[1] is None:
            raise ValueError("Cannot pass in None for indexing")
        k0, k1 = self._as_list(key[0]), self._as_list(key[1])
        used_rows = self.row_indices[k0, k1]
        mean_used = self.mean_img[k0, k1]
        var_used = self.var_img[k0, k1]
        u_used = self._u[used_rows.reshape((-1,), order=self.order)]
        return u_used, mean_used, var_used, used_rows.shape

    def temporal_crop(self, key: Union[np.ndarray, slice, list, int]) -> np.ndarray:
        if key is None:
            raise ValueError("Cannot use None for indexing")
        return self._combined_temporal[:, self._as_list(key)]

    def __getitem__(self, key) -> np.ndarray:
        """self[frames], self[frames, rows], self[frames, rows, cols]; no dimension expansion."""
        if key is None:
            raise ValueError("Cannot use None for indexing")
        if not isinstance(key, tuple):
            key = (key,)
        everything = slice(None, None, None)
        if len(key) == 1:
            skey = (everything, everything)
        elif len(key) == 2:
            skey = (key[1], everything)
        elif len(key) == 3:
            skey = (key[1], key[2])
        else:
            raise ValueError("Too many values to unpack in __getitem__")
        if self._dev is not None:
            return self._getitem_device(skey, key[0])
        spatial, mean_used, var_used, fov = self.spatial_crop(skey)
        temporal = self.temporal_crop(key[0])
        out = spatial.dot(temporal).reshape(fov + (-1,), order=self.order)
        out = out * np.expand_dims(var_used, axis=var_used.ndim) + np.expand_dims(mean_used, axis=mean_used.ndim)
        out = np.transpose(out, axes=(out.ndim - 1, *range(out.ndim - 1)))
        return out.squeeze().astype(self.dtype)


def save_npz(filename, pmd: PMDArray):
    """Write the reference's .npz layout (notebook cell 8; README.md:31-46)."""
    U = pmd.u
    np.savez(
        filename,
        fov_shape=pmd.shape[1:],
        fov_order=pmd.order,
        U_data=U.data,
        U_indices=U.indices,
        U_indptr=U.indptr,
        U_shape=U.shape,
        U_format=type(U),
        R=pmd.r,
        s=pmd.s,
        Vt=pmd.v,
        mean_img=pmd.mean_img,
        noise_var_img=pmd.var_img,
    )


def load_npz(filename) -> PMDArray:
    """Read the reference's .npz layout (notebook cell 10; README.md:48-62).  ``U_format`` is
    an object entry in reference-written files, so it is never loaded (no unpickling)."""
    with np.load(filename, allow_pickle=False) as data:
        u = scipy.sparse.csr_matrix(
            (data["U_data"], data["U_indices"], data["U_indptr"]), shape=tuple(data["U_shape"])
        ).tocoo()
        v = data["Vt"]
        fov = data["fov_shape"]
        order = str(data["fov_order"].item())
        return PMDArray(
            u, data["R"], data["s"], v, (v.shape[1], int(fov[0]), int(fov[1])), order,
            data["mean_img"], data["noise_var_img"],
        )
