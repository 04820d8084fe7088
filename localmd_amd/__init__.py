"""
localmd_amd -- MI355X (gfx950) implementation of localmd's blockwise-PMD hot path.

Public surface mirrors /root/reference/localmd/__init__.py:1-7.
"""
from .decomposition import localmd_decomposition, compute_lowrank_factorized_svd, projected_svd
from .pmdarray import PMDArray, save_npz, load_npz
from .dataset import TiffArray, lazy_data_loader, ArrayDataset

PMDDataset = lazy_data_loader  # the name the reference's README uses (README.md:67)

__all__ = [
    "localmd_decomposition", "compute_lowrank_factorized_svd", "projected_svd", "PMDArray", "TiffArray",
    "lazy_data_loader", "PMDDataset", "ArrayDataset", "save_npz", "load_npz",
]
