"""
Dataset boundary of the decomposition: anything with ``.shape == (T, d1, d2)`` and
``obj[list_of_frames] -> (n, d1, d2) ndarray`` can be decomposed (a plain ``np.ndarray``
qualifies).  ``lazy_data_loader`` mirrors the reference's abstract base
(/root/reference/localmd/dataset.py:7-128): subclasses provide ``dtype``, ``shape`` and
``_compute_at_indices``; ``__getitem__`` normalises int / list / ndarray / slice / range
frame keys and applies optional spatial keys afterwards.
"""
from abc import ABC, abstractmethod
from typing import Tuple, Union

import numpy as np


class lazy_data_loader(ABC):
    @property
    @abstractmethod
    def dtype(self) -> str:
        """Data type of the frames."""

    @property
    @abstractmethod
    def shape(self) -> Tuple[int, int, int]:
        """(n_frames, dims_x, dims_y)."""

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def _normalise_frame_key(self, key):
        n_frames = self.shape[0]
        if isinstance(key, np.ndarray):
            return key.tolist()
        if isinstance(key, (list, int)):
            return key
        if isinstance(key, np.integer):
            return key.item()
        if isinstance(key, (slice, range)):
            for name, bound in (("start", key.start), ("stop", key.stop)):
                if bound is not None and bound > n_frames:
                    raise IndexError(
                        f"Cannot index beyond `n_frames`.\n"
                        f"Desired frame {name} index of <{bound}> lies beyond `n_frames` <{n_frames}>"
                    )
            return slice(key.start, key.stop, 1 if key.step is None else key.step)
        raise IndexError(f"Invalid indexing method, you have passed a: <{type(key)}>")

    def __getitem__(self, item):
        if isinstance(item, tuple):
            if len(item) > len(self.shape):
                raise IndexError(
                    f"Cannot index more dimensions than exist in the array. "
                    f"You have tried to index with <{len(item)}> dimensions, "
                    f"only <{len(self.shape)}> dimensions exist in the array"
                )
            frame_key = item[0]
        else:
            frame_key = item
        frames = self._compute_at_indices(self._normalise_frame_key(frame_key))
        if frames.ndim < len(self.shape):
            frames = frames[None, ...]
        if isinstance(item, tuple):
            if len(item) == 2:
                frames = frames[:, item[1]]
            elif len(item) == 3:
                frames = frames[:, item[1], item[2]]
        return frames.squeeze()

    @abstractmethod
    def _compute_at_indices(self, indices: Union[list, int, slice]) -> np.ndarray:
        """Return the frames selected by ``indices`` (an int, a list of ints or a slice)."""


class ArrayDataset(lazy_data_loader):
    """In-memory (T, d1, d2) array behind the lazy_data_loader interface."""

    def __init__(self, array: np.ndarray):
        if array.ndim != 3:
            raise ValueError("expected a (T, d1, d2) array")
        self._array = array

    @property
    def dtype(self):
        return self._array.dtype

    @property
    def shape(self):
        return tuple(self._array.shape)

    def _compute_at_indices(self, indices):
        return np.asarray(self._array[indices])


class TiffArray(lazy_data_loader):
    """Multipage-TIFF reader (reference: dataset.py:131-181).  Uses ``tifffile`` when it is
    importable, otherwise the built-in reader (localmd_amd/_minitiff.py: grayscale TIFF / BigTIFF, strips or tiles,
    uncompressed / LZW / deflate / PackBits, predictor 2, ImageJ single-IFD hyperstacks; anything else raises
    NotImplementedError naming the unsupported tag).  Every read opens its own file handle, so the ingestion
    path may call ``__getitem__`` from several threads (``thread_safe``)."""

    thread_safe = True

    def __init__(self, filename):
        self.filename = filename
        self._reader = None
        self._shape = None

    def _open(self):
        if self._reader is None:
            try:
                import tifffile  # noqa: F401

                self._reader = "tifffile"
            except ImportError:
                from ._minitiff import MiniTiff

                self._reader = MiniTiff(self.filename)
        return self._reader

    @property
    def dtype(self):
        return np.float32

    @property
    def shape(self):
        if self._shape is None:
            reader = self._open()
            if reader == "tifffile":
                import tifffile

                with tifffile.TiffFile(self.filename) as tf:
                    n = len(tf.pages)
                    x, y = tf.pages[0].shape
                self._shape = (n, x, y)
            else:
                self._shape = reader.shape
        return self._shape

    def _compute_at_indices(self, indices):
        if isinstance(indices, int):
            keys = [indices]
        elif isinstance(indices, list):
            keys = indices
        else:
            keys = list(range(indices.start or 0, indices.stop or self.shape[0], indices.step or 1))
        reader = self._open()
        if reader == "tifffile":
            import tifffile

            data = tifffile.imread(self.filename, key=keys).squeeze()
        else:
            data = reader.read(keys).squeeze()
        return data.astype(self.dtype)
