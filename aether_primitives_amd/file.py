"""util::file (reference: src/util/file.rs:12-107): raw native-endian struct files."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check


def count_structs_in_file(path, dtype=np.complex64):               # file.rs:12-25
    n = C.c_size_t()
    check(_lib.load().aeth_file_count_structs(os.fsencode(path), np.dtype(dtype).itemsize, C.byref(n)))
    return n.value


class BinaryReader:                                                 # file.rs:29-76
    def __init__(self, path, dtype=np.complex64):
        self.path, self.dtype, self.pos = os.fsencode(path), np.dtype(dtype), 0
        count_structs_in_file(path, dtype)                          # binary_reader() checks the size first (:30)

    def read(self, into):                                           # fill the whole slice or fail (:46-57)
        assert into.dtype == self.dtype and into.flags["C_CONTIGUOUS"]
        check(_lib.load().aeth_file_read(self.path, self.pos, into.ctypes.data_as(C.c_void_p), into.size,
                                         self.dtype.itemsize))
        self.pos += into.size
        return into

    def read_vec(self, n):                                          # :60-75
        return self.read(np.empty(n, self.dtype))


class BinaryWriter:                                                 # file.rs:83-110
    def __init__(self, path, dtype=np.complex64):
        self.path, self.dtype, self._append = os.fsencode(path), np.dtype(dtype), 0
        check(_lib.load().aeth_file_write(self.path, None, 0, self.dtype.itemsize, 0))   # create / truncate (:84-90)
        self._append = 1

    def write(self, data):                                          # :101-109
        data = np.ascontiguousarray(data, self.dtype)
        check(_lib.load().aeth_file_write(self.path, data.ctypes.data_as(C.c_void_p), data.size, self.dtype.itemsize, 1))


def binary_reader(path, dtype=np.complex64):
    return BinaryReader(path, dtype)


def binary_writer(path, dtype=np.complex64):
    return BinaryWriter(path, dtype)
