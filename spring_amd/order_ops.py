"""read_order.bin consumers on the GPU (mirrors of generate_order_se / generate_order_pe,
reference src/reorder_compress_quality_id.cpp:101-125, and correct_order, src/encoder.cpp:177-222)."""
import ctypes as C

import numpy as np

from . import _lib
from .reorder import _chk


def generate_order_se(order):
    """order_array[order[i]] = i."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    out = np.zeros(max(len(order), 1), np.uint32)
    ms = C.c_double()
    _chk(_lib.lib().spring_order_invert_se(order.ctypes.data, len(order), out.ctypes.data, C.byref(ms)))
    return out[:len(order)], ms.value


def generate_order_pe(order):
    """for i: if order[i] < n/2: order_array[order[i]] = pos_after_reordering++."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    half = len(order) // 2
    out = np.zeros(max(half, 1), np.uint32)
    ms = C.c_double()
    _chk(_lib.lib().spring_order_invert_pe(order.ctypes.data, len(order), out.ctypes.data, C.byref(ms)))
    return out[:half], ms.value


def correct_order(order, order_N, n_clean):
    """Shifts indices into the clean-read array past the N reads (returns a new array)."""
    order = np.array(order, dtype=np.uint32, copy=True)
    order_N = np.ascontiguousarray(order_N, dtype=np.uint32)
    ms = C.c_double()
    _chk(_lib.lib().spring_order_correct(order.ctypes.data, len(order), order_N.ctypes.data if len(order_N) else None,
                                         len(order_N), n_clean, C.byref(ms)))
    return order, ms.value


def pe_encode(order):
    """pe_encode (pe_encode.cpp:24-84): reordered position -> position in the decompressed paired files."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    out = np.zeros(max(len(order), 1), np.uint32)
    ms = C.c_double()
    _chk(_lib.lib().spring_order_pe_encode(order.ctypes.data, len(order), out.ctypes.data, C.byref(ms)))
    return out[:len(order)], ms.value


def fastq_reorder(fastq: bytes, order):
    """the 4-line records of `fastq` in the order order[0], order[1], ... -> (bytes, kernel ms)."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    buf = np.frombuffer(fastq, dtype=np.uint8)
    L = _lib.lib()
    need, ms = C.c_size_t(), C.c_double()
    _chk(L.spring_fastq_reorder(buf.ctypes.data if len(buf) else None, len(buf), order.ctypes.data if len(order) else None,
                                len(order), None, 0, C.byref(need), None))
    out = np.zeros(max(need.value, 1), np.uint8)
    _chk(L.spring_fastq_reorder(buf.ctypes.data if len(buf) else None, len(buf), order.ctypes.data if len(order) else None,
                                len(order), out.ctypes.data, need.value, C.byref(need), C.byref(ms)))
    return out[:need.value].tobytes(), ms.value
