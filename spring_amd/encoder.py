"""Host-side mirror of the encoder stage (reference src/encoder.h:572-633 encoder_main<N>) on top of
the C ABI in include/spring_encoder.h.  All compute is in the HIP library; no CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from .reorder import ReorderError, ReorderStage


def _chk(rc):
    if rc != 0:
        raise ReorderError("%s (code %d)" % (_lib.lib().spring_reorder_last_error().decode(), rc))


class EncoderStage:
    """encode(): consensus + singleton alignment + noise streams of the contigs a finalized
    ReorderStage holds in HBM.  Streams are fetched with streams()."""

    def __init__(self, device: int = -1):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        _chk(self._L.spring_encoder_create(device, C.byref(self._h)))
        self.info = None
        self.num_thr = 0

    def close(self):
        if self._h:
            self._L.spring_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, reorder: ReorderStage, dnaN: bytes = b"", order_N=None):
        """reorder must be finalized; dnaN / order_N are the images of input_N.dna / read_order_N.bin."""
        order_N = np.ascontiguousarray(order_N if order_N is not None else [], dtype=np.uint32)
        buf = np.frombuffer(dnaN, dtype=np.uint8)
        info = _lib.EncoderInfo()
        _chk(self._L.spring_encoder_encode_reorder(
            self._h, reorder._h, buf.ctypes.data if len(buf) else None, len(buf),
            order_N.ctypes.data if len(order_N) else None, len(order_N), C.byref(info)))
        self.info = info.asdict()
        self.num_thr = reorder.opts.num_thr
        return self.info

    def streams(self):
        """-> dict of the output streams (seq, seq_len_tid, pos, noise, noisepos, order, rlen, rc, unaligned, ...)."""
        i = self.info
        seq = np.zeros(max(i["seq_len"], 1), np.uint8)
        seq_len_tid = np.zeros(max(self.num_thr, 1), np.uint64)
        pos = np.zeros(max(i["n_aligned"], 1), np.uint64)
        noise = np.zeros(max(i["noise_bytes"], 1), np.uint8)
        noisepos = np.zeros(max(i["n_noisepos"], 1), np.uint16)
        order = np.zeros(max(i["n_total"], 1), np.uint32)
        rlen = np.zeros(max(i["n_total"], 1), np.uint16)
        rc = np.zeros(max(i["n_aligned"], 1), np.uint8)
        un = np.zeros(max(i["unaligned_bytes"], 1), np.uint8)
        _chk(self._L.spring_encoder_download(self._h, seq.ctypes.data, seq_len_tid.ctypes.data, pos.ctypes.data,
                                             noise.ctypes.data, noisepos.ctypes.data, order.ctypes.data,
                                             rlen.ctypes.data, rc.ctypes.data, un.ctypes.data))
        return dict(seq=seq[:i["seq_len"]].tobytes(), seq_len_tid=seq_len_tid[:self.num_thr],
                    pos=pos[:i["n_aligned"]], noise=noise[:i["noise_bytes"]].tobytes(),
                    noisepos=noisepos[:i["n_noisepos"]], order=order[:i["n_total"]], rlen=rlen[:i["n_total"]],
                    rc=rc[:i["n_aligned"]], unaligned=un[:i["unaligned_bytes"]].tobytes(),
                    len_unaligned=i["len_unaligned"], matched_s=i["matched_s"], matched_N=i["matched_N"],
                    num_contigs=i["num_contigs"])

    def seq_packed(self):
        """pack_compress_seq without BSC -> (packed bytes tid-major, [tail string per tid])."""
        sl = np.zeros(max(self.num_thr, 1), np.uint64)
        _chk(self._L.spring_encoder_download(self._h, None, sl.ctypes.data, None, None, None, None, None, None, None))
        total = int(sum(int(x) // 4 for x in sl[:self.num_thr]))
        packed = np.zeros(max(total, 1), np.uint8)
        tail = np.zeros(4 * max(self.num_thr, 1), np.uint8)
        _chk(self._L.spring_encoder_download_seq_packed(self._h, packed.ctypes.data, tail.ctypes.data))
        tails = [tail[4 * t:4 * t + int(sl[t]) % 4].tobytes().decode() for t in range(self.num_thr)]
        return packed[:total].tobytes(), tails


def call_encoder(temp_dir: str, cp, num_reads: int, device: int = -1):
    """spring::call_encoder(temp_dir, cp) (reference call_template_functions.cpp:65-142) MINUS its BSC step:
    consumes the files a reorder stage left in temp_dir and writes encoder_main's outputs, but the packed consensus
    stays as read_seq.bin.<tid>.tmp (+ .tail): the caller runs BSC_compress(.tmp -> .bsc) and removes the .tmp, as
    pack_compress_seq does (encoder.cpp:146-150).  cp is a reorder.CompressionParams,
    num_reads = cp.num_reads of the reference (clean + N reads).  -> info dict."""
    bitset_size = (3 * cp.max_readlen - 1) // 64 * 64 + 64
    if cp.max_readlen <= 0 or bitset_size > 1536:
        raise ReorderError("Wrong bitset size.")
    info = _lib.EncoderInfo()
    n_clean = cp.num_reads_clean[0] + (cp.num_reads_clean[1] if cp.paired_end else 0)
    _chk(_lib.lib().spring_encoder_run(temp_dir.encode(), cp.max_readlen, cp.num_thr, num_reads, n_clean, device,
                                       C.byref(info)))
    return info.asdict()


def call_reorder_encoder(temp_dir: str, cp, num_reads: int, opts=None):
    """call_reorder + call_encoder back to back with the intermediate streams kept in HBM (spring.cpp:150-160)."""
    from .reorder import ReorderOpts
    if cp.max_readlen <= 0 or (2 * cp.max_readlen - 1) // 64 * 64 + 64 > 1024:
        raise ReorderError("Wrong bitset size.")
    o = (opts or ReorderOpts(num_thr=cp.num_thr)).to_c()
    info = _lib.EncoderInfo()
    _chk(_lib.lib().spring_reorder_encode_run(temp_dir.encode(), cp.max_readlen, cp.num_thr, int(cp.paired_end),
                                              cp.num_reads_clean[0], cp.num_reads_clean[1], num_reads, C.byref(o),
                                              C.byref(info)))
    return info.asdict()
