"""Host-side mirror of the reference's stage interface.

reference                                   here
---------                                   ----
spring::call_reorder(temp_dir, cp)          call_reorder(temp_dir, cp)         (file contract)
reorder_main<N>: readDnaFile ->             ReorderStage.load_dna / load_synth
  constructdictionary -> reorder ->         .build_dict() .run_chains()
  writetofile                               .finalize() .streams() .emit_dna(tid)
(call_template_functions.cpp:9-63, reorder.h:732-786)
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib


class ReorderError(RuntimeError):
    """Raised where the reference throws std::runtime_error (call_template_functions.cpp:60)."""


@dataclass
class ReorderOpts:
    device: int = -1
    num_chains: int = 0       # K greedy chains (= reference threads); 1 == `-t 1` byte for byte; 0 = auto
    num_thr: int = 1          # number of per-tid output sets (cp.num_thr)
    collect_stats: bool = False
    time_search: bool = False
    force_literal_update: bool = False
    rounds_per_sync: int = 0
    # tuning / experiments (0 = default); the output does not depend on them
    first_shifts: int = 0
    seed_wide: int = 0
    tab_scale: int = 0
    search_wpb: int = 0
    dbg_search_lds: int = 0
    dbg_apply_lds: int = 0
    fused: int = 0            # -1: the two-kernel round; 2: one chain per wavefront everywhere; 3: four per wavefront whatever the chain count
    deep_bins: int = 0        # 1 / -1: force the bin-trimming kernel variant on / off (0 = from the dictionary)
    long_budget: int = 0      # deep pools: compare passes before a search goes to k_long (0 = default, -1 = never)
    devices: tuple = ()       # call_reorder on several GPUs: one pool over these device ordinals (may repeat: host transport)
    mg_host_transport: bool = False
    plan0: tuple = ()         # explicit probe plan (shifts per ordered batch, e.g. (4, 8, 16)); plan1: for a chain whose seed is unmatched
    plan1: tuple = ()
    long_min: int = 0
    long_blocks: int = 0
    entry_flags: int = 0      # -1: deep-bin scans ask the taken bitmap instead of reading the flag in the bin entry (A/B)
    long_split: int = 0       # long searches: chunks of 64 bin entries per part (0 = default 192, -1 = never cut a search into parts)
    debug: bool = False       # stage timings on stderr
    out_writers: int = 0      # call_reorder: threads writing the output files (0 = from the host's thread count)
    alternatives: int = 0     # candidates per match proposal: 1, 2 (a loser takes the next passing read of the bin), 0 = library's choice
    table_mode: int = 0       # 2: dictionary table addressed by the key's minimizer where that applies (experiment; 0 / 1 = by its hash)
    phases: int = 0           # chain schedule: 1 lock-step rounds, 2 two chain groups whose rounds alternate (the output depends on it), 0 = library's choice
    known_absent: int = 0     # four-chain round kernel: chains remember known-absent windows (0 on, -1 off; same results)

    def to_c(self):
        o = _lib.Opts()
        _lib.lib().spring_reorder_default_opts(C.byref(o))
        o.device, o.num_chains, o.num_thr = self.device, self.num_chains, self.num_thr
        o.collect_stats, o.time_search = int(self.collect_stats), int(self.time_search)
        o.force_literal_update, o.rounds_per_sync = int(self.force_literal_update), self.rounds_per_sync
        o.first_shifts, o.seed_wide, o.tab_scale = self.first_shifts, self.seed_wide, self.tab_scale
        o.search_wpb, o.dbg_search_lds, o.dbg_apply_lds = self.search_wpb, self.dbg_search_lds, self.dbg_apply_lds
        o.fused, o.deep_bins = self.fused, self.deep_bins
        o.long_budget = self.long_budget
        if len(self.devices) > 8:
            raise ValueError("at most 8 devices")
        o.table_mode = self.table_mode
        for i, v in enumerate(tuple(self.plan0)[:6]):
            o.plan0[i] = int(v)
        for i, v in enumerate(tuple(self.plan1)[:6]):
            o.plan1[i] = int(v)
        o.long_min, o.long_blocks, o.debug = self.long_min, self.long_blocks, int(self.debug)
        o.long_split, o.entry_flags = self.long_split, self.entry_flags
        o.out_writers = self.out_writers
        o.alternatives = self.alternatives
        o.phases = self.phases
        o.known_absent = self.known_absent
        o.num_devices = len(self.devices)
        for i, d in enumerate(self.devices):
            o.devices[i] = d
        o.mg_host_transport = int(self.mg_host_transport)
        return o


def _chk(rc):
    if rc != 0:
        raise ReorderError("%s (code %d)" % (_lib.lib().spring_reorder_last_error().decode(), rc))


class ReorderStage:
    """One run of the stage on in-memory buffers; device memory is freed by close()."""

    def __init__(self, opts: ReorderOpts = None):
        self.opts = opts or ReorderOpts()
        self._L = _lib.lib()
        self._h = C.c_void_p()
        _chk(self._L.spring_reorder_create(C.byref(self._h), C.byref(self.opts.to_c())))
        self.n = 0
        self.max_readlen = 0

    def close(self):
        if self._h:
            self._L.spring_reorder_destroy(self._h)
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

    # readDnaFile (reorder.h:222-244)
    def load_dna(self, dna: bytes, n: int, max_readlen: int):
        buf = np.frombuffer(dna, dtype=np.uint8)
        _chk(self._L.spring_reorder_load_dna(self._h, buf.ctypes.data if len(buf) else None, len(buf), n,
                                             max_readlen))
        self.n, self.max_readlen = n, max_readlen

    def load_dna_device(self, dptr: int, nbytes: int, n: int, max_readlen: int, fixed_len: bool):
        _chk(self._L.spring_reorder_load_dna_device(self._h, C.c_void_p(dptr), nbytes, n, max_readlen,
                                                    int(fixed_len)))
        self.n, self.max_readlen = n, max_readlen

    def load_synth(self, n, L, G, seed, err_ppm=10000):
        _chk(self._L.spring_reorder_load_synth(self._h, n, L, G, seed, err_ppm))
        self.n, self.max_readlen = n, L

    def load_fastq(self, fastq_1: bytes, fastq_2: bytes = None):
        """FASTQ text -> N split -> packed reads on the device (preprocess.cpp:186-214,:293-304).
        Returns the counts the reference stores in compression_params (num_reads, num_reads_clean, max_readlen)."""
        a = np.frombuffer(fastq_1, dtype=np.uint8)
        b = np.frombuffer(fastq_2, dtype=np.uint8) if fastq_2 is not None else None
        info = _lib.FastqInfo()
        _chk(self._L.spring_reorder_load_fastq(
            self._h, a.ctypes.data if len(a) else None, len(a),
            (b.ctypes.data if len(b) else C.c_void_p(1)) if b is not None else None, len(b) if b is not None else 0,
            C.byref(info)))
        self.n = info.num_reads_clean[0] + info.num_reads_clean[1]
        self.max_readlen = max(int(info.max_readlen), 1)
        return dict(num_reads=list(info.num_reads), num_reads_clean=list(info.num_reads_clean),
                    num_reads_N=list(info.num_reads_N), max_readlen=int(info.max_readlen),
                    ms_device=float(info.ms_device))

    def fastq_N(self, which=0):
        """(input_N.dna bytes, read_order_N.bin array) of input file `which`."""
        nb, cnt = C.c_size_t(), C.c_uint32()
        _chk(self._L.spring_reorder_fastq_N(self._h, which, None, 0, C.byref(nb), None, C.byref(cnt)))
        buf = np.zeros(max(nb.value, 1), np.uint8)
        order = np.zeros(max(cnt.value, 1), np.uint32)
        _chk(self._L.spring_reorder_fastq_N(self._h, which, buf.ctypes.data, nb.value, C.byref(nb), order.ctypes.data,
                                            C.byref(cnt)))
        return buf[:nb.value].tobytes(), order[:cnt.value]

    def build_dict(self):  # constructdictionary (bitset_util.h:74-221)
        _chk(self._L.spring_reorder_build_dict(self._h))

    def run_chains(self):  # reorder() (reorder.h:320-641)
        _chk(self._L.spring_reorder_run_chains(self._h))

    def auto_chains(self):
        """-> (chains, deep): what run_chains() uses for num_chains = 0 (after build_dict)."""
        k, d = C.c_uint32(0), C.c_int32(0)
        _chk(self._L.spring_reorder_auto_chains(self._h, C.byref(k), C.byref(d)))
        return k.value, bool(d.value)

    def finalize(self):
        _chk(self._L.spring_reorder_finalize(self._h))

    def run(self):
        self.build_dict()
        self.run_chains()
        self.finalize()
        return self

    def stats(self):
        s = _lib.Stats()
        _chk(self._L.spring_reorder_get_stats(self._h, C.byref(s)))
        return s.asdict()

    def streams(self):
        """-> dict(order, rc, flag, pos, rlen, order_s, tid_off, tid_off_s): the contents of
        read_order.bin.<tid>, read_rev.txt.<tid>, tempflag.txt.<tid>, temppos.txt.<tid>,
        read_lengths.bin.<tid> (concatenated in tid order) and read_order.bin.singleton."""
        st = self.stats()
        nm, ns, T = st["n_matched"], st["n_single"], self.opts.num_thr
        out = dict(order=np.zeros(max(nm, 1), np.uint32), rc=np.zeros(max(nm, 1), np.uint8),
                   flag=np.zeros(max(nm, 1), np.uint8), pos=np.zeros(max(nm, 1), np.int64),
                   rlen=np.zeros(max(nm, 1), np.uint16), order_s=np.zeros(max(ns, 1), np.uint32),
                   tid_off=np.zeros(T + 1, np.uint64), tid_off_s=np.zeros(T + 1, np.uint64))
        _chk(self._L.spring_reorder_download(self._h, *[out[k].ctypes.data for k in (
            "order", "rc", "flag", "pos", "rlen", "order_s", "tid_off", "tid_off_s")]))
        for k in ("order", "rc", "flag", "pos", "rlen"):
            out[k] = out[k][:nm]
        out["order_s"] = out["order_s"][:ns]
        # inside tid t: where the records of the second chain group's chains begin (a rank of a pool that ran two groups;
        # = tid_off[t + 1] otherwise) -- pool.merge_rank_streams
        out["tid_mid"], out["tid_mid_s"] = np.zeros(T, np.uint64), np.zeros(T, np.uint64)
        _chk(self._L.spring_reorder_tid_split(self._h, out["tid_mid"].ctypes.data, out["tid_mid_s"].ctypes.data))
        out["stats"] = st
        return out

    def emit_dna(self, tid: int) -> bytes:
        """temp.dna.<tid> (tid >= 0) / temp.dna.singleton (tid = -1), built on the device."""
        nb = C.c_size_t()
        _chk(self._L.spring_reorder_emit_dna(self._h, tid, None, 0, C.byref(nb)))
        buf = np.zeros(max(nb.value, 1), np.uint8)
        _chk(self._L.spring_reorder_emit_dna(self._h, tid, buf.ctypes.data, nb.value, C.byref(nb)))
        return buf[:nb.value].tobytes()

    # test hooks
    def dict_lookup(self, which, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        sizes = np.zeros(max(len(keys), 1), np.uint32)
        ids = np.zeros(max(self.n, 1) * 2, np.uint32)
        _chk(self._L.spring_reorder_dict_lookup(self._h, which, keys.ctypes.data, len(keys), sizes.ctypes.data,
                                                ids.ctypes.data, len(ids)))
        return sizes[:len(keys)], ids

    def download_reads(self):
        W = (2 * self.max_readlen - 1) // 64 + 1
        limbs = np.zeros((max(self.n, 1), W), np.uint64)
        ln = np.zeros(max(self.n, 1), np.uint16)
        _chk(self._L.spring_reorder_download_reads(self._h, limbs.ctypes.data, ln.ctypes.data))
        return limbs[:self.n], ln[:self.n]

    def download_dna(self) -> bytes:
        nb = self._L.spring_synth_dna_bytes(self.n, self.max_readlen)  # upper bound (fixed-length records)
        buf = np.zeros(max(nb, 1), np.uint8)
        _chk(self._L.spring_reorder_download_dna(self._h, buf.ctypes.data, nb))
        p = 0
        for _ in range(self.n):  # walk the records to find the end of the stream
            p += 2 + ((int(buf[p]) | (int(buf[p + 1]) << 8)) + 3) // 4
        return buf[:p].tobytes()


def reorder_dna(dna: bytes, n: int, max_readlen: int, opts: ReorderOpts = None):
    """Whole stage on a host .dna record stream -> streams() dict."""
    with ReorderStage(opts) as s:
        s.load_dna(dna, n, max_readlen)
        return s.run().streams()


def synth_dna_host(n, L, G, seed, err_ppm=10000) -> bytes:
    """Host version of the counter-based synthetic generator (identical bytes to load_synth)."""
    L_ = _lib.lib()
    nb = L_.spring_synth_dna_bytes(n, L)
    buf = np.zeros(max(nb, 1), np.uint8)
    _chk(L_.spring_synth_dna_host(buf.ctypes.data, n, L, G, seed, err_ppm))
    return buf[:nb].tobytes()


SYNTH_REPEATS = 0x80000000  # OR into err_ppm (include/spring_reorder.h)
SYNTH_PAIRED = 0x40000000
SYNTH_GENOMIC = 0x20000000  # genome with interspersed repeat families, tandem repeats and low-complexity runs (synth_common.h)


def synth_genome_host(G, seed, flags=0) -> bytes:
    L_ = _lib.lib()
    buf = np.zeros(max(G, 1), np.uint8)
    _chk(L_.spring_synth_genome_host(buf.ctypes.data, G, seed, flags))
    return buf[:G].tobytes()


class CompressionParams:
    """The compression_params fields the stage reads (reference util.h:30-51, reorder.h:747-763)."""

    def __init__(self, max_readlen, num_reads_clean, num_thr=1, paired_end=False):
        self.max_readlen = max_readlen
        self.num_reads_clean = list(num_reads_clean) + [0] * (2 - len(num_reads_clean))
        self.num_thr = num_thr
        self.paired_end = paired_end


def call_reorder(temp_dir: str, cp: CompressionParams, opts: ReorderOpts = None):
    """spring::call_reorder(temp_dir, cp): consumes temp_dir/input_clean_{1,2}.dna, writes the
    per-tid + singleton files of reorder.h:355-368,:643-730.  Raises ReorderError on failure."""
    bitset_size = (2 * cp.max_readlen - 1) // 64 * 64 + 64
    if cp.max_readlen <= 0 or bitset_size > 1024:
        raise ReorderError("Wrong bitset size.")
    o = (opts or ReorderOpts(num_thr=cp.num_thr)).to_c()
    _chk(_lib.lib().spring_reorder_run(temp_dir.encode(), cp.max_readlen, cp.num_thr, int(cp.paired_end),
                                       cp.num_reads_clean[0], cp.num_reads_clean[1], C.byref(o)))
