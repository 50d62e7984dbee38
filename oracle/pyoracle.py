"""ctypes binding of the TEST oracle (oracle/liboracle_reorder.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (spring_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


class OrcStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "unmatched", "search_calls", "probes", "keyok", "cands", "hits", "updates",
        "iterations", "rounds", "lost")]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class OrcOut(C.Structure):
    _fields_ = [
        ("order", C.c_void_p), ("rc", C.c_void_p), ("flag", C.c_void_p), ("pos", C.c_void_p),
        ("rlen", C.c_void_p), ("order_s", C.c_void_p), ("tid_off", C.c_void_p),
        ("tid_off_s", C.c_void_p), ("n_matched", C.c_uint64), ("n_single", C.c_uint64)]


def build():
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_reorder.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_limbs.restype = C.c_int
        L.orc_load_dna.restype = C.c_int64
        L.orc_load_dna.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_reorder_serial.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                         C.POINTER(OrcOut), C.POINTER(OrcStats)]
        L.orc_reorder_rounds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                         C.POINTER(OrcOut), C.POINTER(OrcStats)]
        L.orc_reorder_rounds_alt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p]
        L.orc_reorder_rounds_ph.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                            C.c_void_p, C.c_void_p]
        L.orc_reorder_rounds_ph_alt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p]
        L.orc_phase_split.restype = C.c_uint32
        L.orc_phase_split.argtypes = [C.c_uint32]
        L.orc_reorder_omp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                      C.POINTER(OrcOut), C.POINTER(OrcStats)]
        L.orc_string_to_bits.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_string_to_bits.restype = None
        L.orc_bits_to_string.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        L.orc_bits_to_string.restype = None
        L.orc_reverse_complement.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_reverse_complement.restype = None
        L.orc_pack_read.restype = C.c_size_t
        L.orc_pack_read.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.orc_updaterefcount.restype = None
        L.orc_updaterefcount.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_write_dna_stream.restype = C.c_size_t
        L.orc_write_dna_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_uint64, C.c_void_p]
        L.orc_build_dict.restype = C.c_uint32
        L.orc_build_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_bin_remove.restype = C.c_int64
        L.orc_bin_remove.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64]
        L.orc_bin_live.restype = C.c_int64
        L.orc_bin_live.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_hamming_range.restype = C.c_int
        L.orc_hamming_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_updaterefcount.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_dict_windows.argtypes = [C.c_int, C.POINTER(C.c_int * 2), C.POINTER(C.c_int * 2)]
        _LIB = L
    return _LIB


def ref_lib():
    """The real reference bitset_util build (oracle/_ref), or None if absent."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_bitset.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_build_dict.restype = C.c_int
        R.ref_build_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        R.ref_bin_remove.restype = C.c_int64
        R.ref_bin_remove.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64]
        R.ref_bin_live.restype = C.c_int64
        R.ref_bin_live.argtypes = [C.c_void_p, C.c_uint32]
        R.ref_mask_hamming.restype = C.c_int
        R.ref_mask_hamming.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _REF = R
    return _REF


_UNITS = None


class OrcShadow(C.Structure):
    """orc_shadow (reorder_oracle.h): C function pointers, filled straight from oracle/_ref/libref_units.so."""
    _fields_ = [("user", C.c_void_p), ("claim_first", C.c_void_p), ("remove", C.c_void_p), ("search", C.c_void_p),
                ("update", C.c_void_p), ("pick_seed", C.c_void_p)]


def ref_units():
    """oracle/_ref/libref_units.so: the REAL updaterefcount / search_match / readDnaFile / setglobalarrays
    (reorder.h:33-318), util.cpp helpers and encoder.cpp units compiled by line range (oracle/Makefile), or None."""
    global _UNITS
    if _UNITS is None:
        path = os.path.join(_HERE, "_ref", "libref_units.so")
        if not os.path.exists(path):
            return None
        U = C.CDLL(path)
        U.ref_u_updaterefcount.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]
        U.ref_u_chartobitset.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        U.ref_u_bitsettostring.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        U.ref_u_readDnaFile.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        U.ref_u_reverse_complement.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        U.ref_u_reverse_complement.restype = None
        U.ref_u_write_dna.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_int]
        U.ref_u_read_dna.restype = C.c_long
        U.ref_u_read_dna.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_char_p, C.c_long]
        U.ref_u_read_fastq.restype = C.c_long
        U.ref_u_read_fastq.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p, C.c_long, C.POINTER(C.c_long)]
        U.ref_shadow_create.restype = C.c_void_p
        U.ref_shadow_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_char_p, C.c_int]
        U.ref_shadow_destroy.argtypes = [C.c_void_p]
        U.ref_shadow_destroy.restype = None
        U.ref_shadow_remove.argtypes = [C.c_void_p, C.c_uint32]
        U.ref_shadow_set_remaining.argtypes = [C.c_void_p, C.c_void_p]
        U.ref_shadow_set_remaining.restype = None
        U.ref_shadow_get_remaining.argtypes = [C.c_void_p, C.c_void_p]
        U.ref_shadow_get_remaining.restype = None
        U.ref_shadow_search_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        U.ref_shadow_search_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_int), C.POINTER(C.c_int)]
        U.ref_u_contig.restype = C.c_long
        U.ref_u_contig.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_char_p,
                                   C.POINTER(C.c_uint64), C.c_void_p, C.c_long, C.c_void_p]
        U.ref_u_correct_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p]
        U.ref_u_enc_bits3_roundtrip.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p]
        U.ref_u_readsingletons.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _UNITS = U
    return _UNITS


def reorder_serial_shadow(read, ln, L, basedir, num_thr=2):
    """orc_reorder_serial with every search_match / updaterefcount / bin removal / seed pick cross-checked against the
    REAL reference functions advancing a mirrored state.  -> (streams dict, stats, mm[10])."""
    U = ref_units()
    Lb = lib()
    n = len(ln)
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    sh_obj = U.ref_shadow_create(read.ctypes.data, ln.ctypes.data, n, L, basedir.encode(), num_thr)
    assert sh_obj
    sh = OrcShadow()
    sh.user = sh_obj
    for name in ("claim_first", "remove", "search", "update", "pick_seed"):
        setattr(sh, name, C.cast(getattr(U, "ref_shadow_" + name), C.c_void_p).value)
    o, arrs = _alloc_out(max(n, 1), 1)
    st = OrcStats()
    mm = np.zeros(10, np.uint64)
    Lb.orc_reorder_serial_shadow.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(OrcOut),
                                             C.POINTER(OrcStats), C.POINTER(OrcShadow), C.c_void_p]
    rc = Lb.orc_reorder_serial_shadow(read.ctypes.data, ln.ctypes.data, n, L, C.byref(o), C.byref(st), C.byref(sh),
                                      mm.ctypes.data)
    U.ref_shadow_destroy(sh_obj)
    assert rc == 0
    return _finish(o, arrs, st), st.asdict(), mm


def ref_order_bin():
    """oracle/_ref/ref_order: the real pe_encode.cpp + generate_order_se/pe behind a command line, or None."""
    path = os.path.join(_HERE, "_ref", "ref_order")
    return path if os.path.exists(path) else None


def ref_order(mode, order):
    """Runs the REAL reference function on a read_order.bin image: mode 'se' / 'pe' -> order_array,
    'pe_encode' -> the rewritten read_order.bin (pe_encode.cpp:24-84, reorder_compress_quality_id.cpp:101-125)."""
    import tempfile
    order = np.ascontiguousarray(order, dtype=np.uint32)
    with tempfile.TemporaryDirectory() as d:
        order.tofile(os.path.join(d, "read_order.bin"))
        subprocess.run([ref_order_bin(), mode, d, str(len(order))], check=True)
        out = os.path.join(d, "read_order.bin" if mode == "pe_encode" else "order_array.bin")
        return np.fromfile(out, dtype=np.uint32)


def ref_bsc_bin():
    path = os.path.join(_HERE, "_ref", "ref_bsc")
    return path if os.path.exists(path) else None


def limbs(L):
    return (2 * L - 1) // 64 + 1


def dict_windows(L):
    s = (C.c_int * 2)()
    e = (C.c_int * 2)()
    lib().orc_dict_windows(L, C.byref(s), C.byref(e))
    return list(s), list(e)


def load_dna(dna: bytes, n: int, L: int):
    """readDnaFile: -> (read limbs [n,W] u64, lengths [n] u16)."""
    W = limbs(L)
    read = np.zeros((max(n, 1), W), dtype=np.uint64)
    ln = np.zeros(max(n, 1), dtype=np.uint16)
    buf = np.frombuffer(dna, dtype=np.uint8)
    used = lib().orc_load_dna(buf.ctypes.data if len(buf) else None, len(buf), n, L,
                              read.ctypes.data, ln.ctypes.data)
    if used < 0:
        raise ValueError("malformed .dna stream")
    return read[:n], ln[:n]


def _alloc_out(n, num_thr):
    m = max(n, 1)
    arrs = dict(order=np.zeros(m, np.uint32), rc=np.zeros(m, np.uint8), flag=np.zeros(m, np.uint8),
                pos=np.zeros(m, np.int64), rlen=np.zeros(m, np.uint16), order_s=np.zeros(m, np.uint32),
                tid_off=np.zeros(num_thr + 1, np.uint64), tid_off_s=np.zeros(num_thr + 1, np.uint64))
    o = OrcOut()
    for k, a in arrs.items():
        setattr(o, k, a.ctypes.data)
    return o, arrs


def _finish(o, arrs, st):
    nm, ns = int(o.n_matched), int(o.n_single)
    return dict(order=arrs["order"][:nm].copy(), rc=arrs["rc"][:nm].copy(), flag=arrs["flag"][:nm].copy(),
                pos=arrs["pos"][:nm].copy(), rlen=arrs["rlen"][:nm].copy(),
                order_s=arrs["order_s"][:ns].copy(), tid_off=arrs["tid_off"].copy(),
                tid_off_s=arrs["tid_off_s"].copy(), stats=st.asdict())


def reorder_serial(read, ln, L):
    """Literal `-t 1` restatement of reorder_main<N>()."""
    n = len(ln)
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    o, arrs = _alloc_out(n, 1)
    st = OrcStats()
    rc = lib().orc_reorder_serial(read.ctypes.data, ln.ctypes.data, n, L, C.byref(o), C.byref(st))
    assert rc == 0
    return _finish(o, arrs, st)


def reorder_rounds(read, ln, L, num_chains, num_thr=1, alternatives=1):
    """Deterministic K-chain lock-step schedule (the spec the GPU path follows); alternatives = candidates per match
    proposal, resolved in as many passes (1 = a loser waits for the next round)."""
    n = len(ln)
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    o, arrs = _alloc_out(n, num_thr)
    st = OrcStats()
    rc = lib().orc_reorder_rounds_alt(read.ctypes.data, ln.ctypes.data, n, L, num_chains, num_thr, alternatives,
                                      C.byref(o), C.byref(st))
    assert rc == 0
    return _finish(o, arrs, st)


def reorder_rounds_ph(read, ln, L, num_chains, num_thr=1, alternatives=1):
    """The schedule with two chain groups whose rounds alternate (ReorderOpts.phases = 2)."""
    n = len(ln)
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    o, arrs = _alloc_out(n, num_thr)
    st = OrcStats()
    rc = lib().orc_reorder_rounds_ph_alt(read.ctypes.data, ln.ctypes.data, n, L, num_chains, num_thr, alternatives,
                                         C.byref(o), C.byref(st))
    assert rc == 0, "orc_reorder_rounds_ph: needs num_chains >= 4096 and n >= max(8192, num_chains)"
    return _finish(o, arrs, st)


def check_contigs(read, ln, L, res, reference_update=False):
    """Replay check of a reorder output `res` (streams() / reorder_*() dict) -> dict(contigs, matches, bad, first_bad): every
    matched record must be a match the reference's search_match accepts on the consensus its contig had built by then.
    reference_update: the consensus is kept by the REFERENCE'S OWN updaterefcount<N> (oracle/_ref/libref_units.so) instead of
    the restatement (slower: strings, heap); raises if that library is not built."""
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    order = np.ascontiguousarray(res["order"], dtype=np.uint32)
    rc = np.ascontiguousarray(res["rc"], dtype=np.uint8)
    flag = np.ascontiguousarray(res["flag"], dtype=np.uint8)
    pos = np.ascontiguousarray(res["pos"], dtype=np.int64)
    toff = np.ascontiguousarray(res["tid_off"], dtype=np.uint64)
    out = (C.c_uint64 * 4)()
    upd = None
    if reference_update:
        U = ref_units()
        if U is None:
            raise RuntimeError("oracle/_ref/libref_units.so is not built (make -C oracle ref, needs the reference sources)")
        upd = C.cast(U.ref_u_updaterefcount, C.c_void_p)
    f = lib().orc_check_contigs_upd
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rcode = f(read.ctypes.data, ln.ctypes.data, len(ln), L, order.ctypes.data, rc.ctypes.data, flag.ctypes.data, pos.ctypes.data,
              len(order), toff.ctypes.data, len(toff) - 1, out, upd)
    assert rcode == 0
    return dict(contigs=int(out[0]), matches=int(out[1]), bad=int(out[2]), first_bad=int(out[3]))


def reorder_omp(read, ln, L, num_threads):
    """CPU-baseline port: free-running OpenMP threads (non-deterministic for T > 1, like the reference)."""
    n = len(ln)
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    o, arrs = _alloc_out(n, num_threads)
    st = OrcStats()
    rc = lib().orc_reorder_omp(read.ctypes.data, ln.ctypes.data, n, L, num_threads, C.byref(o), C.byref(st))
    assert rc == 0
    return _finish(o, arrs, st)


def last_omp_phases():
    """(dictionary seconds, chain seconds) of the last reorder_omp call."""
    out = (C.c_double * 2)()
    lib().orc_last_omp_phases(out)
    return float(out[0]), float(out[1])


def write_dna_stream(read, ln, L, order, rc=None):
    """writetofile(): bytes of temp.dna.<tid> (rc given) or temp.dna.singleton (rc None)."""
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    cnt = len(order)
    dst = np.zeros(cnt * (2 + (L + 3) // 4) + 8, dtype=np.uint8)
    rcp = None
    if rc is not None:
        rc = np.ascontiguousarray(rc, dtype=np.uint8)
        rcp = rc.ctypes.data
    nb = lib().orc_write_dna_stream(read.ctypes.data, ln.ctypes.data, L, order.ctypes.data, rcp, cnt,
                                    dst.ctypes.data)
    return dst[:nb].tobytes()


def build_dict(read, ln, L, which):
    n = len(ln)
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    keys = np.zeros(max(n, 1), np.uint64)
    sp = np.zeros(n + 2, np.uint32)
    ids = np.zeros(max(n, 1), np.uint32)
    dn = C.c_uint32()
    nk = lib().orc_build_dict(read.ctypes.data, ln.ctypes.data, n, L, which, keys.ctypes.data,
                              sp.ctypes.data, ids.ctypes.data, C.byref(dn))
    return keys[:nk].copy(), sp[:nk + 1].copy(), ids[:dn.value].copy()


def generate_order_se(order):
    order = np.ascontiguousarray(order, dtype=np.uint32)
    out = np.zeros(max(len(order), 1), np.uint32)
    lib().orc_generate_order_se(C.c_void_p(order.ctypes.data), C.c_uint32(len(order)), C.c_void_p(out.ctypes.data))
    return out[:len(order)]


def generate_order_pe(order):
    order = np.ascontiguousarray(order, dtype=np.uint32)
    out = np.zeros(max(len(order) // 2, 1), np.uint32)
    lib().orc_generate_order_pe(C.c_void_p(order.ctypes.data), C.c_uint32(len(order)), C.c_void_p(out.ctypes.data))
    return out[:len(order) // 2]


def pe_encode(order):
    order = np.ascontiguousarray(order, dtype=np.uint32)
    out = np.zeros(max(len(order), 1), np.uint32)
    lib().orc_pe_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib().orc_pe_encode(order.ctypes.data, len(order), out.ctypes.data)
    return out[:len(order)]


def correct_order(order, order_N, n_clean):
    order = np.array(order, dtype=np.uint32, copy=True)
    order_N = np.ascontiguousarray(order_N, dtype=np.uint32)
    lib().orc_correct_order(C.c_void_p(order.ctypes.data), C.c_uint64(len(order)),
                            C.c_void_p(order_N.ctypes.data if len(order_N) else 0), C.c_uint32(len(order_N)),
                            C.c_uint32(n_clean))
    return order


def preprocess_fastq(text: bytes):
    """Sequence side of preprocess() for one FASTQ file -> dict(clean, ndna, order_N, num_reads, num_clean, num_N, max_readlen).
    Raises ValueError with the reference's message on malformed input."""
    buf = np.frombuffer(text, dtype=np.uint8)
    clean = np.zeros(len(buf) + 16, np.uint8)
    ndna = np.zeros(len(buf) + 16, np.uint8)
    order_N = np.zeros(len(buf) // 4 + 4, np.uint32)
    counts = np.zeros(4, np.uint32)
    cb, nb = C.c_size_t(), C.c_size_t()
    f = lib().orc_preprocess_fastq
    f.restype = C.c_int
    rc = f(C.c_void_p(buf.ctypes.data if len(buf) else 0), C.c_size_t(len(buf)), C.c_void_p(clean.ctypes.data), C.byref(cb),
           C.c_void_p(ndna.ctypes.data), C.byref(nb), C.c_void_p(order_N.ctypes.data), C.c_void_p(counts.ctypes.data))
    if rc == -1:
        raise ValueError("Invalid FASTQ(A) file. Number of lines not multiple of 4(2)")
    if rc == -2:
        raise ValueError("Too long read length (please try --long/-l flag).")
    return dict(clean=clean[:cb.value].tobytes(), ndna=ndna[:nb.value].tobytes(), order_N=order_N[:counts[2]].copy(),
                num_reads=int(counts[0]), num_clean=int(counts[1]), num_N=int(counts[2]), max_readlen=int(counts[3]))


# --------------------------------------------------------------- encoder stage

class OrcEncIn(C.Structure):
    _fields_ = [("max_readlen", C.c_int), ("num_thr", C.c_int), ("read", C.c_void_p), ("len", C.c_void_p),
                ("n_clean", C.c_uint32), ("tid_off", C.c_void_p), ("order", C.c_void_p), ("rc", C.c_void_p),
                ("flag", C.c_void_p), ("pos", C.c_void_p), ("rlen", C.c_void_p), ("order_s", C.c_void_p),
                ("numreads_s", C.c_uint32), ("dnaN", C.c_void_p), ("order_N", C.c_void_p),
                ("numreads_N", C.c_uint32)]


class OrcEncOut(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("seq_len", C.c_uint64), ("seq_len_tid", C.c_void_p), ("pos", C.c_void_p),
                ("noise", C.c_void_p), ("noise_len", C.c_uint64), ("noisepos", C.c_void_p),
                ("n_noisepos", C.c_uint64), ("order", C.c_void_p), ("rlen", C.c_void_p), ("rc", C.c_void_p),
                ("n_aligned", C.c_uint64), ("n_total", C.c_uint64), ("unaligned", C.c_void_p),
                ("unaligned_bytes", C.c_uint64), ("len_unaligned", C.c_uint64), ("matched_s", C.c_uint32),
                ("matched_N", C.c_uint32), ("num_contigs", C.c_uint64), ("num_probes", C.c_uint64),
                ("num_hits", C.c_uint64)]


def _np_from(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype).copy()


def pack_dnaN(strings):
    """write_dnaN_in_bits (util.cpp:322-348) for a list of read strings -> bytes of input_N.dna."""
    code = {"A": 0, "G": 1, "C": 2, "T": 3, "N": 4}
    out = bytearray()
    for s in strings:
        out += int(len(s)).to_bytes(2, "little")
        b = bytearray((len(s) + 1) // 2)
        for i, ch in enumerate(s):
            b[i // 2] |= code[ch] << (4 * (i & 1))
        out += b
    return bytes(out)


def encode(read, ln, L, streams, num_thr=None, dnaN=b"", order_N=None):
    """encoder_main<N>() at -t 1 on in-memory inputs.  `streams` is a reorder result dict
    (order, rc, flag, pos, rlen, order_s, tid_off).  Returns a dict of the output streams."""
    Lb = lib()
    Lb.orc_encode.argtypes = [C.POINTER(OrcEncIn), C.POINTER(OrcEncOut)]
    Lb.orc_encode_free.argtypes = [C.POINTER(OrcEncOut)]
    read = np.ascontiguousarray(read, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.uint16)
    tid_off = np.ascontiguousarray(streams["tid_off"], dtype=np.uint64)
    T = len(tid_off) - 1 if num_thr is None else num_thr
    keep = dict(order=np.ascontiguousarray(streams["order"], np.uint32),
                rc=np.ascontiguousarray(streams["rc"], np.uint8),
                flag=np.ascontiguousarray(streams["flag"], np.uint8),
                pos=np.ascontiguousarray(streams["pos"], np.int64),
                rlen=np.ascontiguousarray(streams["rlen"], np.uint16),
                order_s=np.ascontiguousarray(streams["order_s"], np.uint32),
                order_N=np.ascontiguousarray(order_N if order_N is not None else [], np.uint32),
                dnaN=np.frombuffer(dnaN, dtype=np.uint8))
    i = OrcEncIn()
    i.max_readlen, i.num_thr = L, T
    i.read, i.len, i.n_clean = read.ctypes.data, ln.ctypes.data, len(ln)
    i.tid_off = tid_off.ctypes.data
    for k in ("order", "rc", "flag", "pos", "rlen", "order_s"):
        setattr(i, k, keep[k].ctypes.data)
    i.numreads_s = len(keep["order_s"])
    i.dnaN = keep["dnaN"].ctypes.data if len(keep["dnaN"]) else None
    i.order_N = keep["order_N"].ctypes.data
    i.numreads_N = len(keep["order_N"])
    o = OrcEncOut()
    rc = Lb.orc_encode(C.byref(i), C.byref(o))
    if rc != 0:
        raise ValueError("orc_encode failed: %d" % rc)
    res = dict(seq=_np_from(o.seq, o.seq_len, np.uint8).tobytes(),
               seq_len_tid=_np_from(o.seq_len_tid, T, np.uint64),
               pos=_np_from(o.pos, o.n_aligned, np.uint64),
               noise=_np_from(o.noise, o.noise_len, np.uint8).tobytes(),
               noisepos=_np_from(o.noisepos, o.n_noisepos, np.uint16),
               order=_np_from(o.order, o.n_total, np.uint32),
               rlen=_np_from(o.rlen, o.n_total, np.uint16),
               rc=_np_from(o.rc, o.n_aligned, np.uint8),
               unaligned=_np_from(o.unaligned, o.unaligned_bytes, np.uint8).tobytes(),
               len_unaligned=int(o.len_unaligned), matched_s=int(o.matched_s), matched_N=int(o.matched_N),
               num_contigs=int(o.num_contigs), num_probes=int(o.num_probes), num_hits=int(o.num_hits))
    Lb.orc_encode_free(C.byref(o))
    return res
