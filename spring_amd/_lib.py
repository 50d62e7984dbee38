"""ctypes binding of lib/libspring_reorder_hip.so (C ABI: include/spring_reorder.h)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPRING_AMD_LIB") or os.path.join(HERE, "lib", "libspring_reorder_hip.so")  # env: A/B runs

EXPORTS = [
    "spring_reorder_default_opts", "spring_reorder_last_error", "spring_reorder_trim_pool", "spring_reorder_run", "spring_reorder_create",
    "spring_reorder_destroy", "spring_reorder_load_dna", "spring_reorder_load_dna_device",
    "spring_reorder_build_dict", "spring_reorder_run_chains", "spring_reorder_auto_chains", "spring_reorder_finalize",
    "spring_reorder_mg_begin", "spring_reorder_mg_search", "spring_reorder_mg_slice", "spring_reorder_mg_apply",
    "spring_reorder_mg_end", "spring_reorder_mg_exchange_virtual", "spring_reorder_debug_check_seed_state",
    "spring_mg_rccl_unique_id", "spring_mg_comm_create_rccl", "spring_mg_comm_create_host", "spring_mg_comm_destroy",
    "spring_reorder_mg_run",
    "spring_reorder_get_stats", "spring_reorder_download", "spring_reorder_tid_split", "spring_reorder_emit_dna",
    "spring_reorder_dict_lookup", "spring_reorder_download_reads", "spring_synth_dna_bytes",
    "spring_reorder_load_fastq", "spring_reorder_fastq_N",
    "spring_order_invert_se", "spring_order_invert_pe", "spring_order_correct", "spring_order_pe_encode",
    "spring_fastq_reorder",
    "spring_synth_dna_host", "spring_synth_genome_host", "spring_synth_dna_device", "spring_reorder_load_synth", "spring_reorder_download_dna",
    "spring_encoder_create", "spring_encoder_destroy", "spring_encoder_set_split_tables", "spring_encoder_encode_reorder", "spring_encoder_download",
    "spring_encoder_download_seq_packed", "spring_encoder_get_info", "spring_reorder_encode_run", "spring_encoder_encode_host", "spring_encoder_run",
]


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("num_chains", C.c_uint32), ("num_thr", C.c_int32),
                ("collect_stats", C.c_int32), ("time_search", C.c_int32), ("force_literal_update", C.c_int32),
                ("rounds_per_sync", C.c_int32), ("long_budget", C.c_int32),
                ("first_shifts", C.c_int32), ("seed_wide", C.c_int32), ("tab_scale", C.c_int32),
                ("search_wpb", C.c_int32), ("dbg_search_lds", C.c_int32), ("dbg_apply_lds", C.c_int32),
                ("fused", C.c_int32), ("deep_bins", C.c_int32),
                ("num_devices", C.c_int32), ("devices", C.c_int32 * 8), ("mg_host_transport", C.c_int32),
                ("table_mode", C.c_int32), ("plan0", C.c_int32 * 6), ("plan1", C.c_int32 * 6), ("long_min", C.c_int32),
                ("long_blocks", C.c_int32), ("debug", C.c_int32), ("long_split", C.c_int32), ("entry_flags", C.c_int32),
                ("out_writers", C.c_int32), ("alternatives", C.c_int32), ("phases", C.c_int32), ("known_absent", C.c_int32)]


class FastqInfo(C.Structure):
    _fields_ = [("num_reads", C.c_uint32 * 2), ("num_reads_clean", C.c_uint32 * 2), ("num_reads_N", C.c_uint32 * 2),
                ("max_readlen", C.c_uint32), ("pad", C.c_uint32), ("ms_device", C.c_double)]


class EncoderInfo(C.Structure):
    _fields_ = ([(k, C.c_uint64) for k in ("n_aligned", "n_total", "seq_len", "noise_bytes", "n_noisepos",
                                           "unaligned_bytes", "len_unaligned", "num_contigs")]
                + [(k, C.c_uint32) for k in ("matched_s", "matched_N", "align_passes", "max_bin")]
                + [("ms_device", C.c_double), ("ms_phase", C.c_double * 8)])

    def asdict(self):
        d = {}
        for k, _ in self._fields_:
            v = getattr(self, k)
            d[k] = list(v) if hasattr(v, "__len__") else v
        return d


class Stats(C.Structure):
    _fields_ = ([(k, C.c_uint64) for k in ("n_reads", "n_matched", "n_single", "unmatched", "probes", "keyok",
                                          "cands", "hits", "iterations", "rounds", "lost")]
                + [("numkeys", C.c_uint64 * 2), ("dict_numreads", C.c_uint64 * 2)]
                + [(k, C.c_double) for k in ("ms_unpack", "ms_dict", "ms_chains", "ms_finalize", "ms_total",
                                             "ms_search_kernel")]
                + [("search_launches", C.c_uint64), ("device_bytes", C.c_uint64),
                   ("ms_exchange", C.c_double), ("ms_resolve_mark", C.c_double), ("chains", C.c_uint64),
                   ("deep_pool", C.c_uint64), ("long_searches", C.c_uint64),
                   ("table_minz", C.c_uint64), ("table_marked_lines", C.c_uint64), ("long_splits", C.c_uint64), ("alternatives", C.c_uint64), ("phases", C.c_uint64),
                   ("ms_search_busy", C.c_double)])

    def asdict(self):
        d = {}
        for k, _ in self._fields_:
            v = getattr(self, k)
            d[k] = list(v) if hasattr(v, "__len__") else v
        return d


# spring_mg_allgather_fn (include/spring_reorder.h): host_buf, slice_off, slice_bytes, total_bytes, user -> 0 on success
MG_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p)

_lib = None


def lib():
    """Loads the HIP library; fails loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "spring_amd: %s is missing. Build it with `python -m spring_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u8p = C.c_void_p, C.c_void_p
    L.spring_reorder_default_opts.argtypes = [C.POINTER(Opts)]
    L.spring_reorder_last_error.restype = C.c_char_p
    L.spring_reorder_trim_pool.restype = None
    L.spring_reorder_run.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                     C.POINTER(Opts)]
    L.spring_reorder_create.argtypes = [C.POINTER(vp), C.POINTER(Opts)]
    L.spring_reorder_destroy.argtypes = [vp]
    L.spring_reorder_destroy.restype = None
    L.spring_reorder_load_dna.argtypes = [vp, u8p, C.c_size_t, C.c_uint32, C.c_uint32]
    L.spring_reorder_load_dna_device.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int32]
    L.spring_reorder_build_dict.argtypes = [vp]
    L.spring_reorder_run_chains.argtypes = [vp]
    L.spring_reorder_auto_chains.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    L.spring_reorder_finalize.argtypes = [vp]
    L.spring_reorder_mg_begin.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.spring_reorder_mg_search.argtypes = [vp]
    L.spring_reorder_mg_slice.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.spring_reorder_mg_apply.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint32)]
    L.spring_reorder_mg_end.argtypes = [vp]
    L.spring_reorder_mg_exchange_virtual.argtypes = [C.POINTER(vp), C.c_uint32]
    L.spring_mg_rccl_unique_id.argtypes = [vp]
    L.spring_mg_comm_create_rccl.argtypes = [C.POINTER(vp), C.c_int32, vp, C.c_uint32, C.c_uint32]
    L.spring_mg_comm_create_host.argtypes = [C.POINTER(vp), MG_ALLGATHER_FN, vp, C.c_uint32, C.c_uint32]
    L.spring_mg_comm_destroy.argtypes = [vp]
    L.spring_mg_comm_destroy.restype = None
    L.spring_reorder_mg_run.argtypes = [vp, vp, C.c_uint32]
    L.spring_reorder_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.spring_reorder_download.argtypes = [vp] + [vp] * 8
    L.spring_reorder_tid_split.argtypes = [vp, vp, vp]
    L.spring_reorder_emit_dna.argtypes = [vp, C.c_int32, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.spring_reorder_dict_lookup.argtypes = [vp, C.c_int32, vp, C.c_uint32, vp, vp, C.c_size_t]
    L.spring_reorder_download_reads.argtypes = [vp, vp, vp]
    L.spring_reorder_load_fastq.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(FastqInfo)]
    L.spring_reorder_fastq_N.argtypes = [vp, C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.POINTER(C.c_uint32)]
    L.spring_order_invert_se.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_double)]
    L.spring_order_invert_pe.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_double)]
    L.spring_fastq_reorder.argtypes = [u8p, C.c_size_t, vp, C.c_uint32, u8p, C.c_size_t, C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_double)]
    L.spring_order_pe_encode.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_double)]
    L.spring_order_correct.argtypes = [vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.spring_synth_dna_bytes.restype = C.c_size_t
    L.spring_synth_dna_bytes.argtypes = [C.c_uint32, C.c_uint32]
    L.spring_synth_dna_host.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32]
    L.spring_synth_genome_host.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32]
    L.spring_synth_dna_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32]
    L.spring_reorder_load_synth.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32]
    L.spring_reorder_download_dna.argtypes = [vp, u8p, C.c_size_t]
    L.spring_encoder_create.argtypes = [C.c_int32, C.POINTER(vp)]
    L.spring_encoder_destroy.argtypes = [vp]
    L.spring_encoder_destroy.restype = None
    L.spring_encoder_encode_reorder.argtypes = [vp, vp, u8p, C.c_uint64, vp, C.c_uint32, C.POINTER(EncoderInfo)]
    L.spring_encoder_download.argtypes = [vp] + [vp] * 9
    L.spring_encoder_download_seq_packed.argtypes = [vp, vp, vp]
    L.spring_encoder_encode_host.argtypes = [vp, C.c_uint32, C.c_int32, vp, u8p, C.c_uint64, vp, vp, vp, vp, vp, u8p,
                                             C.c_uint64, vp, C.c_uint32, u8p, C.c_uint64, vp, C.c_uint32,
                                             C.POINTER(EncoderInfo)]
    L.spring_encoder_run.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32,
                                     C.POINTER(EncoderInfo)]
    L.spring_encoder_get_info.argtypes = [vp, C.POINTER(EncoderInfo)]
    L.spring_reorder_encode_run.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.POINTER(Opts), C.POINTER(EncoderInfo)]
    for name in EXPORTS:
        if name not in ("spring_reorder_last_error", "spring_reorder_destroy", "spring_synth_dna_bytes",
                        "spring_reorder_default_opts", "spring_reorder_trim_pool", "spring_encoder_destroy",
                        "spring_mg_comm_destroy"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L
