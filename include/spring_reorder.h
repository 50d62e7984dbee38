/*
 * include/spring_reorder.h -- C ABI of libspring_reorder_hip.so
 *
 * MI355X-native replacement for SPRING's read-reordering stage.  Every entry
 * point names the reference interface it replaces (paths relative to
 * /root/reference/src).  Plain pointers and sizes only; the library owns all
 * device memory, the caller owns every host buffer it passes in.
 *
 * Return value: 0 on success, negative on error (see SPRING_REORDER_E_*);
 * spring_reorder_last_error() returns a thread-local message.  The reference
 * signals errors by throwing std::runtime_error (call_template_functions.cpp:60,
 * main.cpp:140-166); INTEGRATION.md shows the 10-line shim that converts.
 */
#ifndef SPRING_REORDER_H_
#define SPRING_REORDER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPRING_REORDER_E_ARG (-1)     /* bad argument (e.g. max_readlen > 511: "Wrong bitset size.") */
#define SPRING_REORDER_E_IO (-2)      /* file missing / short / unwritable                         */
#define SPRING_REORDER_E_HIP (-3)     /* HIP runtime error (no device, out of memory, ...)         */
#define SPRING_REORDER_E_STATE (-4)   /* entry points called out of order                          */

typedef struct spring_reorder_ctx spring_reorder_ctx;

/* Options.  Zero-initialise with spring_reorder_default_opts().  Three fields change the bytes written for num_chains > 1
 * -- num_chains, alternatives, phases (stats reports what ran) --; every other field moves work
 * between batches, kernels and table sizes only and is covered by a parity test at a non-default value
 * (tests/test_gpu_parity.py::test_tuning_opts_do_not_change_results).  How the library makes its own choices: DESIGN.md
 * section 3 (one table of every threshold). */
typedef struct {
  int32_t device;       /* HIP device ordinal; -1 = current device */
  uint32_t num_chains;  /* K concurrent greedy chains (= reference threads, reorder.h:351); 1 reproduces `-t 1` byte for
                           byte; 0 = the library's choice (spring_reorder_auto_chains) */
  int32_t num_thr;      /* number of per-tid output sets to emit (cp.num_thr, reorder.h:748) */
  int32_t collect_stats;/* 1: count reference-equivalent probes / key hits / Hamming evaluations (slower kernels) */
  int32_t time_search;  /* 1: HIP events around every round-kernel launch (bench roofline) */
  int32_t force_literal_update; /* 1: consensus update through the literal lane-0 restatement of reorder.h:133-212 (tests) */
  int32_t rounds_per_sync;      /* rounds enqueued between host looks at the running chains; 0 = auto */
  int32_t long_budget;  /* deep-bin pools: compare passes a wavefront spends on one search before the search goes to the
                           long-search kernels; 0 = the library's choice (8, on pools of very deep bins only), > 0: on
                           whenever the deep-bin kernel runs, -1 = never */
  /* ---- tuning / experiments: 0 = default, the output does not depend on any of them */
  int32_t first_shifts;   /* ordered probe batches of k and 16 shifts, then the rest at once: k = 1..16; -1: 4, 4, 8, 8 */
  int32_t seed_wide;      /* -1: a chain whose seed is still unmatched uses plan0 as well (default: plan1 = 16, 16) */
  int32_t tab_scale;      /* dictionary table size multiplier 1 / 2 / 4 (default 2: load <= 0.2) */
  int32_t search_wpb;     /* two-kernel round: chains (wavefronts) per block of the search kernel: 1 / 2 / 4 */
  int32_t dbg_search_lds; /* occupancy experiment: dummy LDS bytes per round / search block */
  int32_t dbg_apply_lds;  /* same for the apply kernel */
  int32_t fused;          /* 0: fused round kernel, mapping chosen from the run; -1: two kernels per round (search, apply);
                             2: one chain per wavefront always; 3: four chains per wavefront wherever that kernel applies */
  int32_t deep_bins;      /* 0: from the dictionary; 1 / -1: the kernel variants with the deep-bin machinery on / off */
  /* ---- spring_reorder_run on several GPUs of one node: one read pool over devices[0 .. num_devices),
   * one host thread and context per entry inside the library, merged per-tid file set; the output equals the single-device
   * output with the same num_chains / alternatives / phases -- and with the library's own choices when the default chain
   * count is a multiple of num_devices (it is for 2, 4, 8 devices from 16 384 chains on).  0 / 1: `device` alone.  An entry
   * may repeat a device (tests): the exchange then goes through host memory, as with mg_host_transport = 1.
   * spring_reorder_encode_run runs on `device` alone and rejects num_devices >= 2. */
  int32_t num_devices;
  int32_t devices[8];
  int32_t mg_host_transport;
  int32_t table_mode;     /* 0 / 1: table addressed by the key's hash; 2: by its minimizer where that applies (32-base windows,
                             reads of 100..192 bases, four-chain kernel without known-absent masks): an experiment */
  int32_t plan0[6];       /* ordered probe batches of a search in shifts: up to six widths of 1..16, sum <= 32, 0-terminated;
                             empty = the library's choice.  plan1: for a chain whose seed has no match yet */
  int32_t plan1[6];
  int32_t long_min;       /* long searches: bin entries that must still be ahead of a search for it to be handed over (0 = 512) */
  int32_t long_blocks;    /* long searches: grid unit of the long-search kernels (0 = 256, the CUs) */
  int32_t debug;          /* 1: stage / phase timings on stderr */
  int32_t long_split;     /* long searches: chunks of 64 bin entries per part (0 = 192, -1 = never cut a search into parts) */
  int32_t entry_flags;    /* deep-bin pools: -1 = bin entries do not carry their read's taken bit (the scans ask the bitmap) */
  int32_t out_writers;    /* spring_reorder_run: threads that write the output files; 0 = hardware threads / 4, in [4, 24] */
  int32_t alternatives;   /* candidates per match proposal (specification: orc_reorder_rounds_alt).  1: a chain that loses its
                             proposed read searches again next round.  2: the search also records the next passing read of the
                             winning bin -- what the reference's thread tries next after losing the read_lock race
                             (reorder.h:303-311) -- and a second resolution pass hands it to the loser.  Both are legal `-t K`
                             interleavings; THE OUTPUT DEPENDS ON IT for num_chains > 1.  <= 0: the library's choice (2 on
                             contended pools), reported in stats.alternatives.  2 needs fused rounds and < 2^27 - 1 reads */
  int32_t phases;         /* chain groups (specifications: orc_reorder_rounds, orc_reorder_rounds_ph).  1: all chains advance
                             in lock-step rounds.  2: the chains run as two groups whose rounds alternate, one group's round
                             kernel beside the other's; a group searches on the pool as it was after its own last round and
                             loses a read the other group took in between; group 0 takes its contig seeds from the upper half
                             of the read ids, group 1 from the lower half.  That seed rule is a deliberate deviation: a
                             deterministic, decodable schedule, NOT something reference threads can do (they scan the whole
                             pool from the top, reorder.h:576-599).  THE OUTPUT DEPENDS ON IT for num_chains > 1.  <= 0: the
                             library's choice, reported in stats.phases, the same on one GPU and in a pool.  2 needs the fused
                             round, >= 4096 chains, 8192 .. 2^31 - 1 reads, not fewer reads than chains; in a pool over G GPUs
                             both groups' chain counts must be multiples of 256 G */
  int32_t known_absent;   /* four-chain round kernel, reads up to 192 bases: chains remember which windows of their consensus
                             are known absent from the dictionaries and skip them (the table is immutable).  0 = on, -1 = off */
} spring_reorder_opts;

typedef struct {
  /* work counters (reference-equivalent, see DESIGN.md "Algorithmic bytes") */
  uint64_t n_reads, n_matched, n_single, unmatched;
  uint64_t probes, keyok, cands, hits, iterations, rounds, lost;
  uint64_t numkeys[2], dict_numreads[2];
  /* device-side wall clock of the stages, milliseconds (HIP events on the library's stream) */
  double ms_unpack, ms_dict, ms_chains, ms_finalize, ms_total;
  /* search kernel: summed launch durations and launch count (only when time_search=1) */
  double ms_search_kernel;
  uint64_t search_launches;
  uint64_t device_bytes;   /* bytes of HBM the context holds */
  /* multi-GPU pools with time_search = 1: summed durations of the per-round exchange (all-gather of the proposal
   * words) and of the kernels every rank runs over ALL chains after it (k_mg_resolve + k_mg_mark); ms_search_kernel is
   * the round kernel over the rank's own chains */
  double ms_exchange, ms_resolve_mark;
  uint64_t chains;         /* K the chain phase ran with (opts.num_chains, or what the default rule chose) */
  uint64_t deep_pool;      /* bit 0: the dictionary averages >= 1.3 reads per key (the deep-coverage default applies); bit 1: it has a
                              heavy tail -- >= 0.5 % of its reads in bins of >= 256 entries (repeat families of a real genome) */
  uint64_t long_searches;  /* searches a wavefront handed over to a block of 16 (k_long; deep-coverage pools only) */
  uint64_t table_minz;     /* 1: the dictionary table is addressed by minimizers (opts.table_mode) */
  uint64_t table_marked_lines; /* ... and this many of its lines were over-subscribed: their keys live at the redirect address */
  uint64_t long_splits;    /* long searches that were split into parts over several blocks (k_long) */
  uint64_t alternatives;   /* candidates per match proposal the chain phase ran with (opts.alternatives, or the library's choice) */
  uint64_t phases;         /* chain groups the chain phase ran with (opts.phases, or the library's choice) */
  double ms_search_busy;   /* time_search = 1: the time during which at least one round kernel was running (the union of the
                              launches' intervals); = ms_search_kernel unless two chain groups run side by side (phases = 2) */
} spring_reorder_stats;

void spring_reorder_default_opts(spring_reorder_opts *o);
const char *spring_reorder_last_error(void);
/* Contexts return their device blocks to a per-device pool on destroy (repeated runs then skip
 * hipMalloc/hipFree, which cost about as much as the stage at 100 M reads); this releases the pool. */
void spring_reorder_trim_pool(void);

/* ---------------------------------------------------------------------------
 * Drop-in stage: replaces
 *     void spring::call_reorder(const std::string &temp_dir, compression_params &cp)
 *     (call_template_functions.h:9, body call_template_functions.cpp:9-63)
 *     -> reorder_main<bitset_size>(temp_dir, cp)            (reorder.h:732-786)
 * Reads  temp_dir/input_clean_1.dna (+ _2.dna when paired_end) and DELETES them
 * (reorder.h:232,241); writes, for every tid in [0,num_thr):
 *   read_order.bin.<tid> (u32 LE)      read_rev.txt.<tid> (gzip, 'd'/'r')
 *   tempflag.txt.<tid> (gzip, '0'/'1') temppos.txt.<tid> (gzip, i64 LE)
 *   read_lengths.bin.<tid> (gzip, u16) temp.dna.<tid> (u16 len + 2-bit bases, RC applied)
 * and temp.dna.singleton, read_order.bin.singleton, temp.dna.singleton.count
 * (reorder.h:355-368, :643-730), i.e. exactly what encoder_main<> opens
 * (encoder.h:580-593).  Fields used from compression_params: max_readlen,
 * num_thr, paired_end, num_reads_clean[0..1] (reorder.h:747-763).
 */
int spring_reorder_run(const char *temp_dir, uint32_t max_readlen, int32_t num_thr, int32_t paired_end,
                       uint32_t num_reads_clean_0, uint32_t num_reads_clean_1,
                       const spring_reorder_opts *opts /* NULL = defaults */);

/* ---------------------------------------------------------------------------
 * Stage pieces on in-memory buffers (tests, bench, pipelines that keep reads
 * in memory).  Call order: create -> load_* -> build_dict -> run_chains ->
 * finalize -> download/emit -> destroy.
 */
int spring_reorder_create(spring_reorder_ctx **ctx, const spring_reorder_opts *opts);
void spring_reorder_destroy(spring_reorder_ctx *ctx);

/* readDnaFile (reorder.h:222-244): `dna` = n records of u16 len + ceil(len/4)
 * bytes (util.cpp:269-294), file 1 records followed by file 2 records.
 * Host buffer -> HBM copy, then the unpack kernel builds the limb array. */
int spring_reorder_load_dna(spring_reorder_ctx *ctx, const uint8_t *dna, size_t nbytes, uint32_t n,
                            uint32_t max_readlen);

/* Same, but the record stream is already resident in HBM (d_dna is a device
 * pointer the caller owns; fixed_len != 0 promises every record has
 * len == max_readlen so record i starts at i*(2+ceil(L/4))). */
int spring_reorder_load_dna_device(spring_reorder_ctx *ctx, const void *d_dna, size_t nbytes, uint32_t n,
                                   uint32_t max_readlen, int32_t fixed_len);

/* constructdictionary (bitset_util.h:74-221) for both dictionaries of
 * reorder.h:751-759: keys -> sort -> unique -> exact hash table + CSR bins. */
int spring_reorder_build_dict(spring_reorder_ctx *ctx);

/* reorder() (reorder.h:320-641): the greedy chain search, K chains in
 * lock-step rounds (search_match reorder.h:246-318, updaterefcount :110-220). */
int spring_reorder_run_chains(spring_reorder_ctx *ctx);
/* The chain count run_chains() uses when opts.num_chains = 0 (valid after build_dict): n / 1024, or n / 128 when the
 * dictionary averages >= 1.3 reads per key (*deep = 1), at most 65536 (131072 on deep pools other than the very deepest).  A multi-GPU caller that wants the pool to equal
 * the single-GPU default passes this value (rounded up to a multiple of the world size) to mg_begin / mg_run. */
int spring_reorder_auto_chains(spring_reorder_ctx *ctx, uint32_t *chains, int32_t *deep);

/* ---- FASTQ front end.  The sequence side of preprocess() (src/preprocess.cpp:186-214,:293-304;
 * read_fastq_block src/util.cpp:31-54; write_dna_in_bits / write_dnaN_in_bits src/util.cpp:269-294,:322-348) on
 * the GPU: 4-line FASTQ in host memory -- text, or a gzip file image (magic 1f 8b, any number of members:
 * preprocess.cpp:154-183 reads .gz input too), which is inflated on the host first -> reads without N packed 2 bits/base straight into the
 * record stream the reorder stage consumes (what the reference writes to input_clean_{1,2}.dna; file-2 reads
 * follow file-1 reads), reads with N packed 4 bits/base (input_N.dna[.2]) with their position in their file
 * (read_order_N.bin[.2]).  Errors mirror the reference's exceptions ("Invalid FASTQ(A) file. Number of lines not
 * multiple of 4(2)", "Too long read length ...", "Number of reads in paired files do not match.").
 * After it the context is loaded (build_dict next); spring_reorder_download_dna returns the clean stream. */
typedef struct {
  uint32_t num_reads[2], num_reads_clean[2], num_reads_N[2];
  uint32_t max_readlen; /* over all reads, with or without N (preprocess.cpp:312-315) */
  uint32_t pad;
  double ms_device;     /* device time from the first parse kernel to the end of packing + unpack (HIP events) */
} spring_fastq_info;
int spring_reorder_load_fastq(spring_reorder_ctx *ctx, const uint8_t *fastq_1, size_t nbytes_1, const uint8_t *fastq_2,
                              size_t nbytes_2 /* fastq_2 = NULL: single end */, spring_fastq_info *info);
/* N reads of input file `which` (0/1): 4-bit packed records + positions; any pointer may be NULL. */
int spring_reorder_fastq_N(spring_reorder_ctx *ctx, int32_t which, uint8_t *n_dna, size_t cap, size_t *nbytes,
                           uint32_t *order_N, uint32_t *count);

/* ---- single-pool multi-GPU (one process per GPU).  Every rank loads the same
 * reads and builds the same dictionaries; rank r owns chains [r*K/world, (r+1)*K/world), K =
 * total_chains.  Per round: mg_search -> caller all-gathers the proposal words (mg_slice says which
 * bytes are this rank's) -> mg_apply.  Loop until *alive == 0, then mg_end + finalize/download as
 * usual (each rank then holds the streams of its own chains; tid of chain c is c % num_thr, so tid t
 * of the whole job is the concatenation over ranks of every rank's tid-t segment).  The result is
 * bit-identical to run_chains() with num_chains = total_chains on one GPU, whatever `world` is.
 * d_prop: device buffer of total_chains*8 bytes owned by the caller (e.g. a torch tensor it can hand
 * to torch.distributed), or NULL to let the library allocate it. */
int spring_reorder_mg_begin(spring_reorder_ctx *ctx, uint32_t rank, uint32_t world, uint32_t total_chains, void *d_prop);
int spring_reorder_mg_search(spring_reorder_ctx *ctx);
int spring_reorder_mg_slice(spring_reorder_ctx *ctx, void **d_prop, size_t *slice_off, size_t *slice_bytes,
                            size_t *total_bytes);
int spring_reorder_mg_apply(spring_reorder_ctx *ctx, int32_t check_alive, uint32_t *alive);
int spring_reorder_mg_end(spring_reorder_ctx *ctx);
/* test hook, between mg_apply and the next mg_search: violations[0] = bitmap words with an untaken read above the cursor,
 * violations[1] = blocks below the cursor's block whose untaken-read count differs from the bitmap (what the seed pick,
 * reorder.h:576-592 on the GPU, relies on). */
int spring_reorder_debug_check_seed_state(spring_reorder_ctx *ctx, uint64_t *violations);
/* all-gather between `world` contexts living in ONE process on one device (tests). */
int spring_reorder_mg_exchange_virtual(spring_reorder_ctx **ctxs, uint32_t world);

/* ---- the same pool with the exchange INSIDE the library (no host round trip per round).
 * spring_reorder_mg_run = mg_begin + rounds until no chain is running + mg_end; the per-round all-gather of the
 * proposal words goes through a communicator made once per process and reused by any number of runs:
 *   spring_mg_comm_create_rccl  ncclAllGather (in place) on the library's own stream.  RCCL is loaded at run time
 *                               (librccl.so.1), so single-GPU users never pay for it.  Rank 0 makes the 128-byte
 *                               id with spring_mg_rccl_unique_id and the caller hands it to every rank (bench.py:
 *                               one torch.distributed broadcast; MPI or a file work as well); the call is
 *                               collective over the ranks (ncclCommInitRank).
 *   spring_mg_comm_create_host  a caller-supplied all-gather on a host staging buffer: `fn` receives the whole
 *                               buffer with this rank's slice filled in and must fill the others (tests: gloo
 *                               between two processes that share one GPU; any transport the caller has).
 * Termination needs no extra collective: every rank recounts the running chains from the gathered words.
 * The reference has no counterpart (its chains are OpenMP threads sharing remainingreads[], reorder.h:343-344,
 * :402-421); the contract is the one of mg_begin: output == run_chains() with num_chains = total_chains. */
#define SPRING_RCCL_ID_BYTES 128
typedef struct spring_mg_comm spring_mg_comm;
typedef int (*spring_mg_allgather_fn)(void *host_buf, size_t slice_off, size_t slice_bytes, size_t total_bytes,
                                      void *user);
int spring_mg_rccl_unique_id(void *id128);
int spring_mg_comm_create_rccl(spring_mg_comm **comm, int32_t device, const void *id128, uint32_t rank, uint32_t world);
int spring_mg_comm_create_host(spring_mg_comm **comm, spring_mg_allgather_fn fn, void *user, uint32_t rank,
                               uint32_t world);
void spring_mg_comm_destroy(spring_mg_comm *comm);
int spring_reorder_mg_run(spring_reorder_ctx *ctx, spring_mg_comm *comm, uint32_t total_chains);

/* Gathers the per-chain emissions into the per-tid streams (the replay side of
 * writetofile, reorder.h:643-730). */
int spring_reorder_finalize(spring_reorder_ctx *ctx);

int spring_reorder_get_stats(spring_reorder_ctx *ctx, spring_reorder_stats *st);

/* Copies the streams to host arrays sized n_matched / n_single (get_stats) and
 * num_thr+1 for the offsets: what the reference writes to read_order.bin.<tid>,
 * read_rev.txt.<tid>, tempflag.txt.<tid>, temppos.txt.<tid>, read_lengths.bin.<tid>
 * (concatenated in tid order) and read_order.bin.singleton.  Any pointer may be NULL. */
int spring_reorder_download(spring_reorder_ctx *ctx, uint32_t *order, char *rc, char *flag, int64_t *pos,
                            uint16_t *rlen, uint32_t *order_s, uint64_t *tid_off, uint64_t *tid_off_s);
/* A rank of a multi-GPU pool that ran the chains as two groups (stats.phases = 2) owns a slice of each group: inside tid t
 * its records [tid_off[t], mid[t]) belong to chains of the first group, [mid[t], tid_off[t + 1]) to chains of the second
 * (likewise mid_s for the singleton stream).  The merged tid-t stream of the pool -- chain ids ascending -- is every rank's
 * first part, ranks ascending, then every rank's second part.  mid / mid_s: num_thr entries, either may be NULL.  With one
 * group mid[t] = tid_off[t + 1]. */
int spring_reorder_tid_split(spring_reorder_ctx *ctx, uint64_t *mid, uint64_t *mid_s);

/* temp.dna.<tid> (tid >= 0) or temp.dna.singleton (tid = -1) byte stream, built
 * on the device (reverse complement + 2-bit repack, reorder.h:667-687,
 * util.cpp:269-294).  dst may be NULL to query *nbytes. */
int spring_reorder_emit_dna(spring_reorder_ctx *ctx, int32_t tid, uint8_t *dst, size_t cap, size_t *nbytes);

/* Test hook: bins of dictionary `which` for the given keys, as
 * bbhashdict::findpos would return them on the untouched dictionary
 * (bitset_util.cpp:20-35).  bin_size[i] = 0xffffffff when the key is absent;
 * ids are concatenated into bin_ids (capacity ids_cap). */
int spring_reorder_dict_lookup(spring_reorder_ctx *ctx, int32_t which, const uint64_t *keys, uint32_t nkeys,
                               uint32_t *bin_size, uint32_t *bin_ids, size_t ids_cap);

/* Test hook: limb array + lengths as the unpack kernel produced them. */
int spring_reorder_download_reads(spring_reorder_ctx *ctx, uint64_t *limbs /* n*W */, uint16_t *len);

/* ---------------------------------------------------------------------------
 * consumers of read_order.bin that are pure permutation / prefix-sum work.
 * Host arrays in and out; *kernel_ms (may be NULL) = device time of the kernels alone.
 *   spring_order_invert_se : generate_order_se (reorder_compress_quality_id.cpp:117-125)
 *                            order_array[order[i]] = i, i < n
 *   spring_order_invert_pe : generate_order_pe (reorder_compress_quality_id.cpp:101-115)
 *                            for i < n: if (order[i] < n/2) order_array[order[i]] = pos++   (n/2 outputs)
 *   spring_order_correct   : correct_order (encoder.cpp:177-222): every index into the clean-read array
 *                            (m entries: the per-tid order streams and the singleton order) is shifted by the
 *                            number of N reads that precede that clean read in the original file;
 *                            order_N = original positions of the nN reads with N, n_clean = clean reads.
 *   spring_order_pe_encode : pe_encode (pe_encode.cpp:24-84): read_order.bin of a paired-end run -> position
 *                            of every reordered read in the decompressed files: file-1 reads keep their
 *                            reordered rank among file-1 reads, a file-2 read gets its mate's rank + n/2.
 */
int spring_order_invert_se(const uint32_t *order, uint32_t n, uint32_t *order_array, double *kernel_ms);
int spring_order_invert_pe(const uint32_t *order, uint32_t n, uint32_t *order_array, double *kernel_ms);
int spring_order_correct(uint32_t *order, uint64_t m, const uint32_t *order_N, uint32_t nN, uint32_t n_clean,
                         double *kernel_ms);
int spring_order_pe_encode(const uint32_t *order, uint32_t n, uint32_t *new_order, double *kernel_ms);

/* ---------------------------------------------------------------------------
 * reorder-only output.  Output record k = 4-line record order[k] of the FASTQ text (host
 * buffers; records keep their bytes, a missing final newline is added).  With order = the encoder stage's
 * read_order.bin (single-end) this is the order in which the decompressor emits reads without
 * --preserve-order.  out == NULL: *out_bytes receives the size needed.
 */
int spring_fastq_reorder(const uint8_t *fastq, size_t nbytes, const uint32_t *order, uint32_t n, uint8_t *out,
                         size_t out_cap, size_t *out_bytes, double *kernel_ms);

/* ---------------------------------------------------------------------------
 * Synthetic input for bench.py / tests: uniform random
 * genome of G bases, n reads of length L at uniform positions, i.i.d.
 * substitutions at rate err_ppm/1e6, 50 % reverse complemented, no N.  A
 * counter-based generator (splitmix64 of the index), so the host and the
 * device version produce identical bytes.  Output = .dna record stream.
 */
#define SPRING_SYNTH_REPEATS 0x80000000u /* OR into err_ppm: genome whose eighths 0,2,4,6 are exact copies
                                            (the repeat-rich "hard" distribution of SURVEY.md 8(d)) */
#define SPRING_SYNTH_PAIRED 0x40000000u  /* OR into err_ppm: paired-end pool (n even).  Read i < n/2 is read i of file 1,
                                            read n/2 + i its mate: the two ends of a fragment of ~N(400, 50) bases, on
                                            opposite strands (BASELINE config 4; reorder.h:233-242 lays a paired pool
                                            out as file-1 reads followed by file-2 reads) */
#define SPRING_SYNTH_GENOMIC 0x20000000u /* OR into err_ppm: a genome with the repeat structure of a real one -- per 512-base
                                            segment: 20 % a copy of one of 64 Zipf-sized interspersed-repeat families (5-20 %
                                            divergence from the family consensus), 5 % a tandem repeat (unit 2..40 bases), 3 %
                                            low-complexity runs, the rest unique sequence (synth_common.h) */
size_t spring_synth_dna_bytes(uint32_t n, uint32_t L);
int spring_synth_dna_host(uint8_t *dst, uint32_t n, uint32_t L, uint64_t G, uint64_t seed, uint32_t err_ppm);
/* the genome the reads are drawn from, as G letters (tests of the generator itself); flags = SPRING_SYNTH_REPEATS or 0 */
int spring_synth_genome_host(uint8_t *dst, uint64_t G, uint64_t seed, uint32_t flags);
/* generates into a device buffer the caller owns (e.g. a torch uint8 tensor); stream = 0. */
int spring_synth_dna_device(void *d_dst, uint32_t n, uint32_t L, uint64_t G, uint64_t seed, uint32_t err_ppm);
/* generates straight into HBM owned by the context and loads it (fixed_len). */
int spring_reorder_load_synth(spring_reorder_ctx *ctx, uint32_t n, uint32_t L, uint64_t G, uint64_t seed,
                              uint32_t err_ppm);
/* copies the device-generated record stream back (tests). */
int spring_reorder_download_dna(spring_reorder_ctx *ctx, uint8_t *dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SPRING_REORDER_H_ */
