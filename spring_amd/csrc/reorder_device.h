// spring_amd/csrc/reorder_device.h -- device-side data layout shared by the
// kernels (reorder_kernels.hip) and the host pipeline (reorder_pipeline.cpp).
#ifndef SPRING_REORDER_DEVICE_H_
#define SPRING_REORDER_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sr {

constexpr int MAX_READ_LEN = 511;   // params.h:22
constexpr int MAX_SEARCH = 1000;    // params.h:26 MAX_SEARCH_REORDER
constexpr int THRESH = 4;           // params.h:27 THRESH_REORDER
constexpr uint32_t DEEP_BIN = 16;   // bins with at least this many reads are tail-trimmed between rounds
constexpr uint32_t MID_BIN = 64;    // reads in bins of at least this size are counted too (DictBuild::ndeep[2]): the heavy-tail rule
constexpr uint32_t BIG_BIN = 256;   // reads in bins of at least this size are counted (DictBuild::ndeep[1]): long searches, k_long
constexpr uint32_t CHUNK = 64;      // emission slots a chain reserves per global atomic
constexpr uint32_t MARK_BLOCK = 256; // chains per block of k_mg_mark = per class-list segment (k_round_mc)
constexpr uint32_t MC_WAVES_PER_BLOCK = MARK_BLOCK / 4 + 3;  // wavefronts of four chains a segment can need (each class rounded up)
constexpr int UBLK_SHIFT = 12;      // DevParams::ublk counts untaken reads per 2^12 reads (64 bitmap words = 512 bytes: what a
                                    // seed pick reads of the cursor's block; 2^14 cost four times the L1 requests per pick)
constexpr int LDS_PAD = 10;         // zero limbs either side of ref/revref in LDS
constexpr int LDS_LIMBS = 16 + 2 * LDS_PAD;

enum { MODE_SEARCH = 0, MODE_NEED_SEED = 1 };
enum { PROP_NONE = 0, PROP_MATCH = 1, PROP_SEED = 2, PROP_FRESH = 3 /* fused rounds: nothing proposed yet */ };
// per-chain proposal word exchanged between ranks in single-pool multi-GPU mode:
// kind << 32 | rid, bit 35 = "this seed is the lowest of the round" (moves the cursor), bit 36 below; with the alternatives
// schedule (DevParams::alts = 2) a PK_MATCH word also carries the chain's SECOND candidate, alt + 1 in bits 37..63 (0: none;
// pools of fewer than 2^27 - 1 reads) -- so every consumer of the words, the multi-GPU exchange included, moves 8 bytes as before
enum { PK_NONE = 0, PK_MATCH = 1, PK_SEED = 2, PK_NOSEED = 4, PK_DONE = 5 };
constexpr unsigned long long PK_CURSOR_BIT = 1ull << 35;
constexpr unsigned long long PK_WILLNEED_BIT = 1ull << 36;  // PK_NONE after a failed left search: the chain needs a seed next
constexpr int PK_ALT_SHIFT = 37;
constexpr uint32_t ALT_MAX_READS = (1u << 27) - 1;
// resv[] with alternatives: pass 0 (every first candidate, seeds) writes the chain id, pass 1 (k_alt_resolve: the second
// candidate of a chain that lost pass 0) writes ALT_KEY | chain id -- an earlier pass beats a later one, inside a pass the
// lowest chain id wins (specification: orc_reorder_rounds_alt); chain ids stay below 2^28
constexpr uint32_t ALT_KEY = 1u << 28;

// Per-chain state (one greedy chain == one reference OpenMP thread, reorder.h:351-431).
// The 64-byte header is wave-uniform: the chain kernels fetch it with ONE scalar load into 16 SGPRs (no VGPRs, no
// LDS round trip per field), update it with scalar ALU and write it back as 16-byte quarters (lane q stores quarter
// q).  The search half only changes the last two quarters (dwords 8..15).
struct __attribute__((aligned(16))) ChainHot {
  long long ref_pos;     // reorder.h:397
  int32_t ref_len;
  uint32_t e_slot;       // next free slot in this chain's matched-record chunk
  uint32_t prev, first_rid;
  uint32_t n_emit, n_single;
  uint32_t s_slot;       // next free slot in the singleton chunk
  uint32_t num_reads_thr, num_unmatched_past;   // early-stop window (reorder.h:380-381)
  uint32_t prop_rid;
  union {  // dword 12
    struct {
      uint32_t done : 1, prev_unmatched : 1, left_search : 1, stop_searching : 1;
      uint32_t mode : 1, retrying : 1, finishing : 1, cursor_writer : 1;
      uint32_t cnt_buf : 1;
      uint32_t cnt_wide : 1;   // the committed count buffer is in cnt (else cnt8)
      uint32_t prop_kind : 2;
      uint32_t prop_rev : 3;   // bit 0 reverse match, bit 1 dictionary of the winning probe, bit 2 "a repeated search may resume at this probe"
      uint32_t prop_shift : 9;
    };
    uint32_t flags;
  };
  uint32_t alt1;   // alternatives schedule: second candidate of the proposed match + 1 (0: none)
  uint32_t pad[2];
};
static_assert(sizeof(ChainHot) == 64, "ChainHot must be one 64-byte line");

struct __attribute__((aligned(64))) Chain {
  ChainHot h;
  uint64_t ref[16];      // consensus, 2 bits/base (reorder.h:371)
  uint64_t revref[16];   // its reverse complement
  uint64_t st_probes, st_keyok, st_cands, st_iter, st_lost, st_hits, n_unmatched, st_long /* searches finished by k_long */;
#ifdef SR_PHASE_TIMING  // experiment builds (tools/xbuild.sh): shader clocks per phase of k_round, [32 + k] = visits
  uint64_t pt[64];
#endif
};
#ifndef SR_PHASE_TIMING
static_assert(sizeof(Chain) == 384, "Chain layout");
#endif

struct Globals {
  long long cursor;   // every read above it is taken (== min over threads of remainingpos, reorder.h:402)
  uint32_t alive;     // chains not done
  uint32_t e_alloc;   // next unallocated chunk of the matched-record buffer
  uint32_t s_alloc;   // next unallocated chunk of the singleton buffer
  uint32_t pad;
  long long cursor_b; // the second chain group's cursor (two-group schedule: DevParams::phases)
};

// The dictionary table (both dictionaries in one): 2^(64-bshift) buckets of 32 bytes, [tag x4 | payload x4]; a probe loads
// the 16 tag bytes (buck[2 b]) and the payload word of a slot (word 8 b + 4 + s) on a fingerprint match only -- by then
// the line is in L1 / L2.
// Home bucket of a key with hash h = mix64(key):
//   minz = 0: h >> bshift;
//   minz = 1 (32-base windows, reads of 100..192 bases): four consecutive buckets = one 128-byte neighbourhood, chosen by
//     the key's minimizer (the smallest canonical 16-mer of the window, minz_of_key), the bucket inside it by two bits of
//     h.  Consecutive windows of a consensus share their minimizer for ~9 shifts on average, so the ~151 tag fetches of
//     a failing search touch a few dozen cache lines instead of 151.  A neighbourhood that more than MINZ_HEAVY keys ask for
//     (repeats, low-complexity sequence) keeps none of them: its four buckets carry TAG_MARK in slot 0 and its keys live
//     at tab_redirect(h), where a lookup that sees the mark continues.
struct TabView {
  const uint4 *buck;
  int bshift;
  int minz;
  int lshift;   // minz: neighbourhood = mixed minimizer >> lshift (32 - log2(buckets / 4), 2..31)
};
constexpr uint32_t TAG_MARK = 0xfffffffcu;  // fingerprint 0x3fffffff is never handed out (fp30_of)
constexpr int MINZ_HEAVY = 12;              // keys a line may be the home of
constexpr int MINZ_K = 16;                  // minimizer length in bases
constexpr int MINZ_WL = 32;                 // ... of windows of this length only (MINZ_WL - MINZ_K + 1 = 17 k-mers per window)

constexpr int LONG_MAX_BINS = 1024;  // one bin per probe code of a search (k_long: one thread per probe)
constexpr int LONG_MAX_PARTS = 64;
// What the three kernels of a round's long searches hand each other (k_long_list -> k_long_scan -> k_long_fin,
// reorder_kernels.hip): one head per search of the round (index = its place in longq), its verified multi-read bins in
// lbin / lbcode (lbin_stride entries per search), one word per part in lparts.  A search whose first turn lists more than
// long_part chunks of 64 bin entries is cut into up to LONG_MAX_PARTS parts = ranges of its bins in priority order.
struct LongHead {
  uint32_t nb, nparts, best_single, bestrid;  // bins listed, parts, lowest code of a single-read bin that hit (+ its read)
  uint32_t bestpart, pad[3];                  // lowest part with a pass so far: a hint that lets the parts behind it stop
  uint32_t blo[LONG_MAX_PARTS + 4];           // part p scans bins [blo[p], blo[p + 1])
  uint32_t rescode[LONG_MAX_PARTS], resrid[LONG_MAX_PARTS];  // part p's first pass inside its bin's window: probe code (0x7fffffff: none), read
  uint32_t capped[LONG_MAX_PARTS];            // part p: a bin ahead of its pass (any bin, without one) stopped at the window
};
struct DevParams {
  // reads
  const uint64_t *reads;  // n * S limbs (S = limb stride, power of two >= W, <= 16)
  const uint16_t *lens;
  uint32_t n;
  int L, W, S, Lpad, maxshift, uniform_len;
  int plan[2][6];     // ordered probe batches of a search: widths in shifts (<= 16 each, sum <= 32), 0-terminated;
                      // [0] a chain in its stride, [1] a chain whose seed has no match yet; the tail covers the rest
  int seed_wide;      // 1: plan[1] is in use
  int deep_bins;      // 1: the dictionary has deep bins -> the kernel variants that trim dead bin tails while they scan
  int search_wpb;     // chains (wavefronts) per block of k_search: 1, 2 or 4
  int dbg_search_lds, dbg_apply_lds;   // occupancy experiments: dummy dynamic LDS bytes per block
  int wl;             // length of both dictionary windows in bases (dend - dstart + 1)
  // dictionaries (reorder.h:751-759)
  int dstart[2], dend[2];
  uint32_t numkeys[2];
  TabView tab;                // ONE table for both dictionaries (tab_find)
  const ulonglong2 *urec[2];  // {key, start | count<<32} per unique key (multi-read bins)
  const uint32_t *ids[2];
  // Liveness inside the bins (deep-bin pools, fused rounds; else null / 0xffffffff): epos[l][read] = index of the read's entry
  // in ids[l] (0xffffffff: none), and bit 31 of that entry is set when the read is taken (k_mg_mark, k_init_seeds) -- a bin
  // scan then knows a dead entry from the id it has loaded anyway instead of asking the bitmap (one random request per
  // entry: two thirds of k_long's requests on a bin of more than MAX_SEARCH entries).  idmask strips the bit (n < 2^31).
  uint32_t *epos[2];
  uint32_t idmask;
  const ulonglong2 *sig[2];   // k_long only (else null): {first limb, last limb} of the read of every entry of ids[l] (k_build_sig)
  // shared mutable state
  uint64_t *taken;    // bitmap, bit r set <=> read r claimed (== !remainingreads[r], reorder.h:343)
  uint32_t *ublk;     // untaken reads per block of 2^UBLK_SHIFT reads, exact between rounds (seed selection, find_seed)
  uint32_t *resv;     // lowest chain id that proposed read r this round (0xffffffff = none)
  uint32_t *needy;    // bitmap over chains waiting for a seed (padded with zero words to a multiple of 256 words)
  // rounds whose shared state is kept by k_mg_mark (fused rounds, multi-GPU pools; null in the two-kernel round):
  uint32_t *needy_cnt;       // set bits per 64 words of needy (2048 chains) as k_mg_mark of the LAST round left them
  uint32_t *needy_cnt_next;  // the buffer k_mg_mark of THIS round fills (zeroed by the previous k_mg_mark)
  int fused;                 // 1: k_round (apply + search in one kernel)
  int alts;                  // candidates per match proposal: 1, or 2 (the alternatives schedule; deep-bin kernel variants, fused rounds)
  int mc;                    // 1: four chains per wavefront (k_round_mc) where it applies
  int ka, ka_lo;             // ka = 1: k_round_mc's chains keep known-absent window masks (reorder_round_mc.h: search_ka); four limbs per
                             // strand live in Chain::revref[8..15] between rounds, the forward strand's from limb ka_lo
  // k_round_mc runs chains of one class per wavefront (the four chains of a wavefront take the union of their
  // paths): k_mg_mark sorts the running local chains of every block of MARK_BLOCK consecutive chain ids by what the
  // next round will ask of them -- 0 left search after a failed right search, 1 first search of a new seed, 2 search
  // after a proposed match, 3 seed pick -- into ord[block * MARK_BLOCK ...] (local chain indices, class 0 first) and
  // writes the four class sizes to ord_cnt[block].  No atomics, rewritten every round; done chains are in no list.
  uint32_t *ord;
  uint4 *ord_cnt;
  // long searches (deep-bin pools, k_long): [0] = searches k_round handed over this round, [1] = k_long's ticket counter, [2 + i] = their local chain
  // indices; k_mg_mark zeroes the count.  long_budget: 64-lane compare passes (balanced scan) / bin entries walked by
  // one lane (tail) a wavefront of k_round spends on a search before it hands it over; 0 = never.
  // long_min: bin entries that must still be ahead of the search at that point (else the wavefront carries on).
  uint32_t *longq;
  int long_budget, long_min, long_blocks, long_part;
  // lctl[0] = parts listed this round, [1] = k_long_scan's ticket counter (k_mg_mark zeroes both with the queue), [3] = split
  // searches of the run; lparts[i] = search << 6 | part
  uint32_t *lctl, *lparts;
  struct LongHead *lhead;
  uint2 *lbin;          // {start, count} of a listed bin
  uint16_t *lbcode;     // its probe code
  uint32_t lbin_stride;
#ifdef SR_PHASE_TIMING
  unsigned long long *dbg;  // [0..63] phase clocks / visits summed over the wavefronts that ran > 1M clocks, [64] how many
#endif
  Globals *glob;
  long long *cursor;  // the seed cursor the launch works with (&glob->cursor; the second chain group: &glob->cursor_b)
  // chains: this context owns global chains [c0, c0+K) of Ktot (single GPU: c0 = 0, Ktot = K)
  uint32_t K, c0, Ktot;
  // The two-group schedule (phases = 2; specification: orc_reorder_rounds_ph (the test suite's CPU restatement of the schedule)).
  // The chains run as two groups -- [0, Kh) and [Kh, K), Kh a multiple of 2048 -- whose rounds alternate: group g searches
  // on ITS view of the pool (taken = view_g: everything marked up to its own last mark step), while the other group's mark
  // step runs; its mark step (k_ph_mark) then resolves its proposals against the other group's view as well (taken_other:
  // a read the other group took in between is lost), sets its winners in its own view, leaves them in won[] for the other
  // group's next mark step, and folds the other group's last winners (won_other[]) into its own view.  So a launch of one
  // group never reads what a launch of the other group that may run beside it writes, and the two round kernels fill each
  // other's drain.  Seeds: group 0 from reads [seed_lo, n) downwards, group 1 from [0, seed_lo of group 0).
  // A launch covers the chains [g0, g0 + Kg) (phases = 1: all of them).
  // In a multi-GPU pool a rank owns a slice of EACH group (the groups are the same whatever the number of ranks): its local
  // chains [g0, g0 + Kg) are the global chains [c0 + g0, c0 + g0 + Kg) -- c0 is per launch, so that `cid = c0 + li` holds in
  // every kernel -- of the group's global range [gg0, gg0 + gKg) (one GPU: c0 = 0, gg0 = g0, gKg = Kg; one group: gg0 = 0,
  // gKg = Ktot).  The mark step runs over the group's global range on every rank (replicated, as k_mg_mark does).
  int phases;
  uint32_t g0, Kg, g0_other, Kg_other;
  uint32_t gg0, gKg, gKg_other;
  uint32_t seed_lo, seed_hi;   // this group's seeds are reads [seed_lo, seed_hi) (phases = 1: [0, n))
  uint32_t nb_lo, nb_hi;       // ... and its chains the blocks [nb_lo, nb_hi) of 2048 chains (needy_cnt)
  const uint64_t *taken_other; // the other group's view (k_ph_mark only)
  uint32_t *won, *won_other;   // [gKg] / [gKg_other]: read a chain secured in its group's last mark step | kind << 31 (1: a match), 0xffffffff none
  unsigned long long *prop;   // [Ktot] proposals of the round (multi-GPU mode only, else null)
  uint32_t *alive_wave;       // [ceil(Ktot / 64)] chains not done per 64 chains, rewritten every round by k_mg_mark
  Chain *chains;
  int4 *cnt;          // [K][2][Lpad] per-position counts (A,C,T,G), ping-pong: wide format (some count > 255)
  uint32_t *cnt8;     // same, one byte per count: the format of almost every update (4x fewer bytes moved)
  // append-order emission buffers.  A chain fills private CHUNK-slot chunks; a matched record is ONE 16-byte store
  // {read id, rc | flag << 8, pos}; e_chunk[c] = {owning chain (local index), sequence number of the chunk's first
  // record} is written once per chunk, so the final scatter needs no per-record tags.  Singletons: the read id.
  uint4 *e_rec; uint2 *e_chunk;
  uint32_t *s_rec; uint2 *s_chunk;
  // final streams (tid-major, chain ascending inside a tid)
  uint32_t *f_order; char *f_rc; char *f_flag; long long *f_pos; uint16_t *f_len; uint32_t *f_order_s;
};

// launchers (reorder_kernels.hip)
void launch_unpack(hipStream_t st, const uint8_t *dna, const uint64_t *off, uint32_t n, int L, int W, int S,
                   uint32_t rec_fixed, uint64_t *reads, uint16_t *lens, uint32_t *bad_len = nullptr);
void launch_flag_in_dict(hipStream_t st, const uint16_t *lens, uint32_t n, int dend, uint32_t *flag);
void launch_keys(hipStream_t st, const uint64_t *reads, const uint16_t *lens, const uint32_t *slot, uint32_t n,
                 int S, int dstart, int dend, uint64_t *keys, uint32_t *vals);
// unique keys of both dictionaries merged by hash (mval = dict << 63 | index of the key in its dictionary)
struct DictBuild {
  const uint32_t *ustart, *ucount, *ids;
  ulonglong2 *urec;
  uint32_t *deep, *ndeep;   // ndeep[0] bins listed in deep[], ndeep[1] reads in bins of >= BIG_BIN entries, ndeep[2] ... of >= MID_BIN
};
void launch_tab_insert(hipStream_t st, const uint64_t *mhash, const uint64_t *mval, uint64_t nmerged, DictBuild d0,
                       DictBuild d1, uint32_t *fpt, int bshift);
// minimizer-addressed table: bucket + {tag, payload} word of every merged entry (also writes the bin records), then,
// after a sort by bucket, the two insert passes
void launch_minz_prepare(hipStream_t st, const uint64_t *mhash, const uint64_t *mval, uint64_t nmerged, DictBuild d0,
                         DictBuild d1, int lshift, uint32_t *bucket, uint64_t *tagpay);
void launch_tab_insert_minz(hipStream_t st, const uint32_t *bucket_sorted, const uint64_t *tagpay_sorted, uint64_t nmerged,
                            uint32_t *fpt, int bshift, uint32_t *marked);
hipError_t sort_pairs_u32_u64(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint64_t *vin, uint64_t *vout, size_t n, unsigned end_bit);
hipError_t merge_by_hash(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint64_t *k0, const uint64_t *k1,
                         const uint64_t *v0, const uint64_t *v1, uint64_t *kout, uint64_t *vout, size_t n0, size_t n1);
void launch_iota_tag(hipStream_t st, uint64_t *v, uint64_t n, uint64_t tag);
void launch_build_epos(hipStream_t st, const uint32_t *ids, uint64_t m, uint32_t *epos);
void launch_trim_bins(hipStream_t st, const uint32_t *deep, const uint32_t *ndeep, uint32_t ndeep_host,
                      ulonglong2 *urec, const uint32_t *ids, const uint64_t *taken, ulonglong2 *sig /* or null: moved with the ids */,
                      uint32_t *epos = nullptr /* or null: follows the ids, and bit 31 of an id says "taken" */);
void launch_build_sig(hipStream_t st, const uint32_t *ids, uint64_t m, const uint64_t *reads, int S, int W, ulonglong2 *sig);
void launch_dict_lookup(hipStream_t st, TabView tab, const ulonglong2 *urec, int which,
                        const uint64_t *reads, int S, int dstart, int dend, const uint64_t *keys, uint32_t nkeys,
                        uint32_t *start, uint32_t *count);
void launch_fill_u32(hipStream_t st, uint32_t *p, uint64_t n, uint32_t v);
void launch_init_taken(hipStream_t st, uint64_t *taken, uint64_t nwords, uint32_t n, uint32_t *ublk);
void launch_init_chains(hipStream_t st, const DevParams &P, bool seeds = true);
void launch_check_seed_state(hipStream_t st, const DevParams &P, uint64_t nwords, unsigned long long *bad);
// two-kernel round (one GPU): search -> apply
void launch_search(hipStream_t st, const DevParams &P, bool stats);
void launch_apply(hipStream_t st, const DevParams &P, bool literal);
// fused round: round (apply of the last proposals + search, own chains) -> [multi-GPU: all-gather of prop, resolve] -> mark
void launch_round(hipStream_t st, const DevParams &P, bool stats, bool mg);
void launch_mg_resolve(hipStream_t st, const DevParams &P);
void launch_mg_mark(hipStream_t st, const DevParams &P);
void launch_ph_mark(hipStream_t st, const DevParams &P);       // mark step of one chain group (two-group schedule)
void launch_delay(hipStream_t st, uint32_t microseconds);      // a one-thread kernel that waits (staggers the two groups' first rounds)
void launch_chain_summary(hipStream_t st, const DevParams &P, uint2 *sum, unsigned long long *tot /* [8] */);
void launch_scatter(hipStream_t st, const DevParams &P, uint64_t cap_m, uint64_t cap_s, const uint64_t *off_m,
                    const uint64_t *off_s);
void launch_rec_size(hipStream_t st, const uint32_t *order, const uint16_t *lens, uint64_t cnt, uint32_t *sz);
void launch_emit_dna(hipStream_t st, const uint64_t *reads, const uint16_t *lens, int S, const uint32_t *order,
                     const char *rc, uint64_t cnt, const uint64_t *off, uint32_t rec_fixed, uint8_t *dst);
void launch_synth(hipStream_t st, uint8_t *dst, uint32_t n, uint32_t L, uint64_t G, uint64_t seed, uint32_t thr24);

// FASTQ front end (fastq_kernels.hip)
constexpr int NL_CHUNK_BYTES = 4096;
void launch_nl_count(hipStream_t st, const uint8_t *t, uint64_t nbytes, uint32_t *blk_cnt, uint64_t nblk);
void launch_nl_fill(hipStream_t st, const uint8_t *t, uint64_t nbytes, const uint64_t *blk_off, uint64_t *line_end,
                    uint64_t nblk);
void launch_read_info(hipStream_t st, const uint8_t *t, const uint64_t *line_end, uint64_t nreads, uint32_t *len,
                      uint32_t *fclean, uint32_t *szc, uint32_t *fN, uint32_t *szN, uint32_t *lenc, uint32_t *err);
void launch_pack_reads(hipStream_t st, const uint8_t *t, const uint64_t *line_end, uint64_t nreads, const uint32_t *len,
                       const uint32_t *fclean, const uint32_t *cidx, const uint64_t *coff, const uint32_t *nidx,
                       const uint64_t *noff, uint32_t cidx_base, uint64_t coff_base, uint32_t file_read_base,
                       uint8_t *out_clean, uint64_t *out_off, uint8_t *out_N, uint32_t *out_orderN, uint64_t *out_offN);
hipError_t reduce_max_u32(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint32_t *out, size_t n);
hipError_t reduce_min_u32(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint32_t *out, size_t n);

hipError_t sort_pairs(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout,
                      const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit);
hipError_t rle(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint64_t *in, size_t n, uint64_t *uniq,
               uint32_t *counts, uint32_t *nruns);
hipError_t excl_scan_u32(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint32_t *out, size_t n);
hipError_t excl_scan_u32_to_u64(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint64_t *out,
                                size_t n);

}  // namespace sr
#endif
