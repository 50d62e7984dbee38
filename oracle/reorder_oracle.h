/*
 * oracle/reorder_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of SPRING's read-reordering stage
 * (/root/reference/src/reorder.h + bitset_util.{h,cpp}).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (spring_amd/csrc) never links or calls it.
 *
 * Two algorithms are exported:
 *   orc_reorder_serial  -- literal single-thread restatement of
 *                          reorder_main<N>() at `-t 1` (mutable CSR bins with
 *                          the reference's tail-sentinel encoding).
 *   orc_reorder_rounds  -- the deterministic K-chain schedule this repo's GPU
 *                          path implements (immutable bins + taken[] flags,
 *                          lock-step rounds, lowest chain id wins a contested
 *                          read).  K = 1 must equal orc_reorder_serial
 *                          byte for byte (tests/test_oracle.py).
 *
 * Pinning status: see oracle/README.md ("parity partially pinned").
 */
#ifndef SPRING_ORACLE_REORDER_H_
#define SPRING_ORACLE_REORDER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_READ_LEN 511 /* params.h:22 */
#define ORC_WMAX 16          /* ceil(2*511/64) */

typedef struct {
  uint64_t unmatched;       /* contig seeds ("were unmatched", reorder.h:633) */
  uint64_t search_calls;    /* search_match() invocations                    */
  uint64_t probes;          /* dictionary probes that passed the bounds test  */
  uint64_t keyok;           /* probes whose key exists and bin is non-empty   */
  uint64_t cands;           /* Hamming evaluations                            */
  uint64_t hits;            /* Hamming <= THRESH_REORDER                      */
  uint64_t updates;         /* updaterefcount() calls                         */
  uint64_t iterations;      /* while(!done) iterations summed over chains     */
  uint64_t rounds;          /* lock-step rounds (rounds schedule only)        */
  uint64_t lost;            /* proposals that lost a contested read           */
} orc_stats;

/* limbs per read: (2*L-1)/64+1  (call_template_functions.cpp:10) */
int orc_limbs(int max_readlen);

/* readDnaFile (reorder.h:222-244): records of u16 len + ceil(len/4) bytes.
 * read must hold n*W zeroed limbs.  Returns bytes consumed or -1. */
int64_t orc_load_dna(const uint8_t *dna, size_t nbytes, uint32_t n, int max_readlen,
                     uint64_t *read, uint16_t *len);

/* write_dna_in_bits (util.cpp:269-294) for ACGT strings; returns bytes written. */
size_t orc_pack_read(const char *s, int len, uint8_t *dst);

/* Outputs (caller allocates n entries each).  Matched stream = what the
 * reference writes to read_order.bin.<tid>/read_rev.txt.<tid>/tempflag.txt.<tid>
 * /temppos.txt.<tid>/read_lengths.bin.<tid>, concatenated in tid order
 * (tid_off[num_thr+1] delimits them).  order_s = read_order.bin.singleton. */
typedef struct {
  uint32_t *order;
  char *rc;
  char *flag;
  int64_t *pos;
  uint16_t *rlen;
  uint32_t *order_s;
  uint64_t *tid_off;   /* num_thr+1 entries */
  uint64_t *tid_off_s; /* num_thr+1 entries (singleton stream per tid) */
  uint64_t n_matched;
  uint64_t n_single;
} orc_out;

int orc_reorder_serial(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen,
                       orc_out *out, orc_stats *st);

/* orc_reorder_serial with every step cross-checked by callbacks that run the REAL reference functions on a mirrored
 * state (oracle/ref_units_driver.cpp::ref_shadow_*); mm[10] = mismatch / call counters, see reorder_oracle.c */
typedef struct {
  void *user;
  int (*claim_first)(void *user, uint32_t current);
  int (*remove)(void *user, uint32_t current);
  int (*search)(void *user, const uint64_t *ref_shifted, int rev, int shift, int ref_len, int flag, uint32_t k);
  int (*update)(void *user, uint32_t rid, int reset, int rev, int shift, const int32_t *cnt, int stride,
                const uint64_t *ref, const uint64_t *revref, int ref_len);
  int64_t (*pick_seed)(void *user);
} orc_shadow;
int orc_reorder_serial_shadow(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen,
                              orc_out *out, orc_stats *st, const orc_shadow *sh, uint64_t *mm);

int orc_reorder_rounds(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen,
                       uint32_t num_chains, int num_thr, orc_out *out, orc_stats *st);
/* the same schedule with A candidates per match proposal resolved in A passes (A = 1: orc_reorder_rounds) */
int orc_reorder_rounds_alt(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen,
                       uint32_t num_chains, int num_thr, int alternatives, orc_out *out, orc_stats *st);

/* the schedule with two chain groups whose rounds alternate (spring_reorder_opts::phases = 2; see reorder_oracle.c):
 * group 0 = chains [0, orc_phase_split(K)), group 1 the rest; needs K >= 4096 and n >= 8192; _alt: with A candidates per match proposal */
uint32_t orc_phase_split(uint32_t num_chains);
int orc_reorder_rounds_ph(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen,
                          uint32_t num_chains, int num_thr, orc_out *out, orc_stats *st);
int orc_reorder_rounds_ph_alt(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen,
                              uint32_t num_chains, int num_thr, int alternatives, orc_out *out, orc_stats *st);

/* Replay check of ANY reorder output (any chain count / schedule, any size; OpenMP over the contigs): every contig's records
 * re-derived with the reference's state machine -- each matched read at its recorded orientation and position must be one
 * search_match would have accepted on the consensus its predecessors built (see reorder_oracle.c).  res[4] = contigs, matched
 * records, contigs that do not verify, index of the first bad record (~0: none). */
int orc_check_contigs(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen, const uint32_t *order, const char *rc,
                      const char *flag, const int64_t *pos, uint64_t n_matched, const uint64_t *tid_off, int num_thr, uint64_t *res);
/* ... with the consensus kept by `upd` instead of the restatement: the reference's own updaterefcount<N> compiled in place
 * (oracle/_ref/libref_units.so::ref_u_updaterefcount has this signature; cnt = int32 [4][stride], rows A C T G) */
typedef int (*orc_update_fn)(int L, const uint64_t *cur, int32_t *cnt, int stride, uint64_t *ref, uint64_t *revref, int *ref_len,
                             int reset, int rev, int shift, int cur_readlen);
int orc_check_contigs_upd(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen, const uint32_t *order, const char *rc,
                          const char *flag, const int64_t *pos, uint64_t n_matched, const uint64_t *tid_off, int num_thr, uint64_t *res,
                          orc_update_fn upd);

/* CPU-baseline port: T free-running OpenMP threads like the reference's `-t T`; NOT deterministic
 * for T > 1 (like the reference).  Outputs laid out per thread (tid_off has T+1 entries). */
int orc_reorder_omp(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen, int num_threads,
                    orc_out *out, orc_stats *st);

/* writetofile (reorder.h:643-730): byte stream of temp.dna.<tid> for a slice
 * of the matched stream (rc may be NULL => all 'd', i.e. the singleton file).
 * Returns bytes written into dst (capacity must be >= cnt*(2+ceil(L/4))). */
size_t orc_write_dna_stream(const uint64_t *read, const uint16_t *len, int max_readlen,
                            const uint32_t *order, const char *rc, uint64_t cnt, uint8_t *dst);

/* constructdictionary (bitset_util.h:74-221) restated: returns arrays so the
 * test can compare with the real reference build in oracle/_ref.
 * keys_out: numkeys sorted unique keys; startpos_out: numkeys+1; read_id_out:
 * dict_numreads.  Buffers sized n (+1).  Returns numkeys, *dict_numreads set. */
uint32_t orc_build_dict(const uint64_t *read, const uint16_t *len, uint32_t n, int max_readlen,
                        int which, uint64_t *keys_out, uint32_t *startpos_out,
                        uint32_t *read_id_out, uint32_t *dict_numreads);

/* bbhashdict::findpos/remove (bitset_util.cpp:20-63) on one bin laid out in
 * read_id[0..cap) with startpos {0,cap}.  Used to pin the tail encoding
 * against the real reference.  Returns live count after the call. */
int64_t orc_bin_remove(uint32_t *read_id, uint32_t cap, uint8_t *empty_bin, int64_t current);
int64_t orc_bin_live(const uint32_t *read_id, uint32_t cap);

/* dictionary windows (reorder.h:751-759) */
void orc_dict_windows(int max_readlen, int start[2], int end[2]);

#ifdef __cplusplus
}
#endif
/* pe_encode (pe_encode.cpp:24-84) */
void orc_pe_encode(const uint32_t *order, uint32_t numreads, uint32_t *order_array);

#endif
