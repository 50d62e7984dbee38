/* oracle/encoder_oracle.h -- TEST INFRASTRUCTURE (see encoder_oracle.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it. */
#ifndef ENCODER_ORACLE_H_
#define ENCODER_ORACLE_H_
#include <stddef.h>
#include <stdint.h>

/* In-memory image of the encoder stage's inputs (files of encoder.h:580-593) */
typedef struct {
  int max_readlen, num_thr;
  /* clean-read pool as the reorder stage holds it (2 bits/base, (2L-1)/64+1 limbs per read) */
  const uint64_t *read;
  const uint16_t *len;
  uint32_t n_clean;
  /* reorder output streams, tid-major: records tid_off[t] .. tid_off[t+1] belong to tid t */
  const uint64_t *tid_off; /* num_thr + 1 */
  const uint32_t *order;   /* read_order.bin.<tid> (index into the clean pool) */
  const char *rc, *flag;   /* read_rev.txt.<tid>, tempflag.txt.<tid> */
  const int64_t *pos;      /* temppos.txt.<tid> */
  const uint16_t *rlen;    /* read_lengths.bin.<tid> */
  /* singletons: read_order.bin.singleton (temp.dna.singleton is read[order_s[i]]) */
  const uint32_t *order_s;
  uint32_t numreads_s;
  /* reads with N: input_N.dna records (util.cpp:322-348) and read_order_N.bin */
  const uint8_t *dnaN;
  const uint32_t *order_N;
  uint32_t numreads_N;
} orc_enc_in;

typedef struct {
  char *seq;              /* read_seq.bin.<tid> texts, tid-major */
  uint64_t seq_len;
  uint64_t *seq_len_tid;  /* num_thr */
  uint64_t *pos;          /* read_pos.bin (absolute), n_aligned */
  char *noise;            /* read_noise.txt */
  uint64_t noise_len;
  uint16_t *noisepos;     /* read_noisepos.bin */
  uint64_t n_noisepos;
  uint32_t *order;        /* read_order.bin: n_aligned entries then the unaligned ones (n_total) */
  uint16_t *rlen;         /* read_lengths.bin, n_total */
  char *rc;               /* read_rev.txt, n_aligned */
  uint64_t n_aligned, n_total;
  uint8_t *unaligned;     /* read_unaligned.txt */
  uint64_t unaligned_bytes, len_unaligned;
  uint32_t matched_s, matched_N;
  uint64_t num_contigs, num_probes, num_hits;
} orc_enc_out;

int orc_encode(const orc_enc_in *in, orc_enc_out *out);
void orc_encode_free(orc_enc_out *out);
void orc_enc_bits3(const char *s, int n, uint64_t *b, int W);
int orc_enc_hamming3(const uint64_t *a, const uint64_t *b, int W, int len);
void orc_enc_dict_windows(int L, int start[2], int end[2]);
uint32_t orc_enc_build_dict(const uint64_t *read3, const uint16_t *len, uint32_t n, int L, int which, uint64_t *keys_out,
                            uint32_t *startpos_out, uint32_t *read_id_out, uint32_t *dict_numreads);
uint64_t orc_pack_seq(const char *seq, uint64_t len, uint8_t *packed, char *tail);
#endif
