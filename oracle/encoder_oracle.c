/*
 * oracle/encoder_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of SPRING's encoder stage at `-t 1` (SURVEY.md section 8
 * row f2): encoder_main<N>/encode<N> (encoder.h:124-494, :572-633), buildcontig
 * and writecontig (encoder.cpp:32-109), readsingletons (encoder.h:541-570),
 * correct_order (encoder.cpp:177-222), the 2-bit packing of pack_compress_seq
 * (encoder.cpp:111-156, without the BSC call) -- on in-memory images of the
 * stage's input files.  Per-tid input streams are processed tid 0, 1, ... one
 * after the other, which is what the reference does at `-t 1` and one legal
 * interleaving of what it does at `-t T`.
 *
 * Pinning status: encoder.h needs Boost.Iostreams (absent here) -> not
 * buildable; the shared primitives (constructdictionary with bpb = 3,
 * generatemasks with bpb = 3, findpos/remove) are pinned against the real
 * bitset_util build in tests/test_oracle_vs_ref.py.  "parity partially pinned".
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "encoder_oracle.h"
#include "orc_internal.h"

#define MAX_SEARCH_ENCODER 1000 /* params.h:33 */
#define THRESH_ENCODER 24       /* params.h:34 */
#define CONTIG_LIST_LIMIT 10000000u /* encoder.h:215 */
#define ENC_WMAX 24             /* 1536 bits (call_template_functions.cpp:66-140) */

/* ---------------------------------------------------------- small helpers */

static char revchar(char c) { /* util.h chartorevchar */
  switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}
static void revcomp(const char *s, char *d, int n) { /* util.cpp:376-381 */
  for (int j = 0; j < n; j++) d[j] = revchar(s[n - j - 1]);
}

/* 3-bit code of a base: basemask of setglobalarrays (encoder.h:499-519); bit k of
 * the value is bit 3i+k of the bitset: A=0, N=1, G=2, C=4, T=6 */
static inline unsigned code3(char c) {
  switch (c) { case 'A': return 0; case 'C': return 4; case 'G': return 2; case 'T': return 6; default: return 1; }
}
static void string_to_bits3(const char *s, int n, uint64_t *b, int W) { /* stringtobitset, bitset_util.h:57-62 */
  memset(b, 0, sizeof(uint64_t) * W);
  for (int i = 0; i < n; i++) {
    unsigned v = code3(s[i]);
    for (int k = 0; k < 3; k++)
      if (v >> k & 1) b[(3 * i + k) >> 6] |= 1ull << ((3 * i + k) & 63);
  }
}
static void bits3_to_string(const uint64_t *b, int W, char *s, int n) { /* bitsettostring, encoder.h:107-122 */
  static const char revinttochar[8] = {'A', 'N', 'G', 0, 'C', 0, 'T', 0};
  for (int j = 0; j < n; j++) s[j] = revinttochar[orc__window64(b, W, 3 * j, 3)];
}
/* ((a ^ b) & mask[0][L-len]).count(): bases [0,len) of 3-bit bitsets (generatemasks, bitset_util.h:223-236) */
static int hamming3(const uint64_t *a, const uint64_t *b, int W, int len) {
  int bhi = 3 * len, c = 0;
  for (int i = 0; i < W && i * 64 < bhi; i++) {
    uint64_t x = a[i] ^ b[i];
    if (bhi < i * 64 + 64) x &= (1ULL << (bhi - i * 64)) - 1;
    c += __builtin_popcountll(x);
  }
  return c;
}
static void shr3(uint64_t *b, int W) {
  for (int i = 0; i < W; i++) b[i] = (b[i] >> 3) | (i + 1 < W ? b[i + 1] << 61 : 0);
}
static void shl3(uint64_t *b, int W) {
  for (int i = W - 1; i >= 0; i--) b[i] = (b[i] << 3) | (i ? b[i - 1] >> 61 : 0);
}
static void mask_bits(uint64_t *b, int W, int nbits) { /* & mask[0][0] */
  for (int i = 0; i < W; i++) {
    if (i * 64 >= nbits) b[i] = 0;
    else if (nbits < i * 64 + 64) b[i] &= (1ULL << (nbits - i * 64)) - 1;
  }
}
static void or_base(uint64_t *b, int pos, char c) { /* |= basemask[pos][c] */
  unsigned v = code3(c);
  for (int k = 0; k < 3; k++)
    if (v >> k & 1) b[(3 * pos + k) >> 6] |= 1ull << ((3 * pos + k) & 63);
}

/* enc_noise (encoder.h:522-541) */
static char enc_noise(char ref, char rd) {
  static const char *row_A = "C0G1T2N3", *row_C = "A0G1T2N3", *row_G = "T0A1C2N3", *row_T = "G0C1A2N3",
                    *row_N = "A0G1C2T3";
  const char *row = ref == 'A' ? row_A : ref == 'C' ? row_C : ref == 'G' ? row_G : ref == 'T' ? row_T : row_N;
  for (int i = 0; i < 8; i += 2)
    if (row[i] == rd) return row[i + 1];
  return 0;
}

/* growable byte buffer */
typedef struct { uint8_t *p; size_t n, cap; } buf_t;
static void buf_put(buf_t *b, const void *src, size_t k) {
  if (b->n + k > b->cap) {
    size_t c = b->cap ? b->cap * 2 : 4096;
    while (c < b->n + k) c *= 2;
    b->p = (uint8_t *)realloc(b->p, c);
    b->cap = c;
  }
  memcpy(b->p + b->n, src, k);
  b->n += k;
}

/* write_dnaN_in_bits (util.cpp:322-348) */
static void put_dnaN(buf_t *b, const char *s, int n) {
  uint16_t rl = (uint16_t)n;
  uint8_t arr[256];
  int nb = (n + 1) / 2;
  memset(arr, 0, sizeof(arr));
  for (int i = 0; i < n; i++) {
    unsigned v = s[i] == 'A' ? 0 : s[i] == 'C' ? 2 : s[i] == 'G' ? 1 : s[i] == 'T' ? 3 : 4;
    arr[i / 2] |= (uint8_t)(v << (4 * (i & 1)));
  }
  buf_put(b, &rl, 2);
  buf_put(b, arr, nb);
}

/* contig_reads (encoder.h:80-86) */
typedef struct { char *read; int64_t pos; char rc; uint32_t order; uint16_t len; } cread_t;

/* std::list::sort is a stable merge sort (encoder.h:222-224, :351-353) */
static void stable_sort_pos(cread_t *a, cread_t *tmp, size_t n) {
  if (n < 2) return;
  size_t h = n / 2;
  stable_sort_pos(a, tmp, h);
  stable_sort_pos(a + h, tmp, n - h);
  size_t i = 0, j = h, k = 0;
  while (i < h && j < n) tmp[k++] = (a[j].pos < a[i].pos) ? a[j++] : a[i++];
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, sizeof(cread_t) * n);
}

/* buildcontig (encoder.cpp:32-74); reads are clean (no N) */
static char *buildcontig(const cread_t *c, size_t list_size, size_t *ref_size) {
  static const char longtochar[4] = {'A', 'C', 'G', 'T'};
  if (list_size == 1) {
    char *r = (char *)malloc(c[0].len + 1);
    memcpy(r, c[0].read, c[0].len);
    *ref_size = c[0].len;
    return r;
  }
  int64_t currentpos = 0, currentsize = 0, to_insert;
  size_t cap = 1024;
  long(*count)[4] = (long(*)[4])calloc(cap, sizeof(long[4]));
  for (size_t k = 0; k < list_size; k++) {
    if (k == 0)
      to_insert = c[k].len;
    else {
      currentpos = c[k].pos;
      to_insert = currentpos + c[k].len > currentsize ? currentpos + c[k].len - currentsize : 0;
    }
    if ((size_t)(currentsize + to_insert) > cap) {
      size_t nc = cap;
      while (nc < (size_t)(currentsize + to_insert)) nc *= 2;
      count = (long(*)[4])realloc(count, nc * sizeof(long[4]));
      memset(count + cap, 0, (nc - cap) * sizeof(long[4]));
      cap = nc;
    }
    currentsize += to_insert;
    for (int i = 0; i < c[k].len; i++) {
      char ch = c[k].read[i];
      int v = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
      count[currentpos + i][v] += 1;
    }
  }
  char *ref = (char *)malloc((size_t)currentsize + 1);
  for (int64_t i = 0; i < currentsize; i++) {
    long max = 0, indmax = 0;
    for (long j = 0; j < 4; j++)
      if (count[i][j] > max) { max = count[i][j]; indmax = j; }
    ref[i] = longtochar[indmax];
  }
  free(count);
  *ref_size = (size_t)currentsize;
  return ref;
}

typedef struct {
  buf_t seq, pos, noise, noisepos, order, rc, rlen;
} streams_t;

/* writecontig (encoder.cpp:76-109) */
static void writecontig(const char *ref, size_t ref_size, const cread_t *c, size_t n, streams_t *o, uint64_t *abs_pos) {
  buf_put(&o->seq, ref, ref_size);
  for (size_t k = 0; k < n; k++) {
    long currentpos = (long)c[k].pos, prevj = 0;
    for (long j = 0; j < c[k].len; j++)
      if (c[k].read[j] != ref[currentpos + j]) {
        char e = enc_noise(ref[currentpos + j], c[k].read[j]);
        buf_put(&o->noise, &e, 1);
        uint16_t pv = (uint16_t)(j - prevj);
        buf_put(&o->noisepos, &pv, 2);
        prevj = j;
      }
    buf_put(&o->noise, "\n", 1);
    uint64_t ap = *abs_pos + (uint64_t)currentpos;
    buf_put(&o->pos, &ap, 8);
    buf_put(&o->order, &c[k].order, 4);
    buf_put(&o->rlen, &c[k].len, 2);
    buf_put(&o->rc, &c[k].rc, 1);
  }
  *abs_pos += ref_size;
}

int orc_encode(const orc_enc_in *in, orc_enc_out *out) {
  memset(out, 0, sizeof(*out));
  const int L = in->max_readlen, T = in->num_thr;
  const int W2 = (2 * L - 1) / 64 + 1;       /* clean-read limbs, 2 bits per base */
  const int W = (3 * L - 1) / 64 + 1;        /* call_template_functions.cpp:65 */
  if (W > ENC_WMAX) return -1;
  const uint32_t ns = in->numreads_s, nN = in->numreads_N, np = ns + nN;

  /* ---- readsingletons (encoder.h:541-570) */
  uint64_t *sread = (uint64_t *)calloc((size_t)(np ? np : 1) * W, 8);
  uint16_t *slen = (uint16_t *)calloc(np ? np : 1, 2);
  uint32_t *order_s = (uint32_t *)calloc(np ? np : 1, 4);
  char s[1024], s1[1024];
  for (uint32_t i = 0; i < ns; i++) {
    uint32_t rid = in->order_s[i];
    static const char dec[4] = {'A', 'G', 'C', 'T'}; /* util.cpp:296-320 */
    int n = in->len[rid];
    for (int j = 0; j < n; j++) s[j] = dec[orc__window64(in->read + (size_t)rid * W2, W2, 2 * j, 2)];
    slen[i] = (uint16_t)n;
    string_to_bits3(s, n, sread + (size_t)i * W, W);
    order_s[i] = rid;
  }
  {
    const uint8_t *p = in->dnaN; /* read_dnaN_from_bits, util.cpp:350-374 */
    static const char int2dna[5] = {'A', 'G', 'C', 'T', 'N'};
    for (uint32_t i = ns; i < np; i++) {
      uint16_t n;
      memcpy(&n, p, 2);
      p += 2;
      for (int j = 0; j < n; j++) {
        unsigned v = (p[j / 2] >> (4 * (j & 1))) & 15;
        s[j] = int2dna[v > 4 ? 4 : v];
      }
      p += (n + 1) / 2;
      slen[i] = n;
      string_to_bits3(s, n, sread + (size_t)i * W, W);
      order_s[i] = in->order_N[i - ns];
    }
  }

  /* ---- correct_order (encoder.cpp:177-222): singletons and the per-tid order files */
  const uint32_t n_clean = in->n_clean, numreads_total = n_clean + nN;
  uint32_t *cumN = (uint32_t *)malloc(sizeof(uint32_t) * (n_clean ? n_clean : 1));
  {
    uint8_t *flagN = (uint8_t *)calloc(numreads_total ? numreads_total : 1, 1);
    for (uint32_t i = 0; i < nN; i++) flagN[order_s[ns + i]] = 1;
    uint32_t pc = 0, cnt = 0;
    for (uint32_t i = 0; i < numreads_total; i++) {
      if (flagN[i]) cnt++;
      else cumN[pc++] = cnt;
    }
    free(flagN);
  }
  for (uint32_t i = 0; i < ns; i++) order_s[i] += cumN[order_s[i]];

  /* ---- dictionaries (encoder.h:606-621) */
  dict_t dict[2];
  memset(dict, 0, sizeof(dict));
  {
    int ds[2], de[2];
    orc_enc_dict_windows(L, ds, de);
    for (int l = 0; l < 2; l++) { dict[l].start = ds[l]; dict[l].end = de[l]; }
  }
  dict[0].bpb = dict[1].bpb = 3;
  if (np > 0)
    for (int l = 0; l < 2; l++) orc__dict_build(&dict[l], sread, slen, np, W);

  /* ---- encode (encoder.h:124-363) */
  uint8_t *remaining = (uint8_t *)malloc(np ? np : 1);
  memset(remaining, 1, np ? np : 1);
  streams_t o;
  memset(&o, 0, sizeof(o));
  uint64_t *seq_len_tid = (uint64_t *)calloc(T, 8);
  size_t ccap = 1024, csz = 0;
  cread_t *contig = (cread_t *)malloc(sizeof(cread_t) * ccap), *tmp = (cread_t *)malloc(sizeof(cread_t) * ccap);
  uint64_t nprobe = 0, nhit = 0, ncontig = 0;
  uint64_t pos_base = 0; /* file_len_seq_thr prefix (encoder.h:465-480) */
  for (int tid = 0; tid < T; tid++) {
    uint64_t abs_pos = pos_base; /* per-tid abs_pos + the prefix added at encoder.h:465-480 */
    size_t seq_before = o.seq.n;
    csz = 0;
    for (uint64_t i = in->tid_off[tid];; i++) {
      int done = (i >= in->tid_off[tid + 1]);
      char c = done ? '0' : in->flag[i];
      if (c == '0' || done || csz > CONTIG_LIST_LIMIT) {
        if (csz != 0) {
          ncontig++;
          stable_sort_pos(contig, tmp, csz);
          int64_t first_pos = contig[0].pos;
          for (size_t k = 0; k < csz; k++) contig[k].pos -= first_pos;
          size_t ref_size;
          char *ref = buildcontig(contig, csz, &ref_size);
          if ((int64_t)ref_size >= L && np > 0) {
            uint64_t fwd[ENC_WMAX], rev[ENC_WMAX];
            string_to_bits3(ref, L, fwd, W);
            revcomp(ref, s1, L);
            string_to_bits3(s1, L, rev, W);
            for (long j = 0; j < (long)ref_size - L + 1; j++) {
              for (int r = 0; r < 2; r++) {
                for (int l = 0; l < 2; l++) {
                  const uint64_t *b = r ? rev : fwd;
                  uint64_t ull = orc__window64(b, W, 3 * dict[l].start, 3 * (dict[l].end - dict[l].start + 1));
                  int64_t idx = orc__dict_lookup(&dict[l], ull); /* bphf->lookup + key check (:264-284) */
                  nprobe++;
                  if (idx < 0) continue;
                  if (dict[l].empty_bin[idx]) continue;
                  int64_t di[2];
                  orc__findpos(&dict[l], di, (uint64_t)idx);
                  uint32_t del[MAX_SEARCH_ENCODER];
                  int ndel = 0;
                  for (int64_t k = di[1] - 1; k >= di[0] && k >= di[1] - MAX_SEARCH_ENCODER; k--) {
                    uint32_t rid = dict[l].read_id[k];
                    int h = hamming3(b, sread + (size_t)rid * W, W, slen[rid]);
                    if (h <= THRESH_ENCODER && remaining[rid]) {
                      remaining[rid] = 0;
                      nhit++;
                      if (csz + 1 > ccap) {
                        ccap *= 2;
                        contig = (cread_t *)realloc(contig, sizeof(cread_t) * ccap);
                        tmp = (cread_t *)realloc(tmp, sizeof(cread_t) * ccap);
                      }
                      cread_t *cr = &contig[csz++];
                      cr->read = (char *)malloc(slen[rid]);
                      bits3_to_string(sread + (size_t)rid * W, W, s, slen[rid]);
                      if (r) revcomp(s, cr->read, slen[rid]);
                      else memcpy(cr->read, s, slen[rid]);
                      cr->pos = r ? (j + L - slen[rid]) : j;
                      cr->rc = r ? 'r' : 'd';
                      cr->order = order_s[rid];
                      cr->len = slen[rid];
                      del[ndel++] = rid;
                    }
                  }
                  for (int d = 0; d < ndel; d++) /* delete from dictionaries (:325-341) */
                    for (int l1 = 0; l1 < 2; l1++)
                      if (slen[del[d]] > dict[l1].end) {
                        const uint64_t *rb = sread + (size_t)del[d] * W;
                        uint64_t key = orc__window64(rb, W, 3 * dict[l1].start, 3 * (dict[l1].end - dict[l1].start + 1));
                        int64_t ix = orc__dict_lookup(&dict[l1], key);
                        int64_t dj[2];
                        orc__findpos(&dict[l1], dj, (uint64_t)ix);
                        orc__bin_remove(&dict[l1], dj, (uint64_t)ix, del[d]);
                      }
                }
              }
              if (j != (long)ref_size - L) { /* shift bitsets (:344-357) */
                shr3(fwd, W);
                mask_bits(fwd, W, 3 * L);
                or_base(fwd, L - 1, ref[j + L]);
                shl3(rev, W);
                mask_bits(rev, W, 3 * L);
                or_base(rev, 0, revchar(ref[j + L]));
              }
            }
          }
          stable_sort_pos(contig, tmp, csz);
          writecontig(ref, ref_size, contig, csz, &o, &abs_pos);
          free(ref);
          for (size_t k = 0; k < csz; k++) free(contig[k].read);
        }
        csz = 0; /* the next contig starts with this record (:365-368) */
      }
      if (done) break;
      /* record i -> current (temp.dna.<tid> holds the read with RC already applied, reorder.h:652-678) */
      if (csz + 1 > ccap) {
        ccap *= 2;
        contig = (cread_t *)realloc(contig, sizeof(cread_t) * ccap);
        tmp = (cread_t *)realloc(tmp, sizeof(cread_t) * ccap);
      }
      uint32_t rid = in->order[i];
      static const char dec[4] = {'A', 'G', 'C', 'T'};
      int n = in->rlen[i];
      for (int j = 0; j < n; j++) s[j] = dec[orc__window64(in->read + (size_t)rid * W2, W2, 2 * j, 2)];
      cread_t *cr = &contig[csz++];
      cr->read = (char *)malloc(n ? n : 1);
      if (in->rc[i] == 'r') revcomp(s, cr->read, n);
      else memcpy(cr->read, s, n);
      cr->pos = in->pos[i];
      cr->rc = in->rc[i];
      cr->order = rid + cumN[rid]; /* correct_order on read_order.bin.<tid> */
      cr->len = (uint16_t)n;
    }
    seq_len_tid[tid] = o.seq.n - seq_before;
    pos_base += seq_len_tid[tid];
  }
  free(contig); free(tmp);

  out->seq = (char *)o.seq.p; out->seq_len = o.seq.n; out->seq_len_tid = seq_len_tid;
  out->pos = (uint64_t *)o.pos.p; out->noise = (char *)o.noise.p; out->noise_len = o.noise.n;
  out->noisepos = (uint16_t *)o.noisepos.p; out->n_noisepos = o.noisepos.n / 2;
  out->n_aligned = o.rc.n;
  out->rc = (char *)o.rc.p;
  /* remaining singletons appended to order / readlength, unaligned text (encoder.h:425-452) */
  buf_t un = {0, 0, 0};
  uint32_t matched_s = ns, matched_N = nN;
  uint64_t len_unaligned = 0;
  for (uint32_t i = 0; i < np; i++)
    if (remaining[i]) {
      if (i < ns) matched_s--; else matched_N--;
      buf_put(&o.order, &order_s[i], 4);
      buf_put(&o.rlen, &slen[i], 2);
      bits3_to_string(sread + (size_t)i * W, W, s, slen[i]);
      put_dnaN(&un, s, slen[i]);
      len_unaligned += slen[i];
    }
  out->order = (uint32_t *)o.order.p; out->rlen = (uint16_t *)o.rlen.p; out->n_total = o.order.n / 4;
  out->unaligned = un.p; out->unaligned_bytes = un.n; out->len_unaligned = len_unaligned;
  out->matched_s = matched_s; out->matched_N = matched_N;
  out->num_contigs = ncontig; out->num_probes = nprobe; out->num_hits = nhit;
  free(sread); free(slen); free(order_s); free(cumN); free(remaining);
  orc__dict_free(&dict[0]); orc__dict_free(&dict[1]);
  return 0;
}

void orc_encode_free(orc_enc_out *o) {
  free(o->seq); free(o->seq_len_tid); free(o->pos); free(o->noise); free(o->noisepos); free(o->order);
  free(o->rlen); free(o->rc); free(o->unaligned);
  memset(o, 0, sizeof(*o));
}

/* pack_compress_seq (encoder.cpp:111-156) minus the BSC call: 4 bases per byte, A0 C1 G2 T3,
 * first base in the low bits; the len%4 tail stays text */
uint64_t orc_pack_seq(const char *seq, uint64_t len, uint8_t *packed, char *tail) {
  for (uint64_t i = 0; i < len / 4; i++) {
    unsigned v = 0;
    for (int k = 0; k < 4; k++) {
      char c = seq[4 * i + k];
      unsigned b = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
      v |= b << (2 * k);
    }
    packed[i] = (uint8_t)v;
  }
  for (uint64_t k = 0; k < len % 4; k++) tail[k] = seq[len / 4 * 4 + k];
  return len / 4;
}


/* ---- one contig through buildcontig + writecontig, exposed for tests/test_oracle_vs_ref_units.py (pinned against the
 * REAL encoder.cpp:32-109).  reads: NUL-separated strings; out receives the seven streams back to back, sizes[7] =
 * seq, pos, noise, noisepos, order, RC, readlength. */
long orc_enc_contig(const char *reads, const int64_t *pos, const char *rc, const uint32_t *order, uint32_t count,
                    uint64_t *abs_pos, uint8_t *out, long cap, uint64_t *sizes) {
  cread_t *c = (cread_t *)calloc(count ? count : 1, sizeof(cread_t));
  const char *p = reads;
  for (uint32_t i = 0; i < count; i++) {
    size_t n = strlen(p);
    c[i].read = (char *)p;
    c[i].len = (uint16_t)n;
    c[i].pos = pos[i];
    c[i].rc = rc[i];
    c[i].order = order[i];
    p += n + 1;
  }
  streams_t o;
  memset(&o, 0, sizeof(o));
  size_t ref_size = 0;
  char *ref = buildcontig(c, count, &ref_size);
  writecontig(ref, ref_size, c, count, &o, abs_pos);
  free(ref);
  buf_t *bs[7] = {&o.seq, &o.pos, &o.noise, &o.noisepos, &o.order, &o.rc, &o.rlen};
  long w = 0;
  for (int i = 0; i < 7; i++) {
    if (w + (long)bs[i]->n > cap) { w = -1; break; }
    memcpy(out + w, bs[i]->p, bs[i]->n);
    sizes[i] = bs[i]->n;
    w += (long)bs[i]->n;
  }
  for (int i = 0; i < 7; i++) free(bs[i]->p);
  free(c);
  return w;
}

/* ---- bpb = 3 primitives exposed for tests/test_oracle_vs_ref.py (pinned against the real bitset_util) */
void orc_enc_bits3(const char *s, int n, uint64_t *b, int W) { string_to_bits3(s, n, b, W); }
int orc_enc_hamming3(const uint64_t *a, const uint64_t *b, int W, int len) { return hamming3(a, b, W, len); }
void orc_enc_dict_windows(int L, int start[2], int end[2]) { /* encoder.h:606-616 */
  if (L > 50) { start[0] = 0; end[0] = 20; start[1] = 21; end[1] = 41; }
  else { start[0] = 0; end[0] = 20 * L / 50; start[1] = 20 * L / 50 + 1; end[1] = 41 * L / 50; }
}
uint32_t orc_enc_build_dict(const uint64_t *read3, const uint16_t *len, uint32_t n, int L, int which, uint64_t *keys_out,
                            uint32_t *startpos_out, uint32_t *read_id_out, uint32_t *dict_numreads) {
  int s[2], e[2];
  orc_enc_dict_windows(L, s, e);
  dict_t d;
  memset(&d, 0, sizeof(d));
  d.start = s[which]; d.end = e[which]; d.bpb = 3;
  orc__dict_build(&d, read3, len, n, (3 * L - 1) / 64 + 1);
  memcpy(keys_out, d.keys, sizeof(uint64_t) * d.numkeys);
  memcpy(startpos_out, d.startpos, sizeof(uint32_t) * ((size_t)d.numkeys + 1));
  memcpy(read_id_out, d.read_id, sizeof(uint32_t) * d.dict_numreads);
  *dict_numreads = d.dict_numreads;
  uint32_t nk = d.numkeys;
  orc__dict_free(&d);
  return nk;
}
