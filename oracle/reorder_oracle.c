/*
 * oracle/reorder_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of SPRING's reorder stage.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/src).
 * See reorder_oracle.h for who may call this and oracle/README.md for the
 * pinning status.
 */
#include "reorder_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_SEARCH_REORDER 1000           /* params.h:26 */
#define THRESH_REORDER 4                  /* params.h:27 */
#define MAX_NUM_READS 4294967290u         /* params.h:24 */
#define STOP_CRITERIA_REORDER 0.5f        /* params.h:30 */

/* ------------------------------------------------------------------ limbs */

int orc_limbs(int L) { return (2 * L - 1) / 64 + 1; } /* call_template_functions.cpp:10 */

/* bits [bitpos, bitpos+nbits) of a W-limb little-endian bitset, nbits<=64 */
static inline uint64_t window64(const uint64_t *b, int W, int bitpos, int nbits) {
  int li = bitpos >> 6, off = bitpos & 63;
  uint64_t v = li < W ? b[li] >> off : 0;
  if (off && li + 1 < W) v |= b[li + 1] << (64 - off);
  if (nbits < 64) v &= ((1ULL << nbits) - 1);
  return v;
}

/* std::bitset<W*64> operator>>=(2) / operator<<=(2) (reorder.h:556-557) */
static inline void shr2(uint64_t *b, int W) {
  for (int i = 0; i < W; i++) b[i] = (b[i] >> 2) | (i + 1 < W ? b[i + 1] << 62 : 0);
}
static inline void shl2(uint64_t *b, int W) {
  for (int i = W - 1; i >= 0; i--) b[i] = (b[i] << 2) | (i ? b[i - 1] >> 62 : 0);
}

/* ((a ^ b) & mask[lo][L-hi]).count() where the mask covers bases [lo,hi)
 * (generatemasks, bitset_util.h:223-236; use at reorder.h:291-301) */
static inline int hamming_range(const uint64_t *a, const uint64_t *b, int W, int lo, int hi) {
  if (hi <= lo) return 0;
  int blo = 2 * lo, bhi = 2 * hi, c = 0;
  for (int i = blo >> 6; i < W && i * 64 < bhi; i++) {
    uint64_t x = a[i] ^ b[i];
    int s = i * 64;
    if (blo > s) x &= ~0ULL << (blo - s);
    if (bhi < s + 64) x &= (1ULL << (bhi - s)) - 1;
    c += __builtin_popcountll(x);
  }
  return c;
}

/* ----------------------------------------------------- chars <-> 2 bits */

/* bitsettostring (reorder.h:76-92): code 0..3 -> A,G,C,T */
static void bits_to_string(const uint64_t *b, int W, char *s, int readlen) {
  static const char revinttochar[4] = {'A', 'G', 'C', 'T'};
  for (int j = 0; j < readlen; j++) s[j] = revinttochar[window64(b, W, 2 * j, 2)];
  s[readlen] = '\0';
}

/* chartobitset (bitset_util.h:238-244) with basemask of reorder.h:94-108:
 * A=00, G: bit 2i, C: bit 2i+1, T: both */
static void string_to_bits(const char *s, int readlen, uint64_t *b, int W) {
  memset(b, 0, sizeof(uint64_t) * W);
  for (int i = 0; i < readlen; i++) {
    uint64_t v;
    switch (s[i]) {
      case 'A': v = 0; break;
      case 'G': v = 1; break;
      case 'C': v = 2; break;
      default: v = 3; break; /* 'T' */
    }
    b[(2 * i) >> 6] |= v << ((2 * i) & 63);
  }
}

/* reverse_complement (util.cpp:376-381, table util.h:23-29) */
static void reverse_complement(const char *s, char *s1, int readlen) {
  for (int j = 0; j < readlen; j++) {
    char c = s[readlen - j - 1];
    s1[j] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
  }
  s1[readlen] = '\0';
}

size_t orc_pack_read(const char *s, int len, uint8_t *dst) { /* util.cpp:269-294 */
  uint16_t readlen = (uint16_t)len;
  memcpy(dst, &readlen, 2);
  int nb = (len + 3) / 4;
  for (int i = 0; i < nb; i++) dst[2 + i] = 0;
  for (int i = 0; i < len; i++) {
    uint8_t v = s[i] == 'A' ? 0 : s[i] == 'C' ? 2 : s[i] == 'G' ? 1 : 3;
    dst[2 + i / 4] |= (uint8_t)(v << (2 * (i % 4)));
  }
  return 2 + (size_t)nb;
}

int64_t orc_load_dna(const uint8_t *dna, size_t nbytes, uint32_t n, int L, uint64_t *read,
                     uint16_t *len) { /* reorder.h:222-244 */
  int W = orc_limbs(L);
  size_t p = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (p + 2 > nbytes) return -1;
    uint16_t l;
    memcpy(&l, dna + p, 2);
    p += 2;
    size_t nb = ((uint32_t)l + 3) / 4;
    if (p + nb > nbytes || (int)l > L) return -1;
    memcpy((uint8_t *)(read + (size_t)i * W), dna + p, nb); /* raw copy into bitset memory */
    len[i] = l;
    p += nb;
  }
  return (int64_t)p;
}

void orc_dict_windows(int L, int start[2], int end[2]) { /* reorder.h:751-759 */
  start[0] = L > 100 ? L / 2 - 32 : L / 2 - L * 32 / 100;
  end[0] = L / 2 - 1;
  start[1] = L / 2;
  end[1] = L > 100 ? L / 2 - 1 + 32 : L / 2 - 1 + L * 32 / 100;
}

/* ------------------------------------------------------------ dictionary */

#include "orc_internal.h" /* dict_t (shared with encoder_oracle.c) */

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}

/* returns bin index or -1 (the reference gets >= numkeys or a false positive
 * that fails the key check at reorder.h:282-285) */
static inline int64_t dict_lookup(const dict_t *d, uint64_t key) {
  if (d->numkeys == 0) return -1;
  uint64_t h = mix64(key) & d->hmask;
  for (;;) {
    uint32_t v = d->htab[h];
    if (!v) return -1;
    if (d->keys[v - 1] == key) return (int64_t)v - 1;
    h = (h + 1) & d->hmask;
  }
}

static inline uint64_t read_key(const uint64_t *r, int W, const dict_t *d) {
  /* (read & mask1) >> bpb*start  (bitset_util.h:64-72,:94-95); bpb = 2 (reorder) or 3 (encoder) */
  int bpb = d->bpb ? d->bpb : 2;
  return window64(r, W, bpb * d->start, bpb * (d->end - d->start + 1));
}

/* constructdictionary (bitset_util.h:74-221) */
static void dict_build(dict_t *d, const uint64_t *read, const uint16_t *len, uint32_t n, int W) {
  uint64_t *ull = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  d->dict_numreads = 0;
  for (uint32_t i = 0; i < n; i++) /* :83-105 */
    if ((int)len[i] > d->end) ull[d->dict_numreads++] = read_key(read + (size_t)i * W, W, d);
  uint64_t *allkeys = (uint64_t *)malloc(sizeof(uint64_t) * (d->dict_numreads ? d->dict_numreads : 1));
  memcpy(allkeys, ull, sizeof(uint64_t) * d->dict_numreads);
  qsort(ull, d->dict_numreads, sizeof(uint64_t), cmp_u64); /* :123 */
  uint32_t k = 0;
  if (d->dict_numreads) {
    for (uint32_t i = 1; i < d->dict_numreads; i++)
      if (ull[i] != ull[k]) ull[++k] = ull[i];
    d->numkeys = k + 1;
  } else {
    d->numkeys = 0;
  }
  d->keys = ull;
  uint64_t cap = 2;
  while (cap < 2ull * d->numkeys) cap <<= 1;
  d->hmask = cap - 1;
  d->htab = (uint32_t *)calloc(cap, sizeof(uint32_t));
  for (uint32_t i = 0; i < d->numkeys; i++) {
    uint64_t h = mix64(d->keys[i]) & d->hmask;
    while (d->htab[h]) h = (h + 1) & d->hmask;
    d->htab[h] = i + 1;
  }
  /* counting sort into CSR bins, ascending read id inside a bin (:172-215) */
  d->startpos = (uint32_t *)calloc((size_t)d->numkeys + 1, sizeof(uint32_t));
  d->empty_bin = (uint8_t *)calloc(d->numkeys ? d->numkeys : 1, 1);
  d->read_id = (uint32_t *)malloc(sizeof(uint32_t) * (d->dict_numreads ? d->dict_numreads : 1));
  for (uint32_t j = 0; j < d->dict_numreads; j++) d->startpos[dict_lookup(d, allkeys[j]) + 1]++;
  for (uint32_t i = 1; i < d->numkeys; i++) d->startpos[i] += d->startpos[i - 1];
  uint32_t i = 0;
  for (uint32_t j = 0; j < d->dict_numreads; j++) {
    while ((int)len[i] <= d->end) i++;
    d->read_id[d->startpos[dict_lookup(d, allkeys[j])]++] = i;
    i++;
  }
  for (int64_t keynum = d->numkeys; keynum >= 1; keynum--) d->startpos[keynum] = d->startpos[keynum - 1];
  if (d->numkeys) d->startpos[0] = 0;
  free(allkeys);
}

static void dict_free(dict_t *d) {
  free(d->keys); free(d->startpos); free(d->read_id); free(d->empty_bin); free(d->htab);
}

/* bbhashdict::findpos (bitset_util.cpp:20-35) */
static inline void findpos(const dict_t *d, int64_t *dictidx, uint64_t startposidx) {
  dictidx[0] = d->startpos[startposidx];
  uint32_t endidx = d->startpos[startposidx + 1];
  if (d->read_id[endidx - 1] == MAX_NUM_READS)
    dictidx[1] = endidx - 1;
  else if (d->read_id[endidx - 1] == MAX_NUM_READS + 1)
    dictidx[1] = dictidx[0] + d->read_id[endidx - 2];
  else
    dictidx[1] = endidx;
}

/* bbhashdict::remove (bitset_util.cpp:37-63) */
static inline void bin_remove(dict_t *d, int64_t *dictidx, uint64_t startposidx, int64_t current) {
  int64_t size = dictidx[1] - dictidx[0];
  if (size == 1) {
    d->empty_bin[startposidx] = 1;
    return;
  }
  /* std::lower_bound(read_id+dictidx[0], read_id+dictidx[1], current) */
  int64_t lo = dictidx[0], hi = dictidx[1];
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if ((int64_t)d->read_id[mid] < current) lo = mid + 1; else hi = mid;
  }
  for (int64_t i = lo; i < dictidx[1] - 1; i++) d->read_id[i] = d->read_id[i + 1];
  uint32_t endidx = d->startpos[startposidx + 1];
  if (dictidx[1] == endidx)
    d->read_id[endidx - 1] = MAX_NUM_READS;
  else if (d->read_id[endidx - 1] == MAX_NUM_READS) {
    d->read_id[endidx - 1] = MAX_NUM_READS + 1;
    d->read_id[endidx - 2] = (uint32_t)(size - 1);
  } else
    d->read_id[endidx - 2]--;
}

uint32_t orc_build_dict(const uint64_t *read, const uint16_t *len, uint32_t n, int L, int which,
                        uint64_t *keys_out, uint32_t *startpos_out, uint32_t *read_id_out,
                        uint32_t *dict_numreads) {
  int s[2], e[2];
  orc_dict_windows(L, s, e);
  dict_t d;
  memset(&d, 0, sizeof(d));
  d.start = s[which];
  d.end = e[which];
  dict_build(&d, read, len, n, orc_limbs(L));
  memcpy(keys_out, d.keys, sizeof(uint64_t) * d.numkeys);
  memcpy(startpos_out, d.startpos, sizeof(uint32_t) * ((size_t)d.numkeys + 1));
  memcpy(read_id_out, d.read_id, sizeof(uint32_t) * d.dict_numreads);
  *dict_numreads = d.dict_numreads;
  uint32_t nk = d.numkeys;
  dict_free(&d);
  return nk;
}

int64_t orc_bin_live(const uint32_t *read_id, uint32_t cap) {
  dict_t d;
  uint32_t sp[2] = {0, cap};
  d.startpos = sp;
  d.read_id = (uint32_t *)read_id;
  int64_t di[2];
  findpos(&d, di, 0);
  return di[1] - di[0];
}

int64_t orc_bin_remove(uint32_t *read_id, uint32_t cap, uint8_t *empty_bin, int64_t current) {
  dict_t d;
  uint32_t sp[2] = {0, cap};
  d.startpos = sp;
  d.read_id = read_id;
  d.empty_bin = empty_bin;
  int64_t di[2];
  findpos(&d, di, 0);
  bin_remove(&d, di, 0, current);
  findpos(&d, di, 0);
  return di[1] - di[0];
}

/* -------------------------------------------------------- updaterefcount */

typedef struct {
  int32_t cnt[4][ORC_MAX_READ_LEN + 1];
  uint64_t ref[ORC_WMAX], revref[ORC_WMAX];
  int ref_len;
} cons_t;

static inline int chartoint(char a) { return (a & 0x06) >> 1; } /* reorder.h:121-123: A0 C1 T2 G3 */

/* updaterefcount (reorder.h:110-220), loops kept literal (order-sensitive,
 * including the in-place aliasing of the first reverse case). */
static void updaterefcount(const uint64_t *cur, cons_t *c, int resetcount, int rev, int shift,
                           int cur_readlen, int max_readlen, int W, orc_stats *st) {
  static const char inttochar[4] = {'A', 'C', 'T', 'G'};
  char s[ORC_MAX_READ_LEN + 1], s1[ORC_MAX_READ_LEN + 1], *current;
  int32_t(*count)[ORC_MAX_READ_LEN + 1] = c->cnt;
  int ref_len = c->ref_len;
  st->updates++;
  bits_to_string(cur, W, s, cur_readlen);
  if (!rev)
    current = s;
  else {
    reverse_complement(s, s1, cur_readlen);
    current = s1;
  }
  if (resetcount) { /* :133-142 */
    for (int j = 0; j < 4; j++) memset(count[j], 0, sizeof(int32_t) * max_readlen);
    for (int i = 0; i < cur_readlen; i++) count[chartoint(current[i])][i] = 1;
    ref_len = cur_readlen;
  } else {
    if (!rev) { /* :144-156 */
      for (int i = 0; i < ref_len - shift; i++) {
        for (int j = 0; j < 4; j++) count[j][i] = count[j][i + shift];
        if (i < cur_readlen) count[chartoint(current[i])][i] += 1;
      }
      for (int i = ref_len - shift; i < cur_readlen; i++) {
        for (int j = 0; j < 4; j++) count[j][i] = 0;
        count[chartoint(current[i])][i] = 1;
      }
      ref_len = ref_len - shift > cur_readlen ? ref_len - shift : cur_readlen;
    } else { /* :157-200 */
      if (cur_readlen - shift >= ref_len) {
        for (int i = cur_readlen - shift - ref_len; i < cur_readlen - shift; i++) {
          for (int j = 0; j < 4; j++) count[j][i] = count[j][i - (cur_readlen - shift - ref_len)];
          count[chartoint(current[i])][i] += 1;
        }
        for (int i = 0; i < cur_readlen - shift - ref_len; i++) {
          for (int j = 0; j < 4; j++) count[j][i] = 0;
          count[chartoint(current[i])][i] = 1;
        }
        for (int i = cur_readlen - shift; i < cur_readlen; i++) {
          for (int j = 0; j < 4; j++) count[j][i] = 0;
          count[chartoint(current[i])][i] = 1;
        }
        ref_len = cur_readlen;
      } else if (ref_len + shift <= max_readlen) {
        for (int i = ref_len - cur_readlen + shift; i < ref_len; i++)
          count[chartoint(current[i - (ref_len - cur_readlen + shift)])][i] += 1;
        for (int i = ref_len; i < ref_len + shift; i++) {
          for (int j = 0; j < 4; j++) count[j][i] = 0;
          count[chartoint(current[i - (ref_len - cur_readlen + shift)])][i] = 1;
        }
        ref_len = ref_len + shift;
      } else {
        for (int i = 0; i < max_readlen - shift; i++)
          for (int j = 0; j < 4; j++) count[j][i] = count[j][i + (ref_len + shift - max_readlen)];
        for (int i = max_readlen - cur_readlen; i < max_readlen - shift; i++)
          count[chartoint(current[i - (max_readlen - cur_readlen)])][i] += 1;
        for (int i = max_readlen - shift; i < max_readlen; i++) {
          for (int j = 0; j < 4; j++) count[j][i] = 0;
          count[chartoint(current[i - (max_readlen - cur_readlen)])][i] = 1;
        }
        ref_len = max_readlen;
      }
    }
    for (int i = 0; i < ref_len; i++) { /* :204-212 */
      int max = 0, indmax = 0;
      for (int j = 0; j < 4; j++)
        if (count[j][i] > max) {
          max = count[j][i];
          indmax = j;
        }
      current[i] = inttochar[indmax];
    }
  }
  string_to_bits(current, ref_len, c->ref, W); /* :214-217 */
  char revcurrent[ORC_MAX_READ_LEN + 1];
  reverse_complement(current, revcurrent, ref_len);
  string_to_bits(revcurrent, ref_len, c->revref, W);
  c->ref_len = ref_len;
}

/* -------------------------------------------------------- output helpers */

typedef struct {
  uint32_t *order; char *rc; char *flag; int64_t *pos; uint16_t *rlen;
  uint64_t n, cap;
  uint32_t *order_s; uint64_t ns, caps;
} outbuf_t;

static void ob_push(outbuf_t *o, uint32_t order, char rc, char flag, int64_t pos, uint16_t rlen) {
  if (o->n == o->cap) {
    o->cap = o->cap ? o->cap * 2 : 64;
    o->order = (uint32_t *)realloc(o->order, o->cap * sizeof(uint32_t));
    o->rc = (char *)realloc(o->rc, o->cap);
    o->flag = (char *)realloc(o->flag, o->cap);
    o->pos = (int64_t *)realloc(o->pos, o->cap * sizeof(int64_t));
    o->rlen = (uint16_t *)realloc(o->rlen, o->cap * sizeof(uint16_t));
  }
  o->order[o->n] = order; o->rc[o->n] = rc; o->flag[o->n] = flag; o->pos[o->n] = pos; o->rlen[o->n] = rlen;
  o->n++;
}
static void ob_push_s(outbuf_t *o, uint32_t order) {
  if (o->ns == o->caps) {
    o->caps = o->caps ? o->caps * 2 : 64;
    o->order_s = (uint32_t *)realloc(o->order_s, o->caps * sizeof(uint32_t));
  }
  o->order_s[o->ns++] = order;
}
static void ob_free(outbuf_t *o) {
  free(o->order); free(o->rc); free(o->flag); free(o->pos); free(o->rlen); free(o->order_s);
}

/* ----------------------------------------------------------- serial path */

typedef struct {
  const uint64_t *read;
  const uint16_t *len;
  uint32_t n;
  int L, W, maxshift;
  dict_t dict[2];
  uint8_t *remainingreads;
  orc_stats *st;
} ctx_t;

/* search_match (reorder.h:246-318), single-thread (locks always succeed).
 * `ref` is the already shifted bitset (ref >> 2*shift or revref << 2*shift). */
static int search_match(ctx_t *x, const uint64_t *ref, uint32_t *k, int rev, int shift, int ref_len) {
  int flag = 0;
  int64_t dictidx[2];
  x->st->search_calls++;
  for (int l = 0; l < 2; l++) {
    dict_t *d = &x->dict[l];
    if (!rev) {
      if (d->end + shift >= ref_len) continue;
    } else {
      if (d->end >= ref_len + shift || d->start <= shift) continue;
    }
    uint64_t ull = read_key(ref, x->W, d); /* :269-270 */
    x->st->probes++;
    int64_t startposidx = dict_lookup(d, ull); /* :271-273 */
    if (startposidx < 0) continue;
    findpos(d, dictidx, (uint64_t)startposidx);
    if (d->empty_bin[startposidx]) continue; /* :277-281 */
    uint64_t ull1 = read_key(x->read + (size_t)d->read_id[dictidx[0]] * x->W, x->W, d);
    if (ull == ull1) { /* :285 */
      x->st->keyok++;
      for (int64_t i = dictidx[1] - 1; i >= dictidx[0] && i >= dictidx[1] - MAX_SEARCH_REORDER; i--) {
        uint32_t rid = d->read_id[i];
        x->st->cands++;
        int hamming;
        if (!rev) {
          int m = ref_len - shift < (int)x->len[rid] ? ref_len - shift : (int)x->len[rid];
          hamming = hamming_range(ref, x->read + (size_t)rid * x->W, x->W, 0, m);
        } else {
          int m = ref_len + shift < (int)x->len[rid] ? ref_len + shift : (int)x->len[rid];
          hamming = hamming_range(ref, x->read + (size_t)rid * x->W, x->W, shift, m);
        }
        if (hamming <= THRESH_REORDER) {
          x->st->hits++;
          if (x->remainingreads[rid]) {
            x->remainingreads[rid] = 0;
            *k = rid;
            flag = 1;
          }
          if (flag == 1) break;
        }
      }
    }
    if (flag == 1) break;
  }
  return flag;
}

static int reorder_serial_impl(const uint64_t *read, const uint16_t *len, uint32_t n, int L, orc_out *out,
                               orc_stats *st, const orc_shadow *sh, uint64_t *mm);

int orc_reorder_serial(const uint64_t *read, const uint16_t *len, uint32_t n, int L, orc_out *out,
                       orc_stats *st) {
  return reorder_serial_impl(read, len, n, L, out, st, NULL, NULL);
}

/* The same run with every step mirrored on a shadow state that only reference code advances
 * (oracle/ref_units_driver.cpp): mm[0] search_match results that differ, mm[1] working ref/revref or ref_len that
 * differ at a search, mm[2] ref/revref/ref_len after updaterefcount, mm[3] count[][] after updaterefcount,
 * mm[4] seed picks, mm[5] bin removals / first claim the reference could not do, mm[6..9] calls made
 * (search, update, remove, seed). */
int orc_reorder_serial_shadow(const uint64_t *read, const uint16_t *len, uint32_t n, int L, orc_out *out,
                              orc_stats *st, const orc_shadow *sh, uint64_t *mm) {
  memset(mm, 0, sizeof(uint64_t) * 10);
  return reorder_serial_impl(read, len, n, L, out, st, sh, mm);
}

#define SH_SEARCH(refp, rev)                                                                     \
  if (sh) {                                                                                      \
    int b_ = sh->search(sh->user, refp, rev, shift, c->ref_len, flag, k);                        \
    mm[0] += b_ & 1; mm[1] += (b_ >> 1) & 1; mm[6]++;                                            \
  }
#define SH_UPDATE(rid, reset, rev, sft)                                                          \
  if (sh) {                                                                                      \
    int b_ = sh->update(sh->user, (uint32_t)(rid), reset, rev, sft, &c->cnt[0][0],               \
                        ORC_MAX_READ_LEN + 1, c->ref, c->revref, c->ref_len);                    \
    mm[2] += b_ & 1; mm[3] += (b_ >> 1) & 1; mm[7]++;                                            \
  }

static int reorder_serial_impl(const uint64_t *read, const uint16_t *len, uint32_t n, int L, orc_out *out,
                               orc_stats *st, const orc_shadow *sh, uint64_t *mm) {
  ctx_t x;
  memset(&x, 0, sizeof(x));
  memset(st, 0, sizeof(*st));
  x.read = read; x.len = len; x.n = n; x.L = L; x.W = orc_limbs(L);
  x.maxshift = L / 2; /* reorder.h:750 */
  x.st = st;
  int s[2], e[2];
  orc_dict_windows(L, s, e);
  for (int l = 0; l < 2; l++) { x.dict[l].start = s[l]; x.dict[l].end = e[l]; }
  if (n > 0) /* reorder.h:772 */
    for (int l = 0; l < 2; l++) dict_build(&x.dict[l], read, len, n, x.W);
  x.remainingreads = (uint8_t *)malloc(n ? n : 1);
  memset(x.remainingreads, 1, n);
  const int W = x.W;

  outbuf_t ob;
  memset(&ob, 0, sizeof(ob));
  cons_t *c = (cons_t *)calloc(1, sizeof(cons_t));
  uint64_t ref[ORC_WMAX], revref[ORC_WMAX];

  /* reorder() thread body (reorder.h:351-627) with one thread */
  uint32_t firstread = 0, unmatched = 0;
  int stop_searching = 0;
  uint32_t num_reads_thr = 0, num_unmatched_past_1M_thr = 0;
  int flag = 0, done = 0, prev_unmatched = 0, left_search_start = 0, left_search = 0;
  int64_t current = 0, prev = 0, first_rid = 0;
  int64_t ref_pos = 0, cur_read_pos = 0;
  int64_t remainingpos = (int64_t)n - 1;
  int64_t dictidx[2];
  current = firstread;
  if (n == 0)
    done = 1;
  else if (x.remainingreads[current] == 0)
    done = 1;
  else {
    x.remainingreads[current] = 0;
    unmatched++;
    if (sh) mm[5] += sh->claim_first(sh->user, (uint32_t)current) != 0;
  }
  if (!done) {
    updaterefcount(read + (size_t)current * W, c, 1, 0, 0, len[current], L, W, st);
    SH_UPDATE(current, 1, 0, 0)
    cur_read_pos = 0; ref_pos = 0; first_rid = current; prev_unmatched = 1; prev = current;
  }
  while (!done) {
    st->iterations++;
    if (num_reads_thr % 1000000 == 0) { /* :433-438 */
      if ((float)num_unmatched_past_1M_thr > STOP_CRITERIA_REORDER * 1000000) stop_searching = 1;
      num_unmatched_past_1M_thr = 0;
    }
    num_reads_thr++;
    if (!left_search_start) { /* :458-475 */
      for (int l = 0; l < 2; l++) {
        dict_t *d = &x.dict[l];
        if ((int)len[current] <= d->end) continue;
        uint64_t ull = read_key(read + (size_t)current * W, W, d);
        int64_t startposidx = dict_lookup(d, ull);
        findpos(d, dictidx, (uint64_t)startposidx);
        bin_remove(d, dictidx, (uint64_t)startposidx, current);
      }
      if (sh) { mm[5] += sh->remove(sh->user, (uint32_t)current) != 0; mm[8]++; }
    } else
      left_search_start = 0;
    flag = 0;
    uint32_t k = 0;
    if (!stop_searching) {
      memcpy(ref, c->ref, sizeof(uint64_t) * W);
      memcpy(revref, c->revref, sizeof(uint64_t) * W);
      for (int shift = 0; shift < x.maxshift; shift++) { /* :479-558 */
        flag = search_match(&x, ref, &k, 0, shift, c->ref_len);
        SH_SEARCH(ref, 0)
        if (flag == 1) {
          current = k;
          int ref_len_old = c->ref_len;
          updaterefcount(read + (size_t)current * W, c, 0, 0, shift, len[current], L, W, st);
          SH_UPDATE(current, 0, 0, shift)
          if (!left_search) {
            cur_read_pos = ref_pos + shift;
            ref_pos = cur_read_pos;
          } else {
            cur_read_pos = ref_pos + ref_len_old - shift - len[current];
            ref_pos = ref_pos + ref_len_old - shift - c->ref_len;
          }
          if (prev_unmatched) ob_push(&ob, (uint32_t)prev, 'd', '0', 0, len[prev]);
          ob_push(&ob, (uint32_t)current, left_search ? 'r' : 'd', '1', cur_read_pos, len[current]);
          prev_unmatched = 0;
          break;
        }
        flag = search_match(&x, revref, &k, 1, shift, c->ref_len);
        SH_SEARCH(revref, 1)
        if (flag == 1) {
          current = k;
          int ref_len_old = c->ref_len;
          updaterefcount(read + (size_t)current * W, c, 0, 1, shift, len[current], L, W, st);
          SH_UPDATE(current, 0, 1, shift)
          if (!left_search) {
            cur_read_pos = ref_pos + ref_len_old + shift - len[current];
            ref_pos = ref_pos + ref_len_old + shift - c->ref_len;
          } else {
            cur_read_pos = ref_pos - shift;
            ref_pos = cur_read_pos;
          }
          if (prev_unmatched) ob_push(&ob, (uint32_t)prev, 'd', '0', 0, len[prev]);
          ob_push(&ob, (uint32_t)current, left_search ? 'd' : 'r', '1', cur_read_pos, len[current]);
          prev_unmatched = 0;
          break;
        }
        shl2(revref, W); /* :556 */
        shr2(ref, W);    /* :557 */
      }
    }
    if (flag == 0) { /* :559-615 */
      num_unmatched_past_1M_thr++;
      if (!left_search) {
        left_search = 1;
        left_search_start = 1;
        updaterefcount(read + (size_t)first_rid * W, c, 1, 1, 0, len[first_rid], L, W, st);
        SH_UPDATE(first_rid, 1, 1, 0)
        ref_pos = 0;
        cur_read_pos = 0;
      } else {
        left_search = 0;
        for (int64_t j = remainingpos; j >= 0; j--) {
          if (x.remainingreads[j] == 1) {
            current = j;
            remainingpos = j - 1;
            x.remainingreads[j] = 0;
            flag = 1;
            unmatched++;
            break;
          }
        }
        if (sh) { mm[4] += sh->pick_seed(sh->user) != (flag ? current : -1); mm[9]++; }
        if (flag == 0) {
          if (prev_unmatched) ob_push_s(&ob, (uint32_t)prev);
          done = 1;
        } else {
          updaterefcount(read + (size_t)current * W, c, 1, 0, 0, len[current], L, W, st);
          SH_UPDATE(current, 1, 0, 0)
          ref_pos = 0;
          cur_read_pos = 0;
          if (prev_unmatched) ob_push_s(&ob, (uint32_t)prev);
          prev_unmatched = 1;
          first_rid = current;
          prev = current;
        }
      }
    }
  }
  st->unmatched = unmatched;
  memcpy(out->order, ob.order, ob.n * sizeof(uint32_t));
  memcpy(out->rc, ob.rc, ob.n);
  memcpy(out->flag, ob.flag, ob.n);
  memcpy(out->pos, ob.pos, ob.n * sizeof(int64_t));
  memcpy(out->rlen, ob.rlen, ob.n * sizeof(uint16_t));
  memcpy(out->order_s, ob.order_s, ob.ns * sizeof(uint32_t));
  out->n_matched = ob.n;
  out->n_single = ob.ns;
  if (out->tid_off) { out->tid_off[0] = 0; out->tid_off[1] = ob.n; }
  if (out->tid_off_s) { out->tid_off_s[0] = 0; out->tid_off_s[1] = ob.ns; }
  ob_free(&ob);
  free(c);
  free(x.remainingreads);
  if (n > 0) for (int l = 0; l < 2; l++) dict_free(&x.dict[l]);
  return 0;
}

/* -------------------------------------------------- K-chain rounds schedule
 *
 * The schedule the GPU path implements (DESIGN.md "Chain schedule"):
 *   - K chains; chain i starts at seed i*floor(N/K) (reorder.h:405-421, with
 *     the critical section entered in chain-id order).
 *   - bins are immutable; a read is live in a bin iff !taken[rid].  In the
 *     single-thread reference a claimed read is always removed from its bins
 *     before the next search (reorder.h:458-472), so "live" == "remaining".
 *   - each round has two phases.  Phase A: every chain, looking at taken[] as
 *     it was at the start of the round, either searches (reorder.h:479-558
 *     order: shift, fwd before rev, dict 0 before 1, bin tail first, at most
 *     1000 live entries) or, if it needs a new contig seed, proposes the
 *     (r+1)-th highest untaken read at or below the global cursor, r being
 *     its rank among seed-needing chains (reorder.h:576-592; one cursor is
 *     equivalent to per-thread remainingpos because every read above any
 *     thread's remainingpos is already claimed).  Proposals do
 *     resv[rid] = min(resv[rid], chain).  Phase B: a chain whose proposal
 *     holds resv wins and applies it; a loser retries the same iteration in
 *     the next round (in the reference a thread that loses the read_lock race
 *     keeps scanning, reorder.h:303-311; both are legal `-t K` interleavings).
 *   Alternatives (A > 1, orc_reorder_rounds_alt): the search also records the next A-1
 *     candidates of the winning probe's bin that pass the Hamming test (same scan, same
 *     1000-live-entry window -- exactly the reads the reference's thread would try next
 *     after losing the read_lock race, reorder.h:303-311).  The proposals are resolved in
 *     A passes: in pass p every chain that has not secured a read yet and has a p-th
 *     candidate proposes it; a read secured in an earlier pass stays with its owner; inside
 *     a pass the lowest chain id wins.  A chain that secures its p-th candidate applies
 *     that read (same probe, same alignment); a chain that secures nothing retries next
 *     round.  Seed proposals have one candidate.  A = 1 is the schedule above.
 *   K = 1 degenerates to the reference's `-t 1` order exactly (no other chain to lose to).
 */

enum { MODE_SEARCH = 0, MODE_NEED_SEED = 1 };
enum { PROP_NONE = 0, PROP_MATCH = 1, PROP_SEED = 2 };

typedef struct {
  cons_t c;
  int64_t current, prev, first_rid, ref_pos, cur_read_pos;
  int done, prev_unmatched, left_search, stop_searching, mode, retrying;
  uint32_t num_reads_thr, num_unmatched_past_1M_thr, unmatched;
  int prop_kind, prop_shift, prop_rev;
  uint32_t prop_rid;
  uint32_t alt[8]; /* further passing candidates of the winning bin, in scan order */
  int nalt;
  int got; /* two-group schedule: index of the candidate the chain secured in this half-step, -1 none */
  outbuf_t ob;
} chain_t;

typedef struct {
  const uint64_t *read;
  const uint16_t *len;
  uint32_t n;
  int L, W, maxshift;
  dict_t dict[2];
  uint8_t *taken;
  orc_stats *st;
} rctx_t;

/* one full search of a chain against the round-start taken[]; alt / nalt (may be null / 0 wanted): the next passing
 * candidates of the winner's bin within the same MAX_SEARCH_REORDER window (not counted in the work counters) */
static int rounds_search(rctx_t *x, const cons_t *c, uint32_t *k, int *oshift, int *orev, uint32_t *alt, int want_alt,
                         int *nalt) {
  if (nalt) *nalt = 0;
  uint64_t ref[ORC_WMAX], revref[ORC_WMAX];
  const int W = x->W;
  memcpy(ref, c->ref, sizeof(uint64_t) * W);
  memcpy(revref, c->revref, sizeof(uint64_t) * W);
  for (int shift = 0; shift < x->maxshift; shift++) {
    for (int rev = 0; rev < 2; rev++) {
      const uint64_t *r = rev ? revref : ref;
      x->st->search_calls++;
      for (int l = 0; l < 2; l++) {
        dict_t *d = &x->dict[l];
        if (!rev) {
          if (d->end + shift >= c->ref_len) continue;
        } else {
          if (d->end >= c->ref_len + shift || d->start <= shift) continue;
        }
        uint64_t key = read_key(r, W, d);
        x->st->probes++;
        int64_t b = dict_lookup(d, key);
        if (b < 0) continue;
        int live = 0;
        for (int64_t i = (int64_t)d->startpos[b + 1] - 1;
             i >= (int64_t)d->startpos[b] && live < MAX_SEARCH_REORDER; i--) {
          uint32_t rid = d->read_id[i];
          if (x->taken[rid]) continue;
          if (!live) x->st->keyok++;
          live++;
          x->st->cands++;
          int lo = rev ? shift : 0;
          int m = rev ? c->ref_len + shift : c->ref_len - shift;
          if ((int)x->len[rid] < m) m = x->len[rid];
          if (hamming_range(r, x->read + (size_t)rid * W, W, lo, m) <= THRESH_REORDER) {
            x->st->hits++;
            *k = rid; *oshift = shift; *orev = rev;
            for (i--; want_alt > 0 && *nalt < want_alt && i >= (int64_t)d->startpos[b] && live < MAX_SEARCH_REORDER; i--) {
              const uint32_t r2 = d->read_id[i];
              if (x->taken[r2]) continue;
              live++;
              int m2 = rev ? c->ref_len + shift : c->ref_len - shift;
              if ((int)x->len[r2] < m2) m2 = x->len[r2];
              if (hamming_range(r, x->read + (size_t)r2 * W, W, lo, m2) <= THRESH_REORDER) alt[(*nalt)++] = r2;
            }
            return 1;
          }
        }
      }
    }
    shl2(revref, W);
    shr2(ref, W);
  }
  return 0;
}

int orc_reorder_rounds(const uint64_t *read, const uint16_t *len, uint32_t n, int L, uint32_t K,
                       int num_thr, orc_out *out, orc_stats *st) {
  return orc_reorder_rounds_alt(read, len, n, L, K, num_thr, 1, out, st);
}

int orc_reorder_rounds_alt(const uint64_t *read, const uint16_t *len, uint32_t n, int L, uint32_t K,
                           int num_thr, int A, orc_out *out, orc_stats *st) {
  if (K == 0 || num_thr <= 0 || A < 1 || A > 8) return -1;
  rctx_t x;
  memset(&x, 0, sizeof(x));
  memset(st, 0, sizeof(*st));
  x.read = read; x.len = len; x.n = n; x.L = L; x.W = orc_limbs(L); x.maxshift = L / 2; x.st = st;
  const int W = x.W;
  int s[2], e[2];
  orc_dict_windows(L, s, e);
  for (int l = 0; l < 2; l++) { x.dict[l].start = s[l]; x.dict[l].end = e[l]; }
  if (n > 0) for (int l = 0; l < 2; l++) dict_build(&x.dict[l], read, len, n, W);
  x.taken = (uint8_t *)calloc(n ? n : 1, 1);
  uint32_t *resv = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  memset(resv, 0xff, sizeof(uint32_t) * (n ? n : 1));
  chain_t *ch = (chain_t *)calloc(K, sizeof(chain_t));
  uint32_t *seedlist = (uint32_t *)malloc(sizeof(uint32_t) * K);
  int64_t cursor = (int64_t)n - 1;
  uint32_t alive = 0;

  uint32_t firstread = 0;
  for (uint32_t i = 0; i < K; i++) { /* reorder.h:405-431 */
    chain_t *c = &ch[i];
    c->current = firstread;
    if (n == 0 || x.taken[firstread])
      c->done = 1;
    else {
      x.taken[firstread] = 1;
      c->unmatched++;
      updaterefcount(read + (size_t)c->current * W, &c->c, 1, 0, 0, len[c->current], L, W, st);
      c->first_rid = c->prev = c->current;
      c->prev_unmatched = 1;
      alive++;
    }
    firstread += n / K;
  }

  while (alive) {
    st->rounds++;
    /* ---- phase A: proposals from the round-start state */
    uint32_t nneed = 0;
    for (uint32_t i = 0; i < K; i++)
      if (!ch[i].done && ch[i].mode == MODE_NEED_SEED) nneed++;
    uint32_t nseed = 0;
    if (nneed) {
      for (int64_t j = cursor; j >= 0 && nseed < nneed; j--)
        if (!x.taken[j]) seedlist[nseed++] = (uint32_t)j;
    }
    uint32_t rank = 0;
    for (uint32_t i = 0; i < K; i++) {
      chain_t *c = &ch[i];
      if (c->done) continue;
      c->prop_kind = PROP_NONE;
      if (c->mode == MODE_NEED_SEED) {
        if (rank < nseed) {
          c->prop_kind = PROP_SEED;
          c->prop_rid = seedlist[rank];
        } else { /* no reads left (reorder.h:593-599) */
          if (c->prev_unmatched) ob_push_s(&c->ob, (uint32_t)c->prev);
          c->done = 1;
          alive--;
        }
        rank++;
        continue;
      }
      if (!c->retrying) {
        st->iterations++;
        if (c->num_reads_thr % 1000000 == 0) {
          if ((float)c->num_unmatched_past_1M_thr > STOP_CRITERIA_REORDER * 1000000) c->stop_searching = 1;
          c->num_unmatched_past_1M_thr = 0;
        }
        c->num_reads_thr++;
      }
      c->nalt = 0;
      if (!c->stop_searching) {
        uint32_t k; int sh, rv;
        if (rounds_search(&x, &c->c, &k, &sh, &rv, c->alt, A - 1, &c->nalt)) {
          c->prop_kind = PROP_MATCH; c->prop_rid = k; c->prop_shift = sh; c->prop_rev = rv;
        }
      }
    }
    /* resolution in A passes; resv[rid] = pass << 28 | chain of the owner (an earlier pass beats a later one, inside
     * a pass the lowest chain id wins); secured[i] = 1 + index of the candidate chain i got */
    for (int p = 0; p < A; p++) {
      for (uint32_t i = 0; i < K; i++) {
        chain_t *c = &ch[i];
        if (c->done || c->prop_kind == PROP_NONE) continue;
        int got = 0;
        for (int q = 0; q < p && !got; q++) {
          if (q > c->nalt) break;
          const uint32_t rq = q ? c->alt[q - 1] : c->prop_rid;
          got = resv[rq] == (((uint32_t)q << 28) | i);
        }
        if (got || p > c->nalt || (p > 0 && c->prop_kind != PROP_MATCH)) continue;
        const uint32_t rp = p ? c->alt[p - 1] : c->prop_rid, key = ((uint32_t)p << 28) | i;
        if (resv[rp] > key) resv[rp] = key;
      }
    }
    /* ---- phase B: resolve + apply */
    for (uint32_t i = 0; i < K; i++) {
      chain_t *c = &ch[i];
      if (c->done) continue;
      if (c->prop_kind != PROP_NONE) {
        int got = -1;
        for (int q = 0; q <= c->nalt && got < 0; q++) {
          const uint32_t rq = q ? c->alt[q - 1] : c->prop_rid;
          if (resv[rq] == (((uint32_t)q << 28) | i)) got = q;
        }
        if (got < 0) { /* lost every candidate */
          st->lost++;
          if (c->mode == MODE_SEARCH) c->retrying = 1;
          continue;
        }
        if (got > 0) c->prop_rid = c->alt[got - 1]; /* same probe, same alignment, the next read of the bin */
      }
      if (c->prop_kind == PROP_MATCH) {
        const int shift = c->prop_shift;
        x.taken[c->prop_rid] = 1;
        c->retrying = 0;
        c->current = c->prop_rid;
        int ref_len_old = c->c.ref_len;
        updaterefcount(read + (size_t)c->current * W, &c->c, 0, c->prop_rev, shift, len[c->current], L, W, st);
        char rcch;
        if (!c->prop_rev) { /* reorder.h:490-497,:508 */
          if (!c->left_search) {
            c->cur_read_pos = c->ref_pos + shift;
            c->ref_pos = c->cur_read_pos;
          } else {
            c->cur_read_pos = c->ref_pos + ref_len_old - shift - len[c->current];
            c->ref_pos = c->ref_pos + ref_len_old - shift - c->c.ref_len;
          }
          rcch = c->left_search ? 'r' : 'd';
        } else { /* reorder.h:528-535,:546 */
          if (!c->left_search) {
            c->cur_read_pos = c->ref_pos + ref_len_old + shift - len[c->current];
            c->ref_pos = c->ref_pos + ref_len_old + shift - c->c.ref_len;
          } else {
            c->cur_read_pos = c->ref_pos - shift;
            c->ref_pos = c->cur_read_pos;
          }
          rcch = c->left_search ? 'd' : 'r';
        }
        if (c->prev_unmatched) ob_push(&c->ob, (uint32_t)c->prev, 'd', '0', 0, len[c->prev]);
        ob_push(&c->ob, (uint32_t)c->current, rcch, '1', c->cur_read_pos, len[c->current]);
        c->prev_unmatched = 0;
      } else if (c->prop_kind == PROP_SEED) { /* reorder.h:580-587,:600-613 */
        x.taken[c->prop_rid] = 1;
        c->current = c->prop_rid;
        c->unmatched++;
        updaterefcount(read + (size_t)c->current * W, &c->c, 1, 0, 0, len[c->current], L, W, st);
        c->ref_pos = 0; c->cur_read_pos = 0;
        if (c->prev_unmatched) ob_push_s(&c->ob, (uint32_t)c->prev);
        c->prev_unmatched = 1;
        c->first_rid = c->current;
        c->prev = c->current;
        c->mode = MODE_SEARCH;
      } else if (c->mode == MODE_SEARCH) { /* search failed (reorder.h:559-575) */
        c->retrying = 0;
        c->num_unmatched_past_1M_thr++;
        if (!c->left_search) {
          c->left_search = 1;
          updaterefcount(read + (size_t)c->first_rid * W, &c->c, 1, 1, 0, len[c->first_rid], L, W, st);
          c->ref_pos = 0; c->cur_read_pos = 0;
        } else {
          c->left_search = 0;
          c->mode = MODE_NEED_SEED;
        }
      }
    }
    if (nseed) cursor = (int64_t)seedlist[nseed - 1] - 1;
  }

  /* assemble: chain i -> tid i % num_thr, chains ascending inside a tid */
  uint64_t nm = 0, ns = 0;
  for (int t = 0; t < num_thr; t++) {
    if (out->tid_off) out->tid_off[t] = nm;
    if (out->tid_off_s) out->tid_off_s[t] = ns;
    for (uint32_t i = (uint32_t)t; i < K; i += (uint32_t)num_thr) {
      outbuf_t *o = &ch[i].ob;
      memcpy(out->order + nm, o->order, o->n * sizeof(uint32_t));
      memcpy(out->rc + nm, o->rc, o->n);
      memcpy(out->flag + nm, o->flag, o->n);
      memcpy(out->pos + nm, o->pos, o->n * sizeof(int64_t));
      memcpy(out->rlen + nm, o->rlen, o->n * sizeof(uint16_t));
      memcpy(out->order_s + ns, o->order_s, o->ns * sizeof(uint32_t));
      nm += o->n;
      ns += o->ns;
    }
  }
  if (out->tid_off) out->tid_off[num_thr] = nm;
  if (out->tid_off_s) out->tid_off_s[num_thr] = ns;
  out->n_matched = nm;
  out->n_single = ns;
  for (uint32_t i = 0; i < K; i++) { st->unmatched += ch[i].unmatched; ob_free(&ch[i].ob); }
  free(ch); free(seedlist); free(resv); free(x.taken);
  if (n > 0) for (int l = 0; l < 2; l++) dict_free(&x.dict[l]);
  return 0;
}

/* -------------------------------------------------- K-chain schedule with two chain groups (opts.phases = 2)
 *
 * The rounds schedule above leaves the GPU draining between rounds (every chain waits for the slowest of the round).
 * Here the chains form two groups, [0, Kh) and [Kh, K) with Kh = K/2 rounded to the nearest multiple of 2048 (at least 2048), whose rounds
 * ALTERNATE: half-step h belongs to group h & 1.  In its half-step a group does what a round does above, with two
 * differences that make its search independent of the other group's half-step in front of it (so that on the GPU the
 * two can run side by side):
 *   - phase A looks at the pool as it was after the group's OWN last half-step (view[g] = the truth after half-step
 *     h - 2), not at the other group's claims of half-step h - 1;
 *   - phase B resolves the group's proposals among themselves (lowest chain id) AND against the truth: a read the other
 *     group claimed in half-step h - 1 is lost (the chain retries, as after losing to a lower chain id).
 *   Seeds: group 0 takes the (r+1)-th highest untaken read of [nmid, n) at or below its cursor, group 1 of [0, nmid),
 *   nmid = n/2 rounded down to a multiple of 4096, r = the chain's rank among its group's seed-needing chains; a group
 *   whose range is used up lets its chains finish (reorder.h:593-599) while the other carries on.  (Needs n >= K: every chain
 *   starts on a seed of its own, so either group has chains to use up its range.)
 * One candidate per proposal.  Every read is still claimed exactly once, by a chain whose search saw it untaken and
 * within Hamming distance: a legal interleaving of the reference's `-t K` run like the rounds schedule.
 */
static void ph_apply(rctx_t *x, chain_t *c, const uint64_t *read, const uint16_t *len, int L, int W, orc_stats *st) {
  if (c->prop_kind == PROP_MATCH) {
    const int shift = c->prop_shift;
    c->retrying = 0;
    c->current = c->prop_rid;
    int ref_len_old = c->c.ref_len;
    updaterefcount(read + (size_t)c->current * W, &c->c, 0, c->prop_rev, shift, len[c->current], L, W, st);
    char rcch;
    if (!c->prop_rev) { /* reorder.h:490-497,:508 */
      if (!c->left_search) {
        c->cur_read_pos = c->ref_pos + shift;
        c->ref_pos = c->cur_read_pos;
      } else {
        c->cur_read_pos = c->ref_pos + ref_len_old - shift - len[c->current];
        c->ref_pos = c->ref_pos + ref_len_old - shift - c->c.ref_len;
      }
      rcch = c->left_search ? 'r' : 'd';
    } else { /* reorder.h:528-535,:546 */
      if (!c->left_search) {
        c->cur_read_pos = c->ref_pos + ref_len_old + shift - len[c->current];
        c->ref_pos = c->ref_pos + ref_len_old + shift - c->c.ref_len;
      } else {
        c->cur_read_pos = c->ref_pos - shift;
        c->ref_pos = c->cur_read_pos;
      }
      rcch = c->left_search ? 'd' : 'r';
    }
    if (c->prev_unmatched) ob_push(&c->ob, (uint32_t)c->prev, 'd', '0', 0, len[c->prev]);
    ob_push(&c->ob, (uint32_t)c->current, rcch, '1', c->cur_read_pos, len[c->current]);
    c->prev_unmatched = 0;
  } else if (c->prop_kind == PROP_SEED) { /* reorder.h:580-587,:600-613 */
    c->current = c->prop_rid;
    c->unmatched++;
    updaterefcount(read + (size_t)c->current * W, &c->c, 1, 0, 0, len[c->current], L, W, st);
    c->ref_pos = 0; c->cur_read_pos = 0;
    if (c->prev_unmatched) ob_push_s(&c->ob, (uint32_t)c->prev);
    c->prev_unmatched = 1;
    c->first_rid = c->current;
    c->prev = c->current;
    c->mode = MODE_SEARCH;
  } else if (c->mode == MODE_SEARCH) { /* search failed (reorder.h:559-575) */
    c->retrying = 0;
    c->num_unmatched_past_1M_thr++;
    if (!c->left_search) {
      c->left_search = 1;
      updaterefcount(read + (size_t)c->first_rid * W, &c->c, 1, 1, 0, len[c->first_rid], L, W, st);
      c->ref_pos = 0; c->cur_read_pos = 0;
    } else {
      c->left_search = 0;
      c->mode = MODE_NEED_SEED;
    }
  }
  (void)x;
}

uint32_t orc_phase_split(uint32_t K) {
  const uint64_t h = (((uint64_t)K / 2 + 1024) / 2048) * 2048;
  return (uint32_t)(h < 2048 ? 2048 : h);
}

int orc_reorder_rounds_ph(const uint64_t *read, const uint16_t *len, uint32_t n, int L, uint32_t K,
                          int num_thr, orc_out *out, orc_stats *st) {
  return orc_reorder_rounds_ph_alt(read, len, n, L, K, num_thr, 1, out, st);
}

/* ... with A candidates per match proposal (orc_reorder_rounds_alt): a candidate is SECURED when it holds the group's resv[]
 * entry of its pass (an earlier pass beats a later one, the lowest chain id wins a pass) AND the read is still free in
 * truth; a chain that has not secured a candidate of an earlier pass proposes its next one */
int orc_reorder_rounds_ph_alt(const uint64_t *read, const uint16_t *len, uint32_t n, int L, uint32_t K,
                              int num_thr, int A, orc_out *out, orc_stats *st) {
  const uint32_t Kh = orc_phase_split(K), nmid = (n / 2) & ~4095u;
  if (K < 4096 || Kh >= K || num_thr <= 0 || nmid == 0 || A < 1 || A > 8 || n < K) return -1; /* (n < K: only chain 0 would run) */
  rctx_t x;
  memset(&x, 0, sizeof(x));
  memset(st, 0, sizeof(*st));
  x.read = read; x.len = len; x.n = n; x.L = L; x.W = orc_limbs(L); x.maxshift = L / 2; x.st = st;
  const int W = x.W;
  int s[2], e[2];
  orc_dict_windows(L, s, e);
  for (int l = 0; l < 2; l++) { x.dict[l].start = s[l]; x.dict[l].end = e[l]; }
  for (int l = 0; l < 2; l++) dict_build(&x.dict[l], read, len, n, W);
  uint8_t *truth = (uint8_t *)calloc(n, 1), *view[2];
  view[0] = (uint8_t *)calloc(n, 1); view[1] = (uint8_t *)calloc(n, 1);
  uint32_t *resv[2];
  for (int g = 0; g < 2; g++) { resv[g] = (uint32_t *)malloc(sizeof(uint32_t) * n); memset(resv[g], 0xff, sizeof(uint32_t) * n); }
  chain_t *ch = (chain_t *)calloc(K, sizeof(chain_t));
  uint32_t *seedlist = (uint32_t *)malloc(sizeof(uint32_t) * K);
  uint32_t *won[2], nwon[2] = {0, 0}; /* reads claimed in the group's last half-step */
  won[0] = (uint32_t *)malloc(sizeof(uint32_t) * K); won[1] = (uint32_t *)malloc(sizeof(uint32_t) * K);
  const uint32_t glo[2] = {0, Kh}, ghi[2] = {Kh, K};
  const int64_t slo[2] = {(int64_t)nmid, 0};
  int64_t cursor[2] = {(int64_t)n - 1, (int64_t)nmid - 1};
  uint32_t alive = 0;

  uint32_t firstread = 0;
  for (uint32_t i = 0; i < K; i++) { /* reorder.h:405-431 */
    chain_t *c = &ch[i];
    c->current = firstread;
    if (truth[firstread])
      c->done = 1;
    else {
      truth[firstread] = view[0][firstread] = view[1][firstread] = 1;
      c->unmatched++;
      updaterefcount(read + (size_t)c->current * W, &c->c, 1, 0, 0, len[c->current], L, W, st);
      c->first_rid = c->prev = c->current;
      c->prev_unmatched = 1;
      alive++;
    }
    firstread += n / K;
  }

  for (uint64_t h = 0; alive; h++) {
    const int g = (int)(h & 1);
    if (!g) st->rounds++;
    x.taken = view[g];
    /* ---- phase A: proposals of group g from its own view */
    uint32_t nneed = 0;
    for (uint32_t i = glo[g]; i < ghi[g]; i++)
      if (!ch[i].done && ch[i].mode == MODE_NEED_SEED) nneed++;
    uint32_t nseed = 0;
    if (nneed) {
      for (int64_t j = cursor[g]; j >= slo[g] && nseed < nneed; j--)
        if (!view[g][j]) seedlist[nseed++] = (uint32_t)j;
    }
    uint32_t rank = 0;
    for (uint32_t i = glo[g]; i < ghi[g]; i++) {
      chain_t *c = &ch[i];
      if (c->done) continue;
      c->prop_kind = PROP_NONE;
      if (c->mode == MODE_NEED_SEED) {
        if (rank < nseed) {
          c->prop_kind = PROP_SEED;
          c->prop_rid = seedlist[rank];
        } else { /* the group's range is used up (reorder.h:593-599) */
          if (c->prev_unmatched) ob_push_s(&c->ob, (uint32_t)c->prev);
          c->done = 1;
          alive--;
        }
        rank++;
        continue;
      }
      if (!c->retrying) {
        st->iterations++;
        if (c->num_reads_thr % 1000000 == 0) {
          if ((float)c->num_unmatched_past_1M_thr > STOP_CRITERIA_REORDER * 1000000) c->stop_searching = 1;
          c->num_unmatched_past_1M_thr = 0;
        }
        c->num_reads_thr++;
      }
      c->nalt = 0;
      if (!c->stop_searching) {
        uint32_t k; int sh, rv;
        if (rounds_search(&x, &c->c, &k, &sh, &rv, c->alt, A - 1, &c->nalt)) {
          c->prop_kind = PROP_MATCH; c->prop_rid = k; c->prop_shift = sh; c->prop_rev = rv;
        }
      }
    }
    /* resolution in A passes (truth does not change before phase B: "free in truth" = not claimed up to half-step h - 1) */
#define PH_SECURED(C, I, Q) (resv[g][(Q) ? (C)->alt[(Q) - 1] : (C)->prop_rid] == (((uint32_t)(Q) << 28) | (I)) && \
                             !truth[(Q) ? (C)->alt[(Q) - 1] : (C)->prop_rid])
    for (int p = 0; p < A; p++) {
      for (uint32_t i = glo[g]; i < ghi[g]; i++) {
        chain_t *c = &ch[i];
        if (c->done || c->prop_kind == PROP_NONE) continue;
        int got = 0;
        for (int q = 0; q < p && !got; q++) {
          if (q > c->nalt) break;
          got = PH_SECURED(c, i, q);
        }
        if (got || p > c->nalt || (p > 0 && c->prop_kind != PROP_MATCH)) continue;
        const uint32_t rp = p ? c->alt[p - 1] : c->prop_rid, key = ((uint32_t)p << 28) | i;
        if (resv[g][rp] > key) resv[g][rp] = key;
      }
    }
    /* ---- phase B: the first secured candidate is applied */
    nwon[g] = 0;
    /* (the secured candidates are judged first: truth must stay as it was while they are) */
    for (uint32_t i = glo[g]; i < ghi[g]; i++) {
      chain_t *c = &ch[i];
      if (c->done || c->prop_kind == PROP_NONE) continue;
      c->got = -1;
      for (int q = 0; q <= c->nalt && c->got < 0; q++)
        if (PH_SECURED(c, i, q)) c->got = q;
    }
    for (uint32_t i = glo[g]; i < ghi[g]; i++) {
      chain_t *c = &ch[i];
      if (c->done) continue;
      if (c->prop_kind != PROP_NONE) {
        const int got = c->got;
        if (got < 0) {
          st->lost++;
          if (c->mode == MODE_SEARCH) c->retrying = 1;
          continue;
        }
        if (got > 0) c->prop_rid = c->alt[got - 1]; /* same probe, same alignment, the next read of the bin */
        truth[c->prop_rid] = 1;
        won[g][nwon[g]++] = c->prop_rid;
      }
      ph_apply(&x, c, read, len, L, W, st);
    }
#undef PH_SECURED
    /* the group's view for its next half-step: the truth as of now (its own claims + the other group's last ones) */
    for (uint32_t j = 0; j < nwon[g]; j++) view[g][won[g][j]] = 1;
    for (uint32_t j = 0; j < nwon[g ^ 1]; j++) view[g][won[g ^ 1][j]] = 1;
    if (nseed) cursor[g] = (int64_t)seedlist[nseed - 1] - 1;
  }

  /* assemble: chain i -> tid i % num_thr, chains ascending inside a tid */
  uint64_t nm = 0, ns = 0;
  for (int t = 0; t < num_thr; t++) {
    if (out->tid_off) out->tid_off[t] = nm;
    if (out->tid_off_s) out->tid_off_s[t] = ns;
    for (uint32_t i = (uint32_t)t; i < K; i += (uint32_t)num_thr) {
      outbuf_t *o = &ch[i].ob;
      memcpy(out->order + nm, o->order, o->n * sizeof(uint32_t));
      memcpy(out->rc + nm, o->rc, o->n);
      memcpy(out->flag + nm, o->flag, o->n);
      memcpy(out->pos + nm, o->pos, o->n * sizeof(int64_t));
      memcpy(out->rlen + nm, o->rlen, o->n * sizeof(uint16_t));
      memcpy(out->order_s + ns, o->order_s, o->ns * sizeof(uint32_t));
      nm += o->n;
      ns += o->ns;
    }
  }
  if (out->tid_off) out->tid_off[num_thr] = nm;
  if (out->tid_off_s) out->tid_off_s[num_thr] = ns;
  out->n_matched = nm;
  out->n_single = ns;
  for (uint32_t i = 0; i < K; i++) { st->unmatched += ch[i].unmatched; ob_free(&ch[i].ob); }
  free(ch); free(seedlist); free(truth);
  for (int g = 0; g < 2; g++) { free(view[g]); free(resv[g]); free(won[g]); }
  for (int l = 0; l < 2; l++) dict_free(&x.dict[l]);
  return 0;
}

/* -------------------------------------------------- replay check of a reorder output (any schedule, any size)
 *
 * Which read a chain takes next depends on the schedule (the reference's own `-t K` output differs from run to run); that
 * every emitted match was one the reference COULD have made does not.  orc_check_contigs replays every contig of an output
 * from its streams alone, with the reference's own state machine: the seed (flag '0', 'd', pos 0) resets the consensus
 * (updaterefcount, reorder.h:133-142); for every following record the orientation character and the position give back
 * (reverse?, shift) by inverting reorder.h:490-497 / :528-535 for the search direction the contig is in, and the read must
 * then be exactly what search_match accepts (reorder.h:246-318): a shift in [0, maxshift), a dictionary whose probe is
 * valid at that shift (reorder.h:264-268) and whose window of the shifted consensus EQUALS the read's own key (so the
 * dictionary would have returned the read's bin), and at most THRESH_REORDER differing bits over the compared range
 * (reorder.h:291-302); then the consensus is updated (updaterefcount -- the function that tests/test_oracle_vs_ref_units.py
 * pins against the reference's own code) and ref_pos advanced as the reference does.  A contig searches rightwards first and
 * leftwards after one reset to the reverse complement of its first read (reorder.h:559-575): the replay switches direction at
 * the first record that does not verify rightwards and, should a later record then fail, tries every other switch point.
 * What it cannot see is global: whether a better-placed candidate was free at the time, and whether a singleton had a match.
 * Returns 0; res: contigs, matched records, contigs that no switch point verifies, index of the first bad record. */
/* the consensus update of the replay: the restatement above, or -- upd != NULL -- the REFERENCE'S OWN updaterefcount<N>
 * (oracle/_ref/libref_units.so::ref_u_updaterefcount, compiled from reorder.h:110-220 where it lies) on the same state */
static void chk_update(orc_update_fn upd, const uint64_t *cur, cons_t *c, int reset, int rev, int shift, int n, int L, int W,
                       orc_stats *st) {
  if (upd) upd(L, cur, &c->cnt[0][0], ORC_MAX_READ_LEN + 1, c->ref, c->revref, &c->ref_len, reset, rev, shift, n);
  else updaterefcount(cur, c, reset, rev, shift, n, L, W, st);
}
static int chk_step(orc_update_fn upd, const uint64_t *read, const uint16_t *len, int L, int W, int maxshift, const int ds[2], const int de[2],
                    cons_t *c, int64_t *ref_pos, int left, uint32_t r, char rcch, int64_t p, orc_stats *st) {
  const int n = len[r], R_old = c->ref_len;
  const int rev = left ? (rcch == 'd') : (rcch == 'r');
  int64_t sh;
  if (!rev) sh = !left ? p - *ref_pos : *ref_pos + R_old - n - p;        /* reorder.h:490-497 */
  else sh = !left ? p - *ref_pos - R_old + n : *ref_pos - p;             /* reorder.h:528-535 */
  if (sh < 0 || sh >= maxshift) return 0;
  const int shift = (int)sh;
  uint64_t x[ORC_WMAX];
  memcpy(x, rev ? c->revref : c->ref, sizeof(uint64_t) * W);
  for (int k = 0; k < shift; k++) { if (rev) shl2(x, W); else shr2(x, W); }  /* reorder.h:556-557 */
  const uint64_t *rd = read + (size_t)r * W;
  int findable = 0;
  for (int l = 0; l < 2 && !findable; l++) {
    if (!rev) { if (de[l] + shift >= R_old) continue; }
    else if (de[l] >= R_old + shift || ds[l] <= shift) continue;      /* reorder.h:264-268 */
    if (n <= de[l]) continue;                                          /* the read is not in dictionary l (bitset_util.h:101) */
    const int nb = 2 * (de[l] - ds[l] + 1);
    if (window64(x, W, 2 * ds[l], nb) == window64(rd, W, 2 * ds[l], nb)) findable = 1;
  }
  if (!findable) return 0;
  int lo = rev ? shift : 0, m = rev ? R_old + shift : R_old - shift;
  if (n < m) m = n;
  if (hamming_range(x, rd, W, lo, m) > THRESH_REORDER) return 0;
  chk_update(upd, rd, c, 0, rev, shift, n, L, W, st);
  if (!rev) *ref_pos = !left ? p : *ref_pos + R_old - shift - c->ref_len;
  else *ref_pos = !left ? *ref_pos + R_old + shift - c->ref_len : p;
  return 1;
}

/* replay records [a, b) of one contig with the left search starting at record sw (sw == b: never; sw < 0: at the first
 * record that fails rightwards); returns the index of the first record that does not verify, or -1 */
static int64_t chk_replay(orc_update_fn upd, const uint64_t *read, const uint16_t *len, int L, int W, const int ds[2], const int de[2],
                          const uint32_t *order, const char *rc, const int64_t *pos, int64_t a, int64_t b, int64_t sw,
                          cons_t *c, orc_stats *st) {
  const uint32_t first = order[a];
  int64_t ref_pos = 0;
  int left = 0;
  chk_update(upd, read + (size_t)first * W, c, 1, 0, 0, len[first], L, W, st);
  for (int64_t i = a + 1; i < b; i++) {
    if (!left && i == sw) {
      left = 1;
      chk_update(upd, read + (size_t)first * W, c, 1, 1, 0, len[first], L, W, st);  /* reorder.h:567 */
      ref_pos = 0;
    }
    if (chk_step(upd, read, len, L, W, L / 2, ds, de, c, &ref_pos, left, order[i], rc[i], pos[i], st)) continue;
    if (left || sw >= 0) return i;
    left = 1;  /* (a failed step leaves consensus and ref_pos untouched) */
    chk_update(upd, read + (size_t)first * W, c, 1, 1, 0, len[first], L, W, st);
    ref_pos = 0;
    if (!chk_step(upd, read, len, L, W, L / 2, ds, de, c, &ref_pos, left, order[i], rc[i], pos[i], st)) return i;
  }
  return -1;
}

int orc_check_contigs(const uint64_t *read, const uint16_t *len, uint32_t n, int L, const uint32_t *order, const char *rc,
                      const char *flag, const int64_t *pos, uint64_t nm, const uint64_t *tid_off, int num_thr,
                      uint64_t *res /* [4] */) {
  return orc_check_contigs_upd(read, len, n, L, order, rc, flag, pos, nm, tid_off, num_thr, res, NULL);
}
int orc_check_contigs_upd(const uint64_t *read, const uint16_t *len, uint32_t n, int L, const uint32_t *order, const char *rc,
                          const char *flag, const int64_t *pos, uint64_t nm, const uint64_t *tid_off, int num_thr,
                          uint64_t *res /* [4] */, orc_update_fn upd) {
  (void)n;
  if (upd && nm) {  /* (the reference's tables are built on the first call: make it before the threads start) */
    cons_t *c0 = (cons_t *)calloc(1, sizeof(cons_t));
    upd(L, read + (size_t)order[0] * orc_limbs(L), &c0->cnt[0][0], ORC_MAX_READ_LEN + 1, c0->ref, c0->revref, &c0->ref_len, 1, 0, 0, len[order[0]]);
    free(c0);
  }
  const int W = orc_limbs(L);
  int ds[2], de[2];
  orc_dict_windows(L, ds, de);
  /* contig starts: every flag '0'; a contig ends at the next one or at the end of its tid's stream */
  uint64_t nc = 0;
  for (uint64_t i = 0; i < nm; i++) nc += flag[i] == '0';
  int64_t *st0 = (int64_t *)malloc(sizeof(int64_t) * (nc + 1)), *en0 = (int64_t *)malloc(sizeof(int64_t) * (nc + 1));
  uint64_t k = 0, bad_structure = 0;
  for (int t = 0; t < num_thr; t++) {
    const uint64_t lo = tid_off[t], hi = tid_off[t + 1];
    if (hi > lo && flag[lo] != '0') bad_structure++;
    for (uint64_t i = lo; i < hi; i++)
      if (flag[i] == '0') {
        if (k && en0[k - 1] < 0) en0[k - 1] = (int64_t)i;
        st0[k] = (int64_t)i; en0[k] = -1; k++;
      }
    if (k && en0[k - 1] < 0) en0[k - 1] = (int64_t)hi;
  }
  uint64_t bad = bad_structure, first_bad = ~0ULL, matches = 0;
#pragma omp parallel
  {
    cons_t *c = (cons_t *)calloc(1, sizeof(cons_t));
    orc_stats st;
    memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 256) reduction(+ : bad, matches) reduction(min : first_bad)
    for (int64_t q = 0; q < (int64_t)k; q++) {
      const int64_t a = st0[q], b = en0[q];
      matches += (uint64_t)(b - a - 1);
      int ok = rc[a] == 'd' && pos[a] == 0 && b - a >= 2;
      int64_t f = a;
      if (ok) {
        f = chk_replay(upd, read, len, L, W, ds, de, order, rc, pos, a, b, -1, c, &st);
        for (int64_t sw = a + 1; f >= 0 && sw <= b; sw++)  /* (rare: the greedy switch point was a coincidence) */
          if (chk_replay(upd, read, len, L, W, ds, de, order, rc, pos, a, b, sw, c, &st) < 0) f = -1;
        ok = f < 0;
      }
      if (!ok) { bad++; if ((uint64_t)f < first_bad) first_bad = (uint64_t)f; }
    }
    free(c);
  }
  res[0] = k; res[1] = matches; res[2] = bad; res[3] = first_bad;
  free(st0); free(en0);
  return 0;
}

/* ------------------------------------------------------------ writetofile */

size_t orc_write_dna_stream(const uint64_t *read, const uint16_t *len, int L, const uint32_t *order,
                            const char *rc, uint64_t cnt, uint8_t *dst) { /* reorder.h:667-687 */
  int W = orc_limbs(L);
  size_t p = 0;
  char s[ORC_MAX_READ_LEN + 1], s1[ORC_MAX_READ_LEN + 1];
  for (uint64_t i = 0; i < cnt; i++) {
    uint32_t current = order[i];
    if (!rc || rc[i] == 'd') {
      uint16_t l = len[current];
      size_t nb = ((uint32_t)l + 3) / 4;
      memcpy(dst + p, &l, 2);
      memcpy(dst + p + 2, read + (size_t)current * W, nb);
      p += 2 + nb;
    } else {
      bits_to_string(read + (size_t)current * W, W, s, len[current]);
      reverse_complement(s, s1, len[current]);
      p += orc_pack_read(s1, len[current], dst + p);
    }
  }
  return p;
}

/* exported for tests/test_oracle_vs_ref.py: Hamming over bases [lo,hi) */
int orc_hamming_range(const uint64_t *a, const uint64_t *b, int W, int lo, int hi) {
  return hamming_range(a, b, W, lo, hi);
}

/* exported for unit tests of the consensus arithmetic: one updaterefcount()
 * call on caller-held state (cnt is [4][512] int32, A C T G rows). */
void orc_updaterefcount(const uint64_t *cur, int32_t *cnt, uint64_t *ref, uint64_t *revref,
                        int *ref_len, int resetcount, int rev, int shift, int cur_readlen,
                        int max_readlen) {
  cons_t *c = (cons_t *)calloc(1, sizeof(cons_t));
  orc_stats st;
  memset(&st, 0, sizeof(st));
  int W = orc_limbs(max_readlen);
  memcpy(c->cnt, cnt, sizeof(c->cnt));
  c->ref_len = *ref_len;
  updaterefcount(cur, c, resetcount, rev, shift, cur_readlen, max_readlen, W, &st);
  memcpy(cnt, c->cnt, sizeof(c->cnt));
  memcpy(ref, c->ref, sizeof(uint64_t) * W);
  memcpy(revref, c->revref, sizeof(uint64_t) * W);
  *ref_len = c->ref_len;
  free(c);
}


/* ---- unit contexts for tests/test_oracle_vs_ref.py: the oracle's two search primitives on caller-driven state,
 * compared there with the REAL search_match<> (oracle/_ref/libref_units.so).
 *   serial flavour: mutable bins (bin_remove) + remainingreads[], one search_match() call;
 *   rounds flavour: immutable bins + taken[], one whole rounds_search() (the shift loop). */
typedef struct { ctx_t x; orc_stats st; } orc_unit_t;

void *orc_unit_create(const uint64_t *read, const uint16_t *len, uint32_t n, int L) {
  orc_unit_t *u = (orc_unit_t *)calloc(1, sizeof(orc_unit_t));
  ctx_t *x = &u->x;
  x->read = read; x->len = len; x->n = n; x->L = L; x->W = orc_limbs(L); x->maxshift = L / 2; x->st = &u->st;
  int s[2], e[2];
  orc_dict_windows(L, s, e);
  for (int l = 0; l < 2; l++) { x->dict[l].start = s[l]; x->dict[l].end = e[l]; }
  if (n > 0) for (int l = 0; l < 2; l++) dict_build(&x->dict[l], read, len, n, x->W);
  x->remainingreads = (uint8_t *)malloc(n ? n : 1);
  memset(x->remainingreads, 1, n);
  return u;
}
void orc_unit_free(void *p) {
  orc_unit_t *u = (orc_unit_t *)p;
  if (u->x.n > 0) for (int l = 0; l < 2; l++) dict_free(&u->x.dict[l]);
  free(u->x.remainingreads);
  free(u);
}
/* reorder.h:458-472 for one read */
void orc_unit_remove(void *p, uint32_t current) {
  ctx_t *x = &((orc_unit_t *)p)->x;
  int64_t dictidx[2];
  for (int l = 0; l < 2; l++) {
    dict_t *d = &x->dict[l];
    if ((int)x->len[current] <= d->end) continue;
    uint64_t ull = read_key(x->read + (size_t)current * x->W, x->W, d);
    int64_t startposidx = dict_lookup(d, ull);
    findpos(d, dictidx, (uint64_t)startposidx);
    bin_remove(d, dictidx, (uint64_t)startposidx, current);
  }
}
void orc_unit_set_remaining(void *p, const uint8_t *r) { ctx_t *x = &((orc_unit_t *)p)->x; memcpy(x->remainingreads, r, x->n); }
void orc_unit_get_remaining(void *p, uint8_t *r) { ctx_t *x = &((orc_unit_t *)p)->x; memcpy(r, x->remainingreads, x->n); }
int orc_unit_search(void *p, const uint64_t *refbits, int rev, int shift, int ref_len, uint32_t *k) {
  return search_match(&((orc_unit_t *)p)->x, refbits, k, rev, shift, ref_len);
}
/* rounds_search on (ref, revref, ref_len, taken[]) -- taken = !remainingreads of the same context, bins untouched */
int orc_unit_rounds_search(void *p, const uint64_t *ref, const uint64_t *revref, int ref_len, const uint8_t *taken,
                           uint32_t *k, int *shift, int *rev) {
  orc_unit_t *u = (orc_unit_t *)p;
  rctx_t r;
  memset(&r, 0, sizeof(r));
  r.read = u->x.read; r.len = u->x.len; r.n = u->x.n; r.L = u->x.L; r.W = u->x.W; r.maxshift = u->x.maxshift;
  r.dict[0] = u->x.dict[0]; r.dict[1] = u->x.dict[1];
  r.taken = (uint8_t *)taken;
  r.st = &u->st;
  cons_t *c = (cons_t *)calloc(1, sizeof(cons_t));
  memcpy(c->ref, ref, sizeof(uint64_t) * r.W);
  memcpy(c->revref, revref, sizeof(uint64_t) * r.W);
  c->ref_len = ref_len;
  int f = rounds_search(&r, c, k, shift, rev, NULL, 0, NULL);
  free(c);
  return f;
}
/* chartobitset / bitsettostring twins (reorder.h:76-92, bitset_util.h:238-244) */
void orc_string_to_bits(const char *s, int len, int L, uint64_t *b) { string_to_bits(s, len, b, orc_limbs(L)); }
void orc_bits_to_string(const uint64_t *b, int len, int L, char *s) { bits_to_string(b, orc_limbs(L), s, len); }
void orc_reverse_complement(const char *s, char *s1, int len) { reverse_complement(s, s1, len); }

/* ------------------------------------------------ OpenMP port (CPU baseline only)
 *
 * Free-running threads like the reference's `-t T` (reorder.h:351-627): one greedy chain per
 * thread, shared taken[] claimed with compare-and-swap where the reference uses try-locks, a
 * per-thread remainingpos for seed picking.  Like the reference it is NOT deterministic for
 * T > 1 (threads race for reads); it exists so bench.py can time a multi-core CPU run of the
 * same algorithm on the GPU box.  Bins are immutable + taken[] (cheaper than the reference's
 * locked bin compaction, so this baseline is if anything faster than the reference).
 * Dictionary build: parallel key extraction + parallel LSD radix sort (the reference sorts
 * serially with std::sort, bitset_util.h:123).
 */
#ifdef _OPENMP
#include <omp.h>

static double g_omp_phase[2]; /* seconds: dictionaries, chains of the last orc_reorder_omp call */
void orc_last_omp_phases(double *out) { out[0] = g_omp_phase[0]; out[1] = g_omp_phase[1]; }

typedef struct { uint64_t k; uint32_t v; } kv_t;

static void par_radix_sort(kv_t *a, kv_t *tmp, size_t n, int nbits, int T) {
  const int passes = (nbits + 7) / 8;
  size_t *hist = (size_t *)malloc(sizeof(size_t) * 256 * (size_t)T);
  for (int p = 0; p < passes; p++) {
    const int sh = 8 * p;
    memset(hist, 0, sizeof(size_t) * 256 * (size_t)T);
#pragma omp parallel num_threads(T)
    {
      const int t = omp_get_thread_num();
      const size_t lo = n * (size_t)t / T, hi = n * (size_t)(t + 1) / T;
      size_t *h = hist + 256 * (size_t)t;
      for (size_t i = lo; i < hi; i++) h[(a[i].k >> sh) & 255]++;
#pragma omp barrier
#pragma omp single
      {
        size_t run = 0;
        for (int d = 0; d < 256; d++)
          for (int tt = 0; tt < T; tt++) { size_t c = hist[256 * (size_t)tt + d]; hist[256 * (size_t)tt + d] = run; run += c; }
      }
      for (size_t i = lo; i < hi; i++) tmp[h[(a[i].k >> sh) & 255]++] = a[i];
    }
    kv_t *sw = a; a = tmp; tmp = sw;
  }
  if (passes & 1) memcpy(tmp, a, sizeof(kv_t) * n); /* result must end in the caller's `a` */
  free(hist);
}

static void dict_build_par(dict_t *d, const uint64_t *read, const uint16_t *len, uint32_t n, int W, int T) {
  kv_t *a = (kv_t *)malloc(sizeof(kv_t) * (n ? n : 1)), *tmp = (kv_t *)malloc(sizeof(kv_t) * (n ? n : 1));
  uint32_t m = 0;
  for (uint32_t i = 0; i < n; i++)
    if ((int)len[i] > d->end) { a[m].v = i; m++; }
#pragma omp parallel for num_threads(T) schedule(static)
  for (uint32_t j = 0; j < m; j++) a[j].k = read_key(read + (size_t)a[j].v * W, W, d);
  double tt = omp_get_wtime();
  par_radix_sort(a, tmp, m, 2 * (d->end - d->start + 1), T);
  if (getenv("ORC_TIMING")) fprintf(stderr, "[orc_omp]   radix sort %.2f s\n", omp_get_wtime() - tt);
  tt = omp_get_wtime();
  d->dict_numreads = m;
  d->keys = (uint64_t *)malloc(sizeof(uint64_t) * (m ? m : 1));
  d->startpos = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)m + 1));
  d->read_id = (uint32_t *)malloc(sizeof(uint32_t) * (m ? m : 1));
  uint32_t nk = 0;
  for (uint32_t j = 0; j < m; j++) {
    if (j == 0 || a[j].k != a[j - 1].k) { d->keys[nk] = a[j].k; d->startpos[nk] = j; nk++; }
    d->read_id[j] = a[j].v;
  }
  d->startpos[nk] = m;
  d->numkeys = nk;
  d->empty_bin = (uint8_t *)calloc(nk ? nk : 1, 1);
  uint64_t cap = 2;
  while (cap < 2ull * nk) cap <<= 1;
  d->hmask = cap - 1;
  if (getenv("ORC_TIMING")) fprintf(stderr, "[orc_omp]   unique/fill %.2f s\n", omp_get_wtime() - tt);
  tt = omp_get_wtime();
  d->htab = (uint32_t *)calloc(cap, sizeof(uint32_t));
#pragma omp parallel for num_threads(T) schedule(static)
  for (uint32_t i = 0; i < nk; i++) { /* keys are distinct: claim the first free slot with a CAS */
    uint64_t h = mix64(d->keys[i]) & d->hmask;
    for (;;) {
      uint32_t z = 0;
      if (__atomic_compare_exchange_n(&d->htab[h], &z, i + 1, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
      h = (h + 1) & d->hmask;
    }
  }
  if (getenv("ORC_TIMING")) fprintf(stderr, "[orc_omp]   hash insert %.2f s\n", omp_get_wtime() - tt);
  free(a); free(tmp);
}

static inline int claim(uint8_t *taken, uint32_t r) {
  uint8_t z = 0;
  return __atomic_compare_exchange_n(&taken[r], &z, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED);
}

int orc_reorder_omp(const uint64_t *read, const uint16_t *len, uint32_t n, int L, int T, orc_out *out,
                    orc_stats *st) {
  if (T <= 0) return -1;
  rctx_t x;
  memset(&x, 0, sizeof(x));
  memset(st, 0, sizeof(*st));
  x.read = read; x.len = len; x.n = n; x.L = L; x.W = orc_limbs(L); x.maxshift = L / 2;
  const int W = x.W;
  int s[2], e[2];
  orc_dict_windows(L, s, e);
  for (int l = 0; l < 2; l++) { x.dict[l].start = s[l]; x.dict[l].end = e[l]; }
  const int timing = getenv("ORC_TIMING") != NULL;
  double t0 = omp_get_wtime();
  if (n > 0) for (int l = 0; l < 2; l++) dict_build_par(&x.dict[l], read, len, n, W, T);
  g_omp_phase[0] = omp_get_wtime() - t0;
  if (timing) fprintf(stderr, "[orc_omp] dict build %.2f s\n", g_omp_phase[0]);
  t0 = omp_get_wtime();
  x.taken = (uint8_t *)calloc(n ? n : 1, 1);
  outbuf_t *obs = (outbuf_t *)calloc((size_t)T, sizeof(outbuf_t));
  uint64_t tot_unmatched = 0, tot_iter = 0;
  uint32_t firstseed[1024];
  for (int t = 0; t < T && t < 1024; t++) firstseed[t] = (uint32_t)t * (n / (uint32_t)T);
#pragma omp parallel num_threads(T) reduction(+ : tot_unmatched, tot_iter)
  {
    const int tid = omp_get_thread_num();
    outbuf_t *ob = &obs[tid];
    orc_stats lst;
    memset(&lst, 0, sizeof(lst));
    cons_t *c = (cons_t *)calloc(1, sizeof(cons_t));
    uint64_t ref[ORC_WMAX], revref[ORC_WMAX];
    int done = 0, prev_unmatched = 0, left_search = 0, stop_searching = 0;
    uint32_t num_reads_thr = 0, num_unmatched_past = 0;
    int64_t current = firstseed[tid & 1023], prev = 0, first_rid = 0, ref_pos = 0, cur_read_pos = 0;
    int64_t remainingpos = (int64_t)n - 1;
    if (n == 0 || !claim(x.taken, (uint32_t)current)) done = 1;
    else {
      tot_unmatched++;
      updaterefcount(read + (size_t)current * W, c, 1, 0, 0, len[current], L, W, &lst);
      first_rid = prev = current; prev_unmatched = 1;
    }
    while (!done) {
      tot_iter++;
      if (num_reads_thr % 1000000 == 0) {
        if ((float)num_unmatched_past > STOP_CRITERIA_REORDER * 1000000) stop_searching = 1;
        num_unmatched_past = 0;
      }
      num_reads_thr++;
      int flag = 0, fshift = 0, frev = 0;
      uint32_t k = 0;
      if (!stop_searching) {
        memcpy(ref, c->ref, sizeof(uint64_t) * W);
        memcpy(revref, c->revref, sizeof(uint64_t) * W);
        for (int shift = 0; shift < x.maxshift && !flag; shift++) {
          for (int rev = 0; rev < 2 && !flag; rev++) {
            const uint64_t *r = rev ? revref : ref;
            for (int l = 0; l < 2 && !flag; l++) {
              dict_t *d = &x.dict[l];
              if (!rev) { if (d->end + shift >= c->ref_len) continue; }
              else { if (d->end >= c->ref_len + shift || d->start <= shift) continue; }
              int64_t b = dict_lookup(d, read_key(r, W, d));
              if (b < 0) continue;
              int live = 0;
              for (int64_t i = (int64_t)d->startpos[b + 1] - 1; i >= (int64_t)d->startpos[b] && live < MAX_SEARCH_REORDER; i--) {
                uint32_t rid = d->read_id[i];
                if (__atomic_load_n(&x.taken[rid], __ATOMIC_RELAXED)) continue;
                live++;
                int lo = rev ? shift : 0, m = rev ? c->ref_len + shift : c->ref_len - shift;
                if ((int)len[rid] < m) m = len[rid];
                if (hamming_range(r, read + (size_t)rid * W, W, lo, m) <= THRESH_REORDER && claim(x.taken, rid)) {
                  k = rid; fshift = shift; frev = rev; flag = 1;
                  break;
                }
              }
            }
          }
          shl2(revref, W);
          shr2(ref, W);
        }
      }
      if (flag) {
        current = k;
        int ref_len_old = c->ref_len;
        updaterefcount(read + (size_t)current * W, c, 0, frev, fshift, len[current], L, W, &lst);
        char rcch;
        if (!frev) {
          if (!left_search) { cur_read_pos = ref_pos + fshift; ref_pos = cur_read_pos; }
          else { cur_read_pos = ref_pos + ref_len_old - fshift - len[current]; ref_pos = ref_pos + ref_len_old - fshift - c->ref_len; }
          rcch = left_search ? 'r' : 'd';
        } else {
          if (!left_search) { cur_read_pos = ref_pos + ref_len_old + fshift - len[current]; ref_pos = ref_pos + ref_len_old + fshift - c->ref_len; }
          else { cur_read_pos = ref_pos - fshift; ref_pos = cur_read_pos; }
          rcch = left_search ? 'd' : 'r';
        }
        if (prev_unmatched) ob_push(ob, (uint32_t)prev, 'd', '0', 0, len[prev]);
        ob_push(ob, (uint32_t)current, rcch, '1', cur_read_pos, len[current]);
        prev_unmatched = 0;
      } else {
        num_unmatched_past++;
        if (!left_search) {
          left_search = 1;
          updaterefcount(read + (size_t)first_rid * W, c, 1, 1, 0, len[first_rid], L, W, &lst);
          ref_pos = 0; cur_read_pos = 0;
        } else {
          left_search = 0;
          int got = 0;
          for (int64_t j = remainingpos; j >= 0; j--) {
            if (!__atomic_load_n(&x.taken[j], __ATOMIC_RELAXED) && claim(x.taken, (uint32_t)j)) {
              current = j; remainingpos = j - 1; got = 1; tot_unmatched++;
              break;
            }
          }
          if (prev_unmatched) ob_push_s(ob, (uint32_t)prev);
          if (!got) done = 1;
          else {
            updaterefcount(read + (size_t)current * W, c, 1, 0, 0, len[current], L, W, &lst);
            ref_pos = 0; cur_read_pos = 0; prev_unmatched = 1; first_rid = current; prev = current;
          }
        }
      }
    }
    free(c);
  }
  g_omp_phase[1] = omp_get_wtime() - t0;
  if (timing) fprintf(stderr, "[orc_omp] chains %.2f s\n", g_omp_phase[1]);
  uint64_t nm = 0, ns = 0;
  for (int t = 0; t < T; t++) {
    outbuf_t *o = &obs[t];
    if (out->tid_off) out->tid_off[t] = nm;
    if (out->tid_off_s) out->tid_off_s[t] = ns;
    memcpy(out->order + nm, o->order, o->n * sizeof(uint32_t));
    memcpy(out->rc + nm, o->rc, o->n);
    memcpy(out->flag + nm, o->flag, o->n);
    memcpy(out->pos + nm, o->pos, o->n * sizeof(int64_t));
    memcpy(out->rlen + nm, o->rlen, o->n * sizeof(uint16_t));
    memcpy(out->order_s + ns, o->order_s, o->ns * sizeof(uint32_t));
    nm += o->n; ns += o->ns;
    ob_free(o);
  }
  if (out->tid_off) out->tid_off[T] = nm;
  if (out->tid_off_s) out->tid_off_s[T] = ns;
  out->n_matched = nm; out->n_single = ns;
  st->unmatched = tot_unmatched; st->iterations = tot_iter;
  free(obs); free(x.taken);
  if (n > 0) for (int l = 0; l < 2; l++) dict_free(&x.dict[l]);
  return 0;
}
#endif /* _OPENMP */

/* ------------------------------------------------ SURVEY 8(f3) restatements (literal loops) */
/* generate_order_se (reorder_compress_quality_id.cpp:117-125) */
void orc_generate_order_se(const uint32_t *order, uint32_t numreads, uint32_t *order_array) {
  for (uint32_t i = 0; i < numreads; i++) order_array[order[i]] = i;
}
/* generate_order_pe (reorder_compress_quality_id.cpp:101-115) */
void orc_generate_order_pe(const uint32_t *order, uint32_t numreads, uint32_t *order_array) {
  uint32_t pos_after_reordering = 0, numreads_by_2 = numreads / 2;
  for (uint32_t i = 0; i < numreads; i++)
    if (order[i] < numreads_by_2) order_array[order[i]] = pos_after_reordering++;
}
/* pe_encode (pe_encode.cpp:24-84), the three loops kept literal */
void orc_pe_encode(const uint32_t *order, uint32_t numreads, uint32_t *order_array) {
  uint32_t numreads_by_2 = numreads / 2;
  uint32_t *inverse_order_array = (uint32_t *)malloc(sizeof(uint32_t) * (numreads ? numreads : 1));
  for (uint32_t i = 0; i < numreads; i++) {
    order_array[i] = order[i];
    inverse_order_array[order[i]] = i;
  }
  uint32_t pos_in_file_1 = 0;
  for (uint32_t i = 0; i < numreads; i++)
    if (order_array[i] < numreads_by_2) order_array[i] = pos_in_file_1++;
  for (uint32_t i = 0; i < numreads; i++)
    if (order_array[i] >= numreads_by_2) {
      uint32_t pos_in_original = order_array[i];
      uint32_t pos_of_pair_in_original = pos_in_original - numreads_by_2;
      uint32_t pos_of_pair_in_reordered = inverse_order_array[pos_of_pair_in_original];
      uint32_t new_order_of_pair = order_array[pos_of_pair_in_reordered];
      order_array[i] = new_order_of_pair + numreads_by_2;
    }
  free(inverse_order_array);
}
/* correct_order (encoder.cpp:177-222) on an in-memory index array */
void orc_correct_order(uint32_t *order, uint64_t m, const uint32_t *order_N, uint32_t numreads_N, uint32_t n_clean) {
  uint32_t numreads_total = n_clean + numreads_N;
  uint8_t *read_flag_N = (uint8_t *)calloc(numreads_total ? numreads_total : 1, 1);
  for (uint32_t i = 0; i < numreads_N; i++) read_flag_N[order_N[i]] = 1;
  uint32_t *cumulative_N_reads = (uint32_t *)malloc(sizeof(uint32_t) * (n_clean ? n_clean : 1));
  uint32_t pos_in_clean = 0, num_N_reads_till_now = 0;
  for (uint32_t i = 0; i < numreads_total; i++) {
    if (read_flag_N[i]) num_N_reads_till_now++;
    else cumulative_N_reads[pos_in_clean++] = num_N_reads_till_now;
  }
  for (uint64_t i = 0; i < m; i++) order[i] += cumulative_N_reads[order[i]];
  free(read_flag_N); free(cumulative_N_reads);
}

/* ------------------------------------------------ sequence side of preprocess()
 * read_fastq_block (util.cpp:31-54) + N split / packing (preprocess.cpp:186-214, :293-304) +
 * write_dnaN_in_bits (util.cpp:322-348), for one uncompressed FASTQ file held in memory.
 * counts[0..3] = num_reads, num_reads_clean, num_reads_N, max_readlen.
 * Returns 0, -1 = "Invalid FASTQ(A) file. Number of lines not multiple of 4(2)", -2 = "Too long read length". */
static int fq_getline(const uint8_t *t, size_t n, size_t *p, size_t *s, size_t *e) {
  if (*p >= n) return 0; /* std::getline fails at EOF with nothing extracted */
  *s = *p;
  while (*p < n && t[*p] != '\n') (*p)++;
  *e = *p;
  if (*p < n) (*p)++; /* consume the delimiter */
  return 1;
}
int orc_preprocess_fastq(const uint8_t *txt, size_t nbytes, uint8_t *clean, size_t *clean_bytes, uint8_t *ndna,
                         size_t *n_bytes, uint32_t *order_N, uint32_t *counts) {
  size_t p = 0, cb = 0, nb = 0;
  uint32_t num_reads = 0, num_clean = 0, num_N = 0, maxlen = 0;
  for (;;) {
    size_t s, e, rs, re;
    if (!fq_getline(txt, nbytes, &p, &s, &e)) break;           /* id */
    if (!fq_getline(txt, nbytes, &p, &rs, &re)) return -1;      /* read */
    if (re > rs && txt[re - 1] == '\r') re--;                   /* remove_CR_from_end */
    if (!fq_getline(txt, nbytes, &p, &s, &e)) return -1;        /* comment */
    if (!fq_getline(txt, nbytes, &p, &s, &e)) return -1;        /* quality */
    size_t len = re - rs;
    if (len > ORC_MAX_READ_LEN) return -2;
    if (len > maxlen) maxlen = (uint32_t)len;
    int hasN = 0;
    for (size_t i = rs; i < re; i++) hasN |= txt[i] == 'N';
    if (!hasN) {
      cb += orc_pack_read((const char *)txt + rs, (int)len, clean + cb);
      num_clean++;
    } else {
      order_N[num_N++] = num_reads;
      uint16_t l16 = (uint16_t)len;
      memcpy(ndna + nb, &l16, 2);
      size_t nbts = (len + 1) / 2;
      for (size_t i = 0; i < nbts; i++) ndna[nb + 2 + i] = 0;
      for (size_t i = 0; i < len; i++) {
        uint8_t c = txt[rs + i];
        uint8_t v = c == 'A' ? 0 : c == 'C' ? 2 : c == 'G' ? 1 : c == 'T' ? 3 : 4;
        ndna[nb + 2 + i / 2] |= (uint8_t)(v << (4 * (i % 2)));
      }
      nb += 2 + nbts;
    }
    num_reads++;
  }
  *clean_bytes = cb; *n_bytes = nb;
  counts[0] = num_reads; counts[1] = num_clean; counts[2] = num_N; counts[3] = maxlen;
  return 0;
}

/* ------------------------------------------------ bridge for encoder_oracle.c */
uint64_t orc__window64(const uint64_t *b, int W, int bitpos, int nbits) { return window64(b, W, bitpos, nbits); }
void orc__dict_build(dict_t *d, const uint64_t *read, const uint16_t *len, uint32_t n, int W) { dict_build(d, read, len, n, W); }
void orc__dict_free(dict_t *d) { dict_free(d); }
int64_t orc__dict_lookup(const dict_t *d, uint64_t key) { return dict_lookup(d, key); }
void orc__findpos(const dict_t *d, int64_t *dictidx, uint64_t startposidx) { findpos(d, dictidx, startposidx); }
void orc__bin_remove(dict_t *d, int64_t *dictidx, uint64_t startposidx, int64_t current) { bin_remove(d, dictidx, startposidx, current); }
