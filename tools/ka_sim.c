/* tools/ka_sim.c -- probe (round 6): how many table fetches does a chain save when it remembers which windows of its
 * consensus are known absent from the (immutable) dictionaries?  A serial run of the TEST oracle (shadow hooks) on a
 * synthetic 25x pool; per search it counts the tag fetches of (a) the round-5 plan (ordered batches of 4 + 8 + 16 shifts,
 * then one fetch per distinct window) and (b) the same search when windows whose absence the chain already knows are
 * skipped and the remaining ones are taken in priority order in batches of 16 / 32 / 64 / 64...
 * Build: gcc -O2 -o /tmp/ka_sim tools/ka_sim.c -Ioracle -Loracle -loracle_reorder -Wl,-rpath,$PWD/oracle
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "reorder_oracle.h"
#include "orc_internal.h"

static uint64_t rs = 88172645463325252ull;
static inline uint64_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

#define MAXW 8
typedef struct {
  int L, W, s[2], e[2], maxshift;
  dict_t d[2];
  /* chain state mirrored from the update hook */
  uint64_t ref[MAXW], revref[MAXW];
  int ref_len;
  uint8_t ka[2][512][2]; /* [strand][offset][dict] known absent */
  int single;            /* contig is still its seed read alone */
  /* running search */
  int in_search, last_shift, last_rev, last_flag;
  /* totals */
  uint64_t n_search, n_hit, n_fail, f_old, f_new, f_new_hit, f_new_fail, f_old_hit, f_old_fail;
  uint64_t hist_batches_old[8], hist_batches_new[8];
  uint64_t kept, inval;
} sim_t;

static inline int base_at(const uint64_t *b, int p) { return (int)((b[p >> 5] >> (2 * (p & 31))) & 3); }
static inline uint64_t win(const uint64_t *b, int W, int o) { return orc__window64(b, W, 2 * o, 64); }

static int absent(sim_t *S, int strand, int o, int l) {
  const uint64_t k = win(strand ? S->revref : S->ref, S->W, o);
  return orc__dict_lookup(&S->d[l], k) < 0;
}

/* priority-ordered list of codes up to (and including) the winning (shift, rev); fail: all */
static void finish_search(sim_t *S) {
  if (!S->in_search) return;
  S->in_search = 0;
  const int R = S->ref_len, wl = 32, ms = S->maxshift;
  const int hit = S->last_flag;
  const int wsh = S->last_shift, wrev = S->last_rev;
  S->n_search++;
  if (hit) S->n_hit++; else S->n_fail++;
  /* ---- (a) round-5 plan */
  {
    uint64_t f = 0;
    int t0 = 0, nb = 0, done = 0;
    const int plan[3] = {4, 8, 16};
    for (int ph = 0; ph < 3 && !done; ph++) {
      for (int sh = t0; sh < t0 + plan[ph] && sh < ms; sh++)
        for (int rev = 0; rev < 2; rev++)
          for (int l = 0; l < 2; l++) {
            int valid = !rev ? (S->e[l] + sh < R) : (S->e[l] < R + sh && S->s[l] > sh);
            if (valid) f++;
          }
      nb++;
      t0 += plan[ph];
      if (hit && wsh < t0) done = 1;
    }
    if (!done) { /* tail: one fetch per distinct window not seen in the batches (pres) */
      nb++;
      for (int i = 0; i < wl + ms - t0; i++) {
        /* forward window at offset s0 + t0 + i */
        int sh0 = t0 + i, sh1 = sh0 - wl;
        int v0 = sh0 < ms && S->e[0] + sh0 < R;
        int v1 = sh1 >= t0 && sh1 < ms && S->e[1] + sh1 < R;
        if (v0 && sh1 >= 0 && sh1 < t0) v0 = !absent(S, 0, S->s[0] + sh0, 0);
        if (v0 || v1) f++;
        int rh1 = t0 + i, rh0 = rh1 - wl;
        int w1 = rh1 < ms && S->e[1] < R + rh1 && S->s[1] > rh1;
        int w0 = rh0 >= t0 && rh0 < ms && S->e[0] < R + rh0 && S->s[0] > rh0;
        if (w1 && rh0 >= 0 && rh0 < t0) w1 = !absent(S, 1, S->s[1] - rh1, 1);
        if (w0 || w1) f++;
      }
    }
    S->f_old += f;
    if (hit) S->f_old_hit += f; else S->f_old_fail += f;
    S->hist_batches_old[nb < 7 ? nb : 7]++;
  }
  /* ---- (b) known-absent windows skipped, compacted batches */
  {
    uint64_t f = 0;
    int nb = 0, inb = 0, cap = 16, stop = 0;
    uint8_t fetched[2][512];
    memset(fetched, 0, sizeof(fetched));
    for (int sh = 0; sh < ms && !stop; sh++)
      for (int rev = 0; rev < 2 && !stop; rev++)
        for (int l = 0; l < 2; l++) {
          int valid = !rev ? (S->e[l] + sh < R) : (S->e[l] < R + sh && S->s[l] > sh);
          if (!valid) continue;
          const int o = rev ? S->s[l] - sh : S->s[l] + sh;
          if (S->ka[rev][o][l]) continue;
          if (fetched[rev][o]) { /* present in this dictionary by the tags seen: no second fetch (eval_probe walks) */ continue; }
          if (inb == cap) { /* batch full: was the winner in it? */
            nb++;
            inb = 0;
            cap = nb == 1 ? 32 : 64;
            if (hit && (wsh < sh || (wsh == sh && wrev < rev))) { stop = 1; break; }
          }
          f++;
          inb++;
          fetched[rev][o] = 1;
          for (int ll = 0; ll < 2; ll++)
            if (absent(S, rev, o, ll)) S->ka[rev][o][ll] = 1;
          if (hit && wsh == sh && wrev == rev) { /* winner reached: the rest of this batch is fetched too */ }
        }
    /* a hit stops after the batch that holds the winner: count the rest of that batch */
    if (hit && !stop) { /* winner was in the last (partial) batch: nothing to add in the ideal model */ }
    if (inb) nb++;
    S->f_new += f;
    if (hit) S->f_new_hit += f; else S->f_new_fail += f;
    S->hist_batches_new[nb < 7 ? nb : 7]++;
  }
}

static int h_claim(void *u, uint32_t c) { (void)u; (void)c; return 0; }
static int h_remove(void *u, uint32_t c) { (void)u; (void)c; return 0; }
static int64_t h_seed(void *u) { (void)u; return -2; }
static int h_search(void *u, const uint64_t *refs, int rev, int shift, int ref_len, int flag, uint32_t k) {
  sim_t *S = (sim_t *)u;
  (void)refs; (void)ref_len; (void)k;
  S->in_search = 1; S->last_shift = shift; S->last_rev = rev; S->last_flag = flag;
  return 0;
}
static int h_update(void *u, uint32_t rid, int reset, int rev, int shift, const int32_t *cnt, int stride,
                    const uint64_t *ref, const uint64_t *revref, int ref_len) {
  sim_t *S = (sim_t *)u;
  (void)rid; (void)cnt; (void)stride;
  const int had_search = S->in_search;
  finish_search(S);
  const int W = S->W, Ro = S->ref_len, Rn = ref_len;
  uint8_t nk[2][512][2];
  memset(nk, 0, sizeof(nk));
  if (reset) {
    if (rev && S->single && had_search) { /* left search of a lone seed: ref <-> revref */
      memcpy(nk[0], S->ka[1], sizeof(nk[0]));
      memcpy(nk[1], S->ka[0], sizeof(nk[1]));
    }
    if (!rev) S->single = 1;
  } else {
    S->single = 0;
    /* ref_new[i] = ref_old[i + so] for i < cpy (modulo argmax flips); fixed length: so = shift */
    int so, cpy;
    const int n = Rn; /* fixed-length pools only in this probe */
    if (!rev) { so = shift; cpy = Ro - shift; }
    else if (n - shift >= Ro) { so = 0; cpy = Ro; }
    else if (Ro + shift <= S->L) { so = 0; cpy = Ro; }
    else { so = Ro + shift - S->L; cpy = S->L - shift; }
    uint8_t chg[512];
    for (int p = 0; p < Rn; p++) chg[p] = p >= cpy || base_at(ref, p) != base_at(S->ref, p + so);
    for (int o = 0; o + 32 <= Rn; o++) {
      int bad = 0;
      for (int q = o; q < o + 32; q++) bad |= chg[q];
      if (bad) { S->inval++; continue; }
      S->kept++;
      for (int l = 0; l < 2; l++) {
        nk[0][o][l] = S->ka[0][o + so][l];
        /* reverse strand: window at j of revref_new = ref_new window at Rn - 32 - j; old: ref_old window at that + so = revref_old offset Ro - 32 - (Rn - 32 - j + so) */
        const int j = Rn - 32 - o, jo = Ro - 32 - (o + so);
        if (jo >= 0 && jo < 512) nk[1][j][l] = S->ka[1][jo][l];
      }
    }
  }
  memcpy(S->ka, nk, sizeof(nk));
  memcpy(S->ref, ref, 8 * W);
  memcpy(S->revref, revref, 8 * W);
  S->ref_len = Rn;
  return 0;
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 1000000u;
  const int L = argc > 2 ? atoi(argv[2]) : 150;
  const double cov = argc > 3 ? atof(argv[3]) : 25.0;
  const double err = argc > 4 ? atof(argv[4]) : 0.01;
  const uint64_t G = (uint64_t)((double)n * L / cov);
  const int W = orc_limbs(L);
  char *genome = (char *)malloc(G + L);
  for (uint64_t i = 0; i < G + L; i++) genome[i] = "ACGT"[rnd() & 3];
  uint64_t *read = (uint64_t *)calloc((size_t)n * W, 8);
  uint16_t *len = (uint16_t *)malloc(2 * (size_t)n);
  char buf[1024], rc[1024];
  const uint32_t thr = (uint32_t)(err * 4294967296.0);
  for (uint32_t i = 0; i < n; i++) {
    const uint64_t p = rnd() % G;
    memcpy(buf, genome + p, L);
    for (int j = 0; j < L; j++)
      if ((uint32_t)rnd() < thr) { char c; do c = "ACGT"[rnd() & 3]; while (c == buf[j]); buf[j] = c; }
    if (rnd() & 1) { orc_reverse_complement(buf, rc, L); memcpy(buf, rc, L); }
    orc_string_to_bits(buf, L, L, read + (size_t)i * W);
    len[i] = (uint16_t)L;
  }
  sim_t *S = (sim_t *)calloc(1, sizeof(sim_t));
  S->L = L; S->W = W; S->maxshift = L / 2;
  orc_dict_windows(L, S->s, S->e);
  for (int l = 0; l < 2; l++) { S->d[l].start = S->s[l]; S->d[l].end = S->e[l]; orc__dict_build(&S->d[l], read, len, n, W); }
  orc_out out;
  memset(&out, 0, sizeof(out));
  out.order = malloc(4 * (size_t)n); out.rc = malloc(n); out.flag = malloc(n); out.pos = malloc(8 * (size_t)n);
  out.rlen = malloc(2 * (size_t)n); out.order_s = malloc(4 * (size_t)n);
  orc_stats st;
  uint64_t mm[10];
  orc_shadow sh = {S, h_claim, h_remove, h_search, h_update, h_seed};
  orc_reorder_serial_shadow(read, len, n, L, &out, &st, &sh, mm);
  finish_search(S);
  printf("reads %u L %d cov %.0f err %.3f: searches %llu (hit %llu, fail %llu), singletons %llu\n", n, L, cov, err,
         (unsigned long long)S->n_search, (unsigned long long)S->n_hit, (unsigned long long)S->n_fail, (unsigned long long)out.n_single);
  printf("tag fetches per search: round-5 plan %.2f (hit %.2f, fail %.2f)   known-absent %.2f (hit %.2f, fail %.2f)\n",
         (double)S->f_old / S->n_search, (double)S->f_old_hit / S->n_hit, (double)S->f_old_fail / S->n_fail,
         (double)S->f_new / S->n_search, (double)S->f_new_hit / S->n_hit, (double)S->f_new_fail / S->n_fail);
  printf("batches old:"); for (int i = 0; i < 8; i++) printf(" %llu", (unsigned long long)S->hist_batches_old[i]);
  printf("\nbatches new:"); for (int i = 0; i < 8; i++) printf(" %llu", (unsigned long long)S->hist_batches_new[i]);
  printf("\nwindows kept %.1f / invalidated %.1f per update\n", (double)S->kept / (S->n_hit + 1), (double)S->inval / (S->n_hit + 1));
  return 0;
}
