// spring_amd/csrc/encoder.hip -- SURVEY 8(f2): the encoder stage on the GPU.
//
// What encoder_main<N>() computes from the reorder stage's streams (reference src/encoder.h:124-494,
// :572-633; src/encoder.cpp:32-109,:177-222), re-designed as data-parallel passes over HBM-resident
// arrays:
//   contigs     heads from the flag stream + the 10 000 001-read cut (encoder.h:215), two scans
//   sort        one stable radix sort by the read's first base in the concatenated consensus
//               (= contig, pos - min pos)                                   (list::sort, encoder.h:222)
//   consensus   one block per 2048 consensus bases, votes in LDS            (buildcontig)
//   pool        singleton + N reads as 2-bit limbs + N mask, forward and reverse complement;
//               two exact hash dictionaries on their 21-base windows        (constructdictionary, bpb 3)
//   align       one thread per consensus 21-mer: 2 lookups in a table holding both dictionaries serve the
//               4 probes (fwd/rev x dict 0/1) it takes part in; every hit proposes
//               atomicMin(T[read], probe key): the first probe in the reference's serial order wins.
//               Bins deeper than MAX_SEARCH_ENCODER are handled by iterating to the fixed point
//               with "live at probe P" = T_prev[read] >= P                  (encode, encoder.h:243-343)
//   merge       aligned singletons appended in take order, second stable sort   (encoder.h:351)
//   noise       count / scan / write of substitutions against the consensus     (writecontig)
//   tail        unaligned reads, corrected order                     (encoder.h:425-452, correct_order)
// The 3-bit bitsets of the reference are represented as 2-bit SPRING codes (code3 = 2 * code2) plus
// an N bit per base: Hamming(3-bit) = popcount(2-bit xor) + #N, keys with an N never match a
// consensus window.  No CPU fallback: every pass is a kernel or a rocPRIM primitive.
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "reorder_device.h"
#include "reorder_internal.h"
#include "spring_encoder.h"

using sr::fail;

#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess)                                                                      \
      return fail(SPRING_REORDER_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

namespace {

constexpr uint32_t LIST_LIMIT = 10000001u;  // a contig is cut after this many reads (encoder.h:215)
constexpr int MAX_SEARCH_E = 1000;          // params.h:33
constexpr int THRESH_E = 24;                // params.h:34
constexpr unsigned long long INF = ~0ull;

// ------------------------------------------------------------------ bit helpers
__device__ __forceinline__ uint64_t lowmask(int nbits) {
  return nbits >= 64 ? ~0ull : nbits <= 0 ? 0ull : ((1ull << nbits) - 1);
}
// 64 bits of a long bit stream starting at bit (stream padded with >= 2 zero words)
__device__ __forceinline__ uint64_t win64(const uint64_t *__restrict__ b, uint64_t bit) {
  const uint64_t w = bit >> 6;
  const int off = (int)(bit & 63);
  uint64_t lo = b[w] >> off;
  if (off) lo |= b[w + 1] << (64 - off);
  return lo;
}
// same on a read of nl limbs (zero beyond)
__device__ __forceinline__ uint64_t win64b(const uint64_t *__restrict__ b, int nl, int bit) {
  const int w = bit >> 6, off = bit & 63;
  uint64_t lo = w < nl ? b[w] >> off : 0ull;
  if (off && w + 1 < nl) lo |= b[w + 1] << (64 - off);
  return lo;
}
__device__ __forceinline__ uint64_t rev2(uint64_t x) {  // reverse the order of the 32 2-bit groups
  const uint64_t y = __brevll(x);
  return ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
}
__device__ __forceinline__ uint64_t spread32(uint32_t x) {  // bit i -> bit 2i
  uint64_t v = x;
  v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
  v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
  v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
  v = (v | (v << 2)) & 0x3333333333333333ull;
  v = (v | (v << 1)) & 0x5555555555555555ull;
  return v;
}
// limb t (bases 32t..32t+31) of the reverse complement of a 2-bit read (complement = 3 - code)
__device__ __forceinline__ uint64_t rc_limb(const uint64_t *__restrict__ r, int nl, int len, int t) {
  const int rem = len - 32 * t;
  if (rem <= 0) return 0;
  const int s0 = len - 32 * (t + 1);
  const uint64_t w = s0 >= 0 ? win64b(r, nl, 2 * s0) : (win64b(r, nl, 0) << (2 * (-s0)));
  return (~rev2(w)) & lowmask(2 * rem);
}
// limb t (bases 64t..64t+63) of a reversed 1-bit-per-base mask
__device__ __forceinline__ uint64_t rcn_limb(const uint64_t *__restrict__ nm, int nl, int len, int t) {
  const int rem = len - 64 * t;
  if (rem <= 0) return 0;
  const int s0 = len - 64 * (t + 1);
  const uint64_t w = s0 >= 0 ? win64b(nm, nl, s0) : (win64b(nm, nl, 0) << (-s0));
  return __brevll(w) & lowmask(rem);
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
// largest c with ref_off[c] <= g (ref_off has C+1 entries, ref_off[C] = seq_len > g)
__device__ __forceinline__ uint32_t find_contig(const uint64_t *__restrict__ ref_off, uint32_t C, uint64_t g) {
  uint32_t lo = 0, hi = C;
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (ref_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------ contigs
__global__ void k_heads(const char *__restrict__ flag, uint32_t M, uint32_t *__restrict__ fs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) fs[i] = flag[i] == '0' ? i : 0u;
}
__global__ void k_tid_heads(const uint64_t *__restrict__ tid_off, int T, uint32_t M, uint32_t *__restrict__ fs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T && tid_off[t] < M) fs[tid_off[t]] = (uint32_t)tid_off[t];
}
__global__ void k_head2(const uint32_t *__restrict__ fstart, uint32_t M, uint32_t *__restrict__ h2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) h2[i] = ((i - fstart[i]) % LIST_LIMIT == 0) ? 1u : 0u;
}
__global__ void k_fill_ll(long long *p, uint32_t n, long long v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_u64(unsigned long long *p, uint64_t n, unsigned long long v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_iota(uint32_t *p, uint64_t n, uint32_t base = 0) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = base + (uint32_t)i;
}
// per-contig min(pos), max(pos+len): one atomic per wave when the wave sits inside one contig
__global__ void k_minmax(const long long *__restrict__ pos, const uint16_t *__restrict__ rlen,
                         const uint32_t *__restrict__ cid1, uint32_t M, long long *__restrict__ cmin,
                         long long *__restrict__ cmax) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = i < M;
  const uint32_t c = ok ? cid1[i] - 1 : 0xffffffffu;
  long long lo = ok ? pos[i] : LLONG_MAX, hi = ok ? pos[i] + rlen[i] : LLONG_MIN;
  const uint32_t c0 = __shfl(c, 0);
  const bool uniform = __all(c == c0) && c0 != 0xffffffffu;
  if (uniform) {
    for (int d = 32; d >= 1; d >>= 1) {
      const long long l2 = __shfl_xor(lo, d), h2 = __shfl_xor(hi, d);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&cmin[c0], lo); atomicMax(&cmax[c0], hi); }
  } else if (ok) {
    atomicMin(&cmin[c], lo);
    atomicMax(&cmax[c], hi);
  }
}
__global__ void k_reflen(const long long *__restrict__ cmin, const long long *__restrict__ cmax, uint32_t C,
                         uint32_t *__restrict__ R, unsigned long long *__restrict__ maxR) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  if (c == C) { R[c] = 0; return; }
  const unsigned long long r = (unsigned long long)(cmax[c] - cmin[c]);
  R[c] = r > 0xffffffffull ? 0xffffffffu : (uint32_t)r;
  atomicMax(maxR, r);
}
__global__ void k_keys1(const long long *__restrict__ pos, const uint32_t *__restrict__ cid1,
                        const long long *__restrict__ cmin, const uint64_t *__restrict__ ref_off, uint32_t M,
                        uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const uint32_t c = cid1[i] - 1;
  // contigs are laid out one after the other, so sorting by (contig, pos - min pos) is sorting by
  // the read's first base in the concatenated consensus
  keys[i] = ref_off[c] + (uint64_t)(pos[i] - cmin[c]);
  vals[i] = i;
}
// sorted record: x = first base in the concatenated consensus, y = read id | len << 32 | rc << 48 | singleton << 49
__global__ void k_srec(const uint64_t *__restrict__ kout, const uint32_t *__restrict__ vout,
                       const uint32_t *__restrict__ order, const char *__restrict__ rc,
                       const uint16_t *__restrict__ rlen, uint32_t M, ulonglong2 *__restrict__ srec) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  const uint32_t i = vout[k];
  ulonglong2 r;
  r.x = kout[k];
  r.y = (uint64_t)order[i] | ((uint64_t)rlen[i] << 32) | ((uint64_t)(rc[i] == 'r') << 48);
  srec[k] = r;
}

// ------------------------------------------------------------------ consensus (buildcontig, encoder.cpp:32-74)
// One block per tile of CONS_TILE consensus bases.  The records overlapping the tile are a contiguous
// range of the sorted records (found by a 256-ary block search); every thread walks whole reads and
// votes with LDS atomics; then one thread per base takes the arg max.
constexpr int CONS_TILE = 2048;
template <bool GE>  // first k in [0, M) with srec[k].x + add > bound (GE = false) or >= bound (GE = true)
__device__ __forceinline__ uint32_t block_search(const ulonglong2 *__restrict__ srec, uint32_t M, uint64_t add,
                                                 uint64_t bound) {
  uint32_t lo = 0, hi = M;  // answer in [lo, hi]
  while (hi - lo > 256) {
    const uint64_t span = hi - lo;
    const uint32_t idx = lo + (uint32_t)(span * (threadIdx.x + 1) / 257);
    const uint64_t v = srec[idx].x + add;
    const bool pred = GE ? v >= bound : v > bound;
    const int nfalse = __syncthreads_count(!pred);
    const uint32_t nlo = nfalse ? lo + (uint32_t)(span * (uint32_t)nfalse / 257) + 1 : lo;
    const uint32_t nhi = nfalse < 256 ? lo + (uint32_t)(span * (uint32_t)(nfalse + 1) / 257) : hi;
    lo = nlo;
    hi = nhi;
  }
  const uint32_t idx = lo + threadIdx.x;
  bool pred = true;
  if (idx < hi) {
    const uint64_t v = srec[idx].x + add;
    pred = GE ? v >= bound : v > bound;
  }
  const int nfalse = __syncthreads_count(!pred);
  return lo + (uint32_t)nfalse;
}
__global__ __launch_bounds__(256) void k_consensus(const ulonglong2 *__restrict__ srec, uint32_t M, uint64_t seq_len,
                                                   const uint64_t *__restrict__ reads, int S, int Lmax, bool oriented,
                                                   uint8_t *__restrict__ refc) {
  __shared__ uint32_t cnt[CONS_TILE * 4];  // [base][A, C, G, T] (chartolong, encoder.cpp:36-45)
  const uint64_t g0 = (uint64_t)blockIdx.x * CONS_TILE;
  for (int i = threadIdx.x; i < CONS_TILE * 4; i += 256) cnt[i] = 0;
  const uint32_t a = block_search<false>(srec, M, (uint64_t)Lmax, g0);        // start + Lmax > g0
  const uint32_t b = block_search<true>(srec, M, 0, g0 + CONS_TILE);          // start >= end of the tile
  __syncthreads();
  for (uint32_t k = a + threadIdx.x; k < b; k += 256) {
    const ulonglong2 r = srec[k];
    const int len = (int)((r.y >> 32) & 0xffff);
    const bool rc = ((r.y >> 48) & 1) && !oriented;  // temp.dna.<tid> reads are stored already reverse-complemented
    const uint64_t *rd = reads + (size_t)(uint32_t)r.y * S;
    const int jlo = r.x < g0 ? (int)(g0 - r.x) : 0;
    const long long jend = (long long)(g0 + CONS_TILE - r.x);
    const int jhi = jend < len ? (int)jend : len;
    int cur = -1;
    uint64_t w = 0;
    for (int j = jlo; j < jhi; j++) {
      const int jj = rc ? len - 1 - j : j;
      if ((jj >> 5) != cur) { cur = jj >> 5; w = rd[cur]; }
      int code = (int)((w >> (2 * (jj & 31))) & 3);
      if (rc) code = 3 - code;
      const int nat = code == 1 ? 2 : code == 2 ? 1 : code;  // SPRING code A0 G1 C2 T3 -> A C G T
      atomicAdd(&cnt[(int)(r.x + j - g0) * 4 + nat], 1u);
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < CONS_TILE; p += 256) {
    if (g0 + p >= seq_len) break;
    const uint4 c = *(const uint4 *)&cnt[p * 4];
    uint32_t mx = 0;
    int ind = 0;  // strict >, order A C G T; nothing covering the base -> 'A' (encoder.cpp:62-72)
    if (c.x > mx) { mx = c.x; ind = 0; }
    if (c.y > mx) { mx = c.y; ind = 1; }
    if (c.z > mx) { mx = c.z; ind = 2; }
    if (c.w > mx) { mx = c.w; ind = 3; }
    refc[g0 + p] = (uint8_t)(ind == 1 ? 2 : ind == 2 ? 1 : ind);  // back to the SPRING code
  }
}
__global__ void k_pack_ref(const uint8_t *__restrict__ refc, uint64_t seq_len, uint64_t *__restrict__ refbits,
                           uint64_t nwords) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint64_t v = 0;
  const uint64_t g0 = w * 32;
  if (g0 + 32 <= seq_len) {
    const uint64_t *q = (const uint64_t *)(refc + g0);  // g0 is a multiple of 32
    for (int h = 0; h < 4; h++) {
      const uint64_t x = q[h];
      for (int k = 0; k < 8; k++) v |= ((x >> (8 * k)) & 3ull) << (2 * (8 * h + k));
    }
  } else {
    for (int k = 0; k < 32 && g0 + k < seq_len; k++) v |= (uint64_t)(refc[g0 + k] & 3) << (2 * k);
  }
  refbits[w] = v;
}

// ------------------------------------------------------------------ singleton pool (readsingletons, encoder.h:541-570)
__global__ void k_pool_clean(const uint64_t *__restrict__ reads, const uint16_t *__restrict__ lens, int S,
                             const uint32_t *__restrict__ order_s, uint32_t ns, uint64_t *__restrict__ sread,
                             uint16_t *__restrict__ slen, uint16_t *__restrict__ ncnt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q = (uint32_t)(t / S);
  if (q >= ns) return;
  const int k = (int)(t % S);
  const uint32_t id = order_s[q];
  sread[(size_t)q * S + k] = reads[(size_t)id * S + k];
  if (k == 0) { slen[q] = lens[id]; ncnt[q] = 0; }
}
// input_N.dna records (write_dnaN_in_bits, util.cpp:322-348): u16 len, then 4 bits per base A0 G1 C2 T3 N4
__global__ void k_pool_N(const uint8_t *__restrict__ dnaN, const uint64_t *__restrict__ offN, uint32_t nN,
                         uint32_t ns, int S, int SM, uint64_t *__restrict__ sread, uint64_t *__restrict__ nmask,
                         uint16_t *__restrict__ slen, uint16_t *__restrict__ ncnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nN) return;
  const uint8_t *p = dnaN + offN[i];
  const int len = p[0] | (p[1] << 8);
  p += 2;
  const uint32_t q = ns + i;
  uint64_t *rd = sread + (size_t)q * S, *nm = nmask + (size_t)q * SM;
  int nc = 0;
  uint64_t acc = 0, accn = 0;
  for (int j = 0; j < len; j++) {
    const unsigned v = (p[j >> 1] >> (4 * (j & 1))) & 15u;
    const bool isN = v >= 4;
    acc |= (uint64_t)(isN ? 0u : v) << (2 * (j & 31));
    accn |= (uint64_t)isN << (j & 63);
    nc += isN;
    if ((j & 31) == 31) { rd[j >> 5] = acc; acc = 0; }
    if ((j & 63) == 63) { nm[j >> 6] = accn; accn = 0; }
  }
  if (len & 31) rd[len >> 5] = acc;
  if (len & 63) nm[len >> 6] = accn;
  slen[q] = (uint16_t)len;
  ncnt[q] = (uint16_t)nc;
}
__global__ void k_pool_rev(const uint64_t *__restrict__ sread, const uint64_t *__restrict__ nmask,
                           const uint16_t *__restrict__ slen, int S, int SM, uint32_t np,
                           uint64_t *__restrict__ srev, uint64_t *__restrict__ nmask_r) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q = (uint32_t)(t / S);
  if (q >= np) return;
  const int k = (int)(t % S);
  const int len = slen[q];
  // an N keeps code 0 forward and gets code 3 here: xor with the window then equals the xor of the
  // complemented window with 0, which is what the reference's reverse bitset sees
  srev[(size_t)q * S + k] = rc_limb(sread + (size_t)q * S, S, len, k);
  if (k < SM) nmask_r[(size_t)q * SM + k] = rcn_limb(nmask + (size_t)q * SM, SM, len, k);
}
// read q enters dictionary l iff len > end (bitset_util.h:83-105); a key with an N can never equal a
// consensus window, so such reads are left out (unobservable)
__global__ void k_pool_flag(const uint16_t *__restrict__ slen, const uint64_t *__restrict__ nmask, int SM,
                            uint32_t np, int start, int end, uint32_t *__restrict__ flag) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= np) return;
  bool ok = (int)slen[q] > end;
  if (ok) ok = (win64b(nmask + (size_t)q * SM, SM, start) & lowmask(end - start + 1)) == 0;
  flag[q] = ok ? 1u : 0u;
}
__global__ void k_pool_keys(const uint64_t *__restrict__ sread, int S, const uint32_t *__restrict__ flag,
                            const uint32_t *__restrict__ slot, uint32_t np, int start, int klen,
                            uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= np || !flag[q]) return;
  keys[slot[q]] = win64b(sread + (size_t)q * S, S, 2 * start) & lowmask(2 * klen);
  vals[slot[q]] = q;
}
// open addressing, 16-byte slots {key | 1<<63, start | count << 32}
__global__ void k_etab_insert(const uint64_t *__restrict__ ukeys, const uint32_t *__restrict__ ustart,
                              const uint32_t *__restrict__ ucount, uint32_t numkeys,
                              unsigned long long *__restrict__ tab, uint64_t tmask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numkeys) return;
  const unsigned long long tag = ukeys[i] | (1ull << 63);
  uint64_t h = mix64(ukeys[i]) & tmask;
  for (;;) {
    const unsigned long long old = atomicCAS(&tab[2 * h], 0ull, tag);
    if (old == 0ull) { tab[2 * h + 1] = (unsigned long long)ustart[i] | ((unsigned long long)ucount[i] << 32); break; }
    h = (h + 1) & tmask;
  }
}

// ------------------------------------------------------------------ alignment (encode, encoder.h:243-343)
struct AlignP {
  const uint64_t *refbits, *ref_off;
  uint32_t C;
  uint64_t seq_len;
  int Lmax, S;
  int dstart[2], dend[2];
  const unsigned long long *tab[2];
  uint64_t tmask[2];
  const uint32_t *ids[2];
  const uint64_t *sread, *srev;
  const uint16_t *slen, *ncnt;
  const unsigned long long *Tprev;
  unsigned long long *Tnew;
};

template <bool LIVE>
__global__ __launch_bounds__(256) void k_align(AlignP A) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A.seq_len) return;
  const uint32_t c = find_contig(A.ref_off, A.C, g);
  const uint64_t off = A.ref_off[c], R = A.ref_off[c + 1] - off, p = g - off;
  const uint64_t Lmax = (uint64_t)A.Lmax;
  if (R < Lmax || p > R - Lmax) return;  // encoder.h:232, :243
#pragma unroll 1
  for (int pr = 0; pr < 4; pr++) {
    const int rev = pr >> 1, l = pr & 1;
    if (!A.tab[l]) continue;
    const int start = A.dstart[l], end = A.dend[l], klen = end - start + 1;
    uint64_t key;
    if (!rev) {
      key = win64(A.refbits, 2 * (g + start)) & lowmask(2 * klen);
    } else {  // bases [start, end] of the reverse complement of the window
      const uint64_t x = win64(A.refbits, 2 * (g + Lmax - 1 - end)) & lowmask(2 * klen);
      key = (~(rev2(x) >> (64 - 2 * klen))) & lowmask(2 * klen);
    }
    const unsigned long long tag = key | (1ull << 63);
    uint64_t h = mix64(key) & A.tmask[l];
    unsigned long long sx;
    for (;;) {
      sx = A.tab[l][2 * h];
      if (sx == 0ull || sx == tag) break;
      h = (h + 1) & A.tmask[l];
    }
    if (sx == 0ull) continue;
    const unsigned long long sy = A.tab[l][2 * h + 1];
    const uint32_t bstart = (uint32_t)sy, bcount = (uint32_t)(sy >> 32);
    const unsigned long long Pkey = (g << 2) | (unsigned long long)pr;
    int seen = 0;
    for (long long k = (long long)bstart + bcount - 1; k >= (long long)bstart; k--) {
      const uint32_t rid = A.ids[l][k];
      if (LIVE && A.Tprev[rid] < Pkey) continue;  // taken before this probe: not in the bin any more
      if (++seen > MAX_SEARCH_E) break;
      const int nc = A.ncnt[rid];
      if (nc > THRESH_E) continue;
      const int len = A.slen[rid];
      const uint64_t bo = rev ? g + Lmax - len : g;
      const uint64_t *rd = (rev ? A.srev : A.sread) + (size_t)rid * A.S;
      int hd = nc;
      const int nl = (len + 31) >> 5;
      for (int t = 0; t < nl; t++) {
        const uint64_t x = (win64(A.refbits, 2 * bo + 64ull * t) ^ rd[t]) & lowmask(2 * (len - 32 * t));
        hd += __popcll(x);
        if (hd > THRESH_E) break;
      }
      if (hd <= THRESH_E) atomicMin(&A.Tnew[rid], Pkey);
    }
  }
}
// bitmap of contig starts over the concatenated consensus (bit seq_len is set too): lets a thread find how far
// its contig extends to the left and right with a few sequential word reads instead of a binary search
__global__ void k_contig_bits(const uint64_t *__restrict__ ref_off, uint32_t C, unsigned long long *__restrict__ cbits) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  const uint64_t g = ref_off[c];
  atomicOr(&cbits[g >> 6], 1ull << (g & 63));
}
// db = x - (start of x's contig), df = (end of x's contig) - x, both capped at `cap` (>= cap means "at least cap")
__device__ __forceinline__ void contig_span(const unsigned long long *__restrict__ cbits, uint64_t x, int cap, int &db,
                                            int &df) {
  const uint64_t w = x >> 6;
  const int o = (int)(x & 63);
  db = cap;
  {
    unsigned long long m = cbits[w] & (o == 63 ? ~0ull : ((1ull << (o + 1)) - 1));
    int dist = 0;  // distance from x down to bit 63 of the word being examined
    uint64_t ww = w;
    for (;;) {
      if (m) { db = dist + (ww == w ? o : 63) - (63 - __clzll(m)); break; }
      dist += (ww == w ? o : 63) + 1;
      if (dist >= cap || ww == 0) break;
      ww--;
      m = cbits[ww];
    }
    if (db > cap) db = cap;
  }
  df = cap;
  {
    unsigned long long m = o == 63 ? 0ull : (cbits[w] >> (o + 1));
    int dist = 1;  // distance from x to bit 0 of the shifted word
    uint64_t ww = w;
    for (;;) {
      if (m) { df = dist + (__ffsll((unsigned long long)m) - 1); break; }
      dist += (ww == w ? 63 - o : 64);
      if (dist > cap) break;
      ww++;
      m = cbits[ww];
    }
    if (df > cap) df = cap;
  }
}

// Both dictionaries have windows of the same length (always when max_readlen > 50), so one table keyed by
// the 21-mer holds the bin of dictionary 0 and the bin of dictionary 1: 32-byte slots
// {key | 1<<63, bin0, bin1, -}.  A thread then owns one 21-mer of the consensus and does 2 lookups (the
// 21-mer and its reverse complement) for the 4 probes it takes part in, instead of 4 lookups per window.
__global__ void k_mtab_insert(const uint64_t *__restrict__ ukeys, const uint32_t *__restrict__ ustart,
                              const uint32_t *__restrict__ ucount, uint32_t numkeys, int l,
                              unsigned long long *__restrict__ tab, uint64_t tmask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numkeys) return;
  const unsigned long long tag = ukeys[i] | (1ull << 63);
  uint64_t h = mix64(ukeys[i]) & tmask;
  for (;;) {
    const unsigned long long old = atomicCAS(&tab[4 * h], 0ull, tag);
    if (old == 0ull || old == tag) {
      tab[4 * h + 1 + l] = (unsigned long long)ustart[i] | ((unsigned long long)ucount[i] << 32);
      break;
    }
    h = (h + 1) & tmask;
  }
}

template <bool LIVE>
__global__ __launch_bounds__(256) void k_align_m(AlignP A, const unsigned long long *__restrict__ mtab, uint64_t mmask,
                                                 const unsigned long long *__restrict__ cbits) {
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // first base of this thread's 21-mer
  if (x >= A.seq_len) return;
  const uint64_t Lmax = (uint64_t)A.Lmax;
  const int klen = A.dend[0] - A.dstart[0] + 1;
  int db, df;  // bases of this contig at and before x / from x to its end, capped at Lmax
  contig_span(cbits, x, A.Lmax, db, df);
  if (df < klen) return;
  const uint64_t kf = win64(A.refbits, 2 * x) & lowmask(2 * klen);
  const uint64_t kr = (~(rev2(kf) >> (64 - 2 * klen))) & lowmask(2 * klen);
#pragma unroll 1
  for (int rev = 0; rev < 2; rev++) {
    const uint64_t key = rev ? kr : kf;
    const unsigned long long tag = key | (1ull << 63);
    uint64_t h = mix64(key) & mmask;
    ulonglong2 s01;
    for (;;) {
      s01 = *(const ulonglong2 *)&mtab[4 * h];  // {key, bin0}
      if (s01.x == 0ull || s01.x == tag) break;
      h = (h + 1) & mmask;
    }
    if (s01.x == 0ull) continue;
#pragma unroll 1
    for (int l = 0; l < 2; l++) {
      const unsigned long long sy = l ? mtab[4 * h + 2] : s01.y;
      const uint32_t bstart = (uint32_t)sy, bcount = (uint32_t)(sy >> 32);
      if (!bcount) continue;
      // the window this probe belongs to: forward key = window bases [dstart, dend]; reverse key = reverse
      // complement of window bases [Lmax-1-dend, Lmax-1-dstart]
      const int back = rev ? A.Lmax - 1 - A.dend[l] : A.dstart[l];
      if (back > db || A.Lmax - back > df) continue;  // the window [g, g + Lmax) must lie inside this contig
      const uint64_t g = x - (uint64_t)back;
      const unsigned long long Pkey = (g << 2) | (unsigned long long)(rev << 1) | (unsigned long long)l;
      int seen = 0;
      for (long long k = (long long)bstart + bcount - 1; k >= (long long)bstart; k--) {
        const uint32_t rid = A.ids[l][k];
        if (LIVE && A.Tprev[rid] < Pkey) continue;
        if (++seen > MAX_SEARCH_E) break;
        const int nc = A.ncnt[rid];
        if (nc > THRESH_E) continue;
        const int len = A.slen[rid];
        const uint64_t bo = rev ? g + Lmax - len : g;
        const uint64_t *rd = (rev ? A.srev : A.sread) + (size_t)rid * A.S;
        int hd = nc;
        const int nl = (len + 31) >> 5;
        for (int t = 0; t < nl; t++) {
          const uint64_t xx = (win64(A.refbits, 2 * bo + 64ull * t) ^ rd[t]) & lowmask(2 * (len - 32 * t));
          hd += __popcll(xx);
          if (hd > THRESH_E) break;
        }
        if (hd <= THRESH_E) atomicMin(&A.Tnew[rid], Pkey);
      }
    }
  }
}
__global__ void k_differs(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b,
                          uint32_t n, uint32_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) *flag = 1u;
}

// aligned singletons, listed by descending pool index so that the stable sort by probe key leaves
// reads taken by one probe in bin order from the tail (encoder.h:286-287)
__global__ void k_flag_aligned_rev(const unsigned long long *__restrict__ T, uint32_t np, uint32_t *__restrict__ f) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < np) f[j] = T[np - 1 - j] != INF ? 1u : 0u;
}
__global__ void k_gather_aligned(const unsigned long long *__restrict__ T, const uint32_t *__restrict__ f,
                                 const uint32_t *__restrict__ slot, uint32_t np, uint64_t *__restrict__ Pk,
                                 uint32_t *__restrict__ qv) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np || !f[j]) return;
  const uint32_t q = np - 1 - j;
  Pk[slot[j]] = T[q];
  qv[slot[j]] = q;
}
__global__ void k_single_rec(const uint64_t *__restrict__ Pk, const uint32_t *__restrict__ qv, uint32_t A, int Lmax,
                             const uint16_t *__restrict__ slen, uint64_t *__restrict__ keys,
                             ulonglong2 *__restrict__ frec) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A) return;
  const uint64_t P = Pk[j], g = P >> 2;
  const int rev = (int)((P >> 1) & 1);
  const uint32_t q = qv[j];
  const int len = slen[q];
  const uint64_t gstart = rev ? g + Lmax - len : g;  // pos = j (+ max_readlen - len when reversed), encoder.h:309-311
  keys[j] = gstart;
  ulonglong2 r;
  r.x = gstart;
  r.y = (uint64_t)q | ((uint64_t)len << 32) | ((uint64_t)rev << 48) | (1ull << 49);
  frec[j] = r;
}

// ------------------------------------------------------------------ noise streams (writecontig, encoder.cpp:76-109)
struct NoiseP {
  const uint32_t *vfin;
  const ulonglong2 *frec;
  uint64_t F;
  const uint64_t *refbits;
  const uint64_t *reads;
  int S, SM;
  const uint64_t *sread, *srev, *nmask, *nmask_r;
  bool oriented;             // stream reads are stored already reverse-complemented
  const uint32_t *oid;       // clean-read id of a stream record's read (null: the gather index is the id)
  const uint32_t *cumN;      // may be null (no N reads)
  const uint32_t *order_sc;  // corrected order of the pool reads
  uint32_t *nm;              // mismatches per record
  const uint64_t *noff;      // exclusive scan of nm
  char *noise;
  uint16_t *noisepos;
  uint64_t *out_pos;
  uint32_t *out_order;
  uint16_t *out_rlen;
  char *out_rc;
};
// enc_noise (encoder.h:522-541) indexed by SPRING codes: [ref A G C T][read A G C T N]
__constant__ char c_enc_noise[4][5] = {{0, '1', '0', '2', '3'}, {'1', 0, '2', '0', '3'}, {'0', '1', 0, '2', '3'},
                                       {'2', '0', '1', 0, '3'}};

template <bool WRITE>
__global__ __launch_bounds__(256) void k_noise(NoiseP N) {
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= N.F) return;
  const ulonglong2 rec = N.frec[N.vfin[f]];
  const uint64_t gpos = rec.x;
  const uint32_t id = (uint32_t)rec.y;
  const int len = (int)((rec.y >> 32) & 0xffff);
  const bool rc = (rec.y >> 48) & 1, single = (rec.y >> 49) & 1;
  const uint64_t *rd = single ? (rc ? N.srev : N.sread) + (size_t)id * N.S : N.reads + (size_t)id * N.S;
  const uint64_t *nmk = single ? (rc ? N.nmask_r : N.nmask) + (size_t)id * N.SM : nullptr;
  const int nl = (len + 31) >> 5;
  uint32_t cnt = 0;
  uint64_t o = 0;
  int prevj = 0;
  if (WRITE) o = N.noff[f];
  for (int t = 0; t < nl; t++) {
    const uint64_t r = (single || !rc || N.oriented) ? rd[t] : rc_limb(rd, N.S, len, t);
    const uint32_t nh = nmk ? (uint32_t)(nmk[t >> 1] >> (32 * (t & 1))) : 0u;
    const uint64_t w = win64(N.refbits, 2 * gpos + 64ull * t);
    const uint64_t x = w ^ r;
    uint64_t m = (((x | (x >> 1)) & 0x5555555555555555ull) | spread32(nh)) & lowmask(2 * (len - 32 * t));
    if (!WRITE) {
      cnt += __popcll(m);
    } else {
      while (m) {
        const int jj = __builtin_ctzll(m) >> 1;
        m &= m - 1;
        const int j = 32 * t + jj;
        const int rcode = (int)((w >> (2 * jj)) & 3);
        const int dcode = ((nh >> jj) & 1) ? 4 : (int)((r >> (2 * jj)) & 3);
        N.noise[o + f + cnt] = c_enc_noise[rcode][dcode];
        N.noisepos[o + cnt] = (uint16_t)(j - prevj);
        prevj = j;
        cnt++;
      }
    }
  }
  if (!WRITE) {
    N.nm[f] = cnt;
  } else {
    N.noise[o + f + cnt] = '\n';
    N.out_pos[f] = gpos;
    N.out_rlen[f] = (uint16_t)len;
    N.out_rc[f] = rc ? 'r' : 'd';
    const uint32_t cid = single ? 0u : (N.oid ? N.oid[id] : id);
    N.out_order[f] = single ? N.order_sc[id] : cid + (N.cumN ? N.cumN[cid] : 0u);
  }
}

// ------------------------------------------------------------------ order correction (correct_order, encoder.cpp:177-222)
__global__ void k_shift_N(const uint64_t *__restrict__ off, const uint32_t *__restrict__ order, uint32_t n, uint64_t byte_base,
                          uint32_t read_base, uint64_t *__restrict__ off_out, uint32_t *__restrict__ order_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  off_out[i] = off[i] + byte_base;
  order_out[i] = order[i] + read_base;
}
__global__ void k_mark_N(const uint32_t *__restrict__ order_N, uint32_t nN, uint32_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nN) flag[order_N[i]] = 1u;
}
__global__ void k_cumulative(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ nbefore, uint32_t total,
                             uint32_t *__restrict__ cum) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total && !flag[i]) cum[i - nbefore[i]] = nbefore[i];
}
__global__ void k_pool_order(const uint32_t *__restrict__ order_s, const uint32_t *__restrict__ order_N, uint32_t ns,
                             uint32_t np, const uint32_t *__restrict__ cumN, uint32_t *__restrict__ order_sc) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= np) return;
  order_sc[q] = q < ns ? order_s[q] + (cumN ? cumN[order_s[q]] : 0u) : order_N[q - ns];
}

// ------------------------------------------------------------------ unaligned reads (encoder.h:425-452)
__global__ void k_flag_rem(const unsigned long long *__restrict__ T, const uint16_t *__restrict__ slen, uint32_t np,
                           uint32_t *__restrict__ fr, uint32_t *__restrict__ usz, uint32_t *__restrict__ ulen) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > np) return;
  const bool rem = q < np && T[q] == INF;
  fr[q] = rem ? 1u : 0u;
  usz[q] = rem ? 2u + (slen[q] + 1u) / 2u : 0u;
  ulen[q] = rem ? slen[q] : 0u;
}
__global__ void k_unaligned(const uint32_t *__restrict__ fr, const uint32_t *__restrict__ uslot,
                            const uint64_t *__restrict__ uoff, uint32_t np, const uint64_t *__restrict__ sread,
                            const uint64_t *__restrict__ nmask, const uint16_t *__restrict__ slen, int S, int SM,
                            const uint32_t *__restrict__ order_sc, uint64_t n_aligned, uint32_t *__restrict__ out_order,
                            uint16_t *__restrict__ out_rlen, uint8_t *__restrict__ un) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= np || !fr[q]) return;
  const int len = slen[q];
  out_order[n_aligned + uslot[q]] = order_sc[q];
  out_rlen[n_aligned + uslot[q]] = (uint16_t)len;
  uint8_t *p = un + uoff[q];
  p[0] = (uint8_t)(len & 0xff);
  p[1] = (uint8_t)(len >> 8);
  const uint64_t *rd = sread + (size_t)q * S, *nm = nmask + (size_t)q * SM;
  for (int b = 0; b < (len + 1) / 2; b++) {  // write_dnaN_in_bits (util.cpp:322-348)
    unsigned v = 0;
    for (int h = 0; h < 2; h++) {
      const int j = 2 * b + h;
      if (j >= len) break;
      const unsigned isN = (unsigned)((nm[j >> 6] >> (j & 63)) & 1);
      const unsigned code = isN ? 4u : (unsigned)((rd[j >> 5] >> (2 * (j & 31))) & 3);
      v |= code << (4 * h);
    }
    p[2 + b] = (uint8_t)v;
  }
}

// ------------------------------------------------------------------ seq outputs
__global__ void k_tid_seq(const uint64_t *__restrict__ tid_off, int T, uint32_t M, const uint32_t *__restrict__ cid1,
                          const uint64_t *__restrict__ ref_off, uint64_t seq_len, uint64_t *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > T) return;
  out[t] = (t < T && tid_off[t] < M) ? ref_off[cid1[tid_off[t]] - 1] : seq_len;
}
__global__ void k_seq_ascii(const uint8_t *__restrict__ refc, uint64_t n, char *__restrict__ out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) out[g] = "AGCT"[refc[g] & 3];
}
// pack_compress_seq (encoder.cpp:111-156): natural code A0 C1 G2 T3, 4 bases per byte, low bits first
__global__ void k_seq_pack(const uint8_t *__restrict__ refc, uint64_t base0, uint64_t nbytes, uint8_t *__restrict__ out) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbytes) return;
  unsigned v = 0;
  for (int k = 0; k < 4; k++) {
    const unsigned s = refc[base0 + 4 * b + k] & 3u;
    const unsigned nat = s == 1 ? 2u : s == 2 ? 1u : s;
    v |= nat << (2 * k);
  }
  out[b] = (uint8_t)v;
}

inline dim3 grid(uint64_t n) { return dim3((unsigned)((n + 255) / 256 ? (n + 255) / 256 : 1)); }
inline int bits_for(uint64_t v) {  // bits needed to hold values 0..v
  int b = 1;
  while (b < 64 && (v >> b)) b++;
  return b;
}

// device buffer from the library's pool
struct DBuf {
  int dev = 0;
  void *p = nullptr;
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  ~DBuf() { release(); }
  void release() { if (p) { sr::dev_free(dev, p); p = nullptr; } }
  hipError_t alloc(int d, size_t bytes) { release(); dev = d; return sr::dev_alloc(d, bytes, &p); }
  template <class T> T *as() const { return (T *)p; }
};

}  // namespace

struct spring_encoder_ctx {
  int dev = 0;
  hipStream_t st = nullptr;      // stream the download entry points use: always `own` once an encode has finished
  hipStream_t own = nullptr;     // created on first use
  int T = 0;
  spring_encoder_info info;
  bool have = false;
  bool split_tables = false;     // spring_encoder_set_split_tables (tests)
  // results (device)
  DBuf refc, pos, noise, noisepos, order, rlen, rc, unaligned;
  std::vector<uint64_t> tid_seq;  // T + 1 offsets into refc
};

#define DALLOC(buf, bytes) HIPCHK((buf).alloc(dev, (bytes)))

extern "C" {

int spring_encoder_create(int device, spring_encoder_ctx **out) {
  if (!out) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(SPRING_REORDER_E_HIP, "no HIP device available (the encoder stage has no CPU fallback)");
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;  // like spring_reorder_create
  if (device >= ndev) return fail(SPRING_REORDER_E_ARG, "device %d out of range", device);
  spring_encoder_ctx *c = new spring_encoder_ctx();
  c->dev = device;
  memset(&c->info, 0, sizeof(c->info));
  *out = c;
  return 0;
}

int spring_encoder_set_split_tables(spring_encoder_ctx *ctx, int32_t on) {
  if (!ctx) return -1;
  ctx->split_tables = on != 0;
  return 0;
}
void spring_encoder_destroy(spring_encoder_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->dev);
  (void)hipDeviceSynchronize();
  if (ctx->own) (void)hipStreamDestroy(ctx->own);
  delete ctx;
}

}  // extern "C"

// where the reads of the records come from
struct EncSrc {
  sr::ReorderView V;      // pool + streams on the device (f_order / f_order_s = gather index into V.reads)
  bool oriented;          // stream reads are stored already reverse-complemented (temp.dna.<tid> image)
  const uint32_t *oid;    // clean-read id of stream record i (null: the gather index is the id)
  const uint32_t *oid_s;  // clean-read id of singleton q (null: V.f_order_s)
  bool n_from_view;       // the reads with N are the ones the FASTQ front end left on the device (V.N_*)
};

static int encode_core(spring_encoder_ctx *ctx, const EncSrc &E, const uint8_t *dnaN, uint64_t dnaN_bytes,
                       const uint32_t *order_N, uint32_t nN, spring_encoder_info *info_out) {
  const sr::ReorderView &V = E.V;
  const bool ndev = E.n_from_view;
  if (ndev) { nN = V.N_count[0] + V.N_count[1]; dnaN_bytes = V.N_bytes[0] + V.N_bytes[1]; }
  else if (nN && (!dnaN || !order_N)) return fail(SPRING_REORDER_E_ARG, "numreads_N > 0 but dnaN / order_N is NULL");
  const int dev = ctx->dev;
  HIPCHK(hipSetDevice(dev));
  hipStream_t st = V.st;
  ctx->st = st;
  ctx->have = false;
  const int T = V.num_thr, Lmax = V.L, S = V.S;
  ctx->T = T;
  const uint64_t M64 = V.nrec;
  const uint32_t ns = (uint32_t)V.nsing;
  if ((uint64_t)ns + nN > 4294967290ull || (uint64_t)V.n + nN > 4294967290ull)
    return fail(SPRING_REORDER_E_ARG, "too many reads");
  const uint32_t M = (uint32_t)M64, np = ns + nN;
  int SM = 1;
  while (SM * 64 < Lmax) SM <<= 1;
  spring_encoder_info &I = ctx->info;
  memset(&I, 0, sizeof(I));

  hipEvent_t ev[10];
  for (auto &e : ev) HIPCHK(hipEventCreate(&e));
  struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 10; i++) (void)hipEventDestroy(e[i]); } } evg{ev};
  HIPCHK(hipEventRecord(ev[0], st));

  // host-side scan of the N records (u16 length prefixes; a sequential dependency of nN steps)
  std::vector<uint64_t> offN(nN ? nN : 1);
  if (!ndev) {
    uint64_t o = 0;
    for (uint32_t i = 0; i < nN; i++) {
      if (o + 2 > dnaN_bytes) return fail(SPRING_REORDER_E_ARG, "input_N.dna image is truncated");
      const uint32_t len = dnaN[o] | (dnaN[o + 1] << 8);
      if ((int)len > Lmax) return fail(SPRING_REORDER_E_ARG, "N read longer than max_readlen");
      offN[i] = o;
      o += 2 + (len + 1) / 2;
    }
    if (o > dnaN_bytes) return fail(SPRING_REORDER_E_ARG, "input_N.dna image is truncated");
  }

  // scratch for the rocPRIM primitives (sized for the largest call below)
  const uint64_t FMAX = (uint64_t)M + np;
  size_t tb = 0, t2 = 0;
  HIPCHK(sr::sort_pairs(st, nullptr, t2, nullptr, nullptr, nullptr, nullptr, FMAX ? FMAX : 1, 64)); tb = std::max(tb, t2);
  HIPCHK(sr::excl_scan_u32_to_u64(st, nullptr, t2, nullptr, nullptr, FMAX + 2)); tb = std::max(tb, t2);
  HIPCHK(sr::excl_scan_u32(st, nullptr, t2, nullptr, nullptr, (size_t)V.n + nN + 2)); tb = std::max(tb, t2);
  HIPCHK(sr::rle(st, nullptr, t2, nullptr, np ? np : 1, nullptr, nullptr, nullptr)); tb = std::max(tb, t2);
  HIPCHK(sr::reduce_max_u32(st, nullptr, t2, nullptr, nullptr, np ? np : 1)); tb = std::max(tb, t2);
  {
    size_t t3 = 0;
    HIPCHK(rocprim::inclusive_scan(nullptr, t3, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)(M ? M : 1),
                                   rocprim::maximum<uint32_t>(), st));
    tb = std::max(tb, t3);
  }
  DBuf tmp;
  DALLOC(tmp, tb + 256);

  // ------------------------------------------------ order correction tables
  DBuf cumN, order_sc, dN, dorderN, doffN;
  if (nN) {
    const uint32_t total = V.n + nN;
    DBuf flagN, nb;
    DALLOC(flagN, (size_t)total * 4); DALLOC(nb, (size_t)total * 4); DALLOC(cumN, (size_t)(V.n ? V.n : 1) * 4);
    DALLOC(dN, dnaN_bytes); DALLOC(dorderN, (size_t)nN * 4); DALLOC(doffN, (size_t)nN * 8);
    if (!ndev) {
      HIPCHK(hipMemcpyAsync(dN.p, dnaN, dnaN_bytes, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(dorderN.p, order_N, (size_t)nN * 4, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(doffN.p, offN.data(), (size_t)nN * 8, hipMemcpyHostToDevice, st));
    } else {  // the two files' N reads one after the other, file-2 positions offset by the reads of file 1
      uint64_t bo = 0;     // (the merge of input_N.dna.2 / read_order_N.bin.2, preprocess.cpp:362-383)
      uint32_t ro = 0;
      for (int j = 0; j < 2; j++) {
        const uint32_t c = V.N_count[j];
        if (!c) continue;
        HIPCHK(hipMemcpyAsync(dN.as<uint8_t>() + bo, V.N_dna[j], V.N_bytes[j], hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_shift_N, grid(c), dim3(256), 0, st, V.N_off[j], V.N_order[j], c, bo,
                           j ? V.fq_num_reads_0 : 0u, doffN.as<uint64_t>() + ro, dorderN.as<uint32_t>() + ro);
        bo += V.N_bytes[j];
        ro += c;
      }
    }
    HIPCHK(hipMemsetAsync(flagN.p, 0, (size_t)total * 4, st));
    hipLaunchKernelGGL(k_mark_N, grid(nN), dim3(256), 0, st, dorderN.as<uint32_t>(), nN, flagN.as<uint32_t>());
    t2 = tb;
    HIPCHK(sr::excl_scan_u32(st, tmp.p, t2, flagN.as<uint32_t>(), nb.as<uint32_t>(), total));
    hipLaunchKernelGGL(k_cumulative, grid(total), dim3(256), 0, st, flagN.as<uint32_t>(), nb.as<uint32_t>(), total,
                       cumN.as<uint32_t>());
    HIPCHK(hipStreamSynchronize(st));  // flagN / nb go back to the pool
  }

  // ------------------------------------------------ contigs
  uint32_t C = 0;
  uint64_t seq_len = 0, maxR = 0;
  DBuf cid1, cmin, cmax, Rl, ref_off, dtid, dmax;
  DALLOC(dtid, (size_t)(T + 1) * 8);
  HIPCHK(hipMemcpyAsync(dtid.p, V.tid_off, (size_t)(T + 1) * 8, hipMemcpyHostToDevice, st));
  if (M) {
    DBuf fs, fstart, h2;
    DALLOC(fs, (size_t)M * 4); DALLOC(fstart, (size_t)M * 4); DALLOC(h2, (size_t)M * 4); DALLOC(cid1, (size_t)M * 4);
    hipLaunchKernelGGL(k_heads, grid(M), dim3(256), 0, st, V.f_flag, M, fs.as<uint32_t>());
    hipLaunchKernelGGL(k_tid_heads, grid(T), dim3(256), 0, st, dtid.as<uint64_t>(), T, M, fs.as<uint32_t>());
    size_t t3 = tb;
    HIPCHK(rocprim::inclusive_scan(tmp.p, t3, fs.as<uint32_t>(), fstart.as<uint32_t>(), (size_t)M,
                                   rocprim::maximum<uint32_t>(), st));
    hipLaunchKernelGGL(k_head2, grid(M), dim3(256), 0, st, fstart.as<uint32_t>(), M, h2.as<uint32_t>());
    t3 = tb;
    HIPCHK(rocprim::inclusive_scan(tmp.p, t3, h2.as<uint32_t>(), cid1.as<uint32_t>(), (size_t)M,
                                   rocprim::plus<uint32_t>(), st));
    HIPCHK(hipMemcpyAsync(&C, cid1.as<uint32_t>() + (M - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    DALLOC(cmin, (size_t)C * 8); DALLOC(cmax, (size_t)C * 8);
    DALLOC(Rl, (size_t)(C + 2) * 4); DALLOC(ref_off, (size_t)(C + 2) * 8); DALLOC(dmax, 8);
    hipLaunchKernelGGL(k_fill_ll, grid(C), dim3(256), 0, st, cmin.as<long long>(), C, LLONG_MAX);
    hipLaunchKernelGGL(k_fill_ll, grid(C), dim3(256), 0, st, cmax.as<long long>(), C, LLONG_MIN);
    hipLaunchKernelGGL(k_minmax, grid(M), dim3(256), 0, st, V.f_pos, V.f_len, cid1.as<uint32_t>(), M,
                       cmin.as<long long>(), cmax.as<long long>());
    HIPCHK(hipMemsetAsync(dmax.p, 0, 8, st));
    hipLaunchKernelGGL(k_reflen, grid(C + 1), dim3(256), 0, st, cmin.as<long long>(), cmax.as<long long>(), C,
                       Rl.as<uint32_t>(), dmax.as<unsigned long long>());
    t2 = tb;
    HIPCHK(sr::excl_scan_u32_to_u64(st, tmp.p, t2, Rl.as<uint32_t>(), ref_off.as<uint64_t>(), (size_t)C + 1));
    HIPCHK(hipMemcpyAsync(&maxR, dmax.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&seq_len, ref_off.as<uint64_t>() + C, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (maxR > 0xffffffffull) return fail(SPRING_REORDER_E_ARG, "a contig consensus exceeds 2^32 bases");
  } else {
    DALLOC(ref_off, 16);
    HIPCHK(hipMemsetAsync(ref_off.p, 0, 16, st));
  }
  HIPCHK(hipEventRecord(ev[1], st));

  // ------------------------------------------------ sort by (contig, relative pos), stable
  DBuf kA, kB, vA, vB, frec;
  DALLOC(kA, (FMAX ? FMAX : 1) * 8); DALLOC(kB, (FMAX ? FMAX : 1) * 8);
  DALLOC(vA, (FMAX ? FMAX : 1) * 4); DALLOC(vB, (FMAX ? FMAX : 1) * 4);
  DALLOC(frec, (FMAX ? FMAX : 1) * 16);
  const int key_bits = bits_for(seq_len);
  if (M) {
    hipLaunchKernelGGL(k_keys1, grid(M), dim3(256), 0, st, V.f_pos, cid1.as<uint32_t>(), cmin.as<long long>(),
                       ref_off.as<uint64_t>(), M, kA.as<uint64_t>(), vA.as<uint32_t>());
    t2 = tb;
    HIPCHK(sr::sort_pairs(st, tmp.p, t2, kA.as<uint64_t>(), kB.as<uint64_t>(), vA.as<uint32_t>(), vB.as<uint32_t>(), M,
                          (unsigned)key_bits));
    hipLaunchKernelGGL(k_srec, grid(M), dim3(256), 0, st, kB.as<uint64_t>(), vB.as<uint32_t>(), V.f_order, V.f_rc,
                       V.f_len, M, frec.as<ulonglong2>());
  }
  HIPCHK(hipEventRecord(ev[2], st));

  // ------------------------------------------------ consensus
  DBuf refbits;
  const uint64_t nwords = (seq_len + 31) / 32;
  DALLOC(ctx->refc, seq_len + 64);
  DALLOC(refbits, (nwords + 40) * 8);
  HIPCHK(hipMemsetAsync(refbits.p, 0, (nwords + 40) * 8, st));
  if (seq_len) {
    hipLaunchKernelGGL(k_consensus, dim3((unsigned)((seq_len + CONS_TILE - 1) / CONS_TILE)), dim3(256), 0, st,
                       frec.as<ulonglong2>(), M, seq_len, V.reads, S, Lmax, E.oriented, ctx->refc.as<uint8_t>());
    hipLaunchKernelGGL(k_pack_ref, grid(nwords), dim3(256), 0, st, ctx->refc.as<uint8_t>(), seq_len,
                       refbits.as<uint64_t>(), nwords);
  }
  HIPCHK(hipEventRecord(ev[3], st));

  // ------------------------------------------------ singleton pool + dictionaries
  DBuf sread, srev, nmask, nmask_r, slen, ncnt, Tprev, Tnew;
  DBuf tab[2], ids[2], mtab;
  uint64_t tmask[2] = {0, 0}, mmask = 0;
  bool have_tab[2] = {false, false};
  int dstart[2], dend[2];
  if (Lmax > 50) { dstart[0] = 0; dend[0] = 20; dstart[1] = 21; dend[1] = 41; }   // encoder.h:606-616
  else { dstart[0] = 0; dend[0] = 20 * Lmax / 50; dstart[1] = 20 * Lmax / 50 + 1; dend[1] = 41 * Lmax / 50; }
  // one table for both dictionaries when their windows have the same length (max_readlen > 50 and most others)
  const bool merged = (dend[0] - dstart[0]) == (dend[1] - dstart[1]) && dend[0] - dstart[0] + 1 <= 31 &&
                      !ctx->split_tables;
  uint32_t max_bin = 0;
  DALLOC(sread, (size_t)(np ? np : 1) * S * 8); DALLOC(srev, (size_t)(np ? np : 1) * S * 8);
  DALLOC(nmask, (size_t)(np ? np : 1) * SM * 8); DALLOC(nmask_r, (size_t)(np ? np : 1) * SM * 8);
  DALLOC(slen, (size_t)(np ? np : 1) * 2); DALLOC(ncnt, (size_t)(np ? np : 1) * 2);
  DALLOC(Tprev, (size_t)(np ? np : 1) * 8); DALLOC(Tnew, (size_t)(np ? np : 1) * 8);
  DALLOC(order_sc, (size_t)(np ? np : 1) * 4);
  if (np) {
    HIPCHK(hipMemsetAsync(sread.p, 0, (size_t)np * S * 8, st));
    HIPCHK(hipMemsetAsync(nmask.p, 0, (size_t)np * SM * 8, st));
    if (ns)
      hipLaunchKernelGGL(k_pool_clean, grid((uint64_t)ns * S), dim3(256), 0, st, V.reads, V.lens, S, V.f_order_s, ns,
                         sread.as<uint64_t>(), slen.as<uint16_t>(), ncnt.as<uint16_t>());
    if (nN)
      hipLaunchKernelGGL(k_pool_N, grid(nN), dim3(256), 0, st, dN.as<uint8_t>(), doffN.as<uint64_t>(), nN, ns, S, SM,
                         sread.as<uint64_t>(), nmask.as<uint64_t>(), slen.as<uint16_t>(), ncnt.as<uint16_t>());
    hipLaunchKernelGGL(k_pool_rev, grid((uint64_t)np * S), dim3(256), 0, st, sread.as<uint64_t>(), nmask.as<uint64_t>(),
                       slen.as<uint16_t>(), S, SM, np, srev.as<uint64_t>(), nmask_r.as<uint64_t>());
    hipLaunchKernelGGL(k_pool_order, grid(np), dim3(256), 0, st, E.oid_s ? E.oid_s : V.f_order_s, dorderN.as<uint32_t>(), ns, np,
                       cumN.as<uint32_t>(), order_sc.as<uint32_t>());
    if (merged) {
      uint64_t cap = 1024;
      while (cap < 4ull * np) cap <<= 1;  // <= 2 np distinct keys: load <= 0.5
      mmask = cap - 1;
      DALLOC(mtab, cap * 32);
      HIPCHK(hipMemsetAsync(mtab.p, 0, cap * 32, st));
    }
    DBuf flag, slot, keys, vals, skeys, ukeys, ucount, ustart, dnruns, dmx;
    DALLOC(flag, (size_t)(np + 1) * 4); DALLOC(slot, (size_t)(np + 1) * 4); DALLOC(keys, (size_t)np * 8);
    DALLOC(vals, (size_t)np * 4); DALLOC(skeys, (size_t)np * 8); DALLOC(ukeys, (size_t)np * 8);
    DALLOC(ucount, (size_t)(np + 1) * 4); DALLOC(ustart, (size_t)(np + 1) * 4); DALLOC(dnruns, 16); DALLOC(dmx, 16);
    for (int l = 0; l < 2; l++) {
      const int klen = dend[l] - dstart[l] + 1;
      if (klen <= 0 || klen > 32) continue;
      HIPCHK(hipMemsetAsync(flag.p, 0, (size_t)(np + 1) * 4, st));
      hipLaunchKernelGGL(k_pool_flag, grid(np), dim3(256), 0, st, slen.as<uint16_t>(), nmask.as<uint64_t>(), SM, np,
                         dstart[l], dend[l], flag.as<uint32_t>());
      t2 = tb;
      HIPCHK(sr::excl_scan_u32(st, tmp.p, t2, flag.as<uint32_t>(), slot.as<uint32_t>(), (size_t)np + 1));
      uint32_t nd = 0;
      HIPCHK(hipMemcpyAsync(&nd, slot.as<uint32_t>() + np, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (!nd) continue;
      hipLaunchKernelGGL(k_pool_keys, grid(np), dim3(256), 0, st, sread.as<uint64_t>(), S, flag.as<uint32_t>(),
                         slot.as<uint32_t>(), np, dstart[l], klen, keys.as<uint64_t>(), vals.as<uint32_t>());
      DALLOC(ids[l], (size_t)nd * 4);
      t2 = tb;
      HIPCHK(sr::sort_pairs(st, tmp.p, t2, keys.as<uint64_t>(), skeys.as<uint64_t>(), vals.as<uint32_t>(),
                            ids[l].as<uint32_t>(), nd, (unsigned)(2 * klen)));
      t2 = tb;
      HIPCHK(sr::rle(st, tmp.p, t2, skeys.as<uint64_t>(), nd, ukeys.as<uint64_t>(), ucount.as<uint32_t>(),
                     dnruns.as<uint32_t>()));
      uint32_t nk = 0;
      HIPCHK(hipMemcpyAsync(&nk, dnruns.p, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      t2 = tb;
      HIPCHK(sr::excl_scan_u32(st, tmp.p, t2, ucount.as<uint32_t>(), ustart.as<uint32_t>(), nk));
      t2 = tb;
      HIPCHK(sr::reduce_max_u32(st, tmp.p, t2, ucount.as<uint32_t>(), dmx.as<uint32_t>(), nk));
      uint32_t mb = 0;
      HIPCHK(hipMemcpyAsync(&mb, dmx.p, 4, hipMemcpyDeviceToHost, st));
      if (merged) {
        hipLaunchKernelGGL(k_mtab_insert, grid(nk), dim3(256), 0, st, ukeys.as<uint64_t>(), ustart.as<uint32_t>(),
                           ucount.as<uint32_t>(), nk, l, mtab.as<unsigned long long>(), mmask);
      } else {
        uint64_t cap = 1024;
        while (cap < 2ull * nk) cap <<= 1;
        tmask[l] = cap - 1;
        DALLOC(tab[l], cap * 16);
        HIPCHK(hipMemsetAsync(tab[l].p, 0, cap * 16, st));
        hipLaunchKernelGGL(k_etab_insert, grid(nk), dim3(256), 0, st, ukeys.as<uint64_t>(), ustart.as<uint32_t>(),
                           ucount.as<uint32_t>(), nk, tab[l].as<unsigned long long>(), tmask[l]);
      }
      HIPCHK(hipStreamSynchronize(st));
      max_bin = std::max(max_bin, mb);
      have_tab[l] = true;
    }
    HIPCHK(hipStreamSynchronize(st));  // the build scratch goes back to the pool
  }
  I.max_bin = max_bin;
  HIPCHK(hipEventRecord(ev[4], st));

  // ------------------------------------------------ alignment: fixed point of "first probe that takes the read"
  uint32_t passes = 0;
  DBuf dflag, cbits;
  DALLOC(dflag, 16);
  if (np) hipLaunchKernelGGL(k_fill_u64, grid(np), dim3(256), 0, st, Tprev.as<unsigned long long>(), (uint64_t)np, INF);
  if (np && seq_len && (have_tab[0] || have_tab[1])) {
    AlignP A;
    A.refbits = refbits.as<uint64_t>(); A.ref_off = ref_off.as<uint64_t>(); A.C = C; A.seq_len = seq_len;
    A.Lmax = Lmax; A.S = S;
    for (int l = 0; l < 2; l++) {
      A.dstart[l] = dstart[l]; A.dend[l] = dend[l];
      A.tab[l] = (have_tab[l] && !merged) ? tab[l].as<unsigned long long>() : nullptr;
      A.tmask[l] = tmask[l]; A.ids[l] = ids[l].as<uint32_t>();
    }
    A.sread = sread.as<uint64_t>(); A.srev = srev.as<uint64_t>(); A.slen = slen.as<uint16_t>(); A.ncnt = ncnt.as<uint16_t>();
    const bool live = max_bin > (uint32_t)MAX_SEARCH_E;
    if (merged) {
      const uint64_t nw = seq_len / 64 + 16;
      DALLOC(cbits, nw * 8);
      HIPCHK(hipMemsetAsync(cbits.p, 0, nw * 8, st));
      hipLaunchKernelGGL(k_contig_bits, grid((uint64_t)C + 1), dim3(256), 0, st, ref_off.as<uint64_t>(), C,
                         cbits.as<unsigned long long>());
    }
    for (;;) {
      if (passes >= 1000) return fail(SPRING_REORDER_E_STATE, "singleton alignment did not reach its fixed point");
      hipLaunchKernelGGL(k_fill_u64, grid(np), dim3(256), 0, st, Tnew.as<unsigned long long>(), (uint64_t)np, INF);
      A.Tprev = Tprev.as<unsigned long long>(); A.Tnew = Tnew.as<unsigned long long>();
      if (merged) {
        if (live) hipLaunchKernelGGL(k_align_m<true>, grid(seq_len), dim3(256), 0, st, A, mtab.as<unsigned long long>(), mmask,
                                     cbits.as<unsigned long long>());
        else hipLaunchKernelGGL(k_align_m<false>, grid(seq_len), dim3(256), 0, st, A, mtab.as<unsigned long long>(), mmask,
                                cbits.as<unsigned long long>());
      } else {
        if (live) hipLaunchKernelGGL(k_align<true>, grid(seq_len), dim3(256), 0, st, A);
        else hipLaunchKernelGGL(k_align<false>, grid(seq_len), dim3(256), 0, st, A);
      }
      passes++;
      uint32_t changed = 0;
      if (live) {
        HIPCHK(hipMemsetAsync(dflag.p, 0, 4, st));
        hipLaunchKernelGGL(k_differs, grid(np), dim3(256), 0, st, Tprev.as<unsigned long long>(),
                           Tnew.as<unsigned long long>(), np, dflag.as<uint32_t>());
        HIPCHK(hipMemcpyAsync(&changed, dflag.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
      }
      std::swap(Tprev.p, Tnew.p);
      if (!changed) break;
    }
  }
  I.align_passes = passes;
  unsigned long long *Tfin = Tprev.as<unsigned long long>();
  HIPCHK(hipEventRecord(ev[5], st));

  // ------------------------------------------------ aligned singletons join their contigs
  uint32_t A_cnt = 0;
  if (np) {
    DBuf fa, sl, Pk, qv, Pks, qvs;
    DALLOC(fa, (size_t)(np + 1) * 4); DALLOC(sl, (size_t)(np + 1) * 4);
    HIPCHK(hipMemsetAsync(fa.p, 0, (size_t)(np + 1) * 4, st));
    hipLaunchKernelGGL(k_flag_aligned_rev, grid(np), dim3(256), 0, st, Tfin, np, fa.as<uint32_t>());
    t2 = tb;
    HIPCHK(sr::excl_scan_u32(st, tmp.p, t2, fa.as<uint32_t>(), sl.as<uint32_t>(), (size_t)np + 1));
    HIPCHK(hipMemcpyAsync(&A_cnt, sl.as<uint32_t>() + np, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (A_cnt) {
      DALLOC(Pk, (size_t)A_cnt * 8); DALLOC(qv, (size_t)A_cnt * 4); DALLOC(Pks, (size_t)A_cnt * 8); DALLOC(qvs, (size_t)A_cnt * 4);
      hipLaunchKernelGGL(k_gather_aligned, grid(np), dim3(256), 0, st, Tfin, fa.as<uint32_t>(), sl.as<uint32_t>(), np,
                         Pk.as<uint64_t>(), qv.as<uint32_t>());
      t2 = tb;
      HIPCHK(sr::sort_pairs(st, tmp.p, t2, Pk.as<uint64_t>(), Pks.as<uint64_t>(), qv.as<uint32_t>(), qvs.as<uint32_t>(),
                            A_cnt, (unsigned)std::min(64, bits_for(seq_len) + 2)));
      hipLaunchKernelGGL(k_single_rec, grid(A_cnt), dim3(256), 0, st, Pks.as<uint64_t>(), qvs.as<uint32_t>(), A_cnt, Lmax,
                         slen.as<uint16_t>(), kB.as<uint64_t>() + M, frec.as<ulonglong2>() + M);
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  const uint64_t F = (uint64_t)M + A_cnt;
  const uint32_t *vfin = vB.as<uint32_t>();
  if (F) {
    hipLaunchKernelGGL(k_iota, grid(F), dim3(256), 0, st, vB.as<uint32_t>(), F, 0u);
    if (A_cnt) {
      t2 = tb;
      HIPCHK(sr::sort_pairs(st, tmp.p, t2, kB.as<uint64_t>(), kA.as<uint64_t>(), vB.as<uint32_t>(), vA.as<uint32_t>(), F,
                            (unsigned)key_bits));
      vfin = vA.as<uint32_t>();
    }
  }
  HIPCHK(hipEventRecord(ev[6], st));

  // ------------------------------------------------ noise, pos, order, rc, readlength
  uint64_t n_noisepos = 0;
  uint32_t n_unal = 0;
  DBuf fr, uslot, usz, uoff, ulen, ulsum;
  if (np) {
    DALLOC(fr, (size_t)(np + 2) * 4); DALLOC(uslot, (size_t)(np + 2) * 4); DALLOC(usz, (size_t)(np + 2) * 4);
    DALLOC(uoff, (size_t)(np + 2) * 8); DALLOC(ulen, (size_t)(np + 2) * 4); DALLOC(ulsum, (size_t)(np + 2) * 8);
    hipLaunchKernelGGL(k_flag_rem, grid(np + 1), dim3(256), 0, st, Tfin, slen.as<uint16_t>(), np, fr.as<uint32_t>(),
                       usz.as<uint32_t>(), ulen.as<uint32_t>());
    t2 = tb;
    HIPCHK(sr::excl_scan_u32(st, tmp.p, t2, fr.as<uint32_t>(), uslot.as<uint32_t>(), (size_t)np + 1));
    t2 = tb;
    HIPCHK(sr::excl_scan_u32_to_u64(st, tmp.p, t2, usz.as<uint32_t>(), uoff.as<uint64_t>(), (size_t)np + 1));
    t2 = tb;
    HIPCHK(sr::excl_scan_u32_to_u64(st, tmp.p, t2, ulen.as<uint32_t>(), ulsum.as<uint64_t>(), (size_t)np + 1));
    uint32_t rem_s = 0;
    HIPCHK(hipMemcpyAsync(&n_unal, uslot.as<uint32_t>() + np, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&rem_s, uslot.as<uint32_t>() + ns, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&I.unaligned_bytes, uoff.as<uint64_t>() + np, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&I.len_unaligned, ulsum.as<uint64_t>() + np, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    I.matched_s = ns - rem_s;
    I.matched_N = nN - (n_unal - rem_s);
  }
  I.n_aligned = F;
  I.n_total = F + n_unal;
  DBuf nm, noff;
  DALLOC(nm, (size_t)(F + 2) * 4); DALLOC(noff, (size_t)(F + 2) * 8);
  DALLOC(ctx->pos, (size_t)(F ? F : 1) * 8); DALLOC(ctx->rc, (size_t)(F ? F : 1));
  DALLOC(ctx->order, (size_t)(I.n_total ? I.n_total : 1) * 4); DALLOC(ctx->rlen, (size_t)(I.n_total ? I.n_total : 1) * 2);
  NoiseP N;
  N.vfin = vfin; N.frec = frec.as<ulonglong2>(); N.F = F;
  N.refbits = refbits.as<uint64_t>(); N.reads = V.reads; N.S = S; N.SM = SM;
  N.sread = sread.as<uint64_t>(); N.srev = srev.as<uint64_t>(); N.nmask = nmask.as<uint64_t>();
  N.nmask_r = nmask_r.as<uint64_t>(); N.oriented = E.oriented; N.oid = E.oid; N.cumN = cumN.as<uint32_t>(); N.order_sc = order_sc.as<uint32_t>();
  N.nm = nm.as<uint32_t>(); N.noff = noff.as<uint64_t>(); N.noise = nullptr; N.noisepos = nullptr;
  N.out_pos = ctx->pos.as<uint64_t>(); N.out_order = ctx->order.as<uint32_t>(); N.out_rlen = ctx->rlen.as<uint16_t>();
  N.out_rc = ctx->rc.as<char>();
  if (F) {
    HIPCHK(hipMemsetAsync(nm.as<uint32_t>() + F, 0, 4, st));
    hipLaunchKernelGGL(k_noise<false>, grid(F), dim3(256), 0, st, N);
    t2 = tb;
    HIPCHK(sr::excl_scan_u32_to_u64(st, tmp.p, t2, nm.as<uint32_t>(), noff.as<uint64_t>(), (size_t)F + 1));
    HIPCHK(hipMemcpyAsync(&n_noisepos, noff.as<uint64_t>() + F, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  I.n_noisepos = n_noisepos;
  I.noise_bytes = n_noisepos + F;
  DALLOC(ctx->noise, I.noise_bytes ? I.noise_bytes : 1);
  DALLOC(ctx->noisepos, (n_noisepos ? n_noisepos : 1) * 2);
  if (F) {
    N.noise = ctx->noise.as<char>();
    N.noisepos = ctx->noisepos.as<uint16_t>();
    hipLaunchKernelGGL(k_noise<true>, grid(F), dim3(256), 0, st, N);
  }
  HIPCHK(hipEventRecord(ev[7], st));

  // ------------------------------------------------ unaligned reads, per-tid seq offsets
  DALLOC(ctx->unaligned, I.unaligned_bytes ? I.unaligned_bytes : 1);
  if (n_unal)
    hipLaunchKernelGGL(k_unaligned, grid(np), dim3(256), 0, st, fr.as<uint32_t>(), uslot.as<uint32_t>(),
                       uoff.as<uint64_t>(), np, sread.as<uint64_t>(), nmask.as<uint64_t>(), slen.as<uint16_t>(), S, SM,
                       order_sc.as<uint32_t>(), F, ctx->order.as<uint32_t>(), ctx->rlen.as<uint16_t>(),
                       ctx->unaligned.as<uint8_t>());
  ctx->tid_seq.assign(T + 1, 0);
  if (M) {
    DBuf dts;
    DALLOC(dts, (size_t)(T + 1) * 8);
    hipLaunchKernelGGL(k_tid_seq, grid(T + 1), dim3(256), 0, st, dtid.as<uint64_t>(), T, M, cid1.as<uint32_t>(),
                       ref_off.as<uint64_t>(), seq_len, dts.as<uint64_t>());
    HIPCHK(hipMemcpyAsync(ctx->tid_seq.data(), dts.p, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  HIPCHK(hipEventRecord(ev[8], st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  for (int i = 0; i < 8; i++) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    I.ms_phase[i] = ms;
    I.ms_device += ms;
  }
  I.seq_len = seq_len;
  I.num_contigs = C;
  // `st` is borrowed from the reorder context and dies with it: everything queued on it is finished here, and the
  // download entry points use a stream this context owns
  HIPCHK(hipStreamSynchronize(st));
  if (!ctx->own) HIPCHK(hipStreamCreate(&ctx->own));
  ctx->st = ctx->own;
  ctx->have = true;
  if (info_out) *info_out = I;
  return 0;
}

extern "C" {

int spring_encoder_encode_reorder(spring_encoder_ctx *ctx, spring_reorder_ctx *reorder, const uint8_t *dnaN,
                                  uint64_t dnaN_bytes, const uint32_t *order_N, uint32_t nN,
                                  spring_encoder_info *info_out) {
  if (!ctx || !reorder) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  EncSrc E;
  int rcv = sr::reorder_view(reorder, &E.V);
  if (rcv) return rcv;
  if (E.V.dev != ctx->dev) return fail(SPRING_REORDER_E_ARG, "encoder and reorder contexts live on different devices");
  E.oriented = false;
  E.oid = nullptr;
  E.oid_s = nullptr;
  // no N reads handed over: take the ones the FASTQ front end kept on the device, if the context came from it
  E.n_from_view = !dnaN && !nN && (E.V.N_count[0] + E.V.N_count[1]) > 0;
  return encode_core(ctx, E, dnaN, dnaN_bytes, order_N, nN, info_out);
}

int spring_encoder_encode_host(spring_encoder_ctx *ctx, uint32_t max_readlen, int32_t num_thr, const uint64_t *tid_count,
                               const uint8_t *dna_stream, uint64_t dna_bytes, const uint32_t *order, const char *rc,
                               const char *flag, const int64_t *pos, const uint16_t *rlen, const uint8_t *dna_single,
                               uint64_t single_bytes, const uint32_t *order_s, uint32_t ns, const uint8_t *dnaN,
                               uint64_t dnaN_bytes, const uint32_t *order_N, uint32_t nN, spring_encoder_info *info_out) {
  if (!ctx || !tid_count || num_thr <= 0) return fail(SPRING_REORDER_E_ARG, "bad argument");
  if (max_readlen == 0 || max_readlen > (uint32_t)sr::MAX_READ_LEN) return fail(SPRING_REORDER_E_ARG, "Wrong bitset size.");
  const int dev = ctx->dev;
  HIPCHK(hipSetDevice(dev));
  if (!ctx->own) HIPCHK(hipStreamCreate(&ctx->own));
  hipStream_t st = ctx->own;
  std::vector<uint64_t> tid_off(num_thr + 1, 0);
  for (int t = 0; t < num_thr; t++) tid_off[t + 1] = tid_off[t] + tid_count[t];
  const uint64_t M = tid_off[num_thr];
  if (M + ns > 4294967290ull) return fail(SPRING_REORDER_E_ARG, "too many reads");
  if (M && (!dna_stream || !order || !rc || !flag || !pos || !rlen)) return fail(SPRING_REORDER_E_ARG, "NULL stream");
  if (ns && (!dna_single || !order_s)) return fail(SPRING_REORDER_E_ARG, "NULL singleton image");
  // record offsets: temp.dna.<tid> records follow read_lengths.bin; the singleton records carry their own length
  const uint64_t R = M + ns;
  std::vector<uint64_t> off(R ? R : 1);
  uint64_t o = 0;
  for (uint64_t i = 0; i < M; i++) { off[i] = o; o += 2 + (rlen[i] + 3u) / 4u; }
  if (o != dna_bytes) return fail(SPRING_REORDER_E_ARG, "temp.dna image does not match read_lengths (%llu vs %llu bytes)",
                                  (unsigned long long)o, (unsigned long long)dna_bytes);
  uint64_t so = 0;
  for (uint32_t q = 0; q < ns; q++) {
    if (so + 2 > single_bytes) return fail(SPRING_REORDER_E_ARG, "temp.dna.singleton image is truncated");
    const uint32_t len = dna_single[so] | (dna_single[so + 1] << 8);
    if (len > max_readlen) return fail(SPRING_REORDER_E_ARG, "singleton longer than max_readlen");
    off[M + q] = dna_bytes + so;
    so += 2 + (len + 3) / 4;
  }
  if (so > single_bytes) return fail(SPRING_REORDER_E_ARG, "temp.dna.singleton image is truncated");
  const int L = (int)max_readlen, W = (2 * L - 1) / 64 + 1;
  int S = 1;
  while (S < W) S <<= 1;
  DBuf d_dna, d_off, d_reads, d_lens, d_src, d_ssrc, d_oid, d_oids, d_rc, d_flag, d_pos, d_len;
  DALLOC(d_dna, dna_bytes + single_bytes + 16); DALLOC(d_off, (R ? R : 1) * 8);
  DALLOC(d_reads, (R ? R : 1) * S * 8); DALLOC(d_lens, (R ? R : 1) * 2);
  DALLOC(d_src, (M ? M : 1) * 4); DALLOC(d_ssrc, (size_t)(ns ? ns : 1) * 4); DALLOC(d_oid, (M ? M : 1) * 4);
  DALLOC(d_oids, (size_t)(ns ? ns : 1) * 4); DALLOC(d_rc, M ? M : 1); DALLOC(d_flag, M ? M : 1);
  DALLOC(d_pos, (M ? M : 1) * 8); DALLOC(d_len, (M ? M : 1) * 2);
  if (dna_bytes) HIPCHK(hipMemcpyAsync(d_dna.p, dna_stream, dna_bytes, hipMemcpyHostToDevice, st));
  if (so) HIPCHK(hipMemcpyAsync(d_dna.as<uint8_t>() + dna_bytes, dna_single, so, hipMemcpyHostToDevice, st));
  if (R) HIPCHK(hipMemcpyAsync(d_off.p, off.data(), R * 8, hipMemcpyHostToDevice, st));
  if (M) {
    HIPCHK(hipMemcpyAsync(d_oid.p, order, M * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_rc.p, rc, M, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_flag.p, flag, M, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_pos.p, pos, M * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_len.p, rlen, M * 2, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_iota, grid(M), dim3(256), 0, st, d_src.as<uint32_t>(), M, 0u);
  }
  if (ns) {
    HIPCHK(hipMemcpyAsync(d_oids.p, order_s, (size_t)ns * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_iota, grid(ns), dim3(256), 0, st, d_ssrc.as<uint32_t>(), (uint64_t)ns, (uint32_t)M);
  }
  if (R) sr::launch_unpack(st, d_dna.as<uint8_t>(), d_off.as<uint64_t>(), (uint32_t)R, L, W, S, 0, d_reads.as<uint64_t>(),
                           d_lens.as<uint16_t>());
  EncSrc E;
  E.V.dev = dev; E.V.st = st; E.V.n = (uint32_t)R; E.V.L = L; E.V.W = W; E.V.S = S;
  E.V.reads = d_reads.as<uint64_t>(); E.V.lens = d_lens.as<uint16_t>(); E.V.nrec = M; E.V.nsing = ns;
  E.V.f_order = d_src.as<uint32_t>(); E.V.f_order_s = d_ssrc.as<uint32_t>(); E.V.f_rc = d_rc.as<char>();
  E.V.f_flag = d_flag.as<char>(); E.V.f_pos = d_pos.as<long long>(); E.V.f_len = d_len.as<uint16_t>();
  E.V.tid_off = tid_off.data(); E.V.num_thr = num_thr;
  E.oriented = true;
  E.n_from_view = false;
  for (int j = 0; j < 2; j++) { E.V.N_dna[j] = nullptr; E.V.N_off[j] = nullptr; E.V.N_order[j] = nullptr; E.V.N_count[j] = 0; E.V.N_bytes[j] = 0; }
  E.V.fq_num_reads_0 = 0;
  E.oid = d_oid.as<uint32_t>();
  E.oid_s = d_oids.as<uint32_t>();
  const int r = encode_core(ctx, E, dnaN, dnaN_bytes, order_N, nN, info_out);
  (void)hipStreamSynchronize(st);
  return r;
}

int spring_encoder_get_info(spring_encoder_ctx *ctx, spring_encoder_info *info) {
  if (!ctx || !info) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  if (!ctx->have) return fail(SPRING_REORDER_E_STATE, "nothing encoded yet");
  *info = ctx->info;
  return 0;
}

int spring_encoder_download(spring_encoder_ctx *ctx, char *seq, uint64_t *seq_len_tid, uint64_t *pos, char *noise,
                            uint16_t *noisepos, uint32_t *order, uint16_t *rlen, char *rc, uint8_t *unaligned) {
  if (!ctx || !ctx->have) return fail(SPRING_REORDER_E_STATE, "nothing encoded yet");
  const int dev = ctx->dev;
  HIPCHK(hipSetDevice(dev));
  const spring_encoder_info &I = ctx->info;
  hipStream_t st = ctx->st;
  if (seq && I.seq_len) {
    DBuf a;
    DALLOC(a, I.seq_len);
    hipLaunchKernelGGL(k_seq_ascii, grid(I.seq_len), dim3(256), 0, st, ctx->refc.as<uint8_t>(), I.seq_len, a.as<char>());
    HIPCHK(hipMemcpyAsync(seq, a.p, I.seq_len, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  if (seq_len_tid)
    for (int t = 0; t < ctx->T; t++) seq_len_tid[t] = ctx->tid_seq[t + 1] - ctx->tid_seq[t];
  if (pos && I.n_aligned) HIPCHK(hipMemcpyAsync(pos, ctx->pos.p, I.n_aligned * 8, hipMemcpyDeviceToHost, st));
  if (noise && I.noise_bytes) HIPCHK(hipMemcpyAsync(noise, ctx->noise.p, I.noise_bytes, hipMemcpyDeviceToHost, st));
  if (noisepos && I.n_noisepos) HIPCHK(hipMemcpyAsync(noisepos, ctx->noisepos.p, I.n_noisepos * 2, hipMemcpyDeviceToHost, st));
  if (order && I.n_total) HIPCHK(hipMemcpyAsync(order, ctx->order.p, I.n_total * 4, hipMemcpyDeviceToHost, st));
  if (rlen && I.n_total) HIPCHK(hipMemcpyAsync(rlen, ctx->rlen.p, I.n_total * 2, hipMemcpyDeviceToHost, st));
  if (rc && I.n_aligned) HIPCHK(hipMemcpyAsync(rc, ctx->rc.p, I.n_aligned, hipMemcpyDeviceToHost, st));
  if (unaligned && I.unaligned_bytes)
    HIPCHK(hipMemcpyAsync(unaligned, ctx->unaligned.p, I.unaligned_bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

int spring_encoder_download_seq_packed(spring_encoder_ctx *ctx, uint8_t *packed, char *tail) {
  if (!ctx || !ctx->have) return fail(SPRING_REORDER_E_STATE, "nothing encoded yet");
  if (!packed || !tail) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  const int dev = ctx->dev;
  HIPCHK(hipSetDevice(dev));
  hipStream_t st = ctx->st;
  uint64_t total = 0;
  for (int t = 0; t < ctx->T; t++) total += (ctx->tid_seq[t + 1] - ctx->tid_seq[t]) / 4;
  DBuf d;
  DALLOC(d, total ? total : 1);
  uint64_t o = 0;
  std::vector<uint8_t> tcodes(4);
  for (int t = 0; t < ctx->T; t++) {
    const uint64_t b0 = ctx->tid_seq[t], len = ctx->tid_seq[t + 1] - b0, nb = len / 4;
    if (nb) hipLaunchKernelGGL(k_seq_pack, grid(nb), dim3(256), 0, st, ctx->refc.as<uint8_t>(), b0, nb, d.as<uint8_t>() + o);
    o += nb;
    memset(tail + 4 * t, 0, 4);
    if (len % 4) {
      HIPCHK(hipMemcpyAsync(tcodes.data(), ctx->refc.as<uint8_t>() + b0 + nb * 4, len % 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      for (uint64_t k = 0; k < len % 4; k++) tail[4 * t + k] = "AGCT"[tcodes[k] & 3];
    }
  }
  if (total) HIPCHK(hipMemcpyAsync(packed, d.p, total, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
