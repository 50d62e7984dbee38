// spring_amd/csrc/reorder_kernels.hip
//
// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for SPRING's read-reordering
// stage, plus their launch wrappers.  Reference behaviour: /root/reference/src
// reorder.h + bitset_util.{h,cpp}; each kernel names the loop it replaces.
// No MFMA here: everything is 64-bit integer / bit work bounded by random HBM
// access.
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "reorder_device.h"
#include "synth_common.h"

namespace sr {

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// orders LDS traffic between lanes of one wavefront (no block barrier: the
// waves of a block run independent chains and may exit early)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// a wave-uniform value pinned to scalar registers (the compiler otherwise turns `lane_bit ? P.x[1] : P.x[0]` back
// into a per-lane vector load from the kernel-argument buffer: a memory round trip in front of every probe)
__device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T *uni_ptr(T *p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}

// 64 bits starting at bit `bitpos` (may be negative / beyond the read) of a limb
// array staged in LDS with LDS_PAD zero limbs on both sides; `s` points at limb 0.
__device__ __forceinline__ uint64_t lds_window(const uint64_t *s, int bitpos) {
  int li = bitpos >> 6, off = bitpos & 63;
  uint64_t lo = s[li], hi = s[li + 1];
  return off ? (lo >> off) | (hi << (64 - off)) : lo;
}

__device__ __forceinline__ bool is_taken(const uint64_t *__restrict__ taken, uint32_t rid) {
  return (taken[rid >> 6] >> (rid & 63)) & 1ull;
}
// Liveness of a bin entry for this launch.  r: the entry as loaded (with DevParams::epos bit 31 = its read is taken) -> the
// read id.  One chain group: the flag IS the bitmap.  Two groups (DevParams::phases = 2): the flag is set once the read is
// taken in BOTH groups' views (k_ph_mark sets it when it folds the other group's winners into its own view), so a set flag
// still means dead -- for every launch, now and later: such an entry may be trimmed -- and a clear one means "ask this
// group's bitmap" (the read may have been taken in this view only).  trimmable: dead for every group.
__device__ __forceinline__ bool entry_dead(const DevParams &P, uint32_t &r, bool &trimmable) {
  if (P.idmask != 0xffffffffu) {
    const bool f = (r >> 31) != 0;
    r &= 0x7fffffffu;
    trimmable = f;
    if (f || P.phases != 2) return f;
    return is_taken(P.taken, r);
  }
  const bool d = is_taken(P.taken, r);
  // (no flags: with two groups a dead entry is trimmed only when the other group's view has it too; that bitmap may be
  // written beside this launch -- bits are only ever set, and a set bit holds for every later launch of either group)
  trimmable = d && (P.phases != 2 || is_taken(P.taken_other, r));
  return d;
}

// exact key -> bin map in two levels (replaces boomphf lookup + findpos + key re-check,
// reorder.h:271-285), ONE table for both dictionaries (TabView, reorder_device.h):
//   32-byte buckets [tag0..3 | pay0..3]; tag = (30-bit hash fingerprint << 2) | dict << 1 | single, tag 0 = empty slot.
//         98 % of the probes are absent keys and end at the 16 tag bytes.  The payload word is fetched on a fingerprint
//         match only (a second 16-byte load of every bucket doubles the L1 traffic of a gather, tools/wave_hop_bench.hip);
//         it sits in the cache line the tags came from.
//   The two dictionary windows are adjacent and equally long (reorder.h:751-759: start1 = end0 + 1), so the
//   window dictionary 1 looks up at shift s is the window dictionary 0 looks up at shift s + wl (forward;
//   reverse: dict 0 at s == dict 1 at s + wl).  A key of both dictionaries sits in the same bucket (same
//   hash), so ONE bucket fetch answers both probes: a failing search of a 150-base consensus needs 151
//   fetches instead of 237 (k_search, tail).
//   single = 1: the bin holds exactly one read and pay IS that read id; the key is verified
//         against the read's own window (as the reference does with the first read of a bin,
//         reorder.h:282-285), so the hot path is bucket -> read: no offsets/ids hops.
//   single = 0: pay indexes urec[dict], a 16-byte record {key, start | count << 32}.
// Home bucket: top bits of the hash, or (minz) the line of the key's minimizer + two hash bits (TabView).
// fingerprint = low 30 bits of the hash; 0 (empty) and 0x3fffffff (TAG_MARK) are never handed out
__device__ __forceinline__ uint32_t fp30_of(uint64_t h) {
  const uint32_t f = (uint32_t)h & 0x3fffffffu;
  return f == 0u ? 1u : f == 0x3fffffffu ? 0x3ffffffeu : f;
}
__device__ __forceinline__ uint64_t bucket_of(uint64_t h, int bshift) { return h >> bshift; }
__device__ __forceinline__ uint64_t bucket_mask(int bshift) { return (1ull << (64 - bshift)) - 1; }
// mix64 is a bijection (murmur3 finaliser): the key is recovered from the sorted hashes
__device__ __forceinline__ uint64_t unmix64(uint64_t x) {
  x ^= x >> 33; x *= 0x9cb4b2f8129337dbull; x ^= x >> 33; x *= 0x4f74430c22a54005ull; x ^= x >> 33;
  return x;
}
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
// ---- minimizers (TabView::minz).  Order value of a 16-mer (32 bits, SPRING code A0 G1 C2 T3, first base in the low
// bits): strand-symmetric -- the smaller of the k-mer and its reverse complement -- so that a consensus window and the
// window of the reverse consensus over the same bases have the same minimizer and ONE array of window minimizers
// serves the forward and the reverse probes of a search; xor before the multiplication so that poly-A (0) is not
// everybody's minimum.  Cheap on purpose: it only has to put the k-mers of a window in some fixed pseudo-random order.
__device__ __forceinline__ uint32_t kmer_order(uint32_t x) {
  uint32_t r = __builtin_bitreverse32(x);                           // bases reversed, the two bits of a base swapped ...
  r = ~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1));       // ... swapped back; complement = 3 - code
  return ((x < r ? x : r) ^ 0x5bd1e995u) * 0x9e3779b1u;
}
// the value a window's line is derived from: min over its 17 k-mers, mixed (a minimum is small: its top bits are not uniform)
__device__ __forceinline__ uint32_t minz_of_key(uint64_t key) {
  uint32_t m = 0xffffffffu;
#pragma unroll
  for (int q = 0; q <= MINZ_WL - MINZ_K; q++) m = min(m, kmer_order((uint32_t)(key >> (2 * q))));
  return fmix32(m);
}
// home bucket of a key; mz = minz_of_key(key) (only read when T.minz)
__device__ __forceinline__ uint64_t tab_home(const TabView &T, uint64_t h, uint32_t mz) {
  return T.minz ? (((uint64_t)(mz >> T.lshift) << 2) | ((h >> 30) & 3ull)) : bucket_of(h, T.bshift);
}
// where the keys of an over-subscribed line live instead (a function of the fingerprint and the two bucket bits only:
// that is what the insert kernel still knows of the hash)
__device__ __forceinline__ uint64_t tab_redirect_x(uint32_t x32, int bshift) { return mix64(0x6a09e667f3bcc908ull ^ x32) >> bshift; }
__device__ __forceinline__ uint64_t tab_redirect(const TabView &T, uint64_t h) {
  return tab_redirect_x(fp30_of(h) | ((uint32_t)((h >> 30) & 3ull) << 30), T.bshift);
}
// kind of the (skip+1)-th slot of dictionary l whose fingerprint matches: 0 = none (key absent), 1 = multi,
// 2 = single.  `other` is set when the other dictionary may hold the key too: a slot of it with the same
// fingerprint in a bucket this call looked at, or a full bucket (its slots may continue in the next one);
// other == false proves the key absent from the other dictionary.
// MZ = false: the caller only ever sees hash-addressed tables (the one-chain-per-wavefront kernels: no mark, no redirect)
template <bool MZ = false>
__device__ __forceinline__ int tab_find(const TabView &T, uint64_t h, uint32_t mz, int l, int skip,
                                        uint32_t &pay, bool &other) {
  const uint32_t mine = (fp30_of(h) << 2) | ((uint32_t)l << 1), theirs = mine ^ 2u;
  const uint64_t bmask = bucket_mask(T.bshift);
  uint64_t b = MZ ? tab_home(T, h, mz) : bucket_of(h, T.bshift);
  bool first = MZ && T.minz != 0;
  for (;;) {
    // tags only: 98 % of the probes end here
#if defined(SR_TAGS_NT)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t tv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(&T.buck[b * 2]));
    const uint4 t = make_uint4(tv.x, tv.y, tv.z, tv.w);
#elif defined(SR_TAGS_SC)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t tv;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(tv) : "v"(&T.buck[b * 2]) : "memory");
    const uint4 t = make_uint4(tv.x, tv.y, tv.z, tv.w);
#else
    const uint4 t = T.buck[b * 2];
#endif
    if (MZ && first) {  // an over-subscribed line: its keys live on the chain that starts at tab_redirect
      first = false;
      if (t.x == TAG_MARK) { b = tab_redirect(T, h); continue; }
    }
    other = other || (t.x & ~1u) == theirs || (t.y & ~1u) == theirs || (t.z & ~1u) == theirs || t.w != 0;
    // slots fill in order and never empty again, so the matches of `mine` all lie before the first free slot and a
    // free last slot ends the key's run; one bit per matching slot instead of a branch per slot
    uint32_t m = (uint32_t)((t.x & ~1u) == mine) | ((uint32_t)((t.y & ~1u) == mine) << 1) |
                 ((uint32_t)((t.z & ~1u) == mine) << 2) | ((uint32_t)((t.w & ~1u) == mine) << 3);
    const int c = __popc(m);
    if (c > skip) {
      for (; skip > 0; skip--) m &= m - 1;
      const int I = __ffs((int)m) - 1;
      const uint32_t Tg = I == 0 ? t.x : I == 1 ? t.y : I == 2 ? t.z : t.w;
      pay = reinterpret_cast<const uint32_t *>(T.buck)[b * 8 + 4 + I];
      return 1 + (int)(Tg & 1u);
    }
    if (t.w == 0) return 0;
    skip -= c;
    b = (b + 1) & bmask;
  }
}
// dictionary window of a read straight from its limbs ((read & mask1) >> 2*start, bitset_util.h:94-95)
__device__ __forceinline__ uint64_t read_window(const uint64_t *__restrict__ r, int S, int dstart, int klen2) {
  const int bitpos = 2 * dstart, li = bitpos >> 6, off = bitpos & 63;
  uint64_t v = r[li] >> off;
  if (off && li + 1 < S) v |= r[li + 1] << (64 - off);
  if (klen2 < 64) v &= (1ull << klen2) - 1;
  return v;
}

// Phase clocks of the round kernel (experiment builds only, -DSR_PHASE_TIMING): PT(k) adds the shader clocks since
// the last mark to bucket k of the wavefront's LDS table; k_round adds the table to Chain::pt when it ends.
#ifdef SR_PHASE_TIMING
#ifndef SR_PT_LONG
#define SR_PT_LONG 1000000  // clocks: wavefronts that ran longer are summed separately (DevParams::dbg)
#endif
__shared__ uint32_t g_pt_lds[66];  // k_round: one wavefront per block.  [k] clocks, [32 + k] visits, [64] last mark
#define PT(k)                                                                                         \
  do {                                                                                                \
    const uint32_t t_ = (uint32_t)clock64();                                                          \
    const unsigned long long ex_ = __ballot(1);                                                       \
    if ((int)(threadIdx.x & 63) == __ffsll(ex_) - 1) {                                                \
      g_pt_lds[(k)] += t_ - g_pt_lds[64]; g_pt_lds[32 + (k)] += 1u; g_pt_lds[64] = t_;                 \
    }                                                                                                 \
  } while (0)
#define PTW(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); PT(k); } while (0)
#define PT_FLUSH(c)                                                              \
  do {                                                                           \
    wave_sync();                                                                 \
    (c)->pt[threadIdx.x & 63] += g_pt_lds[threadIdx.x & 63];                     \
  } while (0)
#else
#define PT(k) do {} while (0)
#define PTW(k) do {} while (0)
#define PT_FLUSH(c) do {} while (0)
#endif

// ------------------------------------------------------- K1 unpack (readDnaFile)
// reorder.h:222-244: u16 len + ceil(len/4) raw bytes -> zero padded limbs.
// bad_len (may be null; fixed-record streams): set when a record's length field is not L -- the stream is then
// not the fixed-length stream its size suggested and the host falls back to walking the records.
__global__ void k_unpack(const uint8_t *__restrict__ dna, const uint64_t *__restrict__ off, uint32_t n,
                         int L, int W, int S, uint32_t rec_fixed, uint64_t *__restrict__ reads,
                         uint16_t *__restrict__ lens, uint32_t *__restrict__ bad_len) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t i = t / (uint32_t)S;
  int j = (int)(t % (uint32_t)S);
  if (i >= n) return;
  uint64_t o = off ? off[i] : i * (uint64_t)rec_fixed;
  uint32_t len = (uint32_t)dna[o] | ((uint32_t)dna[o + 1] << 8);
  if (bad_len && !off && len != (uint32_t)L) {
    if (j == 0) *bad_len = 1u;
    len = (uint32_t)L;  // (stay inside the record; the result is discarded)
  }
  if (j == 0) lens[i] = (uint16_t)len;
  uint32_t nb = (len + 3) / 4;
  uint64_t v = 0;
  if (j < W) {
#pragma unroll
    for (int b = 0; b < 8; b++) {
      uint32_t pos = 8u * j + b;
      if (pos < nb) v |= (uint64_t)dna[o + 2 + pos] << (8 * b);
    }
  }
  reads[i * S + j] = v;
}

// ------------------------------------------------ K2 key extraction (bitset_util.h:83-105)
__global__ void k_flag_in_dict(const uint16_t *__restrict__ lens, uint32_t n, int dend,
                               uint32_t *__restrict__ flag) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = lens[i] > dend ? 1u : 0u;
}

__global__ void k_keys(const uint64_t *__restrict__ reads, const uint16_t *__restrict__ lens,
                       const uint32_t *__restrict__ slot /* exclusive scan of flags, or null */, uint32_t n,
                       int S, int dstart, int dend, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (slot && !(lens[i] > dend)) return;
  const uint64_t *r = reads + (uint64_t)i * S;
  int bitpos = 2 * dstart, nbits = 2 * (dend - dstart + 1);
  int li = bitpos >> 6, offb = bitpos & 63;
  uint64_t v = r[li] >> offb;
  if (offb && li + 1 < S) v |= r[li + 1] << (64 - offb);
  if (nbits < 64) v &= (1ull << nbits) - 1;
  uint32_t o = slot ? slot[i] : i;
  keys[o] = mix64(v);  // the dictionary is sorted by hash: equal keys stay adjacent, buckets come out in order
  vals[o] = i;
}

// ------------------------------------------------ K3 table insert (bitset_util.h:122-217)
// one thread per unique (key, dictionary) pair; the pairs of both dictionaries arrive merged by hash = by
// bucket.  Pass 0 (OVERFLOW = false): the first four pairs of a bucket take its slots 0..3 directly (rank =
// number of predecessors with the same bucket, found by looking back at most 4 entries) -- no atomics,
// streaming writes; every pair writes its {key,start,count} record.  Pass 1 (OVERFLOW = true): the few pairs
// of rank >= 4 (under 1 % at load 0.2) claim the next free slot further on with CAS, after pass 0 has placed
// all native pairs.  A lookup scans slots in order and stops at the first empty one, so which free slot an
// overflow pair gets does not matter.
template <bool OVERFLOW>
__global__ void k_tab_insert(const uint64_t *__restrict__ mhash, const uint64_t *__restrict__ mval, uint64_t nm,
                             DictBuild d0, DictBuild d1, uint32_t *fpt, int bshift) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nm) return;
  const uint64_t h = mhash[i];
  const uint64_t b0 = bucket_of(h, bshift);
  int rank = 0;
  while (rank < 4 && i > (uint64_t)rank && bucket_of(mhash[i - 1 - rank], bshift) == b0) rank++;
  if (OVERFLOW != (rank >= 4)) return;
  const uint64_t mv = mval[i];
  const uint32_t l = (uint32_t)(mv >> 63), u = (uint32_t)mv;
  const DictBuild &d = l ? d1 : d0;
  const uint32_t st = d.ustart[u], cn = d.ucount[u];
  const bool single = cn == 1;
  const uint32_t tag = (fp30_of(h) << 2) | (l << 1) | (single ? 1u : 0u);
  const uint32_t pay = single ? d.ids[st] : u;
  if (cn >= DEEP_BIN) d.deep[atomicAdd(d.ndeep, 1u)] = u;  // bins worth trimming (k_trim_bins); none on low-coverage data
  if (cn >= BIG_BIN) atomicAdd(d.ndeep + 1, cn);
  if (cn >= MID_BIN) atomicAdd(d.ndeep + 2, cn);
  d.urec[u] = make_ulonglong2(unmix64(h), (uint64_t)st | ((uint64_t)cn << 32));
  if (!OVERFLOW) {
    fpt[b0 * 8 + rank] = tag;
    fpt[b0 * 8 + 4 + rank] = pay;
    return;
  }
  const uint64_t bmask = bucket_mask(bshift);
  uint64_t b = (b0 + 1) & bmask;  // the home bucket is full by construction
  for (;;) {
    uint32_t *bk = fpt + b * 8;
    for (int sl = 0; sl < 4; sl++) {
      if (atomicCAS(bk + sl, 0u, tag) == 0u) {
        bk[4 + sl] = pay;
        return;
      }
    }
    b = (b + 1) & bmask;
  }
}

// ---- the minimizer-addressed table (TabView::minz) is built from the same merged list in three steps:
// k_minz_prepare (streaming, merged = hash order): the bin record of every unique key, its home bucket and its
// {tag, payload} word; a radix sort of (bucket, word) by bucket; k_tab_insert_minz, two passes over the sorted list
// (pass 0: streaming -- the keys of a line that more than MINZ_HEAVY keys call home are left out and the line is marked;
// the first four keys of a bucket take its slots; pass 1: the keys of marked lines, from tab_redirect on, and the keys
// past the fourth of a bucket, from the next bucket on, claim the first free slot with CAS).
__global__ void k_minz_prepare(const uint64_t *__restrict__ mhash, const uint64_t *__restrict__ mval, uint64_t nm,
                               DictBuild d0, DictBuild d1, int lshift, uint32_t *__restrict__ bucket,
                               uint64_t *__restrict__ tagpay) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nm) return;
  const uint64_t h = mhash[i], key = unmix64(h);
  const uint64_t mv = mval[i];
  const uint32_t l = (uint32_t)(mv >> 63), u = (uint32_t)mv;
  const DictBuild &d = l ? d1 : d0;
  const uint32_t st = d.ustart[u], cn = d.ucount[u];
  const bool single = cn == 1;
  const uint32_t tag = (fp30_of(h) << 2) | (l << 1) | (single ? 1u : 0u);
  const uint32_t pay = single ? d.ids[st] : u;
  if (cn >= DEEP_BIN) d.deep[atomicAdd(d.ndeep, 1u)] = u;
  if (cn >= BIG_BIN) atomicAdd(d.ndeep + 1, cn);
  if (cn >= MID_BIN) atomicAdd(d.ndeep + 2, cn);
  d.urec[u] = make_ulonglong2(key, (uint64_t)st | ((uint64_t)cn << 32));
  bucket[i] = ((minz_of_key(key) >> lshift) << 2) | (uint32_t)((h >> 30) & 3ull);
  tagpay[i] = (uint64_t)tag | ((uint64_t)pay << 32);
}
template <bool SECOND>
__global__ void k_tab_insert_minz(const uint32_t *__restrict__ bk, const uint64_t *__restrict__ tp, uint64_t nm,
                                  uint32_t *fpt, int bshift, uint32_t *marked /* [1]: neighbourhoods marked */) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nm) return;
  const uint32_t b0 = bk[i], line = b0 >> 2;
  // is this line the home of more than MINZ_HEAVY keys?  (its entries are adjacent in the sorted list)
  int back = 0;
  while (back <= MINZ_HEAVY && i > (uint64_t)back && (bk[i - 1 - back] >> 2) == line) back++;
  bool heavy = back > MINZ_HEAVY;
  if (!heavy) {
    const uint64_t at = i - back + MINZ_HEAVY;  // the (MINZ_HEAVY + 1)-th entry of the line, if it has one
    heavy = at < nm && (bk[at] >> 2) == line;
  }
  int rank = 0;
  if (!heavy) while (rank < 4 && rank < back && bk[i - 1 - rank] == b0) rank++;
  const uint64_t w = tp[i];
  const uint32_t tag = (uint32_t)w, pay = (uint32_t)(w >> 32);
  if (!SECOND) {
    if (heavy) {
      if (back == 0) {
        for (int s = 0; s < 4; s++) fpt[((uint64_t)line * 4 + s) * 8] = TAG_MARK;
        atomicAdd(marked, 1u);
      }
    } else if (rank < 4) {
      fpt[(uint64_t)b0 * 8 + rank] = tag;
      fpt[(uint64_t)b0 * 8 + 4 + rank] = pay;
    }
    return;
  }
  if (!heavy && rank < 4) return;
  const uint64_t bmask = bucket_mask(bshift);
  uint64_t b = heavy ? tab_redirect_x((tag >> 2) | ((b0 & 3u) << 30), bshift) : (((uint64_t)b0 + 1) & bmask);
  for (;;) {
    uint32_t *bk = fpt + b * 8;
    for (int sl = 0; sl < 4; sl++) {
      if (atomicCAS(bk + sl, 0u, tag) == 0u) {
        bk[4 + sl] = pay;
        return;
      }
    }
    b = (b + 1) & bmask;
  }
}
__global__ void k_iota_tag(uint64_t *v, uint64_t n, uint64_t tag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i | tag;
}

// Compaction of deep bins.  A taken read stays taken, so dropping it from its bin never changes a result: the
// reference does exactly that (bbhashdict::remove, bitset_util.cpp:37-63), and without it every probe of a deep bin
// (PhiX-like coverage: ~900 reads per bin, nine in ten taken late in a run) walks the taken entries again, two dependent
// loads each.  Runs between rounds, over the list of deep bins only (collected by k_tab_insert): one wavefront per
// bin keeps the untaken entries, in order, at the front of the bin and shrinks its count.  (The in-scan trim of the
// TRIM kernel variants only cuts the dead tail.)
__global__ __launch_bounds__(256) void k_trim_bins(const uint32_t *__restrict__ deep, const uint32_t *__restrict__ ndeep,
                                                   ulonglong2 *__restrict__ urec, uint32_t *__restrict__ ids,
                                                   const uint64_t *__restrict__ taken, ulonglong2 *__restrict__ sig /* or null */,
                                                   uint32_t *__restrict__ epos /* or null */) {
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= *ndeep) return;
  const uint32_t u = deep[i];
  const ulonglong2 rec = urec[u];
  const uint32_t start = (uint32_t)rec.y, count = (uint32_t)(rec.y >> 32);
  uint32_t w = 0;  // entries kept so far (wave-uniform); always <= the position being read
  for (uint32_t base = 0; base < count; base += 64) {
    const uint32_t j = base + lane;
    const uint32_t r = j < count ? ids[start + j] : 0u;
    // (with epos: bit 31 of the id is the read's taken bit -- a live entry's id has it clear)
    const bool live = j < count && (epos ? !(r >> 31) : !is_taken(taken, r));
    ulonglong2 sg = make_ulonglong2(0, 0);
    if (sig && live) sg = sig[start + j];
    const uint64_t m = __ballot(live);
    const uint32_t pos = w + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    if (live && pos != j) {  // (every lane has read its entry before any lane writes)
      ids[start + pos] = r;
      if (sig) sig[start + pos] = sg;
      if (epos) epos[r] = start + pos;
    }
    w += (uint32_t)__popcll(m);
  }
  if (lane == 0 && w != count) urec[u].y = (uint64_t)start | ((uint64_t)w << 32);
}

// test hook: start/count of the bin of each key; single-read bins report count = 1 | 0x80000000
// and the read id in start[]
__global__ void k_dict_lookup(TabView tab, const ulonglong2 *__restrict__ urec, int which,
                              const uint64_t *__restrict__ reads, int S, int dstart, int klen2,
                              const uint64_t *__restrict__ keys, uint32_t nkeys, uint32_t *__restrict__ start,
                              uint32_t *__restrict__ count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nkeys) return;
  const uint64_t key = keys[i], h = mix64(key);
  const uint32_t mz = tab.minz ? minz_of_key(key) : 0u;
  uint32_t s = 0, c = 0xffffffffu;
  for (int skip = 0;; skip++) {
    uint32_t pay;
    bool other = false;
    const int kind = tab_find<true>(tab, h, mz, which, skip, pay, other);
    if (kind == 0) break;
    if (kind == 1) {
      const ulonglong2 r = urec[pay];
      if (r.x != key) continue;
      s = (uint32_t)r.y; c = (uint32_t)(r.y >> 32);
      break;
    }
    if (read_window(reads + (uint64_t)pay * S, S, dstart, klen2) != key) continue;
    s = pay; c = 1u | 0x80000000u;
    break;
  }
  start[i] = s;
  count[i] = c;
}

// ---------------------------------------------------------------- misc init
__global__ void k_fill_u32(uint32_t *p, uint64_t n, uint32_t v) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void k_init_taken(uint64_t *taken, uint64_t nwords, uint32_t n, uint32_t *ublk) {
  uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  if ((w & ((1u << (UBLK_SHIFT - 6)) - 1)) == 0) {  // untaken reads of the block this word starts (find_seed)
    const uint64_t first = w << 6;
    ublk[w >> (UBLK_SHIFT - 6)] = first >= n ? 0u : (uint32_t)(n - first < (1ull << UBLK_SHIFT) ? n - first : (1ull << UBLK_SHIFT));
  }
  uint64_t v = 0;
  if ((w + 1) * 64 > n) {  // bits >= n never become seeds
    int valid = (int)((int64_t)n - (int64_t)w * 64);
    v = valid <= 0 ? ~0ull : (valid >= 64 ? 0ull : (~0ull << valid));
  }
  taken[w] = v;
}

// --------------------------------------------------------- consensus update
//
// updaterefcount (reorder.h:110-220) for one chain by one wavefront.
// Counts live in HBM as int4 per position (A,C,T,G order of reorder.h:120),
// two buffers per chain (read cur, write nxt), so lanes never race on shifted
// columns and a speculative update of a chain that then loses its read is
// simply never committed (cnt_buf is not flipped).  Every case is
// position-parallel, including the in-place aliasing case of the reference
// (reverse match, read longer than ref_len+shift), which has the closed form
//   new[p] = old[p mod d] + sum_{k=1..p/d} onehot(base[p mod d + k d]),  d = n-shift-ref_len.
// The LITERAL template variant (tests only) instead runs the reference loops
// verbatim from lane 0 on an LDS copy.

struct WaveLds {
  uint64_t rd[16];    // limbs of the read being merged
  uint8_t code[512];  // consensus codes (2-bit SPRING code per position)
};
struct WaveLdsLiteral {
  int32_t cnt[4][512];
};

__device__ __forceinline__ int cidx_of_code(int code) {  // SPRING code A0 G1 C2 T3 -> count row A0 C1 T2 G3
  return code == 0 ? 0 : code == 1 ? 3 : code == 2 ? 1 : 2;
}
__device__ __forceinline__ int code_of_cidx(int c) {  // inttochar {A,C,T,G} -> SPRING code
  return c == 0 ? 0 : c == 1 ? 2 : c == 2 ? 3 : 1;
}
// base i of the string `current` of updaterefcount: the read, or its reverse complement
__device__ __forceinline__ int cur_base(const uint64_t *rd, int i, int n, bool rev) {
  int q = rev ? n - 1 - i : i;
  int code = (int)((rd[q >> 5] >> (2 * (q & 31))) & 3ull);
  return rev ? 3 - code : code;
}
__device__ __forceinline__ void add_hot(int4 &v, int ci) {
  v.x += ci == 0; v.y += ci == 1; v.z += ci == 2; v.w += ci == 3;
}
__device__ __forceinline__ int argmax_code(const int4 &v) {  // reorder.h:204-212: strict >, A,C,T,G order
  int mx = 0, ind = 0;
  if (v.x > mx) { mx = v.x; ind = 0; }
  if (v.y > mx) { mx = v.y; ind = 1; }
  if (v.z > mx) { mx = v.z; ind = 2; }
  if (v.w > mx) { mx = v.w; ind = 3; }
  return code_of_cidx(ind);
}

// 16 bits -> the even bit positions of 32 bits
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

// packs LDS codes[0..R) into limbs and stores ref + revref; with lds_refs (the fused round kernel) the limbs
// also go to the search's LDS copy [2][LDS_LIMBS] (zero padded either side).  Block k of 64 bases gives four
// wave-uniform ballots (bit planes of ref and of revref); lane 4k+t keeps the 32-bit half it will interleave
// (t bit 0 = high half, bit 1 = revref), so the bit interleave runs once for all 32 limbs instead of per block.
__device__ __forceinline__ void pack_consensus(WaveLds *ws, int R, int lane, Chain *c, uint64_t *lds_refs = nullptr) {
  wave_sync();
  const int nblk = (R + 63) >> 6;
  const int myk = lane >> 2;
  const bool hi = lane & 1, isrev = lane & 2;
  uint32_t a = 0, b = 0;  // bit plane 0 / 1 of this lane's 32 bases; blocks >= nblk stay zero
  for (int k = 0; k < nblk; k++) {
    const int p = k * 64 + lane;
    int cf = 0, cr = 0;
    if (p < R) {
      cf = ws->code[p];
      cr = 3 - ws->code[R - 1 - p];
    }
    const uint64_t f0 = __ballot(cf & 1), f1 = __ballot(cf & 2);
    const uint64_t r0 = __ballot(cr & 1), r1 = __ballot(cr & 2);
    if (myk == k) {
      const uint64_t x = isrev ? r0 : f0, y = isrev ? r1 : f1;
      a = (uint32_t)(hi ? x >> 32 : x);
      b = (uint32_t)(hi ? y >> 32 : y);
    }
  }
  if (lane < 32) {
    const uint32_t lo = spread16(a & 0xFFFFu) | (spread16(b & 0xFFFFu) << 1);
    const uint32_t up = spread16(a >> 16) | (spread16(b >> 16) << 1);
    const uint64_t limb = ((uint64_t)up << 32) | lo;
    uint64_t *dst = isrev ? c->revref : c->ref;
    dst[2 * myk + (hi ? 1 : 0)] = limb;
    if (lds_refs) lds_refs[(isrev ? LDS_LIMBS : 0) + LDS_PAD + 2 * myk + (hi ? 1 : 0)] = limb;
  }
}

// literal loops of reorder.h:133-212 on the LDS copy (lane 0 only; tests)
__device__ void update_literal_lane0(WaveLds *ws, WaveLdsLiteral *wl, bool reset, bool rev, int shift, int n,
                                     int &ref_len, int M) {
  int32_t(*count)[512] = wl->cnt;
  const uint64_t *rd = ws->rd;
#define CB(i) cidx_of_code(cur_base(rd, (i), n, rev))
  if (reset) {
    for (int j = 0; j < 4; j++)
      for (int i = 0; i < M; i++) count[j][i] = 0;
    for (int i = 0; i < n; i++) count[CB(i)][i] = 1;
    ref_len = n;
    for (int i = 0; i < n; i++) ws->code[i] = (uint8_t)cur_base(rd, i, n, rev);
    return;
  }
  if (!rev) {
    for (int i = 0; i < ref_len - shift; i++) {
      for (int j = 0; j < 4; j++) count[j][i] = count[j][i + shift];
      if (i < n) count[CB(i)][i] += 1;
    }
    for (int i = ref_len - shift; i < n; i++) {
      for (int j = 0; j < 4; j++) count[j][i] = 0;
      count[CB(i)][i] = 1;
    }
    ref_len = max(ref_len - shift, n);
  } else {
    if (n - shift >= ref_len) {
      for (int i = n - shift - ref_len; i < n - shift; i++) {
        for (int j = 0; j < 4; j++) count[j][i] = count[j][i - (n - shift - ref_len)];
        count[CB(i)][i] += 1;
      }
      for (int i = 0; i < n - shift - ref_len; i++) {
        for (int j = 0; j < 4; j++) count[j][i] = 0;
        count[CB(i)][i] = 1;
      }
      for (int i = n - shift; i < n; i++) {
        for (int j = 0; j < 4; j++) count[j][i] = 0;
        count[CB(i)][i] = 1;
      }
      ref_len = n;
    } else if (ref_len + shift <= M) {
      for (int i = ref_len - n + shift; i < ref_len; i++) count[CB(i - (ref_len - n + shift))][i] += 1;
      for (int i = ref_len; i < ref_len + shift; i++) {
        for (int j = 0; j < 4; j++) count[j][i] = 0;
        count[CB(i - (ref_len - n + shift))][i] = 1;
      }
      ref_len = ref_len + shift;
    } else {
      for (int i = 0; i < M - shift; i++)
        for (int j = 0; j < 4; j++) count[j][i] = count[j][i + (ref_len + shift - M)];
      for (int i = M - n; i < M - shift; i++) count[CB(i - (M - n))][i] += 1;
      for (int i = M - shift; i < M; i++) {
        for (int j = 0; j < 4; j++) count[j][i] = 0;
        count[CB(i - (M - n))][i] = 1;
      }
      ref_len = M;
    }
  }
#undef CB
  for (int i = 0; i < ref_len; i++) {
    int mx = 0, ind = 0;
    for (int j = 0; j < 4; j++)
      if (count[j][i] > mx) { mx = count[j][i]; ind = j; }
    ws->code[i] = (uint8_t)code_of_cidx(ind);
  }
}

// Computes one updaterefcount() into the chain's spare count buffer and into
// ws->code; nothing chain-visible is modified (commit = pack_consensus + header
// write by the caller).  NP = positions per lane (ceil(Lpad/64) <= NP).
// Returns the new ref_len.
// Count columns travel as one byte per count (cnt8) whenever every count of the new state fits, else as int4
// (cnt): the caller first asks for bytes (out_wide = false) and repeats the update in the wide format if
// `overflow` comes back set -- seen in practice only on very deep coverage.  cur_wide = format of the committed
// buffer.  4x fewer bytes on the dominant traffic of k_apply.
__device__ __forceinline__ int4 unpack8(uint32_t u) {
  return make_int4((int)(u & 255u), (int)((u >> 8) & 255u), (int)((u >> 16) & 255u), (int)(u >> 24));
}
__device__ __forceinline__ uint32_t pack8(const int4 &t) {
  return (uint32_t)t.x | ((uint32_t)t.y << 8) | ((uint32_t)t.z << 16) | ((uint32_t)t.w << 24);
}
template <int NP, bool LITERAL>
__device__ __forceinline__ int wave_update_compute(const DevParams &P, uint32_t cid, WaveLds *ws, WaveLdsLiteral *wl,
                                                   uint32_t rid, int n, bool reset, bool rev, int shift, int R,
                                                   int cb, bool cur_wide, bool out_wide, bool &overflow, int lane) {
  const int M = P.L, W = P.W;
  const int4 *__restrict__ cur = P.cnt + ((uint64_t)cid * 2 + cb) * P.Lpad;
  int4 *__restrict__ nxt = P.cnt + ((uint64_t)cid * 2 + (cb ^ 1)) * P.Lpad;
  const uint32_t *__restrict__ cur8 = P.cnt8 + ((uint64_t)cid * 2 + cb) * P.Lpad;
  uint32_t *__restrict__ nxt8 = P.cnt8 + ((uint64_t)cid * 2 + (cb ^ 1)) * P.Lpad;
#define LOAD_CNT(I) (cur_wide ? cur[(I)] : unpack8(cur8[(I)]))
  // case parameters: out position p < hiP gets (p<cpy_hi ? cur[p+src_off] : 0) + (add_lo<=p<add_hi ? onehot(base[p-add_lo]) : 0)
  int hiP, cpy_hi, src_off, add_lo, add_hi, Rn, d = 0;
  bool alias = false;
  if (reset) { hiP = M; cpy_hi = 0; src_off = 0; add_lo = 0; add_hi = n; Rn = n; }
  else if (!rev) { Rn = max(R - shift, n); hiP = Rn; cpy_hi = R - shift; src_off = shift; add_lo = 0; add_hi = n; }
  else if (n - shift >= R) { Rn = n; hiP = n; cpy_hi = R; src_off = 0; add_lo = 0; add_hi = n; d = n - shift - R; alias = d > 0; }
  else if (R + shift <= M) { Rn = R + shift; hiP = Rn; cpy_hi = R; src_off = 0; add_lo = R - n + shift; add_hi = Rn; }
  else { Rn = M; hiP = M; cpy_hi = M - shift; src_off = R + shift - M; add_lo = M - n; add_hi = M; }

  // issue every load first: read limbs + this lane's old columns
  uint64_t myl = 0;
  if (lane < 16 && lane < W) myl = P.reads[(uint64_t)rid * P.S + lane];
  int4 v[NP];
  if (!LITERAL) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int p = k * 64 + lane;
      v[k] = make_int4(0, 0, 0, 0);
      if (!alias) { if (p < cpy_hi) v[k] = LOAD_CNT(p + src_off); }
      else if (p >= d && p < n - shift) v[k] = LOAD_CNT(p % d);
    }
  }
  if (lane < 16) ws->rd[lane] = myl;
  wave_sync();
  PTW(1);

  if (LITERAL) {
    for (int p = lane; p < M; p += 64) {
      int4 t = reset ? make_int4(0, 0, 0, 0) : LOAD_CNT(p);
      wl->cnt[0][p] = t.x; wl->cnt[1][p] = t.y; wl->cnt[2][p] = t.z; wl->cnt[3][p] = t.w;
    }
    wave_sync();
    int Rl = R;
    if (lane == 0) update_literal_lane0(ws, wl, reset, rev, shift, n, Rl, M);
    Rl = __shfl(Rl, 0, 64);
    wave_sync();
    int mxl = 0;
    for (int p = lane; p < M; p += 64) mxl = max(max(mxl, max(wl->cnt[0][p], wl->cnt[1][p])), max(wl->cnt[2][p], wl->cnt[3][p]));
    overflow = __any(mxl > 255);
    for (int p = lane; p < M; p += 64) {
      const int4 t = make_int4(wl->cnt[0][p], wl->cnt[1][p], wl->cnt[2][p], wl->cnt[3][p]);
      if (out_wide) nxt[p] = t; else nxt8[p] = pack8(t);
    }
    return Rl;
  }
  int mx = 0;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    const int p = k * 64 + lane;
    if (p < hiP) {
      int4 t = v[k];
      int code = 0;
      if (!alias) {
        if (p >= add_lo && p < add_hi) {
          code = cur_base(ws->rd, p - add_lo, n, rev);
          add_hot(t, cidx_of_code(code));
        }
      } else {  // reverse case 1 with d > 0 (reorder.h:159-174), closed form of the in-place loop
        if (p >= d && p < n - shift) {
          const int r = p % d;
          for (int q = r + d; q <= p; q += d) add_hot(t, cidx_of_code(cur_base(ws->rd, q, n, rev)));
        } else {
          t = make_int4(0, 0, 0, 0);
          add_hot(t, cidx_of_code(cur_base(ws->rd, p, n, rev)));
        }
      }
      if (out_wide) nxt[p] = t; else nxt8[p] = pack8(t);
      mx = max(max(mx, max(t.x, t.y)), max(t.z, t.w));
      if (p < Rn) ws->code[p] = (uint8_t)(reset ? code : argmax_code(t));  // reset: consensus = the read itself
    }
  }
  overflow = __any(mx > 255);
  PT(2);
  return Rn;
#undef LOAD_CNT
}

// The chain header is wave-uniform.  load_hot: ONE scalar load puts its 16 dwords into SGPRs (the chain pointer is
// wave-uniform; the scalar cache is invalidated at every kernel start and a kernel never re-reads a header it has
// written); every update is scalar ALU on those registers.  store_hot: lane q writes quarter q (one store
// instruction); [q_lo, q_hi) = the quarters that changed.
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void load_hot(const Chain *c, ChainHot &h) {
  u32x16_t v;
  const Chain *cu = uni_ptr(c);  // (a chain belongs to one wavefront: the pointer is wave-uniform, say so)
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(cu) : "memory");
  // (field by field: the header must fall apart into independent scalars, not travel as one 512-bit value)
  h.ref_pos = (long long)(((unsigned long long)v[1] << 32) | v[0]);
  h.ref_len = (int32_t)v[2]; h.e_slot = v[3];
  h.prev = v[4]; h.first_rid = v[5]; h.n_emit = v[6]; h.n_single = v[7];
  h.s_slot = v[8]; h.num_reads_thr = v[9]; h.num_unmatched_past = v[10]; h.prop_rid = v[11];
  h.flags = v[12];
  h.alt1 = v[13];
  h.pad[0] = h.pad[1] = 0;
}
__device__ __forceinline__ void store_hot(Chain *c, const ChainHot &h, int lane, int q_lo = 0, int q_hi = 4) {
  if (lane >= q_lo && lane < q_hi) {
    const uint32_t plo = (uint32_t)(unsigned long long)h.ref_pos, phi = (uint32_t)((unsigned long long)h.ref_pos >> 32);
    uint4 q;
    q.x = lane == 0 ? plo : lane == 1 ? h.prev : lane == 2 ? h.s_slot : h.flags;
    q.y = lane == 0 ? phi : lane == 1 ? h.first_rid : lane == 2 ? h.num_reads_thr : h.alt1;
    q.z = lane == 0 ? (uint32_t)h.ref_len : lane == 1 ? h.n_emit : lane == 2 ? h.num_unmatched_past : 0u;
    q.w = lane == 0 ? h.e_slot : lane == 1 ? h.n_single : lane == 2 ? h.prop_rid : 0u;
    reinterpret_cast<uint4 *>(&c->h)[lane] = q;
  }
}
__device__ __forceinline__ uint32_t uni_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ----------------------------------------------------------- chain start-up
// reorder.h:405-431 with the critical section entered in chain-id order.
template <int NP>
__global__ __launch_bounds__(256) void k_init_chains(DevParams P) {
  __shared__ WaveLds lds[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t li = P.g0 + blockIdx.x * 4 + wave;  // (the launch's local chains [g0, g0 + Kg): all of them unless a pool runs two groups)
  if (li >= P.g0 + P.Kg) return;
  const uint32_t cid = P.c0 + li;  // global chain id
  Chain *c = &P.chains[li];
  WaveLds *ws = &lds[wave];
  const uint32_t step = P.n / P.Ktot;
  const uint32_t seed = cid * step;
  const bool start = P.n > 0 && (cid == 0 || step > 0);
  ChainHot h;
  memset(&h, 0, sizeof(h));
  if (lane == 0) {  // the chain's first chunk of either append buffer is pre-assigned
    P.e_chunk[li] = make_uint2(li, 0u);
    P.s_chunk[li] = make_uint2(li, 0u);
  }
  if (!start) {
    h.done = 1;
    store_hot(c, h, lane);
    if (lane == 0 && P.prop) P.prop[cid] = (unsigned long long)PK_DONE << 32;
    return;
  }
  const int n = P.uniform_len ? P.L : (int)P.lens[seed];
  bool ovf0;  // a fresh seed's counts are 0 / 1: bytes
  const int Rn = wave_update_compute<NP, false>(P, li, ws, nullptr, seed, n, true, false, 0, 0, 0, false, false, ovf0, lane);
  pack_consensus(ws, Rn, lane, c);
  h.prev = seed; h.first_rid = seed; h.prev_unmatched = 1;
  h.e_slot = li * CHUNK; h.s_slot = li * CHUNK;  // first chunk is pre-assigned; Globals.*_alloc start at K*CHUNK
  h.ref_len = Rn; h.cnt_buf = 1; h.cnt_wide = 0;
  if (P.fused) h.prop_kind = PROP_FRESH;
  store_hot(c, h, lane);
  if (lane == 0) {
    c->n_unmatched = 1;
    if (P.prop) P.prop[cid] = (unsigned long long)PK_NONE << 32;
  }
}
// a read has been taken: its bin entries say so (DevParams::epos)
__device__ __forceinline__ void mark_dead(const DevParams &P, uint32_t rid) {
#pragma unroll
  for (int l = 0; l < 2; l++) {
    if (!P.epos[l]) continue;
    const uint32_t e = P.epos[l][rid];
    if (e != 0xffffffffu) atomicOr(const_cast<uint32_t *>(P.ids[l]) + e, 0x80000000u);
  }
}
__global__ void k_build_epos(const uint32_t *__restrict__ ids, uint64_t m, uint32_t *__restrict__ epos) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) epos[ids[i]] = (uint32_t)i;
}
void launch_build_epos(hipStream_t st, const uint32_t *ids, uint64_t m, uint32_t *epos) {
  if (m) hipLaunchKernelGGL(k_build_epos, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, ids, m, epos);
}
// every rank marks the initial seeds of ALL chains (taken[] is replicated)
__global__ void k_init_seeds(DevParams P) {
  const uint32_t cid = blockIdx.x * blockDim.x + threadIdx.x;
  if (cid >= P.Ktot) return;
  const uint32_t step = P.n / P.Ktot;
  if (!(P.n > 0 && (cid == 0 || step > 0))) return;
  const uint32_t seed = cid * step;
  atomicOr((unsigned long long *)&P.taken[seed >> 6], 1ull << (seed & 63));
  atomicSub(&P.ublk[seed >> UBLK_SHIFT], 1u);
  mark_dead(P, seed);
}

// (rank+1)-th highest untaken read at or below the cursor, rank = number of seed-needing chains
// with a lower id (reorder.h:576-592 with one global cursor).  Wave-cooperative.  Returns -1 when
// the pool is exhausted; *is_last = this chain proposes the lowest seed of the round.
__device__ __forceinline__ long long find_seed(const DevParams &P, uint32_t cid, int lane, bool *is_last) {
  int r = 0, tot = 0;
  long long top = *P.cursor;  // issued together with the loads below
  const uint32_t nw = (P.Ktot + 31) / 32;
  const uint32_t myw = cid >> 5;
  if (P.needy_cnt) {
    // multi-GPU pools (k_mg_bits fills needy_cnt): needy_cnt[b] = seed-needing chains among chains
    // [2048 b, 2048 b + 2048), then the 64 bitmap words of this chain's own block -- two independent loads per
    // lane, whatever the total number of chains
    const uint32_t myblk = myw >> 6;  // (ranks count the chains of this launch's group: blocks [nb_lo, nb_hi) of 2048 chains)
    {
      const uint32_t w = myblk * 64 + lane;
      const uint32_t v = w < nw ? P.needy[w] : 0u;
      if (w < myw) r += __popc(v);
      else if (w == myw) r += __popc(v & ((1u << (cid & 31)) - 1u));
    }
    for (uint32_t b = P.nb_lo + lane; b < P.nb_hi; b += 64) {
      const uint32_t v = P.needy_cnt[b];
      tot += (int)v;
      if (b < myblk) r += (int)v;
    }
  } else {
    // one GPU: every lane takes a contiguous run of the bitmap (padded to 64 x 4 words) as independent 16-byte
    // loads -- one round trip instead of a word-at-a-time walk
    const uint32_t wpl = ((nw + 255) / 256) * 4;  // words per lane, multiple of 4
    const uint4 *nv = reinterpret_cast<const uint4 *>(P.needy) + (size_t)lane * (wpl / 4);
    const uint32_t w0 = lane * wpl;
#pragma unroll 8
    for (uint32_t j = 0; j < wpl / 4; j++) {
      const uint4 q = nv[j];
      const uint32_t vv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t w = w0 + 4 * j + k, v = vv[k];
        tot += __popc(v);
        if (w < myw) r += __popc(v);
        else if (w == myw) r += __popc(v & ((1u << (cid & 31)) - 1u));
      }
    }
  }
  const uint32_t rank = (uint32_t)wave_sum_i(r), nneedy = (uint32_t)wave_sum_i(tot);
  *is_last = rank + 1 == nneedy;
  uint32_t need = rank + 1;
  // At most two dependent steps per 1 M reads, whatever the rank and however few reads are left.  P.ublk[b] = reads of
  // block b (reads [b << UBLK_SHIFT, (b + 1) << UBLK_SHIFT)) not yet claimed by a MATCH (kept by whoever sets such a
  // taken bit).  Seeds are always taken from the top and every read above the cursor is taken, so below the cursor's
  // block ublk[] is the exact number of untaken reads; the cursor's own block is counted from its bitmap words
  // (WPL per lane, highest first), which are fetched together with the counts of the 63 blocks below it.
  if (top < (long long)P.seed_lo) return -1;  // (seeds come from reads [seed_lo, ...): the whole pool unless the chains run in two groups)
  const long long blo = (long long)(P.seed_lo >> UBLK_SHIFT);
  const long long bt = top >> UBLK_SHIFT;
  constexpr int WPB_ = 1 << (UBLK_SHIFT - 6);  // bitmap words per block
  constexpr int WPL = WPB_ / 64;               // ... per lane
  static_assert(WPL >= 1 && WPL * 64 == WPB_, "a block is a whole number of bitmap words per lane");
  auto pick_in_block = [&](long long blk, const uint64_t *uu, int cnt, uint32_t want) -> long long {
    const long long wtop = blk * WPB_ + (WPB_ - 1);
    const int inc2 = wave_incl_scan_i(cnt, lane);
    const uint64_t m = __ballot((uint32_t)inc2 >= want);
    const int wl = __ffsll((unsigned long long)m) - 1;
    int kth = (int)want - __shfl(inc2 - cnt, wl, 64);  // kth highest untaken read of lane wl's words
    long long seed = -1;
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      uint64_t v = shfl_u64(uu[k], wl);
      const int c = __popcll(v);
      if (seed < 0) {
        if (kth <= c) {
          for (int t = 1; t < kth; t++) v &= ~(1ull << (63 - __clzll(v)));
          seed = (wtop - WPL * wl - k) * 64 + (63 - __clzll(v));
        } else kth -= c;
      }
    }
    return seed;
  };
  auto load_block = [&](long long blk, uint64_t *uu) -> int {
    const long long wl0 = blk * WPB_ + (WPB_ - 1) - WPL * lane;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < WPL; k++) {
      uu[k] = ~P.taken[wl0 - k];  // (the bitmap is padded to whole blocks; bits >= n are set)
      cnt += __popcll(uu[k]);
    }
    return cnt;
  };
  uint64_t uu[WPL];
  {
    const long long b = bt - 1 - lane;
    int u = b >= blo ? (int)P.ublk[b] : 0;  // (in flight together with the cursor block's words)
    const int cnt = load_block(bt, uu);
    const uint32_t tot0 = (uint32_t)wave_sum_i(cnt);
    if (tot0 >= need) return pick_in_block(bt, uu, cnt, need);
    need -= tot0;
    for (long long b0 = bt - 1; b0 >= blo; b0 -= 64) {
      if (b0 != bt - 1) { const long long bb = b0 - lane; u = bb >= blo ? (int)P.ublk[bb] : 0; }
      const int incl = wave_incl_scan_i(u, lane);
      const uint32_t total = (uint32_t)__shfl(incl, 63, 64);
      if (total < need) { need -= total; continue; }
      const uint64_t mb = __ballot((uint32_t)incl >= need);
      const int wb = __ffsll((unsigned long long)mb) - 1;
      need -= (uint32_t)__shfl(incl - u, wb, 64);  // the need-th highest untaken read of block b0 - wb
      const int c2 = load_block(b0 - wb, uu);
      return pick_in_block(b0 - wb, uu, c2, need);
    }
  }
  return -1;
}

// A candidate read is compared STAGE_LIMBS limbs at a time: its limbs go from global memory straight into the
// wavefront's LDS staging area (global_load_lds; the deep-bin variants 16 bytes per lane and instruction), all in flight together --
// one memory round trip per candidate instead of one per limb, and no registers held for the data while it is in
// flight (the round kernel runs at 64 VGPRs; limbs preloaded into registers spill).
constexpr int STAGE_LIMBS = 5;
constexpr int STAGE_WORDS = 2 * STAGE_LIMBS * 64;  // uint32_t per wavefront
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;  // the staging rows are addressed as LDS (no generic-pointer checks)
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;
constexpr int STAGE_ODD = 512;  // dword offset of the odd limb's two rows, behind the two quad rows
typedef const __attribute__((address_space(1))) void glb_void_t;
#ifndef SR_ROUND_WAVES
#define SR_ROUND_WAVES 8  // minimum waves per SIMD the round kernel is compiled for (64 VGPRs)
#endif

// Hamming of candidate r against the shifted consensus over bases [lo, min(mref, len_r))
// (mask[0][..] / mask[shift][..] of reorder.h:291-301); with check_key also the reference's key re-check of
// a single-read bin (reorder.h:282-285): -1 = the read's window is not `key` (fingerprint collision), else 0 / 1.
// spec_taken (may be null): the candidate's limbs are fetched BEFORE it is known whether the read is still free -- its word
// of the taken bitmap is loaded beside them and comes back in *spec_taken (one memory round trip per candidate instead of
// two; a taken candidate costs a wasted 64-byte read).  -2: the read is taken (nothing was compared).
template <bool QUAD, bool UNROLL = false>
__device__ __forceinline__ int cmp_candidate(const DevParams &P, const uint64_t *sx, int bitshift, int lo, int mref, int ds,
                                           int klen2, uint32_t r, bool check_key, lds_u32_t *stage, int lane,
                                           const uint64_t *__restrict__ spec_taken = nullptr) {
  const int W = P.W;
  const int clen = P.uniform_len ? P.L : (int)P.lens[r];
  const int m = clen < mref ? clen : mref;
  const uint64_t *__restrict__ rdp = P.reads + (uint64_t)r * P.S;
  const int blo = 2 * lo, bhi = 2 * m;
  // The key re-check of a single-read bin needs no load of its own: `key` is the consensus window that the
  // alignment puts on the read's own key window [ka, kb), so the keys are equal iff the XOR below is zero on those
  // bits (a valid probe keeps the window inside the compared range).
  const int ka = 2 * ds, kb = ka + klen2;
  uint32_t kdiff = 0;
  int hd = 0;
  // only the first and the last limb of the compared range [blo, bhi) are partial; the limbs between need no mask
  const int first = blo >> 6, last = (bhi - 1) >> 6;
  for (int i0 = 0; i0 < W; i0 += STAGE_LIMBS) {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(rdp + i0);
    // Plain: dword d of the chunk -> staging row d (256 bytes).  The instruction offset moves the global address and
    // the LDS address together, so a row's base is given less that offset (one address register for all the loads).
    // QUAD (the deep-bin variants): gfx950 loads 16 bytes per lane straight into LDS (global_load_lds_dwordx4: LDS
    // address = uniform base + 16 * lane): the chunk's first 4 nq dwords go into nq rows of 1024 bytes, an odd limb's
    // two dwords into two 256-byte rows behind them -- four requests per candidate and lane at W = 5 instead of ten.
    // A balanced deep-bin scan has 64 lanes on 64 different reads and the L1 looks every one of them up per
    // instruction: +3-6 % on deep pools; at the headline size, where one or two lanes compare, it costs 1 % (same box:
    // 407 / 410 ms), so the other variants keep the dwords.
    const int nd = 2 * min(W - i0, STAGE_LIMBS), nq = QUAD ? nd >> 2 : 0;
#define STAGE_ROW(D) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(stage + (D) * 63), 4, (D) * 4, 0)
#define STAGE_QUAD(Q) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(stage + (Q) * 252), 16, (Q) * 16, 0)
#define STAGE_WORD(D, ROW) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(stage + STAGE_ODD + (ROW) * 64 - (D)), 4, (D) * 4, 0)
    static_assert(STAGE_LIMBS == 5, "ten dword rows / two quad rows and the odd limb below");
    if (!QUAD) {
      STAGE_ROW(0); STAGE_ROW(1);
      if (nd > 2) { STAGE_ROW(2); STAGE_ROW(3); }
      if (nd > 4) { STAGE_ROW(4); STAGE_ROW(5); }
      if (nd > 6) { STAGE_ROW(6); STAGE_ROW(7); }
      if (nd > 8) { STAGE_ROW(8); STAGE_ROW(9); }
    } else if (nd == 10) { STAGE_QUAD(0); STAGE_QUAD(1); STAGE_WORD(8, 0); STAGE_WORD(9, 1); }
    else if (nd == 8) { STAGE_QUAD(0); STAGE_QUAD(1); }
    else if (nd == 6) { STAGE_QUAD(0); STAGE_WORD(4, 0); STAGE_WORD(5, 1); }
    else if (nd == 4) { STAGE_QUAD(0); }
    else { STAGE_WORD(0, 0); STAGE_WORD(1, 1); }
#undef STAGE_ROW
#undef STAGE_QUAD
#undef STAGE_WORD
    if (spec_taken && i0 == 0) {
      const uint64_t tw = spec_taken[r >> 6];  // (issued behind the staging loads, waited for with them)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PT(13);
      if ((tw >> (r & 63)) & 1ull) return -2;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PT(13);
    }
    const int ihi = min(i0 + STAGE_LIMBS - 1, last);
    // unrolled (all LDS reads of a chunk issued first) the round kernels spill: 420 -> 569 ms at the headline size;
    // k_long has the registers (UNROLL: every limb of the chunk, the ones outside [first, last] masked to nothing)
#pragma unroll UNROLL ? STAGE_LIMBS : 1
    for (int i = UNROLL ? i0 : max(i0, first); i <= (UNROLL ? min(i0 + STAGE_LIMBS, W) - 1 : ihi); i++) {
      if (UNROLL && (i < first || i > last)) continue;
      const int u = i - i0;
      const uint64_t xr = !QUAD      ? (uint64_t)stage[(2 * u) * 64 + lane] | ((uint64_t)stage[(2 * u + 1) * 64 + lane] << 32)
                          : u < 2 * nq ? *(const lds_u64_t *)(stage + (u >> 1) * 256 + lane * 4 + (u & 1) * 2)
                                       : (uint64_t)stage[STAGE_ODD + lane] | ((uint64_t)stage[STAGE_ODD + 64 + lane] << 32);
      uint64_t y = lds_window(sx, i * 64 + bitshift) ^ xr;
      if (i == first) y &= ~0ull << (blo & 63);
      if (i == last) y &= ~0ull >> (63 - ((bhi - 1) & 63));
      hd += __popcll(y);
      if (check_key && i >= (ka >> 6) && i <= ((kb - 1) >> 6)) {
        if (i == (ka >> 6)) y &= ~0ull << (ka & 63);
        if (i == ((kb - 1) >> 6)) y &= ~0ull >> (63 - ((kb - 1) & 63));
        kdiff |= (uint32_t)y | (uint32_t)(y >> 32);
      }
    }
  }
  PT(14);
  if (kdiff) return -1;
  return hd <= THRESH;
}

// ---- one probe of search_match (reorder.h:262-316): dictionary l, direction rev, at `shift`.  The window's
// key and hash come from the caller because one consensus window is the probe key of both dictionaries
// (at shifts wl apart).  sx = ref (forward) or revref (reverse) in LDS.
// DEFER (probe_batch's balanced scan): a multi-read bin whose key is verified is not walked here -- its extent comes back
// in `pend` (start, count, record index) and the caller deals its entries out over the lanes.
struct PendBin { bool on; uint32_t start, count, pay; };
// SPEC (k_round_mc): candidate reads are fetched speculatively beside their taken bit (cmp_candidate).
// pre != 0: the caller has seen the key's first slot in the tags of its HOME bucket already (tab_find's result for skip = 0):
// pre = kind (1 / 2) | slot << 2 -- the walk is skipped for it, the payload word comes from the line the tags came from.
template <bool TRIM, bool DEFER = false, bool SPEC = false, bool MZ = false>
__device__ __forceinline__ void eval_probe(const DevParams &P, const uint64_t *sx, int l, int rev, int shift,
                                           int ref_len, uint64_t key, uint64_t hsh, uint32_t mz, bool &hit, uint32_t &rid,
                                           bool &keyok, uint32_t &ncand, bool &other, lds_u32_t *s_best, lds_u32_t *stage,
                                           int lane, PendBin *pend = nullptr, int *walk_left = nullptr, int pre = 0) {
  // (l differs between lanes: P.x[l] would be a vector load from the kernel-argument buffer -- select instead)
  const int ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
  const int klen2 = 2 * P.wl;
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) u64x2_t g_urec_t;  // (an integer-made pointer is generic: say "global")
  typedef const __attribute__((address_space(1))) uint32_t g_u32_t;
  g_urec_t *urec = (g_urec_t *)(l ? uni_ptr(P.urec[1]) : uni_ptr(P.urec[0]));
  g_u32_t *ids = (g_u32_t *)(l ? uni_ptr(P.ids[1]) : uni_ptr(P.ids[0]));
  const int bitshift = rev ? -2 * shift : 2 * shift;
  const int lo = rev ? shift : 0;
  const int mref = rev ? ref_len + shift : ref_len - shift;
  auto within_thresh = [&](uint32_t r, bool check_key) -> int {
    return cmp_candidate<TRIM>(P, sx, bitshift, lo, mref, ds, klen2, r, check_key, stage, lane, SPEC ? P.taken : nullptr);
  };
  // *s_best (LDS) = lowest priority code that has hit so far in this batch of probes: the lanes run in lock step, so
  // a lane still walking a bin after another lane with a lower code has hit can never be the winner and leaves
  // (the wavefront waits for its slowest lane; each further candidate is three dependent memory round trips)
  const uint32_t mycode = (uint32_t)((shift << 2) | (rev << 1) | l);
  auto beaten = [&]() -> bool { return *(volatile lds_u32_t *)s_best < mycode; };
  for (int skip = 0;; skip++) {
    if (skip && beaten()) break;
    uint32_t pay;
    int kind;
    if (MZ && skip == 0 && pre) {
      kind = pre & 3;
      pay = reinterpret_cast<const uint32_t *>(P.tab.buck)[tab_home(P.tab, hsh, mz) * 8 + 4 + (pre >> 2)];
    } else kind = tab_find<MZ>(P.tab, hsh, mz, l, skip, pay, other);
    if (skip == 0) PTW(12);
    if (kind == 0) break;  // key absent
    // a single-read bin (kind 2: pay is the read id) runs through the same scan as a bin of one entry; its key is
    // verified on the read itself, and only once the read is known to be untaken: a taken read contributes
    // nothing whether this slot is the key's bin or a fingerprint collision, and the bitmap (12.5 MB, cache-
    // resident) is tested before a random 64-byte read is spent on it
    const bool single = kind == 2;
    uint32_t start = 0, count = 1;
    if (!single) {
      const u64x2_t rec = urec[pay];
      if (rec.x != key) continue;  // fingerprint collision
      start = (uint32_t)rec.y; count = (uint32_t)(rec.y >> 32);
      if (DEFER) { pend->on = true; pend->start = start; pend->count = count; pend->pay = pay; break; }
    }
    bool verified = !single, gave_up = false;
    int live = 0, top_live = -1;
    for (int j = (int)count - 1; j >= 0 && live < MAX_SEARCH; j--) {  // bin tail first, <=1000 live
      if (j != (int)count - 1 && beaten()) { gave_up = true; break; }
      // (long searches, see k_long: a lane that has walked its share of bin entries calls the search off for the whole
      // wavefront -- code 0 beats every probe of the tail, the only caller that passes a budget)
      if (walk_left && --(*walk_left) < 0) {
        // ... if this bin alone still holds a long walk (P.long_min / 64 entries); else the lane finishes the bin
        if (j >= P.long_min >> 6) {
          __hip_atomic_fetch_min(s_best, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          gave_up = true;
          break;
        }
        *walk_left = j;
      }
      uint32_t r = single ? pay : ids[start + j];
      if (TRIM && !single) {  // (with DevParams::epos the entry itself says whether its read is taken)
        bool trimmable;
        const bool dead = entry_dead(P, r, trimmable);
        if (top_live < 0 && !trimmable) top_live = j;  // (the dead tail above it is dead for every launch)
        if (dead) continue;
      } else if (!SPEC && is_taken(P.taken, r)) continue;
      const int wt = within_thresh(r, single);
      if (SPEC && wt == -2) continue;  // taken
      if (wt < 0) break;  // fingerprint collision (single-read bin)
      verified = true;
      live++; keyok = true; ncand++;
      if (wt) { hit = true; rid = r; __hip_atomic_fetch_min(s_best, mycode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); break; }
    }
    if (gave_up) break;  // (no trim below: the scan did not reach the bin's live tail)
    if (!verified) continue;  // taken or colliding single-read slot: the key's own bin may sit in a later slot
    // Chains consume a bin from its tail, and a taken read stays taken: the entries above the first live one are
    // dead for good, so the bin's count shrinks to it (exact; the reference gets the same from
    // bbhashdict::remove, bitset_util.cpp:37-63).  Later scans of a deep bin then start at its live tail
    // instead of walking the dead one entry by entry.  TRIM variants of the kernels only: they are launched when the
    // dictionary has deep bins (pools of a few hundred x coverage and more: +23 % at 1600x); on shallow data the extra
    // state in the hot loop costs 5 %.
    if (TRIM && !single && (uint32_t)(top_live + 1) < count)
      atomicMin((uint32_t *)(uint64_t)(&urec[pay]) + 3, (uint32_t)(top_live + 1));
    break;
  }
}

// ---- the alternatives schedule (DevParams::alts = 2; executable specification: orc_reorder_rounds_alt): the SECOND candidate of a search
// whose winner is read `wrid` of the probe `code` -- the next entry of the winner's bin, in the order the reference scans
// it (from the tail), that is free and within the Hamming threshold, inside the same MAX_SEARCH_REORDER window (the reads
// the reference's thread would try next after losing the read_lock race, reorder.h:303-311).  0xffffffff: none (a
// single-read bin, or nothing passes).  One routine for every way a winner is found (serial walk, balanced scan, k_long):
// it runs after the search, by the whole wavefront, on wave-uniform arguments: the bin is looked up again (its lines were
// just touched), the entries ahead of the winner are only counted, the entries behind it are compared 64 at a time --
// the first four live ones on their own first (on deep-coverage pools the next live entry nearly always passes).
template <bool QUAD>
__device__ __forceinline__ uint32_t find_alt(const DevParams &P, const uint64_t *sref, const uint64_t *srev, int ref_len,
                                             int code, uint32_t wrid, lds_u32_t *stage, int lane) {
  const int l = code & 1, rev = (code >> 1) & 1, shift = code >> 2;
  const int ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
  const int klen2 = 2 * P.wl;
  const uint64_t kmask = klen2 < 64 ? ((1ull << klen2) - 1) : ~0ull;
  const uint64_t *sx = rev ? srev : sref;
  const uint64_t key = lds_window(sx, rev ? 2 * (ds - shift) : 2 * (ds + shift)) & kmask;
  const uint64_t hsh = mix64(key);
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) u64x2_t g_urec_t;
  typedef const __attribute__((address_space(1))) uint32_t g_u32_t;
  g_urec_t *urec = (g_urec_t *)(l ? uni_ptr(P.urec[1]) : uni_ptr(P.urec[0]));
  g_u32_t *ids = (g_u32_t *)(l ? uni_ptr(P.ids[1]) : uni_ptr(P.ids[0]));
  uint32_t start = 0, count = 0;
  for (int skip = 0;; skip++) {  // the key's bin (every lane walks the same slots: one request each)
    uint32_t pay;
    bool other = false;
    const int kind = uni_i32(tab_find<false>(P.tab, hsh, 0u, l, skip, pay, other));
    if (kind == 0) return 0xffffffffu;
    if (kind == 2) { if (uni_u32(pay) == wrid) return 0xffffffffu; continue; }  // the winner's own single-read bin / a colliding slot
    const u64x2_t rec = urec[uni_u32(pay)];
    if (rec.x != key) continue;  // fingerprint collision
    start = uni_u32((uint32_t)rec.y); count = uni_u32((uint32_t)(rec.y >> 32));
    break;
  }
  const int bitshift = rev ? -2 * shift : 2 * shift;
  const int lo = rev ? shift : 0;
  const int mref = rev ? ref_len + shift : ref_len - shift;
  int livec = 0;      // live entries of the chunks before this one
  bool seen = false;  // the winner's entry has been passed
  for (uint32_t base = 0; base < count; base += 64) {
    const long long j = (long long)count - 1 - ((long long)base + lane);
    uint32_t r = 0;
    bool live = false;
    if (j >= 0) {
      r = ids[start + (uint32_t)j];
      bool tr_;
      live = !entry_dead(P, r, tr_);
    }
    const uint64_t Lm = __ballot(live);
    uint64_t cand = Lm;
    if (!seen) {
      const uint64_t Wm = __ballot(j >= 0 && r == wrid);
      if (Wm) { seen = true; cand = Lm & ~((2ull << (__ffsll((unsigned long long)Wm) - 1)) - 1ull); }
      else cand = 0;
    }
    // an entry is looked at while fewer than MAX_SEARCH live entries (the winner among them) lie ahead of it
    const int before = livec + __popcll(Lm & ((1ull << lane) - 1ull));
    const uint64_t Em = __ballot(((cand >> lane) & 1ull) && before < MAX_SEARCH);
    if (Em) {
      uint64_t first = Em;  // its four lowest bits
      { uint64_t t = Em; t &= t - 1; t &= t - 1; t &= t - 1; t &= t - 1; first = Em & ~t; }
      for (int stg = 0; stg < 2; stg++) {
        const uint64_t m = stg ? (Em & ~first) : first;
        if (!m) break;
        bool ps = false;
        if ((m >> lane) & 1ull) ps = cmp_candidate<QUAD>(P, sx, bitshift, lo, mref, ds, klen2, r, false, stage, lane) == 1;
        const uint64_t Pm = __ballot(ps);
        if (Pm) return (uint32_t)__shfl((int)r, __ffsll((unsigned long long)Pm) - 1, 64);
      }
    }
    livec += __popcll(Lm);
    if (livec >= MAX_SEARCH) break;
  }
  return 0xffffffffu;
}

// priority code of a probe: the reference tries shift ascending, forward before reverse, dictionary 0
// before 1 (reorder.h:479-558, :262-316); the lowest code that hits is the reference's winner
__device__ __forceinline__ int probe_code(int shift, int rev, int l) { return (shift << 2) | (rev << 1) | l; }
__device__ __forceinline__ bool probe_valid(const DevParams &P, int l, int rev, int shift, int ref_len) {
  if (shift >= P.maxshift) return false;
  const int de = l ? uni_i32(P.dend[1]) : uni_i32(P.dend[0]), ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
  return rev ? (de < ref_len + shift && ds > shift) : (de + shift < ref_len);
}

struct BatchOut {
  bool capped;       // a probe ahead of the winner (any probe, when nothing hit) stopped at MAX_SEARCH live candidates
  uint32_t found, rid;
  int code;          // probe_code of the winner
  uint64_t st_p, st_k, st_c;
};

// ---- shifts [sh_base, sh_base + nsh), nsh <= 16, in lane order: lane = 4*(shift - sh_base) + 2*rev + dict,
// so lane order == priority order and the first set bit of the hit ballot is the reference's winner.
// STATS counts what the reference would have executed: every valid probe up to and including the winner.
template <bool STATS, bool TRIM>
__device__ __forceinline__ void probe_batch(const DevParams &P, const uint64_t *sref, const uint64_t *srev,
                                            int sh_base, int nsh, int lane, int ref_len, uint8_t *pres, lds_u32_t *s_best,
                                            lds_u32_t *stage, int min_code, uint8_t *owner_of /* [64], LDS */, BatchOut &out,
                                            int *budget = nullptr /* wave-uniform: passes of the balanced scan left */) {
  const int l = lane & 1, rev = (lane >> 1) & 1;
  const int klen2 = 2 * P.wl;
  const uint64_t kmask = klen2 < 64 ? ((1ull << klen2) - 1) : ~0ull;
  const uint64_t *sx = rev ? srev : sref;
  const int shift = sh_base + (lane >> 2);
  // min_code (a chain that lost its last proposal searches again on an unchanged consensus): every probe ahead of
  // the last winner failed then and fails now -- taken reads stay taken -- so the search resumes at the winner's
  // code; a skipped probe leaves "the other dictionary may hold this window" behind for the tail
  const bool valid0 = (lane >> 2) < nsh && probe_valid(P, l, rev, shift, ref_len);
  const bool skipped = TRIM && valid0 && probe_code(shift, rev, l) < min_code;  // (TRIM: see search_step)
  const bool valid = valid0 && !skipped;
  bool hit = false, keyok = false, other = skipped;
  uint32_t rid = 0, ncand = 0;
  constexpr bool BAL = TRIM && !STATS;  // the balanced scan: deep-bin pools, production build
  PendBin pend;
  pend.on = false; pend.start = pend.count = pend.pay = 0;
  if (lane == 0) *s_best = 0x7fffffffu;
  wave_sync();
  if (valid) {
    const int ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
    const uint64_t key = lds_window(sx, rev ? 2 * (ds - shift) : 2 * (ds + shift)) & kmask;
    eval_probe<TRIM, BAL>(P, sx, l, rev, shift, ref_len, key, mix64(key), 0u /* (hash-addressed tables only: the pipeline refuses table_mode = 2 for the one-chain kernels) */, hit, rid, keyok, ncand, other, s_best, stage, lane, &pend);
  }
  bool bal_capped = false;
  if (BAL) {
    // ---- balanced scan of the verified multi-read bins (deep-bin pools: every bin keeps reads that can never pass
    // the Hamming test, and a lane walking them one after the other -- three dependent memory steps each -- is what a
    // round waits for).  The bins' entries are dealt out in the reference's order -- probes by priority (= lane
    // order), entries from the bin's tail -- 64 at a time, one per lane; the first passing entry in that order whose
    // probe has not used up its MAX_SEARCH live comparisons is the winner among these bins.
    const uint64_t hs = __ballot(hit);  // single-read bins that hit: only bins ahead of the first of them matter
    const int wins = hs ? __ffsll((unsigned long long)hs) - 1 : 64;
    uint64_t pm = __ballot(pend.on && lane < wins);
    int jn = (int)pend.count - 1, livec = 0, top_live = -2;  // owner state: next entry, live so far, first live entry (-2: unknown)
    const int klen2 = 2 * P.wl;
    typedef const __attribute__((address_space(1))) uint32_t g_u32_t;
    while (pm) {
      if (budget && --(*budget) < 0) {  // a long search: k_long finishes it (the caller tests the budget) ...
        // ... unless little is left of this batch's bins: a hand-over adds k_long's own latency (30 us and more) to the
        // round, which only pays for a search that would outlast the round's other chains by a multiple of that
        const int left = wave_sum_i(((pm >> lane) & 1ull) ? min(jn + 1, 1 << 20) : 0);
        if (left >= P.long_min) break;
        *budget = 16;
      }
      const bool own = (pm >> lane) & 1ull;
      const int rem = own ? jn + 1 : 0;
      const int incl = wave_incl_scan_i(rem > 64 ? 64 : rem, lane);  // (64 is all a chunk can take from one bin)
      const int off = incl - (rem > 64 ? 64 : rem);
      const int take = own && off < 64 ? min(rem, 64 - off) : 0;
      const int total = min(__shfl(incl, 63, 64), 64);
      // owner of item t: the owning lane whose run [off, off + take) covers t
      owner_of[lane] = 0xff;
      wave_sync();
      if (take > 0) owner_of[off] = (uint8_t)lane;
      wave_sync();
      int ow = owner_of[lane] == 0xff ? -1 : (int)owner_of[lane];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(ow, o, 64); if (lane >= o) ow = max(ow, t); }
      wave_sync();
      bool lv = false, ps = false, nt = false;  // live, passes, not trimmable (live, or dead in this group's view only)
      uint32_t r = 0;
      // (every lane takes part in the shuffles: a lane outside a branch supplies nothing)
      const int src = ow >= 0 ? ow : 0;
      const int pj = __shfl(jn, src, 64), poff = __shfl(off, src, 64);
      const uint32_t pst = (uint32_t)__shfl((int)pend.start, src, 64);
      if (lane < total && ow >= 0) {
        const int j = pj - (lane - poff);
        const int pl = ow & 1, prev = (ow >> 1) & 1, psh = sh_base + (ow >> 2);
        g_u32_t *pids = (g_u32_t *)(pl ? uni_ptr(P.ids[1]) : uni_ptr(P.ids[0]));
        r = pids[pst + (uint32_t)j];
        bool trimmable;
        const bool dead = entry_dead(P, r, trimmable);
        nt = !trimmable;
        if (!dead) {
          lv = true;
          const int pds = pl ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
          ps = cmp_candidate<true>(P, prev ? srev : sref, prev ? -2 * psh : 2 * psh, prev ? psh : 0,
                             prev ? ref_len + psh : ref_len - psh, pds, klen2, r, false, stage, lane) == 1;
        }
      }
      const uint64_t Lm = __ballot(lv), Pm = __ballot(ps), Tm = __ballot(nt);
      bool myhit = false, fin = own && rem == 0;  // (a bin already trimmed to nothing)
      int hit_t = 0;
      if (take > 0) {
        const uint64_t mine = (take >= 64 ? ~0ull : ((1ull << take) - 1)) << off;
        const uint64_t ml = Lm & mine, mp = Pm & mine, mt = Tm & mine;
        if (top_live == -2 && mt) top_live = jn - (__ffsll((unsigned long long)mt) - 1 - off);
        if (mp) {
          hit_t = __ffsll((unsigned long long)mp) - 1;
          const int before = livec + __popcll(ml & ((1ull << hit_t) - 1));
          if (before < MAX_SEARCH) myhit = true; else { bal_capped = true; fin = true; }
        }
        livec += __popcll(ml);
        jn -= take;
        if (!myhit) {
          if (livec >= MAX_SEARCH) { bal_capped = true; fin = true; }
          if (jn < 0) { fin = true; if (top_live == -2) top_live = -1; }
        }
      }
      const uint64_t hb = __ballot(myhit);
      if (hb) {  // the lowest owner that hit wins (only the last owner of a chunk can be unfinished)
        const int wl_ = __ffsll((unsigned long long)hb) - 1;
        const int ht = __shfl(hit_t, wl_, 64);
        const uint32_t wr = (uint32_t)__shfl((int)r, ht, 64);
        if (lane == wl_) { hit = true; rid = wr; }
        break;
      }
      pm &= ~__ballot(fin);
    }
    // dead tail of the bins this scan reached the live part of (same rule as the serial walk)
    if (pend.on && top_live >= -1 && (uint32_t)(top_live + 1) < pend.count) {
      const uint64_t ur = reinterpret_cast<uint64_t>(l ? P.urec[1] : P.urec[0]);
      atomicMin(reinterpret_cast<uint32_t *>(ur + (uint64_t)pend.pay * 16) + 3, (uint32_t)(top_live + 1));
    }
  }
  const uint64_t hm = __ballot(hit);
  const int win = hm ? __ffsll((unsigned long long)hm) - 1 : 63;
  out.found = hm != 0;
  out.code = probe_code(sh_base + (win >> 2), (win >> 1) & 1, win & 1);
  out.rid = (uint32_t)__shfl((int)rid, win, 64);
  out.capped = TRIM && (__any(valid && (hm == 0 || lane < win) && !hit && ncand >= (uint32_t)MAX_SEARCH) || __any(bal_capped));
  // what this fetch told about the OTHER dictionary's probe of the same window, for the tail (probe_tail)
  if ((lane >> 2) < nsh && shift < 32) pres[4 * shift + (lane & 3)] = other ? 1 : 0;
  out.st_p = out.st_k = out.st_c = 0;
  if (STATS) {
    const uint64_t le = win == 63 ? ~0ull : ((1ull << (win + 1)) - 1);
    out.st_p = __popcll(__ballot(valid) & le);
    out.st_k = __popcll(__ballot(keyok) & le);
    out.st_c = (uint64_t)wave_sum_i(lane <= win ? (int)ncand : 0);
  }
}

// ---- every remaining probe (shifts [t0, maxshift)) at once, one bucket fetch per distinct consensus window.
// A forward window at consensus offset off is the key of dictionary 0 at shift off - start0 and of dictionary
// 1 at shift off - start1 = (off - start0) - wl; a reverse window of dictionary 0 at start0 - off and of
// dictionary 1 at start1 - off.  Windows already fetched by the ordered batches (as the dict-1 probe of a
// forward shift < t0, or the dict-0 probe of a reverse shift < t0) are skipped unless that fetch saw a slot
// of the other dictionary (`pres`, one byte per ordered probe).  The needed windows are compacted into `list` (LDS) so that a
// failing search of a 150-base consensus costs 32 + 64 + 55 fetches in three dependent steps (was 237 in
// six).  Lanes no longer run in priority order: the winner is the hit with the lowest probe_code.
constexpr int TAIL_CAP = 576;  // 2 * (32 + MAX_READ_LEN / 2) windows at most
template <bool STATS, bool TRIM>
__device__ __forceinline__ void probe_tail(const DevParams &P, const uint64_t *sref, const uint64_t *srev,
                                           uint16_t *list, uint16_t *stat, int t0, const uint8_t *pres, int lane,
                                           int ref_len, lds_u32_t *s_best, lds_u32_t *stage, int min_code, BatchOut &out,
                                           int *walk_left = nullptr /* per lane: bin entries it may still walk (long searches) */) {
  const int wl = P.wl, s0 = P.dstart[0], s1 = P.dstart[1], ms = P.maxshift;
  const uint64_t kmask = 2 * wl < 64 ? ((1ull << (2 * wl)) - 1) : ~0ull;
  // did the ordered batches' fetch for shift sp (slot x: 1 = forward dict 1, 2 = reverse dict 0) leave the other
  // dictionary's presence open?  (t0 <= 32: probe_batch recorded every shift below t0)
  auto present = [&](int sp, int x) -> bool { return pres[4 * sp + x] != 0; };
  const int nF = wl + ms - t0;
  int T = 0;
  for (int base = 0; base < 2 * nF; base += 64) {
    const int idx = base + lane;
    bool v0 = false, v1 = false;
    int rev = 0, i = 0;
    if (idx < 2 * nF) {
      rev = idx >= nF;
      i = rev ? idx - nF : idx;
      if (!rev) {
        const int sh0 = t0 + i, sh1 = sh0 - wl;
        v0 = probe_valid(P, 0, 0, sh0, ref_len);
        v1 = sh1 >= t0 && probe_valid(P, 1, 0, sh1, ref_len);
        if (v0 && sh1 >= 0 && sh1 < t0) v0 = present(sh1, 1);
      } else {
        const int sh1 = t0 + i, sh0 = sh1 - wl;
        v1 = probe_valid(P, 1, 1, sh1, ref_len);
        v0 = sh0 >= t0 && probe_valid(P, 0, 1, sh0, ref_len);
        if (v1 && sh0 >= 0 && sh0 < t0) v1 = present(sh0, 2);
      }
      // (resumed search: nothing ahead of the last winner)
      const int sh0c = rev ? t0 + i - wl : t0 + i, sh1c = rev ? t0 + i : t0 + i - wl;
      if (TRIM) {
        v0 = v0 && probe_code(sh0c, rev, 0) >= min_code;
        v1 = v1 && probe_code(sh1c, rev, 1) >= min_code;
      }
    }
    const bool need = v0 || v1;
    const uint64_t m = __ballot(need);
    if (need) {
      const int at = T + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      list[at] = (uint16_t)((rev << 15) | ((int)v1 << 14) | ((int)v0 << 13) | i);
      if (STATS) { stat[2 * at] = 0; stat[2 * at + 1] = 0; }
    }
    T += __popcll(m);
  }
  if (lane == 0) *s_best = 0x7fffffffu;
  wave_sync();
  int best = 0x7fffffff, capmin = 0x7fffffff;  // capmin: lowest code of this lane's probes that stopped at MAX_SEARCH
  uint32_t brid = 0;
  for (int base = 0; base < T; base += 64) {
    const int j = base + lane;
    if (j < T) {
      const uint32_t e = list[j];
      const int rev = (int)(e >> 15), i = (int)(e & 0x1ffu);
      int sh0, sh1, off;
      if (!rev) { sh0 = t0 + i; sh1 = sh0 - wl; off = s0 + sh0; }
      else { sh1 = t0 + i; sh0 = sh1 - wl; off = s1 - sh1; }
      const uint64_t *sx = rev ? srev : sref;
      const uint64_t key = lds_window(sx, 2 * off) & kmask;
      const uint64_t hsh = mix64(key);
      const uint32_t mzv = 0u;  // (hash-addressed tables only)
      // the probe with the lower priority code first: forward dictionary 1 (its shift is wl lower), reverse dictionary 0
#pragma nounroll
      for (int k = 0; k < 2; k++) {
        const int l = rev ? k : 1 - k;
        if (!((e >> (13 + l)) & 1u)) continue;
        const int sh = l ? sh1 : sh0;
        bool hit = false, keyok = false, other = false;
        uint32_t rid = 0, ncand = 0;
        if ((base || k) && *(volatile lds_u32_t *)s_best < (uint32_t)probe_code(sh, rev, l)) continue;  // cannot win any more
        eval_probe<TRIM>(P, sx, l, rev, sh, ref_len, key, hsh, mzv, hit, rid, keyok, ncand, other, s_best, stage, lane, nullptr, walk_left);
        if (STATS) stat[2 * j + l] = (uint16_t)(((int)keyok << 15) | (int)ncand);
        if (TRIM && !hit && ncand >= (uint32_t)MAX_SEARCH) capmin = min(capmin, probe_code(sh, rev, l));
        if (hit) {
          const int code = probe_code(sh, rev, l);
          if (code < best) { best = code; brid = rid; }
          break;
        }
      }
    }
  }
  int wmin = best;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmin = min(wmin, __shfl_xor(wmin, o, 64));
  out.found = wmin != 0x7fffffff;
  out.code = wmin;
  out.capped = TRIM && __any(capmin < wmin);
  const uint64_t wm = __ballot(best == wmin);
  out.rid = (uint32_t)__shfl((int)brid, __ffsll((unsigned long long)wm) - 1, 64);
  out.st_p = out.st_k = out.st_c = 0;
  if (STATS) {  // what the reference would have executed: every valid probe of the tail up to the winner
    int cp = 0, ck = 0, cc = 0;
    for (int idx = lane; idx < (ms - t0) * 4; idx += 64) {
      const int sh = t0 + (idx >> 2), rev = (idx >> 1) & 1, l = idx & 1;
      if (probe_valid(P, l, rev, sh, ref_len) && probe_code(sh, rev, l) <= wmin) cp++;
    }
    wave_sync();
    for (int j = lane; j < T; j += 64) {
      const uint32_t e = list[j];
      const int rev = (int)(e >> 15), i = (int)(e & 0x1ffu);
      const int sh0 = rev ? t0 + i - wl : t0 + i, sh1 = rev ? t0 + i : t0 + i - wl;
      for (int l = 0; l < 2; l++) {
        if (!((e >> (13 + l)) & 1u)) continue;
        if (probe_code(l ? sh1 : sh0, rev, l) > wmin) continue;
        const uint32_t sv = stat[2 * j + l];
        ck += (int)(sv >> 15); cc += (int)(sv & 0x7fffu);
      }
    }
    out.st_p = (uint64_t)wave_sum_i(cp); out.st_k = (uint64_t)wave_sum_i(ck); out.st_c = (uint64_t)wave_sum_i(cc);
  }
}

// ------------------------------------------------------------ K4 search (phase A)
//
// One wavefront per chain.  search_match + shift loop (reorder.h:246-318, :479-558): two ordered batches
// (shifts [0, fs) and [fs, fs + 16): lane order == the reference's priority order, the first set bit of the
// hit ballot is the reference's winner), then the tail (every remaining window at once, winner = lowest
// priority code).  Chains that need a new contig seed instead pick the (rank+1)-th highest untaken read at
// or below the global cursor (reorder.h:576-592).
// WORD: publish the proposal as a word of P.prop (multi-GPU pools: resolved after the all-gather; fused rounds:
// read by k_mg_mark).  DIRECT: reserve the read at once (atomicMin on resv[]), everything is on this GPU.
// `h` is the chain's header as it stands (all lanes hold the same copy); ref / revref are already in s_refs.
template <bool STATS, bool WORD, bool DIRECT, bool TRIM, bool LONG = false>
__device__ __forceinline__ int search_step(const DevParams &P, Chain *c, uint32_t cid, ChainHot &h, int lane,
                                            uint64_t *s_refs /* [2][LDS_LIMBS] */, uint16_t *s_list, uint16_t *s_stat,
                                            uint8_t *s_pres /* [128] */, lds_u32_t *s_best, lds_u32_t *s_stage /* [STAGE_WORDS] */) {
  if (h.mode == MODE_NEED_SEED) {
    bool is_last;
    const long long seed_v = find_seed(P, cid, lane, &is_last);
    const bool have = __builtin_amdgcn_readfirstlane((int)(seed_v >= 0)) != 0;
    const uint32_t seed = uni_u32((uint32_t)seed_v);
    const bool last = __builtin_amdgcn_readfirstlane((int)is_last) != 0;
    if (have) {
      h.prop_kind = PROP_SEED;
      h.prop_rid = seed;
      // the last-ranked needy chain proposes the lowest seed of the round: it alone moves the cursor
      h.cursor_writer = last;
      if (lane == 0) {
        if (WORD) P.prop[cid] = ((unsigned long long)PK_SEED << 32) | seed | (last ? PK_CURSOR_BIT : 0ull);
        if (DIRECT) atomicMin(&P.resv[seed], cid);
      }
    } else {
      h.prop_kind = PROP_NONE;
      h.finishing = 1;  // no reads left (reorder.h:593-599); applied in phase B
      // the last-ranked needy chain finds nothing: every untaken read went to the chains ranked before it, so after
      // this round the whole pool is taken -- the cursor goes to -1 (find_seed relies on "every read above the
      // cursor is taken" AND on the cursor having passed every seed ever handed out)
      h.cursor_writer = last;
      if (WORD && lane == 0) P.prop[cid] = ((unsigned long long)PK_NOSEED << 32) | (last ? PK_CURSOR_BIT : 0ull);
    }
    store_hot(c, h, lane, 2, 4);
    PTW(11);
    return -2;
  }

  // ---- search mode
  if (!h.retrying) {  // iteration start bookkeeping (reorder.h:433-439), once per iteration
    if (h.num_reads_thr % 1000000u == 0) {
      if ((float)h.num_unmatched_past > 0.5f * 1000000) h.stop_searching = 1;
      h.num_unmatched_past = 0;
    }
    h.num_reads_thr++;
  }
  const bool new_iter = !h.retrying;
  if (h.stop_searching) {
    h.prop_kind = PROP_NONE;
    store_hot(c, h, lane, 2, 4);
    if (lane == 0) {
      if (WORD) P.prop[cid] = ((unsigned long long)PK_NONE << 32) | (h.left_search ? PK_WILLNEED_BIT : 0ull);
      if (STATS && new_iter) c->st_iter++;
    }
    return -3;
  }

  // (the header lives in scalar registers; its last two quarters go back once, at the end of the search)
  const int ref_len = h.ref_len;
  const bool left_search = h.left_search;
  const uint64_t *sref = s_refs + LDS_PAD, *srev = s_refs + LDS_LIMBS + LDS_PAD;
  wave_sync();
  // The ordered batches of a search, narrow first: most chains match within the first few shifts and every lane
  // past the winner is a wasted 64-byte request; every further batch is a further dependent round trip (8 + 16 is
  // the measured optimum).  A fresh seed (nothing matched to it yet) fails nine searches out of
  // ten and needs every window anyway: it gets the wide plan.  P.plan[which] = batch widths in shifts (each <= 16, sum <= 32), 0-terminated.
  const int *plan = P.plan[(h.prev_unmatched && P.seed_wide) ? 1 : 0];
  // a search repeated after a lost proposal resumes at the last winner's code (see probe_batch) -- unless some probe
  // ahead of that winner had stopped at MAX_SEARCH live candidates (then a deeper candidate may have come into its
  // reach: bit 2 of prop_rev says it had not), or the reference-equivalent work is being counted.  Only in the
  // kernel variant for dictionaries with deep bins (TRIM): that is where proposals are lost in numbers (a third of
  // them at 1 600x coverage, 0.3 % at 25x, where the bookkeeping costs more than it saves)
  int min_code = 0;
  if (TRIM && !STATS && !new_iter) {
    const int pr = (int)h.prop_rev;
    if (pr & 4) min_code = ((int)h.prop_shift << 2) | ((pr & 1) << 1) | ((pr >> 1) & 1);
  }
  bool capped = false;
  BatchOut o;
  o.found = 0;
  uint64_t st_p = 0, st_k = 0, st_c = 0;
  int t0 = 0;
  // Long searches (deep-bin pools, fused rounds): a failing search over full bins compares hundreds of candidates
  // one 64-lane pass after the other and a round waits for the few dozen wavefronts that do (45 % of a PhiX-like
  // run).  A wavefront that has spent P.long_budget passes of the balanced scan, or walked that many entries of one
  // lane's bins in the tail, stops, queues its chain and leaves: k_long, launched behind this kernel, runs the whole
  // search again with a block of 16 wavefronts.  The search is a pure function of (consensus, taken[]), so who
  // computes it does not show in the result.
  // (LONG: its own instantiation of the round kernel -- the budget bookkeeping costs the deep-bin variant 20 bytes of
  // scratch per lane, 2-4 % on the pools that never hand over)
  static_assert(!LONG || (TRIM && !STATS && WORD), "long searches: production deep-bin variant of the fused round");
  const bool lng = LONG && P.long_budget > 0;
  int budget = lng ? P.long_budget : 0x7fffffff;
  bool handed_over = false;
  PT(6);
#pragma nounroll
  for (int ph = 0; ph < 6 && plan[ph] > 0 && t0 < P.maxshift; ph++) {
    probe_batch<STATS, TRIM>(P, sref, srev, t0, plan[ph], lane, ref_len, s_pres, s_best, s_stage, min_code,
                             reinterpret_cast<uint8_t *>(s_list), o, LONG ? &budget : nullptr);
    if (LONG && __builtin_amdgcn_readfirstlane(budget) < 0) { handed_over = true; break; }
    capped = capped || o.capped;
    st_p += o.st_p; st_k += o.st_k; st_c += o.st_c;
    t0 += plan[ph];
    if (ph == 0) PT(7); else PT(8);
    if (o.found) break;
  }
  if (!handed_over && !o.found && t0 < P.maxshift) {
    wave_sync();  // s_pres
    // (what the ordered batches left of the budget; "no budget" is a huge one rather than a null pointer: a pointer that is
    // selected at run time keeps the variable in scratch memory)
    int walk_left = (lng && t0 > 0) ? budget : 0x7fffffff;
    probe_tail<STATS, TRIM>(P, sref, srev, s_list, s_stat, t0, s_pres, lane, ref_len, s_best, s_stage, min_code, o,
                            LONG ? &walk_left : nullptr);
    if (lng && t0 > 0) {
      wave_sync();
      handed_over = __builtin_amdgcn_readfirstlane((int)*(volatile lds_u32_t *)s_best) == 0;
    }
    capped = capped || o.capped;
    st_p += o.st_p; st_k += o.st_k; st_c += o.st_c;
    PT(9);
  }
  if (LONG && handed_over) {
    // the iteration bookkeeping above stays; the proposal fields still hold the last proposal (k_long derives the
    // resume point from them exactly as this search did) and k_long writes the new one
    store_hot(c, h, lane, 2, 4);
    if (lane == 0) P.longq[2 + atomicAdd(&P.longq[0], 1u)] = cid - P.c0;
    return -4;
  }
  {
    const bool found = __builtin_amdgcn_readfirstlane((int)o.found) != 0;
    const uint32_t wrid = uni_u32(o.rid);
    const int wcode = uni_i32(o.code);
    const bool cap_u = __builtin_amdgcn_readfirstlane((int)capped) != 0;
    if (found) {
      h.prop_rid = wrid;
      h.prop_shift = (uint32_t)(wcode >> 2);
      h.prop_rev = (uint32_t)(((wcode >> 1) & 1) | ((wcode & 1) << 1) | ((TRIM && !cap_u) ? 4 : 0));  // rev | dict << 1 | resumable << 2
      h.prop_kind = PROP_MATCH;
      // the alternatives schedule: the chain's second candidate travels with the proposal (deep-bin variants only)
      if (TRIM && WORD) h.alt1 = P.alts == 2 ? uni_u32(find_alt<TRIM>(P, sref, srev, ref_len, wcode, wrid, s_stage, lane)) + 1u : 0u;
    } else {
      h.prop_kind = PROP_NONE;
    }
    store_hot(c, h, lane, 2, 4);
  }
  if (lane == 0) {
    if (o.found) {
      if (WORD) P.prop[cid] = ((unsigned long long)PK_MATCH << 32) | o.rid | ((unsigned long long)(TRIM ? h.alt1 : 0u) << PK_ALT_SHIFT);
      if (DIRECT) atomicMin(&P.resv[o.rid], cid);
    } else {
      // a failed left search sends the chain for a new seed (apply step): k_mg_mark puts it on the needy bitmap
      if (WORD) P.prop[cid] = ((unsigned long long)PK_NONE << 32) | (left_search ? PK_WILLNEED_BIT : 0ull);
    }
    if (STATS) {
      c->st_probes += st_p; c->st_keyok += st_k; c->st_cands += st_c;
      if (o.found) c->st_hits++;
      if (new_iter) c->st_iter++;
    }
  }
  PT(10);
  return o.found ? o.code : -1;  // debug builds time the search by outcome; unused otherwise
}

// ------------------------------------------------------------ K4 search (phase A of the two-kernel round)
// WPB = chains (wavefronts) per block.
template <bool STATS, int WPB>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(STATS ? (WPB == 4 ? 5 : 6) : 7, STATS ? (WPB == 4 ? 5 : 6) : 7)))
void k_search(DevParams P) {  // (7: no scratch; the counting builds carry the work counters: 6 / 5)
  __shared__ uint64_t s_refs[WPB][2][LDS_LIMBS];
  __shared__ uint16_t s_list[WPB][TAIL_CAP];
  __shared__ uint16_t s_stat[STATS ? WPB : 1][STATS ? 2 * TAIL_CAP : 2];
  __shared__ uint8_t s_pres[WPB][128];
  __shared__ uint32_t s_best[WPB];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[WPB][STAGE_WORDS];
  const int wave = uni_i32((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: the chain pointer stays in SGPRs)
  const uint32_t li = blockIdx.x * WPB + wave;
  if (li >= P.K) return;
  const uint32_t cid = P.c0 + li;  // global chain id (conflict priority, seed rank)
  Chain *c = &P.chains[li];
  // ref / revref go to LDS; their loads are in flight while the header's scalar load is waited for
  uint64_t r0 = 0, r1 = 0;
  {
    const int i = lane - LDS_PAD;
    if (lane < LDS_LIMBS && i >= 0 && i < P.W) { r0 = c->ref[i]; r1 = c->revref[i]; }
  }
  ChainHot h;
  load_hot(c, h);
  if (h.done) return;
  if (lane < LDS_LIMBS) { s_refs[wave][0][lane] = r0; s_refs[wave][1][lane] = r1; }
  wave_sync();
  search_step<STATS, false, true, false>(P, c, cid, h, lane, &s_refs[wave][0][0], s_list[wave], s_stat[STATS ? wave : 0], s_pres[wave],
                                          (lds_u32_t *)&s_best[wave], (lds_u32_t *)s_stage[wave]);
}

// ------------------------------------------------------------- K5/K6 apply (phase B)
//
// Resolves the round's proposals (lowest chain id holds resv[rid]) and applies
// the winner's step: claim, consensus update, position bookkeeping and
// emission (reorder.h:484-515, :522-553), or the failure path (reorder.h:559-615).
// The consensus update is computed speculatively while the resv[] answer is in
// flight and only committed by the winner.
// Emission slots: each chain fills private CHUNK-slot chunks of the append buffers and only
// touches the global allocator once per CHUNK records (a same-address atomic per record from
// every chain serialises at ~11 ns each and dominated this kernel).
// (called by the whole wavefront: the slot bookkeeping stays wave-uniform, lane 0 does the memory operations)
__device__ __forceinline__ uint32_t take_slot(uint32_t &slot, uint32_t *alloc, uint2 *chunk, uint32_t li, uint32_t next_seq, int lane) {
  const uint32_t s0 = slot;
  uint32_t nx = s0 + 1;
  if ((nx & (CHUNK - 1)) == 0) {
    uint32_t got = 0;
    if (lane == 0) got = atomicAdd(alloc, CHUNK);
    nx = uni_u32(got);
    if (lane == 0) chunk[nx / CHUNK] = make_uint2(li, next_seq);  // the chunk's owner and the sequence number of its first record
  }
  slot = nx;
  return s0;
}
// one 16-byte store per matched record (length: looked up by the final scatter)
__device__ __forceinline__ void emit_rec(const DevParams &P, ChainHot &h, uint32_t li, uint32_t rid, char rc,
                                         char flag, long long pos, int lane) {
  const uint32_t seq = h.n_emit;
  h.n_emit = seq + 1;
  uint32_t slot = h.e_slot;
  const uint32_t idx = take_slot(slot, &P.glob->e_alloc, P.e_chunk, li, seq + 1, lane);
  h.e_slot = slot;
  if (lane == 0)
    P.e_rec[idx] = make_uint4(rid, (uint32_t)(uint8_t)rc | ((uint32_t)(uint8_t)flag << 8), (uint32_t)pos,
                              (uint32_t)((unsigned long long)pos >> 32));
}
__device__ __forceinline__ void emit_single(const DevParams &P, ChainHot &h, uint32_t li, uint32_t rid, int lane) {
  const uint32_t seq = h.n_single;
  h.n_single = seq + 1;
  uint32_t slot = h.s_slot;
  const uint32_t idx = take_slot(slot, &P.glob->s_alloc, P.s_chunk, li, seq + 1, lane);
  h.s_slot = slot;
  if (lane == 0) P.s_rec[idx] = rid;
}

// One chain's phase B.  Resolves the proposal the chain made in the last search (lowest chain id holds resv[rid])
// and applies the winner's step.  Every lane ends up with the same updated header `h` (the fields of the emission
// allocator excepted, which only lane 0 uses); lane 0 stores it.  DEFER: the shared state (taken[], needy[],
// cursor, alive) is updated by k_mg_mark instead (multi-GPU pools, fused rounds).  lds_refs: see pack_consensus.
// Returns false when the chain has nothing more to do this round (it is, or just became, done).
template <int NP, bool LITERAL, bool DEFER, bool OWNER_FIRST = false>
__device__ __forceinline__ bool apply_step(const DevParams &P, Chain *c, uint32_t cid, uint32_t li, ChainHot &h, int lane,
                                           WaveLds *ws, WaveLdsLiteral *wl, uint64_t *lds_refs) {
  const int kind = h.prop_kind;
  if (kind == PROP_FRESH) return true;  // first fused round: nothing proposed yet
  if (h.finishing) {  // seed-needing chain found the pool empty
    if (h.prev_unmatched) emit_single(P, h, li, h.prev, lane);
    h.done = 1; h.finishing = 0;
    if (!DEFER && lane == 0) {  // DEFER: every rank reads PK_NOSEED in the gathered words (k_mg_mark); the chain says PK_DONE from now on
      atomicAnd(&P.needy[cid >> 5], ~(1u << (cid & 31)));
      atomicSub(&P.glob->alive, 1u);
      if (h.cursor_writer) *P.cursor = -1;  // (see search_step: the pool is exhausted)
    }
    h.cursor_writer = 0;
    store_hot(c, h, lane);
    return false;
  }
  // who holds the read we proposed (load in flight while the update is computed)
  uint32_t owner_v = kind != PROP_NONE ? P.resv[h.prop_rid] : cid;
  if (DEFER && P.alts == 2 && kind == PROP_MATCH && h.alt1 && uni_u32(owner_v) != cid) {
    // the alternatives schedule: the first candidate went to another chain -- did pass 1 (k_alt_resolve) secure the second?
    // Same probe, same alignment, the next read of the bin (orc_reorder_rounds_alt); the words of k_mg_mark say the same.
    const uint32_t alt = h.alt1 - 1u;
    if (uni_u32(P.resv[alt]) == (ALT_KEY | cid)) { h.prop_rid = alt; owner_v = cid; }
  }
  const bool fail_path = kind == PROP_NONE && h.mode == MODE_SEARCH;
  bool do_upd = false, ureset = false, urev = false;
  uint32_t urid = 0;
  int ushift = 0;
  if (kind == PROP_MATCH) { do_upd = true; urid = h.prop_rid; urev = h.prop_rev & 1; ushift = h.prop_shift; }
  else if (kind == PROP_SEED) { do_upd = true; urid = h.prop_rid; ureset = true; }
  else if (fail_path && !h.left_search) { do_upd = true; urid = h.first_rid; ureset = true; urev = true; }  // reorder.h:567
  int n = P.L, R_new = h.ref_len;
  const int R_old = h.ref_len;
  bool nw = false;
  // OWNER_FIRST (the deep-bin variants of the fused round: contended pools lose one to two proposals per read): the
  // answer is waited for before the update is computed -- a loser then costs its header, this word and a store instead
  // of the count columns read and the speculative ones written (~20 requests); a winner pays one dependent step
  static_assert(!OWNER_FIRST || DEFER, "OWNER_FIRST: fused rounds only (the cursor is k_mg_mark's)");
  if (OWNER_FIRST && kind != PROP_NONE && uni_u32(owner_v) != cid) {
    if (h.mode == MODE_SEARCH) h.retrying = 1;
    store_hot(c, h, lane);
    if (lane == 0) atomicAdd((unsigned long long *)&c->st_lost, 1ull);
    return true;
  }
  if (do_upd) {
    if (!P.uniform_len) n = uni_i32((int)P.lens[urid]);
    // bytes first; the rare update that would push a count past 255 is redone in the wide format
    R_new = wave_update_compute<NP, LITERAL>(P, li, ws, wl, urid, n, ureset, urev, ushift, R_old, h.cnt_buf,
                                             h.cnt_wide != 0, false, nw, lane);
    if (nw) {
      bool o2;
      R_new = wave_update_compute<NP, LITERAL>(P, li, ws, wl, urid, n, ureset, urev, ushift, R_old, h.cnt_buf,
                                               h.cnt_wide != 0, true, o2, lane);
    }
  }
  if (!DEFER && kind == PROP_SEED && h.cursor_writer) {  // every seed proposed this round ends up taken, win or lose
    if (lane == 0) *P.cursor = (long long)h.prop_rid - 1;
    h.cursor_writer = 0;
  }
  R_new = uni_i32(R_new);
  nw = __builtin_amdgcn_readfirstlane((int)nw) != 0;
  const uint32_t owner = uni_u32(owner_v);
  if (owner != cid) {  // lost the read: retry, nothing committed
    if (h.mode == MODE_SEARCH) h.retrying = 1;
    store_hot(c, h, lane);
    if (lane == 0) atomicAdd((unsigned long long *)&c->st_lost, 1ull);  // no returned value: nothing to wait for
    return true;
  }
  PT(3);
  if (do_upd) {
    pack_consensus(ws, R_new, lane, c, lds_refs);
    h.ref_len = R_new;
    h.cnt_buf ^= 1;
    h.cnt_wide = nw;
  }
  PT(4);
  if (kind == PROP_MATCH) {
    const uint32_t rid = h.prop_rid;
    const int shift = ushift;
    if (!DEFER && lane == 0) {
      atomicOr((unsigned long long *)&P.taken[rid >> 6], 1ull << (rid & 63));
      atomicSub(&P.ublk[rid >> UBLK_SHIFT], 1u);
    }
    const bool left = h.left_search;
    long long ref_pos = h.ref_pos, cur_pos;
    char rcch;
    if (!urev) {  // reorder.h:490-497, :508
      if (!left) { cur_pos = ref_pos + shift; ref_pos = cur_pos; }
      else { cur_pos = ref_pos + R_old - shift - n; ref_pos = ref_pos + R_old - shift - R_new; }
      rcch = left ? 'r' : 'd';
    } else {  // reorder.h:528-535, :546
      if (!left) { cur_pos = ref_pos + R_old + shift - n; ref_pos = ref_pos + R_old + shift - R_new; }
      else { cur_pos = ref_pos - shift; ref_pos = cur_pos; }
      rcch = left ? 'd' : 'r';
    }
    if (h.prev_unmatched) emit_rec(P, h, li, h.prev, 'd', '0', 0, lane);
    emit_rec(P, h, li, rid, rcch, '1', cur_pos, lane);
    h.prev_unmatched = 0; h.ref_pos = ref_pos; h.retrying = 0;
  } else if (kind == PROP_SEED) {  // reorder.h:580-587, :600-613
    const uint32_t rid = h.prop_rid;
    if (lane == 0) {
      if (!DEFER) {
        atomicOr((unsigned long long *)&P.taken[rid >> 6], 1ull << (rid & 63));
        atomicAnd(&P.needy[cid >> 5], ~(1u << (cid & 31)));
      }
      atomicAdd((unsigned long long *)&c->n_unmatched, 1ull);
    }
    if (h.prev_unmatched) emit_single(P, h, li, h.prev, lane);
    h.prev_unmatched = 1; h.first_rid = rid; h.prev = rid;
    h.ref_pos = 0; h.mode = MODE_SEARCH;
  } else if (fail_path) {  // search failed (reorder.h:559-575)
    h.retrying = 0;
    h.num_unmatched_past++;
    if (!h.left_search) { h.left_search = 1; h.ref_pos = 0; }
    else {
      h.left_search = 0; h.mode = MODE_NEED_SEED;
      if (!DEFER && lane == 0) atomicOr(&P.needy[cid >> 5], 1u << (cid & 31));  // DEFER: k_mg_mark, from the PK_WILLNEED word
    }
  }
  store_hot(c, h, lane);
  PT(5);
  return true;
}

template <int NP, bool LITERAL>
__global__ __launch_bounds__(256) void k_apply(DevParams P) {
  __shared__ WaveLds lds[4];
  __shared__ WaveLdsLiteral ldsl[LITERAL ? 4 : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // local chain index (state arrays, emission tags); wave-uniform -> chain pointers live in SGPRs
  const uint32_t li = blockIdx.x * 4 + wave;
  if (li >= P.K) return;
  const uint32_t cid = P.c0 + li;             // global chain id (conflict priority)
  Chain *c = &P.chains[li];
  ChainHot h;
  load_hot(c, h);
  if (h.done) return;
  (void)apply_step<NP, LITERAL, false>(P, c, cid, li, h, lane, &lds[wave], &ldsl[LITERAL ? wave : 0], nullptr);
}

// ------------------------------------------------------------ fused round: apply(t-1) + search(t) in one kernel
// The two phases of consecutive rounds for one chain, back to back: the chain's header and consensus are loaded
// once, the new consensus goes from the update straight into the search's LDS copy, and a round is one chain
// kernel + k_mg_mark instead of two chain kernels.  Same schedule as the two-kernel round (same specification): every
// search of round t sees taken[] with all claims of round t-1, because k_mg_mark(t-1) -- which sets the winners'
// taken bits, the cursor and the needy bitmap from the proposal words -- runs between the two launches; the
// resv[] entries the apply halves read belong to reads that are all taken by then, so the proposals of round t
// never touch them.
// MG (one pool over several GPUs): a rank runs its own chains only and publishes one word per chain.  Its own proposals go
// straight to resv[] (atomicMin), as on one GPU; after the all-gather k_mg_resolve adds the other ranks' words (lowest
// chain id wins whatever the order), then k_mg_mark.  (MG no longer changes the code of the round kernels: round 3 left
// ALL the resolution to k_mg_resolve, a pass over every chain of the pool on every rank.)
template <int NP, bool STATS, bool MG, bool TRIM, bool LONG>
__device__ __forceinline__ void round_body(const DevParams &P) {
  __shared__ uint64_t s_refs[2][LDS_LIMBS];
  __shared__ uint16_t s_list[TAIL_CAP];
  __shared__ uint16_t s_stat[STATS ? 2 * TAIL_CAP : 2];
  __shared__ uint8_t s_pres[128];
  __shared__ uint32_t s_best;
  // the search's staging area; the apply half's scratch (WaveLds) is dead by then and shares the space
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[STAGE_WORDS];
  static_assert(sizeof(WaveLds) <= sizeof(uint32_t) * STAGE_WORDS, "WaveLds overlays s_stage");
  WaveLds &lds = *reinterpret_cast<WaveLds *>(s_stage);
  const int lane = threadIdx.x;
  const uint32_t li = P.g0 + blockIdx.x;  // (two-group schedule: the launch covers the chains [g0, g0 + Kg); else g0 = 0)
  const uint32_t cid = P.c0 + li;
  Chain *c = &P.chains[li];
#ifdef SR_PHASE_TIMING
  for (int i = lane; i < 66; i += 64) g_pt_lds[i] = 0;
  wave_sync();
  if (lane == 0) g_pt_lds[64] = (uint32_t)clock64();
#endif
  // ref / revref go to LDS; their loads are in flight while the header's scalar load is waited for
  uint64_t r0 = 0, r1 = 0;
  {
    const int i = lane - LDS_PAD;
    if (lane < LDS_LIMBS && i >= 0 && i < P.W) { r0 = c->ref[i]; r1 = c->revref[i]; }
  }
  ChainHot h;
  load_hot(c, h);
  if (lane < LDS_LIMBS) { s_refs[0][lane] = r0; s_refs[1][lane] = r1; }
  wave_sync();  // the update below rewrites s_refs
  if (h.done) {
    if (lane == 0) P.prop[cid] = (unsigned long long)PK_DONE << 32;
    return;
  }
  PTW(0);
#ifdef SR_NO_OWNER_FIRST  // experiment builds (tools/xbuild.sh): the A side of the A/B
  constexpr bool OWNER_FIRST = false;
#else
  constexpr bool OWNER_FIRST = TRIM && !STATS;
#endif
  if (!apply_step<NP, false, true, OWNER_FIRST>(P, c, cid, li, h, lane, &lds, nullptr, &s_refs[0][0])) {
    if (lane == 0) P.prop[cid] = (unsigned long long)PK_DONE << 32;
    PT_FLUSH(c);
    return;
  }
  (void)search_step<STATS, true, true, TRIM, LONG>(P, c, cid, h, lane, &s_refs[0][0], s_list, s_stat, s_pres, (lds_u32_t *)&s_best, (lds_u32_t *)s_stage);
  PT_FLUSH(c);
#ifdef SR_PHASE_TIMING
  {
    const int tot = wave_sum_i(lane < 32 ? (int)g_pt_lds[lane] : 0);
    if (tot > SR_PT_LONG) {
      atomicAdd(&P.dbg[lane], (unsigned long long)g_pt_lds[lane]);
      if (lane == 0) atomicAdd(&P.dbg[64], 1ull);
    }
  }
#endif
}
// Two kernels over the one body, because the register budget is a kernel attribute: the shallow-dictionary variant sits at
// 64 VGPRs (8 waves per SIMD) without scratch; the deep-bin variants (TRIM: tail trimming, balanced scan, resumed
// searches, owner-first apply, k_long hand-over) carry more live state and get SR_TRIM_WAVES waves per SIMD.
// Measured (profiles/r04_trim_waves.txt, 20 M-read pools, chains stage, 8 / 7 / 6 waves per SIMD): 400x 110.3 / 109.0 / 114.8 ms,
// 1 600x 124.3 / 122.0 / 128.4, 6 400x 143.4 / 139.5 / 148.2, 25 600x 183.2 / 177.5 / 184.3, PhiX-like 196.7 / 191.8 / 194.4:
// 7 waves (72 VGPRs) -- at 8 the variants spill 8-13 VGPRs to 20-40 bytes of scratch per lane, at 7 none; the variant with
// the k_long hand-over still spills one there and gets 6 (79 VGPRs).  No chain kernel uses scratch memory.
#ifndef SR_TRIM_WAVES
#define SR_TRIM_WAVES 7
#endif
#ifndef SR_LONG_WAVES
#define SR_LONG_WAVES 6
#endif
template <int NP, bool STATS, bool MG, bool TRIM, bool LONG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SR_ROUND_WAVES, SR_ROUND_WAVES))) void k_round(DevParams P) {
  static_assert(!TRIM, "deep-bin variants: k_round_t");
  round_body<NP, STATS, MG, false, false>(P);
}
template <int NP, bool STATS, bool MG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STATS ? 5 : SR_TRIM_WAVES, STATS ? 5 : SR_TRIM_WAVES))) void k_round_t(DevParams P) {
  round_body<NP, STATS, MG, true, false>(P);
}
template <int NP, bool MG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SR_LONG_WAVES, SR_LONG_WAVES))) void k_round_tl(DevParams P) {
  round_body<NP, false, MG, true, true>(P);
}

template <int NP, bool STATS, bool MG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STATS ? 5 : SR_ROUND_WAVES, STATS ? 5 : SR_ROUND_WAVES))) void k_round_nt(DevParams P) {
  round_body<NP, STATS, MG, false, false>(P);
}

// ------------------------------------------------------------ long searches: one block of 16 wavefronts per chain
// k_round hands a search over (search_step) when it turns out long: on deep-bin pools a chain whose consensus carries
// five or more errors fails its search only after comparing every live read of every bin it probes, as the reference
// does (reorder.h:262-316), and one wavefront doing that is what the whole round waits for.  Here the same search --
// same probes, same priority order, same MAX_SEARCH_REORDER rule, hence the same winner -- is spread over a block:
//   1. one thread per probe (code = shift << 2 | rev << 1 | dict, the reference's order): bucket fetch, single-read
//      bins compared at once, verified multi-read bins collected in code order;
//   2. the collected bins are cut into chunks of 64 entries (bin tail first), numbered in priority order; the 16
//      wavefronts take chunk numbers from a ticket counter and publish a passing entry with an atomicMin on its chunk
//      number: the first passing entry whose bin has fewer than MAX_SEARCH_REORDER live entries ahead of it wins (checked
//      for bins of more than that many entries only; a bin whose pass lies outside the window is left for the next one).
//      Nothing beyond the lowest passing chunk is started.  (Round 3 dealt 16 chunks per step between two block barriers,
//      thread 0 folding the results: 70 % of a search's clocks, profiles/r04_genomic.txt.)
// Dead bin tails are not trimmed here (k_round's scans and k_trim_bins do that).
// Signatures (k_long): beside every dictionary entry (same index as ids[l]) the first and the last limb of its read.  The
// Hamming distance over the bases of those two limbs that lie inside the compared range is a lower bound of the distance
// reorder.h:291-301 computes, so an entry whose bound already exceeds THRESH_REORDER is rejected from 16 bytes that arrive
// coalesced with its id -- exactly the entries the full compare would reject, minus a random 64-byte read each.  In the
// bins that matter here (the 32-mers of a repeat family: reads of hundreds of diverged copies) that is nine entries in ten.
__global__ void k_build_sig(const uint32_t *__restrict__ ids, uint64_t m, const uint64_t *__restrict__ reads, int S, int W,
                            ulonglong2 *__restrict__ sig) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint64_t *r = reads + (uint64_t)ids[i] * S;
  sig[i] = make_ulonglong2(r[0], r[W - 1]);
}
void launch_build_sig(hipStream_t st, const uint32_t *ids, uint64_t m, const uint64_t *reads, int S, int W, ulonglong2 *sig) {
  if (m) hipLaunchKernelGGL(k_build_sig, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, ids, m, reads, S, W, sig);
}
// lower bound of cmp_candidate's Hamming distance from the signature limbs (same range and masks as there)
__device__ __forceinline__ int sig_bound(const DevParams &P, const uint64_t *sx, int bitshift, int lo, int mref, uint32_t r,
                                         const ulonglong2 &sg) {
  const int W = P.W;
  const int clen = P.uniform_len ? P.L : (int)P.lens[r];
  const int m = clen < mref ? clen : mref;
  const int blo = 2 * lo, bhi = 2 * m;
  if (bhi <= blo) return 0;
  const int first = blo >> 6, last = (bhi - 1) >> 6;
  int hd = 0;
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int i = t ? W - 1 : 0;
    if ((t && W == 1) || i < first || i > last) continue;
    uint64_t y = lds_window(sx, i * 64 + bitshift) ^ (t ? sg.y : sg.x);
    if (i == first) y &= ~0ull << (blo & 63);
    if (i == last) y &= ~0ull >> (63 - ((bhi - 1) & 63));
    hd += __popcll(y);
  }
  return hd;
}

constexpr int LONG_WAVES = 16;
// (the abort hint of a split search, below: an agent-scope load -- the XCDs have an L2 each; a stale value only costs work)
template <typename T> __device__ __forceinline__ T ag_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifndef SR_LONG_NCH
#define SR_LONG_NCH 2
#endif
constexpr int LONG_NCH = SR_LONG_NCH;  // chunks of 64 bin entries per ticket
static_assert(2 * 64 * LONG_NCH <= STAGE_WORDS, "the packed survivors of a ticket fit the staging rows");
constexpr uint32_t LONG_NONE = 0x7fffffffu;
// a part is scanned by a block of SCAN_WAVES wavefronts, one thread per bin: at most LONG_PART_BINS bins per part.  Small
// blocks, several per CU: a part's serial phases (its header, the barriers and prefix sums of a turn) run while the
// wavefronts of other parts stream chunks -- with one block of 16 wavefronts per CU those phases left the CU idle, and a
// search cut into parts of 128 chunks took 2.5 s instead of 1.7 (100 M genome-like reads, profiles/r05_k_long.txt)
constexpr int SCAN_WAVES = 4;
constexpr int LONG_PART_BINS = 64 * SCAN_WAVES;

// Round 5: three kernels instead of one.  Round 4's k_long was one launch in which a block ran a whole search -- probes,
// then every chunk of every bin -- and a launch lasted as long as its longest search: 62 % of the blocks' time was idle on
// genome-like pools (a few searches per launch scan 2 000-3 500 chunks, ~1 ms, while the average one takes 40 us;
// profiles/r05_k_long.txt), and the split searches that were meant to cure it handed parts from block to block INSIDE the
// launch through agent-scope memory: a protocol that never became reliable.  Now the hand-over IS a kernel boundary:
//   k_long_list  one block per search: the probes (step 1 above), the verified bins written to the search's slot, the
//                search cut into parts = ranges of bins with about P.long_part listed chunks each, one entry per part
//                appended to the round's part list;
//   k_long_scan  one block per part (a ticket counter over the part list): step 2 above on the part's bins -- the first
//                pass inside its bin's window in the reference's order, or none;
//   k_long_fin   one wavefront per search: the lowest part with a pass wins (against the best single-read bin), the
//                proposal is written exactly as search_step writes it.
// Everything a later kernel reads was written by an earlier one; the only thing blocks of one launch tell each other is a
// hint -- bestpart, the lowest part of the search that has a pass so far: parts behind it stop (no result depends on it).

// ---- kernel 1: probes -> bin list -> parts
// NW wavefronts per block = 64 NW probe codes: 8 for reads of up to 256 bases (three searches in flight per CU instead of one)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_long_list(DevParams P) {
  __shared__ uint64_t s_refs[2][LDS_LIMBS];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[NW][STAGE_WORDS];
  __shared__ uint32_t s_bstart[64 * NW], s_bcount[64 * NW], s_bchunk0[64 * NW];
  __shared__ uint16_t s_bcode[64 * NW];
  __shared__ uint32_t s_wcnt[NW];
  __shared__ uint32_t s_best, s_bestrid, s_qi, s_base, s_bstart_part[LONG_MAX_PARTS + 2];
  static_assert(LONG_MAX_BINS >= 64 * NW, "one bin per thread");
  const int tid = threadIdx.x, wave = uni_i32(tid >> 6), lane = tid & 63;
  const uint32_t npend = P.longq[0];
  // (a fixed grid whatever the round handed over: a block that will find no search must cost nothing -- most rounds of the
  // PhiX-like pool queue a few dozen searches, and 4 000 blocks that each took a ticket first cost 50 us per round)
  if (blockIdx.x >= npend) return;
  const int klen2 = 2 * P.wl;
  const uint64_t kmask = klen2 < 64 ? ((1ull << klen2) - 1) : ~0ull;
  lds_u32_t *stage = (lds_u32_t *)s_stage[wave];
  const uint64_t *sref = &s_refs[0][0] + LDS_PAD, *srev = &s_refs[1][0] + LDS_PAD;
  constexpr uint32_t LONG_FIRST = MAX_SEARCH / 64;  // (whole chunks inside the window whatever is live)
  for (;;) {
    if (tid == 0) s_qi = atomicAdd(&P.longq[1], 1u);  // the searches differ in length: taken as the blocks become free
    __syncthreads();
    const uint32_t qi = uni_u32(s_qi);
    if (qi >= npend) break;
    const uint32_t li = uni_u32(P.longq[2 + qi]);
    Chain *c = &P.chains[li];
    if (tid < LDS_LIMBS) {
      const int i = tid - LDS_PAD;
      uint64_t r0 = 0, r1 = 0;
      if (i >= 0 && i < P.W) { r0 = c->ref[i]; r1 = c->revref[i]; }
      s_refs[0][tid] = r0; s_refs[1][tid] = r1;
    }
    ChainHot h;
    load_hot(c, h);
    if (tid == 0) { s_best = LONG_NONE; s_bestrid = 0; }
    __syncthreads();
    const int ref_len = h.ref_len;
    int min_code = 0;
    if (h.retrying) {  // as search_step: a repeated search resumes at the last winner's probe
      const int pr = (int)h.prop_rev;
      if (pr & 4) min_code = ((int)h.prop_shift << 2) | ((pr & 1) << 1) | ((pr >> 1) & 1);
    }
    // ---- one thread per probe
    const int code = tid, l = code & 1, rev = (code >> 1) & 1, shift = code >> 2;
    const bool valid = probe_valid(P, l, rev, shift, ref_len) && code >= min_code;
    bool hit = false, keyok = false, other = false;
    uint32_t rid = 0, ncand = 0;
    PendBin pend;
    pend.on = false; pend.start = pend.count = pend.pay = 0;
    if (valid) {
      const int ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
      const uint64_t *sx = rev ? srev : sref;
      const uint64_t key = lds_window(sx, rev ? 2 * (ds - shift) : 2 * (ds + shift)) & kmask;
      eval_probe<true, true>(P, sx, l, rev, shift, ref_len, key, mix64(key), 0u /* (hash-addressed tables only: the pipeline refuses table_mode = 2 for the one-chain kernels) */, hit, rid, keyok, ncand, other, (lds_u32_t *)&s_best, stage, lane, &pend);
    }
    __syncthreads();
    if (hit && s_best == (uint32_t)code) s_bestrid = rid;  // (eval_probe left the lowest hitting code in s_best)
    const uint32_t best_single = s_best;
    // (a bin behind a single-read bin that hit can never win: it is not listed)
    const bool mine = pend.on && pend.count > 0 && (uint32_t)code < best_single;
    const uint64_t pb = __ballot(mine);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(pb);
    __syncthreads();
    uint32_t nb = 0;
    {
      uint32_t base = 0;
      for (int w = 0; w < NW; w++) { const uint32_t v = s_wcnt[w]; if (w < wave) base += v; nb += v; }
      if (mine) {
        const uint32_t at = base + (uint32_t)__popcll(pb & ((1ull << lane) - 1));
        s_bstart[at] = pend.start; s_bcount[at] = pend.count; s_bcode[at] = (uint16_t)code;
      }
    }
    nb = uni_u32(nb);
    __syncthreads();  // (s_wcnt is about to be reused; s_b* of every wavefront written)
    // ---- the chunks the first turn of a scan lists (a bin of more than MAX_SEARCH entries: LONG_FIRST of them) -> parts
    const uint32_t my_cnt = (uint32_t)tid < nb ? s_bcount[tid] : 0u;
    const uint32_t my_nch = (my_cnt + 63u) / 64u;
    const uint32_t nch1 = my_cnt > (uint32_t)MAX_SEARCH ? (my_nch < LONG_FIRST ? my_nch : LONG_FIRST) : my_nch;
    const uint32_t incl = (uint32_t)wave_incl_scan_i((int)nch1, lane);
    if (lane == 63) s_wcnt[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < NW; w++) { const uint32_t v = s_wcnt[w]; if (w < wave) base += v; total += v; }
    total = uni_u32(total);
    s_bchunk0[tid] = base + incl - nch1;
    // parts: ranges of bins of about P.long_part listed chunks and at most LONG_PART_BINS bins -- cut at equal shares of the
    // weight chunks * LONG_PART_BINS + bins * part (a part full of either weighs part * LONG_PART_BINS); should a part still
    // come out with too many bins (many one-chunk bins beside a few long ones), the search is cut by bin count alone
    const uint32_t part_ch = P.long_part > 0 ? (uint32_t)P.long_part : 0x100000u;
    const unsigned long long wtot = (unsigned long long)total * LONG_PART_BINS + (unsigned long long)nb * part_ch;
    const unsigned long long wfull = (unsigned long long)part_ch * LONG_PART_BINS;
    uint32_t np = nb ? (uint32_t)((wtot + wfull - 1) / wfull) : 0u;
    if (np > (uint32_t)LONG_MAX_PARTS) np = LONG_MAX_PARTS;
    if (nb > np * (uint32_t)LONG_PART_BINS) np = (nb + LONG_PART_BINS - 1) / LONG_PART_BINS;  // (<= 4: nb <= 1024)
    if (tid == 0) { s_base = np ? atomicAdd(&P.lctl[0], np) : 0u; s_best = 0; }  // (s_best: "a part has too many bins")
    __syncthreads();
    LongHead *hd = P.lhead + qi;
    uint32_t first = 0;
    if ((uint32_t)tid <= np && np) {
      // part p starts at the first bin with p / np of the weight ahead of it
      first = (uint32_t)tid == np ? nb : 0u;
      if (tid > 0 && (uint32_t)tid < np) {
        const unsigned long long thr = (unsigned long long)tid * wtot / np;
        uint32_t lo2 = 0, hi2 = nb;  // first b in [0, nb] with weight ahead >= thr (it does not decrease)
        while (lo2 < hi2) {
          const uint32_t mid = (lo2 + hi2) >> 1;
          if ((unsigned long long)s_bchunk0[mid] * LONG_PART_BINS + (unsigned long long)mid * part_ch >= thr) hi2 = mid; else lo2 = mid + 1;
        }
        first = lo2;
      }
      s_bstart_part[tid] = first;
    }
    __syncthreads();
    if ((uint32_t)tid < np && s_bstart_part[tid + 1] - s_bstart_part[tid] > (uint32_t)LONG_PART_BINS) s_best = 1;
    __syncthreads();
    if ((uint32_t)tid <= np && np) {
      if (uni_u32(s_best)) {  // by bin count alone
        const uint32_t npb = (nb + LONG_PART_BINS - 1) / LONG_PART_BINS;
        first = (uint32_t)tid >= npb ? nb : (uint32_t)tid * LONG_PART_BINS;  // (parts beyond npb are empty)
      }
      hd->blo[tid] = first;
      if ((uint32_t)tid < np) P.lparts[uni_u32(s_base) + (uint32_t)tid] = (qi << 6) | (uint32_t)tid;
    }
    if ((uint32_t)tid < nb) {
      P.lbin[(size_t)qi * P.lbin_stride + tid] = make_uint2(s_bstart[tid], s_bcount[tid]);
      P.lbcode[(size_t)qi * P.lbin_stride + tid] = s_bcode[tid];
    }
    if (tid == 0) {
      hd->nb = nb; hd->nparts = np; hd->best_single = best_single; hd->bestrid = s_bestrid; hd->bestpart = 0xffffffffu;
      if (np > 1) atomicAdd(&P.lctl[3], 1u);  // (the run's split searches: stats.long_splits)
    }
    __syncthreads();  // the LDS state belongs to the next search of this block
  }
}

// ---- kernel 2: one part of a search = a range of its bins, in chunks of 64 entries, in the reference's order (bins by
// priority code, a bin from its tail).  A TURN lists a range of chunks [done, target) per bin; chunk0[b] = listed chunks
// ahead of bin b; a wavefront takes the next listed chunks from a ticket counter, finds their bin, compares, and publishes
// a passing entry with a 64-bit atomicMin on the key bin << 32 | chunk << 6 | lane; wavefronts stop taking tickets beyond
// the lowest key.  Keys order the entries as the reference visits them.
// The MAX_SEARCH_REORDER window (reorder.h:287-288) can only matter in a bin of more than that many entries (a "big"
// bin).  Big bins count their live entries (s_binlive) and are listed LONG_FIRST chunks at first; a pass found in one is
// checked after the turn's barrier against the live entries ahead of it (exact recount) -- outside the window the bin is
// left, as the reference leaves it, and the scan goes on behind it.  A big bin ahead of the best pass that has neither
// been listed to its end nor reached the limit is listed further in the next turn (how far: at the targets below): the
// result is the first pass inside its bin's window in key order, whatever the number of turns.
__global__ __launch_bounds__(64 * SCAN_WAVES) void k_long_scan(DevParams P) {
  __shared__ uint64_t s_refs[2][LDS_LIMBS];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[SCAN_WAVES][STAGE_WORDS];
  __shared__ uint32_t s_bstart[64 * SCAN_WAVES], s_bcount[64 * SCAN_WAVES];
  __shared__ uint16_t s_bcode[64 * SCAN_WAVES];
  __shared__ uint32_t s_wcnt[SCAN_WAVES];
  __shared__ uint32_t s_ctl, s_capped, s_qi, s_hint;
  __shared__ uint32_t s_bchunk0[64 * SCAN_WAVES];   // listed chunks ahead of bin b in this turn
  __shared__ uint32_t s_binlive[64 * SCAN_WAVES];   // live entries seen in bin b (bins of more than MAX_SEARCH entries only)
  __shared__ uint32_t s_bdone[64 * SCAN_WAVES];     // chunks of bin b that have been compared (all of them: the bin is out)
  __shared__ unsigned long long s_ticket;           // 64 * the next listed chunk of the turn
  __shared__ unsigned long long s_minpass;          // lowest key (bin << 32 | chunk << 6 | lane) of a passing entry ...
  __shared__ unsigned long long s_valid;            // ... and the lowest one that has been checked against its bin's window
  const int tid = threadIdx.x, wave = uni_i32(tid >> 6), lane = tid & 63;
  const uint32_t nparts_all = P.lctl[0];
  if (blockIdx.x >= nparts_all) return;
  const int klen2 = 2 * P.wl;
  lds_u32_t *stage = (lds_u32_t *)s_stage[wave];
  const uint64_t *sref = &s_refs[0][0] + LDS_PAD, *srev = &s_refs[1][0] + LDS_PAD;
  constexpr uint32_t LONG_FIRST = MAX_SEARCH / 64;
  for (;;) {
    if (tid == 0) s_qi = atomicAdd(&P.lctl[1], 1u);
    __syncthreads();
    const uint32_t ti = uni_u32(s_qi);
    if (ti >= nparts_all) break;
    const uint32_t pd = uni_u32(P.lparts[ti]), qi = pd >> 6, part = pd & 63u;
    LongHead *hd = P.lhead + qi;
    const uint32_t nparts = uni_u32(hd->nparts);
    const uint32_t b_lo = uni_u32(hd->blo[part]), nb = uni_u32(hd->blo[part + 1]) - b_lo;  // this part's bins, numbered from 0
    const uint32_t li = uni_u32(P.longq[2 + qi]);
    const Chain *c = &P.chains[li];
    if (tid < LDS_LIMBS) {
      const int i = tid - LDS_PAD;
      uint64_t r0 = 0, r1 = 0;
      if (i >= 0 && i < P.W) { r0 = c->ref[i]; r1 = c->revref[i]; }
      s_refs[0][tid] = r0; s_refs[1][tid] = r1;
    }
    const int ref_len = uni_i32(c->h.ref_len);
    if ((uint32_t)tid < nb) {
      const uint2 b = P.lbin[(size_t)qi * P.lbin_stride + b_lo + tid];
      s_bstart[tid] = b.x; s_bcount[tid] = b.y; s_bcode[tid] = P.lbcode[(size_t)qi * P.lbin_stride + b_lo + tid];
    }
    if (tid == 0) { s_capped = 0; s_minpass = ~0ull; s_valid = ~0ull; }
    __syncthreads();
    const uint32_t my_cnt = (uint32_t)tid < nb ? s_bcount[tid] : 0u;
    const uint32_t my_nch = (my_cnt + 63u) / 64u;
    const bool my_big = my_cnt > (uint32_t)MAX_SEARCH;
    s_binlive[tid] = 0;
    s_bdone[tid] = 0;
    for (;;) {
      __syncthreads();
      // the range of thread tid's bin in this turn
      const uint32_t my_done = s_bdone[tid], live_prev = s_binlive[tid];
      uint32_t my_tgt = my_nch;
      if (my_big) {
        if (live_prev >= (uint32_t)MAX_SEARCH) my_tgt = my_done;
        else {
          // How far a big bin is listed.  The reads that pass but lie beyond the window are never taken through this bin, so
          // they pile up right behind it while the window itself is emptied: a chunk that straddles or overshoots the window's
          // end nearly always holds a pass, and a pass outside the window costs a turn of its own (the bin is left, everything
          // behind it is listed again -- 30 to 90 such turns in the longest searches when a bin was simply listed four times
          // as far, r04_genomic.txt).  So: (1) the chunks that cannot reach past the window even if every entry is live
          // ((MAX - live) / 64 of them; at first floor(1000 / 64) = 15), whose passes need no check; (2) when that is none,
          // ONE chunk, which the wavefront that compares it clips at the window exactly (it knows the live entries ahead:
          // live so far + its own lanes); (3) only where the share of live entries says the window is still far, a jump
          // of 7/8 of the estimated distance (at most three times what is done) -- the one case left for the check below.
          uint32_t far = LONG_FIRST;
          if (my_done) {
            const uint32_t room = (uint32_t)MAX_SEARCH - live_prev;
            uint32_t ext = room / 64u ? room / 64u : 1u;
            unsigned long long est = live_prev ? ((unsigned long long)room * my_done + live_prev - 1) / live_prev : 3ull * my_done;
            if (est > 3ull * my_done) est = 3ull * my_done;
            if (est >= 2ull * ext + 4ull) { const uint32_t jump = (uint32_t)(est - est / 8 - 1); if (jump > ext) ext = jump; }
            far = my_done + ext;
          }
          my_tgt = far < my_nch ? far : my_nch;
        }
      }
      if (my_tgt < my_done) my_tgt = my_done;
      const uint32_t nch = my_tgt - my_done;
      const uint32_t incl = (uint32_t)wave_incl_scan_i((int)nch, lane);
      if (lane == 63) s_wcnt[wave] = incl;
      __syncthreads();
      uint32_t base = 0, total = 0;
      for (int w = 0; w < SCAN_WAVES; w++) { const uint32_t v = s_wcnt[w]; if (w < wave) base += v; total += v; }
      // (everything a barrier depends on is made a scalar: a branch the compiler takes for divergent is run through with an
      // empty EXEC mask, and an s_barrier inside it still counts)
      total = uni_u32(total);
      s_bchunk0[tid] = base + incl - nch;
      // (the hint is read by ONE thread and handed round through LDS: leaving the turn loop is a decision of the block -- every
      // wavefront loading it for itself, as round 4's split searches did, lets some leave and some stay, and from then on
      // the barriers pair threads at different places: the "unresolved race" of round 4, and its hangs)
      if (tid == 0) { s_ticket = 0; s_hint = nparts > 1 ? ag_ld(&hd->bestpart) : 0xffffffffu; }
      __syncthreads();
      if (total == 0) break;
      // a part ahead of this one has a pass inside its window: nothing here can win (a hint: see the header comment)
      if (uni_u32(s_hint) < part) break;
      for (;;) {
        // No `if (lane == 0)` around the atomics of this loop: with one at the end of an iteration (the atomicMin) and one
        // at the start of the next (the ticket) the compiler threads the two branches, and lanes 1-63 go round the loop on
        // their own with readfirstlane(g) == 0 -- the same chunk for ever (seen in the ISA, round 4).  Every lane takes part
        // in the atomic instead: the 64 lanes add 1 each (the backend folds that into one LDS add of 64), so s_ticket counts
        // in units of 64.
        // A ticket is LONG_NCH consecutive listed chunks: a chunk is three dependent memory round trips (id + signature,
        // bitmap word, the survivors' reads) and a wavefront has nothing else to do meanwhile; with several in flight the
        // round trips are shared, and the survivors of all of them (one entry in twelve passes the signature bound on
        // genome-like pools) are packed into one compare pass.
        const uint32_t g = uni_u32((uint32_t)(atomicAdd(&s_ticket, 1ull) >> 6));
        if (g >= (total + LONG_NCH - 1) / LONG_NCH) break;
        if (nparts > 1 && uni_u32(ag_ld(&hd->bestpart)) < part) break;
        const uint32_t c0 = g * LONG_NCH;
        uint32_t lo = 0, hi = nb;  // the bin of listed chunk c0: the last b with chunk0[b] <= c0 (it has chunks: c0 < total)
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (uni_u32(s_bchunk0[mid]) <= c0) lo = mid; else hi = mid; }
        {
          const uint32_t q0 = uni_u32(s_bdone[lo]) + (c0 - uni_u32(s_bchunk0[lo]));
          // (s_minpass only decreases during a turn: a key seen below this ticket's first means the final one is below it too)
          const unsigned long long mp = __hip_atomic_load(&s_minpass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const uint32_t mp_hi = uni_u32((uint32_t)(mp >> 32)), mp_lo = uni_u32((uint32_t)mp);
          if (mp_hi < lo || (mp_hi == lo && (mp_lo >> 6) < q0)) break;
        }
        typedef const __attribute__((address_space(1))) uint32_t g_u32_t;
        // A: id and signature of every entry of the chunks
        uint32_t kb[LONG_NCH], kq[LONG_NCH], rk[LONG_NCH], kdead[LONG_NCH];
        bool kbig[LONG_NCH], has[LONG_NCH];
        uint32_t kclip[LONG_NCH];  // live entries of the bin ahead of this chunk when it is to be clipped at the window, else ~0
        int kcode[LONG_NCH];
        ulonglong2 sgk[LONG_NCH];
        {
          uint32_t bb = lo;
#pragma unroll
          for (int k = 0; k < LONG_NCH; k++) {
            const uint32_t cc = c0 + k;
            const bool kv = cc < total;
            if (kv) while (bb + 1 < nb && uni_u32(s_bchunk0[bb + 1]) <= cc) bb++;
            kb[k] = bb;
            kq[k] = kv ? uni_u32(s_bdone[bb]) + (cc - uni_u32(s_bchunk0[bb])) : 0u;
            kcode[k] = uni_i32((int)s_bcode[bb]);
            const uint32_t cnt = uni_u32(s_bcount[bb]), st0 = uni_u32(s_bstart[bb]);
            kbig[k] = cnt > (uint32_t)MAX_SEARCH;
            // a big bin listed with a single chunk: the chunk is clipped at the window (the live entries so far are exact,
            // nobody else adds to them in this turn)
            kclip[k] = 0xffffffffu;
            if (kv && kbig[k] && (bb + 1 < nb ? uni_u32(s_bchunk0[bb + 1]) : total) - uni_u32(s_bchunk0[bb]) == 1u) kclip[k] = uni_u32(s_binlive[bb]);
            const long long j = (long long)cnt - 1 - ((long long)kq[k] * 64 + lane);
            has[k] = kv && j >= 0;
            rk[k] = 0; sgk[k] = make_ulonglong2(0, 0); kdead[k] = 0;
            if (has[k]) {
              const int pl = kcode[k] & 1;
              g_u32_t *pids = (g_u32_t *)(pl ? uni_ptr(P.ids[1]) : uni_ptr(P.ids[0]));
              const ulonglong2 *psig = pl ? uni_ptr(P.sig[1]) : uni_ptr(P.sig[0]);
              rk[k] = pids[st0 + (uint32_t)j];
              sgk[k] = psig[st0 + (uint32_t)j];
              kdead[k] = rk[k] >> 31;  // (meaningful with DevParams::epos only)
              rk[k] &= P.idmask;
            }
          }
        }
        // B: signature bound; the bitmap word (one random request per entry) only where it matters -- the number of live
        // entries counts in a big bin only, elsewhere the entries that pass the bound are asked
        bool sp[LONG_NCH];
        uint64_t tw[LONG_NCH];
#pragma unroll
        for (int k = 0; k < LONG_NCH; k++) {
          const int prev = (kcode[k] >> 1) & 1, psh = kcode[k] >> 2;
          sp[k] = has[k] && sig_bound(P, prev ? srev : sref, prev ? -2 * psh : 2 * psh, prev ? psh : 0,
                                      prev ? ref_len + psh : ref_len - psh, rk[k], sgk[k]) <= THRESH;
          tw[k] = ~0ull;
          if (has[k] && (kbig[k] || sp[k])) {
            if (P.idmask != 0xffffffffu && (kdead[k] || P.phases != 2)) tw[k] = kdead[k] ? ~0ull : 0ull;  // (the entry said it: no request)
            else tw[k] = P.taken[rk[k] >> 6];  // (two chain groups: a clear flag may still be a read taken in this group's view)
          }
        }
        // C: the entries that are free and pass the bound, packed in key order (chunk, then lane) into the staging rows
        uint32_t nsrv = 0;
#pragma unroll
        for (int k = 0; k < LONG_NCH; k++) {
          const bool lv = !((tw[k] >> (rk[k] & 63)) & 1ull);
          const uint64_t Lm = __ballot(lv);
          bool sv = lv && sp[k];
          if (kclip[k] != 0xffffffffu && kclip[k] + (uint32_t)__popcll(Lm & ((1ull << lane) - 1)) >= (uint32_t)MAX_SEARCH) sv = false;  // outside the window
          const uint64_t Sm = __ballot(sv);
          if (kbig[k] && Lm) atomicAdd(&s_binlive[kb[k]], lv ? 1u : 0u);
          if (sv) {
            const uint32_t at = nsrv + (uint32_t)__popcll(Sm & ((1ull << lane) - 1));
            stage[at] = rk[k];
            stage[64 * LONG_NCH + at] = (uint32_t)((k << 6) | lane);
          }
          nsrv += (uint32_t)__popcll(Sm);
        }
        nsrv = uni_u32(nsrv);
        if (nsrv == 0) continue;
        // (the compare stages reads through the same rows: everything is taken out first)
        uint32_t pr[LONG_NCH], po[LONG_NCH];
#pragma unroll
        for (int pp = 0; pp < LONG_NCH; pp++) {
          const uint32_t idx = (uint32_t)pp * 64 + lane;
          pr[pp] = idx < nsrv ? stage[idx] : 0u;
          po[pp] = idx < nsrv ? stage[64 * LONG_NCH + idx] : 0u;
        }
#pragma unroll
        for (int pp = 0; pp < LONG_NCH; pp++) {
          if ((uint32_t)pp * 64 >= nsrv) break;
          const uint32_t idx = (uint32_t)pp * 64 + lane;
          const int ok_ = (int)(po[pp] >> 6);
          uint32_t myb = kb[0], myq = kq[0];
          int pcode = kcode[0];
#pragma unroll
          for (int k = 1; k < LONG_NCH; k++) if (ok_ == k) { myb = kb[k]; myq = kq[k]; pcode = kcode[k]; }
          bool ps = false;
          if (idx < nsrv) {
            const int pl = pcode & 1, prev = (pcode >> 1) & 1, psh = pcode >> 2;
            const int pds = pl ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
            ps = cmp_candidate<true, true>(P, prev ? srev : sref, prev ? -2 * psh : 2 * psh, prev ? psh : 0,
                                           prev ? ref_len + psh : ref_len - psh, pds, klen2, pr[pp], false, stage, lane) == 1;
          }
          const uint64_t Pm = __ballot(ps);
          if (Pm) {
            const int fp = __ffsll((unsigned long long)Pm) - 1;
            const uint32_t wb_ = (uint32_t)__shfl((int)myb, fp, 64), wq_ = (uint32_t)__shfl((int)myq, fp, 64);
            const uint32_t wl_ = (uint32_t)__shfl((int)(po[pp] & 63u), fp, 64);
            atomicMin(&s_minpass, ((unsigned long long)wb_ << 32) | ((unsigned long long)wq_ << 6) | wl_);  // (the same value from every lane)
            break;  // (the packed order is the key order: nothing behind the first pass matters)
          }
        }
      }
      __syncthreads();
      // Every listed chunk with a key below the final s_minpass has been compared (tickets go out in key order and a
      // wavefront only stops on a key lower than its chunk's), so the final s_minpass is the first pass of this turn's list.
      const uint32_t c_hi = uni_u32(((const uint32_t *)&s_minpass)[1]), c_lo = uni_u32(((const uint32_t *)&s_minpass)[0]);
      const uint32_t v_hi = uni_u32(((const uint32_t *)&s_valid)[1]), v_lo = uni_u32(((const uint32_t *)&s_valid)[0]);
      const bool fresh = c_hi != v_hi || c_lo != v_lo;  // a pass ahead of the best checked one
      bool ok = true;
      const uint32_t cb = c_hi, cq = c_lo >> 6, cfp = c_lo & 63u;
      if (fresh) {
        const uint32_t cnt = uni_u32(s_bcount[cb]);
        if (cnt > (uint32_t)MAX_SEARCH && uni_u32(s_binlive[cb]) >= (uint32_t)MAX_SEARCH) {
          // the live entries of bin cb ahead of the passing entry (every thread counts its share; rare)
          typedef const __attribute__((address_space(1))) uint32_t g_u32_t;
          g_u32_t *pids = (g_u32_t *)((uni_u32(s_bcode[cb]) & 1) ? uni_ptr(P.ids[1]) : uni_ptr(P.ids[0]));
          const uint32_t st0 = uni_u32(s_bstart[cb]);
          __syncthreads();
          if (tid == 0) s_ctl = 0;
          __syncthreads();
          const unsigned long long ahead = (unsigned long long)cq * 64 + cfp;  // entries of the bin visited before the pass
          uint32_t mycnt = 0;
          for (unsigned long long pp = (unsigned long long)tid; pp < ahead; pp += 64 * SCAN_WAVES)
            { uint32_t rr = pids[st0 + (cnt - 1 - (uint32_t)pp)]; bool tr_; mycnt += !entry_dead(P, rr, tr_); }
          const uint32_t wsum = (uint32_t)wave_sum_i((int)mycnt);
          if (wsum) atomicAdd(&s_ctl, lane == 0 ? wsum : 0u);
          __syncthreads();
          ok = uni_u32(s_ctl) < (uint32_t)MAX_SEARCH;
        }
      }
      __syncthreads();
      // the lists of the next turn
      if (!fresh) {
        s_bdone[tid] = my_tgt;  // nothing stopped early: every listed chunk was compared
      } else if (ok) {
        // the best pass so far: the bins behind it are out, the bins ahead of it are complete up to their targets
        if (tid == 0) {
          s_valid = ((unsigned long long)c_hi << 32) | c_lo;
          if (nparts > 1) atomicMin(&hd->bestpart, part);  // (the parts behind this one can stop)
        }
        s_bdone[tid] = (uint32_t)tid >= cb ? my_nch : my_tgt;
      } else {
        // outside the window: bin cb is left.  The chunks behind it may or may not have been compared (wavefronts stopped at
        // the key): their bins keep their lists and forget this turn's live entries
        // (that the probe of bin cb stopped at the window -- the search's "capped" flag -- is said at the end, and only if the
        // bin lies ahead of the winner: s_binlive[cb] >= MAX_SEARCH is what brought the check about)
        if (tid == 0) s_minpass = s_valid;
        if ((uint32_t)tid < cb) s_bdone[tid] = my_tgt;
        else if ((uint32_t)tid == cb) s_bdone[tid] = my_nch;
        else s_binlive[tid] = live_prev;
      }
    }
    // a bin ahead of the winner (any bin when nothing won) that held MAX_SEARCH live entries stopped its probe at the window
    // (a part that stopped on the hint has no pass and lies behind the winning part: k_long_fin does not look at it)
    {
      const uint32_t v_hi = uni_u32(((const uint32_t *)&s_valid)[1]);
      const uint32_t wb = v_hi != 0xffffffffu ? v_hi : nb;
      if ((uint32_t)tid < wb && my_big && s_binlive[tid] >= (uint32_t)MAX_SEARCH) s_capped = 1;
    }
    __syncthreads();
    if (tid == 0) {
      const uint32_t v_hi = ((const uint32_t *)&s_valid)[1], v_lo = ((const uint32_t *)&s_valid)[0];
      uint32_t wcode = LONG_NONE, wrid = 0;
      if (v_hi != 0xffffffffu) {
        typedef const __attribute__((address_space(1))) uint32_t g_u32_t;
        g_u32_t *pids = (g_u32_t *)((s_bcode[v_hi] & 1) ? P.ids[1] : P.ids[0]);
        wcode = s_bcode[v_hi];
        wrid = pids[s_bstart[v_hi] + (s_bcount[v_hi] - 1 - ((v_lo >> 6) * 64 + (v_lo & 63u)))] & P.idmask;
      }
      hd->rescode[part] = wcode; hd->resrid[part] = wrid; hd->capped[part] = s_capped;
    }
    __syncthreads();  // the LDS state belongs to the next part of this block
  }
}

// ---- kernel 3: one wavefront per search -- the lowest part with a pass against the best single-read bin; the proposal
// (as the end of search_step)
__global__ __launch_bounds__(256) void k_long_fin(DevParams P, int direct) {
  __shared__ uint64_t s_refs[4][2][LDS_LIMBS];                                  // (the alternatives schedule: find_alt)
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[4][STAGE_WORDS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t npend = P.longq[0];
  for (uint32_t qi = blockIdx.x * 4 + (threadIdx.x >> 6); qi < npend; qi += gridDim.x * 4) {
  const uint32_t li = uni_u32(P.longq[2 + qi]), cid = P.c0 + li;
  Chain *c = &P.chains[li];
  const LongHead *hd = P.lhead + qi;
  ChainHot h;
  load_hot(c, h);
  const uint32_t np = uni_u32(hd->nparts);
  const uint32_t rc_ = (uint32_t)lane < np ? hd->rescode[lane] : LONG_NONE;
  const uint32_t cp_ = (uint32_t)lane < np ? hd->capped[lane] : 0u;
  const uint64_t fm = __ballot(rc_ != LONG_NONE);
  const int wp = fm ? __ffsll((unsigned long long)fm) - 1 : 63;  // the winning part (none: every part counts for the flag)
  const bool cap_u = __ballot(cp_ != 0 && lane <= wp) != 0;
  const uint32_t wm = fm ? (uint32_t)__shfl((int)rc_, wp, 64) : LONG_NONE;
  const uint32_t wmr = fm ? uni_u32(hd->resrid[wp]) : 0u;
  const uint32_t ws = uni_u32(hd->best_single);
  const bool found = wm != LONG_NONE || ws != LONG_NONE;
  const uint32_t wcode = wm < ws ? wm : ws;
  const uint32_t wrid = wm < ws ? wmr : uni_u32(hd->bestrid);
  if (found) {
    h.prop_rid = wrid;
    h.prop_shift = wcode >> 2;
    h.prop_rev = ((wcode >> 1) & 1) | ((wcode & 1) << 1) | (cap_u ? 0u : 4u);  // rev | dict << 1 | resumable << 2
    h.prop_kind = PROP_MATCH;
    h.alt1 = 0;
    if (P.alts == 2) {  // the second candidate, as search_step finds it
      if (lane < LDS_LIMBS) {
        const int i = lane - LDS_PAD;
        const bool in = i >= 0 && i < P.W;
        s_refs[wv][0][lane] = in ? c->ref[i] : 0ull;
        s_refs[wv][1][lane] = in ? c->revref[i] : 0ull;
      }
      wave_sync();
      h.alt1 = uni_u32(find_alt<true>(P, &s_refs[wv][0][0] + LDS_PAD, &s_refs[wv][1][0] + LDS_PAD, h.ref_len, (int)wcode, wrid,
                                      (lds_u32_t *)s_stage[wv], lane)) + 1u;
      wave_sync();
    }
  } else {
    h.prop_kind = PROP_NONE;
  }
  store_hot(c, h, lane, 2, 4);
  if (lane == 0) {
    if (found) {
      P.prop[cid] = ((unsigned long long)PK_MATCH << 32) | wrid | ((unsigned long long)h.alt1 << PK_ALT_SHIFT);
      if (direct) atomicMin(&P.resv[wrid], cid);
    } else {
      P.prop[cid] = ((unsigned long long)PK_NONE << 32) | (h.left_search ? PK_WILLNEED_BIT : 0ull);
    }
    c->st_long++;  // (the chain is this wavefront's alone)
  }
  }
}

#include "reorder_round_mc.h"

// ---------------------------------------------- single-pool multi-GPU: kernels after the exchange
// prop[] now holds every rank's proposals.  All ranks run these two kernels over ALL chains and therefore keep
// identical taken[] / resv[] / needy[] / cursor replicas; k_search and k_apply only touch the chains a rank owns.
// (Work per rank that grows with the number of GPUs: two thread-per-chain passes, a few microseconds.)

// lowest chain id wins a contested read
// (the OTHER ranks' words only: a rank settles its own proposals in its round kernel, as a single GPU does)
// (the launch's group: global chains [gg0, gg0 + gKg) less this rank's slice [c0 + g0, c0 + g0 + Kg); one group: all of them)
__global__ void k_mg_resolve(DevParams P) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.gKg - P.Kg) return;
  const uint32_t cid = P.gg0 + (t < P.c0 + P.g0 - P.gg0 ? t : t + P.Kg);
  const unsigned long long pv = P.prop[cid];
  const int pk = (int)(pv >> 32) & 7;
  if (pk == PK_MATCH || pk == PK_SEED) atomicMin(&P.resv[(uint32_t)pv], cid);
}
// pass 1 of the alternatives schedule (after every pass-0 reservation of the round -- the round kernel's, k_long_fin's, and in
// a multi-GPU pool k_mg_resolve's -- and before k_mg_mark): a chain whose first candidate went to another chain proposes
// its second one; a read secured in pass 0 stays with its owner (ALT_KEY | chain > every chain id)
__global__ void k_alt_resolve(DevParams P) {
  const uint32_t cid = blockIdx.x * blockDim.x + threadIdx.x;
  if (cid >= P.Ktot) return;
  const unsigned long long pv = P.prop[cid];
  if (((int)(pv >> 32) & 7) != PK_MATCH || !(pv >> PK_ALT_SHIFT)) return;
  if (P.resv[(uint32_t)pv] == cid) return;
  atomicMin(&P.resv[(uint32_t)(pv >> PK_ALT_SHIFT) - 1u], ALT_KEY | cid);
}
// winners claim their read on every replica; the lowest seed of the round moves the cursor; the needy bitmap
// (+ its per-2048-chain counts) for the seed ranking of the NEXT round's k_search: chains whose left search just
// failed (k_apply sends them for a seed) and chains whose seed went to a lower chain id; chains still running
__global__ __launch_bounds__(256) void k_mg_mark(DevParams P) {
  const uint32_t cid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool needy = false, alive = false;
  int cls = -1;  // class of the chain's next round (k_round_mc), local chains that are still running only
  if (cid < P.Ktot) {
    const unsigned long long pv = P.prop[cid];
    const int pk = (int)(pv >> 32) & 7;
    alive = pk != PK_DONE;
    if (alive && cid >= P.c0 && cid - P.c0 < P.K) cls = pk == PK_MATCH ? 2 : pk == PK_NONE ? ((pv & PK_WILLNEED_BIT) ? 3 : 0) : 3;
    if (pk == PK_MATCH || pk == PK_SEED) {
      uint32_t rid = (uint32_t)pv;
      bool won = P.resv[rid] == cid;
      if (!won && pk == PK_MATCH && (pv >> PK_ALT_SHIFT)) {  // the alternatives schedule: the second candidate, secured in pass 1
        const uint32_t alt = (uint32_t)(pv >> PK_ALT_SHIFT) - 1u;
        if (P.resv[alt] == (ALT_KEY | cid)) { won = true; rid = alt; }
      }
      if (won) {
        atomicOr((unsigned long long *)&P.taken[rid >> 6], 1ull << (rid & 63));
        mark_dead(P, rid);
        // (matches only: seeds all come from the top of the pool -- thousands of same-address atomics per round --
        // and find_seed counts the cursor's block from the bitmap)
        if (pk == PK_MATCH) atomicSub(&P.ublk[rid >> UBLK_SHIFT], 1u);
      }
      if (pv & PK_CURSOR_BIT) *P.cursor = (long long)(uint32_t)pv - 1;  // every seed proposed this round ends up taken
      needy = pk == PK_SEED && !won;
      if (pk == PK_SEED && won && cls >= 0) cls = 1;
    } else if (pk == PK_NONE) {
      needy = (pv & PK_WILLNEED_BIT) != 0;
    } else if (pk == PK_NOSEED && (pv & PK_CURSOR_BIT)) {
      *P.cursor = -1;  // the last-ranked needy chain found nothing: the pool is exhausted (search_step)
    }
  }
  const uint64_t nb = __ballot(needy);
  const uint32_t na = (uint32_t)__popcll(__ballot(alive));
  const uint32_t w0 = (cid & ~63u) >> 5;  // a wavefront covers two bitmap words
  if (lane == 0 && (cid & ~63u) < P.Ktot) {
    P.needy[w0] = (uint32_t)nb;
    if ((cid & ~63u) + 32 < ((P.Ktot + 31) & ~31u)) P.needy[w0 + 1] = (uint32_t)(nb >> 32);
    if (nb) atomicAdd(&P.needy_cnt_next[cid >> 11], (uint32_t)__popcll(nb));
    P.alive_wave[cid >> 6] = na;  // summed by the host when it checks for the end (a same-address atomic per wavefront
                                  // was most of this kernel's time)
  }
  // the counters the NEXT round's k_mg_mark accumulates into (nobody reads them before that)
  if (cid < (P.Ktot + 2047) / 2048) P.needy_cnt[cid] = 0;
  if (cid == 0 && P.longq) { P.longq[0] = 0; P.longq[1] = 0; }  // this round's long searches are done (k_long ran before this kernel)
  if (P.longq && cid < 2) P.lctl[cid] = 0;
  if (P.ord) {  // class lists of this block's chains (k_round_mc): class 0 first, no atomics
    static_assert(MARK_BLOCK == 256, "k_mg_mark runs 256 chains per block");
    __shared__ uint32_t s_wc[4][4];  // [wave][class]
    const int wv = threadIdx.x >> 6;
    uint32_t mypos = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t m = __ballot(cls == k);
      if (lane == 0) s_wc[wv][k] = (uint32_t)__popcll(m);
      if (cls == k) mypos = (uint32_t)__popcll(m & ((1ull << lane) - 1));
    }
    __syncthreads();
    if (cls >= 0) {
      uint32_t base = 0;
      for (int k = 0; k < cls; k++) base += s_wc[0][k] + s_wc[1][k] + s_wc[2][k] + s_wc[3][k];
      for (int w = 0; w < wv; w++) base += s_wc[w][cls];
      P.ord[(size_t)blockIdx.x * MARK_BLOCK + base + mypos] = cid - P.c0;
    }
    if (threadIdx.x == 0)
      P.ord_cnt[blockIdx.x] = make_uint4(s_wc[0][0] + s_wc[1][0] + s_wc[2][0] + s_wc[3][0], s_wc[0][1] + s_wc[1][1] + s_wc[2][1] + s_wc[3][1],
                                         s_wc[0][2] + s_wc[1][2] + s_wc[2][2] + s_wc[3][2], s_wc[0][3] + s_wc[1][3] + s_wc[2][3] + s_wc[3][3]);
  }
}
// ---------------------------------------------- the two-group schedule (DevParams::phases = 2): one group's mark step
// Specification: orc_reorder_rounds_ph (the test suite's CPU restatement of the schedule).  The launch covers the chains [g0, g0 + Kg) of ONE group
// (one GPU: c0 = 0).  What k_mg_mark does for every chain -- winners claim their read, the lowest seed of the round moves the
// cursor, needy bitmap + counts, running chains, the class lists of k_round_mc -- for this group's chains, on this group's
// view (taken / resv / cursor / needy counts are the group's own); and what only two groups need:
//  * a proposal that holds its resv[] entry still loses when the OTHER group has taken the read since this group's
//    search looked at the pool (taken_other, complete up to the other group's last mark step -- the step in front of this
//    one: the host orders the mark steps A, B, A, B ... by events); resv[] then gets RESV_LOST, which is what the chain's
//    apply half reads in the next round kernel (it must not look at taken_other: the other group's next mark step may run
//    beside it);
//  * won[]: the read every chain secured (bit 31: a match), for the other group's next mark step;
//  * won_other[]: the other group's winners of its last mark step go into THIS group's view, and out of the block counts
//    of this group's seed region (ublk[b] belongs to the group that picks seeds from block b: only its mark step writes it,
//    so its round kernel -- which may run beside the other group's mark step -- reads settled counts).
constexpr uint32_t RESV_LOST = 0xfffffffeu;
// ONE wavefront per block of MARK_BLOCK chains, four chains per lane: the kernel runs beside the other group's round
// kernel, whose one-wavefront workgroups take every slot the moment it frees -- a workgroup of four wavefronts waits
// for four free slots on one CU and starves (k_ph_mark as a copy of k_mg_mark, 256 threads per block, took 145 us beside
// the round kernel of a deep pool and 11 us alone).
__global__ __launch_bounds__(64) void k_ph_mark(DevParams P) {
  constexpr int CPL = MARK_BLOCK / 64;  // chains per lane
  const int lane = threadIdx.x;
  const uint32_t gend = P.gg0 + P.gKg;  // (global chain ids: a multi-GPU pool runs this step over the whole group on every rank)
  const uint32_t cbase = P.gg0 + blockIdx.x * MARK_BLOCK;  // sub-block j holds chains cbase + 64 j + lane
  unsigned long long pv[CPL];
  uint32_t rs[CPL];
  bool inr[CPL];
#pragma unroll
  for (int j = 0; j < CPL; j++) {  // (all proposal words first, then all reservation words: two round trips, not eight)
    const uint32_t cid = cbase + 64 * j + lane;
    inr[j] = cid < gend;
    pv[j] = inr[j] ? P.prop[cid] : ((unsigned long long)PK_DONE << 32);
  }
#pragma unroll
  for (int j = 0; j < CPL; j++) {
    const int pk = (int)(pv[j] >> 32) & 7;
    rs[j] = (pk == PK_MATCH || pk == PK_SEED) ? P.resv[(uint32_t)pv[j]] : 0xffffffffu;
  }
  int cls[CPL];
#pragma unroll
  for (int j = 0; j < CPL; j++) {
    const uint32_t cid = cbase + 64 * j + lane;
    const int pk = (int)(pv[j] >> 32) & 7;
    bool needy = false;
    const bool alive = pk != PK_DONE;
    cls[j] = -1;
    if (inr[j]) {
      uint32_t wonv = 0xffffffffu;
      if (alive && cid - P.c0 - P.g0 < P.Kg) cls[j] = pk == PK_MATCH ? 2 : pk == PK_NONE ? ((pv[j] & PK_WILLNEED_BIT) ? 3 : 0) : 3;  // (this rank's chains)
      if (pk == PK_MATCH || pk == PK_SEED) {
        uint32_t rid = (uint32_t)pv[j];
        bool won = rs[j] == cid;
        if (won && is_taken(P.taken_other, rid)) { won = false; P.resv[rid] = RESV_LOST; }
        if (!won && pk == PK_MATCH && (pv[j] >> PK_ALT_SHIFT)) {  // the alternatives schedule: the second candidate, secured in pass 1 (k_ph_alt_resolve)
          const uint32_t alt = (uint32_t)(pv[j] >> PK_ALT_SHIFT) - 1u;
          if (P.resv[alt] == (ALT_KEY | cid)) {
            if (is_taken(P.taken_other, alt)) P.resv[alt] = RESV_LOST;
            else { won = true; rid = alt; }
          }
        }
        if (won) {
          atomicOr((unsigned long long *)&P.taken[rid >> 6], 1ull << (rid & 63));
          if (pk == PK_MATCH && rid >= P.seed_lo && rid < P.seed_hi) atomicSub(&P.ublk[rid >> UBLK_SHIFT], 1u);
          wonv = rid | (pk == PK_MATCH ? 0x80000000u : 0u);
        }
        if (pv[j] & PK_CURSOR_BIT) *P.cursor = (long long)(uint32_t)pv[j] - 1;  // every seed proposed this round ends up taken (by someone)
        needy = pk == PK_SEED && !won;
        if (pk == PK_SEED && won && cls[j] >= 0) cls[j] = 1;
      } else if (pk == PK_NONE) {
        needy = (pv[j] & PK_WILLNEED_BIT) != 0;
      } else if (pk == PK_NOSEED && (pv[j] & PK_CURSOR_BIT)) {
        *P.cursor = -1;
      }
      P.won[cid - P.gg0] = wonv;
    }
    const uint64_t nb = __ballot(needy);
    const uint32_t na = (uint32_t)__popcll(__ballot(alive && inr[j]));
    const uint32_t cb = cbase + 64 * j;  // (g0 is a multiple of 2048: a sub-block covers two whole bitmap words of its group)
    if (lane == 0 && cb < gend) {
      P.needy[cb >> 5] = (uint32_t)nb;
      if (cb + 32 < ((gend + 31) & ~31u)) P.needy[(cb >> 5) + 1] = (uint32_t)(nb >> 32);
      if (nb) atomicAdd(&P.needy_cnt_next[cb >> 11], (uint32_t)__popcll(nb));
      P.alive_wave[cb >> 6] = na;
    }
  }
  const uint32_t t = blockIdx.x * 64 + lane;  // (one thread per ...)
  for (uint32_t j = t; j < P.gKg_other; j += gridDim.x * 64) {  // the other group's last winners: into this group's view
    const uint32_t w = P.won_other[j];
    if (w == 0xffffffffu) continue;
    const uint32_t rid = w & 0x7fffffffu;
    atomicOr((unsigned long long *)&P.taken[rid >> 6], 1ull << (rid & 63));
    if ((w >> 31) && rid >= P.seed_lo && rid < P.seed_hi) atomicSub(&P.ublk[rid >> UBLK_SHIFT], 1u);
    mark_dead(P, rid);  // taken in both views from here on: its bin entries say so (entry_dead)
  }
  if (t == 0 && P.longq) { P.longq[0] = 0; P.longq[1] = 0; }  // this group's long searches of the round are done (as in k_mg_mark)
  if (P.longq && t < 2) P.lctl[t] = 0;
  if (t < P.nb_hi - P.nb_lo) P.needy_cnt[P.nb_lo + t] = 0;  // what this group's NEXT mark step accumulates into
  {  // class lists of this block's chains (k_round_mc; the order of k_mg_mark: class 0 first, chain ids ascending in a class)
    const uint32_t segi = P.gg0 / MARK_BLOCK + blockIdx.x;
    uint32_t tot[4], base = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t run = base;
#pragma unroll
      for (int j = 0; j < CPL; j++) {
        const uint64_t m = __ballot(cls[j] == k);
        if (cls[j] == k) P.ord[(size_t)segi * MARK_BLOCK + run + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = cbase + 64 * j + lane - P.c0;
        run += (uint32_t)__popcll(m);
      }
      tot[k] = run - base;
      base = run;
    }
    if (lane == 0) P.ord_cnt[segi] = make_uint4(tot[0], tot[1], tot[2], tot[3]);
  }
}
// The same step with four wavefronts per block of MARK_BLOCK chains and one chain per thread (k_mg_mark's shape): a quarter of
// the dependent steps per thread.  Beside the four-chain round kernel of a shallow pool (short-lived wavefronts, 5 per SIMD)
// its workgroups find their slots -- 12 us against 9 alone, and the mark step is on every round's critical path there
// (100 M x 150 bp: chains stage 335 ms with this kernel, 353 with the one above) --, so shallow pools run this one.
__global__ __launch_bounds__(256) void k_ph_mark_wide(DevParams P) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t cid = P.gg0 + t, gend = P.gg0 + P.gKg;  // (global chain ids, as in k_ph_mark)
  const int lane = threadIdx.x & 63;
  bool needy = false, alive = false;
  int cls = -1;
  // The step sits on its group's critical path (round kernel -> mark -> round kernel) and is pure latency: independent loads
  // are issued together -- the proposal word and this thread's entry of the other group's winners first, then the reservation
  // word and the other view's bitmap word of the proposed read (not one behind the other's verdict) -- 20 -> 13 us beside the
  // other group's round kernel
  unsigned long long pv = (unsigned long long)PK_DONE << 32;
  uint32_t wo = 0xffffffffu;
  if (cid < gend) pv = P.prop[cid];
  if (t < P.gKg_other) wo = P.won_other[t];
  const int pk = (int)(pv >> 32) & 7;
  uint32_t rs = 0xffffffffu;
  unsigned long long tw = 0;
  if (pk == PK_MATCH || pk == PK_SEED) {
    rs = P.resv[(uint32_t)pv];
    tw = P.taken_other[(uint32_t)pv >> 6];
  }
  if (cid < gend) {
    uint32_t wonv = 0xffffffffu;
    alive = pk != PK_DONE;
    if (alive && cid - P.c0 - P.g0 < P.Kg) cls = pk == PK_MATCH ? 2 : pk == PK_NONE ? ((pv & PK_WILLNEED_BIT) ? 3 : 0) : 3;
    if (pk == PK_MATCH || pk == PK_SEED) {
      uint32_t rid = (uint32_t)pv;
      bool won = rs == cid;
      if (won && ((tw >> (rid & 63)) & 1ull)) { won = false; P.resv[rid] = RESV_LOST; }
      if (!won && pk == PK_MATCH && (pv >> PK_ALT_SHIFT)) {  // the alternatives schedule: the second candidate, secured in pass 1 (k_ph_alt_resolve)
        const uint32_t alt = (uint32_t)(pv >> PK_ALT_SHIFT) - 1u;
        if (P.resv[alt] == (ALT_KEY | cid)) {
          if (is_taken(P.taken_other, alt)) P.resv[alt] = RESV_LOST;
          else { won = true; rid = alt; }
        }
      }
      if (won) {
        atomicOr((unsigned long long *)&P.taken[rid >> 6], 1ull << (rid & 63));
        if (pk == PK_MATCH && rid >= P.seed_lo && rid < P.seed_hi) atomicSub(&P.ublk[rid >> UBLK_SHIFT], 1u);
        wonv = rid | (pk == PK_MATCH ? 0x80000000u : 0u);
      }
      if (pv & PK_CURSOR_BIT) *P.cursor = (long long)(uint32_t)pv - 1;  // every seed proposed this round ends up taken (by someone)
      needy = pk == PK_SEED && !won;
      if (pk == PK_SEED && won && cls >= 0) cls = 1;
    } else if (pk == PK_NONE) {
      needy = (pv & PK_WILLNEED_BIT) != 0;
    } else if (pk == PK_NOSEED && (pv & PK_CURSOR_BIT)) {
      *P.cursor = -1;
    }
    P.won[t] = wonv;
  }
  for (uint32_t j = t; j < P.gKg_other; j += gridDim.x * blockDim.x) {  // the other group's last winners: into this group's view
    const uint32_t w = j == t ? wo : P.won_other[j];
    if (w == 0xffffffffu) continue;
    const uint32_t rid = w & 0x7fffffffu;
    atomicOr((unsigned long long *)&P.taken[rid >> 6], 1ull << (rid & 63));
    if ((w >> 31) && rid >= P.seed_lo && rid < P.seed_hi) atomicSub(&P.ublk[rid >> UBLK_SHIFT], 1u);
    mark_dead(P, rid);  // taken in both views from here on: its bin entries say so (entry_dead)
  }
  if (t == 0 && P.longq) { P.longq[0] = 0; P.longq[1] = 0; }  // this group's long searches of the round are done (as in k_mg_mark)
  if (P.longq && t < 2) P.lctl[t] = 0;
  const uint64_t nb = __ballot(needy);
  const uint32_t na = (uint32_t)__popcll(__ballot(alive));
  const uint32_t cb = cid & ~63u;  // (g0 is a multiple of 2048: a wavefront covers two whole bitmap words of its group)
  if (lane == 0 && cb < gend) {
    P.needy[cb >> 5] = (uint32_t)nb;
    if (cb + 32 < ((gend + 31) & ~31u)) P.needy[(cb >> 5) + 1] = (uint32_t)(nb >> 32);
    if (nb) atomicAdd(&P.needy_cnt_next[cid >> 11], (uint32_t)__popcll(nb));
    P.alive_wave[cid >> 6] = na;
  }
  if (t < P.nb_hi - P.nb_lo) P.needy_cnt[P.nb_lo + t] = 0;  // what this group's NEXT mark step accumulates into
  {  // class lists of this block's chains (k_round_mc; as in k_mg_mark)
    __shared__ uint32_t s_wc[4][4];  // [wave][class]
    const int wv = threadIdx.x >> 6;
    const uint32_t segi = P.gg0 / MARK_BLOCK + blockIdx.x;
    uint32_t mypos = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t m = __ballot(cls == k);
      if (lane == 0) s_wc[wv][k] = (uint32_t)__popcll(m);
      if (cls == k) mypos = (uint32_t)__popcll(m & ((1ull << lane) - 1));
    }
    __syncthreads();
    if (cls >= 0) {
      uint32_t base = 0;
      for (int k = 0; k < cls; k++) base += s_wc[0][k] + s_wc[1][k] + s_wc[2][k] + s_wc[3][k];
      for (int w = 0; w < wv; w++) base += s_wc[w][cls];
      P.ord[(size_t)segi * MARK_BLOCK + base + mypos] = cid - P.c0;
    }
    if (threadIdx.x == 0)
      P.ord_cnt[segi] = make_uint4(s_wc[0][0] + s_wc[1][0] + s_wc[2][0] + s_wc[3][0], s_wc[0][1] + s_wc[1][1] + s_wc[2][1] + s_wc[3][1],
                                   s_wc[0][2] + s_wc[1][2] + s_wc[2][2] + s_wc[3][2], s_wc[0][3] + s_wc[1][3] + s_wc[2][3] + s_wc[3][3]);
  }
}
// pass 1 of the alternatives schedule for one group (k_alt_resolve; before the group's k_ph_mark, behind the other group's last
// mark step): a chain that did not secure its first candidate -- it went to a lower chain id of the group, or to the
// other group since the search -- proposes its second one
__global__ void k_ph_alt_resolve(DevParams P) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.gKg) return;
  const uint32_t cid = P.gg0 + t;
  const unsigned long long pv = P.prop[cid];
  if (((int)(pv >> 32) & 7) != PK_MATCH || !(pv >> PK_ALT_SHIFT)) return;
  const uint32_t rid = (uint32_t)pv;
  if (P.resv[rid] == cid && !is_taken(P.taken_other, rid)) return;
  atomicMin(&P.resv[(uint32_t)(pv >> PK_ALT_SHIFT) - 1u], ALT_KEY | cid);
}
// waits on the device: the second group's first round starts half a round after the first group's
__global__ void k_delay(uint32_t us) {
  const uint64_t t0 = wall_clock64();  // 100 MHz
  while (wall_clock64() - t0 < (uint64_t)us * 100u) __builtin_amdgcn_s_sleep(64);
}
// first round: every local chain in class 2
// (over the global chains [gg0, gg0 + gKg) of the launch's group, gg0 a multiple of MARK_BLOCK; this rank's are [c0 + g0, c0 + g0 + Kg))
__global__ void k_init_ord(DevParams P) {
  const uint32_t cid = P.gg0 + blockIdx.x * blockDim.x + threadIdx.x;  // global chain id; block = the mark step's block
  const uint32_t blk = P.gg0 / MARK_BLOCK + blockIdx.x;
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const bool local = cid < P.gg0 + P.gKg && cid - P.c0 - P.g0 < P.Kg;
  const uint64_t m = __ballot(local);
  uint32_t wbase = 0;
  if ((threadIdx.x & 63) == 0 && m) wbase = atomicAdd(&s_n, (uint32_t)__popcll(m));
  wbase = (uint32_t)__shfl((int)wbase, 0, 64);
  if (local) P.ord[(size_t)blk * MARK_BLOCK + wbase + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1))] = cid - P.c0;
  __syncthreads();
  if (threadIdx.x == 0) P.ord_cnt[blk] = make_uint4(0u, 0u, s_n, 0u);
}

// per chain: {records emitted, singletons}; totals of the per-chain counters (finalize reads 8 bytes per chain
// instead of the 384-byte chain records)
__global__ void k_chain_summary(DevParams P, uint2 *__restrict__ sum, unsigned long long *__restrict__ tot /* [8] */) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < P.K) {
    const Chain &c = P.chains[i];
    sum[i] = make_uint2(c.h.n_emit, c.h.n_single);
    v[0] = c.n_unmatched; v[1] = c.st_probes; v[2] = c.st_keyok; v[3] = c.st_cands; v[4] = c.st_iter; v[5] = c.st_lost; v[6] = c.st_hits; v[7] = c.st_long;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint64_t w = wave_sum_u64(v[k]);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&tot[k], (unsigned long long)w);
  }
}

// ------------------------------------------------------------ K7 finalize / emit
// slot i of the append buffers belongs to chunk i / CHUNK; its record is number first_seq + i % CHUNK of the owning chain
__global__ void k_scatter_matched(DevParams P, uint64_t cap, const uint64_t *__restrict__ off_m) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  const uint2 ci = P.e_chunk[i / CHUNK];
  if (ci.x == 0xffffffffu) return;  // chunk never handed out
  const uint32_t seq = ci.y + (uint32_t)(i % CHUNK);
  if (seq >= P.chains[ci.x].h.n_emit) return;  // unused slot of the chain's last chunk
  const uint4 r = P.e_rec[i];
  const uint64_t d = off_m[ci.x] + seq;
  P.f_order[d] = r.x; P.f_rc[d] = (char)(r.y & 0xffu); P.f_flag[d] = (char)((r.y >> 8) & 0xffu);
  P.f_pos[d] = (long long)((unsigned long long)r.z | ((unsigned long long)r.w << 32));
  P.f_len[d] = P.uniform_len ? (uint16_t)P.L : P.lens[r.x];
}
__global__ void k_scatter_single(DevParams P, uint64_t cap, const uint64_t *__restrict__ off_s) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  const uint2 ci = P.s_chunk[i / CHUNK];
  if (ci.x == 0xffffffffu) return;
  const uint32_t seq = ci.y + (uint32_t)(i % CHUNK);
  if (seq >= P.chains[ci.x].h.n_single) return;
  P.f_order_s[off_s[ci.x] + seq] = P.s_rec[i];
}

// record sizes of a temp.dna stream (writetofile, reorder.h:667-687)
__global__ void k_rec_size(const uint32_t *__restrict__ order, const uint16_t *__restrict__ lens, uint64_t cnt,
                           uint32_t *__restrict__ sz) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt) sz[i] = 2u + ((uint32_t)lens[order[i]] + 3u) / 4u;
}
// 'd': u16 len + raw bytes; 'r': reverse complement re-packed (util.cpp:269-294, :376-381)
__global__ void k_emit_dna(const uint64_t *__restrict__ reads, const uint16_t *__restrict__ lens, int S,
                           const uint32_t *__restrict__ order, const char *__restrict__ rc, uint64_t cnt,
                           const uint64_t *__restrict__ off, uint32_t rec_fixed, uint8_t *__restrict__ dst) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const uint32_t rid = order[i];
  const uint32_t len = lens[rid];
  const uint64_t *r = reads + (uint64_t)rid * S;
  uint8_t *o = dst + (off ? off[i] : i * (uint64_t)rec_fixed);
  o[0] = (uint8_t)(len & 0xff); o[1] = (uint8_t)(len >> 8);
  const uint32_t nb = (len + 3) / 4;
  const bool rev = rc && rc[i] == 'r';
  for (uint32_t b = 0; b < nb; b++) {
    uint32_t v = 0;
    if (!rev) v = (uint32_t)(r[b >> 3] >> (8 * (b & 7))) & 0xffu;
    else {
      for (uint32_t q = 0; q < 4; q++) {
        uint32_t j = 4 * b + q;
        if (j < len) {
          uint32_t s = len - 1 - j;
          uint32_t code = (uint32_t)(r[s >> 5] >> (2 * (s & 31))) & 3u;
          v |= (3u - code) << (2 * q);
        }
      }
    }
    o[2 + b] = (uint8_t)v;
  }
}

// The same records for fixed-length pools (every record rec bytes, no offsets), one thread per 32-bit WORD of the output:
// coalesced 4-byte stores instead of 2 + ceil(L/4) single-byte stores per lane at a stride of a record -- the byte-wise kernel
// took 105 ms for the 3.9 GB of temp.dna.* of 100 M reads, all of it in front of the first device-to-host copy of the
// drop-in's output leg (profiles/r05_files.txt).
__device__ __forceinline__ uint32_t emit_dna_byte(const uint64_t *__restrict__ r, uint32_t len, bool rev, uint32_t o) {
  if (o == 0) return len & 0xffu;
  if (o == 1) return len >> 8;
  const uint32_t b = o - 2;
  if (!rev) return (uint32_t)(r[b >> 3] >> (8 * (b & 7))) & 0xffu;
  // bases 4b .. 4b+3 of the reverse complement = complement of source bases len-1-4b downwards (util.cpp:376-381)
  const int lo = (int)len - 4 - 4 * (int)b;  // source base of q = 3; negative: that many of the byte's last bases lie past the read
  uint32_t w8;
  if (lo >= 0) {
    const int p = 2 * lo, li = p >> 6, sh = p & 63;
    uint64_t v = r[li] >> sh;
    if (sh > 56) v |= r[li + 1] << (64 - sh);  // (sh > 56 implies bits of the next limb: li + 1 < W because 2 lo + 8 <= 2 len)
    w8 = (uint32_t)v & 0xffu;
  } else {
    w8 = ((uint32_t)r[0] << (-2 * lo)) & 0xffu;
  }
  const uint32_t x = ~w8 & 0xffu;  // complement, then the four bases in reverse order
  uint32_t v = ((x & 3u) << 6) | ((x & 0xcu) << 2) | ((x >> 2) & 0xcu) | (x >> 6);
  if (lo < 0) v &= 0xffu >> (-2 * lo);  // bases past the read's end stay 0 (write_dna_in_bits pads with zeros)
  return v;
}
__global__ void k_emit_dna_fixed(const uint64_t *__restrict__ reads, const uint16_t *__restrict__ lens, int S,
                                 const uint32_t *__restrict__ order, const char *__restrict__ rc, uint64_t cnt,
                                 uint32_t rec, uint8_t *__restrict__ dst) {
  const uint64_t W = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, total = cnt * rec;
  if (4 * W >= total) return;
  uint64_t i = 4 * W / rec;
  uint32_t o = (uint32_t)(4 * W - i * rec);
  uint32_t word = 0;
  const uint32_t nbytes = (uint32_t)(total - 4 * W < 4 ? total - 4 * W : 4);
  uint64_t ci = ~0ull;
  const uint64_t *r = nullptr;
  uint32_t len = 0;
  bool rev = false;
  for (uint32_t k = 0; k < nbytes; k++) {
    if (i != ci) {
      const uint32_t rid = order[i];
      len = lens[rid];
      r = reads + (uint64_t)rid * S;
      rev = rc && rc[i] == 'r';
      ci = i;
    }
    word |= emit_dna_byte(r, len, rev, o) << (8 * k);
    if (++o == rec) { o = 0; i++; }
  }
  if (nbytes == 4) *reinterpret_cast<uint32_t *>(dst + 4 * W) = word;
  else for (uint32_t k = 0; k < nbytes; k++) dst[4 * W + k] = (uint8_t)(word >> (8 * k));
}

// ------------------------------------------------------------- synthetic reads
__global__ void k_synth(uint8_t *__restrict__ dst, uint32_t n, uint32_t L, uint64_t G, uint64_t seed,
                        uint32_t thr24) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t rec = 2u + (L + 3) / 4;
  uint8_t *o = dst + i * rec;
  uint64_t pos; uint32_t rc;
  syn_read_params(seed, G, L, i, (thr24 & SYN_PAIRED_FLAG) ? n / 2 : 0, &pos, &rc);
  o[0] = (uint8_t)(L & 0xff); o[1] = (uint8_t)(L >> 8);
  for (uint32_t b = 0; b < (L + 3) / 4; b++) {
    uint32_t v = 0;
    for (uint32_t q = 0; q < 4; q++) {
      uint32_t j = 4 * b + q;
      if (j < L) v |= syn_nat_to_spring(syn_read_base(seed, G, L, thr24, i, j, pos, rc)) << (2 * q);
    }
    o[2 + b] = (uint8_t)v;
  }
}

// ================================================================== launchers
#define GRID1(n, bs) dim3((unsigned)(((uint64_t)(n) + (bs) - 1) / (bs)))

void launch_unpack(hipStream_t st, const uint8_t *dna, const uint64_t *off, uint32_t n, int L, int W, int S,
                   uint32_t rec_fixed, uint64_t *reads, uint16_t *lens, uint32_t *bad_len) {
  if (!n) return;
  uint64_t tot = (uint64_t)n * S;
  hipLaunchKernelGGL(k_unpack, GRID1(tot, 256), dim3(256), 0, st, dna, off, n, L, W, S, rec_fixed, reads, lens, bad_len);
}
void launch_flag_in_dict(hipStream_t st, const uint16_t *lens, uint32_t n, int dend, uint32_t *flag) {
  hipLaunchKernelGGL(k_flag_in_dict, GRID1(n, 256), dim3(256), 0, st, lens, n, dend, flag);
}
void launch_keys(hipStream_t st, const uint64_t *reads, const uint16_t *lens, const uint32_t *slot, uint32_t n,
                 int S, int dstart, int dend, uint64_t *keys, uint32_t *vals) {
  hipLaunchKernelGGL(k_keys, GRID1(n, 256), dim3(256), 0, st, reads, lens, slot, n, S, dstart, dend, keys, vals);
}
void launch_tab_insert(hipStream_t st, const uint64_t *mhash, const uint64_t *mval, uint64_t nmerged, DictBuild d0,
                       DictBuild d1, uint32_t *fpt, int bshift) {
  if (!nmerged) return;
  hipLaunchKernelGGL(k_tab_insert<false>, GRID1(nmerged, 256), dim3(256), 0, st, mhash, mval, nmerged, d0, d1, fpt, bshift);
  hipLaunchKernelGGL(k_tab_insert<true>, GRID1(nmerged, 256), dim3(256), 0, st, mhash, mval, nmerged, d0, d1, fpt, bshift);
}
void launch_minz_prepare(hipStream_t st, const uint64_t *mhash, const uint64_t *mval, uint64_t nmerged, DictBuild d0,
                         DictBuild d1, int lshift, uint32_t *bucket, uint64_t *tagpay) {
  if (!nmerged) return;
  hipLaunchKernelGGL(k_minz_prepare, GRID1(nmerged, 256), dim3(256), 0, st, mhash, mval, nmerged, d0, d1, lshift, bucket, tagpay);
}
void launch_tab_insert_minz(hipStream_t st, const uint32_t *bucket_sorted, const uint64_t *tagpay_sorted, uint64_t nmerged,
                            uint32_t *fpt, int bshift, uint32_t *marked) {
  if (!nmerged) return;
  hipLaunchKernelGGL(k_tab_insert_minz<false>, GRID1(nmerged, 256), dim3(256), 0, st, bucket_sorted, tagpay_sorted, nmerged, fpt, bshift, marked);
  hipLaunchKernelGGL(k_tab_insert_minz<true>, GRID1(nmerged, 256), dim3(256), 0, st, bucket_sorted, tagpay_sorted, nmerged, fpt, bshift, marked);
}
void launch_iota_tag(hipStream_t st, uint64_t *v, uint64_t n, uint64_t tag) {
  if (!n) return;
  hipLaunchKernelGGL(k_iota_tag, GRID1(n, 256), dim3(256), 0, st, v, n, tag);
}
void launch_trim_bins(hipStream_t st, const uint32_t *deep, const uint32_t *ndeep, uint32_t ndeep_host,
                      ulonglong2 *urec, const uint32_t *ids, const uint64_t *taken, ulonglong2 *sig, uint32_t *epos) {
  if (!ndeep_host) return;
  hipLaunchKernelGGL(k_trim_bins, GRID1(ndeep_host, 4), dim3(256), 0, st, deep, ndeep, urec, const_cast<uint32_t *>(ids), taken, sig, epos);
}
void launch_dict_lookup(hipStream_t st, TabView tab, const ulonglong2 *urec, int which,
                        const uint64_t *reads, int S, int dstart, int dend, const uint64_t *keys, uint32_t nkeys,
                        uint32_t *start, uint32_t *count) {
  if (!nkeys) return;
  hipLaunchKernelGGL(k_dict_lookup, GRID1(nkeys, 256), dim3(256), 0, st, tab, urec, which, reads, S, dstart,
                     2 * (dend - dstart + 1), keys, nkeys, start, count);
}
void launch_fill_u32(hipStream_t st, uint32_t *p, uint64_t n, uint32_t v) {
  if (!n) return;
  hipLaunchKernelGGL(k_fill_u32, GRID1(n, 256), dim3(256), 0, st, p, n, v);
}
// test hook (spring_reorder_debug_check_seed_state): find_seed relies on (1) every read above the cursor being taken and
// (2) ublk[b] being the exact number of untaken reads of every block b below the cursor's block.  bad[0] / bad[1]
// count the violations between two rounds.
__global__ void k_check_seed_state(DevParams P, uint64_t nwords, unsigned long long *bad) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  const long long top = *P.cursor;
  const uint64_t v = P.taken[w];
  // (1) bits of reads > cursor
  const long long first = (long long)w * 64;
  if (first + 63 > top) {
    const int lo = top < first ? 0 : (int)(top - first + 1);
    const uint64_t must = lo >= 64 ? 0ull : (~0ull << lo);
    if ((v & must) != must) atomicAdd(&bad[0], 1ull);
  }
  // (2) one thread per block: the first word of the block
  constexpr int WPB_ = 1 << (UBLK_SHIFT - 6);
  if ((w % WPB_) == 0 && top >= 0 && (long long)(w / WPB_) < (top >> UBLK_SHIFT)) {
    int cnt = 0;
    for (int k = 0; k < WPB_; k++) cnt += __popcll(~P.taken[w + k]);
    if ((uint32_t)cnt != P.ublk[w / WPB_]) atomicAdd(&bad[1], 1ull);
  }
}
void launch_check_seed_state(hipStream_t st, const DevParams &P, uint64_t nwords, unsigned long long *bad) {
  if (!nwords) return;
  hipLaunchKernelGGL(k_check_seed_state, GRID1(nwords, 256), dim3(256), 0, st, P, nwords, bad);
}
void launch_init_taken(hipStream_t st, uint64_t *taken, uint64_t nwords, uint32_t n, uint32_t *ublk) {
  if (!nwords) return;
  hipLaunchKernelGGL(k_init_taken, GRID1(nwords, 256), dim3(256), 0, st, taken, nwords, n, ublk);
}
#define NP_DISPATCH(CALL)                                  \
  do {                                                     \
    const int np_ = P.Lpad / 64;                           \
    if (np_ <= 1) { CALL(1); } else if (np_ == 2) { CALL(2); } else if (np_ == 3) { CALL(3); } \
    else if (np_ == 4) { CALL(4); } else { CALL(8); }      \
  } while (0)

// seeds: once per context (every chain of the pool); the rest for the local chains [g0, g0 + Kg) of the group [gg0, gg0 + gKg)
void launch_init_chains(hipStream_t st, const DevParams &P, bool seeds) {
  if (seeds && P.Ktot) hipLaunchKernelGGL(k_init_seeds, GRID1(P.Ktot, 256), dim3(256), 0, st, P);
  if (!P.Kg) return;
  if (P.ord) hipLaunchKernelGGL(k_init_ord, GRID1(P.gKg, MARK_BLOCK), dim3(MARK_BLOCK), 0, st, P);
#define CALL(N) hipLaunchKernelGGL(k_init_chains<N>, dim3((P.Kg + 3) / 4), dim3(256), 0, st, P)
  NP_DISPATCH(CALL);
#undef CALL
}
template <int WPB>
static void launch_search_wpb(hipStream_t st, const DevParams &P, bool stats) {
  const dim3 g((P.K + WPB - 1) / WPB), b(64 * WPB);
  const size_t dyn = (size_t)P.dbg_search_lds;  // occupancy experiment: dummy dynamic LDS per block
  if (stats) hipLaunchKernelGGL((k_search<true, WPB>), g, b, dyn, st, P);
  else hipLaunchKernelGGL((k_search<false, WPB>), g, b, dyn, st, P);
}
void launch_search(hipStream_t st, const DevParams &P, bool stats) {
  if (!P.K) return;
#ifdef SR_DEV_PROD_ONLY  // tools/xbuild.sh: experiment builds hold the production k_round only
  (void)st; (void)stats; abort();
#else
  if (P.search_wpb == 1) launch_search_wpb<1>(st, P, stats);
  else if (P.search_wpb == 2) launch_search_wpb<2>(st, P, stats);
  else launch_search_wpb<4>(st, P, stats);
#endif
}
void launch_apply(hipStream_t st, const DevParams &P, bool literal) {
  if (!P.K) return;
#ifdef SR_DEV_PROD_ONLY
  (void)st; (void)literal; abort();
#else
  const dim3 g((P.K + 3) / 4), b(256);
  if (literal) { hipLaunchKernelGGL((k_apply<8, true>), g, b, 0, st, P); return; }
#define CALL(N) hipLaunchKernelGGL((k_apply<N, false>), g, b, (size_t)P.dbg_apply_lds, st, P)  // dbg: occupancy experiment
  NP_DISPATCH(CALL);
#undef CALL
#endif
}
// fused round: one wavefront (= one block) per chain; NP = 3 covers reads up to 192 bases, 8 the rest
void launch_round(hipStream_t st, const DevParams &P, bool stats, bool mg) {
  if (!P.K || !P.Kg) return;
  const dim3 g(P.Kg), b(64);
  const size_t dyn = (size_t)P.dbg_search_lds;
  // four chains per wavefront (k_round_mc) unless the run needs what only the one-chain kernel has: the
  // reference-equivalent work counters, or the deep-bin machinery (tail trimming, balanced scan, resumed searches)
  if (P.mc && !stats && !P.deep_bins) {
    // one class-list segment per block of MARK_BLOCK chain ids that holds local chains, a fixed number of wavefronts each
    // (two-group schedule: the launch covers one group, chains [g0, g0 + Kg); else g0 = 0, Kg = K)
    const uint32_t nseg = (P.c0 + P.g0 + P.Kg + MARK_BLOCK - 1) / MARK_BLOCK - (P.c0 + P.g0) / MARK_BLOCK;
    const dim3 g4(nseg * MC_WAVES_PER_BLOCK);
    if (P.Lpad <= 192) {
      // (P.ka: the chains remember which windows of their consensus are known absent -- search_ka, reorder_round_mc.h)
      if (P.ka) {
        if (mg) hipLaunchKernelGGL((k_round_mc<3, true, true>), g4, b, 0, st, P);
        else hipLaunchKernelGGL((k_round_mc<3, false, true>), g4, b, 0, st, P);
      } else if (mg) hipLaunchKernelGGL((k_round_mc<3, true, false>), g4, b, 0, st, P);
      else hipLaunchKernelGGL((k_round_mc<3, false, false>), g4, b, 0, st, P);
    } else {
      if (mg) hipLaunchKernelGGL((k_round_mc_long<true>), g4, b, 0, st, P);
      else hipLaunchKernelGGL((k_round_mc_long<false>), g4, b, 0, st, P);
    }
    return;
  }
#ifdef SR_DEV_PROD_ONLY
  if (stats || mg || P.deep_bins || P.Lpad > 192) abort();
  hipLaunchKernelGGL((k_round<3, false, false, false>), g, b, dyn, st, P);
#else
#define RCALL2(N, KERN)                                                                      \
  do {                                                                                       \
    if (mg) { if (stats) hipLaunchKernelGGL((KERN<N, true, true>), g, b, dyn, st, P);        \
              else hipLaunchKernelGGL((KERN<N, false, true>), g, b, dyn, st, P); }           \
    else { if (stats) hipLaunchKernelGGL((KERN<N, true, false>), g, b, dyn, st, P);          \
           else hipLaunchKernelGGL((KERN<N, false, false>), g, b, dyn, st, P); }             \
  } while (0)
#define RCALL(N) do { if (P.deep_bins) RCALL2(N, k_round_t); else RCALL2(N, k_round_nt); } while (0)
  if (P.deep_bins && !stats && P.long_budget > 0 && P.longq) {
    // pools of very deep bins: the variant that hands long searches over, then k_long for them (a fixed grid, each
    // block takes queue entries in turn)
#define LCALL(N) do { if (mg) hipLaunchKernelGGL((k_round_tl<N, true>), g, b, dyn, st, P); \
                      else hipLaunchKernelGGL((k_round_tl<N, false>), g, b, dyn, st, P); } while (0)
    if (P.Lpad <= 192) LCALL(3); else LCALL(8);
#undef LCALL
    // the queued searches: probes and bin lists, then every part of every search, then the proposals (fixed grids: how many
    // searches a round hands over is known on the device only; the blocks take queue entries / parts in turn)
    // (long_blocks = 256 = the CUs: what is resident of each kernel -- 3 / 1 list blocks, 5 scan blocks per CU -- is its grid)
    const uint32_t lb = (uint32_t)P.long_blocks;
    if (4 * P.maxshift <= 512) hipLaunchKernelGGL(k_long_list<8>, dim3(std::min<uint32_t>(P.K, 3 * lb)), dim3(512), 0, st, P);
    else hipLaunchKernelGGL(k_long_list<LONG_WAVES>, dim3(std::min<uint32_t>(P.K, lb)), dim3(64 * LONG_WAVES), 0, st, P);
    hipLaunchKernelGGL(k_long_scan, dim3(std::min<uint32_t>(P.K * 4, 5 * lb)), dim3(64 * SCAN_WAVES), 0, st, P);
    hipLaunchKernelGGL(k_long_fin, dim3(std::min<uint32_t>((P.K + 3) / 4, lb)), dim3(256), 0, st, P, 1);  // (a wavefront per search: 4 lb in flight)
  } else if (P.Lpad <= 192) RCALL(3); else RCALL(8);
#undef RCALL2
#undef RCALL
#endif
}
void launch_mg_resolve(hipStream_t st, const DevParams &P) {
  if (P.gKg == P.Kg) return;  // one rank: nothing foreign
  hipLaunchKernelGGL(k_mg_resolve, GRID1(P.gKg - P.Kg, 256), dim3(256), 0, st, P);
}
void launch_mg_mark(hipStream_t st, const DevParams &P) {
  if (P.alts == 2) hipLaunchKernelGGL(k_alt_resolve, GRID1(P.Ktot, 256), dim3(256), 0, st, P);
  hipLaunchKernelGGL(k_mg_mark, GRID1(P.Ktot, 256), dim3(256), 0, st, P);
}
void launch_ph_mark(hipStream_t st, const DevParams &P) {
  if (!P.gKg) return;
  if (P.alts == 2) hipLaunchKernelGGL(k_ph_alt_resolve, GRID1(P.gKg, 64), dim3(64), 0, st, P);   // (one-wavefront workgroups: see k_ph_mark)
  if (P.deep_bins) hipLaunchKernelGGL(k_ph_mark, GRID1(P.gKg, MARK_BLOCK), dim3(64), 0, st, P);
  else hipLaunchKernelGGL(k_ph_mark_wide, GRID1(P.gKg, MARK_BLOCK), dim3(MARK_BLOCK), 0, st, P);
}
void launch_delay(hipStream_t st, uint32_t microseconds) { hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, st, microseconds); }
void launch_chain_summary(hipStream_t st, const DevParams &P, uint2 *sum, unsigned long long *tot) {
  if (P.K) hipLaunchKernelGGL(k_chain_summary, GRID1(P.K, 256), dim3(256), 0, st, P, sum, tot);
}
void launch_scatter(hipStream_t st, const DevParams &P, uint64_t cap_m, uint64_t cap_s, const uint64_t *off_m,
                    const uint64_t *off_s) {
  if (cap_m) hipLaunchKernelGGL(k_scatter_matched, GRID1(cap_m, 256), dim3(256), 0, st, P, cap_m, off_m);
  if (cap_s) hipLaunchKernelGGL(k_scatter_single, GRID1(cap_s, 256), dim3(256), 0, st, P, cap_s, off_s);
}
void launch_rec_size(hipStream_t st, const uint32_t *order, const uint16_t *lens, uint64_t cnt, uint32_t *sz) {
  if (!cnt) return;
  hipLaunchKernelGGL(k_rec_size, GRID1(cnt, 256), dim3(256), 0, st, order, lens, cnt, sz);
}
void launch_emit_dna(hipStream_t st, const uint64_t *reads, const uint16_t *lens, int S, const uint32_t *order,
                     const char *rc, uint64_t cnt, const uint64_t *off, uint32_t rec_fixed, uint8_t *dst) {
  if (!cnt) return;
  if (!off) {  // fixed-length pool: every record rec_fixed bytes
    hipLaunchKernelGGL(k_emit_dna_fixed, GRID1((cnt * rec_fixed + 3) / 4, 256), dim3(256), 0, st, reads, lens, S, order, rc, cnt, rec_fixed, dst);
    return;
  }
  hipLaunchKernelGGL(k_emit_dna, GRID1(cnt, 256), dim3(256), 0, st, reads, lens, S, order, rc, cnt, off, rec_fixed, dst);
}
void launch_synth(hipStream_t st, uint8_t *dst, uint32_t n, uint32_t L, uint64_t G, uint64_t seed, uint32_t thr24) {
  if (!n) return;
  hipLaunchKernelGGL(k_synth, GRID1(n, 256), dim3(256), 0, st, dst, n, L, G, seed, thr24);
}

// ------------------------------------------------- rocPRIM plumbing (sort / RLE / scan)
hipError_t sort_pairs(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint64_t *kin, uint64_t *kout,
                      const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0u, end_bit, st);
}
hipError_t sort_pairs_u32_u64(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint64_t *vin, uint64_t *vout, size_t n, unsigned end_bit) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0u, end_bit, st);
}
hipError_t rle(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint64_t *in, size_t n, uint64_t *uniq,
               uint32_t *counts, uint32_t *nruns) {
  return rocprim::run_length_encode(tmp, tmp_bytes, in, n, uniq, counts, nruns, st);
}
hipError_t merge_by_hash(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint64_t *k0, const uint64_t *k1,
                         const uint64_t *v0, const uint64_t *v1, uint64_t *kout, uint64_t *vout, size_t n0, size_t n1) {
  return rocprim::merge(tmp, tmp_bytes, k0, k1, kout, v0, v1, vout, n0, n1, rocprim::less<uint64_t>(), st);
}
hipError_t excl_scan_u32(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint32_t *out, size_t n) {
  return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), st);
}
hipError_t excl_scan_u32_to_u64(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint64_t *out,
                                size_t n) {
  return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), st);
}

}  // namespace sr
