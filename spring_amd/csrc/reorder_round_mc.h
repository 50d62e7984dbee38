// spring_amd/csrc/reorder_round_mc.h -- included by reorder_kernels.hip (inside namespace sr, after the probe helpers).
//
// k_round_mc: the fused round (phase B of round t-1 + phase A of round t, see k_round) with FOUR chains per
// wavefront: chain = one 16-lane DPP row.  Same schedule, same results, other mapping of work to lanes.
//
// Why: with one chain per wavefront almost every instruction of a step is wave-uniform bookkeeping or runs with a
// few active lanes (a candidate compare: one lane; a first probe batch: 32), and the round kernel sits at 64 VGPRs
// x 8 waves per SIMD with its vector ALU ~65 % busy and its memory steps queueing behind each other (DESIGN.md
// section 6).  Here every instruction serves four chains, the kernel runs 5 waves per SIMD with a 96-VGPR budget
// (20 chains per SIMD in flight instead of 8) and a lane keeps several independent table fetches in flight.
// The code is plain SIMT over a 16-lane group: values that are uniform per chain are ordinary per-lane variables,
// branches on them are group-uniform, cross-lane traffic is width-16 shuffles / ballots shifted to the row.
//
// Scope: shallow dictionaries (no tail trimming / balanced scan), no reference-equivalent work counters; the
// one-chain-per-wavefront k_round keeps those variants (deep-coverage pools, collect_stats).
#ifndef SPRING_REORDER_ROUND_MC_H_
#define SPRING_REORDER_ROUND_MC_H_

namespace mc {

constexpr int G = 16;        // lanes per chain
constexpr int CPW = 64 / G;  // chains per wavefront

// per chain; sized by the instantiation (NQ = 3: reads up to 192 bases, 6 limbs, shifts up to 96 -> 4 pad limbs) so that
// the block stays inside 6 LDS granules of 1280 bytes = 20 blocks per CU at 5 waves per SIMD with the minimizer array
template <int NQ, bool KA>
struct GLds {
  static constexpr int WMAX = NQ <= 3 ? 6 : 16;      // limbs of a read
  static constexpr int PAD = NQ <= 3 ? 4 : LDS_PAD;  // zero limbs either side of ref / revref (lds_window)
  static constexpr int LIMBS = WMAX + 2 * PAD;
  static constexpr int MZ = NQ <= 3 && !KA ? 164 : 4;  // window minimizers (TabView::minz: reads up to 192 bases, not with the known-absent masks)
  uint64_t refs[2][LIMBS];          // ref / revref
  uint64_t rd[WMAX + 2];            // the read being merged, one zero limb either side
  uint64_t nref[WMAX];              // the speculative update's consensus, 4 bases per byte (= the limb format)
  union {
    uint8_t pres[128];              // tail_mc: "the other dictionary may hold this window", by probe code
    // search_ka (k_round_mc<.., KA = true>): known-absent windows, the layout of refs -- bit 2 o + l of ka[s] says "the window
    // at offset o of ref (s = 0) / revref (s = 1) is absent from dictionary l".  The table never changes, so the bit
    // holds for as long as the bases under the window do: commit_consensus_ka moves it with them.
    uint64_t ka[2][KA ? LIMBS : 2];
  };
  uint32_t best;                    // lowest priority code that has hit in the running batch (eval_probe)
  uint32_t pad[3];
  uint32_t mz[MZ];                  // mz[w] = minz value of the consensus window at offset w (minz_mc)
};

__device__ __forceinline__ int gmin_i(int v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, G));
  return v;
}
__device__ __forceinline__ int gsum_i(int v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}
__device__ __forceinline__ int gincl_scan_i(int v, int gl) {
#pragma unroll
  for (int o = 1; o < G; o <<= 1) {
    const int t = __shfl_up(v, o, G);
    if (gl >= o) v += t;
  }
  return v;
}
// ballot over the 16 lanes of this lane's group
__device__ __forceinline__ uint32_t gballot(bool p, int lane) {
  return (uint32_t)(__ballot(p) >> (lane & ~(G - 1))) & 0xffffu;
}
__device__ __forceinline__ uint64_t gshfl_u64(uint64_t v, int src) {
  const uint32_t lo = __shfl((uint32_t)v, src, G), hi = __shfl((uint32_t)(v >> 32), src, G);
  return ((uint64_t)hi << 32) | lo;
}

// 8 bits at bit `bitpos` of the staged read (S.rd, limb 0 at index 1; bitpos >= -64)
__device__ __forceinline__ uint32_t rd_bits8(const uint64_t *rd, int bitpos) {
  const int bp = bitpos + 64, li = bp >> 6, off = bp & 63;
  uint64_t v = rd[li] >> off;
  if (off > 56) v |= rd[li + 1] << (64 - off);
  return (uint32_t)v & 0xffu;
}

typedef uint32_t u32x4a_t __attribute__((ext_vector_type(4), aligned(4)));

// updaterefcount (reorder.h:110-220) for one chain by its 16 lanes; see wave_update_compute for the case analysis.
// Lane gl owns the position quads q = gl + 16 k (positions 4q .. 4q+3).  Fast path (byte counts in, byte counts
// out, no in-place aliasing): one 16-byte load (at the shifted source position), four packed-byte additions, one
// 16-byte store per quad; the new consensus goes to S.nref four bases per byte -- the limb format itself.
// Anything else (wide counts, the aliasing case of the reference) takes the generic per-position path.
template <int NQ, bool KA>
__device__ __forceinline__ int update_mc(const DevParams &P, uint32_t li, GLds<NQ, KA> &S, uint32_t rid, int n, bool reset,
                                         bool rev, int shift, int R, int cb, bool cur_wide, bool out_wide,
                                         bool &overflow, int gl, int *o_src = nullptr, int *o_cpy = nullptr, bool *o_alias = nullptr) {
  const int M = P.L, W = P.W;
  const int4 *__restrict__ cur = P.cnt + ((uint64_t)li * 2 + cb) * P.Lpad;
  int4 *__restrict__ nxt = P.cnt + ((uint64_t)li * 2 + (cb ^ 1)) * P.Lpad;
  const uint32_t *__restrict__ cur8 = P.cnt8 + ((uint64_t)li * 2 + cb) * P.Lpad;
  uint32_t *__restrict__ nxt8 = P.cnt8 + ((uint64_t)li * 2 + (cb ^ 1)) * P.Lpad;
  int hiP, cpy_hi, src_off, add_lo, add_hi, Rn, d = 0;
  bool alias = false;
  if (reset) { hiP = M; cpy_hi = 0; src_off = 0; add_lo = 0; add_hi = n; Rn = n; }
  else if (!rev) { Rn = max(R - shift, n); hiP = Rn; cpy_hi = R - shift; src_off = shift; add_lo = 0; add_hi = n; }
  else if (n - shift >= R) { Rn = n; hiP = n; cpy_hi = R; src_off = 0; add_lo = 0; add_hi = n; d = n - shift - R; alias = d > 0; }
  else if (R + shift <= M) { Rn = R + shift; hiP = Rn; cpy_hi = R; src_off = 0; add_lo = R - n + shift; add_hi = Rn; }
  else { Rn = M; hiP = M; cpy_hi = M - shift; src_off = R + shift - M; add_lo = M - n; add_hi = M; }
  const bool fast = !alias && !cur_wide && !out_wide;
  if (o_src) { *o_src = src_off; *o_cpy = cpy_hi; *o_alias = alias; }  // new position p < cpy_hi = old position p + src_off

  // issue the loads first: the read's limbs and (fast path) this lane's source quads
  const uint64_t myl = gl < W ? P.reads[(uint64_t)rid * P.S + gl] : 0ull;
  u32x4a_t old[NQ];
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const int p0 = 4 * (gl + G * k);
    old[k] = (u32x4a_t)(0u);
    if (fast && p0 < cpy_hi) old[k] = *reinterpret_cast<const u32x4a_t *>(cur8 + p0 + src_off);
  }
  if (gl < GLds<NQ, KA>::WMAX) S.rd[1 + gl] = myl;
  wave_sync();
  const uint64_t *rd = S.rd;

  bool ovf = false;
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const int q = gl + G * k, p0 = 4 * q;
    if (p0 >= P.Lpad) continue;
    uint32_t codes = 0;  // four 2-bit consensus codes
    if (fast) {
      // the four bases this quad adds, 2 bits each, element 0 in bits 0-1 (only meaningful inside [add_lo, add_hi))
      const int i0 = p0 - add_lo;
      uint32_t b8 = 0;
      if (i0 > -4 && i0 < n) {
        if (!rev) b8 = rd_bits8(rd, 2 * i0);
        else {
          const uint32_t x = ~rd_bits8(rd, 2 * (n - 4 - i0)) & 0xffu;  // complement; then reverse the four groups
          b8 = ((x & 3u) << 6) | ((x & 0xcu) << 2) | ((x >> 2) & 0xcu) | (x >> 6);
        }
      }
      u32x4a_t t4;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int p = p0 + e;
        uint32_t t = p < cpy_hi ? old[k][e] : 0u;
        const uint32_t code = (b8 >> (2 * e)) & 3u;
        uint32_t cdx = code;
        if (p >= add_lo && p < add_hi) {
          const uint32_t sh = (0x10081800u >> (8 * code)) & 31u;  // SPRING code A0 G1 C2 T3 -> count byte A0 C1 T2 G3
          ovf = ovf || ((t >> sh) & 255u) == 255u;
          t += 1u << sh;
        }
        if (p >= hiP) t = 0u;
        t4[e] = t;
        if (!reset) {  // reorder.h:204-212: strict >, A,C,T,G order (the earlier row wins a tie; no count: A)
          const uint32_t ka = ((t & 255u) << 2) | 3u, kc = (((t >> 8) & 255u) << 2) | 2u,
                         kt = (((t >> 16) & 255u) << 2) | 1u, kg = (t >> 24) << 2;
          const uint32_t ind = 3u - (max(max(ka, kc), max(kt, kg)) & 3u);
          cdx = (0x1320u >> (4 * ind)) & 3u;  // count row -> SPRING code
        }
        if (p < Rn) codes |= cdx << (2 * e);
      }
      if (p0 < hiP) *reinterpret_cast<uint4 *>(nxt8 + p0) = make_uint4(t4[0], t4[1], t4[2], t4[3]);
    } else {
#define MC_LOAD_CNT(I) (cur_wide ? cur[(I)] : unpack8(cur8[(I)]))
      for (int e = 0; e < 4; e++) {
        const int p = p0 + e;
        if (p >= hiP) continue;
        int4 t = make_int4(0, 0, 0, 0);
        int code = 0;
        if (!alias) {
          if (p < cpy_hi) t = MC_LOAD_CNT(p + src_off);
          if (p >= add_lo && p < add_hi) {
            code = cur_base(rd + 1, p - add_lo, n, rev);
            add_hot(t, cidx_of_code(code));
          }
        } else {  // reverse case 1 with d > 0 (reorder.h:159-174), closed form of the in-place loop
          if (p >= d && p < n - shift) {
            t = MC_LOAD_CNT(p % d);
            for (int qq = p % d + d; qq <= p; qq += d) add_hot(t, cidx_of_code(cur_base(rd + 1, qq, n, rev)));
          } else {
            add_hot(t, cidx_of_code(cur_base(rd + 1, p, n, rev)));
          }
        }
        if (out_wide) nxt[p] = t; else nxt8[p] = pack8(t);
        ovf = ovf || max(max(t.x, t.y), max(t.z, t.w)) > 255;
        if (p < Rn) codes |= (uint32_t)(reset ? code : argmax_code(t)) << (2 * e);
      }
#undef MC_LOAD_CNT
    }
    if (q < 8 * GLds<NQ, KA>::WMAX) reinterpret_cast<uint8_t *>(S.nref)[q] = (uint8_t)codes;
  }
  overflow = gballot(ovf, (int)threadIdx.x) != 0;
  return Rn;
}

// commit of a speculative update: S.nref -> ref (LDS + global), revref = its reverse complement (LDS + global)
template <int NQ, bool KA>
__device__ __forceinline__ void commit_consensus(const DevParams &P, GLds<NQ, KA> &S, Chain *c, int R, int gl) {
  constexpr int LDS_PAD = GLds<NQ, KA>::PAD;
  wave_sync();
  const bool in = gl < P.W;
  const uint64_t limb = in ? S.nref[gl] : 0ull;
  if (gl < GLds<NQ, KA>::WMAX) S.refs[0][LDS_PAD + gl] = limb;
  if (in) c->ref[gl] = limb;
  wave_sync();
  uint64_t rl = 0;
  if (in && 32 * gl < R) {
    // bases [32 gl, 32 gl + 32) of the reverse complement = ref bases R-1-32gl downwards, complemented
    const uint64_t w = lds_window(S.refs[0] + LDS_PAD, 2 * (R - 32 - 32 * gl));
    uint64_t x = __builtin_bitreverse64(w);
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    const int nv = R - 32 * gl;
    rl = ~x & (nv >= 32 ? ~0ull : ((1ull << (2 * nv)) - 1));
  }
  if (gl < GLds<NQ, KA>::WMAX) S.refs[1][LDS_PAD + gl] = rl;
  if (in) c->revref[gl] = rl;
  wave_sync();
}

// ---- known-absent windows (k_round_mc<.., KA = true>; reads up to 192 bases).  99 % of a search's probes are for keys the
// dictionaries do not hold, and the table never changes: "the 32-mer under this window is absent" stays true for as long as
// the bases under the window do.  A chain keeps two bits per window offset and strand (GLds::ka), sets them when the tags of
// a fetched bucket prove a key absent (search_ka), moves them with the consensus when a read is merged and drops those of
// every window that covers a position whose consensus base changed or is new (below), swaps the strands when a lone seed
// turns round for its left search (its consensus is exactly the old reverse consensus), and clears them on a new seed.
// A later search skips the windows it knows about BEFORE a key or an address is formed: the first batch of the search
// that follows a match at shift s has already seen all but s of its shifts, a failing search repeated after a lost
// proposal fetches nothing.  Exactness is untouched: a skipped probe is one whose answer was "absent".
// Between rounds the masks live in the unused half of the chain record (Chain::revref[8..15]: four limbs of either
// strand, from limb DevParams::ka_lo of the forward one -- the offsets a search can ask about, reorder.h:262-270).
__device__ __forceinline__ uint64_t gshfl_down1_u64(uint64_t v) {
  const uint32_t lo = __shfl_down((uint32_t)v, 1, G), hi = __shfl_down((uint32_t)(v >> 32), 1, G);
  return ((uint64_t)hi << 32) | lo;
}
// d / dn: changed positions (both bits of a position set) of this limb and the next one -> the windows of this limb (32
// offsets, 2 bits each) that cover one of them: a window spans the positions [o, o + 31] (shorter windows: conservative)
__device__ __forceinline__ uint64_t ka_stale(uint64_t d, uint64_t dn) {
  return (d ? ~0ull >> __clzll(d) : 0ull) | (dn ? (~0ull << (__ffsll((unsigned long long)dn) - 1)) << 2 : 0ull);
}
// mode 0: the update kept old position p + src_off at new position p < cpy_hi; 1: nothing is known about the new
// consensus; 2: ref and revref change places
template <int NQ, bool KA>
__device__ __forceinline__ void commit_consensus_ka(const DevParams &P, GLds<NQ, KA> &S, Chain *c, int R_old, int R, int src_off,
                                                    int cpy_hi, int mode, int gl) {
  constexpr int LDS_PAD = GLds<NQ, KA>::PAD, WM = GLds<NQ, KA>::WMAX;
  constexpr uint64_t EVEN = 0x5555555555555555ull;
  const int dlt = R_old - R - src_off;  // revref_new[j] = revref_old[j + dlt] for j >= -dlt (dlt <= 0: reorder.h:144-200)
  if (mode == 0 && (src_off > 32 * (LDS_PAD - 1) || dlt < -32 * (LDS_PAD - 1) || dlt > 0)) mode = 1;
  wave_sync();
  const bool in = gl < P.W;
  const uint64_t limb = in ? S.nref[gl] : 0ull;
  uint64_t kf = 0, kr = 0, df = ~0ull, dr = ~0ull;
  if (gl < WM) {
    if (mode == 0) {
      uint64_t x = limb ^ lds_window(S.refs[0] + LDS_PAD, 64 * gl + 2 * src_off);
      x = (x | (x >> 1)) & EVEN;
      x |= x << 1;
      const int b = 2 * cpy_hi - 64 * gl;  // positions >= cpy_hi are new
      if (b < 64) x |= b <= 0 ? ~0ull : ~0ull << b;
      df = x;
      kf = lds_window(S.ka[0] + LDS_PAD, 64 * gl + 2 * src_off);
    } else if (mode == 2) {
      kf = S.ka[1][LDS_PAD + gl];
      kr = S.ka[0][LDS_PAD + gl];
    }
  }
  if (mode == 0) kf &= ~ka_stale(df, gshfl_down1_u64(df));
  wave_sync();
  if (gl < WM) { S.refs[0][LDS_PAD + gl] = limb; S.ka[0][LDS_PAD + gl] = kf; }
  if (in) c->ref[gl] = limb;
  wave_sync();
  uint64_t rl = 0;
  if (in && 32 * gl < R) {
    const uint64_t w = lds_window(S.refs[0] + LDS_PAD, 2 * (R - 32 - 32 * gl));
    uint64_t x = __builtin_bitreverse64(w);
    x = ((x >> 1) & EVEN) | ((x & EVEN) << 1);
    const int nv = R - 32 * gl;
    rl = ~x & (nv >= 32 ? ~0ull : ((1ull << (2 * nv)) - 1));
  }
  if (mode == 0) {
    if (gl < WM) {
      uint64_t x = rl ^ lds_window(S.refs[1] + LDS_PAD, 64 * gl + 2 * dlt);
      x = (x | (x >> 1)) & EVEN;
      x |= x << 1;
      const int b = -2 * dlt - 64 * gl;  // positions < -dlt are new
      if (b > 0) x |= b >= 64 ? ~0ull : ~(~0ull << b);
      dr = x;
      kr = lds_window(S.ka[1] + LDS_PAD, 64 * gl + 2 * dlt);
    }
    kr &= ~ka_stale(dr, gshfl_down1_u64(dr));
  }
  wave_sync();
  if (gl < WM) { S.refs[1][LDS_PAD + gl] = rl; S.ka[1][LDS_PAD + gl] = kr; }
  if (in) c->revref[gl] = rl;
  wave_sync();
}

__device__ __forceinline__ uint32_t take_slot_mc(uint32_t &slot, uint32_t *alloc, uint2 *chunk, uint32_t li, uint32_t next_seq, int gl) {
  const uint32_t s0 = slot;
  uint32_t nx = s0 + 1;
  if ((nx & (CHUNK - 1)) == 0) {
    uint32_t got = 0;
    if (gl == 0) {
      got = atomicAdd(alloc, CHUNK);
      chunk[got / CHUNK] = make_uint2(li, next_seq);
    }
    nx = (uint32_t)__shfl((int)got, 0, G);
  }
  slot = nx;
  return s0;
}
__device__ __forceinline__ void emit_rec_mc(const DevParams &P, ChainHot &h, uint32_t li, uint32_t rid, char rc, char flag,
                                            long long pos, int gl) {
  const uint32_t seq = h.n_emit;
  h.n_emit = seq + 1;
  uint32_t slot = h.e_slot;
  const uint32_t idx = take_slot_mc(slot, &P.glob->e_alloc, P.e_chunk, li, seq + 1, gl);
  h.e_slot = slot;
  if (gl == 0)
    P.e_rec[idx] = make_uint4(rid, (uint32_t)(uint8_t)rc | ((uint32_t)(uint8_t)flag << 8), (uint32_t)pos,
                              (uint32_t)((unsigned long long)pos >> 32));
}
__device__ __forceinline__ void emit_single_mc(const DevParams &P, ChainHot &h, uint32_t li, uint32_t rid, int gl) {
  const uint32_t seq = h.n_single;
  h.n_single = seq + 1;
  uint32_t slot = h.s_slot;
  const uint32_t idx = take_slot_mc(slot, &P.glob->s_alloc, P.s_chunk, li, seq + 1, gl);
  h.s_slot = slot;
  if (gl == 0) P.s_rec[idx] = rid;
}

// phase B of one chain (apply_step with the shared state deferred to k_mg_mark).  false: the chain is done.
template <int NQ, bool KA>
__device__ __forceinline__ bool apply_mc(const DevParams &P, Chain *c, uint32_t cid, uint32_t li, ChainHot &h, GLds<NQ, KA> &S, int gl) {
  const int kind = h.prop_kind;
  if (kind == PROP_FRESH) return true;  // first round: nothing proposed yet
  if (h.finishing) {  // seed-needing chain found the pool empty
    if (h.prev_unmatched) emit_single_mc(P, h, li, h.prev, gl);
    h.done = 1; h.finishing = 0;
    store_hot(c, h, gl);
    return false;
  }
  const uint32_t owner = kind != PROP_NONE ? P.resv[h.prop_rid] : cid;  // in flight while the update is computed
  const bool fail_path = kind == PROP_NONE && h.mode == MODE_SEARCH;
  bool do_upd = false, ureset = false, urev = false;
  uint32_t urid = 0;
  int ushift = 0;
  if (kind == PROP_MATCH) { do_upd = true; urid = h.prop_rid; urev = h.prop_rev & 1; ushift = (int)h.prop_shift; }
  else if (kind == PROP_SEED) { do_upd = true; urid = h.prop_rid; ureset = true; }
  else if (fail_path && !h.left_search) { do_upd = true; urid = h.first_rid; ureset = true; urev = true; }  // reorder.h:567
  int n = P.L, R_new = h.ref_len;
  const int R_old = h.ref_len;
  bool nw = false, ualias = false;
  int usrc = 0, ucpy = 0;
  if (do_upd) {
    if (!P.uniform_len) n = (int)P.lens[urid];
    R_new = update_mc<NQ, KA>(P, li, S, urid, n, ureset, urev, ushift, R_old, (int)h.cnt_buf, h.cnt_wide != 0, false, nw, gl,
                          &usrc, &ucpy, &ualias);
    if (nw) {  // a count would pass 255: redo in the wide format
      bool o2;
      wave_sync();
      R_new = update_mc<NQ, KA>(P, li, S, urid, n, ureset, urev, ushift, R_old, (int)h.cnt_buf, h.cnt_wide != 0, true, o2, gl);
    }
  }
  if (owner != cid) {  // lost the read: retry, nothing committed
    if (h.mode == MODE_SEARCH) h.retrying = 1;
    store_hot(c, h, gl, 3, 4);
    if (gl == 0) atomicAdd((unsigned long long *)&c->st_lost, 1ull);
    return true;
  }
  if (do_upd) {
    if constexpr (KA) {
      // what the chain knows of its windows after this update: carried along (a merged read), turned round (a lone seed
      // starts its left search: the new consensus is the old reverse consensus, reorder.h:562-571), or nothing
      const int kmode = kind == PROP_MATCH ? (ualias ? 1 : 0) : (kind != PROP_SEED && h.prev_unmatched) ? 2 : 1;
      commit_consensus_ka<NQ, KA>(P, S, c, R_old, R_new, usrc, ucpy, kmode, gl);
    } else commit_consensus<NQ, KA>(P, S, c, R_new, gl);
    h.ref_len = R_new;
    h.cnt_buf ^= 1;
    h.cnt_wide = nw;
  }
  if (kind == PROP_MATCH) {
    const uint32_t rid = h.prop_rid;
    const int shift = ushift;
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
    if (h.prev_unmatched) emit_rec_mc(P, h, li, h.prev, 'd', '0', 0, gl);
    emit_rec_mc(P, h, li, rid, rcch, '1', cur_pos, gl);
    h.prev_unmatched = 0; h.ref_pos = ref_pos; h.retrying = 0;
  } else if (kind == PROP_SEED) {  // reorder.h:580-587, :600-613
    const uint32_t rid = h.prop_rid;
    if (gl == 0) atomicAdd((unsigned long long *)&c->n_unmatched, 1ull);
    if (h.prev_unmatched) emit_single_mc(P, h, li, h.prev, gl);
    h.prev_unmatched = 1; h.first_rid = rid; h.prev = rid;
    h.ref_pos = 0; h.mode = MODE_SEARCH;
  } else if (fail_path) {  // search failed (reorder.h:559-575)
    h.retrying = 0;
    h.num_unmatched_past++;
    if (!h.left_search) { h.left_search = 1; h.ref_pos = 0; }
    else { h.left_search = 0; h.mode = MODE_NEED_SEED; }  // (k_mg_mark put the chain on the needy bitmap: PK_WILLNEED)
  }
  store_hot(c, h, gl);
  return true;
}

// find_seed for a 16-lane group (fused rounds: needy_cnt is kept by k_mg_mark).  See find_seed.
__device__ __forceinline__ long long find_seed_mc(const DevParams &P, uint32_t cid, int gl, bool *is_last) {
  int r = 0, tot = 0;
  const long long top = *P.cursor;
  const uint32_t nw = (P.Ktot + 31) / 32, myw = cid >> 5;
  const uint32_t myblk = myw >> 6;  // (ranks count the chains of this launch's group: blocks [nb_lo, nb_hi) of 2048 chains)
#pragma unroll
  for (int k = 0; k < 4; k++) {  // the 64 bitmap words of this chain's own block of 2048 chains
    const uint32_t w = myblk * 64 + 4 * gl + k;
    const uint32_t v = w < nw ? P.needy[w] : 0u;
    if (w < myw) r += __popc(v);
    else if (w == myw) r += __popc(v & ((1u << (cid & 31)) - 1u));
  }
  for (uint32_t b = P.nb_lo + gl; b < P.nb_hi; b += G) {
    const uint32_t v = P.needy_cnt[b];
    tot += (int)v;
    if (b < myblk) r += (int)v;
  }
  const uint32_t rank = (uint32_t)gsum_i(r), nneedy = (uint32_t)gsum_i(tot);
  *is_last = rank + 1 == nneedy;
  uint32_t need = rank + 1;
  if (top < (long long)P.seed_lo) return -1;  // (seeds come from reads [seed_lo, ...) -- the whole pool unless the chains run in two groups)
  const long long blo = (long long)(P.seed_lo >> UBLK_SHIFT);
  constexpr int WPB_ = 1 << (UBLK_SHIFT - 6);  // bitmap words per block
  constexpr int WPG = WPB_ / G;                // ... per lane of the group
  static_assert(WPG >= 1 && WPG * G == WPB_ && WPG <= G, "a block is 1..16 bitmap words per lane");
  // the need-th highest untaken read of block blk (it holds at least that many)
  auto pick = [&](long long blk, uint32_t want) -> long long {
    const long long wtop = blk * WPB_ + (WPB_ - 1);
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < WPG; k++) cnt += __popcll(~P.taken[wtop - WPG * gl - k]);
    const int inc = gincl_scan_i(cnt, gl);
    const uint32_t m = gballot((uint32_t)inc >= want, (int)threadIdx.x);
    const int wl = __ffs((int)m) - 1;
    const uint32_t want2 = want - (uint32_t)__shfl(inc - cnt, wl, G);
    // lane wl's words, one per lane
    const long long wbase = wtop - WPG * wl;
    const uint64_t u = gl < WPG ? ~P.taken[wbase - gl] : 0ull;
    const int c1 = __popcll(u);
    const int inc1 = gincl_scan_i(c1, gl);
    const uint32_t m1 = gballot((uint32_t)inc1 >= want2, (int)threadIdx.x);
    const int wk = __ffs((int)m1) - 1;
    int kth = (int)want2 - __shfl(inc1 - c1, wk, G);
    uint64_t v = gshfl_u64(u, wk);
    for (int t = 1; t < kth; t++) v &= ~(1ull << (63 - __clzll(v)));
    return (wbase - wk) * 64 + (63 - __clzll(v));
  };
  const long long bt = top >> UBLK_SHIFT;
  {
    const long long wtop = bt * WPB_ + (WPB_ - 1);
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < WPG; k++) cnt += __popcll(~P.taken[wtop - WPG * gl - k]);
    const uint32_t tot0 = (uint32_t)gsum_i(cnt);
    if (tot0 >= need) return pick(bt, need);
    need -= tot0;
  }
  for (long long b0 = bt - 1; b0 >= blo; b0 -= G) {
    const long long bb = b0 - gl;
    const int u = bb >= blo ? (int)P.ublk[bb] : 0;
    const int incl = gincl_scan_i(u, gl);
    const uint32_t total = (uint32_t)__shfl(incl, G - 1, G);
    if (total < need) { need -= total; continue; }
    const uint32_t mb = gballot((uint32_t)incl >= need, (int)threadIdx.x);
    const int wb = __ffs((int)mb) - 1;
    need -= (uint32_t)__shfl(incl - u, wb, G);
    return pick(b0 - wb, need);
  }
  return -1;
}

constexpr int INF_CODE = 0x7fffffff;

#ifdef SR_EXP_FAKE
// index (in 16-byte units) of the quad a probe of the window at consensus offset `o` fetches in the locality experiment
__device__ __forceinline__ uint64_t exp_fake_quad(const DevParams &P, const uint64_t *sref, int o, uint64_t key) {
  const uint32_t grp = SR_EXP_FAKE == 2 ? (uint32_t)(o >> 3) : (uint32_t)o;
  uint32_t hh = ((uint32_t)sref[0] * 2654435761u) ^ (grp * 0x9E3779B1u);
  hh ^= hh >> 15; hh *= 0x2c1b3c6du; hh ^= hh >> 12; hh *= 0x297a2d39u; hh ^= hh >> 15;
  const uint64_t nl = (1ull << (64 - P.tab.bshift)) >> 1;  // 64-byte lines of the table
  return (uint64_t)(hh & (uint32_t)(nl - 1)) * 4 + ((mix64(key) >> 30) & 3);
}
#endif

// One step of a probe's bucket chain on the tags alone: t = the tags of a bucket of the chain of the key with hash hsh.
// 1: a slot with the fingerprint of dictionary l (a closer look is needed: eval_probe); 2: none, and the bucket is
// full -- the chain goes on in the next bucket; 0: none, and the chain ends here: the key is absent from dictionary l.
// `other`: a slot with the fingerprint of the OTHER dictionary was seen (what the tail wants to know about this window).
// slot (only when 1): index of the first such slot | its `single` bit << 2.
__device__ __forceinline__ int tags_step(const uint4 &t, uint64_t hsh, int l, bool &other, int &slot) {
  const uint32_t mine = (fp30_of(hsh) << 2) | ((uint32_t)l << 1), theirs = mine ^ 2u;
  const bool full = t.w != 0;
  other = other || (t.x & ~1u) == theirs || (t.y & ~1u) == theirs || (t.z & ~1u) == theirs || (t.w & ~1u) == theirs;
  const uint32_t m = (uint32_t)((t.x & ~1u) == mine) | ((uint32_t)((t.y & ~1u) == mine) << 1) |
                     ((uint32_t)((t.z & ~1u) == mine) << 2) | ((uint32_t)((t.w & ~1u) == mine) << 3);
  if (m) {
    other = other || full;  // (the other dictionary's slot may sit in a later bucket of the chain)
    const int I = __ffs((int)m) - 1;
    const uint32_t Tg = I == 0 ? t.x : I == 1 ? t.y : I == 2 ? t.z : t.w;
    slot = I | (int)((Tg & 1u) << 2);
    return 1;
  }
  return full ? 2 : 0;
}
constexpr int QUICK_HOPS = 2;  // buckets past the home bucket (or the redirect bucket of a marked neighbourhood) looked at on the tags alone

// Window minimizers of a chain's consensus (TabView::minz): S.mz[w] = minz_of_key(the 32-mer at offset w of ref),
// w in [0, R - 32].  The reverse consensus needs no array of its own: kmer_order is strand-symmetric, so the window
// at offset o of revref has the k-mers of the ref window at offset R - 32 - o.  Lane g of row j holds the order
// value of the k-mer at q = 16 j + g; a window has 17 = 16 + 1 k-mers, so its minimum is the suffix minimum of one row
// from lane g on and the prefix minimum of the next row up to lane g: two 4-step scans per row, no LDS round trip.
template <int NQ, bool KA>
__device__ __forceinline__ void minz_mc(GLds<NQ, KA> &S, int R, int gl) {
  static_assert(MINZ_WL - MINZ_K + 1 == G + 1, "a window's k-mers = one row of lanes + 1");
  const uint64_t *sref = S.refs[0] + GLds<NQ, KA>::PAD;
  uint32_t sfx_prev = 0xffffffffu;
  for (int j = 0; j == 0 || G * (j - 1) + MINZ_WL <= R; j++) {
    const int q = G * j + gl;
    uint32_t a = 0xffffffffu;
    if (q + MINZ_K <= R) a = kmer_order((uint32_t)lds_window(sref, 2 * q));
    // inclusive prefix / suffix minimum over the 16-lane row: DPP row shifts (a lane shifted in from outside the row
    // keeps the identity 0xffffffff: bound_ctrl off, old = identity), one VALU instruction per step, no LDS
    uint32_t pfx = a, sfx = a;
#define MC_DPP_MIN(V, CTRL) V = min(V, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)V, CTRL, 0xf, 0xf, false))
    MC_DPP_MIN(pfx, 0x111); MC_DPP_MIN(pfx, 0x112); MC_DPP_MIN(pfx, 0x114); MC_DPP_MIN(pfx, 0x118);  // row_shr:1,2,4,8
    MC_DPP_MIN(sfx, 0x101); MC_DPP_MIN(sfx, 0x102); MC_DPP_MIN(sfx, 0x104); MC_DPP_MIN(sfx, 0x108);  // row_shl:1,2,4,8
#undef MC_DPP_MIN
    const int w = G * (j - 1) + gl;
    if (j > 0 && w + MINZ_WL <= R && w < GLds<NQ, KA>::MZ) S.mz[w] = fmix32(min(sfx_prev, pfx));
    sfx_prev = sfx;
  }
}

// probes with priority codes [c_lo, c_hi) (code = shift << 2 | rev << 1 | dict; at most 4 * G of them): lane gl takes
// codes c_lo + gl + 16 i, fetches the tag quads of all of them first, and only walks into eval_probe where the
// quad does not already prove the key absent (2 % of the probes).  Winner = lowest code that hit.
template <int NQ, bool KA>
__device__ __forceinline__ void batch_mc(const DevParams &P, GLds<NQ, KA> &S, lds_u32_t *stage, int c_lo, int c_hi, int ref_len,
                                         int lane, int gl, int &wcode, uint32_t &wrid) {
  constexpr int LDS_PAD = GLds<NQ, KA>::PAD;
  const uint64_t *sref = S.refs[0] + LDS_PAD, *srev = S.refs[1] + LDS_PAD;
  const bool minz = P.tab.minz != 0;
  const int klen2 = 2 * P.wl;
  const uint64_t kmask = klen2 < 64 ? ((1ull << klen2) - 1) : ~0ull;
  lds_u32_t *s_best = (lds_u32_t *)&S.best;
  if (gl == 0) *s_best = (uint32_t)INF_CODE;
  wave_sync();
  const uint32_t bmask = (uint32_t)bucket_mask(P.tab.bshift);  // (k_round_mc runs on tables of at most 2^32 buckets)
  uint4 tg[4];
  uint64_t key[4];
  uint32_t bk[4];
  bool val[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int code = c_lo + gl + G * i;
    const int l = code & 1, rev = (code >> 1) & 1, shift = code >> 2;
    val[i] = code < c_hi && probe_valid(P, l, rev, shift, ref_len);
    key[i] = 0;
    bk[i] = 0;
    tg[i] = make_uint4(0, 0, 0, 0);
    if (val[i]) {
      const int ds = l ? P.dstart[1] : P.dstart[0];
      key[i] = lds_window(rev ? srev : sref, rev ? 2 * (ds - shift) : 2 * (ds + shift)) & kmask;
      const int o = rev ? ds - shift : ds + shift;  // the window's offset in its strand
#ifdef SR_EXP_FAKE  // experiment (tools/xbuild.sh -DSR_EXP_FAKE=1|2): every probe absent; 2 = runs of 8 windows share a 64-byte line
      tg[i] = P.tab.buck[exp_fake_quad(P, sref, rev ? ref_len - P.wl - o : o, key[i])];
#else
      bk[i] = (uint32_t)tab_home(P.tab, mix64(key[i]), minz ? S.mz[rev ? ref_len - MINZ_WL - o : o] : 0u);
      tg[i] = P.tab.buck[2 * (uint64_t)bk[i]];
#endif
    }
  }
  PT(25);   // (experiment builds: batch set-up and address arithmetic | tag fetch | judge + hops | passes)
  PTW(26);
  // Which of this lane's probes need a closer look (pend), judged on the tags alone: a probe whose bucket is full without a
  // slot of its key (5 % of the probes of a minimizer-addressed table, 0.2 % otherwise) or whose neighbourhood is marked
  // follows its chain here, on tags that mostly sit in the cache line just fetched -- a pass through eval_probe is a
  // chain of four dependent memory steps that every chain of the wavefront waits for.  `pres` for the tail.
  // pre: 4 bits per probe, kind | slot << 2 of the key's first slot when that is in its home bucket (eval_probe then starts
  // at the payload word instead of walking the chain again)
  uint32_t pend = 0, cont = 0, redir = 0, oth = 0, pre = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (!val[i]) continue;
    const int code = c_lo + gl + G * i;
    if (minz && tg[i].x == TAG_MARK) { cont |= 1u << i; redir |= 1u << i; continue; }
    bool other = false;
    int slot = 0;
    const int st = tags_step(tg[i], mix64(key[i]), code & 1, other, slot);
    if (other) oth |= 1u << i;
    if (st == 1) { pend |= 1u << i; pre |= (uint32_t)((1 + ((slot >> 2) & 1)) | ((slot & 3) << 2)) << (4 * i); }
    else if (st == 2) cont |= 1u << i;
  }
  for (int hop = 0; hop < QUICK_HOPS && __ballot(cont != 0); hop++) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      if ((cont >> i) & 1u) {
        bk[i] = ((redir >> i) & 1u) ? (uint32_t)tab_redirect(P.tab, mix64(key[i])) : ((bk[i] + 1u) & bmask);
        tg[i] = P.tab.buck[2 * (uint64_t)bk[i]];
      }
    redir = 0;
    const uint32_t c2 = cont;
    cont = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if ((c2 >> i) & 1u) {
        const int code = c_lo + gl + G * i;
        bool other = false;
        int slot = 0;
        const int st = tags_step(tg[i], mix64(key[i]), code & 1, other, slot);
        if (other) oth |= 1u << i;
        if (st == 1) pend |= 1u << i;  // (past the home bucket: eval_probe walks the chain itself)
        else if (st == 2) cont |= 1u << i;
      }
  }
  pend |= cont;  // chains longer than that: eval_probe walks them
  oth |= cont;
  PTW(27);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int code = c_lo + gl + G * i;
    if (code < c_hi && code < 128) S.pres[code] = ((oth >> i) & 1u) ? 1 : 0;
  }
#ifdef SR_EXP_FAKE
  pend = tg[0].x == 0x12345u && tg[1].y == 0x54321u && tg[2].z == 77u && tg[3].w == 99u;  // (keeps the loads alive) never
#endif
  // the closer looks of all lanes run together, one per lane and pass, lowest code first (a pass is a chain of
  // dependent loads: record / taken bit / candidate read -- four chains' worth of them in flight at once)
  int best = INF_CODE;
  uint32_t brid = 0;
  while (__ballot(pend != 0)) {
    if (pend) {
      const int i = __ffs((int)pend) - 1;
      pend &= pend - 1;
      const int code = c_lo + gl + G * i;
      const int l = code & 1, rev = (code >> 1) & 1, shift = code >> 2;
      if (!(*(volatile lds_u32_t *)s_best < (uint32_t)code)) {
        bool hit = false, keyok = false, other = false;
        uint32_t rid = 0, ncand = 0;
        // (key, hash and minimizer are taken from LDS again: nothing of the quick phase stays live across the passes)
        const int ds = l ? P.dstart[1] : P.dstart[0], o = rev ? ds - shift : ds + shift;
        const uint64_t k = lds_window(rev ? srev : sref, 2 * o) & kmask;
        eval_probe<false, false, false, true>(P, rev ? srev : sref, l, rev, shift, ref_len, k, mix64(k),
                                       minz ? S.mz[rev ? ref_len - MINZ_WL - o : o] : 0u, hit, rid, keyok, ncand, other, s_best, stage,
                                       lane, nullptr, nullptr, (int)((pre >> (4 * i)) & 15u));
        if (other && code < 128) S.pres[code] = 1;
        if (hit) { best = code; brid = rid; pend = 0; }
      } else pend = 0;  // a lower code has hit: nothing of this lane can win any more
    }
  }
  wcode = gmin_i(best);
  const uint32_t wm = gballot(best == wcode, lane);
  wrid = (uint32_t)__shfl((int)brid, __ffs((int)wm) - 1, G);
}

// every remaining probe (shifts [t0, maxshift)), one table fetch per distinct consensus window (see probe_tail)
template <int NQ, bool KA>
__device__ __forceinline__ void tail_mc(const DevParams &P, GLds<NQ, KA> &S, lds_u32_t *stage, int t0, int ref_len, int lane, int gl,
                                        int &wcode, uint32_t &wrid) {
  constexpr int LDS_PAD = GLds<NQ, KA>::PAD;
  const uint64_t *sref = S.refs[0] + LDS_PAD, *srev = S.refs[1] + LDS_PAD;
  const bool minz = P.tab.minz != 0;
  const int wl = P.wl, s0 = P.dstart[0], s1 = P.dstart[1], ms = P.maxshift;
  const uint64_t kmask = 2 * wl < 64 ? ((1ull << (2 * wl)) - 1) : ~0ull;
  lds_u32_t *s_best = (lds_u32_t *)&S.best;
  if (gl == 0) *s_best = (uint32_t)INF_CODE;
  wave_sync();
  const int nF = wl + ms - t0;
  int best = INF_CODE;
  uint32_t brid = 0;
  for (int base = 0; base < 2 * nF; base += 4 * G) {
    uint4 tg[4];
    uint64_t key[4];
    uint32_t bk[4];
    uint32_t desc[4];  // bit 31 rev, 30 v1, 29 v0, low bits i
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = base + gl + G * k;
      bool v0 = false, v1 = false;
      int rev = 0, i = 0;
      if (idx < 2 * nF) {
        rev = idx >= nF;
        i = rev ? idx - nF : idx;
        if (!rev) {
          const int sh0 = t0 + i, sh1 = sh0 - wl;
          v0 = probe_valid(P, 0, 0, sh0, ref_len);
          v1 = sh1 >= t0 && probe_valid(P, 1, 0, sh1, ref_len);
          if (v0 && sh1 >= 0 && sh1 < t0) v0 = S.pres[4 * sh1 + 1] != 0;
        } else {
          const int sh1 = t0 + i, sh0 = sh1 - wl;
          v1 = probe_valid(P, 1, 1, sh1, ref_len);
          v0 = sh0 >= t0 && probe_valid(P, 0, 1, sh0, ref_len);
          if (v1 && sh0 >= 0 && sh0 < t0) v1 = S.pres[4 * sh0 + 2] != 0;
        }
      }
      desc[k] = ((uint32_t)rev << 31) | ((uint32_t)v1 << 30) | ((uint32_t)v0 << 29) | (uint32_t)i;
      key[k] = 0;
      bk[k] = 0;
      tg[k] = make_uint4(0, 0, 0, 0);
      if (v0 || v1) {
        const int off = rev ? s1 - (t0 + i) : s0 + t0 + i;
        key[k] = lds_window(rev ? srev : sref, 2 * off) & kmask;
#ifdef SR_EXP_FAKE
        tg[k] = P.tab.buck[exp_fake_quad(P, sref, rev ? ref_len - wl - off : off, key[k])];
#else
        bk[k] = (uint32_t)tab_home(P.tab, mix64(key[k]), minz ? S.mz[rev ? ref_len - MINZ_WL - off : off] : 0u);
        tg[k] = P.tab.buck[2 * (uint64_t)bk[k]];
#endif
      }
    }
    PT(28);
    PTW(29);
    // pending (window, dictionary) pairs of this lane: bit 2k + j, j = 0 the probe with the lower priority code of
    // window k (forward: dictionary 1, its shift is wl lower; reverse: dictionary 0).  Judged on the tags alone, the
    // chain of a full bucket followed here (batch_mc); `un`: pairs not decided yet; pre: 4 bits per pair as in batch_mc
    uint32_t pend = 0, un = 0, cont = 0, redir = 0, pre = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t e = desc[k];
      if (!(e & (3u << 29))) continue;
      const int rev = (int)(e >> 31);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int l = rev ? j : 1 - j;
        if ((e >> (29 + l)) & 1u) un |= 1u << (2 * k + j);
      }
      if (minz && tg[k].x == TAG_MARK) { cont |= 1u << k; redir |= 1u << k; continue; }
      const uint64_t hsh = mix64(key[k]);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (!((un >> (2 * k + j)) & 1u)) continue;
        const int l = rev ? j : 1 - j;
        bool other = false;
        int slot = 0;
        const int st = tags_step(tg[k], hsh, l, other, slot);
        if (st == 1) { pend |= 1u << (2 * k + j); pre |= (uint32_t)((1 + ((slot >> 2) & 1)) | ((slot & 3) << 2)) << (4 * (2 * k + j)); }
        if (st != 2) un &= ~(1u << (2 * k + j));
      }
      if ((un >> (2 * k)) & 3u) cont |= 1u << k;
    }
    for (int hop = 0; hop < QUICK_HOPS && __ballot(cont != 0); hop++) {
      const uint32_t bmask = (uint32_t)bucket_mask(P.tab.bshift);
#pragma unroll
      for (int k = 0; k < 4; k++)
        if ((cont >> k) & 1u) {
          bk[k] = ((redir >> k) & 1u) ? (uint32_t)tab_redirect(P.tab, mix64(key[k])) : ((bk[k] + 1u) & bmask);
          tg[k] = P.tab.buck[2 * (uint64_t)bk[k]];
        }
      redir = 0;
      const uint32_t c2 = cont;
      cont = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!((c2 >> k) & 1u)) continue;
        const int rev = (int)(desc[k] >> 31);
        const uint64_t hsh = mix64(key[k]);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (!((un >> (2 * k + j)) & 1u)) continue;
          const int l = rev ? j : 1 - j;
          bool other = false;
          int slot = 0;
          const int st = tags_step(tg[k], hsh, l, other, slot);
          if (st == 1) pend |= 1u << (2 * k + j);
          if (st != 2) un &= ~(1u << (2 * k + j));
        }
        if ((un >> (2 * k)) & 3u) cont |= 1u << k;
      }
    }
    pend |= un;  // chains longer than that: eval_probe walks them
    PTW(30);
#ifdef SR_EXP_FAKE
    pend = tg[0].x == 0x12345u && tg[1].y == 0x54321u && tg[2].z == 77u && tg[3].w == 99u;
#endif
    while (__ballot(pend != 0)) {
      if (pend) {
        const int b = __ffs((int)pend) - 1;
        pend &= pend - 1;
        const int k = b >> 1, j = b & 1;
        const uint32_t e = k == 0 ? desc[0] : k == 1 ? desc[1] : k == 2 ? desc[2] : desc[3];
        const int rev = (int)(e >> 31), i = (int)(e & 0xfffffu);
        const int l = rev ? j : 1 - j;
        const int sh0 = rev ? t0 + i - wl : t0 + i, sh1 = rev ? t0 + i : t0 + i - wl;
        const int sh = l ? sh1 : sh0, code = probe_code(sh, rev, l);
        if (code < best && !(*(volatile lds_u32_t *)s_best < (uint32_t)code)) {
          bool hit = false, keyok = false, other = false;
          uint32_t rid = 0, ncand = 0;
          const int off = rev ? s1 - (t0 + i) : s0 + t0 + i;
          const uint64_t ky = lds_window(rev ? srev : sref, 2 * off) & kmask;
          eval_probe<false, false, false, true>(P, rev ? srev : sref, l, rev, sh, ref_len, ky, mix64(ky),
                                         minz ? S.mz[rev ? ref_len - MINZ_WL - off : off] : 0u, hit, rid, keyok, ncand, other, s_best,
                                         stage, lane, nullptr, nullptr, (int)((pre >> (4 * b)) & 15u));
          if (hit) { best = code; brid = rid; }
        }
      }
    }
  }
  wcode = gmin_i(best);
  const uint32_t wm = gballot(best == wcode, lane);
  wrid = (uint32_t)__shfl((int)brid, __ffs((int)wm) - 1, G);
}

// ---- the search with known-absent windows (KA = true).  Probe codes in priority order (code = shift << 2 | rev << 1 |
// dict, reorder.h:479-558), and which lane of a chain's 16 owns which:
//   part 1, shifts below the window length wl: every code is a window of its own; lane gl owns the codes 16 i + gl
//     (stream gl & 3 = rev << 1 | dict, shifts 4 i + (gl >> 2)) -- bit i of its need word;
//   part 2, shifts wl .. maxshift - 1: the window of (forward, dictionary 0) at shift s is the window of (forward,
//     dictionary 1) at s - wl, that of (reverse, 1) at s the one of (reverse, 0) at s - wl (the dictionaries' windows are
//     adjacent and equally long, reorder.h:751-759) -- it has been fetched earlier in the search, and a fetch settles both
//     dictionaries.  So only the streams (forward, 1) and (reverse, 0) fetch here: lane gl owns stream gl & 1 at the
//     shifts wl + 8 i + (gl >> 1), bit 8 + i of its need word -- windows no code of part 1 touches, so the word is
//     complete when the search starts;
//   part 3: the codes of the other two streams at shifts >= wl whose window is STILL not known absent after all that (the
//     tags showed a slot of that dictionary: 1-2 % of the windows) -- bits 16 + i, same mapping, built after part 2.
// A lane takes its needed codes lowest first, m = 1, 2, 4, 4 ... per batch (DevParams::plan in units of four shifts):
// the tag quads of a batch are fetched together, judged on the tags alone (every proven absence goes into GLds::ka, for
// both dictionaries of the window), and only slots with the key's fingerprint get a closer look (eval_probe).  The winner
// is the lowest code that hit; a lane drops its codes above the best hit so far, so the search ends when every lower
// code has been looked at or was known absent -- whatever order the lanes got to them in.
__device__ __forceinline__ uint32_t ka_pack8_up(uint64_t w) {  // bit 8 i of w -> bit i
  const uint32_t lo = (uint32_t)w & 0x01010101u, hi = (uint32_t)(w >> 32) & 0x01010101u;
  return (((lo * 0x01020408u) >> 24) & 15u) | ((((hi * 0x01020408u) >> 24) & 15u) << 4);
}
__device__ __forceinline__ uint32_t ka_pack8_down(uint64_t w) {  // bit 56 - 8 i of w -> bit i
  const uint32_t lo = (uint32_t)w & 0x01010101u, hi = (uint32_t)(w >> 32) & 0x01010101u;
  return (((hi * 0x08040201u) >> 24) & 15u) | ((((lo * 0x08040201u) >> 24) & 15u) << 4);
}
__device__ __forceinline__ uint32_t ka_pack16_up(uint64_t w) {  // bit 16 i of w -> bit i (i < 4)
  const uint32_t lo = (uint32_t)w & 0x00010001u, hi = (uint32_t)(w >> 32) & 0x00010001u;
  return ((lo | (lo >> 15)) & 3u) | (((hi | (hi >> 15)) & 3u) << 2);
}
__device__ __forceinline__ uint32_t ka_pack16_down(uint64_t w) {  // bit 48 - 16 i of w -> bit i (i < 4)
  const uint32_t lo = (uint32_t)w & 0x00010001u, hi = (uint32_t)(w >> 32) & 0x00010001u;
  return ((hi >> 16) & 1u) | ((hi & 1u) << 1) | (((lo >> 16) & 1u) << 2) | ((lo & 1u) << 3);
}
__device__ __forceinline__ uint32_t ka_bits_range(int lo, int hi) {  // bits [lo, hi) of a byte, any lo / hi
  lo = max(lo, 0);
  hi = min(hi, 8);
  return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}
// valid shifts of stream (rev, l): [lo, hi) (probe_valid, reorder.h:264-268)
__device__ __forceinline__ void ka_valid_shifts(const DevParams &P, int l, int rev, int ref_len, int &lo, int &hi) {
  const int de = l ? uni_i32(P.dend[1]) : uni_i32(P.dend[0]), ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
  lo = rev ? max(0, de - ref_len + 1) : 0;
  hi = min(P.maxshift, rev ? ds : ref_len - de);
}
// need bits of part 1 (bits 0..7) for this lane
template <int NQ, bool KA>
__device__ __forceinline__ uint32_t ka_need1(const DevParams &P, GLds<NQ, KA> &S, int ref_len, int gl) {
  constexpr int LDS_PAD = GLds<NQ, KA>::PAD;
  const int l = gl & 1, rev = (gl >> 1) & 1, r = gl >> 2;
  const int ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
  int lo, hi;
  ka_valid_shifts(P, l, rev, ref_len, lo, hi);
  hi = min(hi, P.wl);
  const uint32_t valid = ka_bits_range((lo - r + 3) >> 2, (hi - r + 3) >> 2);
  const int b0 = 2 * (rev ? ds - r : ds + r) + l;
  const uint32_t known = rev ? ka_pack8_down(lds_window(S.ka[1] + LDS_PAD, b0 - 56)) : ka_pack8_up(lds_window(S.ka[0] + LDS_PAD, b0));
  return valid & ~known;
}
// need bits of part 2 (second = false: the streams that fetch) or part 3 (second = true: the streams that ride along), at bit 0
template <int NQ, bool KA>
__device__ __forceinline__ uint32_t ka_need2(const DevParams &P, GLds<NQ, KA> &S, int ref_len, int gl, bool second) {
  constexpr int LDS_PAD = GLds<NQ, KA>::PAD;
  const int rev = gl & 1, l = second ? rev : 1 - rev, r = gl >> 1;
  const int ds = l ? uni_i32(P.dstart[1]) : uni_i32(P.dstart[0]);
  int lo, hi;
  ka_valid_shifts(P, l, rev, ref_len, lo, hi);
  const int wl = P.wl;
  const uint32_t valid = ka_bits_range((lo - wl - r + 7) >> 3, (hi - wl - r + 7) >> 3);
  const int b0 = 2 * (rev ? ds - wl - r : ds + wl + r) + l;
  uint32_t known;
  if (rev) known = ka_pack16_down(lds_window(S.ka[1] + LDS_PAD, b0 - 48)) | (ka_pack16_down(lds_window(S.ka[1] + LDS_PAD, b0 - 112)) << 4);
  else known = ka_pack16_up(lds_window(S.ka[0] + LDS_PAD, b0)) | (ka_pack16_up(lds_window(S.ka[0] + LDS_PAD, b0 + 64)) << 4);
  return valid & ~known;
}
// probe code of need bit b of lane gl
__device__ __forceinline__ int ka_code_of_bit(int b, int gl, int wl) {
  if (b < 8) return 16 * b + gl;
  const int rev = gl & 1, l = b < 16 ? 1 - rev : rev;
  return ((wl + 8 * (b & 7) + (gl >> 1)) << 2) | (rev << 1) | l;
}
// this lane's need bits whose code is below `cur`
__device__ __forceinline__ uint32_t ka_keep_below(int cur, int gl, int wl) {
  if (cur == INF_CODE) return 0xffffffu;
  const uint32_t k1 = ka_bits_range(0, (cur - gl + 15) >> 4);
  const int rev = gl & 1, base = ((wl + (gl >> 1)) << 2) | (rev << 1);
  const uint32_t k2 = ka_bits_range(0, (cur - (base | (1 - rev)) + 31) >> 5), k3 = ka_bits_range(0, (cur - (base | rev) + 31) >> 5);
  return k1 | (k2 << 8) | (k3 << 16);
}

// one batch: up to m <= 4 of this lane's needed codes (lowest first) -- tag quads, verdict on the tags, closer looks
#ifndef SR_KA_M
#define SR_KA_M 4
#endif
constexpr int KA_M = SR_KA_M;  // probes a lane has in flight per batch (registers: 8 per probe)
template <int NQ, bool KA>
__device__ __forceinline__ void batch_ka(const DevParams &P, GLds<NQ, KA> &S, lds_u32_t *stage, uint32_t &need, int m, int ref_len,
                                         int lane, int gl, int &best, uint32_t &brid) {
  constexpr int LDS_PAD = GLds<NQ, KA>::PAD;
  const uint64_t *sref = S.refs[0] + LDS_PAD, *srev = S.refs[1] + LDS_PAD;
  const int wl = P.wl, klen2 = 2 * wl;
  const uint64_t kmask = klen2 < 64 ? ((1ull << klen2) - 1) : ~0ull;
  lds_u32_t *s_best = (lds_u32_t *)&S.best;
  const uint32_t bmask = (uint32_t)bucket_mask(P.tab.bshift);
  uint4 tg[KA_M];
  uint64_t key[KA_M];
  uint32_t bk[KA_M];
  int code[KA_M];
#pragma unroll
  for (int i = 0; i < KA_M; i++) {
    code[i] = -1;
    key[i] = 0;
    bk[i] = 0;
    tg[i] = make_uint4(0, 0, 0, 0);
    if (i < m && need) {
      const int b = __ffs((int)need) - 1;
      need &= need - 1;
      const int cd = ka_code_of_bit(b, gl, wl);
      code[i] = cd;
      const int l = cd & 1, rev = (cd >> 1) & 1, shift = cd >> 2;
      const int ds = l ? P.dstart[1] : P.dstart[0];
      const int o = rev ? ds - shift : ds + shift;  // the window's offset in its strand
      key[i] = lds_window(rev ? srev : sref, 2 * o) & kmask;
      bk[i] = (uint32_t)bucket_of(mix64(key[i]), P.tab.bshift);  // (hash-addressed tables only: DevParams::ka)
      tg[i] = P.tab.buck[2 * (uint64_t)bk[i]];
    }
  }
  // verdict on the tags alone (batch_mc): pend = a slot with the key's fingerprint, cont = the bucket is full without
  // one (the chain goes on), oth = the other dictionary may hold the window's key; pre as in batch_mc
  uint32_t pend = 0, cont = 0, oth = 0, pre = 0;
#pragma unroll
  for (int i = 0; i < KA_M; i++) {
    if (code[i] < 0) continue;
    bool other = false;
    int slot = 0;
    const int st = tags_step(tg[i], mix64(key[i]), code[i] & 1, other, slot);
    if (other) oth |= 1u << i;
    if (st == 1) { pend |= 1u << i; pre |= (uint32_t)((1 + ((slot >> 2) & 1)) | ((slot & 3) << 2)) << (4 * i); }
    else if (st == 2) cont |= 1u << i;
  }
  for (int hop = 0; hop < QUICK_HOPS && __ballot(cont != 0); hop++) {
#pragma unroll
    for (int i = 0; i < KA_M; i++)
      if ((cont >> i) & 1u) {
        bk[i] = (bk[i] + 1u) & bmask;
        tg[i] = P.tab.buck[2 * (uint64_t)bk[i]];
      }
    const uint32_t c2 = cont;
    cont = 0;
#pragma unroll
    for (int i = 0; i < KA_M; i++)
      if ((c2 >> i) & 1u) {
        bool other = false;
        int slot = 0;
        const int st = tags_step(tg[i], mix64(key[i]), code[i] & 1, other, slot);
        if (other) oth |= 1u << i;
        if (st == 1) pend |= 1u << i;  // (past the home bucket: eval_probe walks the chain itself)
        else if (st == 2) cont |= 1u << i;
      }
  }
  pend |= cont;  // chains longer than that: eval_probe walks them
  oth |= cont;
  // what the tags have proven absent: this code's dictionary (no slot of the key, the chain ended) and / or the other one
#pragma unroll
  for (int i = 0; i < KA_M; i++) {
    if (code[i] < 0) continue;
    const int cd = code[i], l = cd & 1, rev = (cd >> 1) & 1, shift = cd >> 2;
    const uint32_t own = ((pend >> i) & 1u) ^ 1u, other = ((oth >> i) & 1u) ^ 1u;
    const uint32_t bits = (own << l) | (other << (1 - l));
    if (bits) {
      const int ds = l ? P.dstart[1] : P.dstart[0];
      const int o = rev ? ds - shift : ds + shift;
      lds_u32_t *w = (lds_u32_t *)(S.ka[rev] + LDS_PAD) + (o >> 4);
      __hip_atomic_fetch_or(w, bits << (2 * (o & 15)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  // the closer looks of all lanes run together, one per lane and pass, lowest code first
  while (__ballot(pend != 0)) {
    if (pend) {
      const int i = __ffs((int)pend) - 1;
      pend &= pend - 1;
      int cd = code[0];
#pragma unroll
      for (int q = 1; q < KA_M; q++) cd = i == q ? code[q] : cd;
      const int l = cd & 1, rev = (cd >> 1) & 1, shift = cd >> 2;
      if (cd < best && !(*(volatile lds_u32_t *)s_best < (uint32_t)cd)) {
        bool hit = false, keyok = false, other = false;
        uint32_t rid = 0, ncand = 0;
        const int ds = l ? P.dstart[1] : P.dstart[0], o = rev ? ds - shift : ds + shift;
        const uint64_t k = lds_window(rev ? srev : sref, 2 * o) & kmask;
        eval_probe<false, false, false, true>(P, rev ? srev : sref, l, rev, shift, ref_len, k, mix64(k),
                                       0u, hit, rid, keyok, ncand, other, s_best, stage,
                                       lane, nullptr, nullptr, (int)((pre >> (4 * i)) & 15u));
        if (hit) { best = cd; brid = rid; }
      }
    }
  }
}

template <int NQ, bool KA>
__device__ __forceinline__ void search_ka(const DevParams &P, GLds<NQ, KA> &S, lds_u32_t *stage, int ref_len, int wide, int lane, int gl,
                                          int &wcode, uint32_t &wrid) {
  lds_u32_t *s_best = (lds_u32_t *)&S.best;
  if (gl == 0) *s_best = (uint32_t)INF_CODE;
  wave_sync();
  const int wl = P.wl;
  uint32_t need = ka_need1<NQ, KA>(P, S, ref_len, gl) | (ka_need2<NQ, KA>(P, S, ref_len, gl, false) << 8);
  int best = INF_CODE;
  uint32_t brid = 0;
  for (int b = 0; __ballot(need != 0); b++) {
    const int w = b < 6 ? (wide ? P.plan[1][b] : P.plan[0][b]) : 0;
    const int m = w > 0 ? min(KA_M, (w + 3) >> 2) : KA_M;
    batch_ka<NQ, KA>(P, S, stage, need, m, ref_len, lane, gl, best, brid);
    need &= ka_keep_below((int)*(volatile lds_u32_t *)s_best, gl, wl);
  }
  wave_sync();  // ka
  if ((int)*(volatile lds_u32_t *)s_best > 4 * wl) {  // the codes that rode along: only where a window is still not known absent
    need = (ka_need2<NQ, KA>(P, S, ref_len, gl, true) << 16) & ka_keep_below((int)*(volatile lds_u32_t *)s_best, gl, wl);
    while (__ballot(need != 0)) {
      batch_ka<NQ, KA>(P, S, stage, need, KA_M, ref_len, lane, gl, best, brid);
      need &= ka_keep_below((int)*(volatile lds_u32_t *)s_best, gl, wl);
    }
  }
  wcode = gmin_i(best);
  const uint32_t wm = gballot(best == wcode, lane);
  wrid = (uint32_t)__shfl((int)brid, __ffs((int)wm) - 1, G);
}

// phase A of one chain (search_step: proposal word + direct reservation of the read)
template <int NQ, bool MG, bool KA>
__device__ __forceinline__ void search_mc(const DevParams &P, Chain *c, uint32_t cid, ChainHot &h, GLds<NQ, KA> &S, lds_u32_t *stage,
                                          int lane, int gl) {
  if (h.mode == MODE_NEED_SEED) {
    bool is_last;
    const long long seed = find_seed_mc(P, cid, gl, &is_last);
    if (seed >= 0) {
      h.prop_kind = PROP_SEED;
      h.prop_rid = (uint32_t)seed;
      h.cursor_writer = is_last;  // the last-ranked needy chain proposes the lowest seed of the round
      if (gl == 0) {
        P.prop[cid] = ((unsigned long long)PK_SEED << 32) | (uint32_t)seed | (is_last ? PK_CURSOR_BIT : 0ull);
        atomicMin(&P.resv[seed], cid);  // (multi-GPU pools too: the rank's own proposals are settled here, k_mg_resolve adds the other ranks')
      }
    } else {
      h.prop_kind = PROP_NONE;
      h.finishing = 1;  // no reads left (reorder.h:593-599); applied in phase B
      // (the last-ranked chain finding nothing means the pool is exhausted: k_mg_mark sends the cursor to -1)
      if (gl == 0) P.prop[cid] = ((unsigned long long)PK_NOSEED << 32) | (is_last ? PK_CURSOR_BIT : 0ull);
    }
    store_hot(c, h, gl, 2, 4);
    PT(23);
    return;
  }
  if (!h.retrying) {  // iteration start bookkeeping (reorder.h:433-439), once per iteration
    if (h.num_reads_thr % 1000000u == 0) {
      if ((float)h.num_unmatched_past > 0.5f * 1000000) h.stop_searching = 1;
      h.num_unmatched_past = 0;
    }
    h.num_reads_thr++;
  }
  if (h.stop_searching) {
    h.prop_kind = PROP_NONE;
    store_hot(c, h, gl, 2, 4);
    if (gl == 0) P.prop[cid] = ((unsigned long long)PK_NONE << 32) | (h.left_search ? PK_WILLNEED_BIT : 0ull);
    return;
  }
  const int ref_len = h.ref_len;
  const int wide = (h.prev_unmatched && P.seed_wide) ? 1 : 0;
  if (!KA && P.tab.minz) minz_mc<NQ, KA>(S, ref_len, gl);  // (batch_mc synchronises before it reads S.mz)
  PT(18);
  int wcode = INF_CODE, t0 = 0;
  uint32_t wrid = 0;
  if constexpr (KA) search_ka<NQ, KA>(P, S, stage, ref_len, wide, lane, gl, wcode, wrid);
  else for (int ph = 0; ph < 6 && t0 < P.maxshift; ph++) {
    const int w = wide ? P.plan[1][ph] : P.plan[0][ph];
    if (w <= 0) break;
    batch_mc<NQ, KA>(P, S, stage, 4 * t0, 4 * (t0 + w), ref_len, lane, gl, wcode, wrid);
    if (ph == 0) PT(19); else if (ph == 1) PT(20); else PT(21);
    t0 += w;
    if (wcode != INF_CODE) break;
  }
  if (!KA && wcode == INF_CODE && t0 < P.maxshift) {
    wave_sync();  // pres
    tail_mc<NQ, KA>(P, S, stage, t0, ref_len, lane, gl, wcode, wrid);
    PT(22);
  }
  if (wcode != INF_CODE) {
    h.prop_rid = wrid;
    h.prop_shift = (uint32_t)(wcode >> 2);
    h.prop_rev = (uint32_t)(((wcode >> 1) & 1) | ((wcode & 1) << 1));  // rev | dict << 1
    h.prop_kind = PROP_MATCH;
    if (gl == 0) {
      P.prop[cid] = ((unsigned long long)PK_MATCH << 32) | wrid;
      atomicMin(&P.resv[wrid], cid);
    }
  } else {
    h.prop_kind = PROP_NONE;
    // a failed left search sends the chain for a new seed (apply step): k_mg_mark puts it on the needy bitmap
    if (gl == 0) P.prop[cid] = ((unsigned long long)PK_NONE << 32) | (h.left_search ? PK_WILLNEED_BIT : 0ull);
  }
  store_hot(c, h, gl, 2, 4);
}

}  // namespace mc

// NQ = position quads per lane: 4 * 16 * NQ >= Lpad (3: reads up to 192 bases, 8: up to 511)
#ifndef SR_MC_WAVES
#define SR_MC_WAVES 5  // waves per SIMD the kernel is compiled for (512 / SR_MC_WAVES VGPRs)
#endif
template <int NQ, bool MG, bool KA>
__device__ __forceinline__ void round_mc_body(const DevParams &P) {
  static_assert(!KA || NQ <= 3, "known-absent masks: reads up to 192 bases (the spare half of Chain::revref)");
  typedef mc::GLds<NQ, KA> GL;
  __shared__ GL s_g[mc::CPW];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[STAGE_WORDS];  // candidate limbs, one row per lane (cmp_candidate)
  const int lane = threadIdx.x, g = lane >> 4, gl = lane & (mc::G - 1);
#ifdef SR_PHASE_TIMING
  for (int i = lane; i < 66; i += 64) g_pt_lds[i] = 0;
  wave_sync();
  if (lane == 0) g_pt_lds[64] = (uint32_t)clock64();
  wave_sync();
#endif
  // this wavefront's chains: four entries of one class of one list segment (classes in the order 0, 1, 2, 3).
  // Tried and dropped (profiles/r03_experiments.txt): one global class-major list of wavefront-loads built by an extra
  // one-block kernel (longest classes dispatched first: 419 vs 417 ms, nothing), with empty placeholder blocks instead
  // of the list (each empty block costs ~1.3 ns of dispatch: 547 ms), and resident wavefronts looping over the list
  // (the loop costs the body 30 VGPRs, an occupancy step: 487-710 ms).
  uint32_t li;
  {
    const uint32_t seg = (P.c0 + P.g0) / MARK_BLOCK + blockIdx.x / MC_WAVES_PER_BLOCK, b = blockIdx.x % MC_WAVES_PER_BLOCK;
    const uint4 n = P.ord_cnt[seg];
    const uint32_t w0 = (n.x + 3) / 4, w1 = w0 + (n.y + 3) / 4, w2 = w1 + (n.z + 3) / 4, w3 = w2 + (n.w + 3) / 4;
    if (b >= w3) return;
    const uint32_t cls = b < w0 ? 0u : b < w1 ? 1u : b < w2 ? 2u : 3u;
    const uint32_t wbase = cls == 0 ? 0u : cls == 1 ? w0 : cls == 2 ? w1 : w2;
    const uint32_t cnt = cls == 0 ? n.x : cls == 1 ? n.y : cls == 2 ? n.z : n.w;
    const uint32_t lbase = cls == 0 ? 0u : cls == 1 ? n.x : cls == 2 ? n.x + n.y : n.x + n.y + n.z;
    const uint32_t j = (b - wbase) * mc::CPW + (uint32_t)g;
    if (j >= cnt) return;  // (whole groups leave; the others never wait for them: no block barriers below)
    li = P.ord[(size_t)seg * MARK_BLOCK + lbase + j];
  }
  const uint32_t cid = P.c0 + li;
  Chain *c = &P.chains[li];
  GL &S = s_g[g];
  ChainHot h;
  {
    const uint4 *hq = reinterpret_cast<const uint4 *>(&c->h);
    const uint4 q0 = hq[0], q1 = hq[1], q2 = hq[2], q3 = hq[3];
    if constexpr (KA) {  // the chain's known-absent masks: four limbs of either strand (commit_consensus_ka); the rest is zero
      static_assert(GL::LIMBS <= mc::G, "one limb per lane");
      const uint64_t kav = gl < 8 ? c->revref[8 + gl] : 0ull;
      if (gl < GL::LIMBS) { S.ka[0][gl] = 0ull; S.ka[1][gl] = 0ull; }
      wave_sync();
      if (gl < 8) S.ka[gl >> 2][GL::PAD + (gl < 4 ? P.ka_lo : 0) + (gl & 3)] = kav;
    }
    for (int i = gl; i < GL::LIMBS; i += mc::G) {
      const int k = i - GL::PAD;
      const bool in = k >= 0 && k < P.W;
      S.refs[0][i] = in ? c->ref[k] : 0ull;
      S.refs[1][i] = in ? c->revref[k] : 0ull;
    }
    if (gl < 2) S.rd[gl ? GL::WMAX + 1 : 0] = 0ull;
    h.ref_pos = (long long)(((unsigned long long)q0.y << 32) | q0.x);
    h.ref_len = (int32_t)q0.z; h.e_slot = q0.w;
    h.prev = q1.x; h.first_rid = q1.y; h.n_emit = q1.z; h.n_single = q1.w;
    h.s_slot = q2.x; h.num_reads_thr = q2.y; h.num_unmatched_past = q2.z; h.prop_rid = q2.w;
    h.flags = q3.x;
    h.alt1 = 0;  // (no alternatives in this kernel)
    h.pad[0] = h.pad[1] = 0;
  }
  wave_sync();
  if (h.done) {
    if (gl == 0) P.prop[cid] = (unsigned long long)PK_DONE << 32;
    return;
  }
  PTW(16);
  if (!mc::apply_mc<NQ, KA>(P, c, cid, li, h, S, gl)) {
    if (gl == 0) P.prop[cid] = (unsigned long long)PK_DONE << 32;
    return;
  }
  PT(17);
  mc::search_mc<NQ, MG, KA>(P, c, cid, h, S, (lds_u32_t *)s_stage, lane, gl);
  if constexpr (KA) if (h.mode == MODE_SEARCH) {  // what the chain knows now, for its next round
    wave_sync();
    if (gl < 8) c->revref[8 + gl] = S.ka[gl >> 2][GL::PAD + (gl < 4 ? P.ka_lo : 0) + (gl & 3)];
  }
  PTW(24);
#ifdef SR_PHASE_TIMING  // (experiment builds) the wavefront's table goes to the chain of its first lane still here
  {
    wave_sync();
    const unsigned long long ex = __ballot(1);
    if (lane == __ffsll(ex) - 1)
      for (int i = 0; i < 64; i++) c->pt[i] += g_pt_lds[i];
  }
#endif
}
// reads up to 192 bases: 5 waves per SIMD (96 VGPRs); longer reads (eight position quads per lane in the update) spill 14
// VGPRs there and get 4 (128 VGPRs): no chain kernel uses scratch memory
template <int NQ, bool MG, bool KA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SR_MC_WAVES, SR_MC_WAVES))) void k_round_mc(DevParams P) {
  static_assert(NQ <= 3, "long reads: k_round_mc_long");
  round_mc_body<NQ, MG, KA>(P);
}
template <bool MG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_round_mc_long(DevParams P) {
  round_mc_body<8, MG, false>(P);
}

#endif
