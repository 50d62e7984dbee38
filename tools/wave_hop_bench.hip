// tools/wave_hop_bench.hip -- what does ONE lock-step round of the search cost when the kernel is reduced to its
// memory-access skeleton?  One wavefront per "chain", K chains per launch, every chain does the dependent hops a
// k_search wave does and nothing else:
//   hop 0: its 384-byte state line group (header + ref + revref) from a K x 384 B array
//   "success" wave: `lanes0` lanes fetch one random 32-byte bucket each (table of tab_gib GiB), then `ncand`
//                   lanes fetch one random 64-byte read (reads array of rd_gib GiB), then lane 0 writes 16 bytes
//   "fail" wave   : nfail dependent bucket hops of 64 lanes each, then lane 0 writes
// A fraction fail_pct of the waves are fail waves (by chain id, spread evenly).  Prints the launch time for
// block sizes of 1 and 4 waves.  It is the floor the real kernel can approach on this memory system.
// hipcc --offload-arch=gfx950 -O3 -o tools/whb tools/wave_hop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
struct Args {
  const uint4 *state; const uint4 *tab; uint64_t bmask; const uint4 *reads; uint64_t rmask; uint4 *out;
  uint32_t K; int lanes0, ncand, nfail, fail_pct; uint32_t salt; int tags_only;  // tags_only: fetch 16 of the bucket's 32 bytes (the payload half only for ~1 lane in 32)
};
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_round(Args a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t c = blockIdx.x * WPB + wave;
  if (c >= a.K) return;
  // hop 0: state (24 x 16 B per chain)
  uint4 h = make_uint4(0, 0, 0, 0);
  if (lane < 24) h = a.state[(uint64_t)c * 24 + lane];
  uint64_t x = mix64(((uint64_t)c << 32) ^ a.salt ^ h.x) + (uint64_t)lane * 0x9E3779B97F4A7C15ull;
  const bool fail = (int)((c * 2654435761u) % 100u) < a.fail_pct;
  uint32_t acc = h.y;
  if (!fail) {
    if (lane < a.lanes0) {
      x = mix64(x);
      if (!a.tags_only) {
        const uint4 t = a.tab[(x & a.bmask) * 2], p = a.tab[(x & a.bmask) * 2 + 1];
        acc ^= t.x ^ t.w ^ p.y;
      } else {
        const uint4 t = a.tab[(x & a.bmask) * 2];
        acc ^= t.x ^ t.w;
        if ((lane & 31) == (int)(t.x & 31)) acc ^= reinterpret_cast<const uint32_t *>(a.tab)[(x & a.bmask) * 8 + 4 + (t.y & 3)];
      }
    }
    // the hit lanes (lowest ncand) fetch a candidate read (64 B)
    uint32_t any = __builtin_amdgcn_readfirstlane(acc);
    if (lane < a.ncand) {
      x = mix64(x ^ acc ^ any);
      const uint4 *r = a.reads + (x & a.rmask) * 4;
      const uint4 r0 = r[0], r1 = r[1], r2 = r[2];
      acc ^= r0.x ^ r1.y ^ r2.z;
    }
  } else {
    for (int b = 0; b < a.nfail; b++) {
      x = mix64(x ^ acc);
      if (!a.tags_only) {
        const uint4 t = a.tab[(x & a.bmask) * 2], p = a.tab[(x & a.bmask) * 2 + 1];
        acc ^= t.x ^ t.w ^ p.y;
      } else {
        const uint4 t = a.tab[(x & a.bmask) * 2];
        acc ^= t.x ^ t.w;
        if ((lane & 31) == (int)(t.x & 31)) acc ^= reinterpret_cast<const uint32_t *>(a.tab)[(x & a.bmask) * 8 + 4 + (t.y & 3)];
      }
      acc ^= (uint32_t)__popcll(__ballot(acc & 1));  // wave-level dependency like the hit ballot
    }
  }
  acc ^= (uint32_t)__popcll(__ballot(acc & 1));
  if (lane == 0) a.out[c] = make_uint4(acc, (uint32_t)x, 0, 0);
}
int main(int argc, char **argv) {
  const double tab_gib = argc > 1 ? atof(argv[1]) : 8.0, rd_gib = argc > 2 ? atof(argv[2]) : 4.0;
  const uint32_t K = argc > 3 ? (uint32_t)atoi(argv[3]) : 65536;
  uint64_t nb = 1; while (nb * 32 * 2 <= (uint64_t)(tab_gib * (1ull << 30))) nb <<= 1;
  uint64_t nr = 1; while (nr * 64 * 2 <= (uint64_t)(rd_gib * (1ull << 30))) nr <<= 1;
  Args a; uint4 *state, *tab, *reads, *out;
  hipMalloc(&state, (size_t)K * 384); hipMalloc(&tab, nb * 32); hipMalloc(&reads, nr * 64); hipMalloc(&out, (size_t)K * 16);
  hipMemset(state, 1, (size_t)K * 384); hipMemset(tab, 2, nb * 32); hipMemset(reads, 3, nr * 64);
  a.state = state; a.tab = tab; a.bmask = nb - 1; a.reads = reads; a.rmask = nr - 1; a.out = out; a.K = K;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("table %.1f GiB (32-byte buckets), reads %.1f GiB (64-byte), K=%u chains\n", nb * 32.0 / (1 << 30), nr * 64.0 / (1 << 30), K);
  struct Cfg { const char *name; int lanes0, ncand, nfail, fail_pct; } cfgs[] = {
      {"state only", 0, 0, 0, 0},
      {"success: 32 buckets", 32, 0, 0, 0},
      {"success: 32 buckets + 1 read", 32, 1, 0, 0},
      {"success: 32 buckets + 3 reads", 32, 3, 0, 0},
      {"success: 16 buckets + 1 read", 16, 1, 0, 0},
      {"success: 64 buckets + 1 read", 64, 1, 0, 0},
      {"all fail: 3 x 64 buckets", 0, 0, 3, 100},
      {"all fail: 6 x 64 buckets", 0, 0, 6, 100},
      {"mix 78% success(32+2) / 22% fail(3x64)", 32, 2, 3, 22},
      {"mix 78% success(32+2) / 22% fail(6x64)", 32, 2, 6, 22},
  };
  for (int tags = 0; tags < 2; tags++)
  for (auto &c : cfgs) {
    a.tags_only = tags;
    a.lanes0 = c.lanes0; a.ncand = c.ncand; a.nfail = c.nfail; a.fail_pct = c.fail_pct;
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 6; rep++) {
      for (int v = 0; v < 2; v++) {
        a.salt = 17 * rep + v;
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(k_round<1>, dim3(K), dim3(64), 0, 0, a);
        else hipLaunchKernelGGL(k_round<4>, dim3((K + 3) / 4), dim3(256), 0, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best[v]) best[v] = ms;
      }
    }
    const double req = (double)K * ((100 - c.fail_pct) / 100.0 * (c.lanes0 + c.ncand) + c.fail_pct / 100.0 * c.nfail * 64 + 6);
    printf("%s %-44s 1 wave/block %7.1f us   4 waves/block %7.1f us   (%.2f M requests, %.1f G req/s)\n", tags ? "[tags 16B]" : "[t+p 32B] ", c.name, best[0] * 1e3,
           best[1] * 1e3, req / 1e6, req / (best[1] * 1e-3) / 1e9);
    (void)0;
  }
  return 0;
}
