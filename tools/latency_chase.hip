// tools/latency_chase.hip -- dependent-load latency on MI355X: one wave, each lane chases its own chain.
// mode 0: random over the whole table; mode 1: random inside a 2 MiB window per lane (TLB-friendly).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void chase(const uint64_t* __restrict__ tab, uint64_t nslots, int iters, int mode, int lanes, uint64_t* out) {
  if ((int)threadIdx.x >= lanes) return;
  uint64_t win = (2ull << 20) / 8; if (win > nslots) win = nslots; const uint64_t base = (mix64(threadIdx.x + 17) % (nslots / win)) * win;
  uint64_t idx = mix64(threadIdx.x + 1) % nslots, acc = 0;
  for (int i = 0; i < iters; i++) {
    uint64_t v = tab[idx];
    acc += v;
    uint64_t h = mix64(v + idx + i);
    idx = mode ? base + (h % win) : (h % nslots);
  }
  out[threadIdx.x] = acc;
}
int main() {
  for (double gb : {0.004, 0.03, 0.2, 4.0, 16.0}) {
    uint64_t nslots = (uint64_t)(gb * (1ull << 30)) / 8;
    uint64_t* tab; uint64_t* out; hipMalloc(&tab, nslots * 8); hipMalloc(&out, 64 * 8); hipMemset(tab, 1, nslots * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) for (int lanes : {1, 64}) {
      const int iters = 20000;
      hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, tab, nslots, 2000, mode, lanes, out);
      hipEventRecord(e0);
      hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, tab, nslots, iters, mode, lanes, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("table %7.3f GiB  mode %s  lanes %2d : %7.1f ns per dependent load\n", gb, mode ? "2MiB-window" : "whole-table", lanes, ms * 1e6 / iters);
    }
    hipFree(tab); hipFree(out);
  }
  return 0;
}
