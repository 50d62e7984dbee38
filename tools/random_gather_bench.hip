// tools/random_gather_bench.hip -- ceiling for the search kernel's access pattern on MI355X:
// every lane reads one random 64-byte bucket (4 x 16 B loads) per iteration from a table of
// `gb` GiB.  Prints achieved sectors/s and GB/s.  hipcc --offload-arch=gfx950 -O3 -o rgb tools/random_gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
template <int BYTES, int DEP>
__global__ __launch_bounds__(256) void gather(const uint64_t* __restrict__ tab, uint64_t bmask, int iters, uint64_t* out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = 0, key = t * 0x9E3779B97F4A7C15ull + 1;
  for (int i = 0; i < iters; i++) {
    key = mix64(key + i + (DEP ? acc : 0));   // DEP=1: next address depends on the loaded data (dependent hops)
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(tab + (key & bmask) * 8);
    if (BYTES >= 16) { ulonglong2 a = p[0]; acc += a.x ^ a.y; }
    if (BYTES >= 32) { ulonglong2 a = p[1]; acc += a.x ^ a.y; }
    if (BYTES >= 64) { ulonglong2 a = p[2], b = p[3]; acc += a.x ^ a.y ^ b.x ^ b.y; }
  }
  if (acc == 0x1234567) out[0] = acc;
}
int main(int argc, char** argv) {
  std::vector<double> sizes = {0.25, 1.0, 4.0, 8.0, 16.0};
  if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; i++) sizes.push_back(atof(argv[i])); }  // GiB
  for (double gb : sizes) {
    uint64_t nb = 1; while (nb * 64 * 2 <= (uint64_t)(gb * (1ull << 30))) nb <<= 1;
    uint64_t* tab; uint64_t* out;
    if (hipMalloc(&tab, nb * 64) != hipSuccess) { printf("alloc fail %.2f\n", gb); continue; }
    hipMalloc(&out, 8);
    hipMemset(tab, 1, nb * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 64;
    auto run = [&](auto kern, const char* name, int bytes) {
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, nb - 1, iters, out);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, nb - 1, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double acc = (double)blocks * 256 * iters;
      printf("table %6.2f GiB  %-14s %8.3f ms  %7.2f G accesses/s  %8.1f GB/s useful  %8.1f GB/s sectors\n", nb * 64.0 / (1ull << 30), name, ms,
             acc / ms / 1e6, acc * bytes / ms / 1e6, acc * 64 / ms / 1e6);
    };
    run(gather<64, 0>, "64B indep", 64);
    run(gather<16, 0>, "16B indep", 16);
    run(gather<64, 1>, "64B dependent", 64);
    hipFree(tab); hipFree(out);
  }
  return 0;
}
