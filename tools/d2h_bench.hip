// tools/d2h_bench.hip -- how fast do device -> pinned host copies go on this box: hipMemcpyAsync in slots of 8 / 32 MiB over
// 1 / 2 / 4 streams, and a copy kernel that stores straight into the mapped pinned memory.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/d2h_bench tools/d2h_bench.hip ; run: /tmp/d2h_bench [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_copy(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = s[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t total = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
  uint8_t *d = nullptr, *h = nullptr;
  CK(hipMalloc(&d, total));
  CK(hipMemset(d, 1, total));
  CK(hipHostMalloc(&h, total, hipHostMallocDefault));
  for (size_t i = 0; i < total; i += 4096) h[i] = 0;
  hipStream_t st[8];
  for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int dir = 0; dir < 2; dir++)
    for (size_t slot : {(size_t)8 << 20, (size_t)32 << 20, (size_t)256 << 20})
      for (int ns : {1, 2, 4, 8}) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        int k = 0;
        for (size_t off = 0; off < total; off += slot, k++) {
          const size_t len = std::min(slot, total - off);
          if (dir == 0) CK(hipMemcpyAsync(h + off, d + off, len, hipMemcpyDeviceToHost, st[k % ns]));
          else CK(hipMemcpyAsync(d + off, h + off, len, hipMemcpyHostToDevice, st[k % ns]));
        }
        for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(st[i]));
        const double t = now() - t0;
        printf("%s hipMemcpyAsync slot %4zu MiB, %d streams: %6.1f GB/s\n", dir ? "H2D" : "D2H", slot >> 20, ns, total / t / 1e9);
      }
  uint8_t *hd = nullptr;
  CK(hipHostGetDevicePointer((void **)&hd, h, 0));
  for (int blocks : {256, 1024, 4096})
    for (int rep = 0; rep < 2; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st[0], (const uint4 *)d, (uint4 *)hd, total / 16);
      CK(hipStreamSynchronize(st[0]));
      const double t = now() - t0;
      printf("D2H copy kernel, %d blocks: %6.1f GB/s\n", blocks, total / t / 1e9);
    }
  // the host side alone: memcpy pinned -> pageable with T threads is not measured here (see tools/files_probe.py)
  return 0;
}
