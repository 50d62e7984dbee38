// tools/numa_d2h_bench.hip -- does it matter on which NUMA node of the host a pinned buffer lives?  Pinned buffers are
// allocated by a thread bound to a CPU of node 0 / node 1 (first CPUs of the nodes given on the command line), then copied to
// from the device in 8 MiB pieces.  build: hipcc --offload-arch=gfx950 -O3 -pthread -o /tmp/nd tools/numa_d2h_bench.hip
#define _GNU_SOURCE
#include <hip/hip_runtime.h>
#include <sched.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t total = (size_t)2 << 30, slot = (size_t)8 << 20;
  uint8_t *d = nullptr;
  CK(hipMalloc(&d, total));
  CK(hipMemset(d, 1, total));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int numa = -1;
  (void)hipDeviceGetAttribute(&numa, hipDeviceAttributeHostNumaId, 0);
  printf("hipDeviceAttributeHostNumaId = %d\n", numa);
  for (int a = 1; a < argc; a++) {
    const int cpu = atoi(argv[a]);
    uint8_t *h = nullptr;
    std::thread([&] {
      cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpu, &set);
      if (sched_setaffinity(0, sizeof(set), &set)) perror("sched_setaffinity");
      CK(hipSetDevice(0));
      CK(hipHostMalloc(&h, total, hipHostMallocDefault));
      for (size_t i = 0; i < total; i += 4096) h[i] = 0;
    }).join();
    for (int dir = 0; dir < 2; dir++)
      for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (size_t off = 0; off < total; off += slot) {
          if (dir == 0) CK(hipMemcpyAsync(h + off, d + off, slot, hipMemcpyDeviceToHost, st));
          else CK(hipMemcpyAsync(d + off, h + off, slot, hipMemcpyHostToDevice, st));
        }
        CK(hipStreamSynchronize(st));
        printf("pinned buffer allocated on cpu %3d: %s %6.1f GB/s\n", cpu, dir ? "H2D" : "D2H", total / (now() - t0) / 1e9);
      }
    CK(hipHostFree(h));
  }
  return 0;
}
