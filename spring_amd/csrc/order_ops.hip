// spring_amd/csrc/order_ops.hip -- SURVEY 8(f3): the consumers of read_order.bin that are pure
// permutation / prefix-sum work, on the GPU.
//   generate_order_se / generate_order_pe   reference src/reorder_compress_quality_id.cpp:101-125
//   correct_order                           reference src/encoder.cpp:177-222
// Host arrays in, host arrays out (the library owns the device buffers); kernel_ms reports the
// device time of the kernels alone (HIP events), for bench/roofline purposes.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "reorder_device.h"
#include "spring_reorder.h"

namespace sr {
int fail(int code, const char *fmt, ...);
}
using sr::fail;

#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess)                                                                      \
      return fail(SPRING_REORDER_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

namespace {

// order_array[order[i]] = i  (generate_order_se)
__global__ void k_invert_se(const uint32_t *__restrict__ order, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[order[i]] = i;
}
__global__ void k_flag_lt(const uint32_t *__restrict__ order, uint32_t n, uint32_t half, uint32_t *__restrict__ f) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] = order[i] < half ? 1u : 0u;
}
// if (order < n/2) order_array[order] = pos_after_reordering++  (generate_order_pe); pos = exclusive scan
__global__ void k_invert_pe(const uint32_t *__restrict__ order, const uint32_t *__restrict__ pos, uint32_t n,
                            uint32_t half, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && order[i] < half) out[order[i]] = pos[i];
}
__global__ void k_mark_N(const uint32_t *__restrict__ order_N, uint32_t nN, uint32_t *__restrict__ flag) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nN) flag[order_N[i]] = 1u;
}
// cumulative_N_reads[pos_in_clean] = number of N reads before that read in the original file
__global__ void k_cumulative(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ nbefore, uint32_t total,
                             uint32_t *__restrict__ cum) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total && !flag[i]) cum[i - nbefore[i]] = nbefore[i];
}
__global__ void k_apply_cum(uint32_t *__restrict__ order, uint64_t m, const uint32_t *__restrict__ cum) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) order[i] += cum[order[i]];
}

// pe_encode (pe_encode.cpp:51-70): rank1 = exclusive scan of "is a file-1 read" over the reordered file
__global__ void k_pe_encode(const uint32_t *__restrict__ order, const uint32_t *__restrict__ inv,
                            const uint32_t *__restrict__ rank1, uint32_t n, uint32_t half, uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = order[i];
  out[i] = o < half ? rank1[i] : rank1[inv[o - half]] + half;
}

struct Buf {
  void *p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t b) { return hipMalloc(&p, b ? b : 16); }
  template <class T> T *as() { return (T *)p; }
};
struct Ev {
  hipEvent_t a = nullptr, b = nullptr;
  ~Ev() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
inline dim3 grid(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" {

int spring_order_invert_se(const uint32_t *order, uint32_t n, uint32_t *order_array, double *kernel_ms) {
  if (n && (!order || !order_array)) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  if (kernel_ms) *kernel_ms = 0;
  if (!n) return 0;
  Buf din, dout; Ev ev;
  HIPCHK(din.alloc((size_t)n * 4)); HIPCHK(dout.alloc((size_t)n * 4));
  HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b));
  HIPCHK(hipMemcpy(din.p, order, (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipEventRecord(ev.a, nullptr));
  hipLaunchKernelGGL(k_invert_se, grid(n), dim3(256), 0, nullptr, din.as<uint32_t>(), n, dout.as<uint32_t>());
  HIPCHK(hipEventRecord(ev.b, nullptr));
  HIPCHK(hipMemcpy(order_array, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipEventSynchronize(ev.b));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
  if (kernel_ms) *kernel_ms = ms;
  return 0;
}

int spring_order_invert_pe(const uint32_t *order, uint32_t n, uint32_t *order_array, double *kernel_ms) {
  if (n && (!order || !order_array)) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  if (kernel_ms) *kernel_ms = 0;
  if (!n) return 0;
  const uint32_t half = n / 2;
  Buf din, dflag, dpos, dout, dtmp; Ev ev;
  HIPCHK(din.alloc((size_t)n * 4)); HIPCHK(dflag.alloc((size_t)n * 4)); HIPCHK(dpos.alloc((size_t)n * 4));
  HIPCHK(dout.alloc((size_t)half * 4));
  HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b));
  HIPCHK(hipMemcpy(din.p, order, (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dout.p, 0, (size_t)(half ? half : 1) * 4));
  size_t tb = 0;
  HIPCHK(sr::excl_scan_u32(nullptr, nullptr, tb, dflag.as<uint32_t>(), dpos.as<uint32_t>(), n));
  HIPCHK(dtmp.alloc(tb));
  HIPCHK(hipEventRecord(ev.a, nullptr));
  hipLaunchKernelGGL(k_flag_lt, grid(n), dim3(256), 0, nullptr, din.as<uint32_t>(), n, half, dflag.as<uint32_t>());
  HIPCHK(sr::excl_scan_u32(nullptr, dtmp.p, tb, dflag.as<uint32_t>(), dpos.as<uint32_t>(), n));
  hipLaunchKernelGGL(k_invert_pe, grid(n), dim3(256), 0, nullptr, din.as<uint32_t>(), dpos.as<uint32_t>(), n, half,
                     dout.as<uint32_t>());
  HIPCHK(hipEventRecord(ev.b, nullptr));
  HIPCHK(hipMemcpy(order_array, dout.p, (size_t)half * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipEventSynchronize(ev.b));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
  if (kernel_ms) *kernel_ms = ms;
  return 0;
}

int spring_order_correct(uint32_t *order, uint64_t m, const uint32_t *order_N, uint32_t nN, uint32_t n_clean,
                         double *kernel_ms) {
  if ((m && !order) || (nN && !order_N)) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  if (kernel_ms) *kernel_ms = 0;
  const uint64_t total64 = (uint64_t)n_clean + nN;
  if (total64 > 4294967290ull) return fail(SPRING_REORDER_E_ARG, "too many reads");
  const uint32_t total = (uint32_t)total64;
  if (!m || !total) return 0;
  Buf dord, dN, dflag, dnb, dcum, dtmp; Ev ev;
  HIPCHK(dord.alloc(m * 4)); HIPCHK(dN.alloc((size_t)nN * 4)); HIPCHK(dflag.alloc((size_t)total * 4));
  HIPCHK(dnb.alloc((size_t)total * 4)); HIPCHK(dcum.alloc((size_t)(n_clean ? n_clean : 1) * 4));
  HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b));
  HIPCHK(hipMemcpy(dord.p, order, m * 4, hipMemcpyHostToDevice));
  if (nN) HIPCHK(hipMemcpy(dN.p, order_N, (size_t)nN * 4, hipMemcpyHostToDevice));
  size_t tb = 0;
  HIPCHK(sr::excl_scan_u32(nullptr, nullptr, tb, dflag.as<uint32_t>(), dnb.as<uint32_t>(), total));
  HIPCHK(dtmp.alloc(tb));
  HIPCHK(hipEventRecord(ev.a, nullptr));
  HIPCHK(hipMemsetAsync(dflag.p, 0, (size_t)total * 4, nullptr));
  if (nN) hipLaunchKernelGGL(k_mark_N, grid(nN), dim3(256), 0, nullptr, dN.as<uint32_t>(), nN, dflag.as<uint32_t>());
  HIPCHK(sr::excl_scan_u32(nullptr, dtmp.p, tb, dflag.as<uint32_t>(), dnb.as<uint32_t>(), total));
  hipLaunchKernelGGL(k_cumulative, grid(total), dim3(256), 0, nullptr, dflag.as<uint32_t>(), dnb.as<uint32_t>(), total,
                     dcum.as<uint32_t>());
  hipLaunchKernelGGL(k_apply_cum, grid(m), dim3(256), 0, nullptr, dord.as<uint32_t>(), m, dcum.as<uint32_t>());
  HIPCHK(hipEventRecord(ev.b, nullptr));
  HIPCHK(hipMemcpy(order, dord.p, m * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipEventSynchronize(ev.b));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
  if (kernel_ms) *kernel_ms = ms;
  return 0;
}

int spring_order_pe_encode(const uint32_t *order, uint32_t n, uint32_t *new_order, double *kernel_ms) {
  if (n && (!order || !new_order)) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  if (kernel_ms) *kernel_ms = 0;
  if (!n) return 0;
  if (n & 1) return fail(SPRING_REORDER_E_ARG, "pe_encode needs an even number of reads (pairs)");
  const uint32_t half = n / 2;
  Buf din, dinv, dflag, drank, dout, dtmp; Ev ev;
  HIPCHK(din.alloc((size_t)n * 4)); HIPCHK(dinv.alloc((size_t)n * 4)); HIPCHK(dflag.alloc((size_t)n * 4));
  HIPCHK(drank.alloc((size_t)n * 4)); HIPCHK(dout.alloc((size_t)n * 4));
  HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b));
  HIPCHK(hipMemcpy(din.p, order, (size_t)n * 4, hipMemcpyHostToDevice));
  size_t tb = 0;
  HIPCHK(sr::excl_scan_u32(nullptr, nullptr, tb, dflag.as<uint32_t>(), drank.as<uint32_t>(), n));
  HIPCHK(dtmp.alloc(tb));
  HIPCHK(hipEventRecord(ev.a, nullptr));
  hipLaunchKernelGGL(k_invert_se, grid(n), dim3(256), 0, nullptr, din.as<uint32_t>(), n, dinv.as<uint32_t>());
  hipLaunchKernelGGL(k_flag_lt, grid(n), dim3(256), 0, nullptr, din.as<uint32_t>(), n, half, dflag.as<uint32_t>());
  HIPCHK(sr::excl_scan_u32(nullptr, dtmp.p, tb, dflag.as<uint32_t>(), drank.as<uint32_t>(), n));
  hipLaunchKernelGGL(k_pe_encode, grid(n), dim3(256), 0, nullptr, din.as<uint32_t>(), dinv.as<uint32_t>(),
                     drank.as<uint32_t>(), n, half, dout.as<uint32_t>());
  HIPCHK(hipEventRecord(ev.b, nullptr));
  HIPCHK(hipMemcpy(new_order, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipEventSynchronize(ev.b));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
  if (kernel_ms) *kernel_ms = ms;
  return 0;
}

}  // extern "C"
