// spring_amd/csrc/fastq_reorder.hip -- SURVEY 8(f4): reorder-only output.
//
// Writes the 4-line records of a FASTQ text in a given order: output record k = input record order[k].
// With order = the encoder stage's read_order.bin (single-end) this is the read order the reference's
// decompressor emits without --preserve-order (aligned reads contig by contig, then the unaligned ones),
// so the accelerated stages can be used on their own as a "reorder the FASTQ" tool and checked externally
// (same multiset of records, clustered by similarity).  Newline index -> record sizes gathered by `order`
// -> exclusive scan -> one wavefront copies one record.  Host text in, host text out.
#include <cstring>

#include <hip/hip_runtime.h>

#include "reorder_device.h"
#include "reorder_internal.h"
#include "spring_reorder.h"

using sr::fail;

#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess)                                                                      \
      return fail(SPRING_REORDER_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

namespace {

// bytes [rec_start, rec_end) of record i (4 lines, the final '\n' included when present)
__device__ __forceinline__ void rec_span(const uint64_t *__restrict__ line_end, uint64_t i, uint64_t nbytes,
                                         uint64_t &a, uint64_t &b) {
  a = i ? line_end[4 * i - 1] + 1 : 0;
  const uint64_t e = line_end[4 * i + 3];  // position of the '\n' that ends the record (== nbytes if missing)
  b = e < nbytes ? e + 1 : nbytes;
}
__global__ void k_rec_sizes(const uint64_t *__restrict__ line_end, const uint32_t *__restrict__ order, uint32_t n,
                            uint64_t nrec, uint64_t nbytes, uint32_t *__restrict__ sz, uint32_t *__restrict__ err) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n) return;
  if (k == n) { sz[k] = 0; return; }
  const uint64_t i = order[k];
  if (i >= nrec) { *err = 1; sz[k] = 0; return; }
  uint64_t a, b;
  rec_span(line_end, i, nbytes, a, b);
  const bool unterminated = line_end[4 * i + 3] >= nbytes;  // gets its '\n' in the output
  const uint64_t s = b - a + (unterminated ? 1 : 0);
  if (s > 0xffffffffull) { *err = 2; sz[k] = 0; return; }
  sz[k] = (uint32_t)s;
}
__global__ __launch_bounds__(256) void k_copy_records(const uint8_t *__restrict__ txt, const uint64_t *__restrict__ line_end,
                                                      const uint32_t *__restrict__ order, uint32_t n, uint64_t nrec,
                                                      uint64_t nbytes, const uint64_t *__restrict__ off,
                                                      uint8_t *__restrict__ out) {
  const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= n) return;
  const uint64_t i = order[k];
  if (i >= nrec) return;
  uint64_t a, b;
  rec_span(line_end, i, nbytes, a, b);
  uint8_t *dst = out + off[k];
  for (uint64_t p = a + lane; p < b; p += 64) dst[p - a] = txt[p];
  if (lane == 0 && line_end[4 * i + 3] >= nbytes) dst[b - a] = '\n';
}

struct DBuf {
  int dev = 0;
  void *p = nullptr;
  ~DBuf() { if (p) sr::dev_free(dev, p); }
  hipError_t alloc(int d, size_t bytes) { dev = d; return sr::dev_alloc(d, bytes, &p); }
  template <class T> T *as() const { return (T *)p; }
};

}  // namespace

extern "C" int spring_fastq_reorder(const uint8_t *fastq, size_t nbytes, const uint32_t *order, uint32_t n,
                                    uint8_t *out, size_t out_cap, size_t *out_bytes, double *kernel_ms) {
  if ((nbytes && !fastq) || (n && !order) || !out_bytes) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  *out_bytes = 0;
  if (kernel_ms) *kernel_ms = 0;
  if (!nbytes || !n) return n && !nbytes ? fail(SPRING_REORDER_E_ARG, "order refers to an empty FASTQ") : 0;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  hipStream_t st = nullptr;
  const uint64_t nblk = (nbytes + sr::NL_CHUNK_BYTES - 1) / sr::NL_CHUNK_BYTES;
  DBuf txt, blk_cnt, blk_off, tmp, dorder, sz, off, derr, dout, le;
  HIPCHK(txt.alloc(dev, nbytes + 16)); HIPCHK(blk_cnt.alloc(dev, nblk * 4)); HIPCHK(blk_off.alloc(dev, nblk * 8));
  HIPCHK(dorder.alloc(dev, (size_t)n * 4)); HIPCHK(sz.alloc(dev, ((size_t)n + 1) * 4)); HIPCHK(off.alloc(dev, ((size_t)n + 1) * 8));
  HIPCHK(derr.alloc(dev, 16));
  size_t tb = 0, t2 = 0;
  HIPCHK(sr::excl_scan_u32_to_u64(st, nullptr, tb, nullptr, nullptr, nblk));
  HIPCHK(sr::excl_scan_u32_to_u64(st, nullptr, t2, nullptr, nullptr, (size_t)n + 1));
  tb = tb > t2 ? tb : t2;
  HIPCHK(tmp.alloc(dev, tb + 16));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  struct Ev { hipEvent_t a, b; ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
  HIPCHK(hipMemcpyAsync(txt.p, fastq, nbytes, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dorder.p, order, (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(derr.p, 0, 4, st));
  HIPCHK(hipEventRecord(e0, st));
  sr::launch_nl_count(st, txt.as<uint8_t>(), nbytes, blk_cnt.as<uint32_t>(), nblk);
  t2 = tb;
  HIPCHK(sr::excl_scan_u32_to_u64(st, tmp.p, t2, blk_cnt.as<uint32_t>(), blk_off.as<uint64_t>(), nblk));
  uint64_t last_off = 0;
  uint32_t last_cnt = 0;
  HIPCHK(hipMemcpyAsync(&last_off, blk_off.as<uint64_t>() + (nblk - 1), 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_cnt, blk_cnt.as<uint32_t>() + (nblk - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint64_t nl = last_off + last_cnt;
  const bool unterminated = fastq[nbytes - 1] != '\n';
  const uint64_t nlines = nl + (unterminated ? 1 : 0);
  if (nlines % 4) return fail(SPRING_REORDER_E_ARG, "Invalid FASTQ(A) file. Number of lines not multiple of 4(2)");
  const uint64_t nrec = nlines / 4;
  HIPCHK(le.alloc(dev, (nlines + 1) * 8));
  sr::launch_nl_fill(st, txt.as<uint8_t>(), nbytes, blk_off.as<uint64_t>(), le.as<uint64_t>(), nblk);
  if (unterminated) {
    const uint64_t e = nbytes;
    HIPCHK(hipMemcpyAsync(le.as<uint64_t>() + nl, &e, 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // `e` lives on this block's stack
  }
  hipLaunchKernelGGL(k_rec_sizes, dim3((n + 1 + 255) / 256), dim3(256), 0, st, le.as<uint64_t>(), dorder.as<uint32_t>(), n,
                     nrec, (uint64_t)nbytes, sz.as<uint32_t>(), derr.as<uint32_t>());
  t2 = tb;
  HIPCHK(sr::excl_scan_u32_to_u64(st, tmp.p, t2, sz.as<uint32_t>(), off.as<uint64_t>(), (size_t)n + 1));
  uint64_t total = 0;
  uint32_t err = 0;
  HIPCHK(hipMemcpyAsync(&total, off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&err, derr.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (err) return fail(SPRING_REORDER_E_ARG, err == 1 ? "order refers to a record past the end of the FASTQ" : "record too long");
  *out_bytes = total;
  if (!out || out_cap < total) return out ? fail(SPRING_REORDER_E_ARG, "output buffer too small (%zu < %llu)", out_cap,
                                                 (unsigned long long)total) : 0;  // out == NULL: size query
  HIPCHK(dout.alloc(dev, total + 16));
  hipLaunchKernelGGL(k_copy_records, dim3((n + 3) / 4), dim3(256), 0, st, txt.as<uint8_t>(), le.as<uint64_t>(),
                     dorder.as<uint32_t>(), n, nrec, (uint64_t)nbytes, off.as<uint64_t>(), dout.as<uint8_t>());
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipMemcpyAsync(out, dout.p, total, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  if (kernel_ms) *kernel_ms = ms;
  return 0;
}
