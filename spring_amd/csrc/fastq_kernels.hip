// spring_amd/csrc/fastq_kernels.hip -- SURVEY 8(f1): the sequence side of SPRING's preprocess stage on the GPU.
// FASTQ text -> read boundaries -> N-split -> 2-bit packed clean reads (input_clean_*.dna records, straight into
// the stream the reorder stage unpacks) + 4-bit packed N reads (input_N.dna) + their file positions
// (read_order_N.bin).  Reference: read_fastq_block (src/util.cpp:31-54), the N-split loop of preprocess
// (src/preprocess.cpp:293-304), write_dna_in_bits / write_dnaN_in_bits (src/util.cpp:269-294, :322-348).
// These are HBM-streaming kernels (bytes in, bytes out), unlike the gather-bound search.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "reorder_device.h"

namespace sr {

constexpr int NL_CHUNK = 4096;  // text bytes per block in the newline passes (256 threads x 16 bytes)

__device__ __forceinline__ int count_nl16(const uint8_t *__restrict__ t, uint64_t base, uint64_t nbytes, uint32_t &mask) {
  mask = 0;
  if (base + 16 <= nbytes) {
    // 16-byte chunks start at multiples of 16 from a hipMalloc'd (256-byte aligned) base
    const uint4 v = *reinterpret_cast<const uint4 *>(t + base);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int b = 0; b < 4; b++)
        if (((w[k] >> (8 * b)) & 0xff) == '\n') mask |= 1u << (4 * k + b);
  } else {
    for (int b = 0; b < 16; b++)
      if (base + b < nbytes && t[base + b] == '\n') mask |= 1u << b;
  }
  return __popc(mask);
}

// pass 1: newlines per NL_CHUNK block
__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *__restrict__ t, uint64_t nbytes, uint32_t *__restrict__ blk_cnt) {
  __shared__ int s[4];
  const uint64_t base = (uint64_t)blockIdx.x * NL_CHUNK + (uint64_t)threadIdx.x * 16;
  uint32_t m;
  int c = base < nbytes ? count_nl16(t, base, nbytes, m) : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = (uint32_t)(s[0] + s[1] + s[2] + s[3]);
}

// pass 2: line_end[k] = byte offset of the k-th '\n'
__global__ __launch_bounds__(256) void k_nl_fill(const uint8_t *__restrict__ t, uint64_t nbytes,
                                                 const uint64_t *__restrict__ blk_off, uint64_t *__restrict__ line_end) {
  __shared__ int s[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * NL_CHUNK + (uint64_t)threadIdx.x * 16;
  uint32_t m = 0;
  const int c = base < nbytes ? count_nl16(t, base, nbytes, m) : 0;
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s[wave] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; w++) wbase += s[w];
  uint64_t k = blk_off[blockIdx.x] + (uint64_t)(wbase + incl - c);
  while (m) {
    const int b = __ffs((int)m) - 1;
    line_end[k++] = base + b;
    m &= m - 1;
  }
}

// 16 lanes per read: geometry of read line (4i+1), length, N test, record sizes
__global__ __launch_bounds__(256) void k_read_info(const uint8_t *__restrict__ t, const uint64_t *__restrict__ line_end,
                                                   uint64_t nreads, uint32_t *__restrict__ len, uint32_t *__restrict__ fclean,
                                                   uint32_t *__restrict__ szc, uint32_t *__restrict__ fN,
                                                   uint32_t *__restrict__ szN, uint32_t *__restrict__ lenc,
                                                   uint32_t *__restrict__ err) {
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l16 = threadIdx.x & 15, grp = (threadIdx.x & 63) >> 4;
  uint64_t s = 0, e = 0;
  if (i < nreads) {
    s = line_end[4 * i] + 1;
    e = line_end[4 * i + 1];
    if (e > s && t[e - 1] == '\r') e--;  // remove_CR_from_end (util.cpp:391-394)
  }
  bool hasN = false, bad = false;
  for (uint64_t p = s + l16; p < e; p += 16) {
    const uint8_t c = t[p];
    hasN |= c == 'N';
    // the reference's tables are only defined for A C G T N (util.cpp:270-274, :328): anything else (lower case,
    // IUPAC codes, '.') would be packed as garbage there; here it is an error
    bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N');
  }
  const uint64_t bal = __ballot(hasN);
  const bool anyN = (bal >> (16 * grp)) & 0xffffull;
  if (__ballot(bad) && bad) atomicOr(err, 2u);
  if (i < nreads && l16 == 0) {
    const uint64_t L = e - s;
    if (L > (uint64_t)MAX_READ_LEN) atomicOr(err, 1u);  // "Too long read length" (preprocess.cpp:190-196)
    const uint32_t L32 = (uint32_t)(L > 0xffffu ? 0xffffu : L);
    len[i] = L32;
    fclean[i] = anyN ? 0u : 1u;
    fN[i] = anyN ? 1u : 0u;
    szc[i] = anyN ? 0u : 2u + (L32 + 3u) / 4u;
    szN[i] = anyN ? 2u + (L32 + 1u) / 2u : 0u;
    lenc[i] = anyN ? 0xffffffffu : L32;  // for the minimum clean length (equal-length fast path)
  }
}

__device__ __forceinline__ uint32_t dna2int(uint8_t c) {  // util.cpp:270-274 (+ N = 4, :328)
  return c == 'A' ? 0u : c == 'G' ? 1u : c == 'C' ? 2u : c == 'T' ? 3u : 4u;
}

// 16 lanes per read: write the record (clean: 2 bits/base; N read: 4 bits/base) at its scanned offset
__global__ __launch_bounds__(256) void k_pack_reads(const uint8_t *__restrict__ t, const uint64_t *__restrict__ line_end,
                                                    uint64_t nreads, const uint32_t *__restrict__ len,
                                                    const uint32_t *__restrict__ fclean, const uint32_t *__restrict__ cidx,
                                                    const uint64_t *__restrict__ coff, const uint32_t *__restrict__ nidx,
                                                    const uint64_t *__restrict__ noff, uint32_t cidx_base,
                                                    uint64_t coff_base, uint32_t file_read_base,
                                                    uint8_t *__restrict__ out_clean, uint64_t *__restrict__ out_off,
                                                    uint8_t *__restrict__ out_N, uint32_t *__restrict__ out_orderN,
                                                    uint64_t *__restrict__ out_offN) {
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l16 = threadIdx.x & 15;
  if (i >= nreads) return;
  const uint64_t s = line_end[4 * i] + 1;
  const uint32_t L = len[i];
  if (fclean[i]) {
    const uint64_t o = coff_base + coff[i];
    uint8_t *dst = out_clean + o;
    if (l16 == 0) {
      dst[0] = (uint8_t)(L & 0xff); dst[1] = (uint8_t)(L >> 8);
      out_off[cidx_base + cidx[i]] = o;
    }
    const uint32_t nb = (L + 3) / 4;
    for (uint32_t b = l16; b < nb; b += 16) {
      uint32_t v = 0;
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t j = 4 * b + q;
        if (j < L) v |= (dna2int(t[s + j]) & 3u) << (2 * q);
      }
      dst[2 + b] = (uint8_t)v;
    }
  } else {
    uint8_t *dst = out_N + noff[i];
    if (l16 == 0) {
      dst[0] = (uint8_t)(L & 0xff); dst[1] = (uint8_t)(L >> 8);
      out_orderN[nidx[i]] = file_read_base + (uint32_t)i;  // pos_N = num_reads[j] + i (preprocess.cpp:299)
      out_offN[nidx[i]] = noff[i];                         // record offsets, for consumers that stay on the device
    }
    const uint32_t nb = (L + 1) / 2;
    for (uint32_t b = l16; b < nb; b += 16) {
      uint32_t v = dna2int(t[s + 2 * b]);
      if (2 * b + 1 < L) v |= dna2int(t[s + 2 * b + 1]) << 4;
      dst[2 + b] = (uint8_t)v;
    }
  }
}

#define GRIDN(n, per) dim3((unsigned)(((uint64_t)(n) + (per) - 1) / (per)))

void launch_nl_count(hipStream_t st, const uint8_t *t, uint64_t nbytes, uint32_t *blk_cnt, uint64_t nblk) {
  if (nblk) hipLaunchKernelGGL(k_nl_count, dim3((unsigned)nblk), dim3(256), 0, st, t, nbytes, blk_cnt);
}
void launch_nl_fill(hipStream_t st, const uint8_t *t, uint64_t nbytes, const uint64_t *blk_off, uint64_t *line_end,
                    uint64_t nblk) {
  if (nblk) hipLaunchKernelGGL(k_nl_fill, dim3((unsigned)nblk), dim3(256), 0, st, t, nbytes, blk_off, line_end);
}
void launch_read_info(hipStream_t st, const uint8_t *t, const uint64_t *line_end, uint64_t nreads, uint32_t *len,
                      uint32_t *fclean, uint32_t *szc, uint32_t *fN, uint32_t *szN, uint32_t *lenc, uint32_t *err) {
  if (nreads) hipLaunchKernelGGL(k_read_info, GRIDN(nreads, 16), dim3(256), 0, st, t, line_end, nreads, len, fclean, szc, fN, szN, lenc, err);
}
void launch_pack_reads(hipStream_t st, const uint8_t *t, const uint64_t *line_end, uint64_t nreads, const uint32_t *len,
                       const uint32_t *fclean, const uint32_t *cidx, const uint64_t *coff, const uint32_t *nidx,
                       const uint64_t *noff, uint32_t cidx_base, uint64_t coff_base, uint32_t file_read_base,
                       uint8_t *out_clean, uint64_t *out_off, uint8_t *out_N, uint32_t *out_orderN, uint64_t *out_offN) {
  if (nreads)
    hipLaunchKernelGGL(k_pack_reads, GRIDN(nreads, 16), dim3(256), 0, st, t, line_end, nreads, len, fclean, cidx, coff, nidx,
                       noff, cidx_base, coff_base, file_read_base, out_clean, out_off, out_N, out_orderN, out_offN);
}
hipError_t reduce_max_u32(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint32_t *out, size_t n) {
  return rocprim::reduce(tmp, tmp_bytes, in, out, 0u, n, rocprim::maximum<uint32_t>(), st);
}
hipError_t reduce_min_u32(hipStream_t st, void *tmp, size_t &tmp_bytes, const uint32_t *in, uint32_t *out, size_t n) {
  return rocprim::reduce(tmp, tmp_bytes, in, out, 0xffffffffu, n, rocprim::minimum<uint32_t>(), st);
}

}  // namespace sr
