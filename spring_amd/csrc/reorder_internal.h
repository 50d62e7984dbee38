// spring_amd/csrc/reorder_internal.h -- what the encoder stage (encoder.hip) needs from a finished
// reorder context without a round trip through the host: device pointers of the read pool and of
// the final streams, plus the library's pooled device allocator.  Internal to the library.
#ifndef SPRING_REORDER_INTERNAL_H_
#define SPRING_REORDER_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spring_reorder.h"

namespace sr {

struct ReorderView {
  int dev;
  hipStream_t st;
  uint32_t n;          // clean reads in the pool
  int L, W, S;         // max read length, limbs per read, limb stride
  const uint64_t *reads;
  const uint16_t *lens;
  uint64_t nrec, nsing;   // matched records, singletons
  const uint32_t *f_order, *f_order_s;
  const char *f_rc, *f_flag;
  const long long *f_pos;
  const uint16_t *f_len;
  const uint64_t *tid_off;   // host, num_thr + 1
  int num_thr;
  // reads with N of the two input files when the context was loaded through the FASTQ front end (device)
  const uint8_t *N_dna[2];
  const uint64_t *N_off[2];
  const uint32_t *N_order[2];
  uint32_t N_count[2];
  uint64_t N_bytes[2];
  uint32_t fq_num_reads_0;   // reads of file 1 (file-2 positions in read_order_N.bin are offset by it)
};
int reorder_view(spring_reorder_ctx *ctx, ReorderView *v);   // fails unless the context is finalized

hipError_t dev_alloc(int dev, size_t bytes, void **out);     // pooled (reorder_pipeline.cpp)
void dev_free(int dev, void *p);
int fail(int code, const char *fmt, ...);

}  // namespace sr
#endif
