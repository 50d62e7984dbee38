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
  const uint64_t *tid_off_s; // host, num_thr + 1: the singletons of tid t are f_order_s[tid_off_s[t] .. tid_off_s[t + 1])
  const uint64_t *tid_mid, *tid_mid_s;  // host, num_thr: inside tid t, where the records of the second chain group's chains begin (a pool
                                        // that runs two groups: the merge over ranks takes every rank's first part, then every second part)
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
// same for a rank of a multi-GPU pool as well (it holds the streams of its own chains only)
int reorder_view_any(spring_reorder_ctx *ctx, ReorderView *v);

// ---- streaming host <-> device transfers (reorder_pipeline.cpp), used by the drop-in stage (reorder_files.cpp)
// Pinned staging chunks, cached for the life of the process (pinning memory costs ~0.3 ms per MB: a 4 GB input
// would pay more for the pinning than for the copy); spring_reorder_trim_pool() releases them.
constexpr size_t PIN_CHUNK = (size_t)32 << 20;
void *pinned_get();           // nullptr when hipHostMalloc fails
void pinned_put(void *p);
// A record stream that can be read piecewise from several threads at once: a memory image or files (pread).
struct DnaSource {
  size_t nbytes = 0;
  // copies bytes [off, off + len) of the stream to dst; 0 on success (thread-safe)
  int (*fill)(void *self, size_t off, void *dst, size_t len) = nullptr;
  void *self = nullptr;
  const uint8_t *image = nullptr;  // set when the stream already IS a host memory image: a fallback that has to walk the records walks it in place
};
// readDnaFile (reorder.h:222-244) from such a source: host threads fill pinned chunks while earlier chunks are on
// their way to the device (double buffering per thread), then the unpack kernel.  A stream of exactly
// n * (2 + ceil(L/4)) bytes is taken as fixed-length and verified on the device; anything else goes through the
// host walk of the records (spring_reorder_load_dna).
int load_dna_source(spring_reorder_ctx *ctx, const DnaSource &src, uint32_t n, uint32_t max_readlen);
// temp.dna.<tid> / temp.dna.singleton record stream built on the device (spring_reorder_emit_dna without the copy):
// *d_out is a pooled device buffer of *nbytes bytes that the caller hands back with emit_dna_free (null when empty).
// tid = -1 with s_cnt != ~0: only singletons [s_first, s_first + s_cnt) (one tid's share, tid_off_s).
int emit_dna_device(spring_reorder_ctx *ctx, int32_t tid, uint8_t **d_out, size_t *nbytes, uint64_t s_first = 0,
                    uint64_t s_cnt = ~0ull, size_t *mid_bytes = nullptr /* tid >= 0: byte offset of record tid_mid[tid] */);
void emit_dna_free(spring_reorder_ctx *ctx, uint8_t *d);

void mg_comm_abort(spring_mg_comm *c);  // a failed rank of an in-process pool unblocks its peers (ncclCommAbort)
hipError_t dev_alloc(int dev, size_t bytes, void **out);     // pooled (reorder_pipeline.cpp)
void dev_free(int dev, void *p);
int fail(int code, const char *fmt, ...);

}  // namespace sr
#endif
