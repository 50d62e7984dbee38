// spring_amd/csrc/call_reorder.h -- C++ mirror of the reference's operator
// interface for this stage:
//     void spring::call_reorder(const std::string &temp_dir, compression_params &cp);
//     (reference src/call_template_functions.h:9, .cpp:9-63)
// Same name, same argument meaning (only the compression_params fields the
// stage reads, reorder.h:747-763), same error behaviour (std::runtime_error,
// "Wrong bitset size." for unsupported read lengths).
#ifndef SPRING_AMD_CALL_REORDER_H_
#define SPRING_AMD_CALL_REORDER_H_

#include <cstdint>
#include <string>

#include "spring_reorder.h"

namespace spring_amd {

struct reorder_params {          // subset of spring::compression_params (util.h:30-51)
  bool paired_end = false;       // cp.paired_end
  uint32_t num_reads_clean[2] = {0, 0};  // cp.num_reads_clean
  uint32_t max_readlen = 0;      // cp.max_readlen
  int num_thr = 1;               // cp.num_thr (number of per-tid output sets)
};

void call_reorder(const std::string &temp_dir, const reorder_params &cp,
                  const spring_reorder_opts *opts = nullptr);

// Mirror of void spring::call_encoder(const std::string &temp_dir, compression_params &cp)
// (reference src/call_template_functions.h:11, .cpp:65-142): consumes the files call_reorder left in temp_dir;
// num_reads = cp.num_reads (clean + N reads).  Throws std::runtime_error ("Wrong bitset size." when
// 3 * max_readlen does not fit 1536 bits).  NOT a complete drop-in: it stops before pack_compress_seq's BSC step
// (encoder.cpp:146-150) -- read_seq.bin.<tid> is left as .tmp + .tail, the caller runs BSC_compress(.tmp -> .bsc)
// and removes the .tmp (INTEGRATION.md section 4 shows the loop).
void call_encoder(const std::string &temp_dir, const reorder_params &cp, uint32_t num_reads, int device = -1);

// Both stages back to back with the intermediate streams kept in HBM (spring.cpp:150-160).
void call_reorder_encoder(const std::string &temp_dir, const reorder_params &cp, uint32_t num_reads,
                          const spring_reorder_opts *opts = nullptr);

}  // namespace spring_amd
#endif
