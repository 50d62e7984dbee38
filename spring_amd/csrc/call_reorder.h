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

}  // namespace spring_amd
#endif
