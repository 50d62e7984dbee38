// oracle/ref_order_driver.cpp -- TEST INFRASTRUCTURE.
//
// Command-line driver around the REAL reference functions that consume read_order.bin, compiled where they
// lie (/root/reference/src/pe_encode.cpp and reorder_compress_quality_id.cpp; neither needs Boost) by
// oracle/Makefile into oracle/_ref/ref_order.  Nothing from the reference is copied here; this file only
// calls it.  The other functions of reorder_compress_quality_id.cpp (which need the id / quality codecs) are
// never referenced from main() and are dropped by the linker (-ffunction-sections + --gc-sections), so no
// stand-in for anything is written.  tests/test_order_ops.py uses it to pin the oracle twins
// (orc_generate_order_se/pe, pe_encode) and the GPU kernels of spring_amd/csrc/order_ops.hip.
//
//   ref_order se <dir> <numreads>         spring::generate_order_se(dir/read_order.bin) -> dir/order_array.bin
//   ref_order pe <dir> <numreads>         spring::generate_order_pe(...)                -> dir/order_array.bin (n/2)
//   ref_order pe_encode <dir> <numreads>  spring::pe_encode(dir, cp): rewrites dir/read_order.bin in place
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pe_encode.h"
#include "reorder_compress_quality_id.h"

int main(int argc, char **argv) {
  if (argc != 4) { fprintf(stderr, "usage: ref_order se|pe|pe_encode <dir> <numreads>\n"); return 2; }
  const std::string mode = argv[1], dir = argv[2];
  const uint32_t n = (uint32_t)strtoull(argv[3], nullptr, 10);
  if (mode == "pe_encode") {
    spring::compression_params cp;
    memset(&cp, 0, sizeof(cp));
    cp.num_reads = n;
    cp.paired_end = true;
    spring::pe_encode(dir, cp);
    return 0;
  }
  const bool pe = mode == "pe";
  if (!pe && mode != "se") return 2;
  std::vector<uint32_t> out(pe ? n / 2 : n, 0xffffffffu);
  if (pe) spring::generate_order_pe(dir + "/read_order.bin", out.data(), n);
  else spring::generate_order_se(dir + "/read_order.bin", out.data(), n);
  std::ofstream f(dir + "/order_array.bin", std::ios::binary);
  f.write((const char *)out.data(), (std::streamsize)out.size() * 4);
  return f.good() ? 0 : 1;
}
