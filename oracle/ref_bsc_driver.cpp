// oracle/ref_bsc_driver.cpp -- TEST INFRASTRUCTURE.
//
// Command-line driver around the REAL reference libbsc (/root/reference/src/libbsc, no Boost), compiled where
// it lies by oracle/Makefile into oracle/_ref/ref_bsc: `ref_bsc <infile> <outfile>` calls
// spring::bsc::BSC_compress exactly as pack_compress_seq / the stream compressors do (encoder.cpp:146-150,
// reorder_compress_streams.cpp).  Used by tools/compression_bsc.py to report real BSC sizes of the encoder
// output for different chain counts (it replaces the xz stand-in of round 1).  Never part of the product.
#include <cstdio>
#include "libbsc/bsc.h"

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: ref_bsc <infile> <outfile>\n"); return 2; }
  spring::bsc::BSC_compress(argv[1], argv[2]);
  return 0;
}
