// oracle/ref_bitset_driver.cpp -- TEST INFRASTRUCTURE.
//
// Thin C-ABI driver around the REAL reference sources, compiled where they lie
// (/root/reference/src/bitset_util.{h,cpp} + BooPHF.h + params.h) by
// oracle/Makefile into oracle/_ref/libref_bitset.so.  Nothing from the
// reference is copied here; this file only calls it.  It exists so
// tests/test_oracle_vs_ref.py can pin the oracle's dictionary construction,
// bin findpos/remove tail encoding, index masks and Hamming masks against the
// reference's own code.  (reorder.h itself needs Boost.Iostreams, which this
// image lacks, so reorder_main<> is NOT buildable here -- see oracle/README.md.)
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include "bitset_util.h"

namespace {

// runs with cwd = dir (BooPHF drops temp files in cwd, BooPHF.h:1211) and goes back afterwards: the caller's
// temp directory is deleted later and a process must not be left sitting in it
struct CwdGuard {
  char old[4096];
  bool ok;
  explicit CwdGuard(const char *dir) { ok = getcwd(old, sizeof(old)) != nullptr && chdir(dir) == 0; }
  ~CwdGuard() { if (ok && chdir(old) != 0) ok = false; }
};

template <size_t BS>
int build_dict(const uint64_t *limbs, const uint16_t *len, uint32_t n, int which_start[2],
               int which_end[2], const char *basedir, int num_thr, int which,
               const uint64_t *probe_keys, uint32_t nprobe, uint32_t *bin_size,
               uint32_t *bin_ids /* concatenated */, uint32_t *numkeys, uint32_t *dict_numreads, int bpb = 2) {
  std::bitset<BS> *read = new std::bitset<BS>[n ? n : 1];
  for (uint32_t i = 0; i < n; i++) std::memcpy((void *)&read[i], limbs + (size_t)i * (BS / 64), BS / 8);
  spring::bbhashdict *dict = new spring::bbhashdict[2];
  for (int l = 0; l < 2; l++) { dict[l].start = which_start[l]; dict[l].end = which_end[l]; }
  omp_set_num_threads(num_thr);
  std::vector<uint16_t> lens(len, len + n);
  spring::constructdictionary<BS>(read, dict, lens.data(), 2, n, bpb, std::string(basedir), num_thr);
  spring::bbhashdict &d = dict[which];
  *numkeys = d.numkeys;
  *dict_numreads = d.dict_numreads;
  size_t o = 0;
  for (uint32_t i = 0; i < nprobe; i++) {
    uint64_t idx = d.bphf->lookup(probe_keys[i]);
    if (idx >= d.numkeys) { bin_size[i] = 0xffffffffu; continue; }
    int64_t di[2];
    d.findpos(di, idx);
    bin_size[i] = (uint32_t)(di[1] - di[0]);
    for (int64_t j = di[0]; j < di[1]; j++) bin_ids[o++] = d.read_id[j];
  }
  delete[] read;
  delete[] dict;
  return 0;
}

template <size_t BS>
int mask_hamming(const uint64_t *a, const uint64_t *b, int L, int i, int j, int bpb = 2) {
  static std::bitset<BS> **mask = nullptr;
  static int maskL = -1, maskB = -1;
  if (maskL != L || maskB != bpb) {
    if (mask) { for (int k = 0; k < maskL; k++) delete[] mask[k]; delete[] mask; }
    mask = new std::bitset<BS> *[L];
    for (int k = 0; k < L; k++) mask[k] = new std::bitset<BS>[L];
    spring::generatemasks<BS>(mask, L, bpb);
    maskL = L;
    maskB = bpb;
  }
  std::bitset<BS> x, y;
  std::memcpy((void *)&x, a, BS / 8);
  std::memcpy((void *)&y, b, BS / 8);
  return (int)((x ^ y) & mask[i][j]).count();
}

}  // namespace

#define DISPATCH(W, CALL)                                                                  \
  switch (W) {                                                                             \
    case 1: return CALL(64); case 2: return CALL(128); case 3: return CALL(192);           \
    case 4: return CALL(256); case 5: return CALL(320); case 6: return CALL(384);          \
    case 7: return CALL(448); case 8: return CALL(512); case 9: return CALL(576);          \
    case 10: return CALL(640); case 11: return CALL(704); case 12: return CALL(768);       \
    case 13: return CALL(832); case 14: return CALL(896); case 15: return CALL(960);       \
    case 16: return CALL(1024); default: return -1;                                        \
  }

extern "C" {

// real constructdictionary<BS>() (bitset_util.h:74-221); for every probe key
// returns the live bin (bbhashdict::findpos) the reference built for it.
// Runs with cwd = basedir because BooPHF drops temp files in cwd (BooPHF.h:1211).
int ref_build_dict(const uint64_t *limbs, const uint16_t *len, uint32_t n, int W, int start0,
                   int end0, int start1, int end1, const char *basedir, int num_thr, int which,
                   const uint64_t *probe_keys, uint32_t nprobe, uint32_t *bin_size,
                   uint32_t *bin_ids, uint32_t *numkeys, uint32_t *dict_numreads) {
  int s[2] = {start0, start1}, e[2] = {end0, end1};
  CwdGuard cwd(basedir);
  if (!cwd.ok) return -2;
#define CALL(BS) build_dict<BS>(limbs, len, n, s, e, basedir, num_thr, which, probe_keys, nprobe, bin_size, bin_ids, numkeys, dict_numreads)
  DISPATCH(W, CALL)
#undef CALL
}

// real bbhashdict::findpos + remove (bitset_util.cpp:20-63) on one bin of
// capacity cap stored in read_id[0..cap).  Returns live count after removal.
int64_t ref_bin_remove(uint32_t *read_id, uint32_t cap, uint8_t *empty_bin, int64_t current) {
  spring::bbhashdict d;
  uint32_t sp[2] = {0, cap};
  d.startpos = sp;
  d.read_id = read_id;
  d.empty_bin = (bool *)empty_bin;
  int64_t di[2];
  d.findpos(di, 0);
  d.remove(di, 0, current);
  d.findpos(di, 0);
  d.startpos = NULL; d.read_id = NULL; d.empty_bin = NULL;  // not ours to free
  return di[1] - di[0];
}

int64_t ref_bin_live(uint32_t *read_id, uint32_t cap) {
  spring::bbhashdict d;
  uint32_t sp[2] = {0, cap};
  d.startpos = sp;
  d.read_id = read_id;
  int64_t di[2];
  d.findpos(di, 0);
  d.startpos = NULL; d.read_id = NULL;
  return di[1] - di[0];
}

// ((a ^ b) & mask[i][j]).count() with the real generatemasks (bitset_util.h:223-236)
int ref_mask_hamming(const uint64_t *a, const uint64_t *b, int W, int L, int i, int j) {
#define CALL(BS) mask_hamming<BS>(a, b, L, i, j)
  DISPATCH(W, CALL)
#undef CALL
}

// the same two entry points with the encoder's 3 bits per base (encoder.h:617-619, :131-140)
int ref_build_dict_bpb(const uint64_t *limbs, const uint16_t *len, uint32_t n, int W, int start0, int end0, int start1,
                       int end1, const char *basedir, int num_thr, int which, const uint64_t *probe_keys,
                       uint32_t nprobe, uint32_t *bin_size, uint32_t *bin_ids, uint32_t *numkeys,
                       uint32_t *dict_numreads, int bpb) {
  int s[2] = {start0, start1}, e[2] = {end0, end1};
  CwdGuard cwd(basedir);
  if (!cwd.ok) return -2;
#define CALL(BS) build_dict<BS>(limbs, len, n, s, e, basedir, num_thr, which, probe_keys, nprobe, bin_size, bin_ids, numkeys, dict_numreads, bpb)
  DISPATCH(W, CALL)
#undef CALL
}
int ref_mask_hamming_bpb(const uint64_t *a, const uint64_t *b, int W, int L, int i, int j, int bpb) {
#define CALL(BS) mask_hamming<BS>(a, b, L, i, j, bpb)
  DISPATCH(W, CALL)
#undef CALL
}

}  // extern "C"
