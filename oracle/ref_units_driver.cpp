// oracle/ref_units_driver.cpp -- TEST INFRASTRUCTURE.
//
// C-ABI driver around the parts of the REAL reference main loop that compile in this image without Boost.
// oracle/Makefile generates three translation units under oracle/_ref/gen/ (never committed) by LINE RANGE from
// the sources where they lie under /root/reference/src -- nothing is edited, nothing is stood in for:
//   reorder_units.gen.h   = reorder.h:33-318 (reorder_global, bitsettostring, setglobalarrays, updaterefcount,
//                           readDnaFile, search_match) behind reorder.h's own non-Boost includes (:18-20, :24-36)
//   encoder_units.gen.h   = encoder.h:34-122 (encoder_global_b, encoder_global, contig_reads, declarations,
//                           bitsettostring) + :496-571 (setglobalarrays, readsingletons)
//   util_units.gen.cpp    = util.cpp:31-54 (read_fastq_block), :269-394 (write/read_dna[N]_in/from_bits,
//                           reverse_complement x2, remove_CR_from_end)
//   encoder_units.gen.cpp = encoder.cpp:32-109 (buildcontig, writecontig), :177-222 (correct_order)
// What stays unbuildable: the reorder() driver loop (reorder.h:320-641: Boost gzip streams at :355-368),
// writetofile (:643-730, Boost at :656-658), encode<>() (encoder.h:124-494, Boost at :153-176), preprocess().
//
// Two kinds of entry points:
//   ref_u_*      one call of one reference function on caller-supplied state;
//   ref_shadow_* a "shadow" of one reorder() thread's state built and advanced ONLY by reference code (real
//                constructdictionary bins, real remainingreads[], real count[][]/ref/revref); the oracle's serial
//                restatement calls the hooks at every step of a full run (oracle/reorder_oracle.c::
//                orc_reorder_serial_shadow) and every one of its search_match / updaterefcount / bin removal /
//                seed pick results is compared with what the reference's function returns on the mirrored state.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <list>
#include <sstream>
#include <string>
#include <vector>
#include <unistd.h>

#include "reorder_units.gen.h"
#include "encoder_units.gen.h"

namespace {

struct CwdGuard {
  char old[4096];
  bool ok;
  explicit CwdGuard(const char *dir) { ok = getcwd(old, sizeof(old)) != nullptr && chdir(dir) == 0; }
  ~CwdGuard() { if (ok && chdir(old) != 0) ok = false; }
};

template <size_t BS>
void to_bits(const uint64_t *limbs, std::bitset<BS> &b) { std::memcpy((void *)&b, limbs, BS / 8); }
template <size_t BS>
void from_bits(const std::bitset<BS> &b, uint64_t *limbs) { std::memcpy(limbs, (const void *)&b, BS / 8); }

template <size_t BS>
spring::reorder_global<BS> *make_rg(int L, uint32_t n) {
  auto *rg = new spring::reorder_global<BS>(L);
  rg->max_readlen = L;
  rg->numreads = n;
  rg->numreads_array[0] = n;
  rg->numreads_array[1] = 0;
  rg->maxshift = L / 2;  // reorder.h:750
  rg->num_thr = 1;
  rg->paired_end = false;
  spring::setglobalarrays<BS>(*rg);
  return rg;
}

// ---- one updaterefcount<BS>() call on caller-held state (cnt = int32 [4][stride], rows A C T G)
template <size_t BS>
int u_updaterefcount(int L, const uint64_t *cur, int32_t *cnt, int stride, uint64_t *ref, uint64_t *revref,
                     int *ref_len, int reset, int rev, int shift, int cur_readlen) {
  static spring::reorder_global<BS> *rg = nullptr;
  static int rgL = -1;
  if (rgL != L) { delete rg; rg = make_rg<BS>(L, 0); rgL = L; }
  std::bitset<BS> c, r, rr;
  to_bits<BS>(cur, c); to_bits<BS>(ref, r); to_bits<BS>(revref, rr);
  int *count[4];
  std::vector<int> rows[4];
  for (int j = 0; j < 4; j++) { rows[j].assign(cnt + (size_t)j * stride, cnt + (size_t)j * stride + L); count[j] = rows[j].data(); }
  spring::updaterefcount<BS>(c, r, rr, count, reset != 0, rev != 0, shift, (uint16_t)cur_readlen, *ref_len, *rg);
  for (int j = 0; j < 4; j++) std::memcpy(cnt + (size_t)j * stride, count[j], sizeof(int) * L);
  from_bits<BS>(r, ref); from_bits<BS>(rr, revref);
  return 0;
}

template <size_t BS>
int u_chartobitset(const char *s, int len, int L, uint64_t *out) {
  spring::reorder_global<BS> *rg = make_rg<BS>(L, 0);
  std::bitset<BS> b;
  std::vector<char> tmp(s, s + len);
  tmp.push_back(0);
  spring::chartobitset<BS>(tmp.data(), len, b, rg->basemask);
  from_bits<BS>(b, out);
  delete rg;
  return 0;
}
template <size_t BS>
int u_bitsettostring(const uint64_t *limbs, int len, int L, char *out) {
  spring::reorder_global<BS> *rg = make_rg<BS>(L, 0);
  std::bitset<BS> b;
  to_bits<BS>(limbs, b);
  spring::bitsettostring<BS>(b, out, (uint16_t)len, *rg);
  delete rg;
  return 0;
}

template <size_t BS>
int u_readDnaFile(const char *f1, const char *f2, uint32_t n0, uint32_t n1, int L, uint64_t *limbs, uint16_t *lens) {
  spring::reorder_global<BS> *rg = make_rg<BS>(L, n0 + n1);
  rg->numreads_array[0] = n0;
  rg->numreads_array[1] = n1;
  rg->paired_end = n1 > 0 || (f2 && f2[0]);
  rg->infile[0] = f1;
  rg->infile[1] = f2 ? f2 : "";
  uint32_t n = n0 + n1;
  std::bitset<BS> *read = new std::bitset<BS>[n ? n : 1];  // zero-initialised like reorder.h:767
  spring::readDnaFile<BS>(read, lens, *rg);
  for (uint32_t i = 0; i < n; i++) from_bits<BS>(read[i], limbs + (size_t)i * (BS / 64));
  delete[] read;
  delete rg;
  return 0;
}

// ---- the shadow of one reorder() thread
struct ShadowBase {
  virtual ~ShadowBase() {}
  virtual int claim_first(uint32_t current) = 0;
  virtual int remove(uint32_t current) = 0;
  virtual int search(const uint64_t *given, int rev, int shift, int ref_len, int flag, uint32_t k) = 0;
  virtual int update(uint32_t rid, int reset, int rev, int shift, const int32_t *cnt, int stride, const uint64_t *ref,
                     const uint64_t *revref, int ref_len) = 0;
  virtual int64_t pick_seed() = 0;
  virtual void set_remaining(const uint8_t *r) = 0;
  virtual void get_remaining(uint8_t *r) = 0;
  virtual int search_raw(const uint64_t *refbits, int rev, int shift, int ref_len, uint32_t *k) = 0;
  virtual int search_loop(const uint64_t *ref, const uint64_t *revref, int ref_len, uint32_t *k, int *shift, int *rev) = 0;
};

template <size_t BS>
struct Shadow : ShadowBase {
  uint32_t n;
  int L;
  spring::reorder_global<BS> *rg;
  std::bitset<BS> *read;
  std::vector<uint16_t> lens;
  spring::bbhashdict *dict;
  std::bitset<BS> **mask;
  std::bitset<BS> *mask1;
  omp_lock_t *dict_lock, *read_lock;
  bool *remainingreads;
  // thread state of reorder.h:370-398
  std::bitset<BS> ref, revref;
  int **count;
  int ref_len;
  int64_t remainingpos;

  Shadow(const uint64_t *limbs, const uint16_t *len, uint32_t n_, int L_, const char *basedir, int num_thr)
      : n(n_), L(L_), lens(len, len + n_) {
    rg = make_rg<BS>(L, n);
    read = new std::bitset<BS>[n ? n : 1];
    for (uint32_t i = 0; i < n; i++) to_bits<BS>(limbs + (size_t)i * (BS / 64), read[i]);
    dict = new spring::bbhashdict[2];
    // reorder.h:751-759
    if (L > 100) { dict[0].start = L / 2 - 32; dict[0].end = L / 2 - 1; dict[1].start = L / 2; dict[1].end = L / 2 - 1 + 32; }
    else { dict[0].start = L / 2 - 32 * L / 100; dict[0].end = L / 2 - 1; dict[1].start = L / 2; dict[1].end = L / 2 - 1 + 32 * L / 100; }
    omp_set_num_threads(num_thr);
    if (n > 0) spring::constructdictionary<BS>(read, dict, lens.data(), 2, n, 2, std::string(basedir), num_thr);
    omp_set_num_threads(1);
    // reorder.h:323-344
    dict_lock = new omp_lock_t[spring::NUM_LOCKS_REORDER];
    read_lock = new omp_lock_t[spring::NUM_LOCKS_REORDER];
    for (int j = 0; j < spring::NUM_LOCKS_REORDER; j++) { omp_init_lock(&dict_lock[j]); omp_init_lock(&read_lock[j]); }
    mask = new std::bitset<BS> *[L];
    for (int i = 0; i < L; i++) mask[i] = new std::bitset<BS>[L];
    spring::generatemasks<BS>(mask, L, 2);
    mask1 = new std::bitset<BS>[2];
    spring::generateindexmasks<BS>(mask1, dict, 2, 2);
    remainingreads = new bool[n ? n : 1];
    std::fill(remainingreads, remainingreads + n, 1);
    count = new int *[4];
    for (int j = 0; j < 4; j++) count[j] = new int[L]();
    ref_len = 0;
    remainingpos = (int64_t)n - 1;
  }
  ~Shadow() override {
    for (int j = 0; j < 4; j++) delete[] count[j];
    delete[] count;
    delete[] remainingreads;
    delete[] mask1;
    for (int i = 0; i < L; i++) delete[] mask[i];
    delete[] mask;
    delete[] dict_lock;
    delete[] read_lock;
    delete[] dict;
    delete[] read;
    delete rg;
  }
  // reorder.h:405-417
  int claim_first(uint32_t current) override {
    if (current >= n || !remainingreads[current]) return 1;
    remainingreads[current] = 0;
    return 0;
  }
  // reorder.h:458-472 with the reference's own lookup / findpos / remove (one thread: try-locks succeed)
  int remove(uint32_t current) override {
    int64_t dictidx[2];
    for (int l = 0; l < 2; l++) {
      if (lens[current] <= dict[l].end) continue;
      std::bitset<BS> b = read[current] & mask1[l];
      uint64_t ull = (b >> 2 * dict[l].start).to_ullong();
      uint64_t startposidx = dict[l].bphf->lookup(ull);
      if (startposidx >= dict[l].numkeys) return 1;
      dict[l].findpos(dictidx, startposidx);
      dict[l].remove(dictidx, startposidx, current);
    }
    return 0;
  }
  int search(const uint64_t *given, int rev, int shift, int given_ref_len, int flag, uint32_t k) override {
    int bad = 0;
    // the oracle's working copy must be what `revref <<= 2; ref >>= 2` (reorder.h:556-557) has made of the thread's own
    std::bitset<BS> r = rev ? (revref << (2 * shift)) : (ref >> (2 * shift));
    uint64_t own[BS / 64];
    from_bits<BS>(r, own);
    if (std::memcmp(own, given, BS / 8) != 0 || given_ref_len != ref_len) bad |= 2;
    uint32_t kk = 0;
    bool f = spring::search_match<BS>(r, mask1, dict_lock, read_lock, mask, lens.data(), remainingreads, read, dict, kk,
                                      rev != 0, shift, ref_len, *rg);
    if ((int)f != flag || (f && kk != k)) bad |= 1;
    return bad;
  }
  int update(uint32_t rid, int reset, int rev, int shift, const int32_t *cnt, int stride, const uint64_t *oref,
             const uint64_t *orevref, int oref_len) override {
    spring::updaterefcount<BS>(read[rid], ref, revref, count, reset != 0, rev != 0, shift, lens[rid], ref_len, *rg);
    int bad = 0;
    uint64_t a[BS / 64], b[BS / 64];
    from_bits<BS>(ref, a); from_bits<BS>(revref, b);
    if (std::memcmp(a, oref, BS / 8) != 0 || std::memcmp(b, orevref, BS / 8) != 0 || ref_len != oref_len) bad |= 1;
    for (int j = 0; j < 4; j++)
      if (std::memcmp(count[j], cnt + (size_t)j * stride, sizeof(int) * L) != 0) bad |= 2;
    return bad;
  }
  // reorder.h:576-592
  int64_t pick_seed() override {
    for (int64_t j = remainingpos; j >= 0; j--)
      if (remainingreads[j] == 1) {
        remainingpos = j - 1;
        remainingreads[j] = 0;
        return j;
      }
    return -1;
  }
  void set_remaining(const uint8_t *r) override { for (uint32_t i = 0; i < n; i++) remainingreads[i] = r[i] != 0; }
  void get_remaining(uint8_t *r) override { for (uint32_t i = 0; i < n; i++) r[i] = remainingreads[i]; }
  int search_raw(const uint64_t *refbits, int rev, int shift, int rl, uint32_t *k) override {
    std::bitset<BS> r;
    to_bits<BS>(refbits, r);
    uint32_t kk = 0;
    bool f = spring::search_match<BS>(r, mask1, dict_lock, read_lock, mask, lens.data(), remainingreads, read, dict, kk,
                                      rev != 0, shift, rl, *rg);
    *k = kk;
    return f;
  }
  // the shift loop reorder.h:479-558 around the real search_match, without the update; the claimed read is handed back
  int search_loop(const uint64_t *r0, const uint64_t *rr0, int rl, uint32_t *k, int *oshift, int *orev) override {
    std::bitset<BS> r, rr;
    to_bits<BS>(r0, r); to_bits<BS>(rr0, rr);
    for (int shift = 0; shift < rg->maxshift; shift++) {
      for (int rev = 0; rev < 2; rev++) {
        uint32_t kk = 0;
        bool f = spring::search_match<BS>(rev ? rr : r, mask1, dict_lock, read_lock, mask, lens.data(), remainingreads, read,
                                          dict, kk, rev != 0, shift, rl, *rg);
        if (f) { remainingreads[kk] = 1; *k = kk; *oshift = shift; *orev = rev; return 1; }
      }
      rr <<= 2;
      r >>= 2;
    }
    return 0;
  }
};

std::string slurp(const std::string &p) {
  std::ifstream f(p, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
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

static int limbs_of(int L) { return (2 * L - 1) / 64 + 1; }  // call_template_functions.cpp:10

extern "C" {

// real updaterefcount<BS> (reorder.h:110-220)
int ref_u_updaterefcount(int L, const uint64_t *cur, int32_t *cnt, int stride, uint64_t *ref, uint64_t *revref,
                         int *ref_len, int reset, int rev, int shift, int cur_readlen) {
#define CALL(BS) u_updaterefcount<BS>(L, cur, cnt, stride, ref, revref, ref_len, reset, rev, shift, cur_readlen)
  DISPATCH(limbs_of(L), CALL)
#undef CALL
}
// real setglobalarrays + chartobitset (reorder.h:94-108, bitset_util.h:238-244) / bitsettostring (reorder.h:76-92)
int ref_u_chartobitset(const char *s, int len, int L, uint64_t *out) {
#define CALL(BS) u_chartobitset<BS>(s, len, L, out)
  DISPATCH(limbs_of(L), CALL)
#undef CALL
}
int ref_u_bitsettostring(const uint64_t *limbs, int len, int L, char *out) {
#define CALL(BS) u_bitsettostring<BS>(limbs, len, L, out)
  DISPATCH(limbs_of(L), CALL)
#undef CALL
}
// real readDnaFile (reorder.h:222-244); deletes its input files like the reference
int ref_u_readDnaFile(const char *f1, const char *f2, uint32_t n0, uint32_t n1, int L, uint64_t *limbs, uint16_t *lens) {
#define CALL(BS) u_readDnaFile<BS>(f1, f2, n0, n1, L, limbs, lens)
  DISPATCH(limbs_of(L), CALL)
#undef CALL
}
// real reverse_complement (util.cpp:376-381)
void ref_u_reverse_complement(const char *s, char *s1, int len) {
  std::vector<char> t(s, s + len);
  t.push_back(0);
  spring::reverse_complement(t.data(), s1, len);
}
// real write_dna_in_bits / write_dnaN_in_bits (util.cpp:269-294, :322-348) for `count` NUL-separated strings -> file
int ref_u_write_dna(const char *strings, uint32_t count, const char *path, int withN) {
  std::ofstream f(path, std::ios::binary);
  const char *p = strings;
  for (uint32_t i = 0; i < count; i++) {
    std::string s(p);
    p += s.size() + 1;
    if (withN) spring::write_dnaN_in_bits(s, f); else spring::write_dna_in_bits(s, f);
  }
  f.close();
  return 0;
}
// real read_dna_from_bits / read_dnaN_from_bits (util.cpp:296-320, :350-374): file -> NUL-separated strings
long ref_u_read_dna(const char *path, uint32_t count, int withN, char *out, long cap) {
  std::ifstream f(path, std::ios::binary);
  long o = 0;
  std::string s;
  for (uint32_t i = 0; i < count; i++) {
    if (withN) spring::read_dnaN_from_bits(s, f); else spring::read_dna_from_bits(s, f);
    if (o + (long)s.size() + 1 > cap) return -1;
    std::memcpy(out + o, s.data(), s.size());
    o += s.size();
    out[o++] = 0;
  }
  return o;
}
// real read_fastq_block (util.cpp:31-54) over a text buffer: reads (NUL-separated) of up to max_reads records;
// returns the number of records, -1 when the reference throws ("Number of lines not multiple of 4")
long ref_u_read_fastq(const char *text, size_t nbytes, uint32_t max_reads, char *reads_out, long cap, long *used) {
  std::istringstream in(std::string(text, nbytes));
  std::vector<std::string> id(max_reads), rd(max_reads), q(max_reads);
  uint32_t got;
  try {
    got = spring::read_fastq_block(&in, id.data(), rd.data(), q.data(), max_reads, false);
  } catch (std::runtime_error &) { return -1; }
  long o = 0;
  for (uint32_t i = 0; i < got; i++) {
    if (o + (long)rd[i].size() + 1 > cap) return -2;
    std::memcpy(reads_out + o, rd[i].data(), rd[i].size());
    o += rd[i].size();
    reads_out[o++] = 0;
  }
  *used = o;
  return got;
}

// ---- shadow
void *ref_shadow_create(const uint64_t *limbs, const uint16_t *len, uint32_t n, int L, const char *basedir, int num_thr) {
  CwdGuard cwd(basedir);  // BooPHF drops temp files in cwd
  if (!cwd.ok) return nullptr;
#define CALL(BS) (ShadowBase *)new Shadow<BS>(limbs, len, n, L, basedir, num_thr)
  switch (limbs_of(L)) {
    case 1: return CALL(64); case 2: return CALL(128); case 3: return CALL(192); case 4: return CALL(256);
    case 5: return CALL(320); case 6: return CALL(384); case 7: return CALL(448); case 8: return CALL(512);
    case 9: return CALL(576); case 10: return CALL(640); case 11: return CALL(704); case 12: return CALL(768);
    case 13: return CALL(832); case 14: return CALL(896); case 15: return CALL(960); case 16: return CALL(1024);
    default: return nullptr;
  }
#undef CALL
}
void ref_shadow_destroy(void *s) { delete (ShadowBase *)s; }
int ref_shadow_claim_first(void *s, uint32_t current) { return ((ShadowBase *)s)->claim_first(current); }
int ref_shadow_remove(void *s, uint32_t current) { return ((ShadowBase *)s)->remove(current); }
int ref_shadow_search(void *s, const uint64_t *given, int rev, int shift, int ref_len, int flag, uint32_t k) {
  return ((ShadowBase *)s)->search(given, rev, shift, ref_len, flag, k);
}
int ref_shadow_update(void *s, uint32_t rid, int reset, int rev, int shift, const int32_t *cnt, int stride,
                      const uint64_t *ref, const uint64_t *revref, int ref_len) {
  return ((ShadowBase *)s)->update(rid, reset, rev, shift, cnt, stride, ref, revref, ref_len);
}
int64_t ref_shadow_pick_seed(void *s) { return ((ShadowBase *)s)->pick_seed(); }
void ref_shadow_set_remaining(void *s, const uint8_t *r) { ((ShadowBase *)s)->set_remaining(r); }
void ref_shadow_get_remaining(void *s, uint8_t *r) { ((ShadowBase *)s)->get_remaining(r); }
int ref_shadow_search_raw(void *s, const uint64_t *refbits, int rev, int shift, int ref_len, uint32_t *k) {
  return ((ShadowBase *)s)->search_raw(refbits, rev, shift, ref_len, k);
}
int ref_shadow_search_loop(void *s, const uint64_t *ref, const uint64_t *revref, int ref_len, uint32_t *k, int *shift,
                           int *rev) {
  return ((ShadowBase *)s)->search_loop(ref, revref, ref_len, k, shift, rev);
}

// ---- encoder units: real buildcontig + writecontig (encoder.cpp:32-109) on one contig.
// reads: NUL-separated strings; the seven output streams come back concatenated in `out` with their sizes in
// sizes[7] (seq, pos, noise, noisepos, order, RC, readlength); *abs_pos is advanced like the reference's.
long ref_u_contig(const char *reads, const int64_t *pos, const char *rc, const uint32_t *order, uint32_t count,
                  const char *tmpdir, uint64_t *abs_pos, uint8_t *out, long cap, uint64_t *sizes) {
  std::list<spring::contig_reads> lst;
  const char *p = reads;
  for (uint32_t i = 0; i < count; i++) {
    spring::contig_reads c;
    c.read = p;
    p += c.read.size() + 1;
    c.pos = pos[i];
    c.RC = rc[i];
    c.order = order[i];
    c.read_length = (uint16_t)c.read.size();
    lst.push_back(c);
  }
  spring::encoder_global eg;
  spring::encoder_global_b<64> egb(1);
  eg.max_readlen = 1;
  std::memset(eg.enc_noise, 0, sizeof(eg.enc_noise));
  spring::setglobalarrays<64>(eg, egb);  // enc_noise table (encoder.h:519-540)
  std::string base = std::string(tmpdir) + "/c.";
  static const char *names[7] = {"seq", "pos", "noise", "noisepos", "order", "RC", "readlength"};
  {
    std::ofstream f_seq(base + names[0]), f_pos(base + names[1], std::ios::binary), f_noise(base + names[2]),
        f_noisepos(base + names[3], std::ios::binary), f_order(base + names[4], std::ios::binary), f_RC(base + names[5]),
        f_readlength(base + names[6], std::ios::binary);
    uint32_t list_size = count;
    std::string ref = spring::buildcontig(lst, list_size);
    spring::writecontig(ref, lst, f_seq, f_pos, f_noise, f_noisepos, f_order, f_RC, f_readlength, eg, *abs_pos);
  }
  long o = 0;
  for (int i = 0; i < 7; i++) {
    std::string s = slurp(base + names[i]);
    if (o + (long)s.size() > cap) return -1;
    std::memcpy(out + o, s.data(), s.size());
    sizes[i] = s.size();
    o += s.size();
    std::remove((base + names[i]).c_str());
  }
  return o;
}

// real correct_order (encoder.cpp:177-222): order_s[numreads_s + numreads_N] in place, the per-tid order files
// `<dir>/read_order.bin.<tid>` rewritten in place (the reference deletes read_order_N.bin)
int ref_u_correct_order(uint32_t *order_s, uint32_t numreads, uint32_t numreads_s, uint32_t numreads_N, int num_thr,
                        const char *dir) {
  spring::encoder_global eg;
  eg.numreads = numreads;
  eg.numreads_s = numreads_s;
  eg.numreads_N = numreads_N;
  eg.num_thr = num_thr;
  eg.infile_order = std::string(dir) + "/read_order.bin";
  eg.infile_order_N = std::string(dir) + "/read_order_N.bin";
  spring::correct_order(order_s, eg);
  return 0;
}

// real 3-bit stringtobitset / bitsettostring with the encoder's basemask (encoder.h:496-517, :105-122)
int ref_u_enc_bits3_roundtrip(const char *s, int len, int L, uint64_t *limbs_out /* 24 limbs */, char *back) {
  if (3 * L > 1536) return -1;
  spring::encoder_global eg;
  eg.max_readlen = L;
  spring::encoder_global_b<1536> egb(L);
  spring::setglobalarrays<1536>(eg, egb);
  std::bitset<1536> b;
  spring::stringtobitset<1536>(std::string(s, len), (uint16_t)len, b, egb.basemask);
  std::memcpy(limbs_out, (const void *)&b, 1536 / 8);
  std::string r = spring::bitsettostring<1536>(b, (uint16_t)len, egb);
  std::memcpy(back, r.data(), len);
  return 0;
}

// real readsingletons (encoder.h:541-570): files `<dir>/temp.dna.singleton`, `<dir>/input_N.dna`,
// `<dir>/read_order.bin.singleton`, `<dir>/read_order_N.bin` -> 3-bit bitsets (24 limbs each), order_s, lengths
int ref_u_readsingletons(const char *dir, uint32_t numreads_s, uint32_t numreads_N, int L, uint64_t *limbs_out,
                         uint32_t *order_s, uint16_t *lens) {
  if (3 * L > 1536) return -1;
  spring::encoder_global eg;
  eg.max_readlen = L;
  eg.numreads_s = numreads_s;
  eg.numreads_N = numreads_N;
  eg.infile = std::string(dir) + "/temp.dna";
  eg.infile_N = std::string(dir) + "/input_N.dna";
  eg.infile_order = std::string(dir) + "/read_order.bin";
  eg.infile_order_N = std::string(dir) + "/read_order_N.bin";
  spring::encoder_global_b<1536> egb(L);
  spring::setglobalarrays<1536>(eg, egb);
  uint32_t m = numreads_s + numreads_N;
  std::bitset<1536> *read = new std::bitset<1536>[m ? m : 1];
  spring::readsingletons<1536>(read, order_s, lens, eg, egb);
  for (uint32_t i = 0; i < m; i++) std::memcpy(limbs_out + (size_t)i * 24, (const void *)&read[i], 1536 / 8);
  delete[] read;
  return 0;
}

}  // extern "C"
