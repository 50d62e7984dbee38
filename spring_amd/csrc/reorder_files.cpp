// spring_amd/csrc/reorder_files.cpp
//
// The drop-in stage: spring_reorder_run() keeps reorder_main<N>()'s temp-dir
// file contract (reference src/reorder.h:732-786 -> files opened by
// encoder_main<>, encoder.h:580-593) so the rest of the SPRING pipeline
// consumes the GPU stage's output unchanged.  Also the C++ mirror of the
// reference's operator interface (call_reorder.h).
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include "call_reorder.h"
#include "spring_encoder.h"
#include "spring_reorder.h"

namespace sr {
int fail(int code, const char *fmt, ...);
}
using sr::fail;

namespace {

// ---- gzip container with stored (uncompressed) deflate blocks.  The reference
// writes these four streams through boost::iostreams::gzip_compressor
// (reorder.h:355-368) and reads them back through gzip_decompressor
// (reorder.h:656-658, encoder.h:147-160); any valid RFC 1952 member is accepted.
uint32_t crc_tab[8][256];
bool crc_ready = false;
void crc_init() {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++)
    for (int t = 1; t < 8; t++) crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xff];
  crc_ready = true;
}
uint32_t crc32_buf(const uint8_t *p, size_t n) {
  if (!crc_ready) crc_init();
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t a, b;
    memcpy(&a, p, 4);
    memcpy(&b, p + 4, 4);
    a ^= c;
    c = crc_tab[7][a & 0xff] ^ crc_tab[6][(a >> 8) & 0xff] ^ crc_tab[5][(a >> 16) & 0xff] ^ crc_tab[4][a >> 24] ^
        crc_tab[3][b & 0xff] ^ crc_tab[2][(b >> 8) & 0xff] ^ crc_tab[1][(b >> 16) & 0xff] ^ crc_tab[0][b >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

int write_raw(const std::string &path, const void *data, size_t n) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return fail(SPRING_REORDER_E_IO, "cannot open %s for writing: %s", path.c_str(), strerror(errno));
  if (n && fwrite(data, 1, n, f) != n) {
    fclose(f);
    return fail(SPRING_REORDER_E_IO, "short write to %s", path.c_str());
  }
  if (fclose(f) != 0) return fail(SPRING_REORDER_E_IO, "close failed for %s", path.c_str());
  return 0;
}

int write_gzip_stored(const std::string &path, const void *data, size_t n) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return fail(SPRING_REORDER_E_IO, "cannot open %s for writing: %s", path.c_str(), strerror(errno));
  const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255};
  bool ok = fwrite(hdr, 1, 10, f) == 10;
  const uint8_t *p = (const uint8_t *)data;
  size_t left = n;
  do {
    const size_t blk = left > 65535 ? 65535 : left;
    const uint8_t bh[5] = {(uint8_t)(left == blk ? 1 : 0), (uint8_t)(blk & 0xff), (uint8_t)(blk >> 8),
                           (uint8_t)(~blk & 0xff), (uint8_t)((~blk >> 8) & 0xff)};
    ok = ok && fwrite(bh, 1, 5, f) == 5;
    if (blk) ok = ok && fwrite(p, 1, blk, f) == blk;
    p += blk;
    left -= blk;
  } while (left);
  const uint32_t crc = crc32_buf((const uint8_t *)data, n), isz = (uint32_t)n;
  uint8_t tr[8];
  memcpy(tr, &crc, 4);
  memcpy(tr + 4, &isz, 4);
  ok = ok && fwrite(tr, 1, 8, f) == 8;
  if (fclose(f) != 0 || !ok) return fail(SPRING_REORDER_E_IO, "write failed for %s", path.c_str());
  return 0;
}

int read_file(const std::string &path, std::vector<uint8_t> &buf) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return fail(SPRING_REORDER_E_IO, "cannot open %s: %s", path.c_str(), strerror(errno));
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  size_t old = buf.size();
  buf.resize(old + (size_t)sz);
  if (sz && fread(buf.data() + old, 1, (size_t)sz, f) != (size_t)sz) {
    fclose(f);
    return fail(SPRING_REORDER_E_IO, "short read from %s", path.c_str());
  }
  fclose(f);
  return 0;
}

int file_size(const std::string &path, size_t *sz) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return fail(SPRING_REORDER_E_IO, "cannot open %s: %s", path.c_str(), strerror(errno));
  fseek(f, 0, SEEK_END);
  *sz = (size_t)ftell(f);
  fclose(f);
  return 0;
}
int read_into(const std::string &path, uint8_t *dst, size_t sz) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return fail(SPRING_REORDER_E_IO, "cannot open %s: %s", path.c_str(), strerror(errno));
  const bool ok = !sz || fread(dst, 1, sz, f) == sz;
  fclose(f);
  return ok ? 0 : fail(SPRING_REORDER_E_IO, "short read from %s", path.c_str());
}

struct CtxGuard {
  spring_reorder_ctx *c = nullptr;
  ~CtxGuard() { spring_reorder_destroy(c); }
};

}  // namespace

extern "C" int spring_reorder_run(const char *temp_dir, uint32_t max_readlen, int32_t num_thr, int32_t paired_end,
                                  uint32_t n0, uint32_t n1, const spring_reorder_opts *opts) {
  const bool dbg = getenv("SPRING_REORDER_DEBUG") != nullptr;  // phase timings on stderr, no effect on results
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[run] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  if (!temp_dir) return fail(SPRING_REORDER_E_ARG, "temp_dir is NULL");
  if (num_thr <= 0) return fail(SPRING_REORDER_E_ARG, "num_thr must be >= 1");
  spring_reorder_opts o;
  if (opts) o = *opts; else spring_reorder_default_opts(&o);
  o.num_thr = num_thr;
  const std::string base(temp_dir);
  const std::string in1 = base + "/input_clean_1.dna", in2 = base + "/input_clean_2.dna";  // reorder.h:738-739
  const uint64_t ntot = (uint64_t)n0 + (paired_end ? n1 : 0);                              // reorder.h:761
  if (ntot > 4294967290ull) return fail(SPRING_REORDER_E_ARG, "too many reads");           // params.h:24
  const uint32_t n = (uint32_t)ntot;

  // both input files into one buffer that is not zero-filled first (4 GB at 100 M reads)
  size_t sz1 = 0, sz2 = 0;
  int r = file_size(in1, &sz1);
  if (r) return r;
  if (paired_end && (r = file_size(in2, &sz2))) return r;
  std::unique_ptr<uint8_t[]> dna(new uint8_t[sz1 + sz2 + 1]);
  if ((r = read_into(in1, dna.get(), sz1))) return r;
  if (paired_end && (r = read_into(in2, dna.get() + sz1, sz2))) return r;  // file-2 reads follow in the same pool (reorder.h:233-242)
  lap("read input files");
  CtxGuard g;
  r = spring_reorder_create(&g.c, &o);
  if (r) return r;
  r = spring_reorder_load_dna(g.c, dna.get(), sz1 + sz2, n, max_readlen);
  if (r) return r;
  dna.reset();
  lap("create + load (H2D)");
  remove(in1.c_str());  // the stage consumes its inputs (reorder.h:232,241)
  if (paired_end) remove(in2.c_str());
  if ((r = spring_reorder_build_dict(g.c))) return r;
  if ((r = spring_reorder_run_chains(g.c))) return r;
  if ((r = spring_reorder_finalize(g.c))) return r;
  spring_reorder_stats st;
  if ((r = spring_reorder_get_stats(g.c, &st))) return r;
  lap("dict + chains + final");

  const size_t nm = st.n_matched, ns = st.n_single;
  // not zero-filled: the download overwrites every byte (1.6 GB at 100 M reads)
  std::unique_ptr<uint32_t[]> order(new uint32_t[nm + 1]), order_s(new uint32_t[ns + 1]);
  std::unique_ptr<char[]> rc(new char[nm + 1]), flag(new char[nm + 1]);
  std::unique_ptr<int64_t[]> pos(new int64_t[nm + 1]);
  std::unique_ptr<uint16_t[]> rlen(new uint16_t[nm + 1]);
  std::vector<uint64_t> toff(num_thr + 1), toff_s(num_thr + 1);
  r = spring_reorder_download(g.c, order.get(), rc.get(), flag.get(), pos.get(), rlen.get(), order_s.get(),
                              toff.data(), toff_s.data());
  if (r) return r;
  lap("download streams");
  // temp.dna.<tid> / temp.dna.singleton are built on the device (reverse complement + repack), one stream after the
  // other; the file sets are then written by one host thread per tid (CRC-32 + file system, the slow half)
  std::vector<std::vector<uint8_t>> dna_t(num_thr + 1);
  for (int t = 0; t <= num_thr; t++) {
    const int32_t which = t < num_thr ? t : -1;
    size_t nb = 0;
    if ((r = spring_reorder_emit_dna(g.c, which, nullptr, 0, &nb))) return r;
    dna_t[t].resize(nb ? nb : 1);
    if ((r = spring_reorder_emit_dna(g.c, which, dna_t[t].data(), nb, &nb))) return r;
    dna_t[t].resize(nb);
  }
  lap("emit temp.dna (D2H)");
  if (!crc_ready) crc_init();
  std::vector<int> trc(num_thr, 0);
  std::vector<std::string> terr(num_thr);
  auto write_tid = [&](int t) {  // all six files must exist for every tid (encoder.h:147-175)
    const std::string ts = "." + std::to_string(t);
    const size_t a = toff[t], c = toff[t + 1] - toff[t];
    int e;
    if ((e = write_raw(base + "/read_order.bin" + ts, order.get() + a, c * 4)) ||
        (e = write_gzip_stored(base + "/read_rev.txt" + ts, rc.get() + a, c)) ||
        (e = write_gzip_stored(base + "/tempflag.txt" + ts, flag.get() + a, c)) ||
        (e = write_gzip_stored(base + "/temppos.txt" + ts, pos.get() + a, c * 8)) ||
        (e = write_gzip_stored(base + "/read_lengths.bin" + ts, rlen.get() + a, c * 2)) ||
        (e = write_raw(base + "/temp.dna" + ts, dna_t[t].data(), dna_t[t].size()))) {
      trc[t] = e;
      terr[t] = spring_reorder_last_error();  // the message is thread-local
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < num_thr; t++) th.emplace_back(write_tid, t);
    write_tid(0);
    for (auto &x : th) x.join();
  }
  for (int t = 0; t < num_thr; t++)
    if (trc[t]) return fail(trc[t], "%s", terr[t].c_str());
  if ((r = write_raw(base + "/temp.dna.singleton", dna_t[num_thr].data(), dna_t[num_thr].size()))) return r;  // reorder.h:704-728
  if ((r = write_raw(base + "/read_order.bin.singleton", order_s.get(), ns * 4))) return r;
  const uint32_t numreads_s = (uint32_t)ns;
  if ((r = write_raw(base + "/temp.dna.singleton.count", &numreads_s, 4))) return r;  // reorder.h:699-701
  lap("write files");
  printf("Reordering done, %llu were unmatched\n", (unsigned long long)st.unmatched);  // reorder.h:633-635
  return 0;
}

namespace {
struct EncGuard {
  spring_encoder_ctx *c = nullptr;
  ~EncGuard() { spring_encoder_destroy(c); }
};
// the files encoder_main leaves behind (encoder.h:365-494), read_seq as .tmp + .tail (state before BSC_compress)
int write_encoder_files(const std::string &base, spring_encoder_ctx *ec, const spring_encoder_info &I, int num_thr) {
  int r;
  std::vector<uint64_t> seq_len_tid(num_thr), pos(I.n_aligned ? I.n_aligned : 1);
  std::vector<char> noise(I.noise_bytes ? I.noise_bytes : 1), rc(I.n_aligned ? I.n_aligned : 1);
  std::vector<uint16_t> noisepos(I.n_noisepos ? I.n_noisepos : 1), rlen(I.n_total ? I.n_total : 1);
  std::vector<uint32_t> order(I.n_total ? I.n_total : 1);
  std::vector<uint8_t> un(I.unaligned_bytes ? I.unaligned_bytes : 1);
  if ((r = spring_encoder_download(ec, nullptr, seq_len_tid.data(), pos.data(), noise.data(), noisepos.data(),
                                   order.data(), rlen.data(), rc.data(), un.data())))
    return r;
  uint64_t packed_total = 0;
  for (int t = 0; t < num_thr; t++) packed_total += seq_len_tid[t] / 4;
  std::vector<uint8_t> packed(packed_total ? packed_total : 1);
  std::vector<char> tail((size_t)num_thr * 4);
  if ((r = spring_encoder_download_seq_packed(ec, packed.data(), tail.data()))) return r;
  uint64_t po = 0;
  for (int t = 0; t < num_thr; t++) {  // pack_compress_seq up to the BSC call (encoder.cpp:111-150)
    const std::string ts = "." + std::to_string(t);
    if ((r = write_raw(base + "/read_seq.bin" + ts + ".tmp", packed.data() + po, seq_len_tid[t] / 4))) return r;
    if ((r = write_raw(base + "/read_seq.bin" + ts + ".tail", tail.data() + 4 * t, seq_len_tid[t] % 4))) return r;
    po += seq_len_tid[t] / 4;
  }
  if ((r = write_raw(base + "/read_pos.bin", pos.data(), I.n_aligned * 8))) return r;          // encoder.h:465-480
  if ((r = write_raw(base + "/read_noise.txt", noise.data(), I.noise_bytes))) return r;         // encoder.h:365-395
  if ((r = write_raw(base + "/read_noisepos.bin", noisepos.data(), I.n_noisepos * 2))) return r;
  if ((r = write_raw(base + "/read_order.bin", order.data(), I.n_total * 4))) return r;         // + unaligned, :425-445
  if ((r = write_raw(base + "/read_rev.txt", rc.data(), I.n_aligned))) return r;
  if ((r = write_raw(base + "/read_lengths.bin", rlen.data(), I.n_total * 2))) return r;
  if ((r = write_raw(base + "/read_unaligned.txt", un.data(), I.unaligned_bytes))) return r;
  if ((r = write_raw(base + "/read_unaligned.txt.count", &I.len_unaligned, 8))) return r;      // encoder.h:457-460
  return 0;
}

// whole file -> memory through zlib (accepts the reference's boost::iostreams gzip members and our stored ones)
int read_gz(const std::string &path, std::vector<uint8_t> &buf) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) return fail(SPRING_REORDER_E_IO, "cannot open %s", path.c_str());
  gzbuffer(f, 1 << 20);
  uint8_t tmp[1 << 16];
  for (;;) {
    const int k = gzread(f, tmp, sizeof(tmp));
    if (k < 0) { gzclose(f); return fail(SPRING_REORDER_E_IO, "gzip error in %s", path.c_str()); }
    if (k == 0) break;
    buf.insert(buf.end(), tmp, tmp + k);
  }
  gzclose(f);
  return 0;
}
bool file_exists(const std::string &p) {
  FILE *f = fopen(p.c_str(), "rb");
  if (f) fclose(f);
  return f != nullptr;
}
}  // namespace

extern "C" int spring_reorder_encode_run(const char *temp_dir, uint32_t max_readlen, int32_t num_thr, int32_t paired_end,
                                         uint32_t n0, uint32_t n1, uint32_t num_reads, const spring_reorder_opts *opts,
                                         spring_encoder_info *info_out) {
  if (!temp_dir) return fail(SPRING_REORDER_E_ARG, "temp_dir is NULL");
  if (num_thr <= 0) return fail(SPRING_REORDER_E_ARG, "num_thr must be >= 1");
  spring_reorder_opts o;
  if (opts) o = *opts; else spring_reorder_default_opts(&o);
  o.num_thr = num_thr;
  const std::string base(temp_dir);
  const std::string in1 = base + "/input_clean_1.dna", in2 = base + "/input_clean_2.dna";
  const std::string inN = base + "/input_N.dna", inON = base + "/read_order_N.bin";  // encoder.h:587,:583
  const uint64_t ntot = (uint64_t)n0 + (paired_end ? n1 : 0);
  if (ntot > 4294967290ull || num_reads < ntot) return fail(SPRING_REORDER_E_ARG, "bad read counts");
  const uint32_t n = (uint32_t)ntot, nN = num_reads - n;  // getDataParams, encoder.cpp:158-175

  std::vector<uint8_t> dna, dnaN, ordN;
  int r = read_file(in1, dna);
  if (r) return r;
  if (paired_end && (r = read_file(in2, dna))) return r;
  if (nN) {
    if ((r = read_file(inN, dnaN))) return r;
    if ((r = read_file(inON, ordN))) return r;
    if (ordN.size() < (size_t)nN * 4) return fail(SPRING_REORDER_E_IO, "%s is too short", inON.c_str());
  }
  CtxGuard g;
  EncGuard e;
  if ((r = spring_reorder_create(&g.c, &o))) return r;
  if ((r = spring_reorder_load_dna(g.c, dna.data(), dna.size(), n, max_readlen))) return r;
  std::vector<uint8_t>().swap(dna);
  if ((r = spring_reorder_build_dict(g.c))) return r;
  if ((r = spring_reorder_run_chains(g.c))) return r;
  if ((r = spring_reorder_finalize(g.c))) return r;
  spring_reorder_stats st;
  if ((r = spring_reorder_get_stats(g.c, &st))) return r;
  printf("Reordering done, %llu were unmatched\n", (unsigned long long)st.unmatched);  // reorder.h:633-635
  if ((r = spring_encoder_create(o.device, &e.c))) return r;
  spring_encoder_info I;
  if ((r = spring_encoder_encode_reorder(e.c, g.c, dnaN.data(), dnaN.size(), (const uint32_t *)ordN.data(), nN, &I)))
    return r;

  if ((r = write_encoder_files(base, e.c, I, num_thr))) return r;
  remove(in1.c_str());
  if (paired_end) remove(in2.c_str());
  if (file_exists(inN)) remove(inN.c_str());    // encoder.h:601
  if (file_exists(inON)) remove(inON.c_str());  // encoder.cpp:218
  printf("Encoding done:\n%u singleton reads were aligned\n%u reads with N were aligned\n", I.matched_s,
         I.matched_N);  // encoder.h:489-491
  if (info_out) *info_out = I;
  return 0;
}

extern "C" int spring_encoder_run(const char *temp_dir, uint32_t max_readlen, int32_t num_thr, uint32_t num_reads,
                                  uint32_t num_reads_clean, int32_t device, spring_encoder_info *info_out) {
  if (!temp_dir) return fail(SPRING_REORDER_E_ARG, "temp_dir is NULL");
  if (num_thr <= 0) return fail(SPRING_REORDER_E_ARG, "num_thr must be >= 1");
  if (num_reads < num_reads_clean) return fail(SPRING_REORDER_E_ARG, "bad read counts");
  const std::string base(temp_dir);
  const std::string f_dna = base + "/temp.dna", f_pos = base + "/temppos.txt", f_flag = base + "/tempflag.txt",
                    f_order = base + "/read_order.bin", f_rc = base + "/read_rev.txt", f_len = base + "/read_lengths.bin",
                    f_N = base + "/input_N.dna", f_oN = base + "/read_order_N.bin";  // encoder.h:580-593
  int r;
  std::vector<uint8_t> cnt;
  if ((r = read_file(f_dna + ".singleton.count", cnt))) return r;  // getDataParams, encoder.cpp:158-175
  if (cnt.size() < 4) return fail(SPRING_REORDER_E_IO, "temp.dna.singleton.count is too short");
  uint32_t ns;
  memcpy(&ns, cnt.data(), 4);
  const uint32_t nN = num_reads - num_reads_clean;
  std::vector<uint8_t> dna, order, rc, flag, pos, rlen, dna_s, order_s, dnaN, ordN;
  std::vector<uint64_t> tid_count(num_thr);
  for (int t = 0; t < num_thr; t++) {
    const std::string ts = "." + std::to_string(t);
    const size_t before = order.size();
    if ((r = read_file(f_order + ts, order))) return r;
    tid_count[t] = (order.size() - before) / 4;
    if ((r = read_file(f_dna + ts, dna))) return r;
    if ((r = read_gz(f_rc + ts, rc))) return r;
    if ((r = read_gz(f_flag + ts, flag))) return r;
    if ((r = read_gz(f_pos + ts, pos))) return r;
    if ((r = read_gz(f_len + ts, rlen))) return r;
  }
  const uint64_t M = order.size() / 4;
  if (rc.size() != M || flag.size() != M || pos.size() != M * 8 || rlen.size() != M * 2)
    return fail(SPRING_REORDER_E_IO, "per-tid streams disagree on the number of reads");
  if ((uint64_t)M + ns != num_reads_clean) return fail(SPRING_REORDER_E_ARG, "streams hold %llu + %u reads, expected %u clean",
                                                      (unsigned long long)M, ns, num_reads_clean);
  if ((r = read_file(f_dna + ".singleton", dna_s))) return r;
  if ((r = read_file(f_order + ".singleton", order_s))) return r;
  if (order_s.size() < (size_t)ns * 4) return fail(SPRING_REORDER_E_IO, "read_order.bin.singleton is too short");
  if (nN) {
    if ((r = read_file(f_N, dnaN))) return r;
    if ((r = read_file(f_oN, ordN))) return r;
    if (ordN.size() < (size_t)nN * 4) return fail(SPRING_REORDER_E_IO, "read_order_N.bin is too short");
  }
  EncGuard e;
  if ((r = spring_encoder_create(device, &e.c))) return r;
  spring_encoder_info I;
  r = spring_encoder_encode_host(e.c, max_readlen, num_thr, tid_count.data(), dna.data(), dna.size(),
                                 (const uint32_t *)order.data(), (const char *)rc.data(), (const char *)flag.data(),
                                 (const int64_t *)pos.data(), (const uint16_t *)rlen.data(), dna_s.data(), dna_s.size(),
                                 (const uint32_t *)order_s.data(), ns, dnaN.data(), dnaN.size(),
                                 (const uint32_t *)ordN.data(), nN, &I);
  if (r) return r;
  // inputs are consumed (encoder.h:412-422, :556, :565, :601; encoder.cpp:163-167, :218) before the outputs of the
  // same name (read_order.bin, read_rev.txt, read_lengths.bin) are written
  for (int t = 0; t < num_thr; t++) {
    const std::string ts = "." + std::to_string(t);
    for (const std::string &f : {f_order, f_dna, f_rc, f_flag, f_pos, f_len}) remove((f + ts).c_str());
  }
  remove((f_dna + ".singleton").c_str());
  remove((f_dna + ".singleton.count").c_str());
  remove((f_order + ".singleton").c_str());
  if (file_exists(f_N)) remove(f_N.c_str());
  if (file_exists(f_oN)) remove(f_oN.c_str());
  if ((r = write_encoder_files(base, e.c, I, num_thr))) return r;
  printf("Encoding done:\n%u singleton reads were aligned\n%u reads with N were aligned\n", I.matched_s, I.matched_N);
  if (info_out) *info_out = I;
  return 0;
}

namespace spring_amd {
void call_reorder(const std::string &temp_dir, const reorder_params &cp, const spring_reorder_opts *opts) {
  const size_t bitset_size_reorder = (2 * (size_t)cp.max_readlen - 1) / 64 * 64 + 64;
  if (cp.max_readlen == 0 || bitset_size_reorder > 1024) throw std::runtime_error("Wrong bitset size.");
  int r = spring_reorder_run(temp_dir.c_str(), cp.max_readlen, cp.num_thr, cp.paired_end ? 1 : 0,
                             cp.num_reads_clean[0], cp.num_reads_clean[1], opts);
  if (r != 0) throw std::runtime_error(std::string("spring_reorder_run: ") + spring_reorder_last_error());
}
void call_encoder(const std::string &temp_dir, const reorder_params &cp, uint32_t num_reads, int device) {
  const size_t bitset_size_encoder = (3 * (size_t)cp.max_readlen - 1) / 64 * 64 + 64;  // call_template_functions.cpp:66
  if (cp.max_readlen == 0 || bitset_size_encoder > 1536) throw std::runtime_error("Wrong bitset size.");
  int r = spring_encoder_run(temp_dir.c_str(), cp.max_readlen, cp.num_thr, num_reads,
                             cp.num_reads_clean[0] + (cp.paired_end ? cp.num_reads_clean[1] : 0), device, nullptr);
  if (r != 0) throw std::runtime_error(std::string("spring_encoder_run: ") + spring_reorder_last_error());
}
void call_reorder_encoder(const std::string &temp_dir, const reorder_params &cp, uint32_t num_reads,
                          const spring_reorder_opts *opts) {
  const size_t bitset_size_reorder = (2 * (size_t)cp.max_readlen - 1) / 64 * 64 + 64;
  if (cp.max_readlen == 0 || bitset_size_reorder > 1024) throw std::runtime_error("Wrong bitset size.");
  int r = spring_reorder_encode_run(temp_dir.c_str(), cp.max_readlen, cp.num_thr, cp.paired_end ? 1 : 0,
                                    cp.num_reads_clean[0], cp.num_reads_clean[1], num_reads, opts, nullptr);
  if (r != 0) throw std::runtime_error(std::string("spring_reorder_encode_run: ") + spring_reorder_last_error());
}
}  // namespace spring_amd
