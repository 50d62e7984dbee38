// spring_amd/csrc/reorder_files.cpp
//
// The drop-in stage: spring_reorder_run() keeps reorder_main<N>()'s temp-dir
// file contract (reference src/reorder.h:732-786 -> files opened by
// encoder_main<>, encoder.h:580-593) so the rest of the SPRING pipeline
// consumes the GPU stage's output unchanged.  Also the C++ mirror of the
// reference's operator interface (call_reorder.h).
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

#include <hip/hip_runtime.h>

#include "call_reorder.h"
#include "reorder_internal.h"
#include "spring_encoder.h"
#include "spring_reorder.h"

#ifndef SR_FILES_EXP
#define SR_FILES_EXP 0
#endif

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
// running CRC-32 (state = the complemented register; start with 0xFFFFFFFF, finish with ^ 0xFFFFFFFF)
uint32_t crc32_update(uint32_t c, const uint8_t *p, size_t n) {
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
  return c;
}
uint32_t crc32_buf(const uint8_t *p, size_t n) {
  if (!crc_ready) crc_init();
  return crc32_update(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
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
  struct stat sb;
  if (fstat(fileno(f), &sb) != 0 || sb.st_size < 0) {
    fclose(f);
    return fail(SPRING_REORDER_E_IO, "cannot size %s", path.c_str());
  }
  const size_t sz = (size_t)sb.st_size, old = buf.size();
  buf.resize(old + sz);
  if (sz && fread(buf.data() + old, 1, sz, f) != sz) {
    fclose(f);
    return fail(SPRING_REORDER_E_IO, "short read from %s", path.c_str());
  }
  fclose(f);
  return 0;
}

int file_size(const std::string &path, size_t *sz) {
  struct stat sb;
  if (stat(path.c_str(), &sb) != 0) return fail(SPRING_REORDER_E_IO, "cannot open %s: %s", path.c_str(), strerror(errno));
  if (sb.st_size < 0) return fail(SPRING_REORDER_E_IO, "cannot size %s", path.c_str());
  *sz = (size_t)sb.st_size;
  return 0;
}

struct CtxGuard {
  spring_reorder_ctx *c = nullptr;
  ~CtxGuard() { spring_reorder_destroy(c); }
};

// ---- input: input_clean_1.dna (+ input_clean_2.dna) as ONE record stream read piecewise with pread, from as many
// threads as the uploader runs (sr::load_dna_source): file 2's records follow file 1's (reorder.h:233-242)
struct FilePair {
  int fd[2] = {-1, -1};
  size_t sz[2] = {0, 0};
  std::string name[2];
  void close_all() { for (int k = 0; k < 2; k++) if (fd[k] >= 0) { close(fd[k]); fd[k] = -1; } }
  ~FilePair() { close_all(); }
  static int fill(void *self, size_t off, void *dst, size_t len) {
    FilePair *f = (FilePair *)self;
    uint8_t *d = (uint8_t *)dst;
    while (len) {
      const int k = off < f->sz[0] ? 0 : 1;
      if (k && off - f->sz[0] >= f->sz[1]) return fail(SPRING_REORDER_E_IO, "read past the end of %s", f->name[1].c_str());
      const size_t o = k ? off - f->sz[0] : off, take = std::min(len, f->sz[k] - o);
      size_t got = 0;
      while (got < take) {
        const ssize_t g = pread(f->fd[k], d + got, take - got, (off_t)(o + got));
        if (g < 0 && errno == EINTR) continue;
        if (g <= 0) return fail(SPRING_REORDER_E_IO, "short read from %s", f->name[k].c_str());
        got += (size_t)g;
      }
      d += take; off += take; len -= take;
    }
    return 0;
  }
};

// ---- output: the streams go device -> pinned slot -> file.  Every output file belongs to one writer thread (its
// slots are written in order, straight from the pinned memory with write / writev: no stdio copy); the files are dealt
// out over the writers largest first, and the submitting thread hands the slots out ACROSS the writers -- the next one to
// the writer with the most bytes still to come, at most three in flight per writer -- so that all of them work from the
// first copy on and finish together (round 4 submitted tid by tid: the first writer's backlog held most of the ring
// while the others waited -- 8.3 GB/s for 5.45 GB; profiles/r05_files.txt).
// (the "copy done" event of a slot is made per copy, under the device whose stream records it -- an event belongs to the
// device that was current when it was created, and the slots of this ring serve every rank of a multi-GPU call -- and
// is destroyed by the writer thread once it has waited for it)
constexpr size_t OUT_SLOT = (size_t)8 << 20;                 // bytes per slot
constexpr int SLOTS_PER_CHUNK = (int)(sr::PIN_CHUNK / OUT_SLOT);
struct Slot { void *pin = nullptr; hipEvent_t ev = nullptr; };
class Ring {  // a bounded set of pinned slots, cut from the library's cached pinned chunks
 public:
  explicit Ring(int nchunks) : cap_(nchunks) {}
  ~Ring() { for (void *c : chunks_) sr::pinned_put(c); }
  bool acquire(Slot *out) {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      if (!free_.empty()) { *out = free_.back(); free_.pop_back(); return true; }
      if ((int)chunks_.size() < cap_) {
        void *c = sr::pinned_get();
        if (!c) {
          if (chunks_.empty()) return false;   // not a single chunk can be pinned
          cap_ = (int)chunks_.size();          // no more pinned memory: carry on with the slots there are
          continue;
        }
        chunks_.push_back(c);
        for (int i = SLOTS_PER_CHUNK - 1; i >= 0; i--) {
          Slot s;
          s.pin = (uint8_t *)c + (size_t)i * OUT_SLOT;
          free_.push_back(s);
        }
        continue;
      }
      cv_.wait(lk);
    }
  }
  void release(Slot s) {
    s.ev = nullptr;
    { std::lock_guard<std::mutex> lk(mu_); free_.push_back(s); }
    cv_.notify_one();
  }
 private:
  int cap_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<void *> chunks_;
  std::vector<Slot> free_;
};

// Events are REUSED: the first hipEventSynchronize on a freshly created event costs ~0.3 ms on this stack (650 fresh events,
// each recorded on an idle stream and waited for: 206 ms; the same with events that had been used before: 9 ms --
// tools/r5 notes in profiles/r05_files.txt), and an output of 5.45 GB is 688 copies.  An event belongs to the device that was
// current when it was created, hence one free list per device.
class EventPool {
 public:
  ~EventPool() { for (auto &v : free_) for (hipEvent_t e : v.second) (void)hipEventDestroy(e); }
  hipEvent_t get(int dev) {  // (device `dev` is current)
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto &v = list(dev);
      if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
  }
  void put(int dev, hipEvent_t e) {
    std::lock_guard<std::mutex> lk(mu_);
    list(dev).push_back(e);
  }
 private:
  std::vector<hipEvent_t> &list(int dev) {
    for (auto &v : free_) if (v.first == dev) return v.second;
    free_.push_back({dev, {}});
    return free_.back().second;
  }
  std::mutex mu_;
  std::vector<std::pair<int, std::vector<hipEvent_t>>> free_;
};

// "the copy into slot k has landed": set by the ONE thread per stream that waits for the copies' events in order, read by
// the writers.  (Round 5, first version: every writer called hipEventSynchronize for its own slots -- 24 threads inside
// the runtime beside the submitting one -- and the copies of a 5.45 GB output ran at 23 GB/s on a link that does 52:
// the writers spent 130-150 ms of a 245 ms leg waiting for copies, profiles/r05_files.txt.)
struct Done {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<char> f;
  int bad = 0;
  void set(size_t k, bool ok) {
    { std::lock_guard<std::mutex> lk(mu); f[k] = 1; if (!ok) bad = 1; }
    cv.notify_all();
  }
  bool wait(size_t k) {  // false: a copy failed
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return f[k] != 0; });
    return bad == 0;
  }
};
class CopyWaiter {  // one per stream that copies are issued on
 public:
  CopyWaiter(Done *d, EventPool *pool, int dev) : done_(d), pool_(pool), dev_(dev) {}
  void start() { th_ = std::thread([this] { run(); }); }
  void push(hipEvent_t ev, size_t k) {
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back({ev, k}); }
    cv_.notify_one();
  }
  void finish() {
    if (!th_.joinable()) return;
    push(nullptr, (size_t)-1);
    th_.join();
  }
  ~CopyWaiter() { finish(); }
 private:
  void run() {
    for (;;) {
      std::pair<hipEvent_t, size_t> e;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        e = q_.front();
        q_.pop_front();
      }
      if (!e.first) return;
      const bool ok = hipEventSynchronize(e.first) == hipSuccess;
      pool_->put(dev_, e.first);
      done_->set(e.second, ok);
    }
  }
  Done *done_;
  EventPool *pool_;
  int dev_;
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::pair<hipEvent_t, size_t>> q_;
};

struct Msg {
  enum Kind { OPEN_RAW, OPEN_GZ, DATA, BYTES, CLOSE, STOP } kind = STOP;
  std::string path;
  Slot slot;
  size_t len = 0;
  size_t seq = 0;              // DATA: index of the copy's "landed" flag (Done)
  std::vector<uint8_t> bytes;  // BYTES: small host data
};
class Writer {
 public:
  Writer(Ring *ring, Done *done) : ring_(ring), done_(done) {}
  void start() { th_ = std::thread([this] { run(); }); }
  void push(Msg &&m) {
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(m)); }
    cv_.notify_one();
  }
  void finish() {
    if (!th_.joinable()) return;
    Msg m;
    m.kind = Msg::STOP;
    push(std::move(m));
    th_.join();
  }
  ~Writer() { finish(); }
  int rc = 0;
  std::string err;
  std::atomic<int> inflight{0};  // slots queued to this writer and not yet written (the submitter's pacing)
  double t_wait = 0, t_put = 0, t_idle = 0;  // seconds: waiting for copies / writing / waiting for messages (opts.debug)
  size_t nslots = 0;
 private:
  void bad(const char *what, const std::string &path) {
    if (!rc) { rc = SPRING_REORDER_E_IO; err = std::string(what) + " " + path + ": " + strerror(errno); }
  }
  void put_iov(struct iovec *iov, int cnt) {  // all of it, whatever the kernel takes per call
    while (cnt > 0) {
      const ssize_t w = writev(fd_, iov, cnt);
      if (w < 0 && errno == EINTR) continue;
      if (w < 0) { bad("short write to", path_); return; }
      size_t left = (size_t)w;
      while (cnt > 0 && left >= iov->iov_len) { left -= iov->iov_len; iov++; cnt--; }
      if (cnt > 0) { iov->iov_base = (uint8_t *)iov->iov_base + left; iov->iov_len -= left; }
    }
  }
  void put(const uint8_t *p, size_t n) {
    if (fd_ < 0 || rc || !n) return;
    if (!gz_) {
      struct iovec v = {(void *)p, n};
      put_iov(&v, 1);
      return;
    }
    crc_ = crc32_update(crc_, p, n);
    isize_ += n;
    // stored deflate blocks, none of them final: the final (empty) one is written when the stream closes
    constexpr int MAXB = 256;
    uint8_t bh[MAXB][5];
    struct iovec iov[2 * MAXB];
    while (n && !rc) {
      int nb = 0;
      while (n && nb < MAXB) {
        const size_t blk = n > 65535 ? 65535 : n;
        bh[nb][0] = 0; bh[nb][1] = (uint8_t)(blk & 0xff); bh[nb][2] = (uint8_t)(blk >> 8);
        bh[nb][3] = (uint8_t)(~blk & 0xff); bh[nb][4] = (uint8_t)((~blk >> 8) & 0xff);
        iov[2 * nb] = {bh[nb], 5};
        iov[2 * nb + 1] = {(void *)p, blk};
        p += blk; n -= blk; nb++;
      }
      put_iov(iov, 2 * nb);
    }
  }
  void run() {
    for (;;) {
      Msg m;
      {
        const auto t0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        m = std::move(q_.front());
        q_.pop_front();
        t_idle += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      switch (m.kind) {
        case Msg::STOP: if (fd_ >= 0) close(fd_); return;
        case Msg::OPEN_RAW: case Msg::OPEN_GZ: {
          path_ = m.path; gz_ = m.kind == Msg::OPEN_GZ; crc_ = 0xFFFFFFFFu; isize_ = 0;
          fd_ = rc ? -1 : open(path_.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
          if (fd_ < 0 && !rc) bad("cannot open for writing", path_);
          if (fd_ >= 0 && gz_) {
            uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255};
            struct iovec v = {hdr, 10};
            put_iov(&v, 1);
          }
          break;
        }
        case Msg::DATA: {
          const auto ta = std::chrono::steady_clock::now();
          if (!done_->wait(m.seq) && !rc) { rc = SPRING_REORDER_E_HIP; err = "device to host copy failed"; }
          const auto tb = std::chrono::steady_clock::now();
#if SR_FILES_EXP != 1  // (experiment builds: 1 = the writers drop the data, 2 = no device-to-host copies; what bounds the output leg)
          put((const uint8_t *)m.slot.pin, m.len);
#endif
          const auto tc = std::chrono::steady_clock::now();
          t_wait += std::chrono::duration<double>(tb - ta).count(); t_put += std::chrono::duration<double>(tc - tb).count(); nslots++;
          ring_->release(m.slot);
          inflight.fetch_sub(1, std::memory_order_release);
          break;
        }
        case Msg::BYTES: put(m.bytes.data(), m.bytes.size()); break;
        case Msg::CLOSE:
          if (fd_ >= 0 && gz_ && !rc) {
            uint8_t fin[13] = {1, 0, 0, 0xff, 0xff};  // final stored block, empty; then CRC-32 and ISIZE
            const uint32_t crc = crc_ ^ 0xFFFFFFFFu, isz = (uint32_t)isize_;
            memcpy(fin + 5, &crc, 4); memcpy(fin + 9, &isz, 4);
            struct iovec v = {fin, 13};
            put_iov(&v, 1);
          }
          if (fd_ >= 0 && close(fd_) != 0) bad("close failed for", path_);
          fd_ = -1;
          break;
      }
    }
  }
  Ring *ring_;
  Done *done_;
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Msg> q_;
  int fd_ = -1;
  bool gz_ = false;
  uint32_t crc_ = 0;
  uint64_t isize_ = 0;
  std::string path_;
};

// One output file: bytes of one or more device ranges (rank by rank), or a few host bytes
struct OutSeg { int dev; hipStream_t st; const uint8_t *d; size_t n; };
struct OutFile {
  std::string path;
  bool gz = false;
  std::vector<OutSeg> segs;
  std::vector<uint8_t> bytes;
  size_t total() const { size_t t = bytes.size(); for (const OutSeg &s : segs) t += s.n; return t; }
};
// the next slot of device bytes of file f (cursor: segment si, offset so) -> writer w.  0, or a failure code
struct SubmitClock { double t_ring = 0, t_copy = 0, t_event = 0; };  // seconds the submitting thread spent per step (opts.debug)
static inline double secs_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
int w_next_slot(Writer &w, Ring &ring, CopyWaiter &cw, EventPool &evs, size_t seq, bool *sent, const OutFile &f, size_t &si, size_t &so, SubmitClock &clk) {
  *sent = false;
  while (si < f.segs.size() && so >= f.segs[si].n) { si++; so = 0; }
  if (si >= f.segs.size()) return 0;
  const OutSeg &s = f.segs[si];
  const size_t len = std::min(OUT_SLOT, s.n - so);
  if (hipSetDevice(s.dev) != hipSuccess) return fail(SPRING_REORDER_E_HIP, "hipSetDevice(%d) failed", s.dev);
  Msg m;
  m.kind = Msg::DATA;
  m.len = len;
  auto tk = std::chrono::steady_clock::now();
  if (!ring.acquire(&m.slot)) return fail(SPRING_REORDER_E_HIP, "cannot pin a staging chunk for the output streams");
  clk.t_ring += secs_since(tk); tk = std::chrono::steady_clock::now();
  if (!(m.slot.ev = evs.get(s.dev))) {  // (device s.dev is current)
    ring.release(m.slot);
    return fail(SPRING_REORDER_E_HIP, "cannot create an event on device %d", s.dev);
  }
  clk.t_event += secs_since(tk); tk = std::chrono::steady_clock::now();
  hipError_t ce = hipSuccess;
#if SR_FILES_EXP != 2
  ce = hipMemcpyAsync(m.slot.pin, s.d + so, len, hipMemcpyDeviceToHost, s.st);
#endif
  clk.t_copy += secs_since(tk); tk = std::chrono::steady_clock::now();
  const hipError_t re = hipEventRecord(m.slot.ev, s.st);
  clk.t_event += secs_since(tk);
  if (ce != hipSuccess || re != hipSuccess) {
    (void)hipStreamSynchronize(s.st);  // (a copy that did start must not land in a slot handed to someone else)
    evs.put(s.dev, m.slot.ev);
    ring.release(m.slot);
    return fail(SPRING_REORDER_E_HIP, "device to host copy of an output stream failed");
  }
  m.seq = seq;
  cw.push(m.slot.ev, seq);
  m.slot.ev = nullptr;  // (the waiter's now)
  w.inflight.fetch_add(1, std::memory_order_relaxed);
  w.push(std::move(m));
  so += len;
  *sent = true;
  return 0;
}
// every file of `files` written by `nw` writer threads
int write_out_files(const std::vector<OutFile> &files, int nw, bool dbg) {
  if (!crc_ready) crc_init();
  nw = std::max(1, std::min(nw, (int)files.size()));
  // files -> writers: largest first, each to the writer with the least bytes so far
  std::vector<size_t> idx(files.size()), load((size_t)nw, 0);
  for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return files[a].total() > files[b].total(); });
  std::vector<std::vector<size_t>> mine((size_t)nw);
  for (size_t i : idx) {
    const size_t w = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
    mine[w].push_back(i);
    load[w] += files[i].total() + 4096;
  }
  // as many slots as the writers may hold in flight (MAX_INFLIGHT each) plus one chunk for the copies on their way
  Ring ring((nw * 3 + SLOTS_PER_CHUNK - 1) / SLOTS_PER_CHUNK + 1);
  // one flag per copy, one waiting thread per stream the copies are issued on (a multi-GPU call: one per rank)
  size_t ncopies = 0;
  std::vector<hipStream_t> streams;
  std::vector<int> sdev;
  for (const OutFile &f : files)
    for (const OutSeg &g : f.segs) {
      ncopies += (g.n + OUT_SLOT - 1) / OUT_SLOT;
      if (std::find(streams.begin(), streams.end(), g.st) == streams.end()) { streams.push_back(g.st); sdev.push_back(g.dev); }
    }
  Done done;
  done.f.assign(ncopies + 1, 0);
  EventPool evs;  // (declared before the waiters: they hand their events back to it until they are joined)
  std::vector<std::unique_ptr<CopyWaiter>> CW;
  for (size_t i = 0; i < std::max<size_t>(streams.size(), 1); i++) CW.emplace_back(new CopyWaiter(&done, &evs, sdev.empty() ? 0 : sdev[i]));
  std::vector<std::unique_ptr<Writer>> W;
  for (int i = 0; i < nw; i++) W.emplace_back(new Writer(&ring, &done));
  try {
    for (auto &c : CW) c->start();
    for (auto &w : W) w->start();
  } catch (const std::system_error &) {
    return fail(SPRING_REORDER_E_IO, "cannot start the writer threads");
  }
  size_t seq = 0;
  // The next slot goes to the writer with the MOST BYTES STILL TO COME among those with fewer than three slots in flight:
  // the copies of all writers share one link (50 GB/s), and handed out in turn every writer gets the same share -- the
  // one with the largest file (temp.dna.singleton, 0.5 GB of 5.45 at 100 M reads) is then fed at a twenty-fourth of it
  // until the small files are done, and the call ends when it does.  Longest first, all writers finish together.
  struct Cur { size_t fi = 0, si = 0, so = 0; bool open = false; size_t left = 0; };
  std::vector<Cur> cur((size_t)nw);
  for (int w = 0; w < nw; w++) cur[(size_t)w].left = load[(size_t)w];
  int ret = 0;
  SubmitClock clk;
  double t_pace = 0;
  const auto t_begin = std::chrono::steady_clock::now();
  constexpr int MAX_INFLIGHT = 3;  // (the ring above is sized for nw x 3)
  for (;;) {
    int w = -1;
    bool any = false;
    for (int k = 0; k < nw; k++) {
      const Cur &c = cur[(size_t)k];
      if (c.fi >= mine[(size_t)k].size()) continue;
      any = true;
      if (W[(size_t)k]->inflight.load(std::memory_order_acquire) >= MAX_INFLIGHT) continue;
      if (w < 0 || c.left > cur[(size_t)w].left) w = k;
    }
    if (!any) break;
    if (w < 0) {  // (every writer has its three slots)
      const auto tp0 = std::chrono::steady_clock::now();
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      t_pace += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count();
      continue;
    }
    Cur &c = cur[(size_t)w];
    const OutFile &f = files[mine[(size_t)w][c.fi]];
    if (!c.open) {
      Msg m;
      m.kind = f.gz ? Msg::OPEN_GZ : Msg::OPEN_RAW;
      m.path = f.path;
      W[(size_t)w]->push(std::move(m));
      if (!f.bytes.empty()) {
        Msg b;
        b.kind = Msg::BYTES;
        b.bytes = f.bytes;
        W[(size_t)w]->push(std::move(b));
      }
      c.open = true; c.si = 0; c.so = 0;
    }
    const size_t so0 = c.so, si0 = c.si;
    {
      size_t si2 = c.si, so2 = c.so;
      while (si2 < f.segs.size() && so2 >= f.segs[si2].n) { si2++; so2 = 0; }
      const size_t wi = si2 < f.segs.size() ? (size_t)(std::find(streams.begin(), streams.end(), f.segs[si2].st) - streams.begin()) : 0;
      bool sent_one = false;
      if ((ret = w_next_slot(*W[(size_t)w], ring, *CW[wi], evs, seq, &sent_one, f, c.si, c.so, clk))) break;
      if (sent_one) seq++;
    }
    const size_t sent = c.si == si0 ? c.so - so0 : c.so;  // (w_next_slot skips exhausted segments first)
    c.left -= std::min(c.left, sent);
    while (c.si < f.segs.size() && c.so >= f.segs[c.si].n) { c.si++; c.so = 0; }
    if (c.si >= f.segs.size()) {
      Msg m;
      m.kind = Msg::CLOSE;
      W[(size_t)w]->push(std::move(m));
      c.fi++; c.open = false;
      c.left -= std::min<size_t>(c.left, 4096);
    }
  }
  const auto t_sub = std::chrono::steady_clock::now();
  if (ret) {  // copies that were issued still land (their flags are set); the writers drain their queues
    for (auto &c : CW) c->finish();
  }
  for (auto &w : W) w->finish();
  for (auto &c : CW) c->finish();
  if (dbg) {
    double tw = 0, tp = 0, ti = 0, mx = 0, mxp = 0;
    size_t ns = 0;
    for (auto &w : W) { tw += w->t_wait; tp += w->t_put; ti += w->t_idle; ns += w->nslots; mx = std::max(mx, w->t_wait + w->t_put); mxp = std::max(mxp, w->t_put); }
    fprintf(stderr, "[files] %d writers, %zu slots: submitted after %.1f ms (waiting for a free writer %.1f ms), joined %.1f ms later; per writer on average "
            "%.1f ms waiting for copies, %.1f ms writing, %.1f ms without a message; busiest writer %.1f ms (most time writing: %.1f ms); the submitting thread: %.1f ms for a free slot, "
            "%.1f ms in hipMemcpyAsync, %.1f ms in event calls\n", nw, ns,
            std::chrono::duration<double, std::milli>(t_sub - t_begin).count(), 1e3 * t_pace,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sub).count(), 1e3 * tw / nw, 1e3 * tp / nw, 1e3 * ti / nw, 1e3 * mx, 1e3 * mxp, 1e3 * clk.t_ring, 1e3 * clk.t_copy, 1e3 * clk.t_event);
  }
  if (ret) return ret;
  for (auto &w : W) if (w->rc) return fail(w->rc, "%s", w->err.c_str());
  return 0;
}

// in-process all-gather between the rank threads of one pool (host transport: repeated devices, or asked for)
struct HostGather {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<uint8_t> buf;
  uint32_t world = 1, arrived = 0, gen = 0;
  bool failed = false;
  void barrier(std::unique_lock<std::mutex> &lk) {
    const uint32_t g = gen;
    if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g || failed; });
  }
  static int fn(void *host_buf, size_t off, size_t bytes, size_t total, void *user) {
    HostGather *h = (HostGather *)user;
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->failed) return 1;
    if (h->buf.size() < total) h->buf.resize(total);
    memcpy(h->buf.data() + off, (const uint8_t *)host_buf + off, bytes);
    h->barrier(lk);  // every slice is in
    if (h->failed) return 1;
    memcpy(host_buf, h->buf.data(), total);
    h->barrier(lk);  // everybody has read: the buffer may be overwritten by the next round
    return h->failed ? 1 : 0;
  }
  void abort_all() {
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
  }
};

}  // namespace

extern "C" int spring_reorder_run(const char *temp_dir, uint32_t max_readlen, int32_t num_thr, int32_t paired_end,
                                  uint32_t n0, uint32_t n1, const spring_reorder_opts *opts) {
  const bool dbg = opts && opts->debug != 0;  // phase timings on stderr, no effect on results
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
  if (o.num_devices < 0 || o.num_devices > 8) return fail(SPRING_REORDER_E_ARG, "num_devices must be 0..8");
  const int world = o.num_devices >= 2 ? o.num_devices : 1;
  const std::string base(temp_dir);
  const uint64_t ntot = (uint64_t)n0 + (paired_end ? n1 : 0);                              // reorder.h:761
  if (ntot > 4294967290ull) return fail(SPRING_REORDER_E_ARG, "too many reads");           // params.h:24
  const uint32_t n = (uint32_t)ntot;
  try {
    // ---- input (reorder.h:738-739)
    FilePair in;
    in.name[0] = base + "/input_clean_1.dna";
    in.name[1] = base + "/input_clean_2.dna";
    int r;
    for (int k = 0; k < (paired_end ? 2 : 1); k++) {
      if ((r = file_size(in.name[k], &in.sz[k]))) return r;
      in.fd[k] = open(in.name[k].c_str(), O_RDONLY);
      if (in.fd[k] < 0) return fail(SPRING_REORDER_E_IO, "cannot open %s: %s", in.name[k].c_str(), strerror(errno));
    }
    sr::DnaSource src;
    src.nbytes = in.sz[0] + in.sz[1];
    src.fill = &FilePair::fill;
    src.self = &in;

    // ---- the stage: one context per device; with several, one pool (reads on every device, chains sharded)
    std::vector<CtxGuard> g((size_t)world);
    std::vector<int> devs((size_t)world, o.device);
    bool repeated = false;
    for (int k = 0; k < world && world > 1; k++) {
      devs[(size_t)k] = o.devices[k];
      for (int j = 0; j < k; j++) repeated = repeated || o.devices[j] == o.devices[k];
    }
    for (int k = 0; k < world; k++) {
      spring_reorder_opts ok = o;
      ok.device = devs[(size_t)k];
      if ((r = spring_reorder_create(&g[(size_t)k].c, &ok))) return r;
    }
    if (world == 1) {
      if ((r = sr::load_dna_source(g[0].c, src, n, max_readlen))) return r;
      lap("read + H2D + unpack");
      if ((r = spring_reorder_build_dict(g[0].c))) return r;
      if ((r = spring_reorder_run_chains(g[0].c))) return r;
      if ((r = spring_reorder_finalize(g[0].c))) return r;
    } else {
      const bool host_transport = repeated || o.mg_host_transport != 0;
      HostGather hg;
      hg.world = (uint32_t)world;
      uint8_t id[SPRING_RCCL_ID_BYTES];
      if (!host_transport && (r = spring_mg_rccl_unique_id(id))) return r;
      std::vector<int> rcs((size_t)world, 0);
      std::vector<std::string> errs((size_t)world);
      // phase 1: load + dictionaries on every device; phase 2 (needs the default chain count of rank 0): the pool
      std::vector<uint32_t> autok((size_t)world, 0);
      auto phase1 = [&](int k) {
        int e = sr::load_dna_source(g[(size_t)k].c, src, n, max_readlen);
        if (!e) e = spring_reorder_build_dict(g[(size_t)k].c);
        if (!e) e = spring_reorder_auto_chains(g[(size_t)k].c, &autok[(size_t)k], nullptr);
        if (e) { rcs[(size_t)k] = e; errs[(size_t)k] = spring_reorder_last_error(); }
      };
      // (a thread that cannot be started must not unwind past its running siblings: std::terminate)
      // ... nor may some ranks enter a collective that the missing one never joins: the threads wait at a gate until all exist)
      auto run_ranks = [&](auto &&body) -> bool {
        std::vector<std::thread> th;
        std::atomic<int> gate{0};  // 0 wait, 1 go, -1 give up
        bool ok = true;
        for (int k = 1; k < world && ok; k++) {
          try {
            th.emplace_back([&gate, &body, k] {
              int g;
              while ((g = gate.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
              if (g > 0) body(k);
            });
          } catch (const std::system_error &) { ok = false; }
        }
        gate.store(ok ? 1 : -1, std::memory_order_release);
        if (ok) body(0);
        for (auto &x : th) x.join();
        return ok;
      };
      if (!run_ranks(phase1)) return fail(SPRING_REORDER_E_IO, "cannot start the rank threads");
      for (int k = 0; k < world; k++) if (rcs[(size_t)k]) return fail(rcs[(size_t)k], "%s", errs[(size_t)k].c_str());
      lap("read + H2D + dictionaries");
      uint32_t Ktot = o.num_chains ? o.num_chains : autok[0];
      Ktot = (Ktot + (uint32_t)world - 1) / (uint32_t)world * (uint32_t)world;
      std::vector<spring_mg_comm *> comms((size_t)world, nullptr);
      std::mutex comms_mu;
      auto phase2 = [&](int k) {
        spring_mg_comm *comm = nullptr;
        int e = host_transport ? spring_mg_comm_create_host(&comm, &HostGather::fn, &hg, (uint32_t)k, (uint32_t)world)
                               : spring_mg_comm_create_rccl(&comm, devs[(size_t)k], id, (uint32_t)k, (uint32_t)world);
        { std::lock_guard<std::mutex> lk(comms_mu); comms[(size_t)k] = comm; }
        if (!e) e = spring_reorder_mg_run(g[(size_t)k].c, comm, Ktot);
        if (!e) e = spring_reorder_finalize(g[(size_t)k].c);
        if (e) {  // the peers must not wait for this rank for ever: host transport and RCCL alike
          rcs[(size_t)k] = e; errs[(size_t)k] = spring_reorder_last_error(); hg.abort_all();
          std::lock_guard<std::mutex> lk(comms_mu);
          for (int j = 0; j < world; j++) if (j != k) sr::mg_comm_abort(comms[(size_t)j]);
        }
        { std::lock_guard<std::mutex> lk(comms_mu); comms[(size_t)k] = nullptr; }
        spring_mg_comm_destroy(comm);
      };
      if (!run_ranks(phase2)) return fail(SPRING_REORDER_E_IO, "cannot start the rank threads");
      for (int k = 0; k < world; k++) if (rcs[(size_t)k]) return fail(rcs[(size_t)k], "%s", errs[(size_t)k].c_str());
    }
    lap("dict + chains + final");

    // ---- output: tid t of the job = every rank's tid-t segment, ranks ascending (chain c -> tid c % num_thr)
    std::vector<sr::ReorderView> v((size_t)world);
    uint64_t unmatched = 0, nsing_total = 0;
    for (int k = 0; k < world; k++) {
      if ((r = sr::reorder_view_any(g[(size_t)k].c, &v[(size_t)k]))) return r;
      spring_reorder_stats st;
      if ((r = spring_reorder_get_stats(g[(size_t)k].c, &st))) return r;
      unmatched += st.unmatched;
      nsing_total += v[(size_t)k].nsing;
    }
    // temp.dna.<tid> / temp.dna.singleton are built on the device (reverse complement + repack), rank by rank; all
    // of them first, so that the copies of every file can be in flight together (40 bytes per read of device memory)
    struct EmitBufs {
      std::vector<std::pair<spring_reorder_ctx *, uint8_t *>> v;
      ~EmitBufs() { for (auto &e : v) sr::emit_dna_free(e.first, e.second); }  // (waits for the stream: the copies have left the buffer)
    } emitted;
    std::vector<OutFile> files;
    auto stream_of = [&](int t, const char *name, bool gz, size_t elem, auto ptr_of) {
      OutFile f;
      f.path = base + "/" + name + "." + std::to_string(t);
      f.gz = gz;
      // (chain ids ascending inside a tid: every rank's chains of the first chain group, then every rank's of the second --
      // one part per rank unless the pool ran two groups, tid_mid)
      for (int part = 0; part < 2; part++)
        for (int k = 0; k < world; k++) {
          const sr::ReorderView &vk = v[(size_t)k];
          const uint64_t a = part ? vk.tid_mid[t] : vk.tid_off[t], c = (part ? vk.tid_off[t + 1] : vk.tid_mid[t]) - a;
          if (c) f.segs.push_back({vk.dev, vk.st, (const uint8_t *)ptr_of(vk) + a * elem, (size_t)(c * elem)});
        }
      files.push_back(std::move(f));
    };
    for (int t = 0; t < num_thr; t++) {  // all six files must exist for every tid (encoder.h:147-175)
      stream_of(t, "read_order.bin", false, 4, [](const sr::ReorderView &x) { return (const void *)x.f_order; });
      stream_of(t, "read_rev.txt", true, 1, [](const sr::ReorderView &x) { return (const void *)x.f_rc; });
      stream_of(t, "tempflag.txt", true, 1, [](const sr::ReorderView &x) { return (const void *)x.f_flag; });
      stream_of(t, "temppos.txt", true, 8, [](const sr::ReorderView &x) { return (const void *)x.f_pos; });
      stream_of(t, "read_lengths.bin", true, 2, [](const sr::ReorderView &x) { return (const void *)x.f_len; });
      OutFile f;
      f.path = base + "/temp.dna." + std::to_string(t);
      std::vector<uint8_t *> dk((size_t)world, nullptr);
      std::vector<size_t> nbk((size_t)world, 0), midk((size_t)world, 0);
      for (int k = 0; k < world; k++) {
        if ((r = sr::emit_dna_device(g[(size_t)k].c, t, &dk[(size_t)k], &nbk[(size_t)k], 0, ~0ull, &midk[(size_t)k]))) return r;
        if (dk[(size_t)k]) emitted.v.push_back({g[(size_t)k].c, dk[(size_t)k]});
      }
      for (int part = 0; part < 2; part++)
        for (int k = 0; k < world; k++) {
          const size_t a = part ? midk[(size_t)k] : 0, c = (part ? nbk[(size_t)k] : midk[(size_t)k]) - a;
          if (c) f.segs.push_back({v[(size_t)k].dev, v[(size_t)k].st, dk[(size_t)k] + a, c});
        }
      files.push_back(std::move(f));
    }
    {  // reorder.h:699-728; the singleton streams are in tid order too (rank by rank inside a tid)
      OutFile fd, fo, fc;
      fd.path = base + "/temp.dna.singleton";
      fo.path = base + "/read_order.bin.singleton";
      fc.path = base + "/temp.dna.singleton.count";
      for (int t = 0; t < num_thr; t++)
        for (int part = 0; part < 2; part++)
        for (int k = 0; k < world; k++) {
          const sr::ReorderView &vk = v[(size_t)k];
          const uint64_t a = part ? vk.tid_mid_s[t] : vk.tid_off_s[t], c = (part ? vk.tid_off_s[t + 1] : vk.tid_mid_s[t]) - a;
          if (!c) continue;
          uint8_t *d = nullptr;
          size_t nb = 0;
          if ((r = sr::emit_dna_device(g[(size_t)k].c, -1, &d, &nb, a, c))) return r;
          if (d) emitted.v.push_back({g[(size_t)k].c, d});
          if (nb) fd.segs.push_back({vk.dev, vk.st, d, nb});
          fo.segs.push_back({vk.dev, vk.st, (const uint8_t *)(vk.f_order_s + a), (size_t)c * 4});
        }
      const uint32_t numreads_s = (uint32_t)nsing_total;
      fc.bytes.assign((const uint8_t *)&numreads_s, (const uint8_t *)&numreads_s + 4);
      files.push_back(std::move(fd));
      files.push_back(std::move(fo));
      files.push_back(std::move(fc));
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const int nw = o.out_writers > 0 ? o.out_writers : (int)std::max(4u, std::min(24u, hw ? hw / 4 : 8u));
    lap("emit temp.dna");
    if ((r = write_out_files(files, nw, dbg))) return r;
    lap("D2H + write files");
    // the stage consumes its inputs (reorder.h:232,241) -- once its outputs exist: a failed call leaves them in place
    // Unlinking a file whose pages sit in the page cache frees them page by page (0.3 s for the 4 GB of 100 M reads on
    // the GPU box, tools/pagecache_write_bench.c) -- at the LAST reference to the inode.  The names go now; the
    // descriptors this call still holds are closed by a detached thread, which is where the pages are freed.
    remove((base + "/input_clean_1.dna").c_str());
    if (paired_end) remove((base + "/input_clean_2.dna").c_str());
    {
      const int fd0 = in.fd[0], fd1 = in.fd[1];
      in.fd[0] = in.fd[1] = -1;
      try {
        std::thread([fd0, fd1] { if (fd0 >= 0) close(fd0); if (fd1 >= 0) close(fd1); }).detach();
      } catch (const std::system_error &) {
        if (fd0 >= 0) close(fd0);
        if (fd1 >= 0) close(fd1);
      }
    }
    lap("unlink inputs");
    printf("Reordering done, %llu were unmatched\n", (unsigned long long)unmatched);  // reorder.h:633-635
    return 0;
  } catch (const std::bad_alloc &) {
    return fail(SPRING_REORDER_E_IO, "out of host memory in the reorder stage");
  } catch (const std::system_error &e) {
    return fail(SPRING_REORDER_E_IO, "reorder stage: %s", e.what());
  }
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
  if (o.num_devices >= 2)
    return fail(SPRING_REORDER_E_ARG, "spring_reorder_encode_run runs on one device (opts.device): a device list is taken by spring_reorder_run only");
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
