// spring_amd/csrc/reorder_pipeline.cpp
//
// Host side of the reorder stage: owns the device memory, sequences the HIP
// kernels of reorder_kernels.hip on one stream and implements the in-memory
// half of the C ABI declared in include/spring_reorder.h.  Mirrors
// reorder_main<N>() (reference src/reorder.h:732-786): load -> dictionaries ->
// chains -> streams.  There is no CPU fallback anywhere in this file: every
// stage runs on the GPU or the call fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <system_error>
#include <thread>
#include <memory>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include <zlib.h>

#include "reorder_device.h"
#include "reorder_internal.h"
#include "spring_reorder.h"
#include "synth_common.h"

namespace sr {
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace sr

using namespace sr;

#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess)                                                                      \
      return fail(SPRING_REORDER_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ---------------------------------------------------------------- device pool
// hipMalloc/hipFree of the ~30 GB a 100 M-read run needs cost ~1 s per run, as much as
// the stage itself.  Blocks released by a context are cached per device and handed to
// the next context (sizes rounded to 2 MiB so repeated runs hit exactly).
// spring_reorder_trim_pool() gives the memory back to the driver.
namespace {
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, void *> free_blocks;
  std::unordered_map<void *, size_t> sizes;
  size_t cached = 0;
};
DevPool g_pool[16];
constexpr size_t POOL_GRAN = 2u << 20;

hipError_t pool_alloc(int dev, size_t bytes, void **out, size_t *actual) {
  const size_t want = (bytes + POOL_GRAN - 1) / POOL_GRAN * POOL_GRAN;
  DevPool &p = g_pool[dev & 15];
  {
    std::lock_guard<std::mutex> lk(p.mu);
    auto it = p.free_blocks.lower_bound(want);
    if (it != p.free_blocks.end() && it->first <= want + want / 4 + POOL_GRAN) {
      *out = it->second; *actual = it->first;
      p.cached -= it->first;
      p.free_blocks.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, want);
  if (e != hipSuccess) {  // give cached blocks back and retry once
    std::lock_guard<std::mutex> lk(p.mu);
    for (auto &kv : p.free_blocks) { (void)hipFree(kv.second); p.sizes.erase(kv.second); }
    p.free_blocks.clear(); p.cached = 0;
    (void)hipGetLastError();
    e = hipMalloc(out, want);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(p.mu);
    p.sizes[*out] = want;
    *actual = want;
  }
  return e;
}
void pool_free(int dev, void *ptr) {
  DevPool &p = g_pool[dev & 15];
  std::lock_guard<std::mutex> lk(p.mu);
  auto it = p.sizes.find(ptr);
  if (it == p.sizes.end()) { (void)hipFree(ptr); return; }
  p.free_blocks.emplace(it->second, ptr);
  p.cached += it->second;
}
}  // namespace

enum { ST_CREATED = 0, ST_LOADED = 1, ST_DICT = 2, ST_CHAINS = 3, ST_FINAL = 4 };

struct DictDev {
  int start = 0, end = 0;
  uint32_t numkeys = 0, numreads = 0;
  ulonglong2 *urec = nullptr;
  uint32_t *ids = nullptr;
  uint32_t *deep = nullptr, *d_ndeep = nullptr;  // bins with >= DEEP_BIN reads (k_trim_bins)
  uint32_t ndeep = 0;
  uint32_t big_reads = 0;   // reads in bins of >= BIG_BIN entries (d_ndeep[1]): the pools whose long searches go to k_long
  uint32_t mid_reads = 0;   // reads in bins of >= MID_BIN entries (d_ndeep[2]): the heavy-tail rule
};

struct spring_reorder_ctx {
  spring_reorder_opts o;
  int dev = 0;
  hipStream_t st = nullptr;
  int stage = ST_CREATED;
  std::vector<void *> allocs;
  uint64_t dev_bytes = 0, peak_bytes = 0;
  // input
  uint8_t *d_dna = nullptr;  // record stream (owned unless borrowed)
  bool dna_borrowed = false;
  size_t dna_bytes = 0;
  uint64_t *d_off = nullptr;
  uint32_t n = 0;
  int L = 0, W = 0, S = 0, Lpad = 0;
  bool uniform = true;
  uint64_t *d_reads = nullptr;
  uint16_t *d_lens = nullptr;
  DictDev dict[2];
  uint4 *fpt = nullptr;   // one bucket table for both dictionaries (reorder_kernels.hip, tab_find): nb buckets of 32 bytes
  int bshift = 63;        // plain home bucket = hash >> bshift (nb = 2^(64 - bshift) buckets)
  int minz = 0, lshift = 31;  // minimizer-addressed table (TabView)
  uint32_t marked_lines = 0;  // ... lines of it whose keys were sent to the redirect address
  bool user_plan0 = false;    // opts.plan0 was given (fill_params)
  DevParams P;
  uint32_t K = 0;
  uint64_t nrec = 0, nsing = 0, cap = 0;
  bool mg = false;
  uint32_t *cnt_buf[2] = {nullptr, nullptr};  // needy_cnt double buffer (reorder_device.h)
  uint64_t round_no = 0;
  // two-group schedule (DevParams::phases = 2): the second group's stream, the stream of the host's look at the running
  // chains, the second group's view of the pool and its reservation words, the winners of either group's last mark step
  hipStream_t st2 = nullptr, st3 = nullptr;
  uint64_t *taken2 = nullptr;
  uint32_t *resv2 = nullptr, *won = nullptr;
  uint32_t Kh = 0, nmid = 0;
  // geometry of the chain groups for this context (DevParams: g0 / Kg / c0 / gg0 / gKg): one group = every local chain; two
  // groups = a slice of each (one GPU: the slices are the groups)
  struct GroupGeom { uint32_t g0 = 0, Kg = 0, c0 = 0, gg0 = 0, gKg = 0; } grp[2];
  struct LongBufs { uint32_t *longq = nullptr, *lctl = nullptr, *lparts = nullptr; LongHead *lhead = nullptr; uint2 *lbin = nullptr; uint16_t *lbcode = nullptr; } lb2;  // group 1's (DevParams::longq ...)
  bool in_source_fallback = false;  // load_dna <-> load_dna_source recursion guard
  // FASTQ front end (f1): reads with N, per input file
  uint8_t *d_N[2] = {nullptr, nullptr};
  uint32_t *d_orderN[2] = {nullptr, nullptr};
  uint64_t *d_offN[2] = {nullptr, nullptr};   // byte offset of every N record inside d_N
  uint64_t N_bytes[2] = {0, 0};
  spring_fastq_info fq;
  double fq_ms = 0;
  std::vector<uint64_t> tid_off, tid_off_s;
  std::vector<uint64_t> tid_mid, tid_mid_s;  // inside tid t: where the records of the second group's chains begin (= tid_off[t + 1] with one group)
  spring_reorder_stats stats;
  hipEvent_t ev[8];
  bool ev_ok = false;

  int dmalloc(void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    size_t actual = 0;
    hipError_t e = pool_alloc(dev, bytes, p, &actual);
    if (e != hipSuccess) return fail(SPRING_REORDER_E_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    allocs.push_back(*p);
    dev_bytes += actual;
    peak_bytes = std::max(peak_bytes, dev_bytes);
    return 0;
  }
  void dfree(void *p) {
    if (!p) return;
    auto it = std::find(allocs.begin(), allocs.end(), p);
    if (it != allocs.end()) allocs.erase(it);
    {
      std::lock_guard<std::mutex> lk(g_pool[dev & 15].mu);
      auto sz = g_pool[dev & 15].sizes.find(p);
      if (sz != g_pool[dev & 15].sizes.end()) dev_bytes -= std::min<uint64_t>(dev_bytes, sz->second);
    }
    (void)hipStreamSynchronize(st);  // the block may be handed to another stream next
    pool_free(dev, p);
  }
};

namespace sr {
int reorder_view(spring_reorder_ctx *ctx, ReorderView *v) {
  if (ctx && ctx->mg) return fail(SPRING_REORDER_E_STATE, "single-pool multi-GPU contexts hold partial streams");
  return reorder_view_any(ctx, v);
}
int reorder_view_any(spring_reorder_ctx *ctx, ReorderView *v) {
  if (!ctx || !v) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  if (ctx->stage != ST_FINAL) return fail(SPRING_REORDER_E_STATE, "reorder context is not finalized");
  v->dev = ctx->dev; v->st = ctx->st; v->n = ctx->n; v->L = ctx->L; v->W = ctx->W; v->S = ctx->S;
  v->reads = ctx->d_reads; v->lens = ctx->d_lens; v->nrec = ctx->nrec; v->nsing = ctx->nsing;
  v->f_order = ctx->P.f_order; v->f_order_s = ctx->P.f_order_s; v->f_rc = ctx->P.f_rc; v->f_flag = ctx->P.f_flag;
  v->f_pos = ctx->P.f_pos; v->f_len = ctx->P.f_len; v->tid_off = ctx->tid_off.data(); v->num_thr = ctx->o.num_thr;
  v->tid_off_s = ctx->tid_off_s.data();
  v->tid_mid = ctx->tid_mid.data(); v->tid_mid_s = ctx->tid_mid_s.data();
  for (int j = 0; j < 2; j++) {  // reads with N kept by the FASTQ front end (none when the reads came as .dna records)
    v->N_dna[j] = ctx->d_N[j]; v->N_off[j] = ctx->d_offN[j]; v->N_order[j] = ctx->d_orderN[j];
    v->N_count[j] = ctx->fq.num_reads_N[j]; v->N_bytes[j] = ctx->N_bytes[j];
  }
  v->fq_num_reads_0 = ctx->fq.num_reads[0];
  return 0;
}
hipError_t dev_alloc(int dev, size_t bytes, void **out) {
  size_t actual = 0;
  return pool_alloc(dev, bytes ? bytes : 16, out, &actual);
}
void dev_free(int dev, void *p) { if (p) pool_free(dev, p); }
}  // namespace sr

// ---------------------------------------------------------------- RCCL, loaded on first use
// The four entry points the pool needs, resolved from librccl.so.1 at run time: a process that already holds RCCL
// (torch.distributed's nccl backend) shares that copy, a single-GPU user never loads it.
namespace {
struct RcclId128 { char b[128]; };  // ncclUniqueId, passed by value
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, RcclId128, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*CommAbort)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return 0;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(SPRING_REORDER_E_HIP, "cannot load librccl.so.1: %s", dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");  // (optional: only used to unblock the peers of a failed rank)
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather)
    return fail(SPRING_REORDER_E_HIP, "librccl lacks an expected entry point");
  g_rccl.h = h;
  return 0;
}
const char *rccl_err(int e) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "rccl error"; }
constexpr int RCCL_UINT64 = 5;  // ncclUint64 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5)
}  // namespace
// The prototypes above are written by hand so that the library builds and runs where RCCL's header is absent and never links
// against librccl.  Where the header IS present at build time they are checked against it: same argument lists (an enum or
// an opaque handle where this file says int / void *, which the x86-64 ABI passes alike), same ncclUniqueId size, same
// datatype code.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#include <type_traits>
static_assert(sizeof(ncclUniqueId) == sizeof(RcclId128) && NCCL_UNIQUE_ID_BYTES == 128, "ncclUniqueId is not 128 bytes");
static_assert((int)ncclUint64 == RCCL_UINT64, "ncclUint64 has another code");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclComm_t) == sizeof(void *),
              "enum / handle sizes differ from the hand-written prototypes");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId *)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclCommAbort), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommAbort");
static_assert(std::is_same<decltype(&ncclAllGather),
                           ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value, "ncclAllGather");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char *(*)(ncclResult_t)>::value, "ncclGetErrorString");
#endif
#endif
namespace {
}  // namespace

// exchange transport of mg_run: an RCCL communicator or a caller-supplied host all-gather
struct spring_mg_comm {
  uint32_t rank = 0, world = 1;
  int dev = 0;
  void *rccl = nullptr;
  std::mutex mu;          // rccl handle: destroy / abort
  bool aborted = false;
  spring_mg_allgather_fn host_fn = nullptr;
  void *host_user = nullptr;
};

#define DMALLOC(ptr, bytes)                                   \
  do {                                                        \
    int r_ = ctx->dmalloc((void **)&(ptr), (bytes));          \
    if (r_) return r_;                                        \
  } while (0)

extern "C" {

void spring_reorder_default_opts(spring_reorder_opts *o) {
  memset(o, 0, sizeof(*o));
  o->device = -1;
  o->num_chains = 0;
  o->num_thr = 1;
}

const char *spring_reorder_last_error(void) { return g_err.c_str(); }

extern "C++" {
namespace {
struct PinCache {
  std::mutex mu;
  std::vector<void *> free_chunks;
};
PinCache g_pin;
}  // namespace
namespace sr {
void *pinned_get() {
  {
    std::lock_guard<std::mutex> lk(g_pin.mu);
    if (!g_pin.free_chunks.empty()) {
      void *p = g_pin.free_chunks.back();
      g_pin.free_chunks.pop_back();
      return p;
    }
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, PIN_CHUNK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void pinned_put(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pin.mu);
  g_pin.free_chunks.push_back(p);
}
}  // namespace sr
}  // extern "C++"

void spring_reorder_trim_pool(void) {
  {
    std::lock_guard<std::mutex> lk(g_pin.mu);
    for (void *p : g_pin.free_chunks) (void)hipHostFree(p);
    g_pin.free_chunks.clear();
  }
  for (int d = 0; d < 16; d++) {
    DevPool &p = g_pool[d];
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.free_blocks.empty()) continue;
    (void)hipSetDevice(d);
    for (auto &kv : p.free_blocks) { (void)hipFree(kv.second); p.sizes.erase(kv.second); }
    p.free_blocks.clear();
    p.cached = 0;
  }
}

int spring_reorder_create(spring_reorder_ctx **out, const spring_reorder_opts *opts) {
  if (!out) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  spring_reorder_opts o;
  if (opts) o = *opts; else spring_reorder_default_opts(&o);
  if (o.num_thr <= 0) return fail(SPRING_REORDER_E_ARG, "num_thr must be >= 1");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(SPRING_REORDER_E_HIP, "no HIP device");
  int dev = o.device;
  if (dev < 0) HIPCHK(hipGetDevice(&dev));
  if (dev >= ndev) return fail(SPRING_REORDER_E_ARG, "device %d out of range (%d devices)", dev, ndev);
  HIPCHK(hipSetDevice(dev));
  spring_reorder_ctx *ctx = new spring_reorder_ctx();
  ctx->o = o;
  ctx->dev = dev;
  memset(&ctx->P, 0, sizeof(ctx->P));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  HIPCHK(hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking));
  for (auto &e : ctx->ev) HIPCHK(hipEventCreate(&e));
  ctx->ev_ok = true;
  *out = ctx;
  return 0;
}

void spring_reorder_destroy(spring_reorder_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->dev);
  if (ctx->st) (void)hipStreamSynchronize(ctx->st);
  if (ctx->st2) (void)hipStreamSynchronize(ctx->st2);  // (the second chain group's stream and the host's look at the running
  if (ctx->st3) (void)hipStreamSynchronize(ctx->st3);  //  chains: idle unless run_chains failed half way)
  for (void *p : ctx->allocs) pool_free(ctx->dev, p);
  if (ctx->ev_ok) for (auto &e : ctx->ev) (void)hipEventDestroy(e);
  if (ctx->st) (void)hipStreamDestroy(ctx->st);
  if (ctx->st2) (void)hipStreamDestroy(ctx->st2);
  if (ctx->st3) (void)hipStreamDestroy(ctx->st3);
  delete ctx;
}

static int setup_geometry(spring_reorder_ctx *ctx, uint32_t n, uint32_t max_readlen) {
  if (max_readlen == 0 || max_readlen > (uint32_t)MAX_READ_LEN)
    return fail(SPRING_REORDER_E_ARG, "Wrong bitset size. (max_readlen=%u, supported 1..511)", max_readlen);
  ctx->n = n;
  ctx->L = (int)max_readlen;
  ctx->W = (2 * ctx->L - 1) / 64 + 1;  // call_template_functions.cpp:10
  int S = 1;
  while (S < ctx->W) S <<= 1;
  ctx->S = S;
  ctx->Lpad = (ctx->L + 63) / 64 * 64;
  // dictionary windows, reorder.h:751-759
  const int L = ctx->L;
  ctx->dict[0].start = L > 100 ? L / 2 - 32 : L / 2 - L * 32 / 100;
  ctx->dict[0].end = L / 2 - 1;
  ctx->dict[1].start = L / 2;
  ctx->dict[1].end = L > 100 ? L / 2 - 1 + 32 : L / 2 - 1 + L * 32 / 100;
  return 0;
}

static int unpack_on_device(spring_reorder_ctx *ctx, uint32_t *d_bad_len = nullptr) {
  if (!ctx->d_reads) DMALLOC(ctx->d_reads, (size_t)std::max<uint32_t>(ctx->n, 1) * ctx->S * sizeof(uint64_t));
  if (!ctx->d_lens) DMALLOC(ctx->d_lens, (size_t)std::max<uint32_t>(ctx->n, 1) * sizeof(uint16_t));
  HIPCHK(hipEventRecord(ctx->ev[0], ctx->st));
  const uint32_t rec = 2u + ((uint32_t)ctx->L + 3u) / 4u;
  launch_unpack(ctx->st, ctx->d_dna, ctx->d_off, ctx->n, ctx->L, ctx->W, ctx->S, rec, ctx->d_reads, ctx->d_lens, d_bad_len);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->st));
  ctx->stage = ST_LOADED;
  return 0;
}

// Host -> device copy of `nbytes` produced piecewise by `fill` (thread-safe): up to 8 host threads, each with its
// own stream and two pinned chunks, fill one chunk while the other is on its way (pageable memory or a file never
// goes through the runtime's own staging: that path runs at ~8 GB/s on a link that does ~50).
static int upload_chunked(spring_reorder_ctx *ctx, uint8_t *d_dst, size_t nbytes,
                          int (*fill)(void *, size_t, void *, size_t), void *self) {
  if (!nbytes) return 0;
  const size_t nchunks = (nbytes + PIN_CHUNK - 1) / PIN_CHUNK;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int nthr = (int)std::min<size_t>(std::min<size_t>(8, hw), nchunks);
  std::atomic<size_t> next{0};
  std::atomic<int> rc{0};
  std::vector<std::string> errs((size_t)nthr);
  auto worker = [&](int ti) {
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    bool used[2] = {false, false};
    auto bail = [&](int code, const char *what) {
      int zero = 0;
      if (rc.compare_exchange_strong(zero, code)) errs[(size_t)ti] = what;
    };
    if (hipSetDevice(ctx->dev) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      bail(SPRING_REORDER_E_HIP, "upload: cannot create a copy stream");
    } else {
      for (int k = 0; k < 2 && !rc.load(); k++) {
        pin[k] = pinned_get();
        if (!pin[k] || hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) bail(SPRING_REORDER_E_HIP, "upload: cannot pin a staging chunk");
      }
      for (int k = 0; !rc.load(); k ^= 1) {
        const size_t i = next.fetch_add(1);
        if (i >= nchunks) break;
        const size_t off = i * PIN_CHUNK, len = std::min(PIN_CHUNK, nbytes - off);
        if (used[k] && hipEventSynchronize(ev[k]) != hipSuccess) { bail(SPRING_REORDER_E_HIP, "upload: event wait failed"); break; }
        if (fill(self, off, pin[k], len)) { bail(SPRING_REORDER_E_IO, g_err.c_str()); break; }
        if (hipMemcpyAsync(d_dst + off, pin[k], len, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipEventRecord(ev[k], st) != hipSuccess) { bail(SPRING_REORDER_E_HIP, "upload: hipMemcpyAsync failed"); break; }
        used[k] = true;
      }
      (void)hipStreamSynchronize(st);
    }
    for (int k = 0; k < 2; k++) { if (ev[k]) (void)hipEventDestroy(ev[k]); pinned_put(pin[k]); }
    if (st) (void)hipStreamDestroy(st);
  };
  std::vector<std::thread> th;
  try {
    for (int t = 1; t < nthr; t++) th.emplace_back(worker, t);
  } catch (const std::system_error &) {  // fewer helpers than hoped for: the calling thread does the rest
  }
  worker(0);
  for (auto &x : th) x.join();
  (void)hipGetLastError();
  if (rc.load()) {
    for (auto &e : errs) if (!e.empty()) return fail(rc.load(), "%s", e.c_str());
    return fail(rc.load(), "upload failed");
  }
  return 0;
}

// Device -> host copies of several arrays at once: the pieces (32 MiB chunks of every array) are dealt out to up to 8
// host threads, each with its own stream and two pinned chunks -- the device-to-pinned copy of one piece runs while
// the thread copies the previous piece out to the caller's (pageable) memory.
struct D2HJob { void *dst; const void *src; size_t nbytes; };
static int download_chunked(spring_reorder_ctx *ctx, const std::vector<D2HJob> &jobs) {
  struct Piece { uint8_t *dst; const uint8_t *src; size_t len; };
  std::vector<Piece> pieces;
  for (const D2HJob &j : jobs)
    for (size_t off = 0; off < j.nbytes; off += PIN_CHUNK)
      pieces.push_back({(uint8_t *)j.dst + off, (const uint8_t *)j.src + off, std::min(PIN_CHUNK, j.nbytes - off)});
  if (pieces.empty()) return 0;
  HIPCHK(hipStreamSynchronize(ctx->st));  // the arrays are complete
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int nthr = (int)std::min<size_t>(std::min<size_t>(8, hw), pieces.size());
  std::atomic<size_t> next{0};
  std::atomic<int> rc{0};
  auto worker = [&]() {
    void *pin[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    const Piece *pending[2] = {nullptr, nullptr};
    if (hipSetDevice(ctx->dev) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) rc = SPRING_REORDER_E_HIP;
    for (int k = 0; k < 2 && !rc.load(); k++) if (!(pin[k] = pinned_get())) rc = SPRING_REORDER_E_HIP;
    // two-deep software pipeline on one stream: issue copy k, then drain copy k-1 while k is in flight
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int k = 0; k < 2 && !rc.load(); k++) if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) rc = SPRING_REORDER_E_HIP;
    auto drain = [&](int k) {
      if (!pending[k]) return;
      if (hipEventSynchronize(ev[k]) != hipSuccess) rc = SPRING_REORDER_E_HIP;
      else memcpy(pending[k]->dst, pin[k], pending[k]->len);
      pending[k] = nullptr;
    };
    for (int k = 0; !rc.load(); k ^= 1) {
      const size_t i = next.fetch_add(1);
      if (i >= pieces.size()) break;
      drain(k);
      if (hipMemcpyAsync(pin[k], pieces[i].src, pieces[i].len, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipEventRecord(ev[k], st) != hipSuccess) { rc = SPRING_REORDER_E_HIP; break; }
      pending[k] = &pieces[i];
      drain(k ^ 1);
    }
    drain(0); drain(1);
    for (int k = 0; k < 2; k++) { if (ev[k]) (void)hipEventDestroy(ev[k]); pinned_put(pin[k]); }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  };
  std::vector<std::thread> th;
  try { for (int t = 1; t < nthr; t++) th.emplace_back(worker); } catch (const std::system_error &) {}
  worker();
  for (auto &x : th) x.join();
  (void)hipGetLastError();
  return rc.load() ? fail(SPRING_REORDER_E_HIP, "download: device to host copy failed") : 0;
}

static int scan_records(const uint8_t *dna, size_t nbytes, uint32_t n, int L, bool &uniform, std::vector<uint64_t> &off);

extern "C++" {
namespace sr {
int load_dna_source(spring_reorder_ctx *ctx, const DnaSource &src, uint32_t n, uint32_t max_readlen) {
  if (!ctx || !src.fill) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  if (ctx->stage != ST_CREATED) return fail(SPRING_REORDER_E_STATE, "load_dna: context already loaded");
  HIPCHK(hipSetDevice(ctx->dev));
  int r = setup_geometry(ctx, n, max_readlen);
  if (r) return r;
  const size_t rec = 2u + (max_readlen + 3u) / 4u, nbytes = src.nbytes;
  if (n && nbytes == (size_t)n * rec) {  // the size of a fixed-length stream: upload, unpack, verify on the device
    ctx->uniform = true;
    ctx->dna_bytes = nbytes;
    DMALLOC(ctx->d_dna, nbytes + 16);
    if ((r = upload_chunked(ctx, ctx->d_dna, nbytes, src.fill, src.self))) return r;
    uint32_t *d_bad = nullptr, bad = 0;
    DMALLOC(d_bad, 16);
    HIPCHK(hipMemsetAsync(d_bad, 0, 4, ctx->st));
    if ((r = unpack_on_device(ctx, d_bad))) return r;
    HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    ctx->dfree(d_bad);
    if (!bad) return 0;
    ctx->stage = ST_CREATED;  // same size, other lengths: walk the records after all
    ctx->dfree(ctx->d_dna); ctx->d_dna = nullptr;
  }
  // variable-length (or malformed) stream: the record starts are sequentially dependent -> a host image is walked
  if (src.image) {  // (the caller's own buffer: no second copy of a multi-GB stream)
    ctx->in_source_fallback = true;
    r = spring_reorder_load_dna(ctx, src.image, nbytes, n, max_readlen);
    ctx->in_source_fallback = false;
    return r;
  }
  std::unique_ptr<uint8_t[]> img;
  try { img.reset(new uint8_t[nbytes + 1]); } catch (const std::bad_alloc &) { return fail(SPRING_REORDER_E_IO, "out of host memory for a %zu-byte record stream", nbytes); }
  {
    const size_t CH = (size_t)64 << 20, nch = (nbytes + CH - 1) / CH;
    const int nthr = (int)std::min<size_t>(std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(nch, 1));
    std::atomic<size_t> next{0};
    std::atomic<int> rc{0};
    std::string err;
    std::mutex emu;
    auto worker = [&]() {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= nch || rc.load()) break;
        const size_t off = i * CH, len = std::min(CH, nbytes - off);
        if (src.fill(src.self, off, img.get() + off, len)) { std::lock_guard<std::mutex> lk(emu); rc = SPRING_REORDER_E_IO; err = g_err; }
      }
    };
    std::vector<std::thread> th;
    try { for (int t = 1; t < nthr; t++) th.emplace_back(worker); } catch (const std::system_error &) {}
    worker();
    for (auto &x : th) x.join();
    if (rc.load()) return fail(rc.load(), "%s", err.c_str());
  }
  ctx->in_source_fallback = true;
  r = spring_reorder_load_dna(ctx, img.get(), nbytes, n, max_readlen);
  ctx->in_source_fallback = false;
  return r;
}
}  // namespace sr
}  // extern "C++"

// walks the record stream once on the host: validates it and decides whether
// every read has len == max_readlen (then no offset array is needed).
static int scan_records(const uint8_t *dna, size_t nbytes, uint32_t n, int L, bool &uniform, std::vector<uint64_t> &off) {
  uniform = true;
  off.resize(n);
  size_t p = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (p + 2 > nbytes) return fail(SPRING_REORDER_E_IO, "record stream ends inside read %u", i);
    uint32_t len = (uint32_t)dna[p] | ((uint32_t)dna[p + 1] << 8);
    if ((int)len > L) return fail(SPRING_REORDER_E_ARG, "read %u has length %u > max_readlen %d", i, len, L);
    if ((int)len != L) uniform = false;
    off[i] = p;
    p += 2 + (len + 3) / 4;
    if (p > nbytes) return fail(SPRING_REORDER_E_IO, "record stream ends inside read %u", i);
  }
  return 0;
}

int spring_reorder_load_dna(spring_reorder_ctx *ctx, const uint8_t *dna, size_t nbytes, uint32_t n,
                            uint32_t max_readlen) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage != ST_CREATED) return fail(SPRING_REORDER_E_STATE, "load_dna: context already loaded");
  if (n && !dna) return fail(SPRING_REORDER_E_ARG, "dna is NULL");
  HIPCHK(hipSetDevice(ctx->dev));
  if (max_readlen && max_readlen <= (uint32_t)MAX_READ_LEN && n && nbytes == (size_t)n * (2u + (max_readlen + 3u) / 4u) &&
      !ctx->in_source_fallback) {
    // the size of a fixed-length stream: no serial walk over the records on the host, the device verifies the lengths
    DnaSource src;
    src.nbytes = nbytes;
    src.self = const_cast<uint8_t *>(dna);
    src.image = dna;
    src.fill = [](void *self, size_t o, void *dst, size_t len) -> int { memcpy(dst, (const uint8_t *)self + o, len); return 0; };
    ctx->in_source_fallback = true;  // (load_dna_source comes back here when the lengths turn out to vary)
    const int rs = load_dna_source(ctx, src, n, max_readlen);
    ctx->in_source_fallback = false;
    return rs;
  }
  int r = setup_geometry(ctx, n, max_readlen);
  if (r) return r;
  std::vector<uint64_t> off;
  r = scan_records(dna, nbytes, n, ctx->L, ctx->uniform, off);
  if (r) return r;
  ctx->dna_bytes = nbytes;
  DMALLOC(ctx->d_dna, nbytes + 16);
  auto mem_fill = [](void *self, size_t o, void *dst, size_t len) -> int { memcpy(dst, (const uint8_t *)self + o, len); return 0; };
  if ((r = upload_chunked(ctx, ctx->d_dna, nbytes, mem_fill, const_cast<uint8_t *>(dna)))) return r;
  if (!ctx->uniform) {
    DMALLOC(ctx->d_off, (size_t)n * sizeof(uint64_t));
    if ((r = upload_chunked(ctx, (uint8_t *)ctx->d_off, (size_t)n * sizeof(uint64_t), mem_fill, off.data()))) return r;
  }
  return unpack_on_device(ctx);
}

int spring_reorder_load_dna_device(spring_reorder_ctx *ctx, const void *d_dna, size_t nbytes, uint32_t n,
                                   uint32_t max_readlen, int32_t fixed_len) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage != ST_CREATED) return fail(SPRING_REORDER_E_STATE, "load_dna_device: context already loaded");
  HIPCHK(hipSetDevice(ctx->dev));
  int r = setup_geometry(ctx, n, max_readlen);
  if (r) return r;
  if (fixed_len) {
    const size_t rec = 2u + (max_readlen + 3u) / 4u;
    if (nbytes < rec * n) return fail(SPRING_REORDER_E_ARG, "device stream too short for %u fixed records", n);
    ctx->uniform = true;
    ctx->d_dna = (uint8_t *)d_dna;
    ctx->dna_borrowed = true;
    ctx->dna_bytes = nbytes;
    return unpack_on_device(ctx);
  }
  // variable length: record starts are sequentially dependent -> walk a host copy
  std::vector<uint8_t> h(nbytes);
  if (nbytes) HIPCHK(hipMemcpy(h.data(), d_dna, nbytes, hipMemcpyDeviceToHost));
  return spring_reorder_load_dna(ctx, h.data(), nbytes, n, max_readlen);
}

// ------------------------------------------------------------------ f1: FASTQ front end
// read_fastq_block + the N-split / packing loop of preprocess() (reference src/util.cpp:31-54,
// src/preprocess.cpp:186-214,:293-304) for the sequence lines, on the GPU: the clean reads land in the
// .dna record stream the unpack kernel reads (no input_clean_*.dna round trip through the file system).
namespace {
struct FqFile {
  uint8_t *d_txt = nullptr;
  uint64_t nbytes = 0, nlines = 0, nreads = 0;
  uint64_t *line_end = nullptr;
  uint32_t *len = nullptr, *fclean = nullptr, *szc = nullptr, *fN = nullptr, *szN = nullptr, *cidx = nullptr, *nidx = nullptr;
  uint64_t *coff = nullptr, *noff = nullptr;
  uint32_t n_clean = 0, n_N = 0, maxlen = 0, min_clean = 0xffffffffu;
  uint64_t clean_bytes = 0, N_bytes = 0;
};
}  // namespace

static int fq_scan_file(spring_reorder_ctx *ctx, const uint8_t *txt, size_t nbytes, FqFile &f, uint32_t *d_err) {
  hipStream_t st = ctx->st;
  f.nbytes = nbytes;
  if (!nbytes) return 0;
  DMALLOC(f.d_txt, nbytes + 16);
  HIPCHK(hipMemcpyAsync(f.d_txt, txt, nbytes, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[6], st));  // device time of the parse kernels (after the H2D copy)
  const uint64_t nblk = (nbytes + NL_CHUNK_BYTES - 1) / NL_CHUNK_BYTES;
  uint32_t *blk_cnt = nullptr;
  uint64_t *blk_off = nullptr;
  void *tmp = nullptr;
  size_t tb = 0;
  DMALLOC(blk_cnt, nblk * 4);
  DMALLOC(blk_off, nblk * 8);
  launch_nl_count(st, f.d_txt, nbytes, blk_cnt, nblk);
  HIPCHK(excl_scan_u32_to_u64(st, nullptr, tb, blk_cnt, blk_off, nblk));
  DMALLOC(tmp, tb);
  HIPCHK(excl_scan_u32_to_u64(st, tmp, tb, blk_cnt, blk_off, nblk));
  uint64_t last_off = 0;
  uint32_t last_cnt = 0;
  HIPCHK(hipMemcpyAsync(&last_off, blk_off + (nblk - 1), 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_cnt, blk_cnt + (nblk - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint64_t nl = last_off + last_cnt;
  const bool unterminated = txt[nbytes - 1] != '\n';  // std::getline still returns the last line
  f.nlines = nl + (unterminated ? 1 : 0);
  if (f.nlines % 4) return fail(SPRING_REORDER_E_ARG, "Invalid FASTQ(A) file. Number of lines not multiple of 4(2)");
  f.nreads = f.nlines / 4;
  if (f.nreads > 4294967290ull) return fail(SPRING_REORDER_E_ARG, "Too many reads.");
  DMALLOC(f.line_end, (f.nlines + 1) * 8);
  launch_nl_fill(st, f.d_txt, nbytes, blk_off, f.line_end, nblk);
  if (unterminated) {
    const uint64_t e = nbytes;
    HIPCHK(hipMemcpyAsync(f.line_end + nl, &e, 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  const size_t nr = f.nreads;
  if (!nr) { ctx->dfree(blk_cnt); ctx->dfree(blk_off); ctx->dfree(tmp); return 0; }
  DMALLOC(f.len, nr * 4); DMALLOC(f.fclean, nr * 4); DMALLOC(f.szc, nr * 4); DMALLOC(f.fN, nr * 4); DMALLOC(f.szN, nr * 4);
  DMALLOC(f.cidx, nr * 4); DMALLOC(f.nidx, nr * 4); DMALLOC(f.coff, nr * 8); DMALLOC(f.noff, nr * 8);
  uint32_t *lenc = nullptr;
  DMALLOC(lenc, nr * 4);
  launch_read_info(st, f.d_txt, f.line_end, nr, f.len, f.fclean, f.szc, f.fN, f.szN, lenc, d_err);
  HIPCHK(hipGetLastError());
  ctx->dfree(tmp); tmp = nullptr;
  size_t t1 = 0, t2 = 0, t3 = 0;
  HIPCHK(excl_scan_u32(st, nullptr, t1, f.fclean, f.cidx, nr));
  HIPCHK(excl_scan_u32_to_u64(st, nullptr, t2, f.szc, f.coff, nr));
  uint32_t *d_max = nullptr;
  DMALLOC(d_max, 16);
  HIPCHK(reduce_max_u32(st, nullptr, t3, f.len, d_max, nr));
  tb = std::max(t1, std::max(t2, t3));
  DMALLOC(tmp, tb);
  HIPCHK(excl_scan_u32(st, tmp, tb, f.fclean, f.cidx, nr));
  HIPCHK(excl_scan_u32(st, tmp, tb, f.fN, f.nidx, nr));
  HIPCHK(excl_scan_u32_to_u64(st, tmp, tb, f.szc, f.coff, nr));
  HIPCHK(excl_scan_u32_to_u64(st, tmp, tb, f.szN, f.noff, nr));
  HIPCHK(reduce_max_u32(st, tmp, tb, f.len, d_max, nr));
  HIPCHK(reduce_min_u32(st, tmp, tb, lenc, d_max + 1, nr));
  uint32_t lc = 0, lf = 0, ln = 0, lfn = 0, lsz = 0, lszn = 0;
  uint64_t lco = 0, lno = 0;
  HIPCHK(hipMemcpyAsync(&lc, f.cidx + (nr - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&lf, f.fclean + (nr - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&ln, f.nidx + (nr - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&lfn, f.fN + (nr - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&lco, f.coff + (nr - 1), 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&lsz, f.szc + (nr - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&lno, f.noff + (nr - 1), 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&lszn, f.szN + (nr - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f.maxlen, d_max, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f.min_clean, d_max + 1, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  f.n_clean = lc + lf; f.n_N = ln + lfn; f.clean_bytes = lco + lsz; f.N_bytes = lno + lszn;
  HIPCHK(hipEventRecord(ctx->ev[7], st));
  HIPCHK(hipEventSynchronize(ctx->ev[7]));
  { float ms = 0; if (hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]) == hipSuccess) ctx->fq_ms += ms; }
  ctx->dfree(blk_cnt); ctx->dfree(blk_off); ctx->dfree(tmp); ctx->dfree(d_max); ctx->dfree(lenc);
  return 0;
}

// A gzip'ed FASTQ buffer (magic 1f 8b; preprocess.cpp:154-183 reads .gz input through
// boost::iostreams::gzip_decompressor) is inflated on the host, every member of a multi-member file (bgzip,
// concatenated .gz) in turn; anything else is taken as text.
static int gunzip_if_needed(const uint8_t *&p, size_t &n, std::vector<uint8_t> &buf) {
  if (n < 2 || p[0] != 0x1f || p[1] != 0x8b) return 0;
  z_stream z;
  memset(&z, 0, sizeof(z));
  if (inflateInit2(&z, 15 + 16) != Z_OK) return fail(SPRING_REORDER_E_IO, "zlib: inflateInit2 failed");
  // nothing thrown in here may cross the C ABI: allocation failures come back as an error code.  The output buffer
  // starts modest and grows geometrically (std::vector), whatever the compressed size is.
  try {
    buf.clear();
    buf.reserve(std::min<size_t>(n * 4, (size_t)1 << 30));
    size_t in_pos = 0;
    std::vector<uint8_t> tmp(1 << 22);
    int zr = Z_OK;
    for (;;) {
      if (z.avail_in == 0 && in_pos < n) {
        const size_t take = std::min<size_t>(n - in_pos, 1u << 30);
        z.next_in = const_cast<uint8_t *>(p + in_pos);
        z.avail_in = (uInt)take;
        in_pos += take;
      }
      z.next_out = tmp.data();
      z.avail_out = (uInt)tmp.size();
      zr = inflate(&z, Z_NO_FLUSH);
      if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) {
        inflateEnd(&z);
        return fail(SPRING_REORDER_E_IO, "gzip error in the FASTQ input (zlib %d)", zr);
      }
      buf.insert(buf.end(), tmp.data(), tmp.data() + (tmp.size() - z.avail_out));
      if (zr == Z_STREAM_END) {
        if (z.avail_in == 0 && in_pos >= n) break;  // last member
        // what follows the member: another member, or zero padding up to the end of the image (tape / block
        // padding; gzip(1) ignores it too) -- anything else is an error below
        const uint8_t *rest = z.next_in;
        size_t nrest = z.avail_in;
        bool zeros = true;
        for (size_t i = 0; i < nrest && zeros; i++) zeros = rest[i] == 0;
        for (size_t i = in_pos; i < n && zeros; i++) zeros = p[i] == 0;
        if (zeros) break;
        if (inflateReset(&z) != Z_OK) { inflateEnd(&z); return fail(SPRING_REORDER_E_IO, "zlib: inflateReset failed"); }
      } else if (zr == Z_BUF_ERROR && z.avail_in == 0 && in_pos >= n) {
        inflateEnd(&z);
        return fail(SPRING_REORDER_E_IO, "gzip error in the FASTQ input (truncated member)");
      }
    }
  } catch (const std::bad_alloc &) {
    inflateEnd(&z);
    return fail(SPRING_REORDER_E_IO, "out of host memory while inflating the gzip FASTQ input");
  } catch (const std::length_error &) {
    inflateEnd(&z);
    return fail(SPRING_REORDER_E_IO, "gzip FASTQ input inflates beyond what one buffer can hold");
  }
  inflateEnd(&z);
  p = buf.data();
  n = buf.size();
  return 0;
}

int spring_reorder_load_fastq(spring_reorder_ctx *ctx, const uint8_t *fastq_1, size_t nbytes_1, const uint8_t *fastq_2,
                              size_t nbytes_2, spring_fastq_info *info) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage != ST_CREATED) return fail(SPRING_REORDER_E_STATE, "load_fastq: context already loaded");
  if ((nbytes_1 && !fastq_1) || (nbytes_2 && !fastq_2)) return fail(SPRING_REORDER_E_ARG, "NULL FASTQ buffer");
  std::vector<uint8_t> unz[2];
  {
    int gr = gunzip_if_needed(fastq_1, nbytes_1, unz[0]);
    if (!gr && fastq_2) gr = gunzip_if_needed(fastq_2, nbytes_2, unz[1]);
    if (gr) return gr;
  }
  HIPCHK(hipSetDevice(ctx->dev));
  hipStream_t st = ctx->st;
  const bool paired = fastq_2 != nullptr;
  uint32_t *d_err = nullptr;
  DMALLOC(d_err, 16);
  HIPCHK(hipMemsetAsync(d_err, 0, 4, st));
  FqFile f[2];
  double ms_dev = 0;
  int r = fq_scan_file(ctx, fastq_1, nbytes_1, f[0], d_err);
  if (r) return r;
  if (paired && (r = fq_scan_file(ctx, fastq_2, nbytes_2, f[1], d_err))) return r;
  uint32_t err = 0;
  HIPCHK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
  if (err & 1u) return fail(SPRING_REORDER_E_ARG, "Too long read length (please try --long/-l flag).");
  if (err & 2u) return fail(SPRING_REORDER_E_ARG, "Invalid character in a read (only A, C, G, T and N are supported).");
  if (paired && f[0].nreads != f[1].nreads) return fail(SPRING_REORDER_E_ARG, "Number of reads in paired files do not match.");
  if (f[0].nreads + f[1].nreads > 4294967290ull) return fail(SPRING_REORDER_E_ARG, "Too many reads.");
  const uint32_t n_clean = f[0].n_clean + f[1].n_clean;
  const uint64_t clean_bytes = f[0].clean_bytes + f[1].clean_bytes;
  const uint32_t maxlen = std::max(f[0].maxlen, f[1].maxlen);
  memset(&ctx->fq, 0, sizeof(ctx->fq));
  for (int j = 0; j < 2; j++) {
    ctx->fq.num_reads[j] = (uint32_t)f[j].nreads; ctx->fq.num_reads_clean[j] = f[j].n_clean; ctx->fq.num_reads_N[j] = f[j].n_N;
  }
  ctx->fq.max_readlen = maxlen;
  if (info) *info = ctx->fq;
  r = setup_geometry(ctx, n_clean, maxlen ? maxlen : 1);
  if (r) return r;
  ctx->dna_bytes = clean_bytes;
  DMALLOC(ctx->d_dna, clean_bytes + 16);
  DMALLOC(ctx->d_off, (size_t)std::max<uint32_t>(n_clean, 1) * 8);
  uint32_t cbase = 0, rbase = 0;
  uint64_t obase = 0;
  HIPCHK(hipEventRecord(ctx->ev[6], st));
  for (int j = 0; j < 2; j++) {
    if (!f[j].nreads) continue;
    DMALLOC(ctx->d_N[j], f[j].N_bytes + 16);
    DMALLOC(ctx->d_orderN[j], (size_t)std::max<uint32_t>(f[j].n_N, 1) * 4);
    DMALLOC(ctx->d_offN[j], (size_t)std::max<uint32_t>(f[j].n_N, 1) * 8);
    ctx->N_bytes[j] = f[j].N_bytes;
    // pos_N counts from the start of its own file (preprocess.cpp:299: num_reads[j] + i)
    launch_pack_reads(st, f[j].d_txt, f[j].line_end, f[j].nreads, f[j].len, f[j].fclean, f[j].cidx, f[j].coff, f[j].nidx,
                      f[j].noff, cbase, obase, 0u, ctx->d_dna, ctx->d_off, ctx->d_N[j], ctx->d_orderN[j], ctx->d_offN[j]);
    HIPCHK(hipGetLastError());
    cbase += f[j].n_clean; obase += f[j].clean_bytes; rbase += (uint32_t)f[j].nreads;
  }
  (void)rbase;
  // every clean read as long as max_readlen => fixed-size records, and the search skips the length loads
  ctx->uniform = n_clean > 0 && std::min(f[0].min_clean, f[1].min_clean) == maxlen;
  if (ctx->uniform) { ctx->dfree(ctx->d_off); ctx->d_off = nullptr; }
  r = unpack_on_device(ctx);
  if (r) return r;
  HIPCHK(hipEventRecord(ctx->ev[7], st));
  HIPCHK(hipStreamSynchronize(st));
  { float ms = 0; if (hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]) == hipSuccess) ctx->fq_ms += ms; }
  ms_dev = ctx->fq_ms;
  ctx->fq.ms_device = ms_dev;
  if (info) info->ms_device = ms_dev;
  for (int j = 0; j < 2; j++) {
    ctx->dfree(f[j].d_txt); ctx->dfree(f[j].line_end); ctx->dfree(f[j].len); ctx->dfree(f[j].fclean); ctx->dfree(f[j].szc);
    ctx->dfree(f[j].fN); ctx->dfree(f[j].szN); ctx->dfree(f[j].cidx); ctx->dfree(f[j].nidx); ctx->dfree(f[j].coff);
    ctx->dfree(f[j].noff);
  }
  ctx->dfree(d_err);
  return 0;
}

int spring_reorder_fastq_N(spring_reorder_ctx *ctx, int32_t which, uint8_t *n_dna, size_t cap, size_t *nbytes,
                           uint32_t *order_N, uint32_t *count) {
  if (!ctx || ctx->stage < ST_LOADED) return fail(SPRING_REORDER_E_STATE, "fastq_N: load_fastq first");
  if (which < 0 || which > 1) return fail(SPRING_REORDER_E_ARG, "which must be 0 or 1");
  HIPCHK(hipSetDevice(ctx->dev));
  const uint32_t nN = ctx->fq.num_reads_N[which];
  if (nbytes) *nbytes = ctx->N_bytes[which];
  if (count) *count = nN;
  if (n_dna && ctx->N_bytes[which]) {
    if (cap < ctx->N_bytes[which]) return fail(SPRING_REORDER_E_ARG, "fastq_N: buffer too small");
    HIPCHK(hipMemcpy(n_dna, ctx->d_N[which], ctx->N_bytes[which], hipMemcpyDeviceToHost));
  }
  if (order_N && nN) HIPCHK(hipMemcpy(order_N, ctx->d_orderN[which], (size_t)nN * 4, hipMemcpyDeviceToHost));
  return 0;
}

size_t spring_synth_dna_bytes(uint32_t n, uint32_t L) { return (size_t)n * (2u + (L + 3u) / 4u); }

int spring_synth_dna_host(uint8_t *dst, uint32_t n, uint32_t L, uint64_t G, uint64_t seed, uint32_t err_ppm) {
  if (L == 0 || L > (uint32_t)MAX_READ_LEN || G < L) return fail(SPRING_REORDER_E_ARG, "bad synth geometry");
  if ((err_ppm & SYN_PAIRED_FLAG) && (n & 1)) return fail(SPRING_REORDER_E_ARG, "a paired pool holds an even number of reads");
  const uint32_t rec = 2u + (L + 3u) / 4u, thr = syn_err_thr24(err_ppm);
  for (uint64_t i = 0; i < n; i++) {
    uint8_t *o = dst + i * rec;
    uint64_t pos; uint32_t rc;
    syn_read_params(seed, G, L, i, (err_ppm & SYN_PAIRED_FLAG) ? n / 2 : 0, &pos, &rc);
    o[0] = (uint8_t)(L & 0xff); o[1] = (uint8_t)(L >> 8);
    for (uint32_t b = 0; b < (L + 3) / 4; b++) {
      uint32_t v = 0;
      for (uint32_t q = 0; q < 4; q++) {
        uint32_t j = 4 * b + q;
        if (j < L) v |= syn_nat_to_spring(syn_read_base(seed, G, L, thr, i, j, pos, rc)) << (2 * q);
      }
      o[2 + b] = (uint8_t)v;
    }
  }
  return 0;
}

int spring_synth_genome_host(uint8_t *dst, uint64_t G, uint64_t seed, uint32_t flags) {
  if (!dst) return fail(SPRING_REORDER_E_ARG, "dst is NULL");
  for (uint64_t p = 0; p < G; p++) {
    uint64_t gp = p;
    if (flags & SYN_REPEAT_FLAG) {
      const uint64_t seg = G / 8, k = seg ? gp / seg : 0;
      if (seg && k < 8 && (k & 1) == 0) gp %= seg;
    }
    dst[p] = (uint8_t)"ACGT"[(flags & SYN_GENOMIC_FLAG) ? syn_genomic_base(seed, gp) : syn_genome_base(seed, gp)];
  }
  return 0;
}

int spring_synth_dna_device(void *d_dst, uint32_t n, uint32_t L, uint64_t G, uint64_t seed, uint32_t err_ppm) {
  if (!d_dst || L == 0 || L > (uint32_t)MAX_READ_LEN || G < L) return fail(SPRING_REORDER_E_ARG, "bad synth arguments");
  if ((err_ppm & SYN_PAIRED_FLAG) && (n & 1)) return fail(SPRING_REORDER_E_ARG, "a paired pool holds an even number of reads");
  launch_synth(nullptr, (uint8_t *)d_dst, n, L, G, seed, syn_err_thr24(err_ppm));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(nullptr));
  return 0;
}

int spring_reorder_load_synth(spring_reorder_ctx *ctx, uint32_t n, uint32_t L, uint64_t G, uint64_t seed,
                              uint32_t err_ppm) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage != ST_CREATED) return fail(SPRING_REORDER_E_STATE, "load_synth: context already loaded");
  if (L == 0 || L > (uint32_t)MAX_READ_LEN || G < L) return fail(SPRING_REORDER_E_ARG, "bad synth geometry");
  if ((err_ppm & SYN_PAIRED_FLAG) && (n & 1)) return fail(SPRING_REORDER_E_ARG, "a paired pool holds an even number of reads");
  HIPCHK(hipSetDevice(ctx->dev));
  int r = setup_geometry(ctx, n, L);
  if (r) return r;
  ctx->uniform = true;
  ctx->dna_bytes = spring_synth_dna_bytes(n, L);
  DMALLOC(ctx->d_dna, ctx->dna_bytes + 16);
  launch_synth(ctx->st, ctx->d_dna, n, L, G, seed, syn_err_thr24(err_ppm));
  HIPCHK(hipGetLastError());
  return unpack_on_device(ctx);
}

int spring_reorder_download_dna(spring_reorder_ctx *ctx, uint8_t *dst, size_t cap) {
  if (!ctx || ctx->stage < ST_LOADED) return fail(SPRING_REORDER_E_STATE, "download_dna: nothing loaded");
  if (cap < ctx->dna_bytes) return fail(SPRING_REORDER_E_ARG, "buffer too small");
  HIPCHK(hipSetDevice(ctx->dev));
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (ctx->dna_bytes) HIPCHK(hipMemcpy(dst, ctx->d_dna, ctx->dna_bytes, hipMemcpyDeviceToHost));
  return 0;
}

int spring_reorder_download_reads(spring_reorder_ctx *ctx, uint64_t *limbs, uint16_t *len) {
  if (!ctx || ctx->stage < ST_LOADED) return fail(SPRING_REORDER_E_STATE, "download_reads: nothing loaded");
  HIPCHK(hipSetDevice(ctx->dev));
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (!ctx->n) return 0;
  if (limbs)
    HIPCHK(hipMemcpy2D(limbs, (size_t)ctx->W * 8, ctx->d_reads, (size_t)ctx->S * 8, (size_t)ctx->W * 8, ctx->n,
                       hipMemcpyDeviceToHost));
  if (len) HIPCHK(hipMemcpy(len, ctx->d_lens, (size_t)ctx->n * 2, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------ dictionaries
static uint64_t pow2ceil(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define DBG_T(label)                                                                         \
  do {                                                                                       \
    if (dbg) { (void)hipStreamSynchronize(st); double t_ = now_ms(); fprintf(stderr, "[dict] %-14s %8.2f ms\n", label, t_ - t_last); t_last = t_; } \
  } while (0)

static TabView tab_view(const spring_reorder_ctx *ctx) {
  TabView t;
  t.buck = ctx->fpt;
  t.bshift = ctx->bshift; t.minz = ctx->minz; t.lshift = ctx->lshift;
  return t;
}

int spring_reorder_build_dict(spring_reorder_ctx *ctx) {
  double t_last = now_ms();
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  const bool dbg = ctx->o.debug != 0;  // stage timings on stderr, no effect on results
  if (ctx->stage != ST_LOADED) return fail(SPRING_REORDER_E_STATE, "build_dict: load reads first");
  HIPCHK(hipSetDevice(ctx->dev));
  hipStream_t st = ctx->st;
  const uint32_t n = ctx->n;
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  uint64_t *uhash[2] = {nullptr, nullptr};          // sorted unique mix64(key) per dictionary
  uint32_t *ustart[2] = {nullptr, nullptr}, *ucount[2] = {nullptr, nullptr};
  for (int l = 0; l < 2; l++) {
    DictDev &d = ctx->dict[l];
    uint32_t m = 0;
    uint32_t *d_slot = nullptr, *d_flag = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    if (n) {
      if (ctx->uniform) {
        m = ctx->L > d.end ? n : 0;
      } else {  // reads with len <= end are not in this dictionary (bitset_util.h:98-105)
        DMALLOC(d_flag, (size_t)n * 4);
        DMALLOC(d_slot, (size_t)n * 4);
        launch_flag_in_dict(st, ctx->d_lens, n, d.end, d_flag);
        HIPCHK(excl_scan_u32(st, nullptr, tmp_bytes, d_flag, d_slot, n));
        DMALLOC(d_tmp, tmp_bytes);
        HIPCHK(excl_scan_u32(st, d_tmp, tmp_bytes, d_flag, d_slot, n));
        uint32_t last_slot = 0, last_flag = 0;
        HIPCHK(hipMemcpyAsync(&last_slot, d_slot + (n - 1), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&last_flag, d_flag + (n - 1), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        m = last_slot + last_flag;
        ctx->dfree(d_tmp); d_tmp = nullptr;
      }
    }
    d.numreads = m;
    d.numkeys = 0;
    d.ndeep = 0; d.big_reads = 0; d.mid_reads = 0;
    if (m == 0) {
      DMALLOC(d.urec, 16);
      DMALLOC(d.ids, 16);
      DMALLOC(d.deep, 16);
      DMALLOC(d.d_ndeep, 16);
      HIPCHK(hipMemsetAsync(d.d_ndeep, 0, 12, st));
      if (d_flag) { ctx->dfree(d_flag); ctx->dfree(d_slot); }
      continue;
    }
    uint64_t *k_in = nullptr, *k_out = nullptr;
    uint32_t *v_in = nullptr, *cnt = nullptr, *d_nruns = nullptr;
    DMALLOC(k_in, (size_t)m * 8);
    DMALLOC(k_out, (size_t)m * 8);
    DMALLOC(v_in, (size_t)m * 4);
    DMALLOC(d.ids, (size_t)m * 4);
    DBG_T("alloc keys");
    launch_keys(st, ctx->d_reads, ctx->d_lens, ctx->uniform ? nullptr : d_slot, n, ctx->S, d.start, d.end, k_in, v_in);
    HIPCHK(hipGetLastError());
    DBG_T("k_keys");
    const unsigned end_bit = 64;  // k_keys emits mix64(key): all 64 bits are significant
    // stable LSD radix sort: equal keys (equal hashes: mix64 is a bijection) keep ascending read id
    // (bitset_util.h:192-210), and the unique keys come out in bucket order for k_tab_insert
    tmp_bytes = 0;
    HIPCHK(sort_pairs(st, nullptr, tmp_bytes, k_in, k_out, v_in, d.ids, m, end_bit));
    DMALLOC(d_tmp, tmp_bytes);
    HIPCHK(sort_pairs(st, d_tmp, tmp_bytes, k_in, k_out, v_in, d.ids, m, end_bit));
    DBG_T("sort");
    ctx->dfree(d_tmp); d_tmp = nullptr;
    ctx->dfree(v_in);
    // unique keys + run lengths (bitset_util.h:122-127), reuse k_in for the unique keys
    DMALLOC(cnt, (size_t)m * 4);
    DMALLOC(d_nruns, 16);
    tmp_bytes = 0;
    HIPCHK(rle(st, nullptr, tmp_bytes, k_out, m, k_in, cnt, d_nruns));
    DMALLOC(d_tmp, tmp_bytes);
    HIPCHK(rle(st, d_tmp, tmp_bytes, k_out, m, k_in, cnt, d_nruns));
    uint32_t numkeys = 0;
    HIPCHK(hipMemcpyAsync(&numkeys, d_nruns, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx->dfree(d_tmp); d_tmp = nullptr;
    ctx->dfree(k_out); ctx->dfree(d_nruns);
    d.numkeys = numkeys;
    DBG_T("rle");
    DMALLOC(ustart[l], (size_t)numkeys * 4);
    tmp_bytes = 0;
    HIPCHK(excl_scan_u32(st, nullptr, tmp_bytes, cnt, ustart[l], numkeys));
    DMALLOC(d_tmp, tmp_bytes);
    HIPCHK(excl_scan_u32(st, d_tmp, tmp_bytes, cnt, ustart[l], numkeys));
    DBG_T("scan");
    uhash[l] = k_in;
    ucount[l] = cnt;
    DMALLOC(d.urec, (size_t)numkeys * 16);
    // deep-bin list: at most one entry per DEEP_BIN reads of the dictionary
    DMALLOC(d.deep, ((size_t)m / DEEP_BIN + 1) * 4);
    DMALLOC(d.d_ndeep, 16);
    HIPCHK(hipMemsetAsync(d.d_ndeep, 0, 12, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx->dfree(d_tmp);
    if (d_flag) { ctx->dfree(d_flag); ctx->dfree(d_slot); }
  }
  // ---- one bucket table for both dictionaries.  The (key, dictionary) pairs are merged by hash = by bucket,
  // so k_tab_insert streams; a key of both dictionaries lands in the same bucket and one bucket fetch serves
  // the probes of both (k_search).  Exact map: 4-slot 32-byte fingerprint buckets + 16-byte records.
  // Load <= 0.2: 98 % of the probes are absent keys and stop at the first empty slot of their home bucket; a
  // full bucket (4 pairs) sends them on to the next one.  Halving the load from 0.4 cut full buckets from 5.6 %
  // to 0.7 % and the chains stage by 2.2 % (a further halving: another 1.5 %, for 17 GB more at 100 M reads).
  // opts.tab_scale = 1 / 2 / 4 overrides (1 = the smallest table, load <= 0.4).
  const uint64_t nk0 = ctx->dict[0].numkeys, nk1 = ctx->dict[1].numkeys, nm = nk0 + nk1;
  const int tab_scale = ctx->o.tab_scale > 0 ? ctx->o.tab_scale : 2;
  const uint64_t nb = std::max<uint64_t>(8, pow2ceil(std::max<uint64_t>(2, (nm * 10 + 15) / 16)) * (uint64_t)pow2ceil(tab_scale));
  ctx->bshift = 64;
  for (uint64_t v = nb; v > 1; v >>= 1) ctx->bshift--;
  // Minimizer addressing (TabView, reorder_device.h; opts.table_mode = 2, same results): windows of 32 bases, reads up to
  // 192 (what k_round_mc keeps a window-minimizer array in LDS for); anything else keeps the hash addressing.
  ctx->minz = (ctx->dict[0].end - ctx->dict[0].start + 1 == MINZ_WL && ctx->L <= 192 && ctx->o.table_mode == 2) ? 1 : 0;
  ctx->lshift = 32 - (64 - ctx->bshift - 2);  // lines = nb / 4
  if (ctx->lshift < 1) return fail(SPRING_REORDER_E_ARG, "build_dict: table too large for minimizer addressing");
  ctx->stats.table_minz = ctx->minz; ctx->stats.table_marked_lines = 0;
  DMALLOC(ctx->fpt, nb * 32);
  HIPCHK(hipMemsetAsync(ctx->fpt, 0, nb * 32, st));
  uint32_t *const tab_words = reinterpret_cast<uint32_t *>(ctx->fpt);
  DBG_T("alloc+memset tab");
  if (nm) {
    uint64_t *mv0 = nullptr, *mv1 = nullptr, *mh = nullptr, *mv = nullptr;
    void *d_tmp = nullptr;
    DMALLOC(mv0, std::max<uint64_t>(nk0, 1) * 8);
    DMALLOC(mv1, std::max<uint64_t>(nk1, 1) * 8);
    launch_iota_tag(st, mv0, nk0, 0ull);
    launch_iota_tag(st, mv1, nk1, 1ull << 63);
    const uint64_t *h_in = nullptr, *v_in = nullptr;
    if (nk0 && nk1) {
      DMALLOC(mh, nm * 8);
      DMALLOC(mv, nm * 8);
      size_t tb = 0;
      HIPCHK(merge_by_hash(st, nullptr, tb, uhash[0], uhash[1], mv0, mv1, mh, mv, nk0, nk1));
      DMALLOC(d_tmp, tb);
      HIPCHK(merge_by_hash(st, d_tmp, tb, uhash[0], uhash[1], mv0, mv1, mh, mv, nk0, nk1));
      h_in = mh; v_in = mv;
    } else {
      h_in = nk0 ? uhash[0] : uhash[1];
      v_in = nk0 ? mv0 : mv1;
    }
    DBG_T("merge");
    DictBuild db[2];
    for (int l = 0; l < 2; l++) {
      DictDev &d = ctx->dict[l];
      db[l].ustart = ustart[l]; db[l].ucount = ucount[l]; db[l].ids = d.ids; db[l].urec = d.urec;
      db[l].deep = d.deep; db[l].ndeep = d.d_ndeep;
    }
    uint32_t *bk_in = nullptr, *bk_out = nullptr, *d_marked = nullptr;
    uint64_t *tp_in = nullptr, *tp_out = nullptr;
    ctx->marked_lines = 0;
    void *d_tmp2 = nullptr;
    if (!ctx->minz) {
      launch_tab_insert(st, h_in, v_in, nm, db[0], db[1], tab_words, ctx->bshift);
    } else {
      DMALLOC(bk_in, nm * 4); DMALLOC(bk_out, nm * 4);
      DMALLOC(tp_in, nm * 8); DMALLOC(tp_out, nm * 8);
      launch_minz_prepare(st, h_in, v_in, nm, db[0], db[1], ctx->lshift, bk_in, tp_in);
      DBG_T("minz prepare");
      const unsigned end_bit = (unsigned)(64 - ctx->bshift);  // bucket index bits
      size_t tb = 0;
      HIPCHK(sort_pairs_u32_u64(st, nullptr, tb, bk_in, bk_out, tp_in, tp_out, nm, end_bit));
      DMALLOC(d_tmp2, tb);
      HIPCHK(sort_pairs_u32_u64(st, d_tmp2, tb, bk_in, bk_out, tp_in, tp_out, nm, end_bit));
      DBG_T("minz sort");
      DMALLOC(d_marked, 16);
      HIPCHK(hipMemsetAsync(d_marked, 0, 4, st));
      launch_tab_insert_minz(st, bk_out, tp_out, nm, tab_words, ctx->bshift, d_marked);
      HIPCHK(hipMemcpyAsync(&ctx->marked_lines, d_marked, 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipGetLastError());
    uint32_t nd[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int l = 0; l < 2; l++)
      HIPCHK(hipMemcpyAsync(nd[l], ctx->dict[l].d_ndeep, 12, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int l = 0; l < 2; l++) { ctx->dict[l].ndeep = nd[l][0]; ctx->dict[l].big_reads = nd[l][1]; ctx->dict[l].mid_reads = nd[l][2]; }
    if (dbg) fprintf(stderr, "[dict] reads %u + %u, keys %u + %u, bins >= %u: %u + %u, reads in bins >= %u: %u + %u, >= %u: %u + %u\n",
                     ctx->dict[0].numreads, ctx->dict[1].numreads, ctx->dict[0].numkeys, ctx->dict[1].numkeys, DEEP_BIN, nd[0][0], nd[1][0],
                     MID_BIN, nd[0][2], nd[1][2], BIG_BIN, nd[0][1], nd[1][1]);
    ctx->stats.table_marked_lines = ctx->marked_lines;
    DBG_T("insert");
    ctx->dfree(mv0); ctx->dfree(mv1); ctx->dfree(mh); ctx->dfree(mv); ctx->dfree(d_tmp);
    ctx->dfree(bk_in); ctx->dfree(bk_out); ctx->dfree(tp_in); ctx->dfree(tp_out); ctx->dfree(d_tmp2); ctx->dfree(d_marked);
  }
  for (int l = 0; l < 2; l++) { ctx->dfree(uhash[l]); ctx->dfree(ustart[l]); ctx->dfree(ucount[l]); }
  DBG_T("free temps");
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  ctx->stage = ST_DICT;
  return 0;
}

int spring_reorder_dict_lookup(spring_reorder_ctx *ctx, int32_t which, const uint64_t *keys, uint32_t nkeys,
                               uint32_t *bin_size, uint32_t *bin_ids, size_t ids_cap) {
  if (!ctx || ctx->stage < ST_DICT) return fail(SPRING_REORDER_E_STATE, "dict_lookup: build_dict first");
  if (which < 0 || which > 1) return fail(SPRING_REORDER_E_ARG, "which must be 0 or 1");
  HIPCHK(hipSetDevice(ctx->dev));
  DictDev &d = ctx->dict[which];
  uint64_t *dk = nullptr;
  uint32_t *ds = nullptr, *dc = nullptr;
  DMALLOC(dk, (size_t)nkeys * 8);
  DMALLOC(ds, (size_t)nkeys * 4);
  DMALLOC(dc, (size_t)nkeys * 4);
  std::vector<uint32_t> hs(nkeys), hc(nkeys), hids(d.numreads);
  if (nkeys) {
    HIPCHK(hipMemcpyAsync(dk, keys, (size_t)nkeys * 8, hipMemcpyHostToDevice, ctx->st));
    launch_dict_lookup(ctx->st, tab_view(ctx), d.urec, which, ctx->d_reads, ctx->S, d.start, d.end, dk, nkeys, ds, dc);
    HIPCHK(hipMemcpyAsync(hs.data(), ds, (size_t)nkeys * 4, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(hc.data(), dc, (size_t)nkeys * 4, hipMemcpyDeviceToHost, ctx->st));
  }
  if (d.numreads) HIPCHK(hipMemcpyAsync(hids.data(), d.ids, (size_t)d.numreads * 4, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->dfree(dk); ctx->dfree(ds); ctx->dfree(dc);
  size_t o = 0;
  for (uint32_t i = 0; i < nkeys; i++) {
    if (hc[i] != 0xffffffffu && (hc[i] & 0x80000000u)) {  // single-read bin: the bucket holds the read id itself
      bin_size[i] = 1;
      if (o + 1 > ids_cap) return fail(SPRING_REORDER_E_ARG, "bin_ids too small");
      bin_ids[o++] = hs[i];
      continue;
    }
    bin_size[i] = hc[i];
    if (hc[i] == 0xffffffffu) continue;
    if (o + hc[i] > ids_cap) return fail(SPRING_REORDER_E_ARG, "bin_ids too small");
    memcpy(bin_ids + o, hids.data() + hs[i], (size_t)hc[i] * 4);
    o += hc[i];
  }
  return 0;
}

// ------------------------------------------------------------------ chains
// fields of DevParams shared by run_chains and mg_begin; the tuning values come from spring_reorder_opts (results
// do not depend on them) -- DESIGN.md section 6
// the dictionary averages >= 1.3 reads per key: deep-coverage pool (a few hundred x and up)
static bool dict_is_deep(const spring_reorder_ctx *ctx) {
  const uint64_t nd = (uint64_t)ctx->dict[0].numreads + ctx->dict[1].numreads, nk = (uint64_t)ctx->dict[0].numkeys + ctx->dict[1].numkeys;
  return nd * 10 >= nk * 13;
}
// the kernel variant with the deep-bin machinery (dead tails trimmed while scanning, balanced scan, resumed searches,
// owner-first apply) pays from ~1.15 reads per key on, before the chain-count rule above (whose threshold is a matter
// of compressed size).  tools/variant_probe2.py, chains stage in ms, four chains per wavefront / one chain / one chain
// with the deep-bin machinery: 100 M reads at 100x (1.09 reads per key) 418 / 423 / 439, at 200x (1.18) 493 / 472 / 471;
// 20 M reads at 100x 130 / 109 / 113, at 200x 159 / 124 / 125; at 400x and up (>= 1.3) the deep variant wins by 15-40 %
// A HEAVY TAIL of bins: at least 0.1 % of the dictionary's reads sit in bins of >= MID_BIN (64) entries although the
// average bin may hold a single read (uniform genomes have none there up to 6 400x; genome-like pools 0.15 % at 5 M reads,
// 1.1 % at 20 M, 1.3 % at 100 M) -- what the repeat families of a real genome do to the dictionary (round 4,
// SPRING_SYNTH_GENOMIC pools: 1.02 reads per key on average, bins of thousands of reads for the 32-mers of the big
// families).  Averages do not see it, and the shallow-dictionary kernels then walk those bins one candidate per
// dependent step: 100 M x 150 bp took 41 s (2.4 Mreads/s) before this rule, against 0.44 s on the uniform genome.
static bool dict_has_heavy_tail(const spring_reorder_ctx *ctx) {
  const uint64_t mid = (uint64_t)ctx->dict[0].mid_reads + ctx->dict[1].mid_reads, nd = (uint64_t)ctx->dict[0].numreads + ctx->dict[1].numreads;
  return nd > 0 && mid * 1000 >= nd;
}
// ... a quarter of them: deep coverage all over (25 600x, PhiX-like pools) -- thousands of chains sit on the same loci and
// most proposals are lost: the alternatives schedule (opts.alternatives) pays
static bool dict_is_contended(const spring_reorder_ctx *ctx) {
  const uint64_t mid = (uint64_t)ctx->dict[0].mid_reads + ctx->dict[1].mid_reads, nd = (uint64_t)ctx->dict[0].numreads + ctx->dict[1].numreads;
  return nd > 0 && mid * 4 >= nd;
}
static bool dict_wants_deep_kernel(const spring_reorder_ctx *ctx) {
  const uint64_t nd = (uint64_t)ctx->dict[0].numreads + ctx->dict[1].numreads, nk = (uint64_t)ctx->dict[0].numkeys + ctx->dict[1].numkeys;
  return nd * 100 >= nk * 115 || dict_has_heavy_tail(ctx);
}
// ... and a quarter of its reads sit in bins of >= BIG_BIN entries: bins of hundreds of reads are the rule (PhiX-like
// pools, tens of thousands x): long searches go to k_long, and more than 65 536 chains make it slower, not faster
static bool dict_is_very_deep(const spring_reorder_ctx *ctx) {
  const uint64_t big = (uint64_t)ctx->dict[0].big_reads + ctx->dict[1].big_reads, nd = (uint64_t)ctx->dict[0].numreads + ctx->dict[1].numreads;
  return dict_is_deep(ctx) && nd > 0 && big * 4 >= nd;
}
static void fill_params(spring_reorder_ctx *ctx, DevParams &P) {
  const spring_reorder_opts &o = ctx->o;
  P.reads = ctx->d_reads; P.lens = ctx->d_lens; P.n = ctx->n;
  P.L = ctx->L; P.W = ctx->W; P.S = ctx->S; P.Lpad = ctx->Lpad;
  P.maxshift = ctx->L / 2;  // reorder.h:750
  memset(P.plan, 0, sizeof(P.plan));
  if (o.first_shifts < 0) { P.plan[0][0] = 4; P.plan[0][1] = 4; P.plan[0][2] = 8; P.plan[0][3] = 8; }  // progressive (experiment)
  else if (o.first_shifts > 0) { P.plan[0][0] = std::min(16, o.first_shifts); P.plan[0][1] = 16; }
  else if (o.fused >= 0 && o.fused != 2 && !o.collect_stats && !o.force_literal_update) {
    // four chains per wavefront keep more searches in flight: a narrower first batch (fewer wasted fetches past the
    // winner) is worth its extra dependent step there (417 vs 426 ms at 100 M x 150 bp; 8 + 16 stays for k_round)
    P.plan[0][0] = 4; P.plan[0][1] = 8; P.plan[0][2] = 16;
  } else { P.plan[0][0] = 8; P.plan[0][1] = 16; }   // one narrow batch, one wide
  P.plan[1][0] = 16; P.plan[1][1] = 16;
  bool user_plan[2] = {false, false};
  for (int w = 0; w < 2; w++) {  // opts.plan0 / plan1: an explicit plan wins over every rule here and below
    const int32_t *up = w ? o.plan1 : o.plan0;
    if (up[0] <= 0) continue;
    int k = 0, sum = 0;
    int tmp[6] = {0, 0, 0, 0, 0, 0};
    for (; k < 6 && up[k] > 0; k++) {
      if (up[k] > 16 || sum + up[k] > 32) break;
      tmp[k] = up[k]; sum += up[k];
    }
    if (k == 0) continue;
    memcpy(P.plan[w], tmp, sizeof(tmp));
    user_plan[w] = true;
  }
  ctx->user_plan0 = user_plan[0];
  P.seed_wide = o.seed_wide < 0 ? 0 : 1;
  P.search_wpb = (o.search_wpb == 1 || o.search_wpb == 2 || o.search_wpb == 4) ? o.search_wpb : 1;  // 1: a block is a chain; its slot frees when that chain is done
  P.dbg_search_lds = std::max(0, o.dbg_search_lds);
  P.dbg_apply_lds = std::max(0, o.dbg_apply_lds);
  P.uniform_len = ctx->uniform ? 1 : 0;
  P.wl = ctx->dict[0].end - ctx->dict[0].start + 1;  // == dict[1].end - dict[1].start + 1 (reorder.h:751-759)
  for (int l = 0; l < 2; l++) {
    P.dstart[l] = ctx->dict[l].start; P.dend[l] = ctx->dict[l].end; P.numkeys[l] = ctx->dict[l].numkeys;
    P.urec[l] = ctx->dict[l].urec; P.ids[l] = ctx->dict[l].ids;
    P.epos[l] = nullptr;
  }
  P.idmask = 0xffffffffu;
  P.tab = tab_view(ctx);
  // deep data (>= 1.3 reads per dictionary key on average: coverage of a few hundred x and up): the chain kernel
  // trims dead bin tails while it scans; opts.deep_bins = 1 / -1 forces the variant on / off (same results)
  P.deep_bins = o.deep_bins ? (o.deep_bins > 0) : dict_wants_deep_kernel(ctx);
  if (o.alternatives == 2) P.deep_bins = 1;  // (the second candidate is found by the deep-bin variants of the round kernel)
  // ... and the next read of a chain sits at shift 0 or 1 nearly always, while every verified bin a batch holds past the
  // winner is scanned for nothing: a narrow first batch (2 + 6 + 8 + 16 shifts instead of 4 + 8 + 16: 1 600x -3 %, 6 400x
  // -4 %, 25 600x -7 %, PhiX-like -5 %; 1 + 3 + 4 + 8 + 16 the same within 1 %, 1 + 1 + 2 + 4 + 8 + 16 slower at 400x)
  if (P.deep_bins && !dict_is_deep(ctx) && o.first_shifts == 0 && !user_plan[0]) {
    // (between the two thresholds bins hold two or three reads: 8 + 16 as for one chain per wavefront elsewhere)
    memset(P.plan[0], 0, sizeof(P.plan[0]));
    P.plan[0][0] = 8; P.plan[0][1] = 16;
  }
  if (P.deep_bins && dict_is_deep(ctx) && o.first_shifts == 0 && o.fused >= 0 && !o.collect_stats && !o.force_literal_update &&
      !user_plan[0]) {
    memset(P.plan[0], 0, sizeof(P.plan[0]));
    P.plan[0][0] = 2; P.plan[0][1] = 6; P.plan[0][2] = 8; P.plan[0][3] = 16;
    // (a fresh seed of such a pool finds its first match like any other chain: 4 + 12 + 16 instead of the wide 16 + 16:
    // PhiX-like -3 %, the 20 M-read pools within 1 %)
    if (!user_plan[1]) {
      memset(P.plan[1], 0, sizeof(P.plan[1]));
      P.plan[1][0] = 4; P.plan[1][1] = 12; P.plan[1][2] = 16;
    }
  }
  // long searches of such pools go from k_round to k_long after this many compare passes (same results for every value)
  // -- in pools where bins of hundreds of reads are the rule (a quarter of the dictionary's reads in bins of >= BIG_BIN
  // entries: PhiX-like, tens of thousands x), where a failing search compares thousands of candidates.  Below that a
  // hand-over does not pay (k_long's own latency is added to the round: 25 600x +-0, 6 400x -4 %), so those pools run
  // the variant of k_round without it; opts.long_budget > 0 forces it on (tests), -1 off.
  // (heavy tail on a dictionary that is shallow on average: the long searches are what a round waits for; pools that are
  // deep on average keep their own rule -- 25 600x: 178 ms without the hand-over, 195 with)
  const bool very_deep = P.deep_bins && (dict_is_very_deep(ctx) || (dict_has_heavy_tail(ctx) && !dict_is_deep(ctx)));
  P.long_budget = o.long_budget < 0 ? 0 : (o.long_budget ? o.long_budget : (very_deep ? 8 : 0));
  P.long_blocks = 256;
  // ... and only when at least this many bin entries are still ahead of it; budgets below 8 hand over unconditionally
  // (tests).  Round 5 (the long searches in three kernels, parts scanned by small blocks: what is handed over no longer
  // waits for the round's longest search): 512 -- genome-like pools, chains stage with 2 048 / 1 024 / 512 / 256 entries:
  // 20 M reads 221 / 210 / 210 / 220 ms, 100 M reads 1 589 / - / 1 516 / 1 558; PhiX-like pool 214 / 208 / 207 (off: 225).
  // (round 3, one block of 16 wavefronts per search: 2 048 / 4 096 / 8 192 entries 219 / 225 / 228 ms on the PhiX-like pool)
  P.long_min = P.long_budget < 8 ? 0 : 512;
  if (o.long_min > 0) P.long_min = o.long_min;
  if (o.long_blocks > 0) P.long_blocks = o.long_blocks;
  // a search whose first turn lists more than this many chunks of 64 bin entries is cut into parts (k_long_list)
  P.long_part = o.long_split > 0 ? o.long_split : (o.long_split < 0 ? 0 : 192);
  P.longq = nullptr;
  P.lctl = nullptr; P.lparts = nullptr; P.lhead = nullptr; P.lbin = nullptr; P.lbcode = nullptr; P.lbin_stride = 0;
  P.sig[0] = P.sig[1] = nullptr;
}
// Default chain count.  ~1000 reads per chain: the compressed size grows with the chain count on ordinary coverage
// (+7 % from n/1024 to 65 536 chains at 16 M reads).  On deep-coverage pools (the dictionary
// averages >= 1.3 reads per key: a few hundred x and up) it is the other way round -- 65 536 chains instead of
// n/1024 give 1.3 % / 2.0 % / 1.2 % SMALLER streams at 400x / 1 600x / 6 400x (real BSC, 20 M reads; flat on a
// PhiX-like pool; n/128 and n/256 within 0.4 % of each other at 4 M reads) and run 13 % / 9 % faster, 2.6x on the
// PhiX-like pool -- so those get ~128 reads per chain.
// Contended deep pools (every proposal fought over by many chains) gain from still more chains: at 20 M reads, 131 072
// instead of 65 536 chains run 1 600x / 6 400x / 25 600x 5 / 13 / 20 % faster at -1.4 / -0.6 / +0.3 % of the compressed
// size (profiles/r03_chain_count_deep.txt), so their cap is 131 072; pools of very deep bins stay at 65 536 (the
// PhiX-like pool takes 220 ms with 65 536 chains, 350 with 131 072).
// Pools with a heavy tail of bins (dict_has_heavy_tail; shallow on average): a round lasts as long as its longest bin scans
// whatever the chain count, so more chains per round are nearly free: 20 M genome-like reads, chains stage 1 453 / 737 /
// 511 / 483 ms with 19 531 / 65 536 / 131 072 / 262 144 chains (profiles/r04_genomic.txt) -- n / 128 up to 131 072 as well.
static uint32_t auto_chains(uint32_t n, bool deep, bool very_deep, bool heavy_tail = false) {
  uint64_t k = (deep || heavy_tail) ? n >> 7 : n >> 10;
  const uint64_t cap = ((deep && !very_deep) || (heavy_tail && !deep)) ? 131072 : 65536;
  if (k < 1) k = 1;
  if (k > cap) k = cap;
  // from the chain count at which the chains run as two groups: a multiple of 2048, so that the same default applies -- and
  // cuts into the same two groups -- on one GPU and on a pool of 2, 4 or 8 (setup_chains)
  if (k >= 16384) k &= ~2047ull;
  return (uint32_t)k;
}

// device state of the chain phase for the chains [c0, c0 + K) of Ktot this context owns; `d_prop`: caller's
// proposal buffer or null.  Used by run_chains (c0 = 0, K = Ktot) and mg_begin.
static int setup_chains(spring_reorder_ctx *ctx, uint32_t K, uint32_t c0, uint32_t Ktot, bool fused, void *d_prop,
                        bool allow_phases = false) {
  hipStream_t st = ctx->st;
  const uint32_t n = ctx->n;
  ctx->K = K;
  DevParams &P = ctx->P;
  fill_params(ctx, P);
  // the bitmap is padded to whole blocks of 2^UBLK_SHIFT reads (find_seed reads a block's words; bits >= n are set)
  const uint64_t nublk = std::max<uint64_t>(((uint64_t)n + (1u << UBLK_SHIFT) - 1) >> UBLK_SHIFT, 1);
  const uint64_t nwords = nublk << (UBLK_SHIFT - 6);
  const size_t nn = std::max<uint32_t>(n, 1);
  // options whose values are refused: before anything is allocated
  if (ctx->o.alternatives > 2) return fail(SPRING_REORDER_E_ARG, "alternatives: <= 0 (library's choice), 1 or 2");
  if (ctx->o.phases > 2) return fail(SPRING_REORDER_E_ARG, "phases: <= 0 (library's choice), 1 or 2");
  // (every refusal of an option combination happens here, before the allocations)
  P.K = K; P.c0 = c0; P.Ktot = Ktot;
  P.fused = fused ? 1 : 0;
  // candidates per match proposal (opts.alternatives): the second one travels in the proposal word beside the first
  // (27 bits) and is resolved by k_alt_resolve in front of k_mg_mark: fused rounds, deep-bin kernel variants
  {
    const bool can = fused && P.deep_bins && n < ALT_MAX_READS && Ktot < ALT_KEY;
    // (<= 0: the library's choice.  Two candidates cost a successful search one more look at its winner's bin (+10 % per round)
    // and pay where most proposals are lost: 20 M reads, chains stage with one / two candidates: 100x 109 / 121 ms, 400x 113 /
    // 117, 1 600x 123 / 124, 6 400x 137 / 134, 25 600x 169 / 152, PhiX-like 184 / 158, genome-like 216 / 240 -- so: pools with
    // a quarter of the dictionary's reads in bins of >= MID_BIN entries; profiles/r05_alternatives.txt)
    const int want = ctx->o.alternatives > 0 ? ctx->o.alternatives : (dict_is_contended(ctx) ? 2 : 1);
    if (ctx->o.alternatives == 2 && !can)
      return fail(SPRING_REORDER_E_ARG, "alternatives = 2 needs the fused round (no literal consensus path, fused >= 0) and a pool of fewer than %u reads", ALT_MAX_READS);
    P.alts = (want == 2 && can) ? 2 : 1;
    ctx->stats.alternatives = (uint64_t)P.alts;
  }
  // The chain schedule (opts.phases, DevParams::phases): two chain groups whose rounds alternate.  What it buys is what a
  // round costs beyond its chains -- the drain of the round kernel, the mark step, the kernel boundaries (~46 us + 158 us per
  // 65 536 chains with one group): two half launches side by side cost what their chains cost.  Chains stage, 25x pools,
  // one group / two groups (one chain per wavefront) / two groups (four chains per wavefront): 15 M reads (14 648 chains)
  // 82 / 81 / 105 ms, 25 M 119 / 109 / 121, 30 M 140 / 129 / 130, 35 M 159 / 149 / 142, 70 M 285 / 234 / 234, 100 M
  // 398 / - / 335; 5 M and 10 M reads are slower with two groups (47 -> 59, 65 -> 70: a group's launch no longer fills
  // the chip).  So: two groups from 16 384 chains on, four chains per wavefront from 32 768 on (with one group: 49 152).
  {
    // (the groups are cut from the GLOBAL chain ids, so a pool's output does not depend on the number of ranks: group 0 = chains
    // [0, half), half = Ktot / 2 to the nearest multiple of 2048; a rank of a pool owns an equal slice of each group)
    const uint32_t world = K ? Ktot / K : 1, rank = K ? c0 / K : 0;
    const uint32_t half = (uint32_t)std::max<uint64_t>((((uint64_t)Ktot / 2 + 1024) / 2048) * 2048, 2048);
    const uint32_t nmid = (uint32_t)(((uint64_t)n / 2) >> UBLK_SHIFT << UBLK_SHIFT);  // group 1's seeds: reads [0, nmid)
    // (n >= Ktot: every chain starts with a seed of its own, reorder.h:405-421 -- with fewer reads than chains only chain 0 runs, the
    // second group would have no chain at all and nobody would ever pick the seeds of its range)
    const bool pool_ok = world == 1 ? (c0 == 0 && !d_prop)
                                    : ((uint64_t)K * world == Ktot && half % (MARK_BLOCK * world) == 0 && (Ktot - half) % (MARK_BLOCK * world) == 0);
    const bool can = allow_phases && fused && K > 0 && pool_ok && Ktot >= 4096 && half < Ktot && n < 0x80000000u && nmid > 0 && n >= Ktot;
    // Deep-bin pools (one chain per wavefront, up to 131 072 chains), one group / two, chains stage in ms: 20 M reads (131 072 chains)
    // 400x 112 / 102, 1 600x 122 / 115, 6 400x 136 / 131, 25 600x (two candidates per proposal) 151 / 150; 10 M reads (78 125 chains)
    // 400x 60 / 58, 6 400x 74 / 79, 25 600x 85 / 90; 5 M reads (39 062) 34 / 35, 44 / 50, 55 / 59; 2.5 M reads (19 531) 21 / 26,
    // 29 / 38, 37 / 45 -- what is left on these pools is the round kernel's own throughput, and a group's launch has to fill the
    // chip on its own: two groups only at the cap of 131 072 chains, and not on contended pools (two candidates per proposal).
    // Pools whose long searches go to the k_long kernels stay with one group: PhiX-like 150 / 165 ms, genome-like 100 M reads
    // 1 475 / 1 619 -- blocks of 4-8 wavefronts starve beside the other group's one-wavefront workgroups.
    // (the rule looks at the pool -- Ktot, the dictionary --, never at a rank's share: the choice changes the output)
    const bool long_kernels = P.deep_bins && P.long_budget > 0;
    const bool pays = P.deep_bins ? (Ktot >= 131072 && P.alts == 1 && !long_kernels) : Ktot >= 16384;
    // (the library's own choice also asks for a multiple of 2048 chains -- which the default chain count is from 16 384 on --
    // whatever the number of ranks: both groups then split evenly over 1, 2, 4 or 8 ranks into slices of whole mark-step
    // blocks, and the choice -- one group or two -- is the same for all of them)
    const int want = ctx->o.phases > 0 ? ctx->o.phases : (pays && Ktot % 2048 == 0 ? 2 : 1);
    if (ctx->o.phases > 2) return fail(SPRING_REORDER_E_ARG, "phases: 0 (library's choice), 1 or 2");
    if (ctx->o.phases == 2 && !can)
      return fail(SPRING_REORDER_E_ARG, "phases = 2 needs the fused round (fused >= 0, no literal consensus path), at least 4096 "
                  "chains and 8192 .. 2^31 - 1 reads, at least as many reads as chains; in a pool over G GPUs both groups' chain "
                  "counts must be multiples of %u x G", MARK_BLOCK);
    P.phases = (want == 2 && can) ? 2 : 1;
    ctx->nmid = nmid;
    if (P.phases == 2) {
      const uint32_t sA = half / world, sB = (Ktot - half) / world;
      ctx->grp[0].g0 = 0; ctx->grp[0].Kg = sA; ctx->grp[0].c0 = rank * sA; ctx->grp[0].gg0 = 0; ctx->grp[0].gKg = half;
      ctx->grp[1].g0 = sA; ctx->grp[1].Kg = sB; ctx->grp[1].c0 = half + rank * sB - sA; ctx->grp[1].gg0 = half; ctx->grp[1].gKg = Ktot - half;
      ctx->Kh = sA;
    } else {
      ctx->grp[0].g0 = 0; ctx->grp[0].Kg = K; ctx->grp[0].c0 = c0; ctx->grp[0].gg0 = 0; ctx->grp[0].gKg = Ktot;
      ctx->grp[1] = spring_reorder_ctx::GroupGeom();
      ctx->grp[1].g0 = K; ctx->grp[1].gg0 = Ktot;
      ctx->Kh = K;
    }
    ctx->stats.phases = (uint64_t)P.phases;
  }
  DMALLOC(P.taken, nwords * 8);
  DMALLOC(P.ublk, nublk * 4);
  DMALLOC(P.resv, nn * 4);
  const size_t needy_bytes = (((size_t)Ktot + 31) / 32 + 255) / 256 * 256 * 4;  // find_seed reads whole 256-word groups
  DMALLOC(P.needy, needy_bytes);
  DMALLOC(P.glob, sizeof(Globals));
  P.cursor = &P.glob->cursor;
  DMALLOC(P.chains, (size_t)K * sizeof(Chain));
  DMALLOC(P.cnt, (size_t)K * 2 * ctx->Lpad * sizeof(int4));
  DMALLOC(P.cnt8, (size_t)K * 2 * ctx->Lpad * sizeof(uint32_t) + 64);  // (k_round_mc reads whole quads: up to 12 bytes past a column)
  // append buffers: n records + one partly filled CHUNK per chain
  const size_t cap = (size_t)n + (size_t)K * CHUNK;
  if (cap > 0xfffffff0ull) return fail(SPRING_REORDER_E_ARG, "n + K*%u exceeds the 32-bit slot space", CHUNK);
  ctx->cap = cap;
  const size_t nchunk = cap / CHUNK + 2;
  DMALLOC(P.e_rec, cap * sizeof(uint4)); DMALLOC(P.e_chunk, nchunk * sizeof(uint2));
  DMALLOC(P.s_rec, cap * 4); DMALLOC(P.s_chunk, nchunk * sizeof(uint2));
  P.mc = ctx->o.fused == 2 ? 0 : 1;  // opts.fused = 2: one chain per wavefront everywhere (A/B, tests)
  // Four chains per wavefront pay when a launch holds several wavefronts per slot (5 120 slots of four chains): with
  // fewer chains the GPU is not full and every wavefront waits for the slowest of its four.  25x, chains stage, four
  // chains / one chain per wavefront: 4 882 chains (5 M reads) 74 / 47 ms, 19 531 (20 M) 106 / 100, 39 062 (40 M)
  // 184 / 179, 65 536 (70 M) 282 / 293, 65 536 (100 M) 405 / 418 (tools/variant_probe2.py; at 60-100x the one-chain
  // kernel's lead below 40 000 chains is 5-16 %)
  // known-absent window masks (reorder_round_mc.h: search_ka): reads up to 192 bases -- four limbs per strand in the spare
  // half of Chain::revref, the forward strand's from the limb of its first window (offsets dstart[0] .. L - wl)
  bool ka_ok = false;
  int ka_lo = 0;
  if (P.Lpad <= 192 && ctx->o.known_absent >= 0 && !ctx->minz) {
    const int wlen = ctx->dict[0].end - ctx->dict[0].start + 1, lo = ctx->dict[0].start >> 5;
    if (wlen <= 32 && ctx->dict[1].start == ctx->dict[0].end + 1 && ctx->dict[1].end - ctx->dict[1].start + 1 == wlen &&
        2 * (ctx->L - wlen) + 2 <= 64 * (lo + 4) && 2 * ctx->dict[1].start + 2 <= 256 && lo + 4 <= 6) {
      ka_ok = true; ka_lo = lo;
    }
  }
  // Round 6, with the masks (chains stage, 25x pools, one chain / four chains per wavefront, two groups): 14 648 chains 82 / 79 ms,
  // 18 432 99 / 101, 22 528 114 / 110, 28 672 130 / 112; at 100x (1.09 reads per key: most probes find their key, the masks
  // save little) 108 / 129, 123 / 141, 131 / 139 -- so on very shallow dictionaries (< 1.05 reads per key) the four-chain
  // kernel takes over at 20 480 chains instead of 32 768 (profiles/r06_mc_threshold.txt)
  const uint64_t nd_ = (uint64_t)ctx->dict[0].numreads + ctx->dict[1].numreads, nk_ = (uint64_t)ctx->dict[0].numkeys + ctx->dict[1].numkeys;
  const uint32_t mc_min = P.phases == 2 ? ((ka_ok && nd_ * 100 < nk_ * 105) ? 20480u : 32768u) : 49152u;
  if (K < mc_min && ctx->o.fused != 3) P.mc = 0;   // (opts.fused = 3: four chains per wavefront whatever the count -- tests)
  if (64 - ctx->bshift > 32) P.mc = 0;            // (k_round_mc keeps bucket indices in 32 bits)
  P.ka = (P.mc && ka_ok) ? 1 : 0;
  P.ka_lo = P.ka ? ka_lo : 0;
  if (ctx->minz && !(fused && P.mc && !ctx->o.collect_stats && !P.deep_bins))
    return fail(SPRING_REORDER_E_ARG, "table_mode = 2 (minimizer-addressed table) is an experiment of the four-chain round kernel: "
                "shallow dictionary, at least 49152 chains or fused = 3, no work counters");
  if (!P.mc && !P.deep_bins && ctx->o.first_shifts == 0 && !ctx->user_plan0) {
    memset(P.plan[0], 0, sizeof(P.plan[0]));   // (fill_params assumed the four-chain kernel: 4 + 8 + 16)
    P.plan[0][0] = 8; P.plan[0][1] = 16;
  }
  P.g0 = 0; P.Kg = K; P.g0_other = 0; P.Kg_other = 0;
  P.gg0 = 0; P.gKg = Ktot; P.gKg_other = 0;
  P.seed_lo = 0; P.seed_hi = n;
  P.nb_lo = 0; P.nb_hi = (Ktot + 2047) / 2048;
  P.taken_other = nullptr; P.won = P.won_other = nullptr;
  ctx->taken2 = nullptr; ctx->resv2 = nullptr; ctx->won = nullptr;
  ctx->lb2 = spring_reorder_ctx::LongBufs();
  if (P.phases == 2) {
    DMALLOC(ctx->taken2, nwords * 8);
    DMALLOC(ctx->resv2, nn * 4);
    DMALLOC(ctx->won, (size_t)Ktot * 4);  // (every chain of the pool: the mark steps are replicated)
  }
  P.prop = nullptr; P.alive_wave = nullptr; P.needy_cnt = P.needy_cnt_next = nullptr;
  ctx->cnt_buf[0] = ctx->cnt_buf[1] = nullptr;
  if (fused && P.deep_bins && ctx->o.entry_flags >= 0 && n < 0x80000000u) {  // liveness inside the bins (DevParams::epos)
    for (int l = 0; l < 2; l++) {
      DMALLOC(P.epos[l], (size_t)std::max<uint32_t>(n, 1) * 4);
      HIPCHK(hipMemsetAsync(P.epos[l], 0xff, (size_t)std::max<uint32_t>(n, 1) * 4, st));
      launch_build_epos(st, ctx->dict[l].ids, ctx->dict[l].numreads, P.epos[l]);
    }
    P.idmask = 0x7fffffffu;
  }
  if (fused) {  // the rounds whose shared state k_mg_mark keeps: proposal words + double-buffered counters
    const size_t nblk = (size_t)Ktot / 2048 + 1;
    if (d_prop) P.prop = (unsigned long long *)d_prop;
    else DMALLOC(P.prop, (size_t)Ktot * 8);
    DMALLOC(ctx->cnt_buf[0], 2 * nblk * 4);
    ctx->cnt_buf[1] = ctx->cnt_buf[0] + nblk;
    DMALLOC(P.alive_wave, ((size_t)Ktot + 63) / 64 * 4);
    HIPCHK(hipMemsetAsync(ctx->cnt_buf[0], 0, 2 * nblk * 4, st));
    HIPCHK(hipMemsetAsync(P.alive_wave, 0, ((size_t)Ktot + 63) / 64 * 4, st));
    P.needy_cnt = ctx->cnt_buf[1];  // what the first round reads: nobody needs a seed yet
    const size_t nmark = ((size_t)Ktot + MARK_BLOCK - 1) / MARK_BLOCK;  // blocks of k_mg_mark = class-list segments
    DMALLOC(P.ord, nmark * MARK_BLOCK * 4);
    DMALLOC(P.ord_cnt, nmark * sizeof(uint4));
    if (P.deep_bins && P.long_budget > 0) {  // queue of the searches k_round hands to k_long
      DMALLOC(P.longq, ((size_t)K + 2) * 4);
      HIPCHK(hipMemsetAsync(P.longq, 0, 8, st));
      // what the three kernels of the long searches hand each other (LongHead, reorder_device.h): a head and a bin list per
      // chain (any chain may hand its search over in a round), one word per part
      P.lbin_stride = (uint32_t)std::min<int>(LONG_MAX_BINS, (4 * P.maxshift + 15) & ~15);
      DMALLOC(P.lctl, 64);
      HIPCHK(hipMemsetAsync(P.lctl, 0, 64, st));
      DMALLOC(P.lparts, (size_t)K * LONG_MAX_PARTS * 4);
      DMALLOC(P.lhead, (size_t)K * sizeof(LongHead));
      DMALLOC(P.lbin, (size_t)K * P.lbin_stride * sizeof(uint2));
      DMALLOC(P.lbcode, (size_t)K * P.lbin_stride * sizeof(uint16_t));
      if (P.phases == 2) {  // the second chain group's own set: its long-search kernels run beside the first group's
        const size_t K1 = (size_t)K - ctx->Kh;
        DMALLOC(ctx->lb2.longq, (K1 + 2) * 4);
        HIPCHK(hipMemsetAsync(ctx->lb2.longq, 0, 8, st));
        DMALLOC(ctx->lb2.lctl, 64);
        HIPCHK(hipMemsetAsync(ctx->lb2.lctl, 0, 64, st));
        DMALLOC(ctx->lb2.lparts, K1 * LONG_MAX_PARTS * 4);
        DMALLOC(ctx->lb2.lhead, K1 * sizeof(LongHead));
        DMALLOC(ctx->lb2.lbin, K1 * P.lbin_stride * sizeof(uint2));
        DMALLOC(ctx->lb2.lbcode, K1 * P.lbin_stride * sizeof(uint16_t));
      }
      for (int l = 0; l < 2; l++) {  // ... and the signatures k_long rejects most bin entries from (16 bytes per dictionary entry)
        ulonglong2 *sg = nullptr;
        const uint64_t m = ctx->dict[l].numreads;
        DMALLOC(sg, std::max<uint64_t>(m, 1) * sizeof(ulonglong2));
        launch_build_sig(st, ctx->dict[l].ids, m, ctx->d_reads, ctx->S, ctx->W, sg);
        P.sig[l] = sg;
      }
    }
  } else {
    P.ord = nullptr; P.ord_cnt = nullptr;
  }
#ifdef SR_PHASE_TIMING
  DMALLOC(P.dbg, 65 * sizeof(unsigned long long));
  HIPCHK(hipMemsetAsync(P.dbg, 0, 65 * sizeof(unsigned long long), ctx->st));
#endif
  HIPCHK(hipEventRecord(ctx->ev[4], st));
  launch_init_taken(st, P.taken, nwords, n, P.ublk);
  launch_fill_u32(st, P.resv, n, 0xffffffffu);
  HIPCHK(hipMemsetAsync(P.needy, 0, needy_bytes, st));
  HIPCHK(hipMemsetAsync(P.chains, 0, (size_t)K * sizeof(Chain), st));
  Globals g;
  memset(&g, 0, sizeof(g));
  g.cursor = (long long)n - 1;
  g.cursor_b = (long long)ctx->nmid - 1;
  g.e_alloc = g.s_alloc = K * CHUNK;
  g.alive = n == 0 ? 0 : (n / Ktot > 0 ? K : (c0 == 0 ? 1 : 0));  // chains that get a seed (reorder.h:405-421)
  HIPCHK(hipMemsetAsync(P.e_chunk, 0xff, nchunk * sizeof(uint2), st));  // owner 0xffffffff: chunk never handed out
  HIPCHK(hipMemsetAsync(P.s_chunk, 0xff, nchunk * sizeof(uint2), st));
  HIPCHK(hipMemcpyAsync(P.glob, &g, sizeof(g), hipMemcpyHostToDevice, st));
  if (P.phases == 2) {
    for (int g = 0; g < 2; g++) {  // (the chains of either slice: their global ids are c0 + li with the slice's own c0)
      DevParams Q = P;
      const auto &a = ctx->grp[g];
      Q.g0 = a.g0; Q.Kg = a.Kg; Q.c0 = a.c0; Q.gg0 = a.gg0; Q.gKg = a.gKg;
      launch_init_chains(st, Q, g == 0);
    }
    // both groups start from the same pool: the chains' first seeds are taken
    HIPCHK(hipMemcpyAsync(ctx->taken2, P.taken, nwords * 8, hipMemcpyDeviceToDevice, st));
    launch_fill_u32(st, ctx->resv2, n, 0xffffffffu);
    HIPCHK(hipMemsetAsync(ctx->won, 0xff, (size_t)Ktot * 4, st));
  } else launch_init_chains(st, P);
  HIPCHK(hipGetLastError());
  ctx->round_no = 0;
  return 0;
}

// pointers of the double-buffered counters for round t (reorder_device.h): k_round(t) ranks seeds with what
// k_mg_mark(t-1) counted; k_mg_mark(t) fills the other buffer and zeroes the one just read
static void set_round_buffers(spring_reorder_ctx *ctx) {
  DevParams &P = ctx->P;
  const int w = (int)(ctx->round_no & 1);
  P.needy_cnt = ctx->cnt_buf[w ^ 1];
  P.needy_cnt_next = ctx->cnt_buf[w];

}

// chains still running after the last k_mg_mark: per-wavefront counts summed here (the stream is synchronised)
static int running_chains(spring_reorder_ctx *ctx, std::vector<uint32_t> &buf, uint32_t *alive) {
  const size_t nw = ((size_t)ctx->P.Ktot + 63) / 64;
  buf.resize(nw);
  HIPCHK(hipMemcpyAsync(buf.data(), ctx->P.alive_wave, nw * 4, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  uint64_t a = 0;
  for (uint32_t v : buf) a += v;
  *alive = (uint32_t)std::min<uint64_t>(a, 0xffffffffull);
  return 0;
}

// DevParams of one chain group's launches in round `round_no` of that group (two-group schedule): the group's slice of this
// context's chains, its view of the pool, its reservation words, cursor, seed range, winners' list and long-search buffers
static DevParams group_params(spring_reorder_ctx *ctx, int g, uint64_t round_no) {
  DevParams Q = ctx->P;
  const auto &a = ctx->grp[g], &b = ctx->grp[g ^ 1];
  Q.g0 = a.g0; Q.Kg = a.Kg; Q.c0 = a.c0; Q.gg0 = a.gg0; Q.gKg = a.gKg;
  Q.g0_other = b.g0; Q.Kg_other = b.Kg; Q.gKg_other = b.gKg;
  Q.seed_lo = g ? 0u : ctx->nmid; Q.seed_hi = g ? ctx->nmid : ctx->n;
  Q.nb_lo = a.gg0 / 2048; Q.nb_hi = (a.gg0 + a.gKg + 2047) / 2048;
  Q.taken = g ? ctx->taken2 : ctx->P.taken; Q.taken_other = g ? ctx->P.taken : ctx->taken2;
  Q.resv = g ? ctx->resv2 : ctx->P.resv;
  Q.cursor = g ? &ctx->P.glob->cursor_b : &ctx->P.glob->cursor;
  Q.won = ctx->won + a.gg0; Q.won_other = ctx->won + b.gg0;
  if (g == 1 && ctx->P.longq) {
    Q.longq = ctx->lb2.longq; Q.lctl = ctx->lb2.lctl; Q.lparts = ctx->lb2.lparts; Q.lhead = ctx->lb2.lhead;
    Q.lbin = ctx->lb2.lbin; Q.lbcode = ctx->lb2.lbcode;
  }
  const int w = (int)(round_no & 1);  // (set_round_buffers, per group: the groups touch disjoint blocks of the buffers)
  Q.needy_cnt = ctx->cnt_buf[w ^ 1]; Q.needy_cnt_next = ctx->cnt_buf[w];
  return Q;
}

// The two-group schedule (DevParams::phases = 2).  Group 0 runs on the context's stream, group 1 on
// a second one; a round of a group = its round kernel, then its mark step, which waits (event) for the other group's
// last mark step: the mark steps strictly alternate A, B, A, B ..., the round kernels overlap.  A third stream carries the
// host's look at the running chains (one batch late, as in the one-group loop), so neither group's stream ever waits
// for the host.
static int run_chains_phased(spring_reorder_ctx *ctx, int R, bool timed) {
  DevParams &P = ctx->P;
  const uint32_t K = P.K;
  const bool stats = ctx->o.collect_stats != 0;
  if (!ctx->st2) HIPCHK(hipStreamCreateWithFlags(&ctx->st2, hipStreamNonBlocking));
  if (!ctx->st3) HIPCHK(hipStreamCreateWithFlags(&ctx->st3, hipStreamNonBlocking));
  hipStream_t sg[2] = {ctx->st, ctx->st2}, sc = ctx->st3;
  struct Group { uint64_t *taken; uint64_t round_no; } gp[2] = {{P.taken, 0}, {ctx->taken2, 0}};
  auto params_of = [&](int g) { return group_params(ctx, g, gp[g].round_no); };
  hipEvent_t ev[2] = {nullptr, nullptr}, bev[2] = {nullptr, nullptr}, ev0 = nullptr, evt = nullptr;
  struct EvFree { hipEvent_t *e; int n; ~EvFree() { for (int i = 0; i < n; i++) if (e[i]) (void)hipEventDestroy(e[i]); } };
  EvFree g1{ev, 2}, g2{bev, 2}, g3{&ev0, 1}, g4{&evt, 1};
  for (int i = 0; i < 2; i++) {
    HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&bev[i], hipEventDisableTiming));
  }
  HIPCHK(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
  std::vector<hipEvent_t> tev;  // opts.time_search: an event pair around every round-kernel launch, on the launch's stream
  struct TevFree { std::vector<hipEvent_t> &v; ~TevFree() { for (auto &e : v) (void)hipEventDestroy(e); } } tev_guard{tev};
  if (timed) {
    tev.resize(4 * (size_t)R);
    for (auto &e : tev) HIPCHK(hipEventCreate(&e));
  }
  const size_t nw = ((size_t)K + 63) / 64;
  uint32_t *h_aw = nullptr;
  HIPCHK(hipHostMalloc((void **)&h_aw, 2 * nw * sizeof(uint32_t), hipHostMallocDefault));
  struct HostFree { void *p; ~HostFree() { if (p) (void)hipHostFree(p); } } h_aw_guard{h_aw};
  // an error half way leaves work queued on all three streams: nothing of the run may be in flight when the events and the
  // pinned buffer above go (this guard is destroyed before them)
  struct SyncOnError {
    spring_reorder_ctx *c; bool ok;
    ~SyncOnError() { if (!ok) { (void)hipStreamSynchronize(c->st2); (void)hipStreamSynchronize(c->st3); (void)hipStreamSynchronize(c->st); } }
  } sync_guard{ctx, false};
  // group 1 starts behind the set-up (on the context's stream) and half a round late
  HIPCHK(hipEventRecord(ev0, sg[0]));
  HIPCHK(hipStreamWaitEvent(sg[1], ev0, 0));
  launch_delay(sg[1], 80);  // (the offset of the steady state sets itself within a few rounds whatever this is: 0 .. 160 us measured alike)
  bool have_b = false;
  uint64_t rounds = 0, launches = 0;
  double ms_search = 0;
  auto enqueue_batch = [&]() -> int {
    for (int r = 0; r < R; r++) {
      for (int g = 0; g < 2; g++) {
        const DevParams Q = params_of(g);
        if (timed) HIPCHK(hipEventRecord(tev[4 * r + 2 * g], sg[g]));
        launch_round(sg[g], Q, stats, false);
        if (timed) HIPCHK(hipEventRecord(tev[4 * r + 2 * g + 1], sg[g]));
        if (g == 1 || have_b) HIPCHK(hipStreamWaitEvent(sg[g], ev[g ^ 1], 0));  // the other group's last mark step
        launch_ph_mark(sg[g], Q);
        HIPCHK(hipEventRecord(ev[g], sg[g]));
        gp[g].round_no++;
      }
      have_b = true;
    }
    rounds += R;
    if (P.deep_bins && (ctx->dict[0].ndeep || ctx->dict[1].ndeep)) {
      // compaction of the deep bins (k_trim_bins moves bin entries: no search may run beside it): both groups meet here, every
      // R rounds -- the entries it drops are dead in both groups' views (their flags, or group 0's bitmap, the smaller one here)
      HIPCHK(hipStreamWaitEvent(sg[0], ev[1], 0));
      for (int l = 0; l < 2; l++)
        launch_trim_bins(sg[0], ctx->dict[l].deep, ctx->dict[l].d_ndeep, ctx->dict[l].ndeep, ctx->dict[l].urec, ctx->dict[l].ids, gp[0].taken, const_cast<ulonglong2 *>(P.sig[l]), P.epos[l]);
      HIPCHK(hipEventRecord(evt, sg[0]));
      HIPCHK(hipStreamWaitEvent(sg[1], evt, 0));
    }
    return 0;
  };
  auto count_batch = [&](int slot) -> int {  // behind both groups' last mark steps of the batch
    HIPCHK(hipStreamWaitEvent(sc, ev[0], 0));
    HIPCHK(hipStreamWaitEvent(sc, ev[1], 0));
    HIPCHK(hipMemcpyAsync(h_aw + (size_t)slot * nw, P.alive_wave, nw * 4, hipMemcpyDeviceToHost, sc));
    HIPCHK(hipEventRecord(bev[slot], sc));
    return 0;
  };
  double ms_busy = 0;
  auto collect_times = [&]() -> int {  // (time_search: the batch is complete)
    // summed launch durations, and the time during which at least one of the launches was running: their union (the two
    // groups' kernels run side by side, so the sum counts most of the batch twice)
    std::vector<std::pair<float, float>> iv(2 * (size_t)R);
    for (int i = 0; i < 2 * R; i++) {
      float ms = 0, s0 = 0;
      HIPCHK(hipEventElapsedTime(&ms, tev[2 * i], tev[2 * i + 1]));
      HIPCHK(hipEventElapsedTime(&s0, tev[0], tev[2 * i]));
      ms_search += ms;
      iv[i] = {s0, s0 + ms};
    }
    std::sort(iv.begin(), iv.end());
    float cs = iv[0].first, ce = iv[0].second;
    for (size_t i = 1; i < iv.size(); i++) {
      if (iv[i].first > ce) { ms_busy += ce - cs; cs = iv[i].first; ce = iv[i].second; }
      else ce = std::max(ce, iv[i].second);
    }
    ms_busy += ce - cs;
    launches += 2 * (uint64_t)R;
    return 0;
  };
  int rr;
  if (timed) {  // a batch at a time (the events are read between batches)
    for (;;) {
      if ((rr = enqueue_batch())) return rr;
      if ((rr = count_batch(0))) return rr;
      HIPCHK(hipEventSynchronize(bev[0]));
      HIPCHK(hipStreamSynchronize(sg[0]));
      HIPCHK(hipStreamSynchronize(sg[1]));
      if ((rr = collect_times())) return rr;
      uint64_t a = 0;
      for (size_t i = 0; i < nw; i++) a += h_aw[i];
      if (!a) break;
    }
  } else {
    if ((rr = enqueue_batch())) return rr;
    if ((rr = count_batch(0))) return rr;
    for (int b = 0;; b ^= 1) {
      if ((rr = enqueue_batch())) return rr;
      if ((rr = count_batch(b ^ 1))) return rr;
      HIPCHK(hipEventSynchronize(bev[b]));
      uint64_t a = 0;
      for (size_t i = 0; i < nw; i++) a += h_aw[(size_t)b * nw + i];
      HIPCHK(hipGetLastError());
      if (ctx->o.debug) fprintf(stderr, "[chains] rounds %llu running %llu (two groups)\n", (unsigned long long)(rounds - R), (unsigned long long)a);
      if (!a) break;
    }
  }
  // everything behind the chain phase is queued on the context's stream
  HIPCHK(hipStreamWaitEvent(sg[0], ev[1], 0));
  HIPCHK(hipEventRecord(ctx->ev[5], sg[0]));
  HIPCHK(hipStreamSynchronize(sg[1]));
  HIPCHK(hipStreamSynchronize(sc));
  HIPCHK(hipStreamSynchronize(sg[0]));
  ctx->round_no = gp[0].round_no;
  ctx->stats.rounds = rounds;
  ctx->stats.ms_search_kernel = ms_search;
  ctx->stats.ms_search_busy = ms_busy;
  ctx->stats.search_launches = launches;
  sync_guard.ok = true;
  return 0;
}

int spring_reorder_auto_chains(spring_reorder_ctx *ctx, uint32_t *chains, int32_t *deep) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage < ST_DICT) return fail(SPRING_REORDER_E_STATE, "auto_chains: build_dict first");
  const bool d = dict_is_deep(ctx);
  if (chains) *chains = auto_chains(ctx->n, d, dict_is_very_deep(ctx), dict_has_heavy_tail(ctx));
  if (deep) *deep = d ? 1 : 0;
  return 0;
}

int spring_reorder_run_chains(spring_reorder_ctx *ctx) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage != ST_DICT) return fail(SPRING_REORDER_E_STATE, "run_chains: build_dict first");
  HIPCHK(hipSetDevice(ctx->dev));
  hipStream_t st = ctx->st;
  const uint32_t n = ctx->n;
  const uint32_t K = ctx->o.num_chains ? ctx->o.num_chains : auto_chains(n, dict_is_deep(ctx), dict_is_very_deep(ctx), dict_has_heavy_tail(ctx));
  const bool stats = ctx->o.collect_stats != 0;
  const bool timed = ctx->o.time_search != 0;
  const bool literal = ctx->o.force_literal_update != 0;
  // one chain kernel per round (k_round + k_mg_mark) unless the literal consensus path or the two-kernel round is asked for
  const bool fused = !literal && ctx->o.fused >= 0;
  ctx->stats.chains = K;
  ctx->stats.deep_pool = (dict_is_deep(ctx) ? 1 : 0) | (dict_has_heavy_tail(ctx) ? 2 : 0);
  int r0 = setup_chains(ctx, K, 0, K, fused, nullptr, true);
  if (r0) return r0;
  DevParams &P = ctx->P;
  int R = ctx->o.rounds_per_sync > 0 ? ctx->o.rounds_per_sync : (K >= 256 ? 16 : 256);
  if (P.phases == 2) {
    int rp = run_chains_phased(ctx, R, timed);
    if (rp) return rp;
    ctx->stage = ST_CHAINS;
    return 0;
  }
  std::vector<hipEvent_t> tev;
  if (timed) {
    tev.resize(2 * (size_t)R);
    for (auto &e : tev) HIPCHK(hipEventCreate(&e));
  }
  uint32_t *h_alive = nullptr;
  std::vector<uint32_t> alive_tmp;
  HIPCHK(hipHostMalloc((void **)&h_alive, sizeof(uint32_t), hipHostMallocDefault));
  struct HostFree { void *p; ~HostFree() { if (p) (void)hipHostFree(p); } } h_alive_guard{h_alive};  // (every early return below)
  uint64_t rounds = 0;
  double ms_search = 0;
  uint64_t launches = 0;
  HIPCHK(hipMemcpyAsync(h_alive, &P.glob->alive, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  auto enqueue_batch = [&]() -> int {
    for (int r = 0; r < R; r++) {
      if (timed) HIPCHK(hipEventRecord(tev[2 * r], st));
      if (fused) {
        set_round_buffers(ctx);
        launch_round(st, P, stats, false);
        if (timed) HIPCHK(hipEventRecord(tev[2 * r + 1], st));
        launch_mg_mark(st, P);
        ctx->round_no++;
      } else {
        launch_search(st, P, stats);
        if (timed) HIPCHK(hipEventRecord(tev[2 * r + 1], st));
        launch_apply(st, P, literal);
      }
    }
    rounds += R;
    for (int l = 0; l < 2; l++)  // shrink deep bins whose tail has been consumed (exact; see k_trim_bins)
      launch_trim_bins(st, ctx->dict[l].deep, ctx->dict[l].d_ndeep, ctx->dict[l].ndeep, ctx->dict[l].urec, ctx->dict[l].ids, P.taken, const_cast<ulonglong2 *>(P.sig[l]), P.epos[l]);
    return 0;
  };
  if (fused && !timed && *h_alive) {
    // The host looks at the chains still running one batch LATE: batch b + 1 is in the queue before the count of batch b is
    // waited for, so the GPU never drains while the host synchronises and launches (a bubble of some tens of microseconds
    // every R rounds).  The price: up to 2 R - 1 rounds over finished chains at the end instead of R - 1 (a round in which
    // every chain is done changes nothing and costs its launches).
    const size_t nw = ((size_t)P.Ktot + 63) / 64;
    uint32_t *h_aw = nullptr;
    HIPCHK(hipHostMalloc((void **)&h_aw, 2 * nw * sizeof(uint32_t), hipHostMallocDefault));
    HostFree h_aw_guard{h_aw};
    hipEvent_t bev[2] = {nullptr, nullptr};
    struct EvFree { hipEvent_t *e; ~EvFree() { for (int i = 0; i < 2; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } bev_guard{bev};
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&bev[i], hipEventDisableTiming));
    auto count_batch = [&](int slot) -> int {  // (queued behind the batch: the per-64-chain counts k_mg_mark keeps)
      HIPCHK(hipMemcpyAsync(h_aw + (size_t)slot * nw, P.alive_wave, nw * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipEventRecord(bev[slot], st));
      return 0;
    };
    int rr;
    if ((rr = enqueue_batch())) return rr;
    if ((rr = count_batch(0))) return rr;
    for (int b = 0;; b ^= 1) {
      if ((rr = enqueue_batch())) return rr;
      if ((rr = count_batch(b ^ 1))) return rr;
      HIPCHK(hipEventSynchronize(bev[b]));
      uint64_t a = 0;
      for (size_t i = 0; i < nw; i++) a += h_aw[(size_t)b * nw + i];
      HIPCHK(hipGetLastError());
      if (ctx->o.debug) fprintf(stderr, "[chains] rounds %llu running %llu\n", (unsigned long long)(rounds - R), (unsigned long long)a);
      if (!a) break;
    }
    *h_alive = 0;
  }
  while (*h_alive) {
    int rr = enqueue_batch();
    if (rr) return rr;
    // chains still running: the two-kernel round keeps the count, the fused round recounts it every round
    if (fused) {
      int ra = running_chains(ctx, alive_tmp, h_alive);
      if (ra) return ra;
    } else {
      HIPCHK(hipMemcpyAsync(h_alive, &P.glob->alive, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    HIPCHK(hipGetLastError());
    if (ctx->o.debug) fprintf(stderr, "[chains] rounds %llu running %u\n", (unsigned long long)rounds, *h_alive);
    if (timed) {
      for (int r = 0; r < R; r++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, tev[2 * r], tev[2 * r + 1]));
        ms_search += ms;
      }
      launches += R;
    }
  }
  HIPCHK(hipEventRecord(ctx->ev[5], st));
  HIPCHK(hipStreamSynchronize(st));
  for (auto &e : tev) (void)hipEventDestroy(e);
  ctx->stats.rounds = rounds;
  ctx->stats.ms_search_kernel = ms_search;
  ctx->stats.ms_search_busy = ms_search;
  ctx->stats.search_launches = launches;
  ctx->stage = ST_CHAINS;
  return 0;
}

// ------------------------------------------------- single-pool multi-GPU
// Every rank holds the full read pool and the dictionary table; rank r owns chains
// [r*K/world, (r+1)*K/world).  One round = mg_search (k_round over the own chains: apply of the last
// proposals + search) -> all-gather of the per-chain proposal words (mg_run: inside the library; the
// step-wise API: by the caller, or spring_reorder_mg_exchange_virtual between contexts of one process) ->
// mg_apply (k_mg_resolve + k_mg_mark over all chains, on every rank identically).
// Output is bit-identical to run_chains() with num_chains = total_chains on one GPU.
int spring_reorder_mg_begin(spring_reorder_ctx *ctx, uint32_t rank, uint32_t world, uint32_t total_chains,
                            void *d_prop) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage != ST_DICT) return fail(SPRING_REORDER_E_STATE, "mg_begin: build_dict first");
  if (world == 0 || rank >= world || total_chains == 0 || total_chains % world)
    return fail(SPRING_REORDER_E_ARG, "mg_begin: total_chains (%u) must be a positive multiple of world (%u)", total_chains, world);
  if (ctx->o.force_literal_update) return fail(SPRING_REORDER_E_ARG, "mg_begin: the literal consensus path only exists in the two-kernel round");
  HIPCHK(hipSetDevice(ctx->dev));
  const uint32_t K = total_chains / world;
  ctx->stats.chains = total_chains;
  ctx->stats.deep_pool = (dict_is_deep(ctx) ? 1 : 0) | (dict_has_heavy_tail(ctx) ? 2 : 0);
  int r = setup_chains(ctx, K, rank * K, total_chains, true, d_prop, true);
  if (r) return r;
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->mg = true;
  ctx->stats.rounds = 0;
  return 0;
}

int spring_reorder_mg_search(spring_reorder_ctx *ctx) {
  if (!ctx || !ctx->mg || ctx->stage != ST_DICT) return fail(SPRING_REORDER_E_STATE, "mg_search: mg_begin first");
  HIPCHK(hipSetDevice(ctx->dev));
  if (ctx->P.phases == 2) {
    // two chain groups, step by step: both groups' round kernels, the caller's exchange of both slices, then the mark steps
    // A, B (mg_apply) -- one of the orders the overlapped schedule of mg_run may take (a group's round kernel needs its own
    // last mark step only)
    for (int g = 0; g < 2; g++) launch_round(ctx->st, group_params(ctx, g, ctx->round_no), ctx->o.collect_stats != 0, true);
  } else {
    set_round_buffers(ctx);
    launch_round(ctx->st, ctx->P, ctx->o.collect_stats != 0, true);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->st));  // the caller's exchange reads this rank's slice next
  return 0;
}

int spring_reorder_mg_slice(spring_reorder_ctx *ctx, void **d_prop, size_t *slice_off, size_t *slice_bytes,
                            size_t *total_bytes) {
  if (!ctx || !ctx->mg) return fail(SPRING_REORDER_E_STATE, "mg_slice: mg_begin first");
  if (ctx->P.phases == 2) return fail(SPRING_REORDER_E_STATE, "mg_slice: with two chain groups a rank owns two slices of the proposal words (mg_run / exchange_virtual move both)");
  if (d_prop) *d_prop = ctx->P.prop;
  if (slice_off) *slice_off = (size_t)ctx->P.c0 * 8;
  if (slice_bytes) *slice_bytes = (size_t)ctx->P.K * 8;
  if (total_bytes) *total_bytes = (size_t)ctx->P.Ktot * 8;
  return 0;
}

int spring_reorder_mg_apply(spring_reorder_ctx *ctx, int32_t check_alive, uint32_t *alive) {
  if (!ctx || !ctx->mg || ctx->stage != ST_DICT) return fail(SPRING_REORDER_E_STATE, "mg_apply: mg_begin first");
  HIPCHK(hipSetDevice(ctx->dev));
  hipStream_t st = ctx->st;
  if (ctx->P.phases == 2) {
    for (int g = 0; g < 2; g++) {
      const DevParams Q = group_params(ctx, g, ctx->round_no);
      launch_mg_resolve(st, Q);
      launch_ph_mark(st, Q);
    }
  } else {
    launch_mg_resolve(st, ctx->P);
    launch_mg_mark(st, ctx->P);
  }
  HIPCHK(hipGetLastError());
  ctx->round_no++;
  ctx->stats.rounds++;
  if (ctx->stats.rounds % 16 == 0)
    for (int l = 0; l < 2; l++)
      launch_trim_bins(st, ctx->dict[l].deep, ctx->dict[l].d_ndeep, ctx->dict[l].ndeep, ctx->dict[l].urec, ctx->dict[l].ids, ctx->P.taken, const_cast<ulonglong2 *>(ctx->P.sig[l]), ctx->P.epos[l]);
  if (check_alive) {
    uint32_t a = 0;
    std::vector<uint32_t> tmp;
    int ra = running_chains(ctx, tmp, &a);
    if (ra) return ra;
    if (alive) *alive = a;
  }
  return 0;
}

int spring_reorder_mg_end(spring_reorder_ctx *ctx) {
  if (!ctx || !ctx->mg || ctx->stage != ST_DICT) return fail(SPRING_REORDER_E_STATE, "mg_end: mg_begin first");
  HIPCHK(hipSetDevice(ctx->dev));
  HIPCHK(hipEventRecord(ctx->ev[5], ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->stage = ST_CHAINS;
  return 0;
}

// test hook: the invariants the seed pick relies on, checked between two rounds of the step-wise API (after mg_apply)
int spring_reorder_debug_check_seed_state(spring_reorder_ctx *ctx, uint64_t *violations /* [2] */) {
  if (!ctx || !ctx->mg || ctx->stage != ST_DICT || !violations) return fail(SPRING_REORDER_E_STATE, "debug_check_seed_state: between mg_begin and mg_end");
  HIPCHK(hipSetDevice(ctx->dev));
  unsigned long long *d_bad = nullptr;
  DMALLOC(d_bad, 16);
  HIPCHK(hipMemsetAsync(d_bad, 0, 16, ctx->st));
  const uint64_t nblk = ((uint64_t)ctx->n + (1ull << UBLK_SHIFT) - 1) >> UBLK_SHIFT;
  launch_check_seed_state(ctx->st, ctx->P, nblk << (UBLK_SHIFT - 6), d_bad);
  unsigned long long h[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h, d_bad, 16, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->dfree(d_bad);
  violations[0] = h[0]; violations[1] = h[1];
  return 0;
}

// in-process all-gather between `world` contexts of one device (tests of G-independence on one GPU)
int spring_reorder_mg_exchange_virtual(spring_reorder_ctx **ctxs, uint32_t world) {
  if (!ctxs || !world) return fail(SPRING_REORDER_E_ARG, "bad arguments");
  for (uint32_t d = 0; d < world; d++) {
    spring_reorder_ctx *dc = ctxs[d];
    if (!dc || !dc->mg) return fail(SPRING_REORDER_E_STATE, "exchange_virtual: mg_begin first");
    HIPCHK(hipSetDevice(dc->dev));
    for (uint32_t s2 = 0; s2 < world; s2++) {
      if (s2 == d) continue;
      spring_reorder_ctx *sc = ctxs[s2];
      if (sc->P.prop == dc->P.prop) continue;  // shared buffer
      for (int g = 0; g < 2; g++) {  // (one slice, or one per chain group)
        const auto &a = sc->grp[g];
        if (!a.Kg) continue;
        const size_t o = (size_t)a.c0 + a.g0;  // the slice's first global chain id
        HIPCHK(hipMemcpyAsync(dc->P.prop + o, sc->P.prop + o, (size_t)a.Kg * 8, hipMemcpyDeviceToDevice, dc->st));
      }
    }
  }
  for (uint32_t d = 0; d < world; d++) HIPCHK(hipStreamSynchronize(ctxs[d]->st));
  return 0;
}

int spring_mg_rccl_unique_id(void *id128) {
  if (!id128) return fail(SPRING_REORDER_E_ARG, "id buffer is NULL");
  int r = rccl_load();
  if (r) return r;
  const int e = g_rccl.GetUniqueId(id128);
  if (e) return fail(SPRING_REORDER_E_HIP, "ncclGetUniqueId: %s", rccl_err(e));
  return 0;
}

int spring_mg_comm_create_rccl(spring_mg_comm **out, int32_t device, const void *id128, uint32_t rank, uint32_t world) {
  if (!out || !id128 || !world || rank >= world) return fail(SPRING_REORDER_E_ARG, "comm_create_rccl: bad arguments");
  int r = rccl_load();
  if (r) return r;
  int dev = device;
  if (dev < 0) HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipSetDevice(dev));
  RcclId128 id;
  memcpy(id.b, id128, sizeof(id.b));
  void *comm = nullptr;
  const int e = g_rccl.CommInitRank(&comm, (int)world, id, (int)rank);
  if (e) return fail(SPRING_REORDER_E_HIP, "ncclCommInitRank: %s", rccl_err(e));
  spring_mg_comm *c = new spring_mg_comm();
  c->rank = rank; c->world = world; c->dev = dev; c->rccl = comm;
  *out = c;
  return 0;
}

int spring_mg_comm_create_host(spring_mg_comm **out, spring_mg_allgather_fn fn, void *user, uint32_t rank, uint32_t world) {
  if (!out || !fn || !world || rank >= world) return fail(SPRING_REORDER_E_ARG, "comm_create_host: bad arguments");
  spring_mg_comm *c = new spring_mg_comm();
  c->rank = rank; c->world = world; c->dev = -1; c->host_fn = fn; c->host_user = user;
  *out = c;
  return 0;
}

void spring_mg_comm_destroy(spring_mg_comm *c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->rccl && !c->aborted && g_rccl.CommDestroy) {
      (void)hipSetDevice(c->dev);
      (void)g_rccl.CommDestroy(c->rccl);
    }
    c->rccl = nullptr;
  }
  delete c;
}
}  // extern "C"
// A rank of an in-process pool failed: its peers sit in (or are about to enter) a collective that will never complete.
// ncclCommAbort ends it -- their stream operations then fail and their threads come back with an error instead of hanging.
extern "C++" {
namespace sr {
void mg_comm_abort(spring_mg_comm *c) {
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->rccl && !c->aborted && g_rccl.CommAbort) {
    (void)g_rccl.CommAbort(c->rccl);
    c->aborted = true;
  }
}
}  // namespace sr
}
extern "C" {

// The pool with two chain groups (DevParams::phases = 2): every rank owns a slice of EACH group, so the
// groups -- and with them the output -- are the same whatever the number of ranks.  Per group and round: the round kernel over
// the rank's slice on the group's stream -> the all-gather of the group's proposal words (in place inside the group's range of
// the buffer; all collectives of the run go through ONE exchange stream in one order, the same on every rank) -> the other
// ranks' words resolved, the group's mark step over the whole group (replicated), which waits for the other group's last mark
// step as on one GPU.  So a group's exchange, resolve and mark run beside the other group's round kernel: nothing but round
// kernels is left on a rank's critical path.
static int mg_run_phased(spring_reorder_ctx *ctx, spring_mg_comm *comm) {
  DevParams &P = ctx->P;
  const bool stats = ctx->o.collect_stats != 0;
  const int R = ctx->o.rounds_per_sync > 0 ? ctx->o.rounds_per_sync : 8;
  const bool timed = ctx->o.time_search != 0;
  if (!ctx->st2) HIPCHK(hipStreamCreateWithFlags(&ctx->st2, hipStreamNonBlocking));
  if (!ctx->st3) HIPCHK(hipStreamCreateWithFlags(&ctx->st3, hipStreamNonBlocking));
  hipStream_t sg[2] = {ctx->st, ctx->st2}, sx = ctx->st3;
  hipEvent_t ev[2] = {nullptr, nullptr}, ek[2] = {nullptr, nullptr}, ex[2] = {nullptr, nullptr}, bev[2] = {nullptr, nullptr}, ev0 = nullptr;
  struct EvFree { hipEvent_t *e; int n; ~EvFree() { for (int i = 0; i < n; i++) if (e[i]) (void)hipEventDestroy(e[i]); } };
  EvFree f1{ev, 2}, f2{ek, 2}, f3{ex, 2}, f4{bev, 2}, f5{&ev0, 1};
  for (int i = 0; i < 2; i++) {
    HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ek[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ex[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&bev[i], hipEventDisableTiming));
  }
  HIPCHK(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
  std::vector<hipEvent_t> tev;  // time_search: per round and group {round kernel start, end, exchange end (on sx), mark end}
  struct TevFree { std::vector<hipEvent_t> &v; ~TevFree() { for (auto &e : v) if (e) (void)hipEventDestroy(e); } } tev_guard{tev};
  if (timed) {
    tev.assign(8 * (size_t)R, nullptr);
    for (auto &e : tev) HIPCHK(hipEventCreate(&e));
  }
  const size_t nw = ((size_t)P.Ktot + 63) / 64;
  uint32_t *h_aw = nullptr;
  HIPCHK(hipHostMalloc((void **)&h_aw, 2 * nw * sizeof(uint32_t), hipHostMallocDefault));
  struct HostFree { void *p; ~HostFree() { if (p) (void)hipHostFree(p); } } h_aw_guard{h_aw};
  void *h_stage = nullptr;  // host transport: one staging buffer per group's range
  const size_t gbytes[2] = {(size_t)ctx->grp[0].gKg * 8, (size_t)ctx->grp[1].gKg * 8};
  if (comm->host_fn) HIPCHK(hipHostMalloc(&h_stage, std::max(gbytes[0], gbytes[1]), hipHostMallocDefault));
  HostFree h_stage_guard{h_stage};
  uint64_t round_no[2] = {0, 0};
  HIPCHK(hipEventRecord(ev0, sg[0]));
  HIPCHK(hipStreamWaitEvent(sg[1], ev0, 0));
  HIPCHK(hipStreamWaitEvent(sx, ev0, 0));
  bool have_b = false;
  uint64_t rounds = 0, timed_rounds = 0;
  double ms_round = 0, ms_xchg = 0, ms_mark = 0, ms_busy = 0;
  auto fail_sync = [&](int code) {  // (nothing of this run may still be queued when the caller tears the context down)
    (void)hipStreamSynchronize(sg[0]); (void)hipStreamSynchronize(sg[1]); (void)hipStreamSynchronize(sx);
    return code;
  };
  auto enqueue_batch = [&]() -> int {
    for (int r = 0; r < R; r++) {
      for (int g = 0; g < 2; g++) {
        const DevParams Q = group_params(ctx, g, round_no[g]);
        const auto &a = ctx->grp[g];
        unsigned long long *grange = P.prop + a.gg0;             // the group's words, all ranks
        unsigned long long *mine = P.prop + a.c0 + a.g0;         // this rank's slice of them
        if (timed) HIPCHK(hipEventRecord(tev[8 * r + 4 * g], sg[g]));
        launch_round(sg[g], Q, stats, true);
        if (timed) HIPCHK(hipEventRecord(tev[8 * r + 4 * g + 1], sg[g]));
        HIPCHK(hipEventRecord(ek[g], sg[g]));
        if (comm->rccl) {
          HIPCHK(hipStreamWaitEvent(sx, ek[g], 0));
          const int e = g_rccl.AllGather(mine, grange, a.Kg, RCCL_UINT64, comm->rccl, sx);
          if (e) return fail(SPRING_REORDER_E_HIP, "ncclAllGather: %s", rccl_err(e));
          if (timed) HIPCHK(hipEventRecord(tev[8 * r + 4 * g + 2], sx));
          HIPCHK(hipEventRecord(ex[g], sx));
          HIPCHK(hipStreamWaitEvent(sg[g], ex[g], 0));
        } else {  // through the caller's all-gather on host memory (ranks that share a device; tests): synchronous
          const size_t off = (size_t)(a.c0 + a.g0 - a.gg0) * 8, slice = (size_t)a.Kg * 8;
          HIPCHK(hipMemcpyAsync((char *)h_stage + off, mine, slice, hipMemcpyDeviceToHost, sg[g]));
          HIPCHK(hipStreamSynchronize(sg[g]));
          if (comm->host_fn(h_stage, off, slice, gbytes[g], comm->host_user))
            return fail(SPRING_REORDER_E_IO, "mg_run: the caller's all-gather reported an error");
          HIPCHK(hipMemcpyAsync(grange, h_stage, gbytes[g], hipMemcpyHostToDevice, sg[g]));
          HIPCHK(hipStreamSynchronize(sg[g]));  // (the staging buffer is reused by the other group)
          if (timed) HIPCHK(hipEventRecord(tev[8 * r + 4 * g + 2], sg[g]));
        }
        launch_mg_resolve(sg[g], Q);
        if (g == 1 || have_b) HIPCHK(hipStreamWaitEvent(sg[g], ev[g ^ 1], 0));  // the other group's last mark step
        launch_ph_mark(sg[g], Q);
        if (timed) HIPCHK(hipEventRecord(tev[8 * r + 4 * g + 3], sg[g]));
        HIPCHK(hipEventRecord(ev[g], sg[g]));
        round_no[g]++;
      }
      have_b = true;
    }
    rounds += R;
    if (P.deep_bins && (ctx->dict[0].ndeep || ctx->dict[1].ndeep)) {  // (as in run_chains_phased: both groups meet)
      HIPCHK(hipStreamWaitEvent(sg[0], ev[1], 0));
      for (int l = 0; l < 2; l++)
        launch_trim_bins(sg[0], ctx->dict[l].deep, ctx->dict[l].d_ndeep, ctx->dict[l].ndeep, ctx->dict[l].urec, ctx->dict[l].ids, P.taken, const_cast<ulonglong2 *>(P.sig[l]), P.epos[l]);
      HIPCHK(hipEventRecord(ev0, sg[0]));
      HIPCHK(hipStreamWaitEvent(sg[1], ev0, 0));
    }
    return 0;
  };
  // the running chains, counted from the mark steps' per-wavefront counts behind both groups' last mark steps of a batch:
  // every rank marks every chain from the same gathered words, so every rank reads the same count and stops after the same batch
  auto count_batch = [&](int slot) -> int {
    HIPCHK(hipStreamWaitEvent(sx, ev[0], 0));
    HIPCHK(hipStreamWaitEvent(sx, ev[1], 0));
    HIPCHK(hipMemcpyAsync(h_aw + (size_t)slot * nw, P.alive_wave, nw * 4, hipMemcpyDeviceToHost, sx));
    HIPCHK(hipEventRecord(bev[slot], sx));
    return 0;
  };
  auto alive_in = [&](int slot) { uint64_t a = 0; for (size_t i = 0; i < nw; i++) a += h_aw[(size_t)slot * nw + i]; return a; };
  int rr = 0;
  if (timed || comm->host_fn) {  // a batch at a time
    for (;;) {
      if ((rr = enqueue_batch())) return fail_sync(rr);
      if ((rr = count_batch(0))) return fail_sync(rr);
      HIPCHK(hipEventSynchronize(bev[0]));
      HIPCHK(hipStreamSynchronize(sg[0]));
      HIPCHK(hipStreamSynchronize(sg[1]));
      if (timed) {
        // per group-round: round kernel | all-gather (from the end of the round kernel to the end of the collective on the exchange
        // stream) | resolve + mark (incl. the wait for the other group's mark step); and the union of the round kernels'
        // intervals: the time during which at least one round kernel ran -- the rank's critical path if everything else hides
        std::vector<std::pair<float, float>> iv(2 * (size_t)R);
        for (int i = 0; i < 2 * R; i++) {
          float a = 0, b = 0, c = 0, s0 = 0;
          (void)hipEventElapsedTime(&a, tev[4 * i], tev[4 * i + 1]);
          (void)hipEventElapsedTime(&b, tev[4 * i + 1], tev[4 * i + 2]);
          (void)hipEventElapsedTime(&c, tev[4 * i + 2], tev[4 * i + 3]);
          (void)hipEventElapsedTime(&s0, tev[0], tev[4 * i]);
          ms_round += a; ms_xchg += b; ms_mark += c;
          iv[(size_t)i] = {s0, s0 + a};
        }
        std::sort(iv.begin(), iv.end());
        float cs = iv[0].first, ce = iv[0].second;
        for (size_t i = 1; i < iv.size(); i++) {
          if (iv[i].first > ce) { ms_busy += ce - cs; cs = iv[i].first; ce = iv[i].second; }
          else ce = std::max(ce, iv[i].second);
        }
        ms_busy += ce - cs;
        timed_rounds += (uint64_t)R;
      }
      if (!alive_in(0)) break;
    }
  } else {  // the host looks at the count one batch late: neither group's stream ever waits for it
    if ((rr = enqueue_batch())) return fail_sync(rr);
    if ((rr = count_batch(0))) return fail_sync(rr);
    for (int b = 0;; b ^= 1) {
      if ((rr = enqueue_batch())) return fail_sync(rr);
      if ((rr = count_batch(b ^ 1))) return fail_sync(rr);
      HIPCHK(hipEventSynchronize(bev[b]));
      const uint64_t a = alive_in(b);
      HIPCHK(hipGetLastError());
      if (ctx->o.debug) fprintf(stderr, "[chains] rounds %llu running %llu (pool, two groups)\n", (unsigned long long)(rounds - R), (unsigned long long)a);
      if (!a) break;
    }
  }
  HIPCHK(hipStreamWaitEvent(sg[0], ev[1], 0));
  HIPCHK(hipStreamSynchronize(sg[1]));
  HIPCHK(hipStreamSynchronize(sx));
  HIPCHK(hipStreamSynchronize(sg[0]));
  ctx->round_no = round_no[0];
  ctx->stats.rounds = rounds;
  if (timed) {  // (per group-round: the pieces of both groups are summed; two round kernels run side by side most of the time)
    ctx->stats.ms_search_kernel = ms_round; ctx->stats.ms_search_busy = ms_busy; ctx->stats.search_launches = 2 * timed_rounds;
    ctx->stats.ms_exchange = ms_xchg; ctx->stats.ms_resolve_mark = ms_mark;
  }
  return 0;
}

// one pool over the communicator's ranks with the exchange inside the library; see include/spring_reorder.h
int spring_reorder_mg_run(spring_reorder_ctx *ctx, spring_mg_comm *comm, uint32_t total_chains) {
  if (!ctx || !comm) return fail(SPRING_REORDER_E_ARG, "mg_run: NULL argument");
  if (comm->rccl && comm->dev != ctx->dev)
    return fail(SPRING_REORDER_E_ARG, "mg_run: the communicator lives on device %d, the context on %d", comm->dev, ctx->dev);
  int r = spring_reorder_mg_begin(ctx, comm->rank, comm->world, total_chains, nullptr);
  if (r) return r;
  if (ctx->P.phases == 2) {  // two chain groups: a slice of each per rank, the exchanges beside the round kernels
    const int rp = mg_run_phased(ctx, comm);
    const std::string keep = g_err;
    const int re = spring_reorder_mg_end(ctx);
    if (rp) { g_err = keep; return rp; }
    return re;
  }
  hipStream_t st = ctx->st;
  DevParams &P = ctx->P;
  const bool stats = ctx->o.collect_stats != 0;
  const int R = ctx->o.rounds_per_sync > 0 ? ctx->o.rounds_per_sync : 8;
  const size_t slice = (size_t)P.K * 8, total = (size_t)P.Ktot * 8;
  uint32_t *h_alive = nullptr;
  std::vector<uint32_t> alive_tmp;
  void *h_stage = nullptr;
  HIPCHK(hipHostMalloc((void **)&h_alive, sizeof(uint32_t), hipHostMallocDefault));
  if (comm->host_fn && hipHostMalloc(&h_stage, total, hipHostMallocDefault) != hipSuccess) {
    (void)hipHostFree(h_alive);
    (void)spring_reorder_mg_end(ctx);
    return fail(SPRING_REORDER_E_HIP, "mg_run: cannot pin the exchange staging buffer");
  }
  int ret = 0;
  *h_alive = 1;
  // time_search: four events per round (round kernel | exchange | resolve + mark), read back at every host check
  const bool timed = ctx->o.time_search != 0;
  std::vector<hipEvent_t> tev;
  if (timed) {
    tev.resize(4 * (size_t)R);
    for (auto &e : tev) if (hipEventCreate(&e) != hipSuccess) { ret = fail(SPRING_REORDER_E_HIP, "mg_run: hipEventCreate failed"); break; }
  }
  double ms_round = 0, ms_xchg = 0, ms_mark = 0;
  uint64_t timed_rounds = 0;
  while (*h_alive && !ret) {
    int done_rounds = 0;
    for (int i = 0; i < R && !ret; i++) {
      set_round_buffers(ctx);
      if (timed) (void)hipEventRecord(tev[4 * i], st);
      launch_round(st, P, stats, true);
      if (timed) (void)hipEventRecord(tev[4 * i + 1], st);
      if (comm->rccl) {  // in place: this rank's words already sit at their offset of the receive buffer
        const int e = g_rccl.AllGather(P.prop + P.c0, P.prop, P.K, RCCL_UINT64, comm->rccl, st);
        if (e) ret = fail(SPRING_REORDER_E_HIP, "ncclAllGather: %s", rccl_err(e));
      } else {
        hipError_t he = hipMemcpyAsync((char *)h_stage + (size_t)P.c0 * 8, P.prop + P.c0, slice, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he != hipSuccess) { ret = fail(SPRING_REORDER_E_HIP, "staging copy failed: %s", hipGetErrorString(he)); break; }
        if (comm->host_fn(h_stage, (size_t)P.c0 * 8, slice, total, comm->host_user)) {
          ret = fail(SPRING_REORDER_E_IO, "mg_run: the caller's all-gather reported an error");
          break;
        }
        he = hipMemcpyAsync(P.prop, h_stage, total, hipMemcpyHostToDevice, st);
        if (he != hipSuccess) { ret = fail(SPRING_REORDER_E_HIP, "staging copy failed: %s", hipGetErrorString(he)); break; }
      }
      if (timed) (void)hipEventRecord(tev[4 * i + 2], st);
      launch_mg_resolve(st, P);
      launch_mg_mark(st, P);
      if (timed) (void)hipEventRecord(tev[4 * i + 3], st);
      done_rounds = i + 1;
      ctx->round_no++;
      ctx->stats.rounds++;
      if (ctx->stats.rounds % 16 == 0)
        for (int l = 0; l < 2; l++)
          launch_trim_bins(st, ctx->dict[l].deep, ctx->dict[l].d_ndeep, ctx->dict[l].ndeep, ctx->dict[l].urec, ctx->dict[l].ids, P.taken, const_cast<ulonglong2 *>(P.sig[l]), P.epos[l]);
    }
    if (ret) break;
    if (timed) {
      (void)hipStreamSynchronize(st);
      for (int i = 0; i < done_rounds; i++) {
        float a = 0, b = 0, c = 0;
        (void)hipEventElapsedTime(&a, tev[4 * i], tev[4 * i + 1]);
        (void)hipEventElapsedTime(&b, tev[4 * i + 1], tev[4 * i + 2]);
        (void)hipEventElapsedTime(&c, tev[4 * i + 2], tev[4 * i + 3]);
        ms_round += a; ms_xchg += b; ms_mark += c;
      }
      timed_rounds += (uint64_t)done_rounds;
    }
    // every rank recounted the running chains from the same gathered words: the same decision everywhere
    ret = running_chains(ctx, alive_tmp, h_alive);
    if (!ret) {
      hipError_t he = hipGetLastError();
      if (he != hipSuccess) ret = fail(SPRING_REORDER_E_HIP, "mg_run: %s", hipGetErrorString(he));
    }
  }
  (void)hipHostFree(h_alive);
  if (h_stage) (void)hipHostFree(h_stage);
  for (auto &e : tev) if (e) (void)hipEventDestroy(e);
  if (timed) {
    ctx->stats.ms_search_kernel = ms_round; ctx->stats.ms_search_busy = ms_round; ctx->stats.search_launches = timed_rounds;
    ctx->stats.ms_exchange = ms_xchg; ctx->stats.ms_resolve_mark = ms_mark;
  }
  if (ret) {
    // a failed rank leaves the pool: its peers are stuck in the exchange until the caller tears the communicator
    // down (RCCL) or makes its all-gather callback fail (host transport) -- the context itself is left consistent
    const std::string keep = g_err;
    (void)hipStreamSynchronize(st);
    (void)spring_reorder_mg_end(ctx);
    g_err = keep;
    return ret;
  }
  return spring_reorder_mg_end(ctx);
}

int spring_reorder_finalize(spring_reorder_ctx *ctx) {
  if (!ctx) return fail(SPRING_REORDER_E_ARG, "ctx is NULL");
  if (ctx->stage != ST_CHAINS) return fail(SPRING_REORDER_E_STATE, "finalize: run_chains first");
  HIPCHK(hipSetDevice(ctx->dev));
  hipStream_t st = ctx->st;
  DevParams &P = ctx->P;
  const uint32_t K = ctx->K;
  const int T = ctx->o.num_thr;
  HIPCHK(hipEventRecord(ctx->ev[6], st));
  // per chain only {records, singletons} come back (8 bytes instead of the 384-byte chain record), the counters summed
  // on the device
  std::vector<uint2> hs(std::max<uint32_t>(K, 1));
  unsigned long long htot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  Globals g;
  {
    uint2 *d_sum = nullptr;
    unsigned long long *d_tot = nullptr;
    DMALLOC(d_sum, (size_t)std::max<uint32_t>(K, 1) * sizeof(uint2));
    DMALLOC(d_tot, 64);
    HIPCHK(hipMemsetAsync(d_tot, 0, 64, st));
    launch_chain_summary(st, P, d_sum, d_tot);
    HIPCHK(hipGetLastError());
    if (K) HIPCHK(hipMemcpyAsync(hs.data(), d_sum, (size_t)K * sizeof(uint2), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(htot, d_tot, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&g, P.glob, sizeof(g), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx->dfree(d_sum); ctx->dfree(d_tot);
  }
#ifdef SR_PHASE_TIMING
  std::vector<Chain> hc(K);
  if (K) HIPCHK(hipMemcpy(hc.data(), P.chains, (size_t)K * sizeof(Chain), hipMemcpyDeviceToHost));
#endif
  // chain i -> tid i % num_thr, chains ascending inside a tid (each per-tid file is a
  // sequence of whole contigs, which is all the encoder needs: encoder.h:215-363)
  std::vector<uint64_t> off_m(K), off_s(K);
  ctx->tid_off.assign(T + 1, 0);
  ctx->tid_off_s.assign(T + 1, 0);
  uint64_t am = 0, as = 0;
  spring_reorder_stats &s = ctx->stats;
  s.unmatched = s.probes = s.keyok = s.cands = s.iterations = s.lost = s.hits = 0;
  ctx->tid_mid.assign(T, 0);
  ctx->tid_mid_s.assign(T, 0);
  for (int t = 0; t < T; t++) {
    ctx->tid_off[t] = am;
    ctx->tid_off_s[t] = as;
    // chain id -> tid id % num_thr, chain ids ascending inside a tid: the local chains of the first group's slice, then those of
    // the second group's (a pool that runs two groups: the slices' global ids are c0 + i with the slice's own c0; one group:
    // one slice, the second one empty).  tid_mid: where the second slice's records begin -- the merge over the ranks of a
    // pool takes every rank's first part, then every rank's second part (pool.py, reorder_files.cpp)
    for (int g = 0; g < 2; g++) {
      const auto &a = ctx->grp[g];
      if (g == 1) { ctx->tid_mid[t] = am; ctx->tid_mid_s[t] = as; }
      if (!a.Kg) continue;
      const uint32_t first = (uint32_t)(((uint32_t)t + (uint32_t)T - (a.c0 + a.g0) % (uint32_t)T) % (uint32_t)T);  // first chain of tid t in the slice
      for (uint32_t i = a.g0 + first; i < a.g0 + a.Kg; i += (uint32_t)T) {
        off_m[i] = am; off_s[i] = as;
        am += hs[i].x; as += hs[i].y;
      }
    }
  }
  ctx->tid_off[T] = am;
  ctx->tid_off_s[T] = as;
  s.unmatched = htot[0]; s.probes = htot[1]; s.keyok = htot[2]; s.cands = htot[3]; s.iterations = htot[4];
  s.lost = htot[5]; s.hits = htot[6]; s.long_searches = htot[7];
  s.table_minz = ctx->minz; s.table_marked_lines = ctx->marked_lines;
#ifdef SR_PHASE_TIMING  // experiment builds: per-phase shader clocks of k_round (tools/xbuild.sh, XPIPE=1)
  {
    unsigned long long pt[64] = {0};
    for (uint32_t i = 0; i < K; i++)
      for (int k = 0; k < 64; k++) pt[k] += hc[i].pt[k];
    unsigned long long tot = 0;
    for (int k = 0; k < 32; k++) tot += pt[k];
    fprintf(stderr, "[phase] bucket: total clocks (share of buckets 0-11), visits, clocks per visit\n");
    for (int k = 0; k < 32; k++)
      if (pt[32 + k])
        fprintf(stderr, "[phase] %2d: %16llu (%5.1f %%) %12llu %9.0f\n", k, pt[k], 100.0 * (double)pt[k] / (double)std::max(tot, 1ull),
                pt[32 + k], (double)pt[k] / (double)pt[32 + k]);
    unsigned long long dbg[65];
    HIPCHK(hipMemcpy(dbg, P.dbg, sizeof(dbg), hipMemcpyDeviceToHost));
    fprintf(stderr, "[phase] %llu wavefronts ran > 1M clocks; their buckets (clocks/visits):", dbg[64]);
    for (int k = 0; k < 32; k++)
      if (dbg[32 + k]) fprintf(stderr, " %d:%llu/%llu", k, dbg[k], dbg[32 + k]);
    fprintf(stderr, "\n");
    // the three chains that spent the most clocks: what a round's slowest wavefronts are doing
    std::vector<std::pair<unsigned long long, uint32_t>> top;
    for (uint32_t i = 0; i < K; i++) {
      unsigned long long t = 0;
      for (int k = 0; k < 32; k++) t += hc[i].pt[k];
      top.push_back({t, i});
    }
    std::partial_sort(top.begin(), top.begin() + std::min<size_t>(3, top.size()), top.end(), std::greater<>());
    for (size_t j = 0; j < std::min<size_t>(3, top.size()); j++) {
      fprintf(stderr, "[phase] chain %u: %llu clocks:", top[j].second, top[j].first);
      for (int k = 0; k < 32; k++)
        if (hc[top[j].second].pt[32 + k]) fprintf(stderr, " %d:%u/%u", k, hc[top[j].second].pt[k], hc[top[j].second].pt[32 + k]);
      fprintf(stderr, "\n");
    }
  }
#endif
  // a rank of a multi-GPU pool only holds the records of the chains it owns
  if ((!ctx->mg && am + as != ctx->n) || am + as > ctx->n || g.e_alloc > ctx->cap + CHUNK || g.s_alloc > ctx->cap + CHUNK)
    return fail(SPRING_REORDER_E_STATE, "internal: emission counts do not add up (%llu+%llu vs n=%u, alloc %u/%u cap %llu)",
                (unsigned long long)am, (unsigned long long)as, ctx->n, g.e_alloc, g.s_alloc,
                (unsigned long long)ctx->cap);
  ctx->nrec = am;
  ctx->nsing = as;
  s.n_reads = ctx->n; s.n_matched = am; s.n_single = as;
  const size_t nm = std::max<uint64_t>(am, 1), ns = std::max<uint64_t>(as, 1);
  DMALLOC(P.f_order, nm * 4); DMALLOC(P.f_rc, nm); DMALLOC(P.f_flag, nm); DMALLOC(P.f_pos, nm * 8);
  DMALLOC(P.f_len, nm * 2); DMALLOC(P.f_order_s, ns * 4);
  uint64_t *d_off_m = nullptr, *d_off_s = nullptr;
  DMALLOC(d_off_m, (size_t)K * 8);
  DMALLOC(d_off_s, (size_t)K * 8);
  HIPCHK(hipMemcpyAsync(d_off_m, off_m.data(), (size_t)K * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off_s, off_s.data(), (size_t)K * 8, hipMemcpyHostToDevice, st));
  launch_scatter(st, P, am ? ctx->cap : 0, as ? ctx->cap : 0, d_off_m, d_off_s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[7], st));
  HIPCHK(hipStreamSynchronize(st));
  s.long_splits = 0;
  if (P.lctl) {
    uint32_t ns = 0;  // (k_mg_mark zeroes [0..1] every round, [3] counts the run's split searches)
    HIPCHK(hipMemcpy(&ns, P.lctl + 3, 4, hipMemcpyDeviceToHost));
    s.long_splits = ns;
    if (P.phases == 2 && ctx->lb2.lctl) {  // (the second chain group's)
      HIPCHK(hipMemcpy(&ns, ctx->lb2.lctl + 3, 4, hipMemcpyDeviceToHost));
      s.long_splits += ns;
    }
  }
  ctx->dfree(d_off_m); ctx->dfree(d_off_s);
  // the append-order buffers are no longer needed
  ctx->dfree(P.e_rec); ctx->dfree(P.e_chunk); ctx->dfree(P.s_rec); ctx->dfree(P.s_chunk);
  P.e_rec = nullptr;
  ctx->stage = ST_FINAL;
  return 0;
}

int spring_reorder_tid_split(spring_reorder_ctx *ctx, uint64_t *mid, uint64_t *mid_s) {
  if (!ctx || ctx->stage != ST_FINAL) return fail(SPRING_REORDER_E_STATE, "tid_split: finalize first");
  const int T = (int)ctx->tid_mid.size();
  for (int t = 0; t < T; t++) {
    if (mid) mid[t] = ctx->tid_mid[t];
    if (mid_s) mid_s[t] = ctx->tid_mid_s[t];
  }
  return 0;
}

int spring_reorder_get_stats(spring_reorder_ctx *ctx, spring_reorder_stats *out) {
  if (!ctx || !out) return fail(SPRING_REORDER_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(ctx->dev));
  HIPCHK(hipStreamSynchronize(ctx->st));
  spring_reorder_stats &s = ctx->stats;
  auto el = [&](int a, int b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) != hipSuccess) ms = 0;
    return (double)ms;
  };
  if (ctx->stage >= ST_LOADED) s.ms_unpack = el(0, 1);
  if (ctx->stage >= ST_DICT) s.ms_dict = el(2, 3);
  if (ctx->stage >= ST_CHAINS) s.ms_chains = el(4, 5);
  if (ctx->stage >= ST_FINAL) s.ms_finalize = el(6, 7);
  s.ms_total = s.ms_unpack + s.ms_dict + s.ms_chains + s.ms_finalize;
  for (int l = 0; l < 2; l++) { s.numkeys[l] = ctx->dict[l].numkeys; s.dict_numreads[l] = ctx->dict[l].numreads; }
  s.n_reads = ctx->n;
  s.device_bytes = ctx->peak_bytes;
  *out = s;
  return 0;
}

int spring_reorder_download(spring_reorder_ctx *ctx, uint32_t *order, char *rc, char *flag, int64_t *pos,
                            uint16_t *rlen, uint32_t *order_s, uint64_t *tid_off, uint64_t *tid_off_s) {
  if (!ctx || ctx->stage != ST_FINAL) return fail(SPRING_REORDER_E_STATE, "download: finalize first");
  HIPCHK(hipSetDevice(ctx->dev));
  DevParams &P = ctx->P;
  const size_t nm = ctx->nrec, ns = ctx->nsing;
  std::vector<D2HJob> jobs;
  if (nm) {
    if (order) jobs.push_back({order, P.f_order, nm * 4});
    if (rc) jobs.push_back({rc, P.f_rc, nm});
    if (flag) jobs.push_back({flag, P.f_flag, nm});
    if (pos) jobs.push_back({pos, P.f_pos, nm * 8});
    if (rlen) jobs.push_back({rlen, P.f_len, nm * 2});
  }
  if (ns && order_s) jobs.push_back({order_s, P.f_order_s, ns * 4});
  const int rd = download_chunked(ctx, jobs);
  if (rd) return rd;
  if (tid_off) memcpy(tid_off, ctx->tid_off.data(), ctx->tid_off.size() * 8);
  if (tid_off_s) memcpy(tid_off_s, ctx->tid_off_s.data(), ctx->tid_off_s.size() * 8);
  return 0;
}

extern "C++" {
namespace sr {
int emit_dna_device(spring_reorder_ctx *ctx, int32_t tid, uint8_t **d_out, size_t *nbytes, uint64_t s_first, uint64_t s_cnt, size_t *mid_bytes) {
  if (!ctx || ctx->stage != ST_FINAL) return fail(SPRING_REORDER_E_STATE, "emit_dna: finalize first");
  if (tid < -1 || tid >= ctx->o.num_thr) return fail(SPRING_REORDER_E_ARG, "tid out of range");
  HIPCHK(hipSetDevice(ctx->dev));
  hipStream_t st = ctx->st;
  DevParams &P = ctx->P;
  const uint32_t *order;
  const char *rc;
  uint64_t cnt;
  if (tid < 0) {
    order = P.f_order_s; rc = nullptr; cnt = ctx->nsing;
    if (s_cnt != ~0ull) {
      if (s_first > cnt || s_cnt > cnt - s_first) return fail(SPRING_REORDER_E_ARG, "emit_dna: singleton range out of bounds");
      order += s_first; cnt = s_cnt;
    }
  } else {
    const uint64_t a = ctx->tid_off[tid], b = ctx->tid_off[tid + 1];
    order = P.f_order + a; rc = P.f_rc + a; cnt = b - a;
  }
  const uint32_t rec = 2u + ((uint32_t)ctx->L + 3u) / 4u;
  uint64_t total = 0;
  uint64_t *d_off = nullptr;
  uint32_t *d_sz = nullptr;
  void *d_tmp = nullptr;
  if (ctx->uniform) total = cnt * rec;
  else if (cnt) {
    DMALLOC(d_sz, cnt * 4);
    DMALLOC(d_off, cnt * 8);
    launch_rec_size(st, order, ctx->d_lens, cnt, d_sz);
    size_t tb = 0;
    HIPCHK(excl_scan_u32_to_u64(st, nullptr, tb, d_sz, d_off, cnt));
    DMALLOC(d_tmp, tb);
    HIPCHK(excl_scan_u32_to_u64(st, d_tmp, tb, d_sz, d_off, cnt));
    uint64_t last_off = 0;
    uint32_t last_sz = 0;
    HIPCHK(hipMemcpyAsync(&last_off, d_off + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&last_sz, d_sz + (cnt - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    total = last_off + last_sz;
  }
  if (mid_bytes) {  // where the records of the second chain group's chains begin in this tid's stream (ctx->tid_mid)
    const uint64_t idx = tid >= 0 ? ctx->tid_mid[tid] - ctx->tid_off[tid] : 0;
    if (ctx->uniform || idx == 0) *mid_bytes = idx * rec;
    else if (idx >= cnt) *mid_bytes = total;
    else {
      uint64_t o = 0;
      HIPCHK(hipMemcpyAsync(&o, d_off + idx, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      *mid_bytes = o;
    }
  }
  if (nbytes) *nbytes = total;
  if (d_out) {
    *d_out = nullptr;
    if (total) {
      uint8_t *d_dst = nullptr;
      DMALLOC(d_dst, total);
      launch_emit_dna(st, ctx->d_reads, ctx->d_lens, ctx->S, order, rc, cnt, d_off, rec, d_dst);
      HIPCHK(hipGetLastError());
      *d_out = d_dst;
    }
  }
  ctx->dfree(d_sz); ctx->dfree(d_off); ctx->dfree(d_tmp);  // (dfree waits for the stream: the kernel above has run)
  return 0;
}
void emit_dna_free(spring_reorder_ctx *ctx, uint8_t *d) { if (ctx && d) ctx->dfree(d); }
}  // namespace sr
}  // extern "C++"

int spring_reorder_emit_dna(spring_reorder_ctx *ctx, int32_t tid, uint8_t *dst, size_t cap, size_t *nbytes) {
  size_t total = 0;
  uint8_t *d = nullptr;
  int r = emit_dna_device(ctx, tid, dst ? &d : nullptr, &total);
  if (r) return r;
  if (nbytes) *nbytes = total;
  if (dst && total) {
    if (cap < total) r = fail(SPRING_REORDER_E_ARG, "emit_dna: buffer too small (%zu < %llu)", cap, (unsigned long long)total);
    else if (hipStreamSynchronize(ctx->st) != hipSuccess || hipMemcpy(dst, d, total, hipMemcpyDeviceToHost) != hipSuccess)
      r = fail(SPRING_REORDER_E_HIP, "emit_dna: device to host copy failed");
  }
  emit_dna_free(ctx, d);
  return r;
}

}  // extern "C"
