// Face B: hyperscan() and check_patterns() with the reference shim's exact C ABI
// (hypergrep/lib/c/hyperscanner.c:154-159, :248-258), so hypergrep/utils.py:116-121 and :339-349 can call
// this library unchanged.  The file is streamed through pinned memory into HBM in large chunks, each
// chunk is scanned by the device pipeline (HgScanner), and hit records come back to fill the same
// batched result ring the reference fills in hs_callback() (hyperscanner.c:83-102).
//
// Ingest is a three-stage pipeline: a reader thread fills pinned slots (plain files: several pread stripes in
// parallel; gzip / zstd: the decoder) and cuts them at line-piece boundaries, the calling thread copies slot k+1 to
// HBM on a copy stream while slot k is scanned, and delivers slot k's hits from the pinned bytes while the reader is
// already filling the next slot.
//
// What stays on the host: file IO, gzip/zstd decoding (the reference does that on the CPU too, through
// zlibWrapper), cutting chunks at line boundaries, copying matched line bytes into the result ring and
// calling back.  No byte of the text is matched on the CPU.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hypergrep_amd.h"
#include "hg_compile.h"
#include "hg_engine.h"
#include "hg_mem.h"

namespace {

// ---------------------------------------------------------------- compiled-database cache
// The reference recompiles the database for every file (hyperscanner.c:296); here identical pattern sets share one
// compiled database for the life of the process (at most kDbCacheMax sets, least recently used evicted; scanners keep
// their own reference).  An entry is re-tuned once, on the first large file scanned with it (maybe_tune below).
struct DbEntry {
  std::shared_ptr<const HgDb> db;
  bool tune_tried = false;
  uint64_t stamp = 0;  // last use
};
constexpr size_t kDbCacheMax = 16;
std::mutex g_mu;
std::map<std::string, DbEntry> g_dbs;
uint64_t g_stamp = 0;
std::atomic<uint64_t> g_cache_hits{0}, g_cache_misses{0}, g_tunes{0};

std::string db_key(const char *const *patterns, const unsigned *flags, const unsigned *ids, unsigned n) {
  std::string k;
  for (unsigned i = 0; i < n; i++) {
    const char *p = patterns[i] ? patterns[i] : "";
    uint32_t len = static_cast<uint32_t>(std::strlen(p)), f = flags ? flags[i] : 0, id = ids ? ids[i] : 0;
    k.append(reinterpret_cast<const char *>(&len), 4);
    k.append(reinterpret_cast<const char *>(&f), 4);
    k.append(reinterpret_cast<const char *>(&id), 4);
    k.append(p, len);
  }
  return k;
}

std::shared_ptr<const HgDb> get_db(const char *const *patterns, const unsigned *flags, const unsigned *ids, unsigned n, std::string *err,
                                   std::string *key_out = nullptr) {
  if (!patterns || n == 0) {
    if (err) *err = "no patterns";
    return nullptr;
  }
  for (unsigned i = 0; i < n; i++)
    if (!patterns[i]) return nullptr;
  std::string key = db_key(patterns, flags, ids, n);
  if (key_out) *key_out = key;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_dbs.find(key);
    if (it != g_dbs.end()) {
      it->second.stamp = ++g_stamp;
      g_cache_hits++;
      return it->second.db;
    }
  }
  g_cache_misses++;
  HgDb *raw = nullptr;
  int bad = -1;
  if (hgc_compile(patterns, flags, ids, n, &raw, err, &bad) != 0) return nullptr;
  std::shared_ptr<const HgDb> db(raw, [](const HgDb *d) { hgc_free(const_cast<HgDb *>(d)); });
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_dbs.find(key);
  if (it != g_dbs.end()) {  // another thread compiled the same set meanwhile: keep one
    it->second.stamp = ++g_stamp;
    return it->second.db;
  }
  while (g_dbs.size() >= kDbCacheMax) {
    auto oldest = g_dbs.begin();
    for (auto j = g_dbs.begin(); j != g_dbs.end(); ++j)
      if (j->second.stamp < oldest->second.stamp) oldest = j;
    g_dbs.erase(oldest);
  }
  DbEntry e;
  e.db = db;
  e.stamp = ++g_stamp;
  g_dbs.emplace(std::move(key), std::move(e));
  return db;
}

// The window prefilter picks, per required literal, WHICH four bytes of it the stream pass looks for.  Statically that is
// a guess from byte frequencies of English-like logs; with a sample of the actual text it is the window that is rarest in
// that text (bench.py's workload: first-level matches 4 % -> 1.2 % of dwords).  The file API does this by itself: the
// first time a pattern set meets a file of at least HYPERGREP_TUNE_MIN_MB (default 32; 0 = never), 1 MiB of the first
// ingest chunk (four spread pieces) is sampled on the host (~20 ms), the tuned copy replaces the cache entry, and this
// and every later call of the pattern set scan with it.  Results never depend on it.
size_t tune_min_bytes() {
  static const size_t v = [] {
    if (const char *env = std::getenv("HYPERGREP_TUNE_MIN_MB")) {
      const long mb = std::atol(env);
      if (mb >= 0 && mb <= (1 << 20)) return static_cast<size_t>(mb) << 20;
    }
    return static_cast<size_t>(32) << 20;
  }();
  return v;
}
std::shared_ptr<const HgDb> maybe_tune(const std::string &key, const std::shared_ptr<const HgDb> &db, const uint8_t *chunk, size_t nbytes) {
  const size_t min_bytes = tune_min_bytes();
  if (db->tuned || !db->nreal_factors || !min_bytes || nbytes < min_bytes) return db;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_dbs.find(key);
    if (it == g_dbs.end() || it->second.tune_tried) return it != g_dbs.end() && it->second.db->tuned ? it->second.db : db;
    it->second.tune_tried = true;  // concurrent calls with the same set scan untuned this once
  }
  constexpr size_t kPiece = 256u << 10, kPieces = 4;
  std::vector<uint8_t> sample;
  sample.reserve(kPiece * kPieces);
  for (size_t i = 0; i < kPieces; i++) {
    const size_t at = (nbytes / kPieces * i) & ~static_cast<size_t>(15);
    const size_t len = std::min(kPiece, nbytes - at);
    sample.insert(sample.end(), chunk + at, chunk + at + len);
  }
  HgDb *raw = nullptr;
  std::string err;
  if (hgc_tune(db.get(), sample.data(), sample.size(), &raw, &err) != 0) return db;
  std::shared_ptr<const HgDb> tuned(raw, [](const HgDb *d) { hgc_free(const_cast<HgDb *>(d)); });
  g_tunes++;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_dbs.find(key);
  if (it != g_dbs.end()) it->second.db = tuned;
  return tuned;
}

// ---------------------------------------------------------------- per-call device context, pooled
// A context = streams + staging buffers (pinned slots, device text buffers) + the scanner of ONE pattern set.  At most
// HYPERGREP_POOL contexts exist at a time (default 16: the reference's thread pool runs up to ncpu-1 scans, each of them
// link-bound here); a call beyond that waits for one to come back.  Idle contexts are kept, most recently used last: a
// call takes one that already holds its pattern set's scanner if there is one (the reference's use: one pattern set, many
// files), else creates a new context while the bound allows, else re-binds the least recently used idle one (buffers and
// streams stay, the scanner is replaced).
struct Ctx {
  std::shared_ptr<const HgDb> db;  // pattern set of `sc`
  HgScanner *sc = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  static constexpr int kSlots = 3, kDevBufs = 2;
  uint8_t *h_slot[kSlots] = {nullptr, nullptr, nullptr};  // pinned staging buffers
  size_t h_cap = 0;
  uint8_t *d_text[kDevBufs] = {nullptr, nullptr};
  size_t d_cap = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_h2d[kDevBufs] = {nullptr, nullptr};
  std::vector<HgHit> hits;
  std::vector<HgHitAux> aux;
  ~Ctx() {
    (void)hipSetDevice(device);
    delete sc;
    for (uint8_t *p : h_slot) hgmem::host_free(p, "h_slot");
    for (uint8_t *p : d_text) hgmem::dev_free(p, "d_text");
    for (hipEvent_t e : ev_h2d)
      if (e) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (stream) (void)hipStreamDestroy(stream);
  }
};
struct Pool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Ctx *> idle;  // least recently used first; intentionally never destroyed at exit (no HIP calls in static dtors)
  size_t live = 0;          // contexts in existence (idle + checked out)
  size_t cap = 16;
  std::atomic<uint64_t> created{0}, reused{0}, rebound{0};
  Pool() {
    if (const char *env = std::getenv("HYPERGREP_POOL")) cap = static_cast<size_t>(std::max(1l, std::min(1024l, std::atol(env))));  // read once
  }
};
Pool &pool() {
  static Pool *p = new Pool();
  return *p;
}
std::atomic<unsigned> g_next_device{0};

}  // namespace
// Device of the next Face B context when the node has `ndev` GPUs: HYPERGREP_DEVICE pins one, else files round-robin.
extern "C" int hg_faceb_next_device(int ndev) {
  if (ndev <= 0) return -1;
  if (const char *env = std::getenv("HYPERGREP_DEVICE")) return std::abs(std::atoi(env)) % ndev;
  return static_cast<int>(g_next_device.fetch_add(1) % static_cast<unsigned>(ndev));
}
// Counters of the caches above (tests, HYPERGREP_TRACE): database cache hits / misses / size, tunes, contexts created /
// reused with their scanner / re-bound to another pattern set / alive.
extern "C" void hg_faceb_stats(uint64_t out[8]) {
  Pool &p = pool();
  out[0] = g_cache_hits;
  out[1] = g_cache_misses;
  out[3] = g_tunes;
  out[4] = p.created;
  out[5] = p.reused;
  out[6] = p.rebound;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    out[2] = g_dbs.size();
  }
  std::lock_guard<std::mutex> lock(p.mu);
  out[7] = p.live;
}
namespace {

Ctx *checkout(const std::shared_ptr<const HgDb> &db, std::string *err) {
  Pool &p = pool();
  Ctx *c = nullptr;
  {
    std::unique_lock<std::mutex> lock(p.mu);
    for (;;) {
      for (size_t i = p.idle.size(); i-- > 0;)
        if (p.idle[i]->db == db) {
          c = p.idle[i];
          p.idle.erase(p.idle.begin() + static_cast<long>(i));
          p.reused++;
          return c;
        }
      if (p.live < p.cap) {
        p.live++;
        break;  // create one (outside the lock)
      }
      if (!p.idle.empty()) {
        c = p.idle.front();
        p.idle.erase(p.idle.begin());
        p.rebound++;
        return c;  // the caller replaces its scanner (ensure_scanner)
      }
      p.cv.wait(lock);
    }
  }
  auto fail = [&](const char *what) {
    *err = what;
    {
      std::lock_guard<std::mutex> lock(p.mu);
      p.live--;
    }
    p.cv.notify_one();
    return static_cast<Ctx *>(nullptr);
  };
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available; hypergrep_amd has no CPU scan path");
  auto fresh = std::make_unique<Ctx>();
  fresh->device = hg_faceb_next_device(ndev);  // files shard over the node's GPUs
  if (hipSetDevice(fresh->device) != hipSuccess || hipStreamCreateWithFlags(&fresh->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&fresh->copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&fresh->ev_h2d[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&fresh->ev_h2d[1], hipEventDisableTiming) != hipSuccess)
    return fail("hipStreamCreate failed");
  p.created++;
  return fresh.release();
}
// healthy = the call ended without a device error: the context goes back for reuse.  After a HIP error the scanner's
// streams and workspace are in an unknown state: everything is synchronised and destroyed instead.
void checkin(Ctx *c, bool healthy) {
  Pool &p = pool();
  if (!healthy) {
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    delete c;
    c = nullptr;
  }
  {
    std::lock_guard<std::mutex> lock(p.mu);
    if (c) p.idle.push_back(c);
    else p.live--;
  }
  p.cv.notify_one();
}
// The context's scanner must be the one of `db` (created on first use; replaced when the context is re-bound).
bool ensure_scanner(Ctx *c, const std::shared_ptr<const HgDb> &db, std::string *err) {
  if (c->sc && c->db == db) return true;
  if (hipSetDevice(c->device) != hipSuccess) return false;
  delete c->sc;  // (hipFree synchronises the device: nothing of the old scanner is in flight afterwards)
  c->sc = nullptr;
  c->db = db;
  return HgScanner::create(db, c->device, &c->sc, err) == HG_OK;
}
// slots: pinned buffers needed (1 for a file that fits one chunk, else all)
bool ensure_buffers(Ctx *c, size_t cap, int slots) {
  if (c->h_cap < cap) {
    for (uint8_t *&p : c->h_slot) {
      hgmem::host_free(p, "h_slot");
      p = nullptr;
    }
    c->h_cap = cap;
  }
  for (int i = 0; i < slots; i++)
    if (!c->h_slot[i] && hgmem::host_alloc(&c->h_slot[i], c->h_cap, "h_slot") != hipSuccess) return false;
  if (c->d_cap < cap) {
    for (uint8_t *&p : c->d_text) {
      hgmem::dev_free(p, "d_text");
      p = nullptr;
    }
    c->d_cap = cap;
  }
  // + 32: the kernels may read the text up to its size rounded up to 16 bytes, and the verify pass's byte-aligned
  // discriminator load up to 3 bytes further (hg_verify_kernel)
  for (int i = 0; i < (slots > 1 ? Ctx::kDevBufs : 1); i++)
    if (!c->d_text[i] && hgmem::dev_alloc(&c->d_text[i], c->d_cap + 32, "d_text") != hipSuccess) return false;
  return true;
}

// ---------------------------------------------------------------- input: plain / gzip / zstd
// gzopen("rb") in the reference (through zstd's zlibWrapper, hyperscanner.c:22,191) auto-detects gzip and
// zstd by magic and reads anything else verbatim; this reader does the same.
struct ZIn { const void *src; size_t size, pos; };
struct ZOut { void *dst; size_t size, pos; };
struct Zstd {
  void *lib = nullptr;
  void *(*create)() = nullptr;
  size_t (*destroy)(void *) = nullptr;
  size_t (*step)(void *, ZOut *, ZIn *) = nullptr;
  unsigned (*is_error)(size_t) = nullptr;
  bool load() {
    if (lib) return true;
    for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
      lib = dlopen(name, RTLD_NOW);
      if (lib) break;
    }
    if (!lib) return false;
    create = reinterpret_cast<void *(*)()>(dlsym(lib, "ZSTD_createDStream"));
    destroy = reinterpret_cast<size_t (*)(void *)>(dlsym(lib, "ZSTD_freeDStream"));
    step = reinterpret_cast<size_t (*)(void *, ZOut *, ZIn *)>(dlsym(lib, "ZSTD_decompressStream"));
    is_error = reinterpret_cast<unsigned (*)(size_t)>(dlsym(lib, "ZSTD_isError"));
    return create && destroy && step && is_error;
  }
};
Zstd g_zstd;

class Reader {
 public:
  ~Reader() { close(); }
  bool open(const char *path) {
    fd_ = ::open(path, O_RDONLY);
    if (fd_ < 0) return false;
    struct stat st;
    if (fstat(fd_, &st) != 0 || S_ISDIR(st.st_mode)) return false;
    size_hint_ = S_ISREG(st.st_mode) ? static_cast<size_t>(st.st_size) : 0;
    unsigned char magic[4] = {0, 0, 0, 0};
    ssize_t got = ::pread(fd_, magic, 4, 0);
    if (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
      gz_ = gzdopen(fd_, "rb");
      if (!gz_) return false;
      fd_ = -1;  // owned by zlib now
      gzbuffer(gz_, 1 << 20);
      kind_ = 1;
    } else if (got == 4 && magic[0] == 0x28 && magic[1] == 0xB5 && magic[2] == 0x2F && magic[3] == 0xFD) {
      std::lock_guard<std::mutex> lock(g_mu);
      if (!g_zstd.load()) return false;
      zds_ = g_zstd.create();
      if (!zds_) return false;
      zin_.resize(1 << 20);
      kind_ = 2;
    }
    return true;
  }
  bool compressed() const { return kind_ != 0; }
  size_t size_hint() const { return size_hint_; }
  // Fill dst with up to n bytes; returns bytes produced, 0 at EOF, -1 on error.
  long read(uint8_t *dst, size_t n) {
    if (kind_ == 0) {
      // regular files: the request is cut into stripes read by parallel preads (one thread copies ~5 GB/s out of the page
      // cache, the host link takes ten times that); pipes and the like: plain read()
      auto pread_all = [this](uint8_t *p, size_t len, off_t at) -> long {
        size_t total = 0;
        while (total < len) {
          ssize_t r = ::pread(fd_, p + total, len - total, at + static_cast<off_t>(total));
          if (r < 0) return -1;
          if (r == 0) break;
          total += static_cast<size_t>(r);
        }
        return static_cast<long>(total);
      };
      if (size_hint_) {
        const size_t left = static_cast<size_t>(off_) < size_hint_ ? size_hint_ - static_cast<size_t>(off_) : 0;
        const size_t want = std::min(n, left);
        if (want == 0) {  // at the size seen at open: the file may have grown meanwhile
          const long g = pread_all(dst, n, off_);
          if (g > 0) off_ += g;
          return g;
        }
        const size_t kStripe = static_cast<size_t>(8) << 20;
        const unsigned workers = static_cast<unsigned>(std::min<size_t>(read_threads(), want / kStripe));
        long got;
        if (workers <= 1) {
          got = pread_all(dst, want, off_);
        } else {
          std::vector<long> part(workers, 0);
          std::vector<std::thread> pool;
          const size_t per = (want / workers + 4095) & ~static_cast<size_t>(4095);
          for (unsigned w = 0; w < workers; w++) {
            const size_t b = std::min(want, per * w), e = w + 1 == workers ? want : std::min(want, per * (w + 1));
            pool.emplace_back([&, w, b, e] { part[w] = pread_all(dst + b, e - b, off_ + static_cast<off_t>(b)); });
          }
          for (auto &t : pool) t.join();
          got = 0;
          for (unsigned w = 0; w < workers; w++) {
            const size_t b = std::min(want, per * w), e = w + 1 == workers ? want : std::min(want, per * (w + 1));
            if (part[w] < 0) return -1;
            got += part[w];
            if (static_cast<size_t>(part[w]) < e - b) break;  // the file shrank: stop at the first short stripe
          }
        }
        if (got > 0) off_ += got;
        return got;
      }
      size_t total = 0;
      while (total < n) {
        ssize_t r = ::read(fd_, dst + total, n - total);
        if (r < 0) return -1;
        if (r == 0) break;
        total += static_cast<size_t>(r);
      }
      return static_cast<long>(total);
    }
    if (kind_ == 1) {
      size_t total = 0;
      while (total < n) {
        unsigned want = static_cast<unsigned>(std::min<size_t>(n - total, 1u << 30));
        int r = gzread(gz_, dst + total, want);
        if (r < 0) return -1;
        if (r == 0) break;
        total += static_cast<size_t>(r);
      }
      return static_cast<long>(total);
    }
    size_t total = 0;
    while (total < n) {
      if (zpos_ == zlen_ && !zeof_) {
        ssize_t r = ::read(fd_, zin_.data(), zin_.size());
        if (r < 0) return -1;
        if (r == 0) zeof_ = true;
        zlen_ = static_cast<size_t>(r > 0 ? r : 0);
        zpos_ = 0;
      }
      if (zpos_ == zlen_ && zeof_) break;
      ZIn in{zin_.data(), zlen_, zpos_};
      ZOut out{dst + total, n - total, 0};
      size_t rc = g_zstd.step(zds_, &out, &in);
      if (g_zstd.is_error(rc)) return -1;
      zpos_ = in.pos;
      total += out.pos;
    }
    return static_cast<long>(total);
  }
  void close() {
    if (gz_) gzclose(gz_);
    gz_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
    if (zds_) g_zstd.destroy(zds_);
    zds_ = nullptr;
  }

 private:
  static unsigned read_threads() {
    if (const char *env = std::getenv("HYPERGREP_READ_THREADS")) {
      long v = std::atol(env);
      if (v >= 1 && v <= 64) return static_cast<unsigned>(v);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw >= 16 ? 8u : (hw >= 4 ? hw / 2 : 1u);
  }
  int fd_ = -1, kind_ = 0;
  off_t off_ = 0;  // plain regular files are read with pread from here
  gzFile gz_ = nullptr;
  void *zds_ = nullptr;
  std::vector<uint8_t> zin_;
  size_t zpos_ = 0, zlen_ = 0, size_hint_ = 0;
  bool zeof_ = false;
};

// ---------------------------------------------------------------- result ring (hyperscanner.c:64-72, :83-102)
struct Ring {
  std::vector<hyperscanner_result_t> slots;
  char *storage = nullptr;  // buffer_count line buffers of buffer_size bytes (hyperscanner.c:277-291); malloc, not a zero-filled
                            // vector: with a large buffer_count most of it is never touched (1 GiB for 4096 x 262140)
  int fill = 0;
  hs_event cb = nullptr;
  unsigned long long delivered = 0;
  Ring() = default;
  Ring(const Ring &) = delete;
  Ring &operator=(const Ring &) = delete;
  ~Ring() { std::free(storage); }
  bool init(int count, int buffer_size, hs_event on_event) {
    cb = on_event;
    try {
      slots.resize(static_cast<size_t>(count));
    } catch (const std::bad_alloc &) {
      return false;
    }
    storage = static_cast<char *>(std::malloc(static_cast<size_t>(count) * static_cast<size_t>(buffer_size)));
    if (!storage) return false;
    for (int i = 0; i < count; i++) slots[i].line = storage + static_cast<size_t>(i) * static_cast<size_t>(buffer_size);
    return true;
  }
  void push(unsigned id, unsigned long long line_number, const uint8_t *line, uint32_t len) {
    hyperscanner_result_t &r = slots[static_cast<size_t>(fill++)];
    r.id = id;
    r.line_number = line_number;
    std::memcpy(r.line, line, len);
    r.line[len] = 0;
    delivered++;
    if (fill == static_cast<int>(slots.size())) flush();
  }
  void flush() {
    if (fill) cb(slots.data(), fill);
    fill = 0;
  }
};

size_t chunk_bytes() {
  if (const char *env = std::getenv("HYPERGREP_CHUNK_MB")) {
    long mb = std::atol(env);
    if (mb >= 1 && mb <= 16384) return static_cast<size_t>(mb) << 20;
  }
  return static_cast<size_t>(256) << 20;
}

}  // namespace

extern "C" int check_patterns(const char *const *patterns, const unsigned int *pattern_flags, const unsigned int *pattern_ids,
                              const unsigned int elements) {
  std::string err;
  return get_db(patterns, pattern_flags, pattern_ids, elements, &err) ? 0 : HYPERSCANNER_DB;
}

extern "C" int hyperscan(char *file_name, const char *const *patterns, const unsigned int *pattern_flags,
                         const unsigned int *pattern_ids, const unsigned int elements, hs_event on_event, const int buffer_size,
                         int buffer_count, unsigned long long max_match_count) {
  if (max_match_count > 0 && max_match_count < static_cast<unsigned long long>(buffer_count)) buffer_count = static_cast<int>(max_match_count);
  if (buffer_count < 1 || buffer_size < 1 || !on_event) return HYPERSCANNER_STATE_MEM;
  Ring ring;
  if (!ring.init(buffer_count, buffer_size, on_event)) return HYPERSCANNER_COMPILE_MEM;

  std::string err, db_key_str;
  std::shared_ptr<const HgDb> db = get_db(patterns, pattern_flags, pattern_ids, elements, &err, &db_key_str);
  if (!db) {
    std::fprintf(stderr, "ERROR: Unable to create database. Exiting.\n");
    return HYPERSCANNER_DB;
  }
  // the file is opened before any device resource is taken: a missing file is HYPERSCANNER_GZ_OPEN with or without a GPU
  // (the reference's scratch allocation cannot fail for want of a device; its order is database, scratch, file)
  Reader in;
  if (!file_name || !in.open(file_name)) return HYPERSCANNER_GZ_OPEN;
  if (buffer_size < 2) return 0;  // gzgets(len <= 1) returns NULL at once: nothing is scanned (hyperscanner.c:199)

  Ctx *ctx = checkout(db, &err);
  if (!ctx) {
    std::fprintf(stderr, "ERROR: Unable to allocate scratch space. Exiting. (%s)\n", err.c_str());
    return HYPERSCANNER_SCRATCH;
  }
  bool healthy = true;  // false after a device error: the context is destroyed, not pooled
  struct Return {
    Ctx *c;
    bool *healthy;
    ~Return() { checkin(c, *healthy); }
  } ret_guard{ctx, &healthy};

  const uint64_t bs1 = static_cast<uint64_t>(buffer_size) - 1;
  size_t cap = chunk_bytes();
  const bool one_chunk = !in.compressed() && in.size_hint() && in.size_hint() < cap;
  if (one_chunk) cap = std::max<size_t>(in.size_hint() + 16, 1 << 16);
  // a chunk must hold a whole piece plus the carry of the previous one — unless the whole file is one chunk anyway
  if (!one_chunk) cap = std::max<size_t>(cap, static_cast<size_t>(std::min<uint64_t>(2 * bs1 + 16, static_cast<uint64_t>(1) << 32)));
  const int nslots = one_chunk ? 1 : Ctx::kSlots;
  if (hipSetDevice(ctx->device) != hipSuccess || !ensure_buffers(ctx, cap, nslots)) {
    std::fprintf(stderr, "ERROR: Unable to allocate scratch space. Exiting. (device buffers)\n");
    healthy = false;
    return HYPERSCANNER_SCRATCH;
  }

  // ---- stage 1: the reader thread fills slots and cuts them at piece boundaries
  struct Slot {
    size_t cut = 0;     // bytes to scan (whole pieces)
    bool last = false;  // nothing follows
    int state = 0;      // 0 free (reader's), 1 ready (consumer's)
  };
  Slot slots[Ctx::kSlots];
  // HYPERGREP_TRACE=1: where the call's wall time went (seconds), printed to stderr at the end
  const bool trace = std::getenv("HYPERGREP_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_call = now();
  double t_read = 0, t_wait_slot = 0, t_scan = 0, t_deliver = 0, t_setup = 0;  // reader: filling slots; consumer: waiting for a slot, copy + scan + hit copy, ring + callbacks
  uint64_t bytes_in = 0;
  std::mutex mu;
  std::condition_variable cv;
  bool abandon = false;  // the consumer stopped early (match limit, error)
  std::thread reader([&] {
    std::vector<uint8_t> carry;  // bytes after the cut, they open the next slot
    bool eof = false;
    for (unsigned k = 0;; k++) {
      Slot &s = slots[k % nslots];
      uint8_t *buf = ctx->h_slot[k % nslots];
      {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return s.state == 0 || abandon; });
        if (abandon) return;
      }
      size_t have = carry.size();
      if (have) std::memcpy(buf, carry.data(), have);
      carry.clear();
      const double t_fill = now();
      while (!eof && have < cap) {
        const long got = in.read(buf + have, cap - have);
        if (got <= 0) {  // end of the stream; a read error ends it like gzgets returning NULL
          eof = true;
          break;
        }
        have += static_cast<size_t>(got);
      }
      t_read += now() - t_fill;
      size_t cut = have;
      if (!eof && have) {  // cut at a piece boundary so that every piece is scanned whole
        const void *nl = memrchr(buf, '\n', have);
        if (nl) cut = static_cast<size_t>(static_cast<const uint8_t *>(nl) - buf) + 1;
        else cut = static_cast<size_t>((have / bs1) * bs1);  // one unterminated line: stop at a forced break
        if (!cut) cut = have;
        carry.assign(buf + cut, buf + have);
      }
      {
        std::lock_guard<std::mutex> lock(mu);
        s.cut = cut;
        s.last = eof && carry.empty();
        s.state = 1;
      }
      cv.notify_all();
      if (eof && carry.empty()) return;
    }
  });
  auto stop_reader = [&] {
    {
      std::lock_guard<std::mutex> lock(mu);
      abandon = true;
    }
    cv.notify_all();
    reader.join();
  };

  // ---- stages 2 and 3: copy to HBM (one chunk ahead when the reader is), scan, deliver
  uint64_t line_base = 0;  // pieces delivered to the scanner so far == reference line_number of the chunk's first piece
  bool stop = false;
  int rc = 0;
  bool copied[Ctx::kSlots] = {false, false, false};
  auto issue_copy = [&](unsigned k) -> bool {  // slot k -> device buffer k % 2, asynchronously
    const int sl = static_cast<int>(k % nslots), db_i = static_cast<int>(k % Ctx::kDevBufs);
    if (copied[sl]) return true;
    copied[sl] = true;
    if (!slots[sl].cut) return true;
    return hipMemcpyAsync(ctx->d_text[nslots > 1 ? db_i : 0], ctx->h_slot[sl], slots[sl].cut, hipMemcpyHostToDevice, ctx->copy_stream) == hipSuccess &&
           hipEventRecord(ctx->ev_h2d[db_i], ctx->copy_stream) == hipSuccess;
  };
  for (unsigned k = 0; !stop; k++) {
    const int sl = static_cast<int>(k % nslots);
    Slot &s = slots[sl];
    bool next_ready = false;
    {
      const double t0 = now();
      std::unique_lock<std::mutex> lock(mu);
      cv.wait(lock, [&] { return s.state == 1; });
      t_wait_slot += now() - t0;
      next_ready = nslots > 1 && !s.last && slots[(k + 1) % nslots].state == 1;
    }
    const uint8_t *host = ctx->h_slot[sl];
    const size_t cut = s.cut;
    const bool last = s.last;
    if (k == 0) {  // the pattern set's first large file picks the prefilter windows from this text (maybe_tune); then the scanner
      const double t0 = now();
      if (cut) db = maybe_tune(db_key_str, db, host, cut);
      if (!ensure_scanner(ctx, db, &err)) {
        std::fprintf(stderr, "ERROR: Unable to allocate scratch space. Exiting. (%s)\n", err.c_str());
        rc = HYPERSCANNER_SCRATCH;
        healthy = false;
        break;
      }
      t_setup = now() - t0;
    }
    if (cut) {
      const double t_begin = now();
      bytes_in += cut;
      uint8_t *d_text = ctx->d_text[nslots > 1 ? k % Ctx::kDevBufs : 0];
      if (!issue_copy(k) || (next_ready && !issue_copy(k + 1)) ||  // the next chunk travels while this one is scanned
          hipStreamWaitEvent(ctx->stream, ctx->ev_h2d[k % Ctx::kDevBufs], 0) != hipSuccess) {
        rc = HYPERSCANNER_SCAN;
        healthy = false;
        break;
      }
      HgScanOutput out{};
      int src = ctx->sc->scan(d_text, cut, buffer_size, line_base, ctx->stream, &out);
      if (src != HG_OK) {
        std::fprintf(stderr, "ERROR: Unable to scan buffer. Exiting. (%s)\n", ctx->sc->last_error().c_str());
        rc = HYPERSCANNER_SCAN;
        healthy = false;
        break;
      }
      ctx->hits.resize(out.n_hits);
      ctx->aux.resize(out.n_hits);
      if (out.n_hits) {
        if (hipMemcpyAsync(ctx->hits.data(), out.d_hits, out.n_hits * sizeof(HgHit), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(ctx->aux.data(), out.d_aux, out.n_hits * sizeof(HgHitAux), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
          rc = HYPERSCANNER_SCAN;
          healthy = false;
          break;
        }
      }
      const double t_scanned = now();
      t_scan += t_scanned - t_begin;
      // deliver line by line; inside a line reports go out by ascending end offset, then id (hs_scan order)
      size_t i = 0;
      while (i < ctx->hits.size() && !stop) {
        size_t j = i;
        while (j < ctx->hits.size() && ctx->hits[j].line_no == ctx->hits[i].line_no) j++;
        if (j - i > 1) {
          std::vector<size_t> order(j - i);
          for (size_t q = 0; q < order.size(); q++) order[q] = i + q;
          std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
            if (ctx->hits[x].to != ctx->hits[y].to) return ctx->hits[x].to < ctx->hits[y].to;
            return ctx->hits[x].id < ctx->hits[y].id;
          });
          for (size_t q : order) ring.push(ctx->hits[q].id, ctx->hits[q].line_no, host + ctx->aux[q].start, ctx->aux[q].len);
        } else {
          ring.push(ctx->hits[i].id, ctx->hits[i].line_no, host + ctx->aux[i].start, ctx->aux[i].len);
        }
        // the reference checks the limit after each line's hs_scan returns (hyperscanner.c:222-224)
        if (max_match_count > 0 && ring.delivered >= max_match_count) stop = true;
        i = j;
      }
      line_base += out.n_pieces;
      t_deliver += now() - t_scanned;
    }
    copied[sl] = false;
    {
      std::lock_guard<std::mutex> lock(mu);
      s.state = 0;  // the slot goes back to the reader
    }
    cv.notify_all();
    if (last) break;
  }
  stop_reader();
  (void)hipStreamSynchronize(ctx->copy_stream);  // a copy issued ahead may still be in flight
  ring.flush();
  if (trace)
    std::fprintf(stderr, "[hypergrep_amd] %s: %.1f MiB in %.4f s; reader filling slots %.4f s; consumer: waiting for data %.4f s, window tuning + scanner %.4f s (%s), copy + scan %.4f s, delivering %llu hits %.4f s\n",
                 file_name, bytes_in / 1048576.0, now() - t_call, t_read, t_wait_slot, t_setup, db->tuned ? "tuned windows" : "static windows", t_scan,
                 static_cast<unsigned long long>(ring.delivered), t_deliver);
  return rc;
}
