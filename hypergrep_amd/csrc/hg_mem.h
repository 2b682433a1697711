// Device / pinned allocations of the library go through these wrappers so that one environment variable shows where
// every byte lives: HG_MEMLOG=<file> appends one line per allocation, free and scan call (name, base, end, size).
// A GPU memory fault reports only an address; this log is what maps it to a buffer (or to the gap right after one).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace hgmem {

inline FILE *log_file() {
  static FILE *f = [] {
    const char *path = std::getenv("HG_MEMLOG");
    return path && *path ? std::fopen(path, "a") : nullptr;
  }();
  return f;
}
inline void note(const char *fmt, ...) {
  FILE *f = log_file();
  if (!f) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  va_list ap;
  va_start(ap, fmt);
  std::vfprintf(f, fmt, ap);
  va_end(ap);
  std::fflush(f);  // the process may die in the next kernel
}

template <typename T>
inline hipError_t dev_alloc(T **ptr, size_t bytes, const char *name) {
  hipError_t e = hipMalloc(reinterpret_cast<void **>(ptr), bytes);
  if (log_file())
    note("alloc dev  %-14s %p .. %p  %zu  %s\n", name, static_cast<void *>(*ptr), static_cast<void *>(reinterpret_cast<char *>(*ptr) + bytes), bytes,
         e == hipSuccess ? "ok" : hipGetErrorString(e));
  return e;
}
template <typename T>
inline hipError_t host_alloc(T **ptr, size_t bytes, const char *name) {
  hipError_t e = hipHostMalloc(reinterpret_cast<void **>(ptr), bytes);
  if (log_file())
    note("alloc host %-14s %p .. %p  %zu  %s\n", name, static_cast<void *>(*ptr), static_cast<void *>(reinterpret_cast<char *>(*ptr) + bytes), bytes,
         e == hipSuccess ? "ok" : hipGetErrorString(e));
  return e;
}
inline void dev_free(void *p, const char *name) {
  if (!p) return;
  if (log_file()) note("free  dev  %-14s %p\n", name, p);
  (void)hipFree(p);
}
inline void host_free(void *p, const char *name) {
  if (!p) return;
  if (log_file()) note("free  host %-14s %p\n", name, p);
  (void)hipHostFree(p);
}

}  // namespace hgmem
