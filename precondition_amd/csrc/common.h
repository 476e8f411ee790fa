// common.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ps_api.h"

#define PS_HIP(expr)                          \
  do {                                        \
    hipError_t e__ = (expr);                  \
    if (e__ != hipSuccess) return (int)e__;   \
  } while (0)

#define PS_LAUNCH_CHECK()                     \
  do {                                        \
    hipError_t e__ = hipGetLastError();       \
    if (e__ != hipSuccess) return (int)e__;   \
  } while (0)

namespace psh {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

// Bump allocator over caller-provided workspace (device memory).
struct Arena {
  char* base;
  size_t cap;
  size_t off = 0;
  bool overflow = false;
  Arena(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    size_t bytes = count * sizeof(T);
    if (base != nullptr && off + bytes > cap) overflow = true;
    T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += bytes;
    return r;
  }
};

}  // namespace psh
