// common.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/ps_api.h"

#define PS_HIP(expr)                          \
  do {                                        \
    hipError_t e__ = (expr);                  \
    if (e__ != hipSuccess) return (int)e__;   \
  } while (0)

#define PS_RC(expr)                           \
  do {                                        \
    int rc__ = (expr);                        \
    if (rc__ != 0) return rc__;               \
  } while (0)

#define PS_DEVICE_CHECK()                     \
  do {                                        \
    int rc__ = psh::bind_device();            \
    if (rc__ != 0) return rc__;               \
  } while (0)

#define PS_LAUNCH_CHECK()                     \
  do {                                        \
    hipError_t e__ = hipGetLastError();       \
    if (e__ != hipSuccess) return (int)e__;   \
  } while (0)

namespace psh {

// One process drives ONE device (one process per GPU, RCCL-style): the library keeps a
// few per-process resources that are bound to a device (the pinned upload ring's
// events, kernel attributes such as the dynamic-LDS limit, the mapped status rings).
// The first compute call binds the process to the current HIP device; a later call made
// with another device current is refused with PS_EDEVICE instead of enqueuing kernels
// on a stream of the wrong device.
inline int bind_device() {
  static std::mutex mu;
  static int bound = -1;
  int d = -1;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return 0;  // no device at all: the first HIP call below reports it
  std::lock_guard<std::mutex> lk(mu);
  if (bound < 0) bound = d;
  return bound == d ? 0 : PS_EDEVICE;
}

// Side streams of the multi-stream drivers (stream groups of the staged Newton execution, of the eigh
// reduction, the second group of the one-sided Jacobi sweeps): ONE pool per host thread, shared by all of
// them.  The runtime maps streams onto a few hardware queues (4 by default): with a pool per driver a
// process that had run a mixed Newton call (1 side stream) and then an eigh call (3 more) held 5 streams,
// two of the eigh groups shared a hardware queue and serialised -- cfg3 measured 219 ms inside the full
// bench against 172 ms standalone.  Streams are created on first use and live as long as the thread.
// (process-wide, not per thread: a host that runs two calls side by side from two threads -- the two phases of the
// exchange, comm.sharded_inverse_pth_roots -- must not double the number of live streams, nor leak a pool per
// short-lived thread; two calls that share a side stream are merely ordered on it.)
constexpr int PS_MAX_SIDE_STREAMS = 7;
inline hipStream_t side_stream(int k) {
  static hipStream_t pool[PS_MAX_SIDE_STREAMS] = {};
  static std::mutex mu;
  if (k < 0 || k >= PS_MAX_SIDE_STREAMS) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!pool[k] && hipStreamCreateWithFlags(&pool[k], hipStreamNonBlocking) != hipSuccess) pool[k] = nullptr;
  return pool[k];
}

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

// Host -> device upload of a small descriptor table without a stream synchronisation: the
// bytes are copied into a slot of a process-wide ring of pinned staging buffers and
// the H2D copy is enqueued from there, so the caller's host vectors may go out of scope
// at once and the host keeps running ahead of the GPU.  A slot is reused only after the
// event recorded behind its last copy has completed (32 slots: practically never waits).
inline int upload_async(hipStream_t st, void* dst, const void* src, size_t bytes) {
  struct Slot { void* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
  static std::mutex mu;
  static Slot slots[32];
  static unsigned next = 0;
  if (bytes == 0) return 0;
  std::lock_guard<std::mutex> lk(mu);
  Slot& s = slots[next++ % 32];
  if (s.used) PS_HIP(hipEventSynchronize(s.ev));
  if (s.cap < bytes) {
    if (s.host) (void)hipHostFree(s.host);
    s.host = nullptr; s.cap = 0;
    const size_t cap = align_up(bytes + bytes / 2, 1 << 16);
    PS_HIP(hipHostMalloc(&s.host, cap, hipHostMallocDefault));
    s.cap = cap;
  }
  if (!s.ev) PS_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  memcpy(s.host, src, bytes);
  PS_HIP(hipMemcpyAsync(dst, s.host, bytes, hipMemcpyHostToDevice, st));
  PS_HIP(hipEventRecord(s.ev, st));
  s.used = true;
  return 0;
}

// ---- tile lists in hardware dispatch order --------------------------------------------
// Workgroups of a launch are dealt round-robin over the 8 XCDs (each with its own L2), so
// entry b of a tile list that the kernel indexes with blockIdx.x runs on XCD b % 8.
// deal_to_xcds splits the tasks of a grouped launch (task i = ntile[i] tiles of cost[i]
// each) into units -- a whole task, or a run of its tiles when the launch is too small to
// give every XCD a few tasks -- assigns the units to the XCDs by longest-processing-time-
// first, and orders every XCD's units by descending per-tile cost (cheap tiles fill the
// tail).  A task's tiles stay on one L2 and the 8 XCDs finish together; a contiguous-chunk
// remap of a cost-sorted or heterogeneous list leaves one XCD with all the expensive tiles.
// The caller expands the units into tiles, pads the shorter lists with no-op entries and
// interleaves them: list[j * 8 + x] = j-th tile of XCD x.
constexpr int NXCD = 8;
struct DealUnit {
  int task, first, count;  // tiles [first, first + count) of the task, in the task's own order
  int64_t tile_cost;
};

inline void deal_to_xcds(const std::vector<int>& ntile, const std::vector<int64_t>& cost,
                         std::vector<DealUnit> (&lists)[NXCD]) {
  int64_t total = 0;
  for (int n : ntile) total += n;
  const int unit_max = (int)std::max<int64_t>(1, total / (NXCD * 8));
  std::vector<DealUnit> units;
  for (size_t i = 0; i < ntile.size(); ++i)
    for (int f = 0; f < ntile[i]; f += unit_max)
      units.push_back({(int)i, f, std::min(unit_max, ntile[i] - f), cost[i]});
  std::vector<int> order(units.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return units[a].tile_cost * units[a].count > units[b].tile_cost * units[b].count;
  });
  int64_t load[NXCD] = {0};
  for (int x = 0; x < NXCD; ++x) lists[x].clear();
  for (int u : order) {
    int best = 0;
    for (int x = 1; x < NXCD; ++x)
      if (load[x] < load[best]) best = x;
    lists[best].push_back(units[u]);
    load[best] += units[u].tile_cost * units[u].count;
  }
  for (int x = 0; x < NXCD; ++x)
    std::stable_sort(lists[x].begin(), lists[x].end(),
                     [](const DealUnit& a, const DealUnit& b) { return a.tile_cost > b.tile_cost; });
}

// Interleaves per-XCD tile lists into dispatch order, padding with `noop`.
template <typename Tile>
inline void interleave_xcd_lists(std::vector<Tile> (&lists)[NXCD], const Tile& noop,
                                 std::vector<Tile>& out) {
  size_t longest = 0;
  for (int x = 0; x < NXCD; ++x) longest = std::max(longest, lists[x].size());
  out.assign(longest * NXCD, noop);
  for (int x = 0; x < NXCD; ++x)
    for (size_t j = 0; j < lists[x].size(); ++j) out[j * NXCD + x] = lists[x][j];
}

}  // namespace psh
