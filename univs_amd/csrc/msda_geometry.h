// Host-side tile geometry shared by the LDS-tiled MSDA kernels (msda_tiled.hip, msda_tiled2.hip).
#pragma once
#include <algorithm>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "msda_common.h"

namespace univs {

constexpr int UNIVS_MSDA_WIN_EDGE_MAX = 64;   // window edge limit (the product ww*wh is bounded by the LDS carve)

static inline long long floor_div(long long a, long long b) {  // b > 0
  return (a >= 0) ? a / b : -((-a + b - 1) / b);
}
static inline long long ceil_div(long long a, long long b) {  // b > 0
  return -floor_div(-a, b);
}

// ---- host-side geometry: built once per (device, level shapes, tile parameters); the 32 most recent are kept.
struct GeoKey {
  int dev, L, TH, TW, R, ring;
  long long cap_px;
  int H[UNIVS_MAX_LEVELS], W[UNIVS_MAX_LEVELS];
  bool operator==(const GeoKey& o) const {
    if (dev != o.dev || L != o.L || TH != o.TH || TW != o.TW || R != o.R || ring != o.ring || cap_px != o.cap_px) return false;
    for (int l = 0; l < L; ++l)
      if (H[l] != o.H[l] || W[l] != o.W[l]) return false;
    return true;
  }
};
// Who may still read an entry's device tables: ONE event per stream that launched with it (a record on stream B must not
// overwrite the record that guards stream A's launch), and a pin set when the entry is used while a stream is being
// captured -- a hipGraph keeps raw pointers to the tables and replays at any later time, so a pinned entry is never evicted.
struct GeoUse {
  std::vector<std::pair<hipStream_t, hipEvent_t>> events;
  bool pinned = false;
  void destroy() {
    for (auto& se : events)
      if (se.second) (void)hipEventDestroy(se.second);
    events.clear();
  }
};
struct GeoEntry {
  GeoKey key;
  int4* table;       // device
  int tiles_y, tiles_x;
  long long qmax;    // max queries of a tile
  long long win_px;  // max window pixels of a (tile, level)
  long long lvl_px[UNIVS_MAX_LEVELS];   // ... per level
  GeoUse use;        // per-stream "last launch" events + the capture pin (geo_mark_use)
};

// Lifetime of a cached geometry: callers hold a shared_ptr for the duration of their launch call (an eviction by another host
// thread cannot free the entry under them), every launch records an event on ITS stream (GeoUse), and an evicted entry's device table
// is freed only once that event has completed -- no hipDeviceSynchronize, nothing of another device is touched.  While a
// stream is being captured a cache MISS returns nullptr (hipMalloc / hipMemcpy are illegal there): warm the cache with one
// eager call per geometry before capturing.
static inline std::mutex& geo_use_mutex() {
  static std::mutex mu;
  return mu;
}
template <class E>
static inline void geo_mark_use(const std::shared_ptr<E>& e, hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
  std::lock_guard<std::mutex> lock(geo_use_mutex());
  if (cs != hipStreamCaptureStatusNone) {                       // (no event can be recorded here; the graph outlives this call)
    e->use.pinned = true;
    return;
  }
  hipEvent_t ev = nullptr;
  for (auto& se : e->use.events)
    if (se.first == st) ev = se.second;
  if (!ev) {
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      e->use.pinned = true;                                     // cannot track this stream: keep the tables for good
      return;
    }
    e->use.events.emplace_back(st, ev);
  }
  (void)hipEventRecord(ev, st);
}
template <class E>
static inline bool geo_pinned(const std::shared_ptr<E>& e) {
  std::lock_guard<std::mutex> lock(geo_use_mutex());
  return e->use.pinned;
}
template <class E>
static inline bool geo_idle(const std::shared_ptr<E>& e) {       // nobody holds it, no graph points at it, every stream is done with it
  if (e.use_count() > 1) return false;
  std::lock_guard<std::mutex> lock(geo_use_mutex());
  if (e->use.pinned) return false;
  for (auto& se : e->use.events) {
    const hipError_t q = hipEventQuery(se.second);
    if (q != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
  }
  return true;
}
static inline bool geo_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
  return cs != hipStreamCaptureStatusNone;
}

// one axis of one level: query interval and window interval of tile t
// ring = 1: the window may include the one-pixel zero ring around the level (pixels -1 and Nq: msda_tiled.hip /
// msda_tiled2.hip stage zeros there); ring = 0: the window is clipped to the level (msda_tiled3.hip folds the
// border cases into the corner weights instead).
static void axis_entry(int t, int ntile, int T, int Nq, int Nf, int R, int cap, int ring, int4& e) {
  // queries: pixel centres (i + 0.5) / Nq inside [t*T/Nf, (t+1)*T/Nf)
  long long lo = std::max<long long>(0, ceil_div(2LL * t * T * Nq - Nf, 2LL * Nf));
  long long hi = (t + 1 == ntile) ? Nq : ceil_div(2LL * (t + 1) * T * Nq - Nf, 2LL * Nf);
  hi = std::min<long long>(std::max(hi, lo), Nq);
  // window: bilinear corners of samples within R pixels of the tile's box, clipped to the zero ring
  const long long num1 = std::min<long long>((long long)(t + 1) * T, Nf);
  long long w0 = std::max<long long>(ring ? -1 : 0, floor_div(2LL * t * T * Nq - (1 + 2LL * R) * Nf, 2LL * Nf));
  long long w1 = std::min<long long>(ring ? Nq : Nq - 1, floor_div(2LL * num1 * Nq - (1 - 2LL * R) * Nf, 2LL * Nf) + 1);
  if (!ring) w0 = std::min<long long>(w0, std::max<long long>(Nq - 2, 0));   // two pixels inside the level
  long long wn = std::max<long long>(w1 - w0 + 1, 2);
  wn = std::min<long long>(wn, cap);
  e.x = (int)lo; e.y = (int)(hi - lo); e.z = (int)w0; e.w = (int)wn;
}

static void geo_free(GeoEntry* e) {
  if (!e) return;
  if (e->table) (void)hipFree(e->table);
  e->use.destroy();
  delete e;
}

static std::shared_ptr<GeoEntry> geometry(const LevelTable& lv, int L, int fine, int TH, int TW, int R, long long cap_px,
                                          hipStream_t st, int ring = 1) {
  static std::mutex mu;
  static std::vector<std::shared_ptr<GeoEntry>> cache, retired;
  GeoKey key{};
  if (hipGetDevice(&key.dev) != hipSuccess) return nullptr;
  key.L = L; key.TH = TH; key.TW = TW; key.R = R; key.ring = ring; key.cap_px = cap_px;
  for (int l = 0; l < L; ++l) { key.H[l] = lv.H[l]; key.W[l] = lv.W[l]; }
  std::lock_guard<std::mutex> lock(mu);
  for (size_t i = 0; i < retired.size();)                       // evicted earlier: free what the GPU has finished with
    if (geo_idle(retired[i])) retired.erase(retired.begin() + i);
    else ++i;
  for (const auto& e : cache)
    if (e->key == key) return e;
  if (geo_capturing(st)) return nullptr;

  GeoEntry* ge = new GeoEntry();
  ge->key = key;
  ge->tiles_y = (lv.H[fine] + TH - 1) / TH;
  ge->tiles_x = (lv.W[fine] + TW - 1) / TW;
  std::vector<int4> tab((size_t)L * (ge->tiles_x + ge->tiles_y));
  for (int l = 0; l < L; ++l) {
    for (int tx = 0; tx < ge->tiles_x; ++tx)
      axis_entry(tx, ge->tiles_x, TW, lv.W[l], lv.W[fine], R, UNIVS_MSDA_WIN_EDGE_MAX, ring, tab[(size_t)l * ge->tiles_x + tx]);
    for (int ty = 0; ty < ge->tiles_y; ++ty)
      axis_entry(ty, ge->tiles_y, TH, lv.H[l], lv.H[fine], R, UNIVS_MSDA_WIN_EDGE_MAX, ring,
                 tab[(size_t)L * ge->tiles_x + (size_t)l * ge->tiles_y + ty]);
  }
  // windows must fit the LDS carve: shrink rows where a (tile, level) would not (samples beyond go
  // through the global fallback, results unchanged)
  ge->qmax = 0; ge->win_px = 4;
  for (int l = 0; l < UNIVS_MAX_LEVELS; ++l) ge->lvl_px[l] = 0;
  for (int l = 0; l < L; ++l) {
    int mw = 2, mqx = 0, mqy = 0;
    for (int tx = 0; tx < ge->tiles_x; ++tx) {
      mw = std::max(mw, tab[(size_t)l * ge->tiles_x + tx].w);
      mqx = std::max(mqx, tab[(size_t)l * ge->tiles_x + tx].y);
    }
    for (int ty = 0; ty < ge->tiles_y; ++ty) {
      int4& e = tab[(size_t)L * ge->tiles_x + (size_t)l * ge->tiles_y + ty];
      e.w = (int)std::max<long long>(2, std::min<long long>(e.w, cap_px / mw));
      mqy = std::max(mqy, e.y);
      ge->win_px = std::max<long long>(ge->win_px, (long long)mw * e.w);
      ge->lvl_px[l] = std::max<long long>(ge->lvl_px[l], (long long)mw * e.w);
    }
    ge->qmax += (long long)mqx * mqy;
  }
  ge->table = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&ge->table), tab.size() * sizeof(int4)) != hipSuccess ||
      hipMemcpy(ge->table, tab.data(), tab.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    geo_free(ge);
    return nullptr;
  }
  std::shared_ptr<GeoEntry> sp(ge, geo_free);
  if (cache.size() >= 32) {   // bounded (image datasets: many resolutions): retire the oldest unpinned entry OF THIS DEVICE
    for (size_t i = 0; i < cache.size(); ++i)
      if (cache[i]->key.dev == key.dev && !geo_pinned(cache[i])) {
        retired.push_back(cache[i]);
        cache.erase(cache.begin() + i);
        break;
      }
  }
  cache.push_back(sp);
  return sp;
}


}  // namespace univs
