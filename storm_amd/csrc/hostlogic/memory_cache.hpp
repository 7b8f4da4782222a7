// memory_cache.hpp -- the library's process-wide cache of device blocks, pinned host blocks and streams (slab arenas with
// best-fit carving and coalescing, size-class lists, footprint limits).  Host-only bookkeeping: every call into the GPU
// runtime goes through `Backend`, so the same code is compiled by hipcc into libstorm_hip.so (HipBackend, storm_hip.hip) and
// by g++ -fsanitize=address,undefined / -fsanitize=thread into the CPU stress test (tests/sanitize/hostlogic_stress.cpp,
// `make -C storm_amd/csrc sanitize`) over a malloc-backed stand-in for the runtime.
//
// Backend (all static):
//   using Stream = ...;                               a null-comparable handle
//   void* alloc(bool pinned, size_t bytes);           nullptr on failure (any sticky runtime error already cleared)
//   void release(bool pinned, void* p);               may synchronise the device
//   [[noreturn]] void alloc_failed(bool pinned, size_t bytes);   throws the library's failure
//   size_t device_total(int dev);                     0 if unknown
//   int current_device(); void set_device(int dev);
//   Stream stream_create();  (throws on failure)      void stream_destroy(Stream);
#pragma once
#include <algorithm>
#include <cstddef>
#include <iterator>
#include <map>
#include <mutex>
#include <vector>

namespace storm {

// Process-wide cache of device buffers, pinned host buffers and streams.  hipFree / hipHostFree / hipStreamDestroy
// synchronise the whole device and hipMalloc / hipHostMalloc cost 0.1-1 ms on an idle device (far more on a busy one, see Arena): with one handle per small scan (the
// rotation sweeps of the reference's batch scripts run thousands of 100-300 frequency scans, each on its own system)
// they serialised concurrent scans (scripts/many_scans_bench.py).  Blocks are kept by size and handed to the next
// handle that asks; only what exceeds the cache limit goes back to the runtime.  A block is given back only after
// the stream that used it has been synchronised (see DevBuf::reserve and the destroy functions).
// A slab of device memory that large blocks are carved from.  With kernels of other handles in flight a hipMalloc took
// 30 ms - 5.5 s on the box whatever its size (the 1001-rate batch of tools/storm_batch with four groups in flight: a run
// that met a dozen of them took 3-7 s instead of 2.1 s; this is also the multi-second stall DESIGN.md section 8 recorded
// in round 1) and a hipFree waits for the whole device, so blocks of 64 MiB and more never reach the runtime one by
// one: the first request allocates a slab while the device is still idle and later ones grow the set by doubling.
struct Arena {
  char* base = nullptr;
  size_t size = 0, free_bytes = 0;
  std::map<size_t, size_t> holes;  // offset -> length, neighbours coalesced
  bool contains(const void* p) const { return p >= base && p < base + size; }
  void* carve(size_t bytes) {  // best fit
    auto best = holes.end();
    for (auto it = holes.begin(); it != holes.end(); ++it)
      if (it->second >= bytes && (best == holes.end() || it->second < best->second)) best = it;
    if (best == holes.end()) return nullptr;
    const size_t off = best->first, len = best->second;
    holes.erase(best);
    if (len > bytes) holes.emplace(off + bytes, len - bytes);
    free_bytes -= bytes;
    return base + off;
  }
  void put_back(void* p, size_t bytes) {
    size_t off = size_t(static_cast<char*>(p) - base), len = bytes;
    auto next = holes.lower_bound(off);
    if (next != holes.begin()) {
      auto prev = std::prev(next);
      if (prev->first + prev->second == off) { off = prev->first; len += prev->second; holes.erase(prev); }
    }
    if (next != holes.end() && off + len == next->first) { len += next->second; holes.erase(next); }
    holes.emplace(off, len);
    free_bytes += bytes;
  }
};

template <class Backend>
class BasicMemoryCache {
 public:
  using Stream = typename Backend::Stream;
  static BasicMemoryCache& get() { static BasicMemoryCache m; return m; }
  static constexpr size_t kMedium = size_t(1) << 20;  // blocks from this size on live in arenas ...
  static constexpr size_t kLarge = size_t(64) << 20;  // ... those from this size on in slabs of their own
  static size_t size_class(size_t bytes) {
    if (bytes <= 256) return 256;
    if (bytes < kMedium) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
    if (bytes < kLarge) return (bytes + kMedium - 1) >> 20 << 20;
    // large blocks: eight classes per power of two, so that the slightly different sizes successive handles ask for
    // (a sweep's staging follows its mode count) leave holes the next one fits
    size_t p2 = size_t(1) << 26;
    while ((p2 << 1) <= bytes) p2 <<= 1;
    const size_t step = p2 >> 3;
    return (bytes + step - 1) / step * step;
  }
  void* take(bool pinned, int dev, size_t bytes, size_t& granted) {
    granted = size_class(bytes);
    if (granted < kMedium) {  // small blocks are recycled by size class; a miss is carved out of a slab as well
      std::lock_guard<std::mutex> lock(mu_);
      auto& m = blocks_[key(pinned, dev)];
      auto it = m.find(granted);
      if (it != m.end()) {
        void* p = it->second;
        cached_[pinned] -= granted;
        if (!pinned) in_use_[dev] += granted;
        m.erase(it);
        return p;
      }
    }
    if (void* p = take_from_arenas(pinned, dev, granted)) return p;
    // no arenas (fixed limit) or no room for another slab: a block of its own, recycled by size
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto& m = blocks_[key(pinned, dev)];
      auto it = m.lower_bound(granted);
      if (it != m.end() && it->first <= granted + granted / 2 + (size_t(1) << 20)) {  // close enough in size
        void* p = it->second;
        granted = it->first;
        cached_[pinned] -= granted;
        if (!pinned) in_use_[dev] += granted;
        m.erase(it);
        return p;
      }
    }
    void* p = Backend::alloc(pinned, granted);
    if (!p) {  // make room and try once more
      trim();
      p = Backend::alloc(pinned, granted);
    }
    if (!p) Backend::alloc_failed(pinned, granted);  // throws
    if (!pinned) { std::lock_guard<std::mutex> lock(mu_); in_use_[dev] += granted; }
    return p;
  }
  void give(bool pinned, int dev, void* p, size_t granted) {
    if (!p) return;
    void* dead_arena = nullptr;
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (!pinned) {
        size_t& used = in_use_[dev];
        used = used >= granted ? used - granted : 0;
      }
      const bool fixed = limit_[pinned] != kAutoLimit;
      Arena* home = nullptr;
      std::vector<Arena>* home_list = nullptr;
      for (int band = 0; band < kBands && !home; ++band) {
        auto& av = arenas_[arena_key(pinned, dev, band)];
        for (auto& a : av)
          if (a.contains(p)) { home = &a; home_list = &av; break; }
      }
      // Small blocks wait in the size-class lists, as do blocks that have no slab; the others go back into their slab.
      const bool listed = granted < kMedium || !home;
      if (listed) {
        bool keep;
        if (fixed) keep = cached_[pinned] + granted <= limit_[pinned];
        else if (home) keep = true;
        else if (pinned) keep = cached_[1] + granted <= (size_t(2) << 30);
        // keep while the library's footprint on the device stays below three quarters of its memory
        else keep = in_use_[dev] + cached_[0] + arena_bytes_locked(false, dev) + granted <= device_total_locked(dev) / 4 * 3;
        if (keep) {
          blocks_[key(pinned, dev)].emplace(granted, p);
          cached_[pinned] += granted;
          return;
        }
      }
      if (home) {
        home->put_back(p, granted);
        p = nullptr;
        // with a fixed limit (storm_memory_cache_limit) an empty slab goes back as soon as it exceeds the limit
        if (fixed && home->free_bytes == home->size && cached_[pinned] + arena_free_locked(pinned) > limit_[pinned]) {
          dead_arena = home->base;
          home_list->erase(home_list->begin() + (home - home_list->data()));
        }
      }
    }
    if (dead_arena) Backend::release(pinned, dead_arena);
    if (!p) return;
    Backend::release(pinned, p);
  }
  Stream take_stream(int dev) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto& v = streams_[dev];
      if (!v.empty()) { Stream st = v.back(); v.pop_back(); return st; }
    }
    return Backend::stream_create();  // throws on failure
  }
  void give_stream(int dev, Stream st) {  // st must be idle
    if (!st) return;
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto& v = streams_[dev];
      if (v.size() < 64) { v.push_back(st); return; }
    }
    Backend::stream_destroy(st);
  }
  // Returns every cached block, every arena nothing is carved from and every idle stream to the runtime (each device's
  // with that device current).
  void trim() { release_idle(true); }
  void set_limit(bool pinned, size_t bytes) {
    bool over;
    {
      std::lock_guard<std::mutex> lock(mu_);
      limit_[pinned] = bytes;
      over = cached_[pinned] + arena_free_locked(pinned) > bytes;
    }
    if (over) trim();  // (gives back idle blocks of both kinds; what is in use follows when it is released)
  }
  size_t cached(bool pinned) {
    std::lock_guard<std::mutex> lock(mu_);
    return cached_[pinned] + arena_free_locked(pinned);
  }

 private:
  static int key(bool pinned, int dev) { return dev * 2 + (pinned ? 1 : 0); }
  // arenas_: keys 0 .. 2 = pinned host memory (band 0 .. 2); 3 + 3 * device + band = device memory
  static constexpr int kBands = 3;  // below 1 MiB, [1, 64) MiB, 64 MiB and more: each band has slabs of its own
  static int band_of(size_t granted) { return granted >= kLarge ? 2 : granted >= kMedium ? 1 : 0; }
  static int arena_key(bool pinned, int dev, int band) { return pinned ? band : kBands + dev * kBands + band; }
  static bool arena_key_pinned(int k) { return k < kBands; }
  static int arena_key_dev(int k) { return k < kBands ? 0 : (k - kBands) / kBands; }
  // idle slab memory: what a trim would hand back (slabs nothing is carved from; the holes of a slab in use stay)
  size_t arena_free_locked(bool pinned) const {
    size_t f = 0;
    for (auto& kv : arenas_)
      if (arena_key_pinned(kv.first) == pinned)
        for (auto& a : kv.second) if (a.free_bytes == a.size) f += a.size;
    return f;
  }
  size_t arena_bytes_locked(bool pinned, int dev) {
    size_t f = 0;
    for (int band = 0; band < kBands; ++band) for (auto& a : arenas_[arena_key(pinned, dev, band)]) f += a.size;
    return f;
  }
  size_t device_total_locked(int dev) {
    auto it = total_.find(dev);
    if (it == total_.end()) {
      size_t bytes = Backend::device_total(dev);
      if (!bytes) bytes = size_t(64) << 30;
      it = total_.emplace(dev, bytes).first;
    }
    return it->second;
  }
  // A block out of the arenas (device memory of `dev`, the current device, or pinned host memory), growing them if need
  // be; nullptr: the caller falls back to a block of its own (arenas are off under a fixed limit, or there is no room for
  // another slab).  Three bands with slabs of their own, so that the many small and medium blocks of short-lived handles
  // do not fragment the slabs the multi-GB staging blocks need.
  void* take_from_arenas(bool pinned, int dev, size_t granted) {
    const int band = band_of(granted);
    const int ak = arena_key(pinned, dev, band);
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (limit_[pinned] != kAutoLimit) return nullptr;
      for (auto& a : arenas_[ak])
        if (void* p = a.carve(granted)) { if (!pinned) in_use_[dev] += granted; return p; }
    }
    std::lock_guard<std::mutex> grow(grow_mu_);  // one slab at a time; whoever waited here looks again first
    size_t want = 0;
    {
      std::lock_guard<std::mutex> lock(mu_);
      for (auto& a : arenas_[ak])
        if (void* p = a.carve(granted)) { if (!pinned) in_use_[dev] += granted; return p; }
      // Slab sizes.  Fresh memory is not free: when the driver has to clear it first (memory another process released
      // shortly before) hipMalloc / hipHostMalloc take ~30 ms per GB -- a 144 GB slab was measured at 4.1-4.4 s, at
      // 0.5 ms when the memory was clean -- and every stream of the process stalls meanwhile.  So slabs follow the need:
      // the first one is small, later ones double up to a ceiling, and the request itself sets the minimum (twice its
      // size for the large band).  Device: 4 GiB up to 32 GiB (blocks of 64 MiB and more), 256 MiB up to 2 GiB (medium),
      // 16 MiB up to 256 MiB (small); never past 3/4 of the device in all.  Pinned host: 1 GiB up to 8 GiB, 128 MiB up
      // to 1 GiB, 8 MiB up to 128 MiB; 64 GiB in all.
      const size_t total = pinned ? size_t(64) << 30 : device_total_locked(dev);
      const size_t have = arena_bytes_locked(pinned, dev);
      const auto& av = arenas_[ak];
      const size_t kFirst[2][kBands] = {{size_t(16) << 20, size_t(256) << 20, size_t(4) << 30}, {size_t(8) << 20, size_t(128) << 20, size_t(1) << 30}};
      const size_t kTop[2][kBands] = {{size_t(256) << 20, size_t(2) << 30, size_t(32) << 30}, {size_t(128) << 20, size_t(1) << 30, size_t(8) << 30}};
      const size_t first = kFirst[pinned][band], top = kTop[pinned][band];
      want = av.empty() ? first : std::min(av.back().size * 2, top);
      want = std::max(want, band == 2 ? granted * 2 : granted + granted / 4);
      want = (want + (size_t(2) << 20) - 1) >> 21 << 21;
      const size_t cap = pinned ? total : total / 4 * 3;
      if (have + cached_[pinned] + want > cap) want = granted;
      if (have + want > cap) return nullptr;
    }
    void* base = nullptr;
    while (!(base = Backend::alloc(pinned, want))) {
      if (want <= granted) return nullptr;
      want = std::max(granted, want / 2);
    }
    std::lock_guard<std::mutex> lock(mu_);
    Arena a;
    a.base = static_cast<char*>(base); a.size = want; a.free_bytes = want;
    a.holes.emplace(0, want);
    void* p = a.carve(granted);
    arenas_[ak].push_back(std::move(a));
    if (!pinned) in_use_[dev] += granted;
    return p;
  }
  // All cached blocks and all empty arenas go back to the runtime.
  void release_idle(bool everything) {
    std::map<int, std::multimap<size_t, void*>> blocks;
    std::map<int, std::vector<void*>> slabs;
    std::map<int, std::vector<Stream>> streams;
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (everything) {
        streams.swap(streams_);   // idle streams go back as well (found by the leak check of the CPU sanitizer build)
        blocks.swap(blocks_);
        cached_[0] = cached_[1] = 0;
        // listed blocks that were carved out of a slab go back into it
        for (auto& kv : blocks) {
          const bool pinned = kv.first & 1;
          const int dev = kv.first >> 1;
          for (auto it = kv.second.begin(); it != kv.second.end();) {
            Arena* home = nullptr;
            for (int band = 0; band < kBands && !home; ++band)
              for (auto& a : arenas_[arena_key(pinned, dev, band)])
                if (a.contains(it->second)) { home = &a; break; }
            if (home) { home->put_back(it->second, it->first); it = kv.second.erase(it); }
            else ++it;
          }
        }
      }
      for (auto& kv : arenas_) {
        const bool pinned = arena_key_pinned(kv.first);
        auto& av = kv.second;
        for (size_t k = 0; k < av.size();) {
          const bool idle = av[k].free_bytes == av[k].size;
          const bool over = limit_[pinned] != kAutoLimit && cached_[pinned] + arena_free_locked(pinned) > limit_[pinned];
          if (idle && (everything || over)) { slabs[kv.first].push_back(av[k].base); av.erase(av.begin() + k); }
          else ++k;
        }
      }
    }
    const int prev = Backend::current_device();
    for (auto& kv : streams) {
      Backend::set_device(kv.first);
      for (Stream st : kv.second) Backend::stream_destroy(st);
    }
    for (auto& kv : blocks) {
      const bool pinned = kv.first & 1;
      if (!pinned) Backend::set_device(kv.first >> 1);
      for (auto& b : kv.second) Backend::release(pinned, b.second);
    }
    for (auto& kv : slabs) {
      const bool pinned = arena_key_pinned(kv.first);
      if (!pinned) Backend::set_device(arena_key_dev(kv.first));
      for (void* p : kv.second) Backend::release(pinned, p);
    }
    Backend::set_device(prev);
  }
  std::mutex mu_, grow_mu_;
  std::map<int, std::multimap<size_t, void*>> blocks_;
  std::map<int, std::vector<Stream>> streams_;
  std::map<int, std::vector<Arena>> arenas_;       // by arena_key(pinned, device, band)
  static constexpr size_t kAutoLimit = ~size_t(0);
  size_t cached_[2] = {0, 0};                      // small blocks kept for reuse: device (all devices together), pinned host
  std::map<int, size_t> in_use_, total_;           // per device: bytes handed out, memory of the device
  size_t limit_[2] = {kAutoLimit, kAutoLimit};     // footprint rules above until storm_memory_cache_limit sets figures
};

}  // namespace storm
