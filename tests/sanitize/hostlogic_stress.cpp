// hostlogic_stress.cpp -- CPU stress test of the library's host-only logic (storm_amd/csrc/hostlogic/) under the sanitizers:
//   * the slab allocator / block cache (memory_cache.hpp) over a stand-in for the GPU runtime that hands out address ranges
//     (PROT_NONE reservations: the cache only does bookkeeping, a dereference would fault) and fails on demand:
//     randomised take / give / trim / set_limit across all three size bands from 8 threads, with an overlap check on every
//     block handed out and a leak check at the end;
//   * the bracket state machine (brackets.hpp) on an analytic discriminant, error paths included;
//   * the multi-device orchestration (multi.hpp: worker threads, slice + halo, ordered gather, cyclic deal, error hand-over)
//     over scalar stand-in "devices", bit-compared with the one-device result, several handles at once.
// Built twice by tests/sanitize/Makefile: -fsanitize=address,undefined and -fsanitize=thread.  Never runs on a GPU box.
#include <sys/mman.h>

#include <atomic>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../storm_amd/csrc/hostlogic/brackets.hpp"
#include "../../storm_amd/csrc/hostlogic/failure.hpp"
#include "../../storm_amd/csrc/hostlogic/memory_cache.hpp"
#include "../../storm_amd/csrc/hostlogic/multi.hpp"

using namespace storm;

#define CHECK(cond)                                                                    \
  do {                                                                                 \
    if (!(cond)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); std::abort(); } \
  } while (0)

// ------------------------------------------------------------------------------------------------ the fake runtime
struct FakeRuntime {
  using Stream = int*;
  static std::mutex mu;
  static std::map<char*, size_t> live;   // reservations handed to the cache
  static size_t device_bytes, pinned_bytes, device_capacity;
  static std::atomic<long> alloc_calls, fail_every, streams_live;
  static thread_local int device;

  static void* alloc(bool pinned, size_t bytes) {
    const long call = ++alloc_calls;
    const long every = fail_every.load();
    if (every > 0 && call % every == 0) return nullptr;   // the runtime says no now and then
    std::lock_guard<std::mutex> lock(mu);
    if (!pinned && device_bytes + bytes > device_capacity) return nullptr;
    void* p = mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    live.emplace(static_cast<char*>(p), bytes);
    (pinned ? pinned_bytes : device_bytes) += bytes;
    return p;
  }
  static void release(bool pinned, void* p) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = live.find(static_cast<char*>(p));
    CHECK(it != live.end());                              // only what alloc() returned, and only once
    (pinned ? pinned_bytes : device_bytes) -= it->second;
    munmap(p, it->second);
    live.erase(it);
  }
  [[noreturn]] static void alloc_failed(bool, size_t bytes) { fail(STORM_ERR_HIP, "fake runtime: out of memory (" + std::to_string(bytes) + ")"); }
  static size_t device_total(int) { return device_capacity; }
  static int current_device() { return device; }
  static void set_device(int d) { device = d; }
  static Stream stream_create() { ++streams_live; return new int(0); }
  static void stream_destroy(Stream s) { --streams_live; delete s; }
};
std::mutex FakeRuntime::mu;
std::map<char*, size_t> FakeRuntime::live;
size_t FakeRuntime::device_bytes = 0, FakeRuntime::pinned_bytes = 0, FakeRuntime::device_capacity = size_t(24) << 30;
std::atomic<long> FakeRuntime::alloc_calls{0}, FakeRuntime::fail_every{0}, FakeRuntime::streams_live{0};
thread_local int FakeRuntime::device = 0;
using Cache = BasicMemoryCache<FakeRuntime>;

// every block the cache hands out, to prove that no two live blocks overlap
struct Ledger {
  std::mutex mu;
  std::map<char*, size_t> blocks;
  void add(void* p, size_t bytes) {
    std::lock_guard<std::mutex> lock(mu);
    char* b = static_cast<char*>(p);
    auto next = blocks.lower_bound(b);
    if (next != blocks.end()) CHECK(b + bytes <= next->first);
    if (next != blocks.begin()) { auto prev = std::prev(next); CHECK(prev->first + prev->second <= b); }
    blocks.emplace(b, bytes);
  }
  void remove(void* p) {
    std::lock_guard<std::mutex> lock(mu);
    CHECK(blocks.erase(static_cast<char*>(p)) == 1);
  }
};

static void arena_unit() {
  Arena a;
  std::vector<char> backing(1 << 16);
  a.base = backing.data(); a.size = a.free_bytes = backing.size();
  a.holes.emplace(0, a.size);
  void* p1 = a.carve(4096); void* p2 = a.carve(8192); void* p3 = a.carve(4096);
  CHECK(p1 && p2 && p3 && a.free_bytes == a.size - 16384);
  a.put_back(p2, 8192);
  CHECK(a.holes.size() == 2);
  void* p4 = a.carve(8192);                 // best fit: the hole p2 left, not the big tail
  CHECK(p4 == p2);
  a.put_back(p1, 4096); a.put_back(p3, 4096); a.put_back(p4, 8192);
  CHECK(a.holes.size() == 1 && a.holes.begin()->second == a.size && a.free_bytes == a.size);   // all coalesced
  CHECK(a.carve(a.size + 1) == nullptr);
}

static void cache_stress(unsigned seed, int n_threads, int iters) {
  Cache& cache = Cache::get();
  Ledger ledger;
  std::atomic<long> failures{0};
  auto worker = [&](int tid) {
    std::mt19937_64 rng(seed * 977u + unsigned(tid));
    FakeRuntime::set_device(tid % 2);
    struct Held { void* p; size_t granted; bool pinned; int dev; };
    std::vector<Held> held;
    for (int it = 0; it < iters; ++it) {
      const unsigned op = unsigned(rng() % 100);
      if (op < 55 || held.empty()) {
        // sizes across the three bands: < 1 MiB, [1, 64) MiB, >= 64 MiB
        const unsigned band = unsigned(rng() % 10);
        size_t bytes;
        if (band < 6) bytes = 1 + size_t(rng() % ((size_t(1) << 20) - 1));
        else if (band < 9) bytes = (size_t(1) << 20) + size_t(rng() % (size_t(48) << 20));
        else bytes = (size_t(64) << 20) + size_t(rng() % (size_t(400) << 20));
        const bool pinned = (rng() % 4) == 0;
        const int dev = FakeRuntime::current_device();
        size_t granted = 0;
        void* p = nullptr;
        try {
          p = cache.take(pinned, dev, bytes, granted);
        } catch (const Failure& f) {
          CHECK(f.code == STORM_ERR_HIP);
          ++failures;
          continue;
        }
        CHECK(p != nullptr && granted >= bytes);
        ledger.add(p, granted);
        held.push_back(Held{p, granted, pinned, dev});
      } else if (op < 93) {
        const size_t k = size_t(rng() % held.size());
        ledger.remove(held[k].p);
        cache.give(held[k].pinned, held[k].dev, held[k].p, held[k].granted);
        held[k] = held.back(); held.pop_back();
      } else if (op < 95) {
        cache.trim();
      } else if (op < 97) {
        // a fixed limit for a while, then the automatic rules again
        cache.set_limit(false, (rng() % 2) ? (size_t(1) << 30) : ~size_t(0));
      } else {
        auto st = cache.take_stream(tid % 2);
        CHECK(st != nullptr);
        cache.give_stream(tid % 2, st);
        (void)cache.cached(false); (void)cache.cached(true);
      }
    }
    for (auto& h : held) { ledger.remove(h.p); cache.give(h.pinned, h.dev, h.p, h.granted); }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker, t);
  for (auto& th : pool) th.join();
  CHECK(ledger.blocks.empty());
  cache.set_limit(false, ~size_t(0));
  cache.set_limit(true, ~size_t(0));
  cache.trim();
  // nothing is handed out any more: after a trim every slab and every cached block is back with the runtime
  CHECK(FakeRuntime::live.empty() && FakeRuntime::device_bytes == 0 && FakeRuntime::pinned_bytes == 0);
  std::printf("  cache stress seed %u: %ld runtime allocations, %ld refused requests, clean\n", seed, FakeRuntime::alloc_calls.load(), failures.load());
}

// ------------------------------------------------------------------------------------------------ brackets
static double toy_det(double x, double shift, double wiggle = 1.0) {
  return std::sin(3.14159265358979323846 * wiggle * (x - shift)) * std::exp(x / 7.0);
}

static void brackets_unit() {
  const size_t n = 400;
  std::vector<double> f(n), d(n);
  for (size_t i = 0; i < n; ++i) { f[i] = 0.25 + 20.0 * double(i) / double(n - 1); d[i] = toy_det(f[i], 0.0); }
  d[100] = std::nan("");                        // a non-finite scan value: its pairs are dropped and flagged
  std::vector<BracketState> br;
  std::vector<uint64_t> idx;
  bool invalid = false;
  filter_sign_swaps(f.data(), d.data(), n, br, idx, invalid);
  CHECK(invalid && br.size() == 20 && idx.size() == 20);
  for (uint64_t ulp : {uint64_t(1), uint64_t(45035996)}) {
    std::vector<BracketState> b2 = br;
    size_t rounds = 0;
    refine_brackets(b2, ulp, [&](const double* xs, size_t m, double* fs) {
      ++rounds;
      for (size_t k = 0; k < m; ++k) fs[k] = toy_det(xs[k], 0.0);
    });
    CHECK(rounds > 2 && rounds < 200);
    for (size_t k = 0; k < b2.size(); ++k) {
      CHECK(b2[k].done && b2[k].evals >= 1);
      CHECK(std::fabs(b2[k].root - std::round(b2[k].root)) < (ulp == 1 ? 1e-13 : 1e-6));
      CHECK(b2[k].lower.x <= b2[k].root && b2[k].root <= b2[k].upper.x);
    }
  }
  // error paths: a NaN during refinement, a bracket that is none, a precision of zero
  auto code_of = [&](auto&& fn) { try { fn(); } catch (const Failure& e) { return e.code; } return int(STORM_OK); };
  std::vector<BracketState> b3 = br;
  CHECK(code_of([&] { refine_brackets(b3, 1, [](const double*, size_t m, double* fs) { for (size_t k = 0; k < m; ++k) fs[k] = std::nan(""); }); }) == STORM_ERR_NON_FINITE);
  std::vector<BracketState> b4(1);
  b4[0].lower = Pt{2.0, 1.0}; b4[0].upper = Pt{1.0, -1.0};
  CHECK(code_of([&] { refine_brackets(b4, 1, [](const double*, size_t, double*) {}); }) == STORM_ERR_BRACKET);
  CHECK(code_of([&] { refine_brackets(b3, 0, [](const double*, size_t, double*) {}); }) == STORM_ERR_INVALID_ARGUMENT);
  // the split: every adjacent pair owned once
  for (size_t nn : {size_t(0), size_t(1), size_t(2), size_t(7), size_t(64), size_t(1001)})
    for (size_t parts : {size_t(1), size_t(2), size_t(3), size_t(8), size_t(13)}) {
      std::vector<int> owned(nn ? nn - 1 : 0, 0);
      size_t prev_hi = 0;
      for (size_t k = 0; k < parts; ++k) {
        size_t lo, hi;
        shard_bounds(nn, parts, k, lo, hi);
        CHECK(lo == prev_hi && lo <= hi && hi <= nn);
        prev_hi = hi;
        for (size_t i = lo; i + 1 < std::min(hi + 1, nn); ++i) owned[i] += 1;
      }
      CHECK(prev_hi == nn);
      for (int c : owned) CHECK(c == 1);
    }
}

// ------------------------------------------------------------------------------------------------ multi
struct ToySolver { double shift; size_t n_int; int fail_in = 0; /* 1: scan, 2: optimize, 3: eigenvectors */ double wiggle = 1.0; };
static thread_local std::string toy_error;
struct ToyOps {
  using Solver = ToySolver*;
  static int scan_brackets(Solver s, const double* freq, size_t n, storm_bracket* out, size_t cap, size_t* n_out, int* invalid,
                           double* det_out, uint64_t* index_out) {
    if (s->fail_in == 1) { toy_error = "toy scan failure"; return STORM_ERR_HIP; }
    std::vector<double> d(n);
    for (size_t i = 0; i < n; ++i) d[i] = toy_det(freq[i], s->shift, s->wiggle);
    if (det_out) std::copy(d.begin(), d.end(), det_out);
    std::vector<BracketState> br;
    std::vector<uint64_t> idx;
    bool inv = false;
    filter_sign_swaps(freq, d.data(), n, br, idx, inv);
    *invalid = inv;
    *n_out = br.size();
    if (br.size() > cap) { toy_error = "cap"; return STORM_ERR_CAPACITY; }
    for (size_t k = 0; k < br.size(); ++k) {
      out[k] = storm_bracket{br[k].lower.x, br[k].lower.f, br[k].upper.x, br[k].upper.f, std::nan(""), 0};
      index_out[k] = idx[k];
    }
    return STORM_OK;
  }
  static int optimize_brackets(Solver s, storm_bracket* b, size_t n, uint64_t ulp) {
    if (s->fail_in == 2) { toy_error = "toy refine failure"; return STORM_ERR_NON_FINITE; }
    std::vector<BracketState> br(n);
    for (size_t k = 0; k < n; ++k) { br[k].lower = Pt{b[k].lo_x, b[k].lo_f}; br[k].upper = Pt{b[k].hi_x, b[k].hi_f}; }
    try {
      refine_brackets(br, ulp, [&](const double* xs, size_t m, double* fs) { for (size_t k = 0; k < m; ++k) fs[k] = toy_det(xs[k], s->shift, s->wiggle); });
    } catch (const Failure& f) { toy_error = f.msg; return f.code; }
    for (size_t k = 0; k < n; ++k)
      b[k] = storm_bracket{br[k].lower.x, br[k].lower.f, br[k].upper.x, br[k].upper.f, br[k].root, br[k].evals};
    return STORM_OK;
  }
  static int scan_and_optimize(Solver s, const double* freq, size_t n, uint64_t ulp, storm_bracket* out, size_t cap, size_t* n_out,
                               int* invalid, double* det_out, uint64_t* index_out) {
    const int rc = scan_brackets(s, freq, n, out, cap, n_out, invalid, det_out, index_out);
    return rc != STORM_OK ? rc : optimize_brackets(s, out, *n_out, ulp);
  }
  static int eigenvector_batch(Solver s, const double* freq, size_t n, double* det_out, double* vec_out) {
    if (s->fail_in == 3) { toy_error = "toy eigenvector failure"; return STORM_ERR_SINGULAR; }
    const size_t len = (s->n_int + 1) * 4;
    for (size_t k = 0; k < n; ++k) {
      det_out[k] = toy_det(freq[k], s->shift, s->wiggle);
      for (size_t i = 0; i < len; ++i) vec_out[k * len + i] = freq[k] * double(i + 1);
    }
    return STORM_OK;
  }
  static size_t n_int(Solver s) { return s->n_int; }
  static std::string last_error() { return toy_error; }
};

static bool same_bits(double a, double b) { return bits_of(a) == bits_of(b); }

static void multi_unit(size_t n, size_t D, uint64_t ulp, double wiggle = 1.0) {
  std::vector<double> f(n);
  for (size_t i = 0; i < n; ++i) f[i] = 0.25 + 30.0 * double(i) / double(n > 1 ? n - 1 : 1);
  ToySolver one{0.125, 37};
  one.wiggle = wiggle;
  std::vector<ToySolver> devs(D, one);
  std::vector<ToySolver*> ptr1{&one}, ptrD;
  for (auto& d : devs) ptrD.push_back(&d);
  auto run = [&](const std::vector<ToySolver*>& sv, bool deal, std::vector<storm_bracket>& out, std::vector<uint64_t>& idx, std::vector<double>& det, int& inv) {
    out.assign(n ? n : 1, storm_bracket{}); idx.assign(n ? n : 1, 0); det.assign(n, -7.0);
    size_t found = 0;
    multi_scan_and_optimize<ToyOps>(sv, f.data(), n, ulp, out.data(), out.size(), &found, &inv, det.data(), idx.data(), deal);
    out.resize(found); idx.resize(found);
  };
  std::vector<storm_bracket> a, b;
  std::vector<uint64_t> ia, ib;
  std::vector<double> da, db;
  int inva = 0, invb = 0;
  run(ptr1, true, a, ia, da, inva);
  for (bool deal : {true, false}) {   // dealt cyclically | refined where found: the one-device result either way
    run(ptrD, deal, b, ib, db, invb);
    CHECK(a.size() == b.size() && ia == ib && inva == invb);
    for (size_t k = 0; k < a.size(); ++k)
      CHECK(same_bits(a[k].root, b[k].root) && same_bits(a[k].lo_x, b[k].lo_x) && same_bits(a[k].hi_f, b[k].hi_f) && a[k].evals == b[k].evals);
    for (size_t i = 0; i < n; ++i) CHECK(same_bits(da[i], db[i]));
  }
  if (!a.empty()) {
    std::vector<double> roots(a.size()), d1(a.size()), d2(a.size()), v1(a.size() * 38 * 4), v2(a.size() * 38 * 4);
    for (size_t k = 0; k < a.size(); ++k) roots[k] = a[k].root;
    multi_eigenvector_batch<ToyOps>(ptr1, roots.data(), roots.size(), d1.data(), v1.data());
    multi_eigenvector_batch<ToyOps>(ptrD, roots.data(), roots.size(), d2.data(), v2.data());
    CHECK(d1 == d2 && v1 == v2);
  }
}

static void multi_errors() {
  std::vector<double> f(200);
  for (size_t i = 0; i < f.size(); ++i) f[i] = 0.25 + 0.1 * double(i);
  for (int where = 1; where <= 3; ++where) {
    std::vector<ToySolver> devs(4, ToySolver{0.0, 9});
    devs[2].fail_in = where;
    std::vector<ToySolver*> ptr;
    for (auto& d : devs) ptr.push_back(&d);
    std::vector<storm_bracket> out(f.size());
    std::vector<uint64_t> idx(f.size());
    size_t found = 0;
    int inv = 0;
    try {
      multi_scan_and_optimize<ToyOps>(ptr, f.data(), f.size(), 1, out.data(), out.size(), &found, &inv, nullptr, idx.data(), where != 2);
      CHECK(where == 3);
      std::vector<double> d(found), v(found * 40);
      std::vector<double> roots(found);
      for (size_t k = 0; k < found; ++k) roots[k] = out[k].root;
      multi_eigenvector_batch<ToyOps>(ptr, roots.data(), found, d.data(), v.data());
      CHECK(false);
    } catch (const Failure& e) {
      CHECK(e.code == (where == 1 ? STORM_ERR_HIP : where == 2 ? STORM_ERR_NON_FINITE : STORM_ERR_SINGULAR));
      CHECK(e.msg.find("device 2: toy") != std::string::npos);   // the failing thread's own message reached the caller
    }
  }
  // a bracket buffer that is too small
  ToySolver one{0.0, 9};
  std::vector<ToySolver*> ptr{&one, &one, &one};
  std::vector<storm_bracket> out(3);
  std::vector<uint64_t> idx(3);
  size_t found = 0;
  int inv = 0;
  try {
    multi_scan_and_optimize<ToyOps>(ptr, f.data(), f.size(), 1, out.data(), out.size(), &found, &inv, nullptr, idx.data());
    CHECK(false);
  } catch (const Failure& e) { CHECK(e.code == STORM_ERR_CAPACITY && found > 3); }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 4000;
  std::printf("hostlogic stress\n");
  arena_unit();
  for (unsigned seed = 1; seed <= 3; ++seed) {
    FakeRuntime::fail_every = seed == 2 ? 5 : 0;            // the second pass with a runtime that refuses every fifth request
    FakeRuntime::device_capacity = seed == 3 ? size_t(3) << 30 : size_t(24) << 30;   // the third with a small device
    cache_stress(seed, 8, iters);
  }
  FakeRuntime::fail_every = 0;
  CHECK(FakeRuntime::streams_live.load() >= 0);
  brackets_unit();
  std::printf("  brackets: ok\n");
  for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(5), size_t(333), size_t(2000)})
    for (size_t D : {size_t(1), size_t(2), size_t(3), size_t(5), size_t(8)}) multi_unit(n, D, n % 2 ? 1 : 45035996);
  multi_unit(60000, 2, 45035996, 700.0);   // ~21 000 brackets: a slice holds more than the first call's room for 8192
  multi_errors();
  {  // several multi-device scans at once (the library allows one handle per host thread)
    std::vector<std::thread> pool;
    for (int t = 0; t < 6; ++t) pool.emplace_back([t] { multi_unit(500 + 37 * size_t(t), 2 + size_t(t) % 4, 1); });
    for (auto& th : pool) th.join();
  }
  std::printf("  multi: ok\nhostlogic stress: all clean\n");
  return 0;
}
