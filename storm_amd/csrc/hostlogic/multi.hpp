// multi.hpp -- one scan over several devices from ONE process (storm_multi_*, include/storm_hip.h): the
// host-thread orchestration, the gather of the per-device bracket lists in ascending order and the cyclic deal of brackets
// and modes.  Host-only: every device-side step goes through `Ops` -- in libstorm_hip.so the single-device C ABI itself
// (storm_scan_brackets, storm_optimize_brackets, storm_eigenvector_batch on one storm_solver per device; storm_hip.hip), in
// the CPU sanitizer tests a scalar stand-in (tests/sanitize/hostlogic_stress.cpp), so that ThreadSanitizer sees the worker
// threads, the shared result arrays and the error hand-over, and AddressSanitizer the slice / halo / deal indexing.
//
// Ops:
//   using Solver = ...*;
//   int scan_brackets(Solver, const double* freq, size_t n, storm_bracket* out, size_t cap, size_t* n_out, int* invalid,
//                     double* det_out, uint64_t* index_out);          (status codes of storm_hip.h)
//   int optimize_brackets(Solver, storm_bracket* br, size_t n, uint64_t ulp);
//   int scan_and_optimize(Solver, ... as scan_brackets, with `uint64_t ulp` after n ...);   (the two in one call)
//   int eigenvector_batch(Solver, const double* freq, size_t n, double* det_out, double* vec_out);
//   size_t n_int(Solver);
//   std::string last_error();                                         (of the calling thread)
#pragma once
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/storm_hip.h"
#include "brackets.hpp"
#include "failure.hpp"

namespace storm {

// Runs job(k) for k < n on n host threads (the caller's thread takes k = 0); the first failure wins and its message
// becomes the caller's storm_last_error().  `last_error()` is read on the thread that failed (it is thread-local).
template <class Job, class LastError>
void on_every_device(size_t n, Job&& job, LastError&& last_error) {
  std::vector<int> rc(n, STORM_OK);
  std::vector<std::string> msg(n);
  auto run = [&](size_t k) {
    try {
      rc[k] = job(k);
      if (rc[k] != STORM_OK) msg[k] = last_error();
    } catch (const std::bad_alloc&) {  // (sizing a part: an exception must not leave a worker thread)
      rc[k] = STORM_ERR_HIP;
      msg[k] = "out of host memory";
    } catch (const Failure& f) {
      rc[k] = f.code;
      msg[k] = f.msg;
    } catch (const std::exception& e) {
      rc[k] = STORM_ERR_INVALID_ARGUMENT;
      msg[k] = e.what();
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(n ? n - 1 : 0);
  try {
    for (size_t k = 1; k < n; ++k) pool.emplace_back(run, k);
  } catch (...) {  // (thread creation failed: join what was started, then report)
    for (auto& th : pool) th.join();
    fail(STORM_ERR_HIP, "could not start a host thread per device");
  }
  run(0);
  for (auto& th : pool) th.join();
  for (size_t k = 0; k < n; ++k)
    if (rc[k] != STORM_OK) fail(rc[k], "device " + std::to_string(k) + ": " + msg[k]);
}

// ErasedSolver::scan_and_optimize (src/dynamic_interface.rs:116-160) over the solvers of D devices.
// deal = false (default of a storm_multi): every device refines the brackets of its own slice in the call that scanned it --
// one round of host threads, nothing between scan and refinement; what a scan with a few dozen modes wants (refinement
// rounds are latency-bound whatever the count).  deal = true: SURVEY 8(e)'s protocol -- scan, gather, deal the brackets
// cyclically, refine -- for a grid whose modes crowd into one slice.  Same brackets either way, bit for bit.
template <class Ops>
void multi_scan_and_optimize(const std::vector<typename Ops::Solver>& solvers, const double* freq, size_t n, uint64_t ulp,
                             storm_bracket* out, size_t cap, size_t* n_out, int* had_invalid, double* det_out,
                             uint64_t* index_out, bool deal = true) {
  if (ulp == 0) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be >= 1");
  const size_t D = solvers.size();
  auto scan = [&](typename Ops::Solver s, const double* f, size_t m, storm_bracket* br, size_t room, size_t* found, int* invalid,
                  double* dets, uint64_t* idx) {
    return deal ? Ops::scan_brackets(s, f, m, br, room, found, invalid, dets, idx)
                : Ops::scan_and_optimize(s, f, m, ulp, br, room, found, invalid, dets, idx);
  };
  // 1. every device scans its slice (+1 halo point) and filters its own adjacent pairs
  struct Part { size_t lo = 0; std::vector<storm_bracket> br; std::vector<uint64_t> idx; std::vector<double> det; int invalid = 0; };
  std::vector<Part> parts(D);
  on_every_device(D, [&](size_t k) {
    Part& p = parts[k];
    size_t lo, hi;
    shard_bounds(n, D, k, lo, hi);
    p.lo = lo;
    const size_t m = std::min(hi + 1, n) - lo;  // with the halo
    if (hi == lo) return int(STORM_OK);
    // room for 8192 brackets first (a scan finds a few dozen); only if a slice has more is the scan repeated with room
    // for one per grid point (zero-filling 56 B per grid point on every call cost 1 ms per 10^5 points)
    size_t room = std::min<size_t>(m, 8192);
    p.br.resize(room); p.idx.resize(room);
    if (det_out) p.det.resize(m);
    size_t found = 0;
    int rc = scan(solvers[k], freq + lo, m, p.br.data(), room, &found, &p.invalid, det_out ? p.det.data() : nullptr, p.idx.data());
    if (rc == STORM_ERR_CAPACITY && room < m) {
      room = m;
      p.br.resize(room); p.idx.resize(room);
      rc = scan(solvers[k], freq + lo, m, p.br.data(), room, &found, &p.invalid, det_out ? p.det.data() : nullptr, p.idx.data());
    }
    p.br.resize(rc == STORM_OK ? found : 0); p.idx.resize(rc == STORM_OK ? found : 0);
    if (rc == STORM_OK && det_out) std::memcpy(det_out + lo, p.det.data(), (hi - lo) * sizeof(double));  // owned points only
    return rc;
  }, Ops::last_error);
  // 2. the gather: slices are contiguous, so device order is ascending bracket order
  std::vector<storm_bracket> all;
  std::vector<uint64_t> idx;
  int invalid = 0;
  for (const Part& p : parts) {
    invalid |= p.invalid;
    for (size_t j = 0; j < p.br.size(); ++j) { all.push_back(p.br[j]); idx.push_back(p.idx[j] + p.lo); }
  }
  if (had_invalid) *had_invalid = invalid;
  *n_out = all.size();
  if (all.size() > cap) fail(STORM_ERR_CAPACITY, "bracket output buffer too small");
  if (!deal) {  // refined where they were found
    for (size_t j = 0; j < all.size(); ++j) {
      out[j] = all[j];
      if (index_out) index_out[j] = idx[j];
    }
    return;
  }
  // 3. the deal: bracket j goes to device j mod D
  std::vector<std::vector<storm_bracket>> share(D);
  for (size_t j = 0; j < all.size(); ++j) share[j % D].push_back(all[j]);
  on_every_device(D, [&](size_t k) {
    return share[k].empty() ? int(STORM_OK) : Ops::optimize_brackets(solvers[k], share[k].data(), share[k].size(), ulp);
  }, Ops::last_error);
  for (size_t j = 0; j < all.size(); ++j) {
    out[j] = share[j % D][j / D];
    if (index_out) index_out[j] = idx[j];
  }
}

// The per-mode eigenvector loop of StormState::scan (src/main.rs:824-831): mode j goes to device j mod D.
template <class Ops>
void multi_eigenvector_batch(const std::vector<typename Ops::Solver>& solvers, const double* freq, size_t n, double* det_out,
                             double* vec_out) {
  if (n == 0) return;
  const size_t D = solvers.size(), len = (Ops::n_int(solvers[0]) + 1) * 4;
  if (D == 1) {  // one device: straight into the caller's block (no staging copy of 32 B x grid points per mode)
    std::vector<double> d(det_out ? 0 : n);
    const int rc = Ops::eigenvector_batch(solvers[0], freq, n, det_out ? det_out : d.data(), vec_out);
    if (rc != STORM_OK) fail(rc, "device 0: " + Ops::last_error());
    return;
  }
  on_every_device(D, [&](size_t k) {
    std::vector<double> f, d;
    for (size_t j = k; j < n; j += D) f.push_back(freq[j]);
    if (f.empty()) return int(STORM_OK);
    d.resize(f.size());
    std::unique_ptr<double[]> v(new double[f.size() * len]);  // (uninitialised: every element is written by the call)
    const int rc = Ops::eigenvector_batch(solvers[k], f.data(), f.size(), d.data(), v.get());
    if (rc != STORM_OK) return rc;
    for (size_t i = 0, j = k; j < n; j += D, ++i) {
      std::memcpy(vec_out + j * len, v.get() + i * len, len * sizeof(double));
      if (det_out) det_out[j] = d[i];
    }
    return int(STORM_OK);
  }, Ops::last_error);
}

}  // namespace storm
