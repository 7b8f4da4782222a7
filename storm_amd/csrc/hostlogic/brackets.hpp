// brackets.hpp -- the host-side bracketing of a scan: FilterSignSwap + finite filter, InverseQuadratic::optimize as a
// lock-step state machine over all brackets, and the frequency-slice split / cyclic deal of a multi-device scan.
// Pure host arithmetic (f64 + u64 bit patterns, no GPU call): the discriminant evaluations of a round come from a
// callback.  Compiled into libstorm_hip.so by hipcc (-ffp-contract=off: the expressions keep the reference's association
// and rounding) and into the CPU sanitizer tests by g++ (tests/sanitize/).
// Reference: src/bracket.rs (all), src/dynamic_interface.rs:116-160.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../../include/storm_hip.h"
#include "failure.hpp"

namespace storm {

struct Pt { double x, f; };

inline double sign_of(double f) {  // f64::signum: by sign bit, NaN stays NaN
  if (f != f) return f;
  return std::signbit(f) ? -1.0 : 1.0;
}
inline uint64_t bits_of(double x) { uint64_t b; std::memcpy(&b, &x, 8); return b; }
inline double from_bits(uint64_t b) { double x; std::memcpy(&x, &b, 8); return x; }

inline double secant(Pt a, Pt b) {  // evaluate_secant(., ., 0), src/bracket.rs:73-75
  const double y = 0.0;
  return (y - b.f) / (a.f - b.f) * a.x + (y - a.f) / (b.f - a.f) * b.x;
}
inline double inverse_quadratic(Pt a, Pt b, Pt c) {  // src/bracket.rs:77-81
  const double y = 0.0;
  return (y - b.f) * (y - c.f) / (a.f - b.f) / (a.f - c.f) * a.x +
         (y - a.f) * (y - c.f) / (b.f - a.f) / (b.f - c.f) * b.x +
         (y - a.f) * (y - b.f) / (c.f - b.f) / (c.f - a.f) * c.x;
}
// ensure_maximal_bracket, src/bracket.rs:96-132; true = converged (x is the root as proposed).
inline bool clamp_proposal(double upper, double lower, double& x, uint64_t ulp) {
  if (!(upper > lower) || !std::isfinite(upper) || !std::isfinite(lower) || !(upper >= 0.0) || !(lower >= 0.0))
    fail(STORM_ERR_BRACKET, "bracket must satisfy 0 <= lower < upper < inf");
  const uint64_t ub = bits_of(upper), lb = bits_of(lower);
  if (ub - lb <= ulp) return true;
  uint64_t xb = bits_of(x);
  if ((ub - lb) <= 2 * ulp) xb = lb + (ub - lb) / 2;
  else if ((ub < xb) || (ub - xb) < ulp) xb = ub - ulp;
  else if ((lb > xb) || (xb - lb) < ulp) xb = lb + ulp;
  x = from_bits(xb);
  return false;
}

struct BracketState {
  Pt lower, upper, previous;
  bool has_prev = false, done = false;
  double root = 0.0, pending = 0.0;
  uint64_t evals = 0;
};

// InverseQuadratic::optimize (src/bracket.rs:148-223) on all brackets at once: per round every open bracket proposes its
// next abscissa (secant first, then inverse quadratic with secant fallback; ensure_maximal_bracket on the bit patterns),
// `eval(xs, n, fs)` evaluates the n proposals in one batch, and each bracket replaces the end point of equal sign.
// Per bracket this is the reference's loop, evaluation for evaluation (`evals` included).
// `eval` may take a fourth argument, `const size_t* who`: the bracket each proposal belongs to (the scaled scan brings every
// value to its bracket's own power of two).
template <class Eval>
void refine_brackets(std::vector<BracketState>& br, uint64_t ulp, Eval&& eval) {
  if (ulp == 0) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be >= 1");
  if (!(ulp < (uint64_t(1) << 54))) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be < 2^54");
  std::vector<double> xs, fs;
  std::vector<size_t> who;
  for (;;) {
    xs.clear(); who.clear();
    for (size_t k = 0; k < br.size(); ++k) {
      BracketState& b = br[k];
      if (b.done) continue;
      double x;
      if (!b.has_prev) {
        x = secant(b.lower, b.upper);
      } else {
        x = inverse_quadratic(b.lower, b.upper, b.previous);
        if (x <= b.lower.x || x >= b.upper.x) x = secant(b.lower, b.upper);
      }
      if (clamp_proposal(b.upper.x, b.lower.x, x, ulp)) {
        b.done = true; b.root = x;
        continue;
      }
      b.pending = x;
      xs.push_back(x); who.push_back(k);
    }
    if (xs.empty()) break;
    fs.resize(xs.size());
    if constexpr (std::is_invocable_v<Eval&, const double*, size_t, double*, const size_t*>) eval(xs.data(), xs.size(), fs.data(), who.data());
    else eval(xs.data(), xs.size(), fs.data());
    for (size_t q = 0; q < who.size(); ++q) {
      BracketState& b = br[who[q]];
      const Pt next{xs[q], fs[q]};
      if (!std::isfinite(next.x) || !std::isfinite(next.f))
        fail(STORM_ERR_NON_FINITE, "non-finite discriminant during bracket refinement");
      b.evals += 1;
      if (sign_of(b.upper.f) == sign_of(next.f)) { b.previous = b.upper; b.upper = next; }
      else { b.previous = b.lower; b.lower = next; }
      b.has_prev = true;
    }
  }
}

// FilterSignSwap + finite filter (src/bracket.rs:236-250, src/dynamic_interface.rs:128-138) over the n scan values.
inline void filter_sign_swaps(const double* freq, const double* dets, size_t n, std::vector<BracketState>& br,
                              std::vector<uint64_t>& idx, bool& invalid) {
  invalid = false;
  for (size_t i = 0; i + 1 < n; ++i) {
    const double a = dets[i], b = dets[i + 1];
    if (sign_of(a) != sign_of(b)) {
      if (std::isfinite(a) && std::isfinite(b)) {
        BracketState st;
        st.lower = Pt{freq[i], a};
        st.upper = Pt{freq[i + 1], b};
        br.push_back(st);
        idx.push_back(i);
      } else {
        invalid = true;
      }
    }
  }
}

// The split of a scan over n_parts devices / ranks (storm_shard_bounds): part k owns the grid points [lo, hi) and
// evaluates [lo, min(hi + 1, n)) -- one halo point, so that every adjacent pair is owned by exactly one part.
inline void shard_bounds(size_t n, size_t n_parts, size_t k, size_t& lo, size_t& hi) {
  const size_t per = n_parts ? (n + n_parts - 1) / n_parts : n;
  lo = std::min(n, k * per);
  hi = std::min(n, lo + per);
}

}  // namespace storm
