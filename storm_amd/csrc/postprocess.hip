// postprocess.hip -- eigenfunctions, normalisation and mode identification on the device.
//
// Reference: Rotating1DPostprocessing::new (src/postprocessing.rs:58-301) and count_windings (:304-355).
// Compiled -ffp-contract=off: every expression keeps the reference's association, so all 13 fields, the
// norm, the crossing counts and the node positions are bit-identical to the CPU path.
//
// HBM-bound, three kernels per batch of modes:
//   post_terms    lane = (mode, point): the mode-inertia integrand of every point          (40 B/pt/mode)
//   post_modes    wave = mode: the integrand summed IN INDEX ORDER (the reference's `norm +=` loop is a
//                 sequential sum; any tree would change the last bits), scale = 1/sqrt(norm), then the
//                 winding count over adjacent point pairs with ballot-ordered node compaction  (40 B/pt/mode)
//   post_fields   lane = (mode, point): all 13 fields recomputed and stored scaled          (136 B/pt/mode)
// Recomputing the point in post_fields costs ~80 flops and saves storing + re-reading 13 unscaled fields.
#include "postprocess.hpp"

#include <algorithm>
#include <type_traits>
#include "shared_div.hpp"

#define PDEV __device__ __forceinline__

namespace storm {
namespace {

// f64::powi with a run-time exponent (compiler-builtins __powidf2): square-and-multiply on |b|, reciprocal
// at the end for b < 0.
PDEV double powi_rt(double a, int b) {
  unsigned pw = b < 0 ? 0u - unsigned(b) : unsigned(b);
  double mul = 1.0;
  for (;;) {
    if (pw & 1u) mul *= a;
    pw >>= 1;
    if (pw == 0) break;
    a *= a;
  }
  return b < 0 ? 1.0 / mul : mul;
}

struct PointFields {
  double v[kPostFields];  // y1 y2 y3 y4 xi_r xi_h xi_tn xi_tp psi dpsi rho p chi  (unscaled)
  double term;            // integrand of the norm (0 for the central point, which the reference skips)
};

enum { Y1, Y2, Y3, Y4, XI_R, XI_H, XI_TN, XI_TP, PSI, DPSI, RHO, PRS, CHI };

PDEV PointFields point_fields(const PostModelDev& md, const PostParams& q, double freq, const double* __restrict__ vec, size_t i) {
  const size_t np = md.n_pts;
  const double* __restrict__ t = md.tab;
  PointFields o;
  const double2 ya = *reinterpret_cast<const double2*>(vec + i * 4);
  const double2 yb = *reinterpret_cast<const double2*>(vec + i * 4 + 2);
  const double y1 = ya.x, y2 = ya.y, y3 = yb.x, y4 = yb.y;
  o.v[Y1] = y1; o.v[Y2] = y2; o.v[Y3] = y3; o.v[Y4] = y4;
  const double rho = t[PP_RHO * np + i];
  const double rotv = t[PP_ROT * np + i];
  const double mrot = q.mf * rotv;
  double xi_r, xi_h;
  if (i != 0) {  // src/postprocessing.rs:113-171
    const double r = t[PP_R * np + i];
    const double vv = t[PP_V * np + i], g1 = t[PP_GAMMA1 * np + i], c1 = t[PP_C1 * np + i];
    const double dphi = t[PP_M * np + i] / (r * r);
    const double rl1 = powi_rt(r, q.ell - 1), rl2 = powi_rt(r, q.ell - 2);
    xi_r = y1 * rl1;
    const double p_prime = y2 * rho * dphi * rl1;
    const double psi_prime = y3 * dphi * rl1;
    o.v[DPSI] = y4 * dphi * rl2;
    const double rsigma = freq - mrot;
    double omega_rsq, rel_rot;
    if (q.ell != 0) {
      const double d = freq - mrot;
      omega_rsq = (q.lambda * d + 2.0 * mrot) * d;
      rel_rot = 2.0 * mrot / (q.lambda * d + 2.0 * mrot);
      const double f = 2.0 * mrot / q.lambda;
      xi_h = 1.0 / (rsigma * rho * (rsigma + f)) * ((p_prime + rho * psi_prime) / r - f * rsigma * rho * xi_r);
    } else {
      omega_rsq = 1.0; rel_rot = 0.0; xi_h = 0.0;
    }
    const double vg = vv / g1;
    const double l2 = (q.lambda * q.lambda) / (omega_rsq * c1);
    const double xdy1 = (vg - 1.0 - q.ellf - q.lambda * rel_rot) * y1 + (-vv / g1 + l2) * y2 + l2 * y3;
    const double chi = rl2 * ((q.ellf + 1.0) * y1 + xdy1) - q.lambda / r * xi_h;
    o.v[CHI] = chi;
    o.v[RHO] = -rho * chi + xi_r * rho / r * (vg + t[PP_ASTAR * np + i]);
    o.v[PRS] = p_prime;
    o.v[PSI] = psi_prime;
    o.term = rho * (r * r) * (xi_r * xi_r + q.lambda * xi_h * xi_h) * t[PP_TRAP * np + i];
  } else {  // central point, src/postprocessing.rs:173-208
    const double kPi = 3.14159265358979323846;
    if (q.ell != 1) {
      xi_r = 0.0; o.v[DPSI] = 0.0; xi_h = 0.0;
    } else {
      const double ddphi0 = 4.0 / 3.0 * kPi * rho;
      xi_r = y1;
      o.v[DPSI] = y4 * ddphi0;
      const double rsigma = freq - mrot;
      const double f = 2.0 * q.mf * rotv / q.lambda;
      xi_h = 1.0 / (rsigma * (rsigma + f)) * ((y2 + y3) * ddphi0 - f * xi_r);
    }
    if (q.ell == 0) {
      const double p_prime = y2 * (rho * rho) * 4.0 / 3.0 * kPi;
      o.v[PRS] = p_prime;
      o.v[PSI] = y3 * rho * 4.0 / 3.0 * kPi;
      const double rho_prime = rho * p_prime / (t[PP_GAMMA1 * np] * t[PP_P * np]);
      o.v[RHO] = rho_prime;
      o.v[CHI] = -rho_prime / rho;
    } else {
      o.v[PRS] = 0.0; o.v[PSI] = 0.0; o.v[RHO] = 0.0; o.v[CHI] = 0.0;
    }
    o.term = 0.0;
  }
  o.v[XI_R] = xi_r; o.v[XI_H] = xi_h;
  // toroidal components, src/postprocessing.rs:155-165 and :210-224
  if (q.tn_zero) o.v[XI_TN] = 0.0;
  else o.v[XI_TN] = 2.0 * rotv / (q.lambda_n1 * (freq - mrot) + 2.0 * q.mf * rotv) * (-q.q_hd_n * xi_r + q.q_h_n * xi_h);
  if (i == 0 && q.ell == 0) o.v[XI_TP] = 0.0;
  else o.v[XI_TP] = 2.0 * rotv / (q.lambda_p1 * (freq - mrot) + 2.0 * q.mf * rotv) * (-q.q_hd_p * xi_r + q.q_h_p * xi_h);
  return o;
}

__global__ __launch_bounds__(256) void post_terms_kernel(PostModelDev md, PostParams q, const double* __restrict__ freq,
                                                         const double* __restrict__ vec, double* __restrict__ terms,
                                                         const PostSystem* __restrict__ systems, const int* __restrict__ mode_sys) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t k = blockIdx.y;
  if (systems) { const PostSystem& ps = systems[mode_sys[k]]; md = ps.model; q = ps.par; }  // block-uniform
  if (i >= md.n_pts) return;
  const PointFields f = point_fields(md, q, freq[k], vec + k * md.n_pts * 4, i);
  terms[k * md.n_pts + i] = f.term;
}

__global__ __launch_bounds__(256) void post_fields_kernel(PostModelDev md, PostParams q, const double* __restrict__ freq,
                                                          const double* __restrict__ vec, const double* __restrict__ scale,
                                                          double* __restrict__ fields, const PostSystem* __restrict__ systems,
                                                          const int* __restrict__ mode_sys) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t k = blockIdx.y;
  if (systems) { const PostSystem& ps = systems[mode_sys[k]]; md = ps.model; q = ps.par; }
  if (i >= md.n_pts) return;
  const PointFields f = point_fields(md, q, freq[k], vec + k * md.n_pts * 4, i);
  const double sc = scale[k];
  double* __restrict__ out = fields + k * kPostFields * md.n_pts + i;
#pragma unroll
  for (int c = 0; c < kPostFields; ++c) out[(size_t)c * md.n_pts] = f.v[c] * sc;
}

constexpr int kSumTile = 1024;  // terms staged per round of the ordered sum (16 per lane)

// One wave per mode.
__global__ __launch_bounds__(64) void post_modes_kernel(PostModelDev md, PostParams q, const double* __restrict__ vec,
                                                        const double* __restrict__ terms, double* __restrict__ scale_out,
                                                        long long* __restrict__ ids, double* __restrict__ nodes, size_t node_cap,
                                                        const PostSystem* __restrict__ systems, const int* __restrict__ mode_sys) {
  __shared__ __attribute__((aligned(16))) double tile[2][kSumTile];
  const int lane = threadIdx.x;
  const size_t k = blockIdx.x;
  if (systems) { const PostSystem& ps = systems[mode_sys[k]]; md = ps.model; q = ps.par; }
  const size_t np = md.n_pts;
  const double* __restrict__ tm = terms + k * np;
  // ---- norm: 0 + t[1] + t[2] + ... in index order (src/postprocessing.rs:97,167-170) ----
  double norm = 0.0;
  {
    const size_t count = np - 1;  // terms 1 .. np-1
    double stage[kSumTile / 64];
    auto fetch = [&](size_t base) {
#pragma unroll
      for (int j = 0; j < kSumTile / 64; ++j) {
        const size_t idx = base + (size_t)j * 64 + lane;
        stage[j] = idx < count ? tm[1 + idx] : 0.0;
      }
    };
    fetch(0);
    int buf = 0;
    for (size_t base = 0; base < count; base += kSumTile) {
#pragma unroll
      for (int j = 0; j < kSumTile / 64; ++j) tile[buf][j * 64 + lane] = stage[j];
      __syncthreads();
      if (base + kSumTile < count) fetch(base + kSumTile);  // in flight while this tile is summed
      const size_t lim = count - base < (size_t)kSumTile ? count - base : (size_t)kSumTile;
      if (lim == (size_t)kSumTile) {
#pragma unroll 16
        for (int j = 0; j < kSumTile; j += 2) {
          const double2 v = *reinterpret_cast<const double2*>(&tile[buf][j]);
          norm += v.x;
          norm += v.y;
        }
      } else {
        for (size_t j = 0; j < lim; ++j) norm += tile[buf][j];
      }
      buf ^= 1;
    }
  }
  const double sc = 1.0 / sqrt(norm);  // src/postprocessing.rs:226
  if (lane == 0) scale_out[k] = sc;

  // ---- count_windings on the normalised solution (src/postprocessing.rs:244-279, 304-355) ----
  const double* __restrict__ e = vec + k * np * 4;
  const double* __restrict__ rr = md.tab + PP_R * np;
  const double* __restrict__ uu = md.tab + PP_U * np;
  const size_t first = q.ell == 1 ? 2 : 0;  // the dipole case drops the two innermost points
  unsigned long long cw = 0, ccw = 0, n_nodes = 0;
  // curve of point i: (a, b) = (y1, y2) | ((1-u/3) y1 + (y3-y4)/3, y2 - y1) | (y1, y2 + y3)
  auto curve = [&](size_t i, double& a, double& b) {
    const double2 ya = *reinterpret_cast<const double2*>(e + i * 4);
    const double2 yb = *reinterpret_cast<const double2*>(e + i * 4 + 2);
    const double y1 = ya.x * sc, y2 = ya.y * sc, y3 = yb.x * sc, y4 = yb.y * sc;
    if (q.ell == 0) { a = y1; b = y2; }
    else if (q.ell == 1) { a = (1.0 - uu[i] / 3.0) * y1 + (y3 - y4) / 3.0; b = y2 - y1; }
    else { a = y1; b = y2 + y3; }
  };
  for (size_t base = first; base + 1 < np; base += 64) {
    const size_t i = base + lane;
    bool cross = false, is_cw = false;
    double xt = 0.0;
    if (i + 1 < np) {
      double a1, b1, a2, b2;
      curve(i, a1, b1);
      curve(i + 1, a2, b2);
      const bool up = a1 <= 0.0 && a2 > 0.0;
      const bool down = !up && a1 >= 0.0 && a2 < 0.0;
      if (up || down) {
        const double x1 = rr[i], x2 = rr[i + 1];
        const double yt = b1 - a1 * (b2 - b1) / (a2 - a1);
        xt = x2 - a1 * (x2 - x1) / (a2 - a1);
        cross = true;
        is_cw = up ? (yt > 0.0) : !(yt > 0.0);
      }
    }
    const unsigned long long mc = __ballot(cross), mw = __ballot(cross && is_cw);
    if (cross && nodes) {
      const size_t slot = n_nodes + __popcll(mc & ((1ULL << lane) - 1ULL));
      if (slot < node_cap) nodes[k * node_cap + slot] = xt;
    }
    n_nodes += __popcll(mc);
    cw += __popcll(mw);
    ccw += __popcll(mc & ~mw);
  }
  if (lane == 0) {
    long long order;
    if (q.ell == 0) order = (long long)ccw;
    else if (q.ell == 1) order = (long long)ccw - (long long)cw + (cw > ccw ? 0 : 1);
    else order = (long long)ccw - (long long)cw;
    ids[k * 4 + 0] = (long long)cw;
    ids[k * 4 + 1] = (long long)ccw;
    ids[k * 4 + 2] = order;
    ids[k * 4 + 3] = (long long)n_nodes;
  }
}


// ---------------------------------------------------------------------------------------
// Mode-coupling integrals (reference src/perturbed.rs:110-263): six M x M matrices, each element a sum over the
// radial grid of 148 flops of products of the left and right modes' post-processing fields.
// Lane = one (left, right) pair, wave = 8 x 8 tile of pairs; every lane walks the grid in index order with its
// six accumulators (the reference's `+=` loop: a sequential sum), so results are bit-identical.
// Staging: per chunk of 8 grid points the wave needs 4 fields of its 8 left modes and 5 of its 8 right modes; they
// are copied global -> LDS by LDS-DMA (global_load_lds_dword: no VGPRs, any 4-byte alignment) one chunk ahead of
// the arithmetic, so the serial walk never waits for memory.  The 19 mode-independent weights of a point have a
// wave-uniform address: they are read through the constant address space (s_load, scalar cache) into SGPRs and feed
// the FP64 instructions as scalar operands -- no LDS broadcast (that was 20 of 29 LDS reads per point), 28 fewer
// VGPRs (121: four waves per SIMD).  FP64 VALU bound (148 flops per (pair, point); fields are read once per tile).
// No MFMA: the contraction could be phrased as ~20 weighted GEMMs, but MFMA accumulates with fused multiply-adds
// in its own order and would change the last bits of every element; FP64 MFMA has the same peak as the FP64
// vector pipe on MI355X, so the only cost of staying on the VALU is the un-fused mul/add split.
// ---------------------------------------------------------------------------------------
constexpr int kCpChunk = 8;                                      // grid points per staged chunk
constexpr int kCpPlane = kCpChunk * 8;                           // doubles: [point][mode of the tile]
constexpr int kCpFieldTile = 9 * kCpPlane;                       // 4 left + 5 right planes

typedef __attribute__((address_space(3))) void* cp_lds_ptr;
typedef __attribute__((address_space(1))) const void* cp_glb_ptr;

// Requests chunk [rc0, rc0 + 8) of the fields this wave needs into `tile` (points past the grid are clamped to
// the last one).  LDS image per plane: [point][mode], so the 8 modes of
// one point sit in 8 consecutive doubles (conflict-free reads, each shared by 8 lanes).  One DMA instruction
// moves 256 B = 4 points x 8 modes; lane -> (point, mode, half of the double).
// `lrow` / `rrow`: the fields row of the left / right mode this LANE stages (see coupling_body).
// NULL_TILE: only the four planes the shortcut of coupling_body reads (left psi, p; right xi_r, rho).
template <bool NULL_TILE>
PDEV void coupling_stage(const CouplingArgs& a, size_t lrow, size_t rrow, size_t rc0, double* tile, int lane) {
  const size_t np = a.n_pts;
  const int pt_in = lane >> 4, word = lane & 1;
  constexpr int kPlanes[9] = {XI_R, XI_H, PSI, PRS, XI_R, XI_H, XI_TP, XI_TN, RHO};  // 4 left, 5 right
#pragma unroll
  for (int pl = 0; pl < 9; ++pl) {
    if (NULL_TILE && !(pl == 2 || pl == 3 || pl == 4 || pl == 8)) continue;
    const double* row = a.fields + ((pl < 4 ? lrow : rrow) * kPostFields + kPlanes[pl]) * np;
#pragma unroll
    for (int h = 0; h < kCpChunk / 4; ++h) {
      size_t pt = rc0 + h * 4 + pt_in;
      pt = pt < np ? pt : np - 1;
      const char* src = reinterpret_cast<const char*>(row + pt) + word * 4;
      __builtin_amdgcn_global_load_lds((cp_glb_ptr)src, (cp_lds_ptr)(tile + pl * kCpPlane + h * 32), 4, 0, 0);
    }
  }
}

// weights: uniform address, constant address space -> scalar loads
typedef const double __attribute__((address_space(4))) * cp_const_ptr;

PDEV void coupling_body(const CouplingArgs& a, unsigned tile_l, unsigned tile_r) {
  __shared__ __attribute__((aligned(16))) double tiles[2][kCpFieldTile];
  const int lane = threadIdx.x;
  const int tl = lane >> 3, tr = lane & 7;
  const size_t M = a.n_modes, np = a.n_pts;
  const size_t l0 = (size_t)tile_l * 8, r0 = (size_t)tile_r * 8;
  size_t lrow, rrow;
  {
    const int mode_in = (lane >> 1) & 7;
    size_t lmode = l0 + mode_in, rmode = r0 + mode_in;
    lmode = lmode < M ? lmode : M - 1;
    rmode = rmode < M ? rmode : M - 1;
    lrow = a.rows ? (size_t)a.rows[lmode] : lmode;
    rrow = a.rows ? (size_t)a.rows[rmode] : rmode;
  }
  const bool live = l0 + tl < M && r0 + tr < M;
  const size_t lm = l0 + tl < M ? l0 + tl : M - 1, rm = r0 + tr < M ? r0 + tr : M - 1;
  const double* __restrict__ q = a.consts + ((size_t)a.ell_index[lm] * a.n_ell + a.ell_index[rm]) * kCouplingConsts;
  const double q2 = q[CQ_Q2], q2_hrd = q[CQ_Q2_HRD], q2_hld = q[CQ_Q2_HLD], q2_h = q[CQ_Q2_H];
  const double q_rt_p = q[CQ_RT_P], q_rt_n = q[CQ_RT_N], q_ht_p = q[CQ_HT_P], q_ht_n = q[CQ_HT_N];
  const double ipr = q[CQ_IPR], iph = q[CQ_IPH];
  const double k6 = ipr + 6.0 * q2;
  const bool no_q2 = __all(q2 == 0.0 && q2_hrd == 0.0);
  const double rfreq = a.freq[rm];
  const cp_const_ptr wts = (cp_const_ptr)a.weights;
  double d_squared = 0.0, d_linear = 0.0, d_zero = 0.0, r_linear = 0.0, r_zero = 0.0, l_zero = 0.0;
  bool finite = true;
  // this lane's nine values of point c of a tile: planes 0-3 at mode tl, 4-8 at mode tr
  auto fetch = [&](const double* __restrict__ t, int c, double (&f)[9]) {
#pragma unroll
    for (int pl = 0; pl < 9; ++pl) f[pl] = t[pl * kCpPlane + c * 8 + (pl < 4 ? tl : tr)];
  };
  // Degrees more than 2 apart: all ten factors are zero.  With finite operands of moderate size every product that
  // contains one of them is +-0, five of the six sums stay +0.0 and what is left of l_zero's term is
  // trap r^2 ((psi_l rho'_r tea - rho psi_l xi_r adtea / r) - p_l xi_r adtea / r): adding or subtracting a zero never
  // changes a non-zero partial sum, and an accumulator that starts at +0.0 never becomes -0.0, so the signs of the
  // dropped zeros cannot reach the result.  "Moderate" (the `tame` flags: |v| < 2^200, so no dropped product overflows
  // into inf * 0 = NaN) is checked per mode by coupling_tame_kernel and for the weights on the host; anything else takes
  // the full expression below.
  const bool all_zero = q2 == 0.0 && q2_hrd == 0.0 && q2_hld == 0.0 && q2_h == 0.0 && q_rt_p == 0.0 && q_rt_n == 0.0 &&
                        q_ht_p == 0.0 && q_ht_n == 0.0 && ipr == 0.0 && iph == 0.0;
  const bool tame_tile = a.tame != nullptr && a.weights_tame && __all(a.tame[lm] != 0 && a.tame[rm] != 0);
  if (tame_tile && __all(all_zero)) {
    coupling_stage<true>(a, lrow, rrow, 1, tiles[0], lane);
    int buf = 0;
    for (size_t rc0 = 1; rc0 < np; rc0 += kCpChunk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (rc0 + kCpChunk < np) coupling_stage<true>(a, lrow, rrow, rc0 + kCpChunk, tiles[buf ^ 1], lane);
      const double* __restrict__ t = tiles[buf];
      const int cnt = np - rc0 < (size_t)kCpChunk ? (int)(np - rc0) : kCpChunk;
      for (int c = 0; c < cnt; ++c) {
        const cp_const_ptr u = wts + (rc0 + c) * kCouplingWeights;
        const double lpsi = t[2 * kCpPlane + c * 8 + tl], lp = t[3 * kCpPlane + c * 8 + tl];
        const double rxr = t[4 * kCpPlane + c * 8 + tr], rrho = t[8 * kCpPlane + c * 8 + tr];
        const double adtea = u[CW_ADTEA];
        double t0 = u[CW_RHO] * lpsi * rxr * adtea, t2 = lp * rxr * adtea;
        shared_div2(t0, t2, SharedDenominator{u[CW_X], u[CW_RX], shared_div_moderate(u[CW_X])});
        l_zero += u[CW_TR2] * (lpsi * rrho * u[CW_TEA] - t0 - t2);
      }
      buf ^= 1;
    }
    if (live) {
      const size_t at = lm + rm * M, mm = M * M;
      a.out[0 * mm + at] = 0.0; a.out[1 * mm + at] = 0.0; a.out[2 * mm + at] = 0.0;
      a.out[3 * mm + at] = 0.0; a.out[4 * mm + at] = 0.0; a.out[5 * mm + at] = l_zero;
      if (!isfinite(l_zero)) atomicOr(a.flags, 1);  // non-finite is absorbing under +=: the last value tells
    }
    return;
  }
  // Different degrees (src/perturbed.rs:134-146): the inner products and the four toroidal factors are zero.  By the
  // same argument as above a tile of such pairs skips the products that contain them (about a quarter of the
  // instructions); OFF_DIAG is that variant of the walk, the other one is the reference's expression term by term.
  const bool off_diag = ipr == 0.0 && iph == 0.0 && q_rt_p == 0.0 && q_rt_n == 0.0 && q_ht_p == 0.0 && q_ht_n == 0.0;
  auto walk = [&](auto off_diag_tag) {
    constexpr bool OFF_DIAG = decltype(off_diag_tag)::value;
    coupling_stage<false>(a, lrow, rrow, 1, tiles[0], lane);
    int buf = 0;
    for (size_t rc0 = 1; rc0 < np; rc0 += kCpChunk) {  // src/perturbed.rs:148-261
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (rc0 + kCpChunk < np) coupling_stage<false>(a, lrow, rrow, rc0 + kCpChunk, tiles[buf ^ 1], lane);
      const double* __restrict__ t = tiles[buf];
      const int cnt = np - rc0 < (size_t)kCpChunk ? (int)(np - rc0) : kCpChunk;
      double f[9];
      for (int c = 0; c < cnt; ++c) {
        fetch(t, c, f);
        const cp_const_ptr u = wts + (rc0 + c) * kCouplingWeights;  // wave-uniform
        const double lxr = f[0], lxh = f[1], lpsi = f[2], lp = f[3];
        const double rxr = f[4], rxh = f[5], rtpv = f[6], rtnv = f[7], rrho = f[8];
        const double rr = lxr * rxr, rh = lxr * rxh, hr = lxh * rxr, hh = lxh * rxh;
        const double eb = u[CW_EB], rho = u[CW_RHO], x = u[CW_X];
        double val;
        if constexpr (OFF_DIAG)
          val = u[CW_VOL] * (rr * 2.0 * q2 * u[CW_U3] + (rh * q2_hrd + hr * q2_hld + hh * 2.0 * q2_h) * eb);
        else
          val = u[CW_VOL] * (rr * ipr * u[CW_U1] + hh * iph * u[CW_U2] + rr * 2.0 * q2 * u[CW_U3] +
                             (rh * q2_hrd + hr * q2_hld + hh * 2.0 * q2_h) * eb);
        finite = finite && isfinite(val);
        d_squared += val;
        d_linear += u[CW_C1] * val;
        d_zero += u[CW_C2] * val;
        const double mrot = u[CW_MROT];
        if constexpr (OFF_DIAG) {
          val = u[CW_VOL2] * ((rh + hr) * q2 * u[CW_U5] + hh * k6 * 2.0 * eb);
          finite = finite && isfinite(val);
          r_linear += mrot * val;
          r_zero += mrot * (mrot * val);
        } else {
          const double rtp = lxr * rtpv, rtn = lxr * rtnv, htp = lxh * rtpv, htn = lxh * rtnv;
          val = u[CW_VOL2] * (ipr * ((hr + rh) * u[CW_U4] + hh * u[CW_U2]) + (rh + hr) * q2 * u[CW_U5] + hh * k6 * 2.0 * eb);
          const double val_t = u[CW_VOL2ROT] * (q_rt_p * rtp + q_rt_n * rtn + q_ht_n * htn + q_ht_p * htp);
          finite = finite && isfinite(val) && isfinite(val_t);
          r_linear += mrot * val + val_t;
          r_zero += mrot * (mrot * val + val_t);
        }
        const double tea = u[CW_TEA], adtea = u[CW_ADTEA], teb = u[CW_TEB], adteb = u[CW_ADTEB];
        const SharedDenominator den{x, u[CW_RX], shared_div_moderate(x)};  // reciprocal: coupling_recip_kernel
        double t0, t1, t2, t3, t4, t5;
        if (no_q2) {
          // the four terms with a factor q2 or q2_hrd are +-0 (or NaN) for every lane of this wave and dividing them by
          // the radius returns them unchanged: two quotients are left
          t0 = rho * lpsi * rxr * adtea;
          t2 = lp * rxr * adtea;
          shared_div2(t0, t2, den);
          t1 = rho * lpsi * q2 * rxr * adteb;
          t3 = lp * q2 * rxr * adteb;
          t4 = lp * q2_hrd * rxh * teb;
          t5 = q2_hrd * rho * lpsi * rxh * teb;
        } else {
          double dv[6] = {rho * lpsi * rxr * adtea, rho * lpsi * q2 * rxr * adteb, lp * rxr * adtea,
                          lp * q2 * rxr * adteb,    lp * q2_hrd * rxh * teb,       q2_hrd * rho * lpsi * rxh * teb};
          shared_div6(dv, den);
          t0 = dv[0]; t1 = dv[1]; t2 = dv[2]; t3 = dv[3]; t4 = dv[4]; t5 = dv[5];
        }
        if constexpr (OFF_DIAG) {
          l_zero += u[CW_TR2] * (lpsi * rrho * tea - t0 + lpsi * q2 * rrho * teb - t1 - t2 - t3 - t4 + t5);
        } else {
          const double sig = rfreq - mrot;
          l_zero += u[CW_TR2] * ((sig * sig) * rho * (ipr * rr + iph * hh) + u[CW_C1] * sig * rho * ipr * (rh + hr + hh) +
                                 lpsi * rrho * tea - t0 + lpsi * q2 * rrho * teb - t1 - t2 - t3 - t4 + t5);
        }
        finite = finite && isfinite(l_zero);
      }
      buf ^= 1;
    }
  };
  if (tame_tile && __all(off_diag)) walk(std::true_type{});
  else walk(std::false_type{});
  if (live) {
    const size_t at = lm + rm * M, mm = M * M;
    a.out[0 * mm + at] = d_squared; a.out[1 * mm + at] = d_linear; a.out[2 * mm + at] = d_zero;
    a.out[3 * mm + at] = r_linear; a.out[4 * mm + at] = r_zero; a.out[5 * mm + at] = l_zero;
    if (!finite) atomicOr(a.flags, 1);
  }
}


// Column CW_RX of the weights: the refined reciprocal of the radius every division of a point shares (shared_div.hpp),
// computed once per point here instead of once per (tile, point) in the walk -- the same instructions on the same
// input, so the same bits.
PDEV void coupling_recip_body(const CouplingArgs& a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < a.n_pts) a.weights[i * kCouplingWeights + CW_RX] = shared_div_prepare(a.weights[i * kCouplingWeights + CW_X]).r;
}

__global__ __launch_bounds__(256) void coupling_recip_kernel(CouplingArgs a) { coupling_recip_body(a); }

__global__ __launch_bounds__(256) void coupling_recip_batch_kernel(const CouplingArgs* __restrict__ table) {
  const CouplingArgs a = table[blockIdx.y];
  if (a.n_modes) coupling_recip_body(a);
}

// tame[k] = 1 if the frequency of mode k and every value of the seven field planes the integrals read are finite and
// smaller than 2^200 in magnitude (exponent test on the high words).  One workgroup per mode.
constexpr unsigned kTameLimit = unsigned(1023 + 200) << 20;
PDEV void coupling_tame_body(const CouplingArgs& a, size_t k) {
  __shared__ unsigned part[4];
  const size_t row = a.rows ? (size_t)a.rows[k] : k;
  constexpr int kPlanes[7] = {XI_R, XI_H, PSI, PRS, XI_TP, XI_TN, RHO};
  unsigned hi = unsigned(__double2hiint(a.freq[k])) & 0x7fffffffu;
  for (int pl = 0; pl < 7; ++pl) {
    const double* __restrict__ v = a.fields + (row * kPostFields + kPlanes[pl]) * a.n_pts;
    for (size_t i = threadIdx.x; i < a.n_pts; i += 256) {
      const unsigned w = unsigned(__double2hiint(v[i])) & 0x7fffffffu;
      hi = w > hi ? w : hi;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)hi, off);
    hi = o > hi ? o : hi;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) hi = part[w] > hi ? part[w] : hi;
    a.tame[k] = hi < kTameLimit ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void coupling_tame_kernel(CouplingArgs a) { coupling_tame_body(a, blockIdx.x); }

__global__ __launch_bounds__(256) void coupling_tame_batch_kernel(const CouplingArgs* __restrict__ table) {
  const CouplingArgs a = table[blockIdx.y];
  if (a.tame == nullptr || blockIdx.x >= a.n_modes) return;
  coupling_tame_body(a, blockIdx.x);
}

__global__ __launch_bounds__(64, 2) void coupling_kernel(CouplingArgs a) { coupling_body(a, blockIdx.x, blockIdx.y); }

// blockIdx.z = problem; the grid covers the largest one, tiles past a problem's own size leave at once
__global__ __launch_bounds__(64, 2) void coupling_batch_kernel(const CouplingArgs* __restrict__ table) {
  const CouplingArgs a = table[blockIdx.z];
  const size_t tiles = (a.n_modes + 7) / 8;
  if (blockIdx.x >= tiles || blockIdx.y >= tiles) return;
  coupling_body(a, blockIdx.x, blockIdx.y);
}

}  // namespace

hipError_t launch_postprocess(const PostArgs& a) {
  if (a.n == 0) return hipSuccess;
  if (a.n > 65535 || a.model.n_pts < 2) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((a.model.n_pts + 255) / 256), (unsigned)a.n, 1);
  post_terms_kernel<<<grid, 256, 0, a.stream>>>(a.model, a.par, a.freq, a.vec, a.terms, a.systems, a.mode_sys);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  post_modes_kernel<<<(unsigned)a.n, 64, 0, a.stream>>>(a.model, a.par, a.vec, a.terms, a.scale, a.ids, a.nodes, a.node_cap, a.systems, a.mode_sys);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.fields) {
    post_fields_kernel<<<grid, 256, 0, a.stream>>>(a.model, a.par, a.freq, a.vec, a.scale, a.fields, a.systems, a.mode_sys);
    e = hipGetLastError();
  }
  return e;
}

namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(const double* __restrict__ src, const long long* __restrict__ index,
                                                          size_t row_elems, double* __restrict__ dst) {
  const size_t k = blockIdx.y;
  const double* __restrict__ from = src + (size_t)index[k] * row_elems;
  double* __restrict__ to = dst + k * row_elems;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < row_elems; e += (size_t)gridDim.x * 256) to[e] = from[e];
}
}  // namespace

hipError_t launch_gather_rows(const double* src, const long long* index, size_t n, size_t row_elems, double* dst,
                              hipStream_t stream) {
  if (n == 0 || row_elems == 0) return hipSuccess;
  if (n > 65535) return hipErrorInvalidValue;
  const unsigned bx = (unsigned)std::min<size_t>((row_elems + 255) / 256, 128);
  gather_rows_kernel<<<dim3(bx, (unsigned)n, 1), 256, 0, stream>>>(src, index, row_elems, dst);
  return hipGetLastError();
}

hipError_t launch_coupling_batch(const CouplingArgs* table, size_t n_problems, size_t max_modes, size_t n_pts,
                                 hipStream_t stream, bool with_tame) {
  if (n_problems == 0 || max_modes == 0) return hipSuccess;
  const unsigned tiles = (unsigned)((max_modes + 7) / 8);
  if (tiles > 65535 || n_problems > 65535) return hipErrorInvalidValue;
  if (with_tame) {
    coupling_tame_batch_kernel<<<dim3((unsigned)max_modes, (unsigned)n_problems, 1), 256, 0, stream>>>(table);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  {
    coupling_recip_batch_kernel<<<dim3((unsigned)((n_pts + 255) / 256), (unsigned)n_problems, 1), 256, 0, stream>>>(table);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  coupling_batch_kernel<<<dim3(tiles, tiles, (unsigned)n_problems), 64, 0, stream>>>(table);
  return hipGetLastError();
}

hipError_t launch_coupling(const CouplingArgs& a) {
  if (a.n_modes == 0) return hipSuccess;
  if (a.n_pts < 2) return hipErrorInvalidValue;
  const unsigned tiles = (unsigned)((a.n_modes + 7) / 8);
  if (tiles > 65535) return hipErrorInvalidValue;
  if (a.tame) {
    coupling_tame_kernel<<<(unsigned)a.n_modes, 256, 0, a.stream>>>(a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  coupling_recip_kernel<<<(unsigned)((a.n_pts + 255) / 256), 256, 0, a.stream>>>(a);
  {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  coupling_kernel<<<dim3(tiles, tiles, 1), 64, 0, a.stream>>>(a);
  return hipGetLastError();
}

}  // namespace storm
