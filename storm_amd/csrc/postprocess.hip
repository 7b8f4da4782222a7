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

}  // namespace storm
