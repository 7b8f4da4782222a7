// storm_host.cpp -- see storm_host.hpp.  Built with -ffp-contract=off.
#include "storm_host.hpp"

#include <algorithm>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>

#include <cmath>
#include <cstring>
#include <limits>

namespace storm_host {

namespace {
constexpr double kGrav = 6.67430e-8;  // "as defined by MESA", gsm.rs:13
constexpr double kPi = 3.14159265358979323846264338327950288;
inline double pow3(double x) { return (x * x) * x; }  // f64::powi(3)

void resize_all(DimensionlessProperties& d, size_t n) {
  d.r_coord.assign(n, 0.); d.v.assign(n, 0.); d.u.assign(n, 0.); d.gamma1.assign(n, 0.);
  d.a_star.assign(n, 0.); d.c1.assign(n, 0.); d.rot.assign(n, 0.);
  d.m_coord.assign(n, 0.); d.rho.assign(n, 0.); d.p.assign(n, 0.);
}

[[noreturn]] void raise_abi(const char* what, int rc) {
  throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + storm_last_error());
}
}  // namespace

double DimensionedProperties::freq_scale() const { return std::sqrt(grav * mass / pow3(radius)); }

DiscreteModel DiscreteModel::from_gsm_arrays(size_t n, const double* r, const double* m_r, const double* rho,
                                             const double* p, const double* gamma1, const double* n2,
                                             const double* omega_rot, double r_star, double m_star) {
  if (n < 2) throw std::runtime_error("GSM model needs at least 2 points");
  DiscreteModel out;
  out.scale = DimensionedProperties{r_star, m_star, kGrav};
  DimensionlessProperties& d = out.dimensionless;
  resize_all(d, n);
  const double rs3 = pow3(r_star);
  const double omega_dyn = std::sqrt(kGrav * m_star / rs3);
  const double four_pi = 4. * kPi;
  const double rs4 = (r_star * r_star) * (r_star * r_star);  // f64::powi(4)
  for (size_t k = 0; k < n; ++k) {
    const double r3 = pow3(r[k]);
    d.m_coord[k] = m_r[k] / m_star;                      // gsm.rs:95-97
    d.rho[k] = rho[k] / m_star * rs3;
    d.p[k] = p[k] / kGrav / (m_star * m_star) * rs4;
    d.rot[k] = omega_rot[k] / omega_dyn;
    d.c1[k] = r3 / rs3 * m_star / m_r[k];
    d.a_star[k] = r3 / (kGrav * m_r[k]) * n2[k];
    d.v[k] = kGrav * m_r[k] * rho[k] / (p[k] * r[k]);
    d.u[k] = four_pi * rho[k] * r3 / m_r[k];
    d.r_coord[k] = r[k] / r_star;
    d.gamma1[k] = gamma1[k];
  }
  // the centre is a removable singularity: analytic limits, gsm.rs:84-90
  d.c1[0] = m_star / rs3 * 3. / (four_pi * rho[0]);
  d.a_star[0] = 0.;
  d.v[0] = 0.;
  d.u[0] = 3.;
  return out;
}

DiscreteModel LinearInterpolator::eval(const std::vector<double>& grid) const {
  const DimensionlessProperties& src = model_.dimensionless;
  DiscreteModel out;
  out.scale = model_.scale;
  DimensionlessProperties& dst = out.dimensionless;
  resize_all(dst, grid.size());
  if (grid.empty()) throw std::runtime_error("LinearInterpolator::eval: empty grid");
  const std::vector<double>& x = src.r_coord;
  // Two cursors, one over the model grid and one over the requested points (both ascending),
  // advanced exactly like the reference loop (interpolate.rs:50-117).
  size_t feed = 0, lo = 0, hi = 0, q = 0;
  while (true) {
    const double want = grid[q];
    if (x[hi] == want) {
      dst.r_coord[q] = x[hi]; dst.v[q] = src.v[hi]; dst.u[q] = src.u[hi]; dst.gamma1[q] = src.gamma1[hi];
      dst.a_star[q] = src.a_star[hi]; dst.c1[q] = src.c1[hi]; dst.rot[q] = src.rot[hi];
      dst.m_coord[q] = src.m_coord[hi]; dst.rho[q] = src.rho[hi]; dst.p[q] = src.p[hi];
      if (++q == grid.size()) break;
      continue;
    }
    if (x[hi] > want) {
      const double pos = (want - x[lo]) / (x[hi] - x[lo]);
      auto lerp = [&](const std::vector<double>& f) { return f[lo] + pos * (f[hi] - f[lo]); };
      dst.r_coord[q] = want;
      dst.v[q] = lerp(src.v); dst.u[q] = lerp(src.u); dst.gamma1[q] = lerp(src.gamma1);
      dst.a_star[q] = lerp(src.a_star); dst.c1[q] = lerp(src.c1); dst.rot[q] = lerp(src.rot);
      dst.m_coord[q] = lerp(src.m_coord); dst.rho[q] = lerp(src.rho); dst.p[q] = lerp(src.p);
      if (++q == grid.size()) break;
      continue;
    }
    lo = hi;
    if (feed == x.size()) break;  // requested points beyond the model stay zero, like the reference
    hi = feed++;
  }
  return out;
}

namespace {
constexpr double kPoly0MaxX =
    2.449489742783178098197284074705891391965947480656670128432692567250960377457315;  // sqrt(6)
}

double Polytrope0::exact(uint64_t ell, int64_t radial_order) const {  // Pekeris 1938, polytrope.rs:178-186
  const double n = double(ell), k = 2. * double(radial_order);
  const double d = -2. + gamma1 / 4. * (k * (k + 5. + 2. * n) + 6. + 4. * n);
  return std::sqrt(d + std::sqrt(d * d + n * (n + 1.)));
}

DiscreteModel Polytrope0::eval(const std::vector<double>& grid) const {
  DiscreteModel out;
  DimensionlessProperties& d = out.dimensionless;
  resize_all(d, grid.size());
  const double phi_max = pow3(kPoly0MaxX) / 3.;  // polytrope.rs:221-225
  for (size_t k = 0; k < grid.size(); ++k) {
    const double xi = grid[k] * kPoly0MaxX;
    const double theta = 1. - xi * xi / 6.;
    const double phi = pow3(xi) / 3.;
    d.rho[k] = 3. / (4. * kPi);
    d.m_coord[k] = phi / phi_max;
    d.p[k] = std::numeric_limits<double>::infinity();
    d.r_coord[k] = grid[k];
    d.v[k] = phi / (xi * theta);
    d.a_star[k] = -phi / (xi * theta * gamma1);
    d.u[k] = 3.; d.gamma1[k] = gamma1; d.c1[k] = 1.; d.rot[k] = 0.;
  }
  if (!grid.empty()) { d.v[0] = 0.; d.a_star[0] = 0.; }  // polytrope.rs:210,219 (index 0 of the *evaluated* grid)
  return out;
}

std::vector<double> linspace(double lower, double upper, size_t n) {
  std::vector<double> out(n);
  for (size_t k = 0; k < n; ++k) out[k] = lower + (upper - lower) * double(k) / double(n - 1);
  return out;
}
std::vector<double> rev_linspace(double lower, double upper, size_t n) {
  std::vector<double> out = linspace(1. / lower, 1. / upper, n);
  for (double& x : out) x = 1. / x;
  return out;
}
std::vector<double> grid_of(const DiscreteModel& model, size_t scale) {
  const std::vector<double>& r = model.dimensionless.r_coord;
  std::vector<double> g;
  g.reserve((r.size() - 1) * scale + 1);
  for (size_t k = 0; k + 1 < r.size(); ++k)
    for (size_t s = 0; s < scale; ++s) g.push_back(r[k] + (r[k + 1] - r[k]) * double(s) / double(scale));
  g.push_back(r.back());
  return g;
}
std::vector<double> grid_of(const ContinuousModel& model, size_t npoints) {
  return linspace(model.inner(), model.outer(), npoints);
}

// ---- Lane-Emden integration for construct_polytrope (polytrope.rs:9-84) ----
namespace {
struct Deriv { double dphi, dtheta; };
Deriv lane_emden_rhs(double xi, double phi, double theta, double n) {
  const double dphi = std::pow(theta, n) * (xi * xi);
  return Deriv{dphi, xi == 0. ? 0. : -phi / (xi * xi)};
}
struct State { double phi, theta; };
State rk4_step(double xi, State s, double n, double h) {
  const double hh = h / 2.;
  const Deriv k1 = lane_emden_rhs(xi, s.phi, s.theta, n);
  const Deriv k2 = lane_emden_rhs(xi + hh, s.phi + hh * k1.dphi, s.theta + hh * k1.dtheta, n);
  const Deriv k3 = lane_emden_rhs(xi + hh, s.phi + hh * k2.dphi, s.theta + hh * k2.dtheta, n);
  const Deriv k4 = lane_emden_rhs(xi + h, s.phi + h * k3.dphi, s.theta + h * k3.dtheta, n);
  return State{s.phi + h / 6. * (k1.dphi + 2. * k2.dphi + 2. * k3.dphi + k4.dphi),
               s.theta + h / 6. * (k1.dtheta + 2. * k2.dtheta + 2. * k3.dtheta + k4.dtheta)};
}
inline double sgn(double f) { return f != f ? f : (std::signbit(f) ? -1. : 1.); }
inline uint64_t bits(double x) { uint64_t b; std::memcpy(&b, &x, 8); return b; }
inline double unbits(uint64_t b) { double x; std::memcpy(&x, &b, 8); return x; }

// InverseQuadratic::optimize (bracket.rs:151-222) for a scalar host function; used only to locate
// the surface of the polytrope (polytrope.rs:49-69).  The scan's brackets are refined inside
// libstorm_hip, not here.
template <class F>
BracketResult refine_host(Point lo, Point hi, F&& fn, uint64_t ulp) {
  bool have_prev = false;
  Point prev{0., 0.};
  uint64_t evals = 0;
  auto sec = [](Point a, Point b) { return (0. - b.f) / (a.f - b.f) * a.x + (0. - a.f) / (b.f - a.f) * b.x; };
  while (true) {
    double x;
    if (!have_prev) {
      x = sec(lo, hi);
    } else {
      const Point &a = lo, &b = hi, &c = prev;
      x = (0. - b.f) * (0. - c.f) / (a.f - b.f) / (a.f - c.f) * a.x +
          (0. - a.f) * (0. - c.f) / (b.f - a.f) / (b.f - c.f) * b.x +
          (0. - a.f) * (0. - b.f) / (c.f - b.f) / (c.f - a.f) * c.x;
      if (x <= lo.x || x >= hi.x) x = sec(lo, hi);
    }
    const uint64_t ub = bits(hi.x), lb = bits(lo.x);
    if (!(hi.x > lo.x) || !(lo.x >= 0.)) throw std::runtime_error("refine_host: invalid bracket");
    if (ub - lb <= ulp) return BracketResult{lo, hi, x, evals};
    uint64_t xb = bits(x);
    if (ub - lb <= 2 * ulp) xb = lb + (ub - lb) / 2;
    else if (ub < xb || ub - xb < ulp) xb = ub - ulp;
    else if (lb > xb || xb - lb < ulp) xb = lb + ulp;
    x = unbits(xb);
    const Point next{x, fn(x)};
    if (!std::isfinite(next.f)) throw std::runtime_error("refine_host: non-finite value");
    ++evals;
    if (sgn(hi.f) == sgn(next.f)) { prev = hi; hi = next; } else { prev = lo; lo = next; }
    have_prev = true;
  }
}
}  // namespace

DiscreteModel construct_polytrope(double n, double gamma1, double step) {
  if (!(n < 5.) || !(n >= 0.)) throw std::runtime_error("polytropic index must be in [0, 5)");
  std::vector<double> xi, theta, phi;
  State s{0., 1.};
  uint64_t count = 0;
  while (s.theta > 0.) {
    theta.push_back(s.theta); phi.push_back(s.phi); xi.push_back(double(count) * step);
    s = rk4_step(double(count) * step, s, n, step);
    ++count;
  }
  const double x_last = double(count - 1) * step;
  const State last{phi.back(), theta.back()};
  const BracketResult br = refine_host(Point{0., last.theta}, Point{step, s.theta},
                                       [&](double h) { return rk4_step(x_last, last, n, h).theta; }, 1);
  const State surf = rk4_step(x_last, last, n, br.lower.x);
  const double max_phi = surf.phi, max_xi = x_last + br.root;
  const size_t len = xi.size();  // the surface point itself is dropped (polytrope.rs:99-101)
  DiscreteModel out;
  DimensionlessProperties& d = out.dimensionless;
  resize_all(d, len);
  for (size_t k = 0; k < len; ++k) {
    d.u[k] = pow3(xi[k]) * std::pow(theta[k], n) / phi[k];
    d.v[k] = (n + 1.) * phi[k] / (xi[k] * theta[k]);
    d.c1[k] = pow3(xi[k]) / pow3(max_xi) * max_phi / phi[k];
    d.a_star[k] = -((n + 1.) / gamma1 - n) * phi[k] / (xi[k] * theta[k]);
    d.r_coord[k] = xi[k] / max_xi;
    d.gamma1[k] = gamma1;
    d.rot[k] = 0.;
    d.m_coord[k] = phi[k] / max_phi;                                                   // polytrope.rs:139
    d.rho[k] = std::pow(theta[k], n) * pow3(max_xi) / (max_phi * 4. * kPi);            // :143
  }
  d.u[0] = 3.; d.v[0] = 0.; d.a_star[0] = 0.;
  d.c1[0] = max_phi / pow3(max_xi) * 3.;
  const double kk = 4. * kPi / (std::pow(d.rho[0], 1. / n - 1.) * (n + 1.) * (max_xi * max_xi));  // :146
  for (size_t k = 0; k < len; ++k) d.p[k] = std::pow(d.rho[k], 1. + 1. / n) * kk;                 // :147
  return out;
}

// ---- ErasedSolver ----
namespace {
void abscissae(DifferenceSchemes scheme, std::vector<double>& out) {  // Stepper::points(), stepper.rs
  constexpr double c2 = 0.28867513459481287, c3 = 0.3872983346207417;
  constexpr double c4a = 0.8611363115940526 / 2., c4b = 0.33998104358485626 / 2.;
  switch (int(scheme) & 3) {
    case 0: out = {0.5}; break;
    case 1: out = {0.5 + -c2, 0.5 + c2}; break;
    case 2: out = {0.5 + -c3, 0.5 + 0., 0.5 + c3}; break;
    default: out = {0.5 + -c4a, 0.5 + -c4b, 0.5 + c4b, 0.5 + c4a}; break;
  }
}
}  // namespace

namespace {
// What DiscretizedSystemImpl::new (discretized.rs:81-123) keeps of the model: the arrays behind one storm_system_desc.
struct SystemArrays {
  std::vector<double> u, vg, as, c1, rot, dx;
  double inner_c1 = 0., inner_rot = 0., outer_u = 0.;
  size_t steps = 0, p = 0;
  SystemArrays(const ContinuousModel& model, DifferenceSchemes scheme, const std::vector<double>& grid) {
    if (grid.empty()) throw std::runtime_error("ErasedSolver::new: empty solver grid");
    std::vector<double> loc;
    abscissae(scheme, loc);
    p = loc.size(); steps = grid.size() - 1;
    // discretized.rs:94-100
    std::vector<double> delta(steps), xs(steps * p);
    for (size_t i = 0; i < steps; ++i) delta[i] = grid[i + 1] - grid[i];
    for (size_t i = 0; i < steps; ++i)
      for (size_t j = 0; j < p; ++j) xs[i * p + j] = grid[i] + loc[j] * delta[i];
    dx.resize(steps * p);
    if (steps) {
      const DiscreteModel at = model.eval(xs);
      const DimensionlessProperties& d = at.dimensionless;
      u = d.u; as = d.a_star; c1 = d.c1; rot = d.rot;
      vg.resize(steps * p);
      for (size_t k = 0; k < steps * p; ++k) vg[k] = d.v[k] / d.gamma1[k];  // adiabatic.rs:52
      for (size_t i = 0; i < steps; ++i)
        for (size_t j = 0; j < p; ++j) dx[i * p + j] = delta[i] / xs[i * p + j];
    }
    const DiscreteModel in = model.eval({model.inner()}), outm = model.eval({model.outer()});  // discretized.rs:119-120
    inner_c1 = in.dimensionless.c1[0];
    inner_rot = in.dimensionless.rot[0];
    outer_u = outm.dimensionless.u[0];
  }
  storm_system_desc desc(Rotating1D system, DifferenceSchemes scheme) const {
    storm_system_desc d{};
    d.scheme = int32_t(scheme);
    d.ell = system.ell; d.m = system.m;
    d.n_int = steps; d.p = p;
    d.u = u.data(); d.v_gamma = vg.data(); d.a_star = as.data(); d.c1 = c1.data();
    d.rot = rot.data(); d.delta_over_x = dx.data();
    d.inner_c1 = inner_c1; d.inner_rot = inner_rot; d.outer_u = outer_u;
    return d;
  }
};
}  // namespace

ErasedSolver::ErasedSolver(const ContinuousModel& model, Rotating1D system, DifferenceSchemes scheme,
                           const std::vector<double>& grid) {
  const SystemArrays arrays(model, scheme, grid);
  const storm_system_desc desc = arrays.desc(system, scheme);
  const int rc = storm_solver_create(&desc, &h_);
  if (rc != STORM_OK) raise_abi("storm_solver_create", rc);
  n_int_ = arrays.steps;
}

ErasedSolver::~ErasedSolver() { storm_solver_destroy(h_); }

MultiSolver::MultiSolver(const ContinuousModel& model, Rotating1D system, DifferenceSchemes scheme,
                         const std::vector<double>& grid, const std::vector<int>& devices) {
  const SystemArrays arrays(model, scheme, grid);
  const storm_system_desc desc = arrays.desc(system, scheme);
  const int rc = storm_multi_create(&desc, devices.data(), devices.size(), &h_);
  if (rc != STORM_OK) raise_abi("storm_multi_create", rc);
  n_int_ = arrays.steps;
  p_ = arrays.p;
}
MultiSolver::~MultiSolver() { storm_multi_destroy(h_); }
std::vector<BracketResult> MultiSolver::scan_and_optimize(const std::vector<double>& grid, Precision precision,
                                                          bool* had_invalid) const {
  std::vector<storm_bracket> raw(grid.size() ? grid.size() : 1);
  size_t n_out = 0;
  int invalid = 0;
  const int rc = storm_multi_scan_and_optimize(h_, grid.data(), grid.size(), precision.as_ulp(), raw.data(), raw.size(),
                                               &n_out, &invalid, nullptr, nullptr);
  if (rc != STORM_OK) raise_abi("storm_multi_scan_and_optimize", rc);
  if (had_invalid) *had_invalid = invalid != 0;
  std::vector<BracketResult> out(n_out);
  for (size_t k = 0; k < n_out; ++k)
    out[k] = BracketResult{Point{raw[k].lo_x, raw[k].lo_f}, Point{raw[k].hi_x, raw[k].hi_f}, raw[k].root, raw[k].evals};
  return out;
}
void MultiSolver::eigenvectors_into(const double* freqs, size_t n, double* out) const {
  const int rc = storm_multi_eigenvector_batch(h_, freqs, n, nullptr, out);
  if (rc != STORM_OK) raise_abi("storm_multi_eigenvector_batch", rc);
}

double ErasedSolver::det(double freq) const {
  double out = 0.;
  const int rc = storm_det_batch(h_, &freq, 1, &out);
  if (rc != STORM_OK) raise_abi("storm_det_batch", rc);
  return out;
}
std::vector<double> ErasedSolver::det(const std::vector<double>& freqs) const {
  std::vector<double> out(freqs.size());
  const int rc = storm_det_batch(h_, freqs.data(), freqs.size(), out.data());
  if (rc != STORM_OK) raise_abi("storm_det_batch", rc);
  return out;
}
std::vector<double> ErasedSolver::eigenvector(double freq) const { return eigenvectors({freq}); }
std::vector<double> ErasedSolver::eigenvectors(const std::vector<double>& freqs) const {
  std::vector<double> out(freqs.size() * (n_int_ + 1) * 4);
  const int rc = storm_eigenvector_batch(h_, freqs.data(), freqs.size(), nullptr, out.data());
  if (rc != STORM_OK) raise_abi("storm_eigenvector_batch", rc);
  return out;
}
void ErasedSolver::eigenvectors_into(const double* freqs, size_t n, double* out) const {
  const int rc = storm_eigenvector_batch(h_, freqs, n, nullptr, out);
  if (rc != STORM_OK) raise_abi("storm_eigenvector_batch", rc);
}
std::vector<BracketResult> ErasedSolver::scan_and_optimize(const std::vector<double>& grid, Precision precision,
                                                           bool* had_invalid) const {
  std::vector<storm_bracket> raw(grid.size() ? grid.size() : 1);
  size_t n_out = 0;
  int invalid = 0;
  const int rc = storm_scan_and_optimize(h_, grid.data(), grid.size(), precision.as_ulp(), raw.data(), raw.size(),
                                         &n_out, &invalid, nullptr, nullptr);
  if (rc != STORM_OK) raise_abi("storm_scan_and_optimize", rc);
  if (had_invalid) *had_invalid = invalid != 0;
  std::vector<BracketResult> out(n_out);
  for (size_t k = 0; k < n_out; ++k)
    out[k] = BracketResult{Point{raw[k].lo_x, raw[k].lo_f}, Point{raw[k].hi_x, raw[k].hi_f}, raw[k].root, raw[k].evals};
  return out;
}

std::vector<Solution> scan(const ContinuousModel& model, const std::vector<double>& solver_grid, uint64_t ell,
                           int64_t m, double lower, double upper, size_t steps, bool inverse, double precision,
                           DifferenceSchemes scheme) {
  ErasedSolver determinant(model, Rotating1D::make(ell, m), scheme, solver_grid);
  const std::vector<double> points = inverse ? rev_linspace(lower, upper, steps) : linspace(lower, upper, steps);
  const std::vector<BracketResult> res = determinant.scan_and_optimize(points, Precision::Relative(precision));
  std::vector<double> roots(res.size());
  for (size_t k = 0; k < res.size(); ++k) roots[k] = res[k].root;
  const std::vector<double> vecs = determinant.eigenvectors(roots);
  const size_t len = (determinant.len() + 1) * 4;
  std::vector<Solution> out(res.size());
  for (size_t k = 0; k < res.size(); ++k)
    out[k] = Solution{res[k], std::vector<double>(vecs.begin() + k * len, vecs.begin() + (k + 1) * len), ell, m};
  return out;
}

// ---- SolverSweep: many systems in lock-step (storm_sweep) ----
SolverSweep::SolverSweep(const std::vector<SweepSystem>& systems, DifferenceSchemes scheme,
                         const std::vector<double>& grid) {
  if (systems.empty()) throw std::runtime_error("SolverSweep: no systems");
  // the model enters through its values at the collocation abscissae only: systems that share a model (the 8 (ell, m)
  // of one rotation rate) share the interpolation
  std::vector<const ContinuousModel*> distinct;
  std::vector<size_t> which(systems.size());
  for (size_t k = 0; k < systems.size(); ++k) {
    if (!systems[k].model) throw std::runtime_error("SolverSweep: null model");
    size_t at = 0;
    while (at < distinct.size() && distinct[at] != systems[k].model) ++at;
    if (at == distinct.size()) distinct.push_back(systems[k].model);
    which[k] = at;
  }
  std::vector<std::unique_ptr<SystemArrays>> arrays(distinct.size());
  {
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex err_mu;
    auto work = [&] {
      for (size_t k; (k = next.fetch_add(1)) < distinct.size();) {
        try { arrays[k] = std::make_unique<SystemArrays>(*distinct[k], scheme, grid); }
        catch (...) { std::lock_guard<std::mutex> lock(err_mu); if (!err) err = std::current_exception(); }
      }
    };
    const size_t n_thr = std::min<size_t>({distinct.size(), size_t(std::max(1u, std::thread::hardware_concurrency())), size_t(16)});
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_thr; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (err) std::rethrow_exception(err);
  }
  std::vector<storm_system_desc> descs(systems.size());
  for (size_t k = 0; k < systems.size(); ++k) {
    descs[k] = arrays[which[k]]->desc(systems[k].system, scheme);
    ell_.push_back(systems[k].system.ell); m_.push_back(systems[k].system.m);
  }
  const int rc = storm_sweep_create(descs.data(), descs.size(), &h_);
  if (rc != STORM_OK) raise_abi("storm_sweep_create", rc);
  n_int_ = arrays[0]->steps;
}
SolverSweep::~SolverSweep() { storm_sweep_destroy(h_); }

namespace {
std::vector<size_t> flatten(const std::vector<std::vector<double>>& lists, size_t n_sys, std::vector<double>& flat) {
  if (lists.size() != n_sys) throw std::runtime_error("SolverSweep: one list per system expected");
  std::vector<size_t> off(n_sys + 1, 0);
  for (size_t u = 0; u < n_sys; ++u) off[u + 1] = off[u] + lists[u].size();
  flat.resize(off[n_sys]);
  for (size_t u = 0; u < n_sys; ++u) std::copy(lists[u].begin(), lists[u].end(), flat.begin() + off[u]);
  return off;
}
}  // namespace

std::vector<std::vector<double>> SolverSweep::det(const std::vector<std::vector<double>>& freqs) const {
  std::vector<double> flat;
  const std::vector<size_t> off = flatten(freqs, size(), flat);
  std::vector<double> out(flat.size());
  const int rc = storm_sweep_det(h_, off.data(), flat.data(), out.data());
  if (rc != STORM_OK) raise_abi("storm_sweep_det", rc);
  std::vector<std::vector<double>> res(size());
  for (size_t u = 0; u < size(); ++u) res[u].assign(out.begin() + off[u], out.begin() + off[u + 1]);
  return res;
}

std::vector<std::vector<BracketResult>> SolverSweep::scan_and_optimize(const std::vector<std::vector<double>>& grids,
                                                                       Precision precision,
                                                                       std::vector<bool>* had_invalid) const {
  std::vector<double> flat;
  const std::vector<size_t> off = flatten(grids, size(), flat);
  std::vector<storm_bracket> raw(flat.size() ? flat.size() : 1);
  std::vector<size_t> ooff(size() + 1, 0);
  std::vector<int> inv(size(), 0);
  const int rc = storm_sweep_scan_and_optimize(h_, off.data(), flat.data(), precision.as_ulp(), raw.data(), raw.size(),
                                               ooff.data(), inv.data(), nullptr, nullptr);
  if (rc != STORM_OK) raise_abi("storm_sweep_scan_and_optimize", rc);
  if (had_invalid) { had_invalid->assign(size(), false); for (size_t u = 0; u < size(); ++u) (*had_invalid)[u] = inv[u] != 0; }
  std::vector<std::vector<BracketResult>> res(size());
  for (size_t u = 0; u < size(); ++u)
    for (size_t k = ooff[u]; k < ooff[u + 1]; ++k)
      res[u].push_back(BracketResult{Point{raw[k].lo_x, raw[k].lo_f}, Point{raw[k].hi_x, raw[k].hi_f}, raw[k].root, raw[k].evals});
  return res;
}

std::vector<std::vector<double>> SolverSweep::eigenvectors(const std::vector<std::vector<double>>& roots) const {
  std::vector<double> flat;
  const std::vector<size_t> off = flatten(roots, size(), flat);
  const size_t len = (n_int_ + 1) * 4;
  std::vector<double> out(flat.size() * len);
  const int rc = storm_sweep_eigenvectors(h_, off.data(), flat.data(), nullptr, out.data());
  if (rc != STORM_OK) raise_abi("storm_sweep_eigenvectors", rc);
  std::vector<std::vector<double>> res(size());
  for (size_t u = 0; u < size(); ++u) res[u].assign(out.begin() + off[u] * len, out.begin() + off[u + 1] * len);
  return res;
}

std::vector<std::vector<Rotating1DPostprocessing>> SolverSweep::modes(const std::vector<const PostProcessor*>& posts,
                                                                      const std::vector<std::vector<double>>& roots,
                                                                      bool want_fields) const {
  return run_modes(posts, roots, want_fields, false);
}
std::vector<std::vector<Rotating1DPostprocessing>> SolverSweep::modes_resident(const std::vector<const PostProcessor*>& posts,
                                                                               const std::vector<std::vector<double>>& roots) const {
  return run_modes(posts, roots, false, true);
}

CouplingMatrices SolverSweep::coupling_matrices(const PostProcessor& post, const std::vector<ModeRef>& modes, int64_t m,
                                                const PerturbedMetric& metric) const {
  const size_t n = modes.size(), np = n_int_ + 1;
  if (resident_off_.empty()) throw std::runtime_error("SolverSweep::coupling_matrices: call modes_resident first");
  for (const std::vector<double>* v : {&metric.alpha, &metric.dalpha, &metric.ddalpha, &metric.beta, &metric.dbeta, &metric.ddbeta})
    if (v->size() != np) throw std::runtime_error("coupling_matrices: metric arrays must have the model's length");
  std::vector<size_t> index(n);
  std::vector<uint64_t> ell(n);
  std::vector<double> freq(n);
  for (size_t k = 0; k < n; ++k) {
    const ModeRef& r = modes[k];
    if (r.system >= size() || resident_off_[r.system] + r.mode >= resident_off_[r.system + 1])
      throw std::runtime_error("coupling_matrices: no such resident mode");
    index[k] = resident_off_[r.system] + r.mode;
    ell[k] = ell_[r.system];
    freq[k] = resident_freq_[index[k]];
  }
  storm_metric_desc md{metric.alpha.data(), metric.dalpha.data(), metric.ddalpha.data(),
                       metric.beta.data(), metric.dbeta.data(), metric.ddbeta.data()};
  CouplingMatrices c;
  double* out = c.allocate(n);
  const int rc = storm_sweep_coupling_matrices(h_, post.handle(), n, index.data(), ell.data(), freq.data(), m, &md, out);
  if (rc != STORM_OK) raise_abi("storm_sweep_coupling_matrices", rc);
  return c;
}

std::vector<CouplingMatrices> SolverSweep::coupling_matrices(const std::vector<CouplingProblem>& problems) const {
  const size_t np = n_int_ + 1, P = problems.size();
  if (resident_off_.empty()) throw std::runtime_error("SolverSweep::coupling_matrices: call modes_resident first");
  std::vector<size_t> off(P + 1, 0), index;
  std::vector<uint64_t> ell;
  std::vector<double> freq;
  std::vector<int64_t> m(P);
  std::vector<storm_metric_desc> md(P);
  std::vector<storm_postproc*> posts(P, nullptr);
  std::vector<CouplingMatrices> res(P);
  std::vector<double*> out_ptr(P, nullptr);
  for (size_t j = 0; j < P; ++j) {
    const CouplingProblem& pr = problems[j];
    if (!pr.model_on_grid || !pr.metric) throw std::runtime_error("coupling_matrices: null problem member");
    const PerturbedMetric& mt = *pr.metric;
    for (const std::vector<double>* v : {&mt.alpha, &mt.dalpha, &mt.ddalpha, &mt.beta, &mt.dbeta, &mt.ddbeta})
      if (v->size() != np) throw std::runtime_error("coupling_matrices: metric arrays must have the model's length");
    for (const ModeRef& r : pr.modes) {
      if (r.system >= size() || resident_off_[r.system] + r.mode >= resident_off_[r.system + 1])
        throw std::runtime_error("coupling_matrices: no such resident mode");
      index.push_back(resident_off_[r.system] + r.mode);
      ell.push_back(ell_[r.system]);
      freq.push_back(resident_freq_[index.back()]);
    }
    off[j + 1] = index.size();
    m[j] = pr.m;
    md[j] = storm_metric_desc{mt.alpha.data(), mt.dalpha.data(), mt.ddalpha.data(), mt.beta.data(), mt.dbeta.data(), mt.ddbeta.data()};
    posts[j] = pr.model_on_grid->handle();
    out_ptr[j] = res[j].allocate(pr.modes.size());  // uninitialised: the library writes every element
  }
  const int rc = storm_sweep_coupling_matrices_batch(h_, P, posts.data(), off.data(), index.data(), ell.data(), freq.data(), m.data(),
                                                     md.data(), out_ptr.data());
  if (rc != STORM_OK) raise_abi("storm_sweep_coupling_matrices_batch", rc);
  return res;
}

std::vector<std::vector<Rotating1DPostprocessing>> SolverSweep::run_modes(const std::vector<const PostProcessor*>& posts,
                                                                          const std::vector<std::vector<double>>& roots,
                                                                          bool want_fields, bool resident) const {
  if (posts.size() != size()) throw std::runtime_error("SolverSweep::modes: one PostProcessor per system expected");
  std::vector<double> flat;
  const std::vector<size_t> off = flatten(roots, size(), flat);
  const size_t n = flat.size(), np = n_int_ + 1;
  std::vector<storm_postproc*> handles(size(), nullptr);
  for (size_t u = 0; u < size(); ++u) {
    if (!posts[u] && off[u + 1] > off[u]) throw std::runtime_error("SolverSweep::modes: null PostProcessor");
    if (posts[u]) handles[u] = posts[u]->handle();
  }
  std::vector<double> fields(want_fields ? n * STORM_POST_FIELDS * np : 0), nodes;
  std::vector<storm_mode_id> ids(n ? n : 1);
  size_t cap = std::min(np, size_t(256));  // see kNodeCapFirst
  for (;; cap = np) {
    nodes.resize(n * cap);
    const int rc = resident ? storm_sweep_modes_resident(h_, handles.data(), off.data(), flat.data(), ids.data(), nodes.data(), cap)
                            : storm_sweep_modes(h_, handles.data(), off.data(), flat.data(), nullptr, nullptr,
                                                want_fields ? fields.data() : nullptr, ids.data(), nodes.data(), cap);
    if (rc != STORM_OK) raise_abi(resident ? "storm_sweep_modes_resident" : "storm_sweep_modes", rc);
    bool fits = true;
    for (size_t k = 0; k < n; ++k) fits = fits && ids[k].n_nodes <= cap;
    if (fits || cap == np) break;
  }
  resident_off_.clear(); resident_freq_.clear();
  if (resident) { resident_off_ = off; resident_freq_ = flat; }
  std::vector<std::vector<Rotating1DPostprocessing>> res(size());
  for (size_t u = 0; u < size(); ++u) {
    const size_t a = off[u], cnt = off[u + 1] - off[u];
    if (!cnt) continue;
    const std::vector<double> f(want_fields ? fields.begin() + a * STORM_POST_FIELDS * np : fields.begin(),
                                want_fields ? fields.begin() + (a + cnt) * STORM_POST_FIELDS * np : fields.begin());
    const std::vector<storm_mode_id> id(ids.begin() + a, ids.begin() + a + cnt);
    const std::vector<double> nd(nodes.begin() + a * cap, nodes.begin() + (a + cnt) * cap);
    res[u] = posts[u]->unpack(cnt, f, id, nd, cap);
  }
  return res;
}

// ---- post-processing ----
PostProcessor::PostProcessor(const DiscreteModel& model) : x_(model.dimensionless.r_coord) {
  const DimensionlessProperties& d = model.dimensionless;
  storm_model_desc desc{};
  desc.n_pts = d.r_coord.size();
  desc.r_coord = d.r_coord.data(); desc.m_coord = d.m_coord.data(); desc.rho = d.rho.data(); desc.p = d.p.data();
  desc.v = d.v.data(); desc.u = d.u.data(); desc.gamma1 = d.gamma1.data(); desc.a_star = d.a_star.data();
  desc.c1 = d.c1.data(); desc.rot = d.rot.data();
  const int rc = storm_postproc_create(&desc, &h_);
  if (rc != STORM_OK) raise_abi("storm_postproc_create", rc);
}
PostProcessor::~PostProcessor() { storm_postproc_destroy(h_); }

std::vector<Rotating1DPostprocessing> PostProcessor::unpack(size_t n, const std::vector<double>& fields,
                                                            const std::vector<storm_mode_id>& ids,
                                                            const std::vector<double>& nodes, size_t node_cap) const {
  const size_t np = x_.size();
  std::vector<Rotating1DPostprocessing> out(n);
  for (size_t k = 0; k < n; ++k) {
    Rotating1DPostprocessing& r = out[k];
    if (!fields.empty()) {
      r.x = x_;  // (ids-only results carry no per-point arrays at all)
      std::vector<double>* dst[STORM_POST_FIELDS] = {&r.y1, &r.y2, &r.y3, &r.y4, &r.xi_r, &r.xi_h, &r.xi_tn, &r.xi_tp,
                                                     &r.psi, &r.dpsi, &r.rho, &r.p, &r.chi};
      for (int f = 0; f < STORM_POST_FIELDS; ++f) {
        const double* src = fields.data() + (k * STORM_POST_FIELDS + f) * np;
        dst[f]->assign(src, src + np);
      }
    }
    r.cross_clockwise = ids[k].cross_clockwise;
    r.cross_counter_clockwise = ids[k].cross_counter_clockwise;
    r.radial_order = ids[k].radial_order;
    const size_t nn = ids[k].n_nodes < node_cap ? size_t(ids[k].n_nodes) : node_cap;
    r.nodes.assign(nodes.begin() + k * node_cap, nodes.begin() + k * node_cap + nn);
  }
  return out;
}

std::vector<Rotating1DPostprocessing> PostProcessor::post_process(const std::vector<Solution>& sols) const {
  const size_t n = sols.size(), np = x_.size();
  if (n == 0) return {};
  std::vector<double> freq(n), vec(n * np * 4);
  for (size_t k = 0; k < n; ++k) {
    if (sols[k].ell != sols[0].ell || sols[k].m != sols[0].m)
      throw std::runtime_error("post_process: one call handles the solutions of one (ell, m)");
    if (sols[k].eigenvector.size() != np * 4) throw std::runtime_error("post_process: eigenvector is not on the model grid");
    freq[k] = sols[k].bracket.root;
    std::memcpy(vec.data() + k * np * 4, sols[k].eigenvector.data(), np * 4 * sizeof(double));
  }
  std::vector<double> fields(n * STORM_POST_FIELDS * np), nodes(n * np);
  std::vector<storm_mode_id> ids(n);
  const int rc = storm_postprocess_batch(h_, sols[0].ell, sols[0].m, freq.data(), n, vec.data(), fields.data(), ids.data(),
                                         nodes.data(), np);
  if (rc != STORM_OK) raise_abi("storm_postprocess_batch", rc);
  return unpack(n, fields, ids, nodes, np);
}

namespace {
// Radial nodes come back as [mode][cap]; a mode has a few to a few hundred of them, not n_pts.  Start with a small
// capacity and repeat the call with the full one only if some mode has more (n_nodes is always the true count).
constexpr size_t kNodeCapFirst = 256;
}  // namespace

std::vector<Rotating1DPostprocessing> PostProcessor::modes(const ErasedSolver& solver, const std::vector<double>& roots,
                                                           bool want_fields) const {
  const size_t n = roots.size(), np = x_.size();
  if (n == 0) return {};
  std::vector<double> fields(want_fields ? n * STORM_POST_FIELDS * np : 0), nodes;
  std::vector<storm_mode_id> ids(n);
  for (size_t cap = std::min(np, kNodeCapFirst);; cap = np) {
    nodes.resize(n * cap);
    const int rc = storm_modes_batch(solver.handle(), h_, roots.data(), n, nullptr, nullptr,
                                     want_fields ? fields.data() : nullptr, ids.data(), nodes.data(), cap);
    if (rc != STORM_OK) raise_abi("storm_modes_batch", rc);
    bool fits = true;
    for (const storm_mode_id& id : ids) fits = fits && id.n_nodes <= cap;
    if (fits || cap == np) return unpack(n, fields, ids, nodes, cap);
  }
}

// ---- perturb_structure (perturbed.rs:324-460) ----
namespace {
// nalgebra's closed-form inverses of 2x2 and 3x3 matrices (row-major m[r][c] here); false if the determinant is zero.
bool invert2(const double (&m)[2][2], double (&o)[2][2]) {
  const double det = m[0][0] * m[1][1] - m[1][0] * m[0][1];
  if (det == 0.) return false;
  o[0][0] = m[1][1] / det; o[0][1] = -m[0][1] / det; o[1][0] = -m[1][0] / det; o[1][1] = m[0][0] / det;
  return true;
}
bool invert3(const double (&m)[3][3], double (&o)[3][3]) {
  const double c0 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
  const double c1 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
  const double c2 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
  const double det = m[0][0] * c0 - m[0][1] * c1 + m[0][2] * c2;
  if (det == 0.) return false;
  o[0][0] = c0 / det;
  o[0][1] = (m[0][2] * m[2][1] - m[2][2] * m[0][1]) / det;
  o[0][2] = (m[0][1] * m[1][2] - m[1][1] * m[0][2]) / det;
  o[1][0] = -c1 / det;
  o[1][1] = (m[0][0] * m[2][2] - m[2][0] * m[0][2]) / det;
  o[1][2] = (m[0][2] * m[1][0] - m[1][2] * m[0][0]) / det;
  o[2][0] = c2 / det;
  o[2][1] = (m[0][1] * m[2][0] - m[2][1] * m[0][0]) / det;
  o[2][2] = (m[0][0] * m[1][1] - m[1][0] * m[0][1]) / det;
  return true;
}
// nalgebra product a*b for N x N: every result column is an axpy over the columns of a, k ascending
template <int N>
void product(const double (&a)[N][N], const double (&b)[N][N], double (&c)[N][N]) {
  for (int j = 0; j < N; ++j) {
    for (int r = 0; r < N; ++r) c[r][j] = a[r][0] * b[0][j];
    for (int k = 1; k < N; ++k)
      for (int r = 0; r < N; ++r) c[r][j] = a[r][k] * b[k][j] + c[r][j];
  }
}
}  // namespace

PerturbedMetric perturb_structure(const DiscreteModel& model, double rot) {
  const DimensionlessProperties& d = model.dimensionless;
  const size_t np = d.r_coord.size();
  if (np < 2) throw std::runtime_error("perturb_structure: the model needs at least two points");
  const std::vector<double>& x = d.r_coord;
  auto dlnrho = [&](size_t i) { return -d.a_star[i] - d.v[i] / d.gamma1[i]; };
  auto kappa = [&](size_t i) { return 4. * kPi / d.m_coord[i] * d.rho[i] * x[i] * dlnrho(i); };
  // the two linear recurrences (P0 and P2 parts of the potential perturbation), implicit midpoint steps
  std::vector<double> p0a(np, 0.), p0b(np, 0.), p2a(np, 0.), p2b(np, 0.);
  p2a[1] = 1.; p2b[1] = 2.;
  for (size_t i = 1; i < np; ++i) {
    const double delta = x[i] - x[i - 1], mid = 0.5 * (x[i] + x[i - 1]);
    const double k = kappa(i), k_prev = i == 1 ? 0. : kappa(i - 1);
    const double w = 0.5 * delta / mid;
    const double coupling = 0.5 * (mid * mid) * (k + k_prev);
    {
      const double a[3][3] = {{w * -1., w * 1., w * 0.}, {w * coupling, w * -2., w * mid}, {w * 0., w * 0., w * 0.}};
      double lhs[3][3], rhs[3][3], inv[3][3], step[3][3];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { const double e = r == c ? 1. : 0.; lhs[r][c] = e - a[r][c]; rhs[r][c] = e + a[r][c]; }
      if (!invert3(lhs, inv)) throw std::runtime_error("perturb_structure: singular step (the reference unwraps try_inverse)");
      product<3>(inv, rhs, step);
      const double ya = p0a[i - 1], yb = p0b[i - 1];
      p0a[i] = (step[0][1] * yb + step[0][0] * ya) + step[0][2];
      p0b[i] = (step[1][1] * yb + step[1][0] * ya) + step[1][2];
    }
    if (i != 1) {
      const double a[2][2] = {{w * -1., w * 1.}, {w * (6. + coupling), w * -2.}};
      double lhs[2][2], rhs[2][2], inv[2][2], step[2][2];
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) { const double e = r == c ? 1. : 0.; lhs[r][c] = e - a[r][c]; rhs[r][c] = e + a[r][c]; }
      if (!invert2(lhs, inv)) throw std::runtime_error("perturb_structure: singular step (the reference unwraps try_inverse)");
      product<2>(inv, rhs, step);
      const double ya = p2a[i - 1], yb = p2b[i - 1];
      p2a[i] = step[0][1] * yb + step[0][0] * ya;
      p2b[i] = step[1][1] * yb + step[1][0] * ya;
    }
  }
  const double a2 = -5. / 6. / (3. * p2a[np - 1] + p2b[np - 1]);
  PerturbedMetric pm;
  pm.rot = rot;
  for (std::vector<double>* v : {&pm.alpha, &pm.dalpha, &pm.ddalpha, &pm.beta, &pm.dbeta, &pm.ddbeta}) v->assign(np, 0.);
  const double rot2 = rot * rot;
  double mass = 0.;
  for (size_t i = 1; i < np; ++i) {
    const double dl = dlnrho(i), k = kappa(i), u = d.u[i], xi = x[i];
    const double trap = i + 1 < np ? 0.5 * (x[i + 1] - x[i - 1]) : 0.5 * (x[np - 1] - x[np - 2]);
    auto metric = [&](double psi, double dpsi, double ddpsi, double& f, double& df, double& ddf) {
      const double g = xi * dpsi / psi;
      f = 2. * xi / d.m_coord[i] * psi * rot2;
      df = f * (1. - u + g);
      ddf = df * (1. - u + g) + f * (-3. * u - u * dl + u * u + g - g * g + (xi * xi) * ddpsi / psi);
    };
    metric(p0a[i] * xi, p0b[i], ((xi * xi) * k * p0a[i] - 2. * p0b[i]) / xi + 1., pm.alpha[i], pm.dalpha[i], pm.ddalpha[i]);
    metric(p2a[i] * xi * a2, p2b[i] * a2, ((6. + (xi * xi) * k) * p2a[i] - 2. * p2b[i]) / xi * a2, pm.beta[i], pm.dbeta[i],
           pm.ddbeta[i]);
    mass += trap * d.rho[i] * (xi * xi) * (3. * pm.alpha[i] + pm.dalpha[i]);
  }
  pm.mass_delta = mass * (4. * kPi);
  return pm;
}

CouplingMatrices PostProcessor::coupling_matrices(const std::vector<ModeToPerturb>& modes, int64_t m,
                                                  const PerturbedMetric& metric) const {
  const size_t n = modes.size(), np = x_.size();
  CouplingMatrices out;
  double* flat = out.allocate(n);
  if (n == 0) return out;
  std::vector<uint64_t> ell(n);
  std::vector<double> freq(n), fields(n * STORM_POST_FIELDS * np);
  for (size_t k = 0; k < n; ++k) {
    const Rotating1DPostprocessing& r = *modes[k].post_processing;
    const std::vector<double>* src[STORM_POST_FIELDS] = {&r.y1, &r.y2, &r.y3, &r.y4, &r.xi_r, &r.xi_h, &r.xi_tn, &r.xi_tp,
                                                         &r.psi, &r.dpsi, &r.rho, &r.p, &r.chi};
    for (int f = 0; f < STORM_POST_FIELDS; ++f) {
      if (src[f]->size() != np) throw std::runtime_error("coupling_matrices: a mode was post-processed on another grid");
      std::memcpy(fields.data() + (k * STORM_POST_FIELDS + f) * np, src[f]->data(), np * sizeof(double));
    }
    ell[k] = modes[k].ell;
    freq[k] = modes[k].freq;
  }
  for (const std::vector<double>* a : {&metric.alpha, &metric.dalpha, &metric.ddalpha, &metric.beta, &metric.dbeta, &metric.ddbeta})
    if (a->size() != np) throw std::runtime_error("coupling_matrices: the metric lives on another grid");
  const storm_metric_desc md{metric.alpha.data(), metric.dalpha.data(), metric.ddalpha.data(),
                             metric.beta.data(), metric.dbeta.data(), metric.ddbeta.data()};
  const int rc = storm_coupling_matrices(h_, n, ell.data(), freq.data(), fields.data(), m, &md, flat);
  if (rc != STORM_OK) raise_abi("storm_coupling_matrices", rc);
  return out;
}

Rotating1DPostprocessing Rotating1DPostprocessing::make(double freq, const std::vector<double>& eigenvector, uint64_t ell,
                                                        int64_t m, const DiscreteModel& model) {
  if (eigenvector.size() % 4 != 0) throw std::runtime_error("eigenvector length is not a multiple of 4");
  PostProcessor pp(model);
  Solution s{BracketResult{Point{freq, 0.}, Point{freq, 0.}, freq, 0}, eigenvector, ell, m};
  return std::move(pp.post_process({s})[0]);
}

}  // namespace storm_host
