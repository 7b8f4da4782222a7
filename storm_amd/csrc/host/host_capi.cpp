// host_capi.cpp -- flat C view of storm_host.hpp for the ctypes tests and bench.py
// (libstorm_host.so).  Not part of the drop-in boundary (that is include/storm_hip.h); it only
// lets Python drive the C++ host mirror of the reference's Rust interface.
#include <cstring>
#include <string>

#include "storm_host.hpp"

using namespace storm_host;

namespace {
thread_local std::string g_err;

struct HostModel {
  DiscreteModel discrete;
  bool is_poly0 = false;
  double gamma1 = 0.;
  std::unique_ptr<ContinuousModel> continuous() const {
    if (is_poly0) return std::make_unique<Polytrope0>(gamma1);
    return std::make_unique<LinearInterpolator>(discrete);
  }
};

template <class F>
int guarded(F&& f) {
  try { f(); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
  catch (...) { g_err = "unknown error"; return -1; }
}
}  // namespace

extern "C" {

const char* storm_host_last_error() { return g_err.c_str(); }

void* storm_host_model_from_gsm(size_t n, const double* r, const double* m_r, const double* rho, const double* p,
                                const double* gamma1, const double* n2, const double* omega_rot, double r_star,
                                double m_star) {
  HostModel* h = new HostModel();
  if (guarded([&] { h->discrete = DiscreteModel::from_gsm_arrays(n, r, m_r, rho, p, gamma1, n2, omega_rot, r_star, m_star); })) {
    delete h; return nullptr;
  }
  return h;
}
void* storm_host_model_polytrope(double n, double gamma1, double step) {
  HostModel* h = new HostModel();
  if (guarded([&] { h->discrete = construct_polytrope(n, gamma1, step); })) { delete h; return nullptr; }
  return h;
}
void* storm_host_model_poly0(double gamma1) {
  HostModel* h = new HostModel();
  h->is_poly0 = true; h->gamma1 = gamma1;
  return h;
}
void storm_host_model_free(void* m) { delete static_cast<HostModel*>(m); }
size_t storm_host_model_len(void* m) { return static_cast<HostModel*>(m)->discrete.len(); }
// field: 0 r_coord, 1 v, 2 u, 3 gamma1, 4 a_star, 5 c1, 6 rot, 7 m_coord, 8 rho, 9 p
int storm_host_model_get(void* m, int field, double* out) {
  DimensionlessProperties& d = static_cast<HostModel*>(m)->discrete.dimensionless;
  std::vector<double>* f[] = {&d.r_coord, &d.v, &d.u, &d.gamma1, &d.a_star, &d.c1, &d.rot, &d.m_coord, &d.rho, &d.p};
  if (field < 0 || field > 9) return -1;
  std::memcpy(out, f[field]->data(), f[field]->size() * sizeof(double));
  return 0;
}
int storm_host_model_set(void* m, int field, const double* in) {
  DimensionlessProperties& d = static_cast<HostModel*>(m)->discrete.dimensionless;
  std::vector<double>* f[] = {&d.r_coord, &d.v, &d.u, &d.gamma1, &d.a_star, &d.c1, &d.rot, &d.m_coord, &d.rho, &d.p};
  if (field < 0 || field > 9) return -1;
  std::memcpy(f[field]->data(), in, f[field]->size() * sizeof(double));
  return 0;
}
// set-rotation-constant with dynamical units (src/main.rs:770-791)
int storm_host_model_set_rotation_constant(void* m, double value) {
  HostModel* h = static_cast<HostModel*>(m);
  if (h->is_poly0) { g_err = "Setting the rotation profile for analytical polytropes is not supported"; return -1; }
  for (double& r : h->discrete.dimensionless.rot) r = value;
  return 0;
}
size_t storm_host_model_grid(void* m, size_t scale, double* out, size_t cap) {
  HostModel* h = static_cast<HostModel*>(m);
  std::vector<double> g = h->is_poly0 ? grid_of(*h->continuous(), scale) : grid_of(h->discrete, scale);
  if (out) std::memcpy(out, g.data(), std::min(cap, g.size()) * sizeof(double));
  return g.size();
}

void* storm_host_solver_new(void* m, uint64_t ell, int64_t mm, int scheme, const double* grid, size_t ngrid) {
  HostModel* h = static_cast<HostModel*>(m);
  ErasedSolver* s = nullptr;
  if (guarded([&] {
        if (scheme < 0 || scheme > 7) throw std::runtime_error("unknown difference scheme");
        auto cm = h->continuous();
        s = new ErasedSolver(*cm, Rotating1D::make(ell, mm), DifferenceSchemes(scheme),
                             std::vector<double>(grid, grid + ngrid));
      })) return nullptr;
  return s;
}
void storm_host_solver_free(void* s) { delete static_cast<ErasedSolver*>(s); }

// MultiSolver: the same solver over several devices from this process (storm_multi_*)
void* storm_host_multi_new(void* m, uint64_t ell, int64_t mm, int scheme, const double* grid, size_t ngrid,
                           const int* devices, size_t n_dev) {
  HostModel* h = static_cast<HostModel*>(m);
  MultiSolver* s = nullptr;
  if (guarded([&] {
        if (scheme < 0 || scheme > 7) throw std::runtime_error("unknown difference scheme");
        auto cm = h->continuous();
        s = new MultiSolver(*cm, Rotating1D::make(ell, mm), DifferenceSchemes(scheme),
                            std::vector<double>(grid, grid + ngrid), std::vector<int>(devices, devices + n_dev));
      })) return nullptr;
  return s;
}
void storm_host_multi_free(void* s) { delete static_cast<MultiSolver*>(s); }
size_t storm_host_multi_len(void* s) { return static_cast<MultiSolver*>(s)->len(); }
size_t storm_host_multi_points(void* s) { return static_cast<MultiSolver*>(s)->points(); }
void* storm_host_multi_handle(void* s) { return static_cast<MultiSolver*>(s)->handle(); }
size_t storm_host_solver_len(void* s) { return static_cast<ErasedSolver*>(s)->len(); }
void* storm_host_solver_handle(void* s) { return static_cast<ErasedSolver*>(s)->handle(); }

int storm_host_solver_det(void* s, const double* f, size_t n, double* out) {
  return guarded([&] {
    std::vector<double> r = static_cast<ErasedSolver*>(s)->det(std::vector<double>(f, f + n));
    std::memcpy(out, r.data(), n * sizeof(double));
  });
}
// precision_kind: 0 = Precision::ULP(precision_ulp), 1 = Precision::Relative(precision_rel)
int storm_host_solver_scan_and_optimize(void* s, const double* f, size_t n, int precision_kind, uint64_t precision_ulp,
                                        double precision_rel, storm_bracket* out, size_t cap, size_t* n_out,
                                        int* had_invalid) {
  return guarded([&] {
    bool inv = false;
    const Precision prec = precision_kind == 0 ? Precision::ULP(precision_ulp) : Precision::Relative(precision_rel);
    std::vector<BracketResult> r =
        static_cast<ErasedSolver*>(s)->scan_and_optimize(std::vector<double>(f, f + n), prec, &inv);
    *n_out = r.size();
    if (had_invalid) *had_invalid = inv ? 1 : 0;
    for (size_t k = 0; k < r.size() && k < cap; ++k)
      out[k] = storm_bracket{r[k].lower.x, r[k].lower.f, r[k].upper.x, r[k].upper.f, r[k].root, r[k].evals};
  });
}
int storm_host_solver_eigenvectors(void* s, const double* f, size_t n, double* out) {
  return guarded([&] {
    static_cast<ErasedSolver*>(s)->eigenvectors_into(f, n, out);
  });
}

// StormState::scan: returns the number of modes; brackets [cap], eigenvectors [cap][(len+1)*4] (may be NULL).
int storm_host_scan(void* m, const double* grid, size_t ngrid, uint64_t ell, int64_t mm, double lower, double upper,
                    size_t steps, int inverse, double precision, int scheme, storm_bracket* out, double* vec_out,
                    size_t cap, size_t* n_out) {
  HostModel* h = static_cast<HostModel*>(m);
  return guarded([&] {
    auto cm = h->continuous();
    std::vector<Solution> sol = scan(*cm, std::vector<double>(grid, grid + ngrid), ell, mm, lower, upper, steps,
                                     inverse != 0, precision, DifferenceSchemes(scheme));
    *n_out = sol.size();
    for (size_t k = 0; k < sol.size() && k < cap; ++k) {
      const BracketResult& b = sol[k].bracket;
      out[k] = storm_bracket{b.lower.x, b.lower.f, b.upper.x, b.upper.f, b.root, b.evals};
      if (vec_out) std::memcpy(vec_out + k * sol[k].eigenvector.size(), sol[k].eigenvector.data(),
                               sol[k].eigenvector.size() * sizeof(double));
    }
  });
}

// `input.as_continuous().eval(&input.grid())` (src/main.rs:892): a new model handle on `grid`.
void* storm_host_model_eval(void* m, const double* grid, size_t ngrid) {
  HostModel* h = static_cast<HostModel*>(m);
  HostModel* o = new HostModel();
  if (guarded([&] { o->discrete = h->continuous()->eval(std::vector<double>(grid, grid + ngrid)); })) { delete o; return nullptr; }
  return o;
}

void* storm_host_postproc_new(void* m) {
  PostProcessor* p = nullptr;
  if (guarded([&] { p = new PostProcessor(static_cast<HostModel*>(m)->discrete); })) return nullptr;
  return p;
}
void storm_host_postproc_free(void* p) { delete static_cast<PostProcessor*>(p); }

namespace {
void flatten(const std::vector<Rotating1DPostprocessing>& r, size_t np, double* fields_out, int64_t* ids_out,
             uint64_t* n_nodes_out, double* nodes_out, size_t node_cap) {
  for (size_t k = 0; k < r.size(); ++k) {
    const std::vector<double>* src[STORM_POST_FIELDS] = {&r[k].y1, &r[k].y2, &r[k].y3, &r[k].y4, &r[k].xi_r, &r[k].xi_h,
                                                         &r[k].xi_tn, &r[k].xi_tp, &r[k].psi, &r[k].dpsi, &r[k].rho,
                                                         &r[k].p, &r[k].chi};
    if (fields_out)
      for (int f = 0; f < STORM_POST_FIELDS; ++f)
        if (src[f]->size() == np) std::memcpy(fields_out + (k * STORM_POST_FIELDS + f) * np, src[f]->data(), np * sizeof(double));
    ids_out[k * 3] = int64_t(r[k].cross_clockwise); ids_out[k * 3 + 1] = int64_t(r[k].cross_counter_clockwise);
    ids_out[k * 3 + 2] = r[k].radial_order;
    n_nodes_out[k] = r[k].nodes.size();
    if (nodes_out) std::memcpy(nodes_out + k * node_cap, r[k].nodes.data(), std::min(node_cap, r[k].nodes.size()) * sizeof(double));
  }
}
}  // namespace

// StormState::post_process over n solutions of one (ell, m): roots [n], eigenvectors [n][n_pts*4].
int storm_host_post_process(void* p, uint64_t ell, int64_t mm, const double* roots, const double* vecs, size_t n,
                            double* fields_out, int64_t* ids_out, uint64_t* n_nodes_out, double* nodes_out, size_t node_cap) {
  PostProcessor* pp = static_cast<PostProcessor*>(p);
  return guarded([&] {
    const size_t len = pp->len() * 4;
    std::vector<Solution> sols(n);
    for (size_t k = 0; k < n; ++k)
      sols[k] = Solution{BracketResult{Point{roots[k], 0.}, Point{roots[k], 0.}, roots[k], 0},
                         std::vector<double>(vecs + k * len, vecs + (k + 1) * len), ell, mm};
    flatten(pp->post_process(sols), pp->len(), fields_out, ids_out, n_nodes_out, nodes_out, node_cap);
  });
}
// Eigenvectors + post-processing in one device pass (PostProcessor::modes).
int storm_host_modes(void* p, void* solver, const double* roots, size_t n, int want_fields, double* fields_out,
                     int64_t* ids_out, uint64_t* n_nodes_out, double* nodes_out, size_t node_cap) {
  PostProcessor* pp = static_cast<PostProcessor*>(p);
  return guarded([&] {
    flatten(pp->modes(*static_cast<ErasedSolver*>(solver), std::vector<double>(roots, roots + n), want_fields != 0),
            pp->len(), want_fields ? fields_out : nullptr, ids_out, n_nodes_out, nodes_out, node_cap);
  });
}

// ---- SolverSweep: per-system lists arrive concatenated with offsets [n_sys + 1] ----
namespace {
std::vector<std::vector<double>> split(const size_t* off, const double* f, size_t n_sys) {
  std::vector<std::vector<double>> lists(n_sys);
  for (size_t u = 0; u < n_sys; ++u) lists[u].assign(f + off[u], f + off[u + 1]);
  return lists;
}
}  // namespace

void* storm_host_sweep_new(void* const* models, const uint64_t* ell, const int64_t* mm, size_t n_sys, int scheme,
                           const double* grid, size_t ngrid) {
  SolverSweep* s = nullptr;
  if (guarded([&] {
        if (scheme < 0 || scheme > 7) throw std::runtime_error("unknown difference scheme");
        // one continuous model per distinct handle, so systems sharing a model share its interpolation
        std::vector<void*> distinct;
        std::vector<std::unique_ptr<ContinuousModel>> cms;
        std::vector<SweepSystem> systems(n_sys);
        for (size_t u = 0; u < n_sys; ++u) {
          size_t at = 0;
          while (at < distinct.size() && distinct[at] != models[u]) ++at;
          if (at == distinct.size()) { distinct.push_back(models[u]); cms.push_back(static_cast<HostModel*>(models[u])->continuous()); }
          systems[u] = SweepSystem{cms[at].get(), Rotating1D::make(ell[u], mm[u])};
        }
        s = new SolverSweep(systems, DifferenceSchemes(scheme), std::vector<double>(grid, grid + ngrid));
      })) return nullptr;
  return s;
}
void storm_host_sweep_free(void* s) { delete static_cast<SolverSweep*>(s); }
size_t storm_host_sweep_len(void* s) { return static_cast<SolverSweep*>(s)->len(); }
void* storm_host_sweep_handle(void* s) { return static_cast<SolverSweep*>(s)->handle(); }

int storm_host_sweep_det(void* s, const size_t* off, const double* f, double* out) {
  SolverSweep* sw = static_cast<SolverSweep*>(s);
  return guarded([&] {
    const auto r = sw->det(split(off, f, sw->size()));
    for (size_t u = 0; u < sw->size(); ++u) std::memcpy(out + off[u], r[u].data(), r[u].size() * sizeof(double));
  });
}
int storm_host_sweep_scan_and_optimize(void* s, const size_t* off, const double* f, int precision_kind,
                                       uint64_t precision_ulp, double precision_rel, storm_bracket* out, size_t cap,
                                       size_t* out_off, int* had_invalid) {
  SolverSweep* sw = static_cast<SolverSweep*>(s);
  return guarded([&] {
    const Precision prec = precision_kind == 0 ? Precision::ULP(precision_ulp) : Precision::Relative(precision_rel);
    std::vector<bool> inv;
    const auto r = sw->scan_and_optimize(split(off, f, sw->size()), prec, &inv);
    out_off[0] = 0;
    for (size_t u = 0; u < sw->size(); ++u) {
      out_off[u + 1] = out_off[u] + r[u].size();
      if (had_invalid) had_invalid[u] = inv[u] ? 1 : 0;
      for (size_t k = 0; k < r[u].size(); ++k)
        if (out_off[u] + k < cap) {
          const BracketResult& b = r[u][k];
          out[out_off[u] + k] = storm_bracket{b.lower.x, b.lower.f, b.upper.x, b.upper.f, b.root, b.evals};
        }
    }
  });
}
int storm_host_sweep_eigenvectors(void* s, const size_t* off, const double* f, double* out) {
  SolverSweep* sw = static_cast<SolverSweep*>(s);
  return guarded([&] {
    const size_t len = (sw->len() + 1) * 4;
    const auto r = sw->eigenvectors(split(off, f, sw->size()));
    for (size_t u = 0; u < sw->size(); ++u) std::memcpy(out + off[u] * len, r[u].data(), r[u].size() * sizeof(double));
  });
}
int storm_host_sweep_modes(void* s, void* const* posts, const size_t* off, const double* f, int want_fields,
                           double* fields_out, int64_t* ids_out, uint64_t* n_nodes_out, double* nodes_out, size_t node_cap) {
  SolverSweep* sw = static_cast<SolverSweep*>(s);
  return guarded([&] {
    std::vector<const PostProcessor*> pp(sw->size());
    for (size_t u = 0; u < sw->size(); ++u) pp[u] = static_cast<const PostProcessor*>(posts[u]);
    const auto r = sw->modes(pp, split(off, f, sw->size()), want_fields != 0);
    const size_t np = sw->len() + 1;
    for (size_t u = 0; u < sw->size(); ++u)
      flatten(r[u], np, want_fields ? fields_out + off[u] * STORM_POST_FIELDS * np : nullptr, ids_out + off[u] * 3,
              n_nodes_out + off[u], nodes_out ? nodes_out + off[u] * node_cap : nullptr, node_cap);
  });
}

// perturb_structure through the mirror: out6 = alpha, dalpha, ddalpha, beta, dbeta, ddbeta (each [n_pts]).
int storm_host_perturb_structure(void* m, double rot, double* const* out6, double* mass_delta) {
  return guarded([&] {
    const PerturbedMetric pm = perturb_structure(static_cast<HostModel*>(m)->discrete, rot);
    const std::vector<double>* src[6] = {&pm.alpha, &pm.dalpha, &pm.ddalpha, &pm.beta, &pm.dbeta, &pm.ddbeta};
    for (int k = 0; k < 6; ++k) std::memcpy(out6[k], src[k]->data(), src[k]->size() * sizeof(double));
    if (mass_delta) *mass_delta = pm.mass_delta;
  });
}

// perturb_deformed's integrals through the mirror: fields [n][13][n_pts], metric6 = alpha, dalpha, ddalpha, beta, dbeta,
// ddbeta; out [6][n*n] column-major per matrix.
int storm_host_coupling_matrices(void* p, size_t n, const uint64_t* ells, const double* freqs, const double* fields,
                                 int64_t mm, const double* const* metric6, double* out) {
  PostProcessor* pp = static_cast<PostProcessor*>(p);
  return guarded([&] {
    const size_t np = pp->len();
    std::vector<Rotating1DPostprocessing> posts(n);
    std::vector<ModeToPerturb> modes(n);
    for (size_t k = 0; k < n; ++k) {
      Rotating1DPostprocessing& r = posts[k];
      std::vector<double>* dst[STORM_POST_FIELDS] = {&r.y1, &r.y2, &r.y3, &r.y4, &r.xi_r, &r.xi_h, &r.xi_tn, &r.xi_tp,
                                                     &r.psi, &r.dpsi, &r.rho, &r.p, &r.chi};
      for (int f = 0; f < STORM_POST_FIELDS; ++f) {
        const double* src = fields + (k * STORM_POST_FIELDS + f) * np;
        dst[f]->assign(src, src + np);
      }
      modes[k] = ModeToPerturb{ells[k], freqs[k], &posts[k]};
    }
    PerturbedMetric pm;
    std::vector<double>* md[6] = {&pm.alpha, &pm.dalpha, &pm.ddalpha, &pm.beta, &pm.dbeta, &pm.ddbeta};
    for (int k = 0; k < 6; ++k) md[k]->assign(metric6[k], metric6[k] + np);
    const CouplingMatrices c = pp->coupling_matrices(modes, mm, pm);
    const MatrixView* src[STORM_COUPLING_MATRICES] = {&c.d_squared, &c.d_linear, &c.d_zero, &c.r_linear, &c.r_zero, &c.l_zero};
    for (int k = 0; k < STORM_COUPLING_MATRICES; ++k) std::memcpy(out + size_t(k) * n * n, src[k]->data(), n * n * sizeof(double));
  });
}

double storm_host_polytrope0_exact(double gamma1, uint64_t ell, int64_t radial_order) {
  return Polytrope0(gamma1).exact(ell, radial_order);
}
void storm_host_linspace(double lo, double hi, size_t n, double* out) {
  std::vector<double> g = linspace(lo, hi, n); std::memcpy(out, g.data(), n * sizeof(double));
}
void storm_host_rev_linspace(double lo, double hi, size_t n, double* out) {
  std::vector<double> g = rev_linspace(lo, hi, n); std::memcpy(out, g.data(), n * sizeof(double));
}

}  // extern "C"
