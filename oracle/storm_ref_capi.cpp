// C API over the CPU ORACLE (storm_ref.hpp) for ctypes.  Test infrastructure only:
// see the header of storm_ref.hpp.  Built by oracle/Makefile into
// oracle/_build/libstorm_ref.so with  g++ -O2 -ffp-contract=off.
#include "storm_ref.hpp"

#include <string>

using namespace storm_ref;

namespace {
thread_local std::string g_err;

struct ModelHandle {
  DiscreteModel discrete;
  bool is_poly0 = false;
  double gamma1 = 0.;
  size_t poly0_points = 0;
  ContinuousModel continuous() const {
    ContinuousModel c;
    c.is_poly0 = is_poly0; c.gamma1 = gamma1; c.discrete = &discrete;
    return c;
  }
};
struct SolverHandle { ErasedSolver solver; };

template <class F>
int guarded(F&& f) {
  try { f(); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
  catch (...) { g_err = "unknown error"; return -1; }
}
}  // namespace

extern "C" {

const char* sref_last_error() { return g_err.c_str(); }

void* sref_model_from_gsm(size_t n, const double* r, const double* m_r, const double* rho, const double* p,
                          const double* gamma1, const double* n2, const double* omega_rot, double radius,
                          double mass) {
  ModelHandle* h = new ModelHandle();
  if (guarded([&] { h->discrete = model_from_gsm(n, r, m_r, rho, p, gamma1, n2, omega_rot, radius, mass); })) {
    delete h; return nullptr;
  }
  return h;
}
void* sref_model_polytrope(double n, double gamma1, double step) {
  ModelHandle* h = new ModelHandle();
  if (guarded([&] { h->discrete = construct_polytrope(n, gamma1, step); })) { delete h; return nullptr; }
  return h;
}
void* sref_model_poly0(double gamma1) {
  ModelHandle* h = new ModelHandle();
  h->is_poly0 = true; h->gamma1 = gamma1;
  return h;
}
void sref_model_free(void* m) { delete static_cast<ModelHandle*>(m); }
size_t sref_model_len(void* m) { return static_cast<ModelHandle*>(m)->discrete.size(); }
// field: 0 r_coord, 1 v, 2 u, 3 gamma1, 4 a_star, 5 c1, 6 rot, 7 m_coord, 8 rho, 9 p
int sref_model_get(void* m, int field, double* out) {
  const DiscreteModel& d = static_cast<ModelHandle*>(m)->discrete;
  const std::vector<double>* f[] = {&d.r_coord, &d.v, &d.u, &d.gamma1, &d.a_star, &d.c1, &d.rot, &d.m_coord, &d.rho, &d.p};
  if (field < 0 || field > 9) return -1;
  std::copy(f[field]->begin(), f[field]->end(), out);
  return 0;
}
// set-rotation-constant in dynamical units (src/main.rs:785-788 with Dynamical units).
void sref_model_set_rot_const(void* m, double value) {
  DiscreteModel& d = static_cast<ModelHandle*>(m)->discrete;
  std::fill(d.rot.begin(), d.rot.end(), value);
}
// Overwrite one model field (same indices as sref_model_get), e.g. a non-uniform rotation profile
// (what set-rotation-overlay does, src/model/gsm.rs:116-127, after unit conversion).
int sref_model_set(void* m, int field, const double* in) {
  DiscreteModel& d = static_cast<ModelHandle*>(m)->discrete;
  std::vector<double>* f[] = {&d.r_coord, &d.v, &d.u, &d.gamma1, &d.a_star, &d.c1, &d.rot, &d.m_coord, &d.rho, &d.p};
  if (field < 0 || field > 9) return -1;
  std::copy(in, in + f[field]->size(), f[field]->begin());
  return 0;
}
double sref_model_freq_scale(void* m) {  // DimensionedProperties::freq_scale, model.rs:127-129
  const DiscreteModel& d = static_cast<ModelHandle*>(m)->discrete;
  if (!d.has_scale) return 0.;
  return std::sqrt(d.grav * d.mass / cube(d.radius));
}
// Model::grid(), src/main.rs:661-679.  `scale` = resample (discrete) or npoints (poly0).
// Returns the number of points; writes up to cap.
size_t sref_model_grid(void* m, size_t scale, double* out, size_t cap) {
  ModelHandle* h = static_cast<ModelHandle*>(m);
  std::vector<double> g = h->is_poly0 ? linspace(0., 1., scale) : model_grid_discrete(h->discrete, scale);
  if (out) std::copy(g.begin(), g.begin() + std::min(cap, g.size()), out);
  return g.size();
}

void* sref_solver_new(void* m, uint64_t ell, int64_t mm, int scheme, const double* grid, size_t ngrid) {
  ModelHandle* h = static_cast<ModelHandle*>(m);
  SolverHandle* s = new SolverHandle();
  if (guarded([&] {
        if (scheme < 0 || scheme > 7) throw std::runtime_error("bad scheme");
        s->solver.sys = discretize(h->continuous(), scheme, ell, mm, std::vector<double>(grid, grid + ngrid));
      })) { delete s; return nullptr; }
  return s;
}
void sref_solver_free(void* s) { delete static_cast<SolverHandle*>(s); }
size_t sref_solver_len(void* s) { return static_cast<SolverHandle*>(s)->solver.sys.len(); }
int sref_solver_points(void* s) { return static_cast<SolverHandle*>(s)->solver.sys.P; }

// Dump the discretised system: per collocation point [len*P] u, v_gamma, a_star, c1, rot, delta/x,
// and the boundary model points {inner c1, inner rot, outer u}.  This is what the host hands the
// HIP library (include/storm_hip.h: storm_system_desc).
int sref_solver_system(void* s, double* u, double* v_gamma, double* a_star, double* c1, double* rot,
                       double* delta_over_x, double* bnd3) {
  const Discretized& d = static_cast<SolverHandle*>(s)->solver.sys;
  for (size_t i = 0; i < d.len(); ++i)
    for (int j = 0; j < d.P; ++j) {
      size_t k = i * d.P + j;
      u[k] = d.pts[k].u; v_gamma[k] = d.pts[k].v_gamma; a_star[k] = d.pts[k].a_star;
      c1[k] = d.pts[k].c1; rot[k] = d.pts[k].rot; delta_over_x[k] = d.delta[i] / d.xs[k];
    }
  bnd3[0] = d.inner.c1; bnd3[1] = d.inner.rot; bnd3[2] = d.outer.u;
  return 0;
}

int sref_det_batch(void* s, const double* f, size_t n, double* out, int threads) {
  const ErasedSolver& sv = static_cast<SolverHandle*>(s)->solver;
  return guarded([&] { parallel_for(n, threads, [&](size_t i) { out[i] = sv.det(f[i]); }); });
}
// Step (transfer) matrix of one interval at one frequency, 16 doubles column-major.
int sref_step_matrix(void* s, size_t step, double f, double* out16) {
  const ErasedSolver& sv = static_cast<SolverHandle*>(s)->solver;
  return guarded([&] {
    Mat<double, 4> m = fill_explicit<double>(sv.sys, step, f);
    std::copy(m.a, m.a + 16, out16);
  });
}

struct sref_bracket { double lo_x, lo_f, hi_x, hi_f, root; uint64_t evals; };

// scan_and_optimize.  dets_out [n] and index_out [cap] may be NULL.
int sref_scan_and_optimize(void* s, const double* f, size_t n, uint64_t ulp, sref_bracket* out, size_t cap,
                           size_t* n_out, int* had_invalid, int threads, double* dets_out, uint64_t* index_out) {
  const ErasedSolver& sv = static_cast<SolverHandle*>(s)->solver;
  return guarded([&] {
    bool inv = false;
    std::vector<double> dets;
    std::vector<uint64_t> idx;
    std::vector<BracketResult> r = sv.scan_and_optimize(std::vector<double>(f, f + n), ulp, &inv, threads, &dets, &idx);
    if (dets_out) std::copy(dets.begin(), dets.end(), dets_out);
    *n_out = r.size();
    if (had_invalid) *had_invalid = inv ? 1 : 0;
    for (size_t i = 0; i < r.size() && i < cap; ++i) {
      out[i] = sref_bracket{r[i].lower.x, r[i].lower.f, r[i].upper.x, r[i].upper.f, r[i].root, r[i].evals};
      if (index_out) index_out[i] = idx[i];
    }
  });
}
uint64_t sref_precision_relative_to_ulp(double rel) { return precision_relative_to_ulp(rel); }

// Bracket refinement of an arbitrary bracket with this solver's det (InverseQuadratic::optimize).
int sref_optimize(void* s, double lo_x, double lo_f, double hi_x, double hi_f, uint64_t ulp, sref_bracket* out) {
  const ErasedSolver& sv = static_cast<SolverHandle*>(s)->solver;
  return guarded([&] {
    BracketResult r = optimize_bracket(Point{lo_x, lo_f}, Point{hi_x, hi_f}, [&](double x) { return sv.det(x); }, ulp);
    *out = sref_bracket{r.lower.x, r.lower.f, r.upper.x, r.upper.f, r.root, r.evals};
  });
}

// eigenvector(f): det and the (len+1)*4 eigenvector.
int sref_eigenvector_batch(void* s, const double* f, size_t n, double* det_out, double* vec_out, int threads) {
  const ErasedSolver& sv = static_cast<SolverHandle*>(s)->solver;
  const size_t stride = (sv.sys.len() + 1) * 4;
  return guarded([&] {
    parallel_for(n, threads, [&](size_t i) {
      double dv;
      std::vector<double> e = sv.eigenvector(f[i], &dv);
      if (det_out) det_out[i] = dv;
      std::copy(e.begin(), e.end(), vec_out + i * stride);
    });
  });
}

// The model evaluated on a grid (what `input.as_continuous().eval(&input.grid())` hands the post-processing,
// src/main.rs:892): returns a new model handle holding the 10 arrays on that grid.
void* sref_model_eval(void* m, const double* grid, size_t ngrid) {
  ModelHandle* h = static_cast<ModelHandle*>(m);
  ModelHandle* o = new ModelHandle();
  if (guarded([&] { o->discrete = h->continuous().eval(std::vector<double>(grid, grid + ngrid)); })) { delete o; return nullptr; }
  return o;
}

// Rotating1DPostprocessing::new for n modes on one (already evaluated) model.  fields_out [n][13][n_pts]
// (order: PostField), ids_out [n][3] = {clockwise, counter-clockwise, radial order}, nodes_out [n][node_cap]
// (may be NULL), n_nodes_out [n].
int sref_postprocess_batch(void* m, uint64_t ell, int64_t mm, const double* freq, size_t n, const double* vec,
                           double* fields_out, int64_t* ids_out, double* nodes_out, size_t node_cap,
                           uint64_t* n_nodes_out, int threads) {
  const DiscreteModel& d = static_cast<ModelHandle*>(m)->discrete;
  const size_t np = d.size();
  return guarded([&] {
    parallel_for(n, threads, [&](size_t k) {
      Postprocessing pp = postprocess(freq[k], vec + k * np * 4, np * 4, ell, mm, d);
      if (fields_out)
        for (int f = 0; f < PF_COUNT; ++f) std::copy(pp.f[f].begin(), pp.f[f].end(), fields_out + (k * PF_COUNT + f) * np);
      ids_out[k * 3] = int64_t(pp.cross_clockwise); ids_out[k * 3 + 1] = int64_t(pp.cross_counter_clockwise);
      ids_out[k * 3 + 2] = pp.radial_order;
      if (n_nodes_out) n_nodes_out[k] = pp.nodes.size();
      if (nodes_out) std::copy(pp.nodes.begin(), pp.nodes.begin() + std::min(node_cap, pp.nodes.size()), nodes_out + k * node_cap);
    });
  });
}
int sref_gaunt(uint64_t ell, int64_t mm, double* out4) {  // {q_hd_n, q_hd_p, q_h_n, q_h_p}, postprocessing.rs:88-91
  const uint64_t n1 = ell == 0 ? 0 : ell - 1;
  out4[0] = q_kl1_hd(ell, n1, mm); out4[1] = q_kl1_hd(ell, ell + 1, mm);
  out4[2] = q_kl1_h(ell, n1, mm); out4[3] = q_kl1_h(ell, ell + 1, mm);
  return 0;
}
double sref_powi(double a, int b) { return powi(a, b); }

// perturb_deformed's six coupling matrices (src/perturbed.rs:110-263).  fields is [M][13][n_pts] in PostField order
// (the layout sref_postprocess_batch / storm_postprocess_batch write), metric6 = {alpha, dalpha, ddalpha, beta,
// dbeta, ddbeta} each [n_pts]; out is [6][M*M] column-major per matrix; flops_out (may be NULL) receives the exact
// add+mul+div count of ONE (left, right, point) term, measured with the op-counting scalar.
int sref_coupling_matrices(void* m, size_t M, const uint64_t* ells, const double* freqs, const double* fields, int64_t mm,
                           const double* const* metric6, double* out, uint64_t* flops_out, int threads) {
  const DiscreteModel& d = static_cast<ModelHandle*>(m)->discrete;
  const size_t np = d.size();
  return guarded([&] {
    PerturbedMetric pm;
    std::vector<double>* dst[6] = {&pm.alpha, &pm.dalpha, &pm.ddalpha, &pm.beta, &pm.dbeta, &pm.ddbeta};
    for (int k = 0; k < 6; ++k) dst[k]->assign(metric6[k], metric6[k] + np);
    std::vector<ModeToPerturb> modes(M);
    for (size_t k = 0; k < M; ++k) {
      const double* f = fields + k * PF_COUNT * np;
      modes[k] = ModeToPerturb{ells[k], freqs[k], f + PF_XI_R * np, f + PF_XI_H * np, f + PF_XI_TN * np, f + PF_XI_TP * np,
                               f + PF_PSI * np, f + PF_RHO * np, f + PF_P * np};
    }
    double* o[CM_COUNT];
    for (int c = 0; c < CM_COUNT; ++c) o[c] = out + size_t(c) * M * M;
    coupling_matrices(d, modes, mm, pm, o, threads);  // rows of the (left, right) loop are independent
    if (flops_out && M) {
      Counted::reset();
      Counted v[CM_COUNT];
      coupling_pair<Counted>(d, CouplingWeights(d, pm), pm, modes[0], modes[0], mm, v);
      *flops_out = Counted::total() / (np - 1);
    }
  });
}

// perturb_structure (src/perturbed.rs:324-460): out6 = {alpha, dalpha, ddalpha, beta, dbeta, ddbeta}, each [n_pts].
int sref_perturb_structure(void* m, double rot, double* const* out6, double* mass_delta) {
  const DiscreteModel& d = static_cast<ModelHandle*>(m)->discrete;
  return guarded([&] {
    PerturbedMetric pm = perturb_structure(d, rot, mass_delta);
    const std::vector<double>* src[6] = {&pm.alpha, &pm.dalpha, &pm.ddalpha, &pm.beta, &pm.dbeta, &pm.ddbeta};
    for (int k = 0; k < 6; ++k) std::copy(src[k]->begin(), src[k]->end(), out6[k]);
  });
}

// Matrix exponential known-answer entry (N = 3 or 4), column-major in place.
int sref_exp(int n, double* a) {
  return guarded([&] {
    if (n == 3) { Mat<double, 3> m; std::copy(a, a + 9, m.a); mat_exp(m); std::copy(m.a, m.a + 9, a); }
    else if (n == 4) { Mat<double, 4> m; std::copy(a, a + 16, m.a); mat_exp(m); std::copy(m.a, m.a + 16, a); }
    else throw std::runtime_error("sref_exp: n must be 3 or 4");
  });
}
int sref_inv4(const double* a, double* out) {
  Mat<double, 4> m, o; std::copy(a, a + 16, m.a);
  if (!inv4(m, o)) return 1;
  std::copy(o.a, o.a + 16, out); return 0;
}
int sref_matmul4(const double* a, const double* b, double* out) {
  Mat<double, 4> x, y; std::copy(a, a + 16, x.a); std::copy(b, b + 16, y.a);
  Mat<double, 4> c = mul(x, y); std::copy(c.a, c.a + 16, out); return 0;
}

// Exact algorithmic flop count (add+mul+div) of one det(f) evaluation, reference form.
// out3 = {adds, muls, divs}.
int sref_count_flops(void* s, double f, uint64_t* out3) {
  const ErasedSolver& sv = static_cast<SolverHandle*>(s)->solver;
  return guarded([&] {
    Counted::reset();
    Counted r = determinant_explicit<Counted>(sv.sys, Counted(f));
    (void)r;
    out3[0] = Counted::n_add; out3[1] = Counted::n_mul; out3[2] = Counted::n_div;
  });
}

double sref_polytrope0_exact(double gamma1, uint64_t ell, int64_t radial_order) {
  return polytrope0_exact(gamma1, ell, radial_order);
}
void sref_linspace(double lo, double hi, size_t n, double* out) {
  std::vector<double> g = linspace(lo, hi, n); std::copy(g.begin(), g.end(), out);
}
void sref_rev_linspace(double lo, double hi, size_t n, double* out) {
  std::vector<double> g = rev_linspace(lo, hi, n); std::copy(g.begin(), g.end(), out);
}

}  // extern "C"
