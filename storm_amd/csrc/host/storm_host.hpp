// storm_host.hpp -- C++ host side above the C ABI, mirroring the reference's Rust interface for
// the scan path name for name, so a StORM user finds the same calls:
//
//   reference (Rust)                                   here (C++)
//   DiscreteModel::from_gsm           model/gsm.rs:68   DiscreteModel::from_gsm_arrays (HDF5 stays with the caller)
//   construct_polytrope               polytrope.rs:90   construct_polytrope
//   Polytrope0                        polytrope.rs:168  Polytrope0
//   LinearInterpolator                interpolate.rs:6  LinearInterpolator
//   ContinuousModel                   model.rs:17       ContinuousModel
//   Rotating1D::new                   adiabatic.rs:19   Rotating1D
//   DifferenceSchemes                 dynamic_interface.rs:24
//   Precision                         bracket.rs:19     Precision
//   ErasedSolver::{new,det,eigenvector,scan_and_optimize}   dynamic_interface.rs:60-160
//   Model::grid, linspace, rev_linspace, StormState::scan   main.rs:661-679,793-838,1232-1238
//   Rotating1DPostprocessing::new     postprocessing.rs:58   Rotating1DPostprocessing::make / post_process
//   StormState::post_process          main.rs:878-899   post_process
//   perturb_deformed (integrals)      perturbed.rs:56-263   PostProcessor::coupling_matrices -> ModeCoupling{d,r,l,...}
//
// The Rust host does exactly this work before it reaches the FFI seam: interpolate
// the model at the collocation abscissae (DiscretizedSystemImpl::new, discretized.rs:81-123) and
// hand the coefficient arrays to storm_solver_create.  Everything numeric below keeps the
// reference's operation order (built with -ffp-contract=off) so inputs to the GPU are bit-identical.
// Errors: where the reference panics, these throw std::runtime_error.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/storm_hip.h"

namespace storm_host {

struct DimensionedProperties {  // model.rs:93-102
  double radius, mass, grav;
  double freq_scale() const;    // model.rs:127-129
};

struct DimensionlessProperties {  // model.rs:70-91
  std::vector<double> r_coord, m_coord, rho, p, v, u, gamma1, a_star, c1, rot;
};

struct DiscreteModel {  // model.rs:6-13
  DimensionlessProperties dimensionless;
  std::optional<DimensionedProperties> scale;
  // The numeric part of from_gsm (gsm.rs:81-103) on datasets the caller has read from the GSM file.
  static DiscreteModel from_gsm_arrays(size_t n, const double* r, const double* m_r, const double* rho,
                                       const double* p, const double* gamma1, const double* n2,
                                       const double* omega_rot, double r_star, double m_star);
  size_t len() const { return dimensionless.r_coord.size(); }
};

DiscreteModel construct_polytrope(double n, double gamma1, double step);  // polytrope.rs:90-165

class ContinuousModel {  // model.rs:17-26
 public:
  virtual ~ContinuousModel() = default;
  virtual double inner() const = 0;
  virtual double outer() const = 0;
  virtual DiscreteModel eval(const std::vector<double>& grid) const = 0;
  virtual std::optional<DimensionedProperties> dimensions() const = 0;
};

class LinearInterpolator : public ContinuousModel {  // interpolate.rs
 public:
  explicit LinearInterpolator(const DiscreteModel& model) : model_(model) {}
  double inner() const override { return model_.dimensionless.r_coord.front(); }
  double outer() const override { return model_.dimensionless.r_coord.back(); }
  DiscreteModel eval(const std::vector<double>& grid) const override;
  std::optional<DimensionedProperties> dimensions() const override { return model_.scale; }
 private:
  const DiscreteModel& model_;
};

class Polytrope0 : public ContinuousModel {  // polytrope.rs:168-248
 public:
  explicit Polytrope0(double gamma1) : gamma1(gamma1) {}
  double gamma1;
  double exact(uint64_t ell, int64_t radial_order) const;
  double inner() const override { return 0.; }
  double outer() const override { return 1.; }
  DiscreteModel eval(const std::vector<double>& grid) const override;
  std::optional<DimensionedProperties> dimensions() const override { return std::nullopt; }
};

struct Rotating1D {  // adiabatic.rs:12-25
  uint64_t ell;
  int64_t m;
  static Rotating1D make(uint64_t ell, int64_t m) { return Rotating1D{ell, m}; }
};

enum class DifferenceSchemes : int {  // dynamic_interface.rs:24-57
  Colloc2 = 0, Colloc4, Colloc6, Colloc8, Magnus2, Magnus4, Magnus6, Magnus8
};

struct Precision {  // bracket.rs:19-40
  bool relative;
  uint64_t ulp;
  double rel;
  static Precision ULP(uint64_t u) {
    if (u == 0) throw std::runtime_error("Precision::ULP needs a non-zero value");
    return Precision{false, u, 0.};
  }
  static Precision Relative(double f) { return Precision{true, 0, f}; }
  uint64_t as_ulp() const { return relative ? storm_precision_relative_to_ulp(rel) : ulp; }
};

struct Point { double x, f; };                                             // bracket.rs:63-68
struct BracketResult { Point lower, upper; double root; uint64_t evals; };  // bracket.rs:6-15

std::vector<double> linspace(double lower, double upper, size_t n);      // main.rs:1232-1234
std::vector<double> rev_linspace(double lower, double upper, size_t n);  // main.rs:1236-1238
// Model::grid (main.rs:661-679): every interval of the model grid split `scale` times.
std::vector<double> grid_of(const DiscreteModel& model, size_t scale);
std::vector<double> grid_of(const ContinuousModel& model, size_t npoints);

class ErasedSolver {  // dynamic_interface.rs:60-160
 public:
  ErasedSolver(const ContinuousModel& model, Rotating1D system, DifferenceSchemes scheme,
               const std::vector<double>& solver_grid);
  ~ErasedSolver();
  ErasedSolver(const ErasedSolver&) = delete;
  ErasedSolver& operator=(const ErasedSolver&) = delete;

  double det(double freq) const;
  std::vector<double> det(const std::vector<double>& freqs) const;  // batched convenience
  std::vector<double> eigenvector(double freq) const;
  // Batched form of the per-mode loop of StormState::scan (main.rs:824-831); row-major [n][(n_int+1)*4].
  std::vector<double> eigenvectors(const std::vector<double>& freqs) const;
  // same, into caller-owned storage of n*(len()+1)*4 doubles (no intermediate copy)
  void eigenvectors_into(const double* freqs, size_t n, double* out) const;
  std::vector<BracketResult> scan_and_optimize(const std::vector<double>& freq_grid, Precision precision,
                                               bool* had_invalid_numbers = nullptr) const;
  size_t len() const { return n_int_; }
  storm_solver* handle() const { return h_; }

 private:
  storm_solver* h_ = nullptr;
  size_t n_int_ = 0;
};

// ErasedSolver over several GPUs of the node from this process (storm_multi_*, include/storm_hip.h): the same two calls,
// the scan's frequency grid cut into one contiguous slice per listed device, brackets and modes dealt out cyclically.
// Results are bit-identical to ErasedSolver's.  This is the form the Rust host would take for `scan`.
class MultiSolver {
 public:
  MultiSolver(const ContinuousModel& model, Rotating1D system, DifferenceSchemes scheme,
              const std::vector<double>& solver_grid, const std::vector<int>& devices);
  ~MultiSolver();
  MultiSolver(const MultiSolver&) = delete;
  MultiSolver& operator=(const MultiSolver&) = delete;
  std::vector<BracketResult> scan_and_optimize(const std::vector<double>& freq_grid, Precision precision,
                                               bool* had_invalid_numbers = nullptr) const;
  void eigenvectors_into(const double* freqs, size_t n, double* out) const;
  size_t len() const { return n_int_; }
  size_t points() const { return p_; }
  storm_multi* handle() const { return h_; }

 private:
  storm_multi* h_ = nullptr;
  size_t n_int_ = 0, p_ = 0;
};

struct Solution {  // main.rs:1225-1230
  BracketResult bracket;
  std::vector<double> eigenvector;
  uint64_t ell;
  int64_t m;
};

// StormState::scan (main.rs:793-838) with frequencies in dynamical units.
std::vector<Solution> scan(const ContinuousModel& model, const std::vector<double>& solver_grid, uint64_t ell,
                           int64_t m, double lower, double upper, size_t steps, bool inverse, double precision,
                           DifferenceSchemes scheme);

// Rotating1DPostprocessing, postprocessing.rs:14-54.
struct Rotating1DPostprocessing {
  std::vector<double> x, y1, y2, y3, y4, xi_r, xi_h, xi_tn, xi_tp, psi, dpsi, rho, p, chi;
  uint64_t cross_clockwise = 0, cross_counter_clockwise = 0;
  int64_t radial_order = 0;
  std::vector<double> nodes;
  // Rotating1DPostprocessing::new for one mode; `model` is the model evaluated on the solver grid.
  static Rotating1DPostprocessing make(double freq, const std::vector<double>& eigenvector, uint64_t ell, int64_t m,
                                       const DiscreteModel& model);
};

struct PerturbedMetric {  // model.rs:104-123
  std::vector<double> alpha, dalpha, ddalpha, beta, dbeta, ddbeta;
  double rot = 0., mass_delta = 0.;
};
// perturb_structure (perturbed.rs:324-460): the deformation of the equilibrium structure by uniform rotation `rot`
// (dynamical units).  A serial recurrence over the grid; stays on the host like in the reference (`deform`, main.rs:840-876).
PerturbedMetric perturb_structure(const DiscreteModel& model, double rot);

struct ModeToPerturb {  // perturbed.rs:26-33
  uint64_t ell;
  double freq;
  const Rotating1DPostprocessing* post_processing;
};
// The six matrices perturb_deformed integrates (perturbed.rs:110-120), each n x n column-major like DMatrix
// ((left, right) at left + right*n).  ModeCoupling's d, r, l are d_squared, r_linear, l_zero (perturbed.rs:313-320);
// the generalised eigenproblem assembled from all six (perturbed.rs:265-311, linalg/qz.rs) is the caller's.
// One n x n matrix of a CouplingMatrices block (column-major like DMatrix: element (left, right) at left + right * n).
struct MatrixView {
  const double* p = nullptr;
  size_t len = 0;
  const double* begin() const { return p; }
  const double* end() const { return p + len; }
  const double* data() const { return p; }
  size_t size() const { return len; }
  double operator[](size_t i) const { return p[i]; }
};
// The six matrices live in ONE block, as the device returns them (no copy per matrix: a rotation sweep moves ~3 MB per
// rotation rate through here).
struct CouplingMatrices {
  size_t n = 0;
  MatrixView d_squared, d_linear, d_zero, r_linear, r_zero, l_zero;
  std::unique_ptr<double[]> storage;  // [6][n * n]
  double at(const MatrixView& mat, size_t left, size_t right) const { return mat[left + right * n]; }
  // Allocates the block (uninitialised) and points the six views into it; returns the block for the library to fill.
  double* allocate(size_t n_modes) {
    n = n_modes;
    storage.reset(new double[6 * n * n + 1]);
    MatrixView* v[6] = {&d_squared, &d_linear, &d_zero, &r_linear, &r_zero, &l_zero};
    for (int q = 0; q < 6; ++q) *v[q] = MatrixView{storage.get() + size_t(q) * n * n, n * n};
    return storage.get();
  }
};

class SolverSweep;

// The model arrays of the post-processing resident on the device (storm_postproc).
class PostProcessor {
 public:
  explicit PostProcessor(const DiscreteModel& model_on_grid);
  ~PostProcessor();
  PostProcessor(const PostProcessor&) = delete;
  PostProcessor& operator=(const PostProcessor&) = delete;
  // StormState::post_process (main.rs:878-899) over solutions of one (ell, m).
  std::vector<Rotating1DPostprocessing> post_process(const std::vector<Solution>& solutions) const;
  // Eigenvectors and post-processing in one device pass (storm_modes_batch); mode ids only unless want_fields.
  std::vector<Rotating1DPostprocessing> modes(const ErasedSolver& solver, const std::vector<double>& roots,
                                              bool want_fields) const;
  // The integrals of perturb_deformed(model, modes, m, metric) (perturbed.rs:56-263).
  CouplingMatrices coupling_matrices(const std::vector<ModeToPerturb>& modes, int64_t m, const PerturbedMetric& metric) const;
  size_t len() const { return x_.size(); }
  storm_postproc* handle() const { return h_; }

 private:
  friend class SolverSweep;
  std::vector<Rotating1DPostprocessing> unpack(size_t n, const std::vector<double>& fields,
                                               const std::vector<storm_mode_id>& ids, const std::vector<double>& nodes,
                                               size_t node_cap) const;
  storm_postproc* h_ = nullptr;
  std::vector<double> x_;
};

// Many systems evaluated in lock-step (storm_sweep): what a batch input that sweeps rotation rates and (ell, m) does
// with one ErasedSolver per `scan` line (main.rs:815-831, scripts/generate-batch-input-deformed.py:19-35), as one
// object.  All systems share the difference scheme and the solver grid; the model, ell and m may differ.  Results are
// bit-identical to one ErasedSolver per system.  Per-system inputs and outputs are vectors indexed by system.
struct SweepSystem {
  const ContinuousModel* model;  // borrowed for the constructor only; systems may share one
  Rotating1D system;
};
class SolverSweep {
 public:
  SolverSweep(const std::vector<SweepSystem>& systems, DifferenceSchemes scheme, const std::vector<double>& solver_grid);
  ~SolverSweep();
  SolverSweep(const SolverSweep&) = delete;
  SolverSweep& operator=(const SolverSweep&) = delete;

  std::vector<std::vector<double>> det(const std::vector<std::vector<double>>& freqs) const;
  std::vector<std::vector<BracketResult>> scan_and_optimize(const std::vector<std::vector<double>>& freq_grids,
                                                            Precision precision,
                                                            std::vector<bool>* had_invalid_numbers = nullptr) const;
  // per system: row-major [modes][(n_int+1)*4]
  std::vector<std::vector<double>> eigenvectors(const std::vector<std::vector<double>>& roots) const;
  // eigenvectors + post-processing on the device; posts[u] = the model of system u on the solver grid
  std::vector<std::vector<Rotating1DPostprocessing>> modes(const std::vector<const PostProcessor*>& posts,
                                                           const std::vector<std::vector<double>>& roots,
                                                           bool want_fields) const;
  // post-process + perturb-deformed with the fields staying on the device: modes_resident returns ids and nodes only
  // (the per-point arrays of the results are empty) and keeps the fields of all modes of the call in HBM;
  // coupling_matrices then integrates a subset of them: mode (u, k) = mode k of system u of that call.
  std::vector<std::vector<Rotating1DPostprocessing>> modes_resident(const std::vector<const PostProcessor*>& posts,
                                                                    const std::vector<std::vector<double>>& roots) const;
  struct ModeRef { size_t system, mode; };
  CouplingMatrices coupling_matrices(const PostProcessor& model_on_grid, const std::vector<ModeRef>& modes, int64_t m,
                                     const PerturbedMetric& metric) const;
  // many blocks (rotation rates) in one launch
  struct CouplingProblem {
    const PostProcessor* model_on_grid;
    std::vector<ModeRef> modes;
    int64_t m;
    const PerturbedMetric* metric;
  };
  std::vector<CouplingMatrices> coupling_matrices(const std::vector<CouplingProblem>& problems) const;
  size_t size() const { return ell_.size(); }
  size_t len() const { return n_int_; }
  storm_sweep* handle() const { return h_; }

 private:
  std::vector<std::vector<Rotating1DPostprocessing>> run_modes(const std::vector<const PostProcessor*>& posts,
                                                               const std::vector<std::vector<double>>& roots,
                                                               bool want_fields, bool resident) const;
  storm_sweep* h_ = nullptr;
  size_t n_int_ = 0;
  std::vector<uint64_t> ell_;
  std::vector<int64_t> m_;
  mutable std::vector<size_t> resident_off_;               // offsets of the resident call
  mutable std::vector<double> resident_freq_;
};

}  // namespace storm_host
