// host_driver.cpp -- the host mirror (storm_amd/csrc/host/storm_host.cpp) under AddressSanitizer / UBSan, on the CPU: the
// model layer it restates (construct_polytrope, LinearInterpolator::eval incl. exact grid hits and out-of-range points,
// Polytrope0, grid_of, linspace / rev_linspace, perturb_structure, Precision) and the ERROR path of every handle type when
// no HIP device is present -- the C ABI must hand back a status, the mirror must throw, nothing may leak or unwind across
// the boundary.  Links the real libstorm_hip.so (un-instrumented); never runs on a GPU box.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../storm_amd/csrc/host/storm_host.hpp"

using namespace storm_host;

#define CHECK(cond)                                                                    \
  do {                                                                                 \
    if (!(cond)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); std::abort(); } \
  } while (0)

template <class F>
static std::string thrown_by(F&& f) {
  try { f(); } catch (const std::exception& e) { return e.what(); }
  return std::string();
}

int main() {
  // ---- models
  DiscreteModel poly = construct_polytrope(3.0, 5.0 / 3.0, 0.01);
  const size_t n = poly.len();
  CHECK(n > 600 && n < 800);
  const auto& d = poly.dimensionless;
  CHECK(d.r_coord.front() == 0.0 && d.v.size() == n && d.rot.size() == n);
  for (size_t i = 1; i < n; ++i) CHECK(d.r_coord[i] > d.r_coord[i - 1]);
  LinearInterpolator lin(poly);
  std::vector<double> g1 = grid_of(poly, 1), g3 = grid_of(poly, 3);
  CHECK(g1.size() == n && g3.size() == 3 * (n - 1) + 1);
  DiscreteModel on3 = lin.eval(g3);
  CHECK(on3.len() == g3.size());
  for (size_t i = 0; i < n; ++i) CHECK(on3.dimensionless.u[3 * i] == d.u[i]);   // exact grid hits copy the node
  for (double v : on3.dimensionless.c1) CHECK(std::isfinite(v));
  Polytrope0 p0(5.0 / 3.0);
  std::vector<double> gp = grid_of(p0, 1000);
  CHECK(gp.size() == 1000 && gp.front() == 0.0 && gp.back() == 1.0);
  DiscreteModel on0 = p0.eval(gp);
  CHECK(on0.len() == 1000 && std::isfinite(p0.exact(2, 3)));
  std::vector<double> f = linspace(1.0, 25.0, 7), r = rev_linspace(0.5, 1.0, 7);
  CHECK(f.front() == 1.0 && f.back() == 25.0 && r.size() == 7 && std::fabs(r.front() - 0.5) < 1e-15);
  CHECK(linspace(1.0, 2.0, 0).empty());
  CHECK(Precision::Relative(1e-8).as_ulp() == 45035996 && Precision::ULP(3).as_ulp() == 3);
  CHECK(!thrown_by([] { Precision::ULP(0); }).empty());
  // ---- perturb_structure (a serial recurrence over the grid, on the host like in the reference)
  PerturbedMetric pm = perturb_structure(poly, 0.1);
  CHECK(pm.alpha.size() == n && pm.ddbeta.size() == n && std::isfinite(pm.mass_delta));
  for (size_t i = 0; i < n; ++i) CHECK(std::isfinite(pm.alpha[i]) && std::isfinite(pm.beta[i]));
  // ---- handles without a device: clean failures
  int n_dev = 0;
  const bool have_gpu = (std::getenv("STORM_SANITIZE_EXPECT_GPU") != nullptr);
  if (!have_gpu) {
    const std::string e1 = thrown_by([&] { ErasedSolver s(lin, Rotating1D::make(2, 0), DifferenceSchemes::Colloc4, g1); });
    CHECK(!e1.empty());
    const std::string e2 = thrown_by([&] { MultiSolver s(lin, Rotating1D::make(1, 1), DifferenceSchemes::Magnus2, g1, {0, 1}); });
    CHECK(!e2.empty());
    const std::string e3 = thrown_by([&] { PostProcessor pp(lin.eval(g1)); });
    CHECK(!e3.empty());
    const std::string e4 = thrown_by([&] {
      std::vector<SweepSystem> sys;
      sys.push_back(SweepSystem{&lin, Rotating1D::make(1, 0)});
      SolverSweep sw(sys, DifferenceSchemes::Colloc2, g1);
    });
    CHECK(!e4.empty());
    // an empty solver grid is rejected before any device is touched
    CHECK(!thrown_by([&] { ErasedSolver s(lin, Rotating1D::make(2, 0), DifferenceSchemes::Colloc4, {}); }).empty());
    std::printf("host driver: no HIP device, four handle types failed cleanly (\"%s\")\n", e1.c_str());
  }
  (void)n_dev;
  std::printf("host driver: all clean\n");
  return 0;
}
