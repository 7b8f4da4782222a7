// oracle_driver.cpp -- the CPU oracle (oracle/storm_ref.hpp through its C API, oracle/storm_ref_capi.cpp) under the
// sanitizers: model construction, discretisation, a threaded scan with refinement, threaded eigenvectors, post-processing,
// the deformation and the coupling integrals, on file-free models (Polytrope0, an n = 3 polytrope), with the reference's own
// test_polytrope known answer (src/dynamic_interface.rs:505-521) as the functional check.  Built twice by
// tests/sanitize/Makefile (-fsanitize=address,undefined / -fsanitize=thread) together with storm_ref_capi.cpp.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct sref_bracket { double lo_x, lo_f, hi_x, hi_f, root; uint64_t evals; };
extern "C" {
const char* sref_last_error();
void* sref_model_polytrope(double n, double gamma1, double step);
void* sref_model_poly0(double gamma1);
void sref_model_free(void* m);
size_t sref_model_len(void* m);
size_t sref_model_grid(void* m, size_t scale, double* out, size_t cap);
void sref_model_set_rot_const(void* m, double value);
void* sref_model_eval(void* m, const double* grid, size_t ngrid);
void* sref_solver_new(void* m, uint64_t ell, int64_t mm, int scheme, const double* grid, size_t ngrid);
void sref_solver_free(void* s);
size_t sref_solver_len(void* s);
int sref_det_batch(void* s, const double* f, size_t n, double* out, int threads);
int sref_scan_and_optimize(void* s, const double* f, size_t n, uint64_t ulp, sref_bracket* out, size_t cap, size_t* n_out,
                           int* had_invalid, int threads, double* dets_out, uint64_t* index_out);
int sref_eigenvector_batch(void* s, const double* f, size_t n, double* det_out, double* vec_out, int threads);
int sref_postprocess_batch(void* m, uint64_t ell, int64_t mm, const double* freq, size_t n, const double* vec, double* fields_out,
                           int64_t* ids_out, double* nodes_out, size_t node_cap, uint64_t* n_nodes_out, int threads);
int sref_coupling_matrices(void* m, size_t M, const uint64_t* ells, const double* freqs, const double* fields, int64_t mm,
                           const double* const* metric6, double* out, uint64_t* flops_out, int threads);
int sref_perturb_structure(void* m, double rot, double* const* out6, double* mass_delta);
double sref_polytrope0_exact(double gamma1, uint64_t ell, int64_t radial_order);
void sref_linspace(double lo, double hi, size_t n, double* out);
uint64_t sref_precision_relative_to_ulp(double rel);
}

#define CHECK(cond)                                                                                                     \
  do {                                                                                                                  \
    if (!(cond)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s (%s)\n", __FILE__, __LINE__, #cond, sref_last_error()); std::abort(); } \
  } while (0)

int main() {
  const int threads = 8;
  // --- test_polytrope: Polytrope0{5/3}, Colloc6, 10^4-point uniform grid, scan [3, 4] -> one root = the analytic frequency
  {
    void* m = sref_model_poly0(5.0 / 3.0);
    std::vector<double> grid(10000), f(5);
    sref_linspace(0.0, 1.0, grid.size(), grid.data());
    sref_linspace(3.0, 4.0, f.size(), f.data());
    void* s = sref_solver_new(m, 0, 0, 2, grid.data(), grid.size());
    CHECK(s && sref_solver_len(s) == 9999);
    std::vector<sref_bracket> br(f.size());
    size_t found = 0;
    int invalid = 0;
    CHECK(sref_scan_and_optimize(s, f.data(), f.size(), 1, br.data(), br.size(), &found, &invalid, threads, nullptr, nullptr) == 0);
    CHECK(found == 1 && !invalid);
    const double exact = sref_polytrope0_exact(5.0 / 3.0, 0, 1);
    CHECK(std::fabs(br[0].root / exact - 1.0) < 1e-13);
    sref_solver_free(s);
    sref_model_free(m);
  }
  // --- an n = 3 polytrope (~700 points): every scheme, threaded scan + eigenvectors + post-processing + coupling integrals
  void* m = sref_model_polytrope(3.0, 5.0 / 3.0, 0.01);
  CHECK(m);
  sref_model_set_rot_const(m, 0.05);
  const size_t np = sref_model_grid(m, 1, nullptr, 0);
  std::vector<double> grid(np);
  sref_model_grid(m, 1, grid.data(), np);
  void* on_grid = sref_model_eval(m, grid.data(), np);
  CHECK(on_grid && sref_model_len(on_grid) == np);
  std::vector<double> f(160);
  sref_linspace(1.0, 12.0, f.size(), f.data());
  std::vector<double> all_fields, all_freqs;
  std::vector<uint64_t> all_ells;
  for (int scheme = 0; scheme < 8; ++scheme) {
    const uint64_t ell = scheme % 3;
    const int64_t mm = ell ? 1 : 0;
    void* s = sref_solver_new(m, ell, mm, scheme, grid.data(), np);
    CHECK(s);
    std::vector<sref_bracket> br(f.size());
    std::vector<double> dets(f.size());
    std::vector<uint64_t> idx(f.size());
    size_t found = 0;
    int invalid = 0;
    CHECK(sref_scan_and_optimize(s, f.data(), f.size(), sref_precision_relative_to_ulp(1e-8), br.data(), br.size(), &found, &invalid,
                                 threads, dets.data(), idx.data()) == 0);
    CHECK(found >= 3 && found < f.size());
    std::vector<double> roots(found), vec(found * np * 4), det(found);
    for (size_t k = 0; k < found; ++k) roots[k] = br[k].root;
    CHECK(sref_eigenvector_batch(s, roots.data(), found, det.data(), vec.data(), threads) == 0);
    std::vector<double> fields(found * 13 * np), nodes(found * 64);
    std::vector<int64_t> ids(found * 3);
    std::vector<uint64_t> n_nodes(found);
    CHECK(sref_postprocess_batch(on_grid, ell, mm, roots.data(), found, vec.data(), fields.data(), ids.data(), nodes.data(), 64,
                                 n_nodes.data(), threads) == 0);
    if (scheme == 1 || scheme == 2) {   // (ell = 1, 2; m = 1): keep a few modes for the coupling integrals
      for (size_t k = 0; k < std::min<size_t>(found, 5); ++k) {
        all_fields.insert(all_fields.end(), fields.begin() + k * 13 * np, fields.begin() + (k + 1) * 13 * np);
        all_freqs.push_back(roots[k]);
        all_ells.push_back(ell);
      }
    }
    sref_solver_free(s);
  }
  std::vector<std::vector<double>> metric(6, std::vector<double>(np));
  double* out6[6];
  const double* in6[6];
  for (int k = 0; k < 6; ++k) { out6[k] = metric[k].data(); in6[k] = metric[k].data(); }
  double mass_delta = 0.0;
  CHECK(sref_perturb_structure(on_grid, 0.05, out6, &mass_delta) == 0);
  const size_t M = all_freqs.size();
  CHECK(M >= 6);
  std::vector<double> cm(6 * M * M);
  uint64_t flops = 0;
  CHECK(sref_coupling_matrices(on_grid, M, all_ells.data(), all_freqs.data(), all_fields.data(), 1, in6, cm.data(), &flops, threads) == 0);
  CHECK(flops > 100 && flops < 400);
  for (double v : cm) CHECK(std::isfinite(v));
  sref_model_free(on_grid);
  sref_model_free(m);
  std::printf("oracle driver: all clean (%zu coupled modes, %llu flops per term)\n", M, (unsigned long long)flops);
  return 0;
}
