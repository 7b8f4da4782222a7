// The rotation-sweep workload (reference scripts/generate-batch-input-deformed.py:19-35) driven from C++ threads through
// the host mirror: per job  model copy with its own rotation rate -> ErasedSolver::new -> scan_and_optimize ->
// PostProcessor -> modes (eigenvectors + mode ids on the device).  No Python, no GIL.
// build: g++ -O2 -std=c++17 -pthread -o tools/many_scans.bin tools/many_scans.cpp -Lstorm_amd/lib -lstorm_host -lstorm_hip
//        -Wl,-rpath,'$ORIGIN/../storm_amd/lib'
// run:   tools/many_scans.bin model.bin [scans=64] [steps=300] [scheme=1] [threads...]
//        a negative thread count -B runs the jobs as SolverSweeps of B systems each (storm_sweep)
//        model.bin = u64 n, f64 R_star, f64 M_star, then r, M_r, rho, P, Gamma_1, N2, Omega_rot (n doubles each)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>

#include "../storm_amd/csrc/host/storm_host.hpp"

using namespace storm_host;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

extern "C" int hipSetDeviceFlags(unsigned flags);  // libamdhip64 (already a dependency of libstorm_hip)

int main(int argc, char** argv) {
  // STORM_SYNC=blocking|yield: how host threads wait for the GPU (default: HIP's choice, spinning).  With as many
  // driver threads as cores, spinning waits starve the runtime's own completion thread.
  if (const char* mode = std::getenv("STORM_SYNC")) {
    const unsigned flag = mode[0] == 'b' ? 0x4u : (mode[0] == 'y' ? 0x2u : 0x1u);
    if (hipSetDeviceFlags(flag) != 0) std::fprintf(stderr, "hipSetDeviceFlags failed\n");
  }
  if (argc < 2) { std::fprintf(stderr, "usage: many_scans model.bin [scans] [steps] [scheme] [threads...]\n"); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("model"); return 2; }
  uint64_t n = 0; double rs = 0, msx = 0;
  if (std::fread(&n, 8, 1, f) != 1 || std::fread(&rs, 8, 1, f) != 1 || std::fread(&msx, 8, 1, f) != 1) return 2;
  std::vector<std::vector<double>> a(7, std::vector<double>(n));
  for (auto& v : a) if (std::fread(v.data(), 8, n, f) != n) return 2;
  std::fclose(f);
  const size_t scans = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 64, steps = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 300;
  const int scheme = argc > 4 ? std::atoi(argv[4]) : 1;
  std::vector<int> thread_counts;
  for (int k = 5; k < argc; ++k) thread_counts.push_back(std::atoi(argv[k]));
  if (thread_counts.empty()) thread_counts = {1, 8, 16};
  const DiscreteModel base = DiscreteModel::from_gsm_arrays(n, a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(),
                                                            a[5].data(), a[6].data(), rs, msx);
  const std::vector<double> grid = base.dimensionless.r_coord;
  const std::vector<double> freqs = linspace(1.0, 25.0, steps);
  const std::pair<uint64_t, int64_t> combos[8] = {{0, 0}, {2, 0}, {4, 0}, {1, 1}, {1, -1}, {2, 1}, {2, -1}, {3, 0}};
  struct Phases { double solver = 0, scan = 0, post = 0, modes = 0, k_eigen = 0, k_build = 0, k_post = 0, k_det = 0; size_t found = 0; };
  const bool profile = std::getenv("STORM_PROFILE") != nullptr;
  auto job = [&](size_t k, Phases& ph) {
    DiscreteModel model = base;
    const double rot = 0.01 + 0.002 * double(k / 8);
    for (double& r : model.dimensionless.rot) r = rot;
    const auto [ell, m] = combos[k % 8];
    auto t = Clock::now();
    LinearInterpolator cm(model);
    ErasedSolver solver(cm, Rotating1D::make(ell, m), DifferenceSchemes(scheme), grid);
    if (profile) storm_solver_set_option(solver.handle(), "profile", 1);
    ph.solver += ms_since(t); t = Clock::now();
    const std::vector<BracketResult> res = solver.scan_and_optimize(freqs, Precision::Relative(1e-8));
    ph.scan += ms_since(t); t = Clock::now();
    if (res.empty()) return;
    PostProcessor pp(cm.eval(grid));
    ph.post += ms_since(t); t = Clock::now();
    std::vector<double> roots(res.size());
    for (size_t i = 0; i < res.size(); ++i) roots[i] = res[i].root;
    const auto ids = pp.modes(solver, roots, false);
    ph.modes += ms_since(t);
    ph.found += ids.size();
    if (profile) {
      double ms[STORM_KERNEL_KINDS]; uint64_t cnt[STORM_KERNEL_KINDS];
      storm_solver_kernel_times(solver.handle(), ms, cnt, 1);
      ph.k_eigen += ms[STORM_KERNEL_EIGEN]; ph.k_build += ms[STORM_KERNEL_BUILD_STEPS]; ph.k_post += ms[STORM_KERNEL_POSTPROCESS];
      ph.k_det += ms[STORM_KERNEL_DET_HEX] + ms[STORM_KERNEL_CHAIN];
    }
  };
  { Phases warm; job(0, warm); }
  // a negative "thread count" -B runs the same jobs as SolverSweeps of B systems each (one host thread):
  // every system of a sweep advances in lock-step, one launch per phase for all of them
  auto sweep_run = [&](size_t batch) {
    const auto t0 = Clock::now();
    double t_new = 0, t_scan = 0, t_post = 0, t_modes = 0;
    size_t found = 0;
    for (size_t first = 0; first < scans; first += batch) {
      const size_t cnt = std::min(batch, scans - first);
      auto t = Clock::now();
      // one model per rotation rate, shared by its (ell, m) systems
      std::vector<std::unique_ptr<DiscreteModel>> models;
      std::vector<std::unique_ptr<LinearInterpolator>> cms;
      std::vector<size_t> model_of(cnt);
      std::vector<SweepSystem> systems(cnt);
      size_t last_rate = size_t(-1);
      for (size_t j = 0; j < cnt; ++j) {
        const size_t k = first + j, rate = k / 8;
        if (rate != last_rate) {
          models.push_back(std::make_unique<DiscreteModel>(base));
          const double rot = 0.01 + 0.002 * double(rate);
          for (double& r : models.back()->dimensionless.rot) r = rot;
          cms.push_back(std::make_unique<LinearInterpolator>(*models.back()));
          last_rate = rate;
        }
        model_of[j] = cms.size() - 1;
        systems[j] = SweepSystem{cms.back().get(), Rotating1D::make(combos[k % 8].first, combos[k % 8].second)};
      }
      SolverSweep sweep(systems, DifferenceSchemes(scheme), grid);
      t_new += ms_since(t); t = Clock::now();
      const auto res = sweep.scan_and_optimize(std::vector<std::vector<double>>(cnt, freqs), Precision::Relative(1e-8));
      t_scan += ms_since(t); t = Clock::now();
      std::vector<std::unique_ptr<PostProcessor>> pps;
      for (auto& cm : cms) pps.push_back(std::make_unique<PostProcessor>(cm->eval(grid)));
      std::vector<const PostProcessor*> posts(cnt);
      std::vector<std::vector<double>> roots(cnt);
      for (size_t j = 0; j < cnt; ++j) {
        posts[j] = pps[model_of[j]].get();
        for (const BracketResult& b : res[j]) roots[j].push_back(b.root);
      }
      t_post += ms_since(t); t = Clock::now();
      const auto ids = sweep.modes(posts, roots, false);
      t_modes += ms_since(t);
      for (const auto& v : ids) found += v.size();
    }
    const double wall = ms_since(t0);
    std::printf("sweep %4zu: %8.1f ms for %zu scans x %zu frequencies = %6.3f ms/scan, %zu modes, %7.0f modes/s | per-scan phases (ms): "
                "sweep-create %.3f scan %.3f postproc-create %.3f modes %.3f\n", batch, wall, scans, steps, wall / scans, found,
                found / wall * 1e3, t_new / scans, t_scan / scans, t_post / scans, t_modes / scans);
    std::fflush(stdout);
  };
  for (int th : thread_counts) {
    if (th < 0) { sweep_run(size_t(-th)); continue; }
    std::atomic<size_t> next{0};
    std::vector<Phases> per(th);
    const auto t0 = Clock::now();
    std::vector<std::thread> pool;
    for (int w = 0; w < th; ++w)
      pool.emplace_back([&, w] { for (size_t k; (k = next.fetch_add(1)) < scans;) job(k, per[w]); });
    for (auto& t : pool) t.join();
    const double wall = ms_since(t0);
    Phases sum;
    for (const Phases& p : per) { sum.k_eigen += p.k_eigen; sum.k_build += p.k_build; sum.k_post += p.k_post; sum.k_det += p.k_det; sum.solver += p.solver; sum.scan += p.scan; sum.post += p.post; sum.modes += p.modes; sum.found += p.found; }
    std::printf("threads %3d: %8.1f ms for %zu scans x %zu frequencies = %6.2f ms/scan, %zu modes, %7.0f modes/s | mean per-scan phases (ms): "
                "solver %.2f scan %.2f postproc-create %.2f modes %.2f\n", th, wall, scans, steps, wall / scans, sum.found,
                sum.found / wall * 1e3, sum.solver / scans, sum.scan / scans, sum.post / scans, sum.modes / scans);
    if (profile)
      std::printf("             device ms per scan (HIP events): eigen %.2f  build_steps %.2f  postprocess %.2f  det_hex+chain %.2f\n",
                  sum.k_eigen / scans, sum.k_build / scans, sum.k_post / scans, sum.k_det / scans);
    std::fflush(stdout);
  }
  return 0;
}
