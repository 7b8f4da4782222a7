// storm_batch -- runs a StORM batch input (the command language of `storm`, reference src/main.rs:95-275,560-640) for
// the rotation-sweep pipeline of scripts/generate-batch-input-deformed.py on the GPU:
//
//     set-rotation-constant V | scan ELL M LOWER UPPER STEPS [PRECISION] [--inverse] [--precision P]
//     [--difference-scheme S] | post-process | deform ROT | perturb-deformed M | output LABEL | clear
//
// The reference executes the lines one by one, building one ErasedSolver per `scan` (main.rs:815-831).  A block of
// lines up to `output` is independent of every other block (`output` clears the solutions, main.rs:1206-1210; the
// rotation profile and the deformation persist and are tracked here), so this driver executes GROUPS of blocks in
// lock-step: all `scan`s of a group as one storm_sweep per difference scheme, all modes of the group in one
// eigenvector + post-processing pass, then per block the structure deformation (host, as in the reference) and the
// coupling integrals (device).  Results are bit-identical to line-by-line execution (tests/test_hip_batch.py).
//
// Not the reference's CLI: the model comes from a flat file (u64 n, f64 R_star, f64 M_star, then r, M_r, rho, P,
// Gamma_1, N2, Omega_rot; tools/make_model_bin.py writes it from a tests/golden fixture) instead of `input FILE.GSM`
// (an `input` line is accepted and ignored), frequencies are in dynamical units only, and `output` writes a text
// record (hex floats, exact) instead of HDF5; the generalised eigenproblem after the integrals (linalg/qz.rs) is the
// caller's.
//
// usage: storm_batch model.bin batch.txt out.txt [--group N] [--threads T] [--summary]
//        N = blocks executed together (default 16), T = host threads for the per-block integrals (default 8),
//        --summary writes one checksum per coupling matrix instead of its entries (timing runs)
//        --pipeline P runs P groups side by side from P host threads (default 4; the phase times printed at the end
//        are then sums over overlapping groups)
//        --shard K/N runs the K-th of N contiguous shares of the blocks (blocks are independent: one process per
//        GPU, no communication), --device D selects the HIP device
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../storm_amd/csrc/host/storm_host.hpp"

using namespace storm_host;
using Clock = std::chrono::steady_clock;
extern "C" int hipSetDevice(int device);  // libamdhip64 (already a dependency of libstorm_hip)

static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

namespace {

struct ScanCmd {
  uint64_t ell; int64_t m; double lower, upper; size_t steps; bool inverse; double precision; int scheme;
  double rot;  // the model's rotation constant when the line is reached
};
struct Block {
  std::vector<ScanCmd> scans;
  bool post = false;
  double post_rot = 0.;                                 // rotation constant at `post-process`
  std::vector<std::pair<int64_t, double>> perturb;      // (m, rotation constant at `perturb-deformed`)
  bool has_metric = false;
  double metric_model_rot = 0., metric_rot = 0.;        // `deform ROT` in effect: the model's constant then, and ROT
  std::string label;
};

[[noreturn]] void die(size_t line, const std::string& msg) {
  std::fprintf(stderr, "storm_batch: line %zu: %s\n", line, msg.c_str());
  std::exit(2);
}

int scheme_of(const std::string& s, size_t line) {
  static const char* names[8] = {"colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"};
  for (int k = 0; k < 8; ++k) if (s == names[k]) return k;
  die(line, "unknown difference scheme " + s);
}

std::vector<Block> parse(const char* path) {
  std::ifstream in(path);
  if (!in) { std::perror(path); std::exit(2); }
  std::vector<Block> blocks;
  Block cur;
  double rot = 0.;              // state that survives `output`
  bool has_metric = false;
  double metric_model_rot = 0., metric_rot = 0.;
  bool rot_set = false;
  std::string text;
  size_t line = 0;
  while (std::getline(in, text)) {
    ++line;
    std::istringstream ss(text);
    std::vector<std::string> w;
    for (std::string t; ss >> t;) w.push_back(t);
    if (w.empty() || w[0][0] == '#') continue;
    const std::string& c = w[0];
    auto num = [&](size_t k) { if (k >= w.size()) die(line, "missing argument"); return std::strtod(w[k].c_str(), nullptr); };
    if (c == "input") {
      if (!cur.scans.empty()) die(line, "Changing input models with already computed solutions is not supported");
    } else if (c == "set-rotation-constant") {
      for (size_t k = 2; k < w.size(); ++k)
        if (w[k] == "--frequency-units" && (k + 1 >= w.size() || w[k + 1] != "dynamical")) die(line, "only dynamical units");
      rot = num(1); rot_set = true;
    } else if (c == "scan") {
      ScanCmd s{};
      std::vector<std::string> pos;
      s.inverse = false; s.precision = 1e-8; s.scheme = 1;
      for (size_t k = 1; k < w.size(); ++k) {
        if (w[k] == "--inverse") s.inverse = true;
        else if (w[k] == "--precision") s.precision = num(++k);
        else if (w[k] == "--difference-scheme") { if (++k >= w.size()) die(line, "missing argument"); s.scheme = scheme_of(w[k], line); }
        else if (w[k] == "--frequency-units") { if (++k >= w.size() || w[k] != "dynamical") die(line, "only dynamical units"); }
        else pos.push_back(w[k]);
      }
      if (pos.size() < 5 || pos.size() > 6) die(line, "scan ELL M LOWER UPPER STEPS [PRECISION]");
      s.ell = std::strtoull(pos[0].c_str(), nullptr, 10); s.m = std::strtoll(pos[1].c_str(), nullptr, 10);
      s.lower = std::strtod(pos[2].c_str(), nullptr); s.upper = std::strtod(pos[3].c_str(), nullptr);
      s.steps = std::strtoull(pos[4].c_str(), nullptr, 10);
      if (pos.size() == 6) s.precision = std::strtod(pos[5].c_str(), nullptr);  // (as the reference's generator writes it)
      s.rot = rot;
      (void)rot_set;
      cur.scans.push_back(s);
    } else if (c == "post-process") {
      cur.post = true; cur.post_rot = rot;
    } else if (c == "deform") {
      has_metric = true; metric_model_rot = rot; metric_rot = num(1);
    } else if (c == "perturb-deformed") {
      if (!has_metric) die(line, "Deformed structure was not computed. Please run `deform` before `perturb-deformed`");
      if (!cur.post) die(line, "Post processing was not ran. Please run `post-process` before running `perturb-deformed`.");
      const int64_t m = std::strtoll(w.size() > 1 ? w[1].c_str() : "0", nullptr, 10);
      for (auto& p : cur.perturb) if (p.first == m) die(line, "Already computed perturbation for m = " + std::to_string(m));
      cur.perturb.push_back({m, rot});
      cur.has_metric = true; cur.metric_model_rot = metric_model_rot; cur.metric_rot = metric_rot;
    } else if (c == "output" || c == "clear") {
      cur.label = c == "output" && w.size() > 1 ? w[1] : "(cleared)";
      if (c == "output") blocks.push_back(cur);
      cur = Block{};
    } else {
      die(line, "unsupported command " + c);
    }
  }
  if (!cur.scans.empty()) { cur.label = "(no output)"; blocks.push_back(cur); }
  return blocks;
}

void put(FILE* f, double v) { std::fprintf(f, " %a", v); }

}  // namespace

static int run(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: storm_batch model.bin batch.txt out.txt [--group N]\n"); return 2; }
  size_t group = 16, threads = 8, shard = 0, shards = 1, pipeline = 4;
  int device = -1;
  int64_t coupling_tile = 0, arithmetic = 0;  // options of the sweeps (include/storm_hip.h): tile of the coupling kernel, FMA build
  bool summary = false;  // --summary: one checksum per matrix instead of its n*n entries (for timing runs)
  for (int k = 4; k < argc; ++k) {
    if (!std::strcmp(argv[k], "--group") && k + 1 < argc) group = std::strtoul(argv[++k], nullptr, 10);
    else if (!std::strcmp(argv[k], "--threads") && k + 1 < argc) threads = std::strtoul(argv[++k], nullptr, 10);
    else if (!std::strcmp(argv[k], "--pipeline") && k + 1 < argc) pipeline = std::strtoul(argv[++k], nullptr, 10);
    else if (!std::strcmp(argv[k], "--summary")) summary = true;
    else if (!std::strcmp(argv[k], "--coupling-tile") && k + 1 < argc) coupling_tile = std::atoi(argv[++k]);
    else if (!std::strcmp(argv[k], "--cache-gb") && k + 1 < argc) {  // fixed cache limit instead of the library's slab policy (for comparison)
      if (storm_memory_cache_limit(size_t(std::atoi(argv[++k])) << 30, size_t(8) << 30) != STORM_OK) return 2;
    }
    else if (!std::strcmp(argv[k], "--arithmetic") && k + 1 < argc) arithmetic = std::atoi(argv[++k]);
    else if (!std::strcmp(argv[k], "--device") && k + 1 < argc) device = std::atoi(argv[++k]);
    else if (!std::strcmp(argv[k], "--shard") && k + 1 < argc) {
      if (std::sscanf(argv[++k], "%zu/%zu", &shard, &shards) != 2 || shards == 0 || shard >= shards) {
        std::fprintf(stderr, "--shard K/N with 0 <= K < N\n"); return 2;
      }
    }
  }
  if (device >= 0 && hipSetDevice(device) != 0) { std::fprintf(stderr, "storm_batch: cannot select device %d\n", device); return 2; }
  if (group == 0) group = 1;
  if (threads == 0) threads = 1;
  FILE* mf = std::fopen(argv[1], "rb");
  if (!mf) { std::perror(argv[1]); return 2; }
  uint64_t n = 0; double rs = 0, msx = 0;
  if (std::fread(&n, 8, 1, mf) != 1 || std::fread(&rs, 8, 1, mf) != 1 || std::fread(&msx, 8, 1, mf) != 1) return 2;
  std::vector<std::vector<double>> a(7, std::vector<double>(n));
  for (auto& v : a) if (std::fread(v.data(), 8, n, mf) != n) return 2;
  std::fclose(mf);
  const DiscreteModel base = DiscreteModel::from_gsm_arrays(n, a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(),
                                                            a[5].data(), a[6].data(), rs, msx);
  const std::vector<double> grid = base.dimensionless.r_coord;  // Model::grid with resample 1
  const size_t np = grid.size();
  std::vector<Block> blocks = parse(argv[2]);
  if (shards > 1) {  // blocks are independent: process K of N takes a contiguous share (one process per GPU)
    const size_t base = blocks.size() / shards, extra = blocks.size() % shards;
    const size_t lo = shard * base + std::min(shard, extra), cnt = base + (shard < extra ? 1 : 0);
    blocks = std::vector<Block>(blocks.begin() + lo, blocks.begin() + lo + cnt);
  }
  FILE* final_out = std::fopen(argv[3], "w");
  if (!final_out) { std::perror(argv[3]); return 2; }

  // Every group in flight stages ~2.7 MB per mode (step matrices, factors, eigenvectors, fields).  The library's default
  // memory policy (slabs carved into blocks, kept while its footprint is below 3/4 of the device) recycles all of it
  // without calling hipMalloc / hipFree mid-run; --cache-gb replaces it by a fixed cache of separately allocated blocks.
  const auto t_all = Clock::now();
  struct Totals { double t_scan = 0, t_modes = 0, t_deform = 0, t_couple = 0, t_write = 0; size_t n_scans = 0, n_modes = 0; } tot;
  std::mutex tot_mu;
  const size_t n_groups = (blocks.size() + group - 1) / group;
  std::vector<std::string> records(n_groups);
  // One group of blocks: everything from solver construction to the group's records (kept in memory so that groups can
  // run side by side and still be written in input order).
  auto process_group = [&](size_t gi) -> int {
    const size_t g0 = gi * group;
    double t_scan = 0, t_modes = 0, t_deform = 0, t_couple = 0, t_write = 0;
    size_t n_scans = 0, n_modes = 0;
    char* mem = nullptr;
    size_t mem_len = 0;
    FILE* out = open_memstream(&mem, &mem_len);
    if (!out) { std::perror("open_memstream"); return 1; }
    const size_t g1 = std::min(blocks.size(), g0 + group);
    // one model (and one post-processing handle) per distinct rotation constant of the group
    std::map<double, std::unique_ptr<DiscreteModel>> models;
    std::map<double, std::unique_ptr<LinearInterpolator>> cms;
    std::map<double, std::unique_ptr<PostProcessor>> pps;
    auto model_of = [&](double rot) -> const LinearInterpolator* {
      auto it = cms.find(rot);
      if (it != cms.end()) return it->second.get();
      auto m = std::make_unique<DiscreteModel>(base);
      for (double& r : m->dimensionless.rot) r = rot;  // set-rotation-constant, main.rs:785-788 (dynamical units)
      cms[rot] = std::make_unique<LinearInterpolator>(*m);
      models[rot] = std::move(m);
      return cms[rot].get();
    };
    auto post_of = [&](double rot) -> const PostProcessor* {
      auto it = pps.find(rot);
      if (it != pps.end()) return it->second.get();
      pps[rot] = std::make_unique<PostProcessor>(model_of(rot)->eval(grid));
      return pps[rot].get();
    };
    // ---- all scans of the group, one sweep per difference scheme ----
    struct Ref { size_t block, scan; };
    std::map<int, std::vector<Ref>> by_scheme;
    for (size_t b = g0; b < g1; ++b)
      for (size_t s = 0; s < blocks[b].scans.size(); ++s) by_scheme[blocks[b].scans[s].scheme].push_back({b, s});
    // results per (block, scan)
    std::vector<std::vector<std::vector<BracketResult>>> found(g1 - g0);
    std::vector<std::vector<std::vector<Rotating1DPostprocessing>>> post(g1 - g0);
    for (size_t b = g0; b < g1; ++b) { found[b - g0].resize(blocks[b].scans.size()); post[b - g0].resize(blocks[b].scans.size()); }
    // If every scan of the group shares scheme and precision (the generator's case) there is ONE sweep, and the 13
    // fields of its modes stay in HBM between post-processing and the coupling integrals; otherwise they travel
    // through the host like in the reference.
    bool one_sweep = by_scheme.size() == 1;
    if (one_sweep) {
      const double p0 = blocks[by_scheme.begin()->second[0].block].scans[by_scheme.begin()->second[0].scan].precision;
      for (const Ref& r : by_scheme.begin()->second) one_sweep = one_sweep && blocks[r.block].scans[r.scan].precision == p0;
    }
    std::unique_ptr<SolverSweep> resident;                         // alive until the group's integrals are done
    std::vector<std::vector<size_t>> system_of(g1 - g0);           // (block, scan) -> system of `resident`
    for (auto& kv : by_scheme) {
      const std::vector<Ref>& refs = kv.second;
      // Precision is one per sweep call: scans are sub-grouped by it (normally a single value)
      std::map<double, std::vector<Ref>> by_prec;
      for (const Ref& r : refs) by_prec[blocks[r.block].scans[r.scan].precision].push_back(r);
      for (auto& pv : by_prec) {
        const std::vector<Ref>& rs2 = pv.second;
        auto t = Clock::now();
        std::vector<SweepSystem> systems;
        std::vector<std::vector<double>> grids;
        for (const Ref& r : rs2) {
          const ScanCmd& s = blocks[r.block].scans[r.scan];
          systems.push_back(SweepSystem{model_of(s.rot), Rotating1D::make(s.ell, s.m)});
          grids.push_back(s.inverse ? rev_linspace(s.lower, s.upper, s.steps) : linspace(s.lower, s.upper, s.steps));
        }
        auto sweep_ptr = std::make_unique<SolverSweep>(systems, DifferenceSchemes(kv.first), grid);
        SolverSweep& sweep = *sweep_ptr;
        // eigenvector staging for all modes of the group in one pass (2 MB per mode at 6199 intervals; the card has 288 GB)
        storm_sweep_set_option(sweep.handle(), "steps_bytes_max", int64_t(64) << 30);
        if (storm_sweep_set_option(sweep.handle(), "coupling_tile", coupling_tile) != STORM_OK ||
            storm_sweep_set_option(sweep.handle(), "arithmetic", arithmetic) != STORM_OK)
          throw std::runtime_error(storm_last_error());
        const auto res = sweep.scan_and_optimize(grids, Precision::Relative(pv.first));
        t_scan += ms_since(t); t = Clock::now();
        std::vector<std::vector<double>> roots(rs2.size());
        std::vector<const PostProcessor*> posts(rs2.size(), nullptr);
        bool fields = false;
        for (size_t k = 0; k < rs2.size(); ++k) {
          const Block& blk = blocks[rs2[k].block];
          found[rs2[k].block - g0][rs2[k].scan] = res[k];
          for (const BracketResult& br : res[k]) roots[k].push_back(br.root);
          // StormState::post_process evaluates the model as it is when the command runs (main.rs:878-899)
          posts[k] = post_of(blk.post ? blk.post_rot : blocks[rs2[k].block].scans[rs2[k].scan].rot);
          fields = fields || !blk.perturb.empty();
          n_modes += res[k].size();
        }
        n_scans += rs2.size();
        // the eigenvector of every mode + its post-processing; the 13 fields only if some block integrates them
        const bool keep = fields && one_sweep;
        const auto pr = keep ? sweep.modes_resident(posts, roots) : sweep.modes(posts, roots, fields);
        for (size_t k = 0; k < rs2.size(); ++k) post[rs2[k].block - g0][rs2[k].scan] = pr[k];
        if (keep) {
          for (size_t b = g0; b < g1; ++b) system_of[b - g0].assign(blocks[b].scans.size(), 0);
          for (size_t k = 0; k < rs2.size(); ++k) system_of[rs2[k].block - g0][rs2[k].scan] = k;
          resident = std::move(sweep_ptr);
        }
        t_modes += ms_since(t);
      }
    }
    // ---- per block: deformation (host) + coupling integrals (device); blocks are independent, so a few host threads
    // each take blocks (every rotation rate has its own post-processing handle and stream) ----
    struct Task { size_t block, which; CouplingMatrices c; size_t n = 0; };
    std::vector<Task> tasks;
    for (size_t b = g0; b < g1; ++b)
      for (size_t w = 0; w < blocks[b].perturb.size(); ++w) {
        Task tk; tk.block = b; tk.which = w;
        tasks.push_back(std::move(tk));
        model_of(blocks[b].metric_model_rot);      // (the maps are filled here, on one thread)
        post_of(blocks[b].perturb[w].second);
      }
    {
      auto t = Clock::now();
      std::atomic<size_t> next{0};
      std::atomic<long long> deform_us{0};
      std::string error;
      std::mutex error_mu;
      std::vector<PerturbedMetric> metrics(tasks.size());
      std::vector<SolverSweep::CouplingProblem> problems(tasks.size());
      // Resident path: the threads only deform (host, as in the reference); the integrals of every block of the group
      // then go out as ONE launch.  Host-fields path: each thread also integrates its blocks (one launch per block).
      auto work = [&] {
        if (device >= 0) (void)hipSetDevice(device);  // per-thread in HIP (the handles guard their own device as well)
        for (size_t at; (at = next.fetch_add(1)) < tasks.size();) {
          try {
            Task& tk = tasks[at];
            const Block& blk = blocks[tk.block];
            const auto& pm = blk.perturb[tk.which];
            const size_t b = tk.block;
            auto t0 = Clock::now();
            // deform: perturb_structure on the model as it was at `deform` (main.rs:840-876)
            metrics[at] = perturb_structure(*models.at(blk.metric_model_rot), blk.metric_rot);
            deform_us += (long long)(ms_since(t0) * 1e3);
            std::vector<ModeToPerturb> modes;
            std::vector<SolverSweep::ModeRef> refs;
            for (size_t s = 0; s < blk.scans.size(); ++s)
              if (blk.scans[s].m == pm.first)
                for (size_t k = 0; k < found[b - g0][s].size(); ++k) {
                  modes.push_back(ModeToPerturb{blk.scans[s].ell, found[b - g0][s][k].root, &post[b - g0][s][k]});
                  if (resident) refs.push_back({system_of[b - g0][s], k});
                }
            tk.n = modes.size();
            if (resident) problems[at] = SolverSweep::CouplingProblem{pps.at(pm.second).get(), refs, pm.first, &metrics[at]};
            else if (!modes.empty()) tk.c = pps.at(pm.second)->coupling_matrices(modes, pm.first, metrics[at]);
          } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lock(error_mu);
            if (error.empty()) error = e.what();
          }
        }
      };
      const size_t n_thr = std::min<size_t>(tasks.size(), threads);
      std::vector<std::thread> pool;
      for (size_t k = 1; k < n_thr; ++k) pool.emplace_back(work);
      work();
      for (auto& th : pool) th.join();
      if (error.empty() && resident && !tasks.empty()) {
        try {
          std::vector<CouplingMatrices> all = resident->coupling_matrices(problems);
          for (size_t at = 0; at < tasks.size(); ++at) tasks[at].c = std::move(all[at]);
        } catch (const std::exception& e) { error = e.what(); }
      }
      if (!error.empty()) { std::fprintf(stderr, "storm_batch: %s\n", error.c_str()); return 1; }
      t_deform += double(deform_us.load()) * 1e-3 / double(std::max<size_t>(1, n_thr));
      t_couple += ms_since(t);
    }
    // ---- the records ----
    auto t_w = Clock::now();
    size_t task_at = 0;
    for (size_t b = g0; b < g1; ++b) {
      const Block& blk = blocks[b];
      std::fprintf(out, "block %zu %s scans %zu\n", b, blk.label.c_str(), blk.scans.size());
      for (size_t s = 0; s < blk.scans.size(); ++s) {
        const auto& br = found[b - g0][s];
        std::fprintf(out, "scan %llu %lld modes %zu\n", (unsigned long long)blk.scans[s].ell, (long long)blk.scans[s].m, br.size());
        for (size_t k = 0; k < br.size(); ++k) {
          std::fprintf(out, "mode");
          put(out, br[k].lower.x); put(out, br[k].lower.f); put(out, br[k].upper.x); put(out, br[k].upper.f); put(out, br[k].root);
          const Rotating1DPostprocessing& p = post[b - g0][s][k];
          std::fprintf(out, " %llu %llu %llu %lld %zu\n", (unsigned long long)br[k].evals, (unsigned long long)p.cross_clockwise,
                       (unsigned long long)p.cross_counter_clockwise, (long long)p.radial_order, p.nodes.size());
        }
      }
      for (size_t w = 0; w < blk.perturb.size(); ++w, ++task_at) {
        const Task& tk = tasks[task_at];
        std::fprintf(out, "coupling %lld n %zu\n", (long long)blk.perturb[w].first, tk.n);
        if (tk.n) {
          const CouplingMatrices& c = tk.c;
          const MatrixView* mats[6] = {&c.d_squared, &c.d_linear, &c.d_zero, &c.r_linear, &c.r_zero, &c.l_zero};
          for (int q = 0; q < 6; ++q) {
            std::fprintf(out, "matrix %d", q);
            if (!summary) for (double v : *mats[q]) put(out, v);
            else { double sum = 0.; for (double v : *mats[q]) sum += v; put(out, sum); }
            std::fprintf(out, "\n");
          }
        }
      }
    }
    t_write += ms_since(t_w);
    (void)np;
    std::fclose(out);
    records[gi].assign(mem, mem_len);
    std::free(mem);
    std::lock_guard<std::mutex> lock(tot_mu);
    tot.t_scan += t_scan; tot.t_modes += t_modes; tot.t_deform += t_deform; tot.t_couple += t_couple; tot.t_write += t_write;
    tot.n_scans += n_scans; tot.n_modes += n_modes;
    return 0;
  };
  // Groups are independent too: `pipeline` host threads each take the next group, so the host side of one group (model
  // interpolation, plane construction, weights, result copies: about half of a group's wall time) overlaps the
  // kernels of another.
  {
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    std::string error;
    std::mutex error_mu;
    auto worker = [&] {
      // HIP's current device is per host thread: every thread that creates handles selects it (handles take their
      // device from hipGetDevice at creation)
      if (device >= 0 && hipSetDevice(device) != 0) {
        std::lock_guard<std::mutex> lock(error_mu);
        if (error.empty()) error = "cannot select the device in a pipeline thread";
        failed = 1;
        return;
      }
      for (size_t gi; !failed.load() && (gi = next.fetch_add(1)) < n_groups;) {
        try {
          if (process_group(gi) != 0) failed = 1;
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> lock(error_mu);
          if (error.empty()) error = e.what();
          failed = 1;
        }
      }
    };
    const size_t n_thr = std::max<size_t>(1, std::min(pipeline, n_groups));
    std::vector<std::thread> pool;
    for (size_t k = 1; k < n_thr; ++k) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    if (!error.empty()) throw std::runtime_error(error);
    if (failed.load()) return 1;
  }
  for (const std::string& r : records) std::fwrite(r.data(), 1, r.size(), final_out);
  std::fclose(final_out);
  const double t_scan = tot.t_scan, t_modes = tot.t_modes, t_deform = tot.t_deform, t_couple = tot.t_couple, t_write = tot.t_write;
  const size_t n_scans = tot.n_scans, n_modes = tot.n_modes;
  const double wall = ms_since(t_all);
  std::fprintf(stderr, "storm_batch: %zu blocks, %zu scans, %zu modes in %.1f ms (%.3f ms per block, %.0f modes/s) | scans %.1f ms, "
               "eigenvectors + post-processing %.1f ms, deform + coupling integrals %.1f ms (deform %.1f), writing %.1f ms\n",
               blocks.size(), n_scans, n_modes, wall, wall / std::max<size_t>(1, blocks.size()), n_modes / wall * 1e3, t_scan, t_modes,
               t_couple, t_deform, t_write);
  return 0;
}

int main(int argc, char** argv) {
  try {
    return run(argc, argv);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "storm_batch: %s (a smaller --group needs less device memory per pass)\n", e.what());
    return 1;
  }
}
