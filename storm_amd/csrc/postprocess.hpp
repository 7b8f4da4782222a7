#pragma once
// Device post-processing of eigenvectors: Rotating1DPostprocessing::new + count_windings
// (reference src/postprocessing.rs:58-355).  SURVEY.md 8(f) rank 1.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace storm {

// Model planes on the device, [kPostPlanes][n_pts] FP64 (DimensionlessProperties, reference src/model.rs:70-91,
// plus the trapezoid weights of src/postprocessing.rs:99-111, which depend on the grid only).
enum PostPlane : int { PP_R = 0, PP_M, PP_RHO, PP_P, PP_V, PP_U, PP_GAMMA1, PP_ASTAR, PP_C1, PP_ROT, PP_TRAP, kPostPlanes };

constexpr int kPostFields = 13;  // storm_post_field of include/storm_hip.h
enum PostField : int { Y1, Y2, Y3, Y4, XI_R, XI_H, XI_TN, XI_TP, PSI, DPSI, RHO, PRS, CHI };  // rows of a mode's field block

struct PostModelDev {
  const double* tab;  // [kPostPlanes][n_pts]
  size_t n_pts;
};

// Per-(ell, m) scalars of src/postprocessing.rs:85-95, evaluated once on the host.
struct PostParams {
  int ell;
  int tn_zero;  // |m| == ell || ell == 1  ->  xi_tn = 0
  double lambda, lambda_n1, lambda_p1, q_hd_n, q_hd_p, q_h_n, q_h_p, mf, ellf;
};

// One entry per system when the modes of several systems share a launch (storm_sweep_modes).
struct PostSystem {
  PostModelDev model;
  PostParams par;
};

struct PostArgs {
  PostModelDev model;
  PostParams par;
  // several systems per launch: mode k belongs to systems[mode_sys[k]] (device arrays; every model has the same n_pts,
  // given in `model.n_pts`); systems == nullptr: the one (model, par) above
  const PostSystem* systems = nullptr;
  const int* mode_sys = nullptr;
  const double* freq;   // device [n]
  const double* vec;    // device [n][n_pts*4], raw eigenvectors (last component 1)
  size_t n;
  double* terms;        // device scratch [n][n_pts]
  double* scale;        // device [n]: 1/sqrt(norm)
  double* fields;       // device [n][13][n_pts] or nullptr (skip the field kernel)
  long long* ids;       // device [n][4]: clockwise, counter-clockwise, radial order, number of nodes
  double* nodes;        // device [n][node_cap] or nullptr
  size_t node_cap;
  hipStream_t stream;
};

// Three launches on args.stream: norm terms -> per-mode ordered sum + windings -> scaled fields (if asked).
hipError_t launch_postprocess(const PostArgs& args);

}  // namespace storm

// ---- mode-coupling integrals of the deformed star: perturb_deformed, reference src/perturbed.rs:110-263 ----
namespace storm {

constexpr int kCouplingMatrices = 6;  // d_squared, d_linear, d_zero, r_linear, r_zero, l_zero
// Per-point weights, identical for every mode pair (evaluated once on the host, same expressions):
enum CouplingWeight : int {
  CW_TR2 = 0,   // trapezoid * r^2
  CW_VOL,       // trapezoid * r^2 * rho
  CW_VOL2,      // vol * 2
  CW_VOL2ROT,   // vol * 2 * rot
  CW_C1,        // 2 * m * rot
  CW_C2,        // m * m * rot * rot
  CW_MROT,      // m * rot
  CW_RHO, CW_X,
  CW_U1,        // 1 + 2 (eps_a + a d eps_a)
  CW_U2,        // 1 + 2 eps_a
  CW_U3,        // eps_b + a d eps_b
  CW_U4,        // 1 + 2 eps_a + a d eps_a
  CW_U5,        // 2 eps_b + a d eps_b
  CW_EB, CW_TEA, CW_ADTEA, CW_TEB, CW_ADTEB,
  CW_RX,        // refined reciprocal of x (shared_div.hpp), filled on the device by the launch
  kCouplingWeights = 20  // rows of 160 B
};
// Per-(ell_left, ell_right) Gaunt factors (src/perturbed.rs:126-146).
enum CouplingConst : int { CQ_Q2 = 0, CQ_Q2_HRD, CQ_Q2_HLD, CQ_Q2_H, CQ_RT_P, CQ_RT_N, CQ_HT_P, CQ_HT_N, CQ_IPR, CQ_IPH, kCouplingConsts };

struct CouplingArgs {
  size_t n_pts, n_modes;
  const double* fields;    // device [..][13][n_pts], post-processing layout: mode k of the problem is row k, or row
  const long long* rows;   // rows[k] when `rows` is not null (a subset of a larger resident set, no gather copy)
  const double* freq;      // device [n_modes]
  const int* ell_index;    // device [n_modes], index into the distinct-ell list
  const double* consts;    // device [n_ell][n_ell][kCouplingConsts]
  int n_ell;
  double* weights;         // device [n_pts][kCouplingWeights]; the launch fills column CW_RX
  double* out;             // device [kCouplingMatrices][n_modes*n_modes], (left, right) at left + right*n_modes
  int* flags;              // device; bit 0 set if a term was not finite (the reference asserts)
  hipStream_t stream;
  // Shortcut for tiles whose ten Gaunt factors are all zero (degrees more than 2 apart): see coupling_body.  `tame`:
  // device [n_modes] scratch the launch fills (1 = every field value and the frequency of the mode are finite and
  // below 2^200 in magnitude), nullptr = never take the shortcut; weights_tame: the same property of `weights` (host).
  int* tame = nullptr;
  int weights_tame = 0;
  // FMA build, matrix-pipe walk: the problem's modes grouped by degree, every group padded to a multiple of 16 positions;
  // device [n_pad], order[pos] = index into the arrays above (0 .. n_modes - 1) or -1 for padding; a tile's first position
  // is never padding.  nullptr: the vector walk (always in the parity build).
  const int* order = nullptr;
  size_t n_pad = 0;
  // Matrix-pipe walk of ONE problem, split along the grid: workgroup (tile, piece) walks the piece's share of the points and
  // writes its sums to partial[piece] (device [n_split][kCouplingMatrices][n_modes*n_modes]); a second launch adds the
  // pieces in ascending order into `out` (deterministic).  n_split = 1: no split, the walk writes `out` itself.  From
  // plan_coupling_matrix; what it buys: a tile count that is not a multiple of the chip's resident workgroups no longer
  // leaves a mostly empty last round (M = 1024: 1089 tiles on 512 slots = 2.13 rounds paid as 3).
  unsigned n_split = 1;
  double* partial = nullptr;
  size_t partial_stride = 0;   // doubles from one piece's block to the next (one problem: 6 n_modes^2; a batch: all problems' outputs)
};

// Tile (16 or 32) and number of pieces of the matrix-pipe walk for one problem of n_pad ordered positions on n_pts grid
// points; matrix_tile / split: 0 = automatic, else forced (options "coupling_matrix_tile", "coupling_split").
struct CouplingPlan { int tile; unsigned n_split; };
CouplingPlan plan_coupling_matrix(size_t n_pad, size_t n_pts, int matrix_tile, int split);
CouplingPlan plan_coupling_batch(size_t max_pad, size_t n_problems, size_t n_pts, int matrix_tile, int split);

// dst[k][0 .. row_elems) = src[index[k]][0 .. row_elems) for k < n (device arrays): picks the post-processing fields of
// a subset of modes out of a larger resident set without leaving HBM.
hipError_t launch_gather_rows(const double* src, const long long* index, size_t n, size_t row_elems, double* dst,
                              hipStream_t stream);

}  // namespace storm

// The coupling launchers, one set per arithmetic build (coupling_kernels.hip; storm = the reference's association,
// storm_fast = FMA contraction allowed).  launch_coupling_batch: several coupling problems in one launch, blockIdx.z
// picks table[z] (device array; args.stream is ignored); max_modes = the largest n_modes in the table, n_pts = the
// (common) grid size of the entries; with_tame: some entry has a `tame` array to fill first.  tile: modes per side of a
// workgroup's tile of pairs, 8 (one wave) or 16 (four waves sharing the staged fields).  max_pad: the largest n_pad of
// the table if EVERY entry carries an ordered mode list (FMA build: the matrix-pipe walk takes the launch), else 0;
// matrix_tile: 0 = automatic, 16 or 32 = that tile of the matrix-pipe walk.  n_split > 1 (matrix-pipe walk only; every entry
// carries n_split, its `partial` and partial_stride = out_total): the pieces are added into out_all[0 .. out_total) -- the
// entries' outputs, back to back -- by a second launch; plan_coupling_batch gives tile and pieces.
#define STORM_DECL_COUPLING(ns)                                                                                         \
  namespace ns {                                                                                                        \
  hipError_t launch_coupling(const storm::CouplingArgs& args, int tile, int matrix_tile);                               \
  hipError_t launch_coupling_batch(const storm::CouplingArgs* table, size_t n_problems, size_t max_modes, size_t n_pts, \
                                   hipStream_t stream, bool with_tame, int tile, size_t max_pad, int matrix_tile,       \
                                   unsigned n_split, const double* partial, double* out_all, size_t out_total, int* flags); \
  }
STORM_DECL_COUPLING(storm)
STORM_DECL_COUPLING(storm_fast)
