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

struct PostArgs {
  PostModelDev model;
  PostParams par;
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
