/* storm_hip.h -- C ABI of libstorm_hip: StORM's frequency-scan hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary under StORM's type-erased solver.  The reference has no FFI of
 * its own; each entry point below replaces the body of one reference function/closure and is
 * what `build.rs` + an `extern "C"` block on the Rust side would bind.
 * All paths in comments are relative to the reference root (VincentVanlaer/StORM @ 2025-10-31).
 *
 * Conventions: plain pointers and sizes only; the caller owns every host buffer; a handle owns
 * its device memory and stream; every entry point returns 0 on success or a negative
 * storm_status and never unwinds/aborts; storm_last_error() gives a thread-local message.
 * Entry points taking the same handle may be called concurrently from several host threads
 * (the reference calls its closures from rayon workers, src/dynamic_interface.rs:61-62); calls
 * are serialised per handle.  Results preserve input order.  All floating point is IEEE f64.
 */
#ifndef STORM_HIP_H
#define STORM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  STORM_OK = 0,
  STORM_ERR_INVALID_ARGUMENT = -1,
  STORM_ERR_HIP = -2,            /* a HIP runtime call failed (no GPU, out of memory, ...) */
  STORM_ERR_SINGULAR = -3,       /* a collocation stage matrix was singular: the reference panics
                                    (try_inverse().unwrap(), src/stepper.rs:88,199,250,...) */
  STORM_ERR_NON_FINITE = -4,     /* a discriminant evaluated during refinement was not finite: the
                                    reference panics (assert!, src/bracket.rs:209-210) */
  STORM_ERR_CAPACITY = -5,       /* output buffer too small; *n_out holds the required count */
  STORM_ERR_BRACKET = -6         /* ensure_maximal_bracket precondition violated (src/bracket.rs:102-112) */
} storm_status;

/* DifferenceSchemes, src/dynamic_interface.rs:24-57, in this order. */
typedef enum {
  STORM_COLLOC2 = 0, STORM_COLLOC4 = 1, STORM_COLLOC6 = 2, STORM_COLLOC8 = 3,
  STORM_MAGNUS2 = 4, STORM_MAGNUS4 = 5, STORM_MAGNUS6 = 6, STORM_MAGNUS8 = 7
} storm_scheme;

typedef struct storm_solver storm_solver; /* opaque; = one ErasedSolver (src/dynamic_interface.rs:60-63) */

/* = BracketResult, src/bracket.rs:6-15 (lower/upper are Point{x,f}, src/bracket.rs:63-68). */
typedef struct {
  double lo_x, lo_f, hi_x, hi_f, root;
  uint64_t evals;
} storm_bracket;

/* What DiscretizedSystemImpl::new (src/system/discretized.rs:81-123) knows after interpolating
 * the model at the collocation abscissae.  Arrays are SoA, interval-major: entry [i*p + j] is
 * collocation point j of grid interval i, exactly the reference's `interpolated_points` /
 * `points` order.  Host pointers, borrowed for the duration of storm_solver_create only. */
typedef struct {
  int32_t scheme;        /* storm_scheme */
  uint64_t ell;          /* Rotating1D::new(ell, m), src/system/adiabatic.rs:19-24 */
  int64_t m;
  size_t n_int;          /* grid intervals = solver grid points - 1 */
  size_t p;              /* collocation points per interval; must match the scheme (1,2,3,4) */
  const double* u;       /* AdiabaticStructureCoefficients, src/system/adiabatic.rs:29-40 */
  const double* v_gamma;
  const double* a_star;
  const double* c1;
  const double* rot;
  const double* delta_over_x; /* delta[i] / points[i*p+j], src/system/discretized.rs:107,198 */
  double inner_c1, inner_rot; /* model point at model.inner(), used by inner_boundary (adiabatic.rs:164-168) */
  double outer_u;             /* model point at model.outer(), used by outer_boundary (adiabatic.rs:204) */
} storm_system_desc;

/* Replaces get_solvers_inner_explicit (src/dynamic_interface.rs:193-221): builds the cached
 * frequency-independent matrices (Rotating1D::eval, src/system/adiabatic.rs:78-114) and uploads
 * them to the current HIP device. */
int storm_solver_create(const storm_system_desc* desc, storm_solver** out);
void storm_solver_destroy(storm_solver* solver);

/* Replaces the `det` closure = determinant_explicit (src/dynamic_interface.rs:212,
 * src/solver.rs:435-542), batched: det_out[k] = det(freq[k]).  Raw, unscaled discriminant; a
 * non-finite value is returned as such (overflow parity).  Host pointers. */
int storm_det_batch(storm_solver* solver, const double* freq, size_t n, double* det_out);

/* Same with device pointers (inputs/outputs resident in HBM), asynchronous on `hip_stream`
 * (a hipStream_t, may be NULL for the solver's own stream, in which case the call returns
 * after the work completed).  On a caller's stream the call waits (on the device) for the handle's asynchronous
 * create-time upload, returns without synchronising and cannot report STORM_ERR_SINGULAR: a singular stage matrix
 * shows as NaN in d_det_out. */
int storm_det_batch_device(storm_solver* solver, const double* d_freq, size_t n, double* d_det_out, void* hip_stream);

/* The discriminant as mantissa * 2^exponent (mantissa in [0.5, 1), like frexp): the running product of the transfer
 * matrices is renormalised by exact powers of two along the grid, so the result does not overflow where the reference's
 * raw product does (det *= pivot, src/solver.rs:341,383,422,504,531; the reference detects overflow after the fact,
 * src/dynamic_interface.rs:130-137).  An ADDITION to the raw value of storm_det_batch, which is unchanged (overflow
 * parity): mathematically the same number; it agrees with the raw discriminant to rounding (relative ~1e-13, the closing
 * elimination may pivot in a different order), signs included, wherever the raw one is finite.  Zero, inf and NaN are
 * returned as such with exponent 0.  Parity build: evaluated through the staged step matrices (any batch size, chunked by
 * "steps_bytes_max"), about twice storm_det_batch's time for large batches.  FMA build: through segment transfer matrices
 * (fast kernels) and a scaled chain over the segments, ~1.3x storm_det_batch's time; frequencies whose segment product
 * overflows fall back to the staged path. */
int storm_det_scaled_batch(storm_solver* solver, const double* freq, size_t n, double* mantissa_out, int64_t* exponent_out);

/* storm_det_batch_device for callers that cannot allocate.  NO counterpart in the reference: it exists for callers
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same kernels and the same choice
 * between them as storm_det_batch_device (options "path", "fused_min_n", "quad_min_n", "hex_min_n", "segments" included),
 * with one difference: the staged path (step matrices in a buffer of 128 B per (frequency, interval): "path" = 2, or a
 * small batch that has no segment plan -- parity build, or fewer than 64 intervals) is replaced by det_hex, which forms
 * the same ordered product.  Results are bit-identical to storm_det_batch_device in the parity build on every path, and
 * in the FMA build wherever both take the same kernels (always, except where det_hex replaces the staged path: there the
 * FMA build's two kernels agree to its tolerance contract).  The two scratch areas the kernels need -- the
 * singular-stage flag word, and the segment transfer matrices (128 B per (frequency, segment) pair) of the FMA build's
 * segment path, which small batches AND the last partial wave round of large batches take -- live in a file-scope device
 * array of STORM_NOALLOC_SCRATCH_BYTES that every GPU holds once per process, so the call reaches no hipMalloc / hipFree /
 * hipHostMalloc and creates no event or stream.  A segment plan whose matrices would not fit the scratch (possible only
 * with a forced "segments" length) is dropped and det_hex takes those frequencies.  Limit: n <= STORM_NOALLOC_MAX_N,
 * STORM_ERR_CAPACITY otherwise.  A singular stage matrix shows
 * as NaN in d_det_out (no status is read back).  hip_stream NULL: the solver's own stream, and the call returns after
 * the work completed; otherwise asynchronous on that stream.  All allocation-free calls on one GPU share the scratch:
 * calls that overlap on one GPU (different streams or handles) race; calls on one stream are safe. */
#define STORM_NOALLOC_MAX_N ((size_t)1 << 24)
#define STORM_NOALLOC_SCRATCH_BYTES (((size_t)16 << 20) + 256)
int storm_det_batch_device_noalloc(storm_solver* solver, const double* d_freq, size_t n, double* d_det_out, void* hip_stream);

/* Replaces ErasedSolver::scan_and_optimize (src/dynamic_interface.rs:116-160): discriminant on
 * every grid point, sign-swap filter on adjacent points (FilterSignSwap, src/bracket.rs:236-250)
 * dropping non-finite pairs (sets *had_invalid), then InverseQuadratic::optimize
 * (src/bracket.rs:151-222) on every bracket with `ulp` = Precision::as_ulp() (src/bracket.rs:30-40).
 * out[0..*n_out) in ascending bracket order.  det_out (may be NULL) receives the n scan values,
 * index_out (may be NULL, cap entries) the grid index of each bracket's lower point. */
int storm_scan_and_optimize(storm_solver* solver, const double* freq, size_t n, uint64_t ulp,
                            storm_bracket* out, size_t cap, size_t* n_out, int* had_invalid,
                            double* det_out, uint64_t* index_out);

/* The two halves of ErasedSolver::scan_and_optimize as separate calls, for callers that redistribute the brackets
 * between the halves (a frequency grid sharded over several GPUs: every GPU scans a slice, the brackets are dealt out
 * evenly for refinement).
 * storm_scan_brackets = the first half (src/dynamic_interface.rs:125-138): discriminants, FilterSignSwap
 * (src/bracket.rs:236-250), finite filter.  out[k] has lo_x/lo_f/hi_x/hi_f set, root = NaN, evals = 0.
 * storm_optimize_brackets = the second half (src/dynamic_interface.rs:139-151): InverseQuadratic::optimize
 * (src/bracket.rs:148-223) on each of the n brackets, in place; only lo_* and hi_* are read.  STORM_ERR_BRACKET where
 * the reference asserts or panics (non-finite end points, end values of equal sign, src/bracket.rs:159-166). */
int storm_scan_brackets(storm_solver* solver, const double* freq, size_t n, storm_bracket* out, size_t cap,
                        size_t* n_out, int* had_invalid, double* det_out, uint64_t* index_out);
int storm_optimize_brackets(storm_solver* solver, storm_bracket* brackets, size_t n, uint64_t ulp);

/* Replaces the `eigenvector` closure = determinant_with_upper + UpperResult::eigenvectors
 * (src/dynamic_interface.rs:213-219, src/solver.rs:100-131,271-433), batched over modes.
 * vec_out is [n][(n_int+1)*4], element [i*4+k] = y_{k+1}(x_i); det_out (may be NULL) is [n].  If vec_out is pinned host
 * memory (hipHostMalloc / hipHostRegister) the eigenvectors are copied into it by DMA, otherwise through the library's
 * pinned staging block and a host copy. */
int storm_eigenvector_batch(storm_solver* solver, const double* freq, size_t n, double* det_out, double* vec_out);

/* ---- Post-processing of found modes (the caller of the scan path: `post-process`, src/main.rs:878-899) ----
 *
 * Replaces Rotating1DPostprocessing::new (src/postprocessing.rs:58-301) and count_windings (:304-355),
 * batched over modes and run on the device, so eigenvectors need not leave HBM to get mode ids. */

/* DimensionlessProperties (src/model.rs:70-91) of the model EVALUATED ON THE SOLVER GRID, i.e. what
 * `input.as_continuous().eval(&input.grid())` returns (src/main.rs:892).  n_pts = n_int + 1 of the solvers
 * used with it; r_coord[0] must be 0 (assert_eq!, src/postprocessing.rs:69).  Host pointers, borrowed for
 * the duration of storm_postproc_create only. */
typedef struct {
  size_t n_pts;
  const double *r_coord, *m_coord, *rho, *p, *v, *u, *gamma1, *a_star, *c1, *rot;
} storm_model_desc;

typedef struct storm_postproc storm_postproc; /* opaque; owns the model arrays in HBM and a stream */

/* Field order of fields_out; = the Box<[f64]> members of Rotating1DPostprocessing, src/postprocessing.rs:14-42
 * (x is the caller's r_coord and is not returned). */
typedef enum {
  STORM_POST_Y1 = 0, STORM_POST_Y2, STORM_POST_Y3, STORM_POST_Y4, STORM_POST_XI_R, STORM_POST_XI_H,
  STORM_POST_XI_TN, STORM_POST_XI_TP, STORM_POST_PSI, STORM_POST_DPSI, STORM_POST_RHO, STORM_POST_P,
  STORM_POST_CHI, STORM_POST_FIELDS = 13
} storm_post_field;

/* cross_clockwise, cross_counter_clockwise, radial_order (src/postprocessing.rs:43-51) and nodes.len(). */
typedef struct {
  uint64_t cross_clockwise, cross_counter_clockwise;
  int64_t radial_order;
  uint64_t n_nodes;
} storm_mode_id;

int storm_postproc_create(const storm_model_desc* model, storm_postproc** out);
void storm_postproc_destroy(storm_postproc* post);
/* Options of a post-processing handle (no counterpart in the reference).  "arithmetic": 0 (default) = the coupling
 * integrals of storm_coupling_matrices / storm_sweep_coupling_matrices keep the reference's association and are
 * bit-identical to its sequential sums (src/perturbed.rs:148-261); 1 = the FMA build (the same integrands re-associated
 * and summed on v_mfma_f64_16x16x4, four grid points per instruction, over the modes ordered by degree inside the library:
 * the caller's order is free): every matrix element within 1e-12 of the largest magnitude of its matrix
 * (tests/test_hip_coupling.py; measured 1e-14), the same bits on every run.  Speed only, never the contract:
 * "coupling_tile": 0 (default, automatic), 8 or 16 modes per side of a workgroup's tile of pairs in the vector walk (the
 * parity build's kernel: never changes a bit there); FMA build: "coupling_matrix" 1 (default) = the matrix-pipe walk, 0 =
 * the vector walk; "coupling_matrix_tile" 0 (automatic), 16 or 32; "coupling_split" 0 (automatic) or 1..64 = pieces the grid
 * is cut into per tile (their sums are added in ascending order by a second launch; 1 = no split).  The fields, norms and
 * mode identification of storm_postprocess_batch are always computed in the reference's arithmetic.
 * storm_sweep_coupling_matrices_batch follows the sweep's own "arithmetic" option. */
int storm_postproc_set_option(storm_postproc* post, const char* key, int64_t value);

/* Rotating1DPostprocessing::new(freq[k], eigenvectors[k], ell, m, model) for k < n.  Host pointers:
 * eigenvectors [n][n_pts*4] as returned by storm_eigenvector_batch; fields_out [n][STORM_POST_FIELDS][n_pts]
 * (may be NULL: ids only); ids_out [n]; nodes_out [n][node_cap] (may be NULL) receives the first
 * min(n_nodes, node_cap) radial node positions of each mode in ascending grid order. */
int storm_postprocess_batch(storm_postproc* post, uint64_t ell, int64_t m, const double* freq, size_t n,
                            const double* eigenvectors, double* fields_out, storm_mode_id* ids_out,
                            double* nodes_out, size_t node_cap);

/* The `eigenvector` closure and the post-processing fused: what StormState::scan's per-mode loop followed by
 * `post-process` computes (src/main.rs:822-831,878-899), with the eigenvectors staying in HBM between the two.
 * ell and m are the solver's; the solver grid must be the model grid (n_int + 1 == n_pts).  det_out, vec_out,
 * fields_out, nodes_out may each be NULL (then nothing is copied back for them). */
int storm_modes_batch(storm_solver* solver, storm_postproc* post, const double* freq, size_t n, double* det_out,
                      double* vec_out, double* fields_out, storm_mode_id* ids_out, double* nodes_out, size_t node_cap);

/* ---- Mode coupling of the deformed star (`perturb-deformed`, src/main.rs:901-940) ----
 *
 * PerturbedMetric (src/model.rs:104-123) as produced by perturb_structure (src/perturbed.rs:324-460, stays on the
 * host): six arrays of n_pts values on the post-processing handle's grid.  Host pointers. */
typedef struct {
  const double *alpha, *dalpha, *ddalpha, *beta, *dbeta, *ddbeta;
} storm_metric_desc;

/* Order of the matrices in `out`. */
typedef enum {
  STORM_COUPLING_D_SQUARED = 0, STORM_COUPLING_D_LINEAR, STORM_COUPLING_D_ZERO, STORM_COUPLING_R_LINEAR,
  STORM_COUPLING_R_ZERO, STORM_COUPLING_L_ZERO, STORM_COUPLING_MATRICES = 6
} storm_coupling_matrix;

/* Replaces the (left, right, radius) triple loop of perturb_deformed (src/perturbed.rs:110-263): the six n x n
 * coupling matrices of n modes with azimuthal order m.  ell[k], freq[k] = ModeToPerturb::{ell, freq}
 * (src/perturbed.rs:26-33); fields = the modes' post-processing results, [n][STORM_POST_FIELDS][n_pts] exactly as
 * storm_postprocess_batch / storm_modes_batch write them.  out is [STORM_COUPLING_MATRICES][n*n], each matrix
 * column-major like nalgebra's DMatrix: element (left, right) at left + right*n.  The generalised eigenproblem built
 * from them (src/perturbed.rs:265-320, src/linalg/qz.rs) stays with the caller.  STORM_ERR_NON_FINITE where the
 * reference's assert!(.. .is_finite()) would fire (src/perturbed.rs:165-170,188-199,253-260). */
int storm_coupling_matrices(storm_postproc* post, size_t n, const uint64_t* ell, const double* freq,
                            const double* fields, int64_t m, const storm_metric_desc* metric, double* out);

/* Device time in ms (HIP events on the launch stream) of the coupling kernel of the last storm_coupling_matrices call
 * on this handle; measurement aid, no counterpart in the reference. */
double storm_postproc_last_kernel_ms(const storm_postproc* post);

/* ---- Several systems per launch: rotation sweeps ----
 *
 * Replaces the reference's one-solver-per-`scan`-line loop for batch inputs that sweep many (ell, m, rotation rate)
 * systems with scans of only 100-300 frequencies each (scripts/generate-batch-input-deformed.py:19-35; each `scan`
 * builds get_solvers_inner_explicit + ErasedSolver::scan_and_optimize, src/main.rs:815-831,
 * src/dynamic_interface.rs:116-160,193-221).  One such scan cannot fill a GPU, so a storm_sweep runs S systems in
 * lock-step: one discriminant launch over every (system, frequency) pair, the refinement rounds of all systems
 * merged, one eigenvector launch over all modes.  Results are bit-identical to S separate storm_solver handles.
 *
 * All systems share the difference scheme and n_int (the grid); ell, m and every coefficient array may differ.
 * Per-system lists are passed concatenated with CSR-style offsets: system u owns entries
 * [offsets[u], offsets[u+1]) of the arrays, offsets has n_sys + 1 entries, offsets[0] = 0.  Host pointers. */
typedef struct storm_sweep storm_sweep;
int storm_sweep_create(const storm_system_desc* descs, size_t n_sys, storm_sweep** out);
void storm_sweep_destroy(storm_sweep* sweep);
size_t storm_sweep_size(const storm_sweep* sweep);  /* n_sys */
size_t storm_sweep_n_int(const storm_sweep* sweep);

/* storm_det_batch for every system: det_out[i] = det_u(freq[i]) for i in system u's range. */
int storm_sweep_det(storm_sweep* sweep, const size_t* offsets, const double* freq, double* det_out);

/* storm_scan_and_optimize for every system.  out receives the brackets of system u at
 * [out_offsets[u], out_offsets[u+1]) (out_offsets: n_sys + 1 entries, written by the call; if the total exceeds cap the
 * call fails with STORM_ERR_CAPACITY after filling out_offsets, so the caller can size `out` and call again).
 * had_invalid (may be NULL) has n_sys entries; det_out (may be NULL) is laid out like freq; index_out (may be NULL,
 * cap entries) holds each bracket's lower grid index inside its own system's scan. */
int storm_sweep_scan_and_optimize(storm_sweep* sweep, const size_t* freq_offsets, const double* freq, uint64_t ulp,
                                  storm_bracket* out, size_t cap, size_t* out_offsets, int* had_invalid,
                                  double* det_out, uint64_t* index_out);

/* storm_eigenvector_batch for every system: vec_out [total modes][(n_int+1)*4] in the order of freq. */
int storm_sweep_eigenvectors(storm_sweep* sweep, const size_t* offsets, const double* freq, double* det_out,
                             double* vec_out);

/* storm_modes_batch for every system; posts[u] is the post-processing handle (model) of system u (handles may
 * repeat; entries of systems without modes may be NULL).  det_out, vec_out, fields_out, nodes_out may each be NULL. */
int storm_sweep_modes(storm_sweep* sweep, storm_postproc* const* posts, const size_t* offsets, const double* freq,
                      double* det_out, double* vec_out, double* fields_out, storm_mode_id* ids_out, double* nodes_out,
                      size_t node_cap);

/* post-process followed by perturb-deformed (src/main.rs:878-940) without the fields leaving HBM.
 * storm_sweep_modes_resident = storm_sweep_modes that returns ids (and nodes) only and keeps the 13 post-processing
 * fields of every mode of the call on the device (they must fit the staging cap in one pass, else
 * STORM_ERR_CAPACITY); a later storm_sweep_modes* call replaces them.
 * storm_sweep_coupling_matrices = storm_coupling_matrices on n of those modes: mode_index[k] is the position of
 * mode k in the freq array of the resident call (the modes of one `output` block with sol.m == m, in scan order,
 * src/main.rs:922-934); ell[k], freq[k], metric and out as in storm_coupling_matrices; post = the model the
 * integrals run on (the input evaluated on the grid when perturb-deformed runs). */
int storm_sweep_modes_resident(storm_sweep* sweep, storm_postproc* const* posts, const size_t* offsets,
                               const double* freq, storm_mode_id* ids_out, double* nodes_out, size_t node_cap);
int storm_sweep_coupling_matrices(storm_sweep* sweep, storm_postproc* post, size_t n, const size_t* mode_index,
                                  const uint64_t* ell, const double* freq, int64_t m,
                                  const storm_metric_desc* metric, double* out);

/* The same for many blocks in ONE launch (a group of rotation rates): problem j owns entries
 * [offsets[j], offsets[j+1]) of mode_index / ell / freq and has its own handle posts[j], azimuthal order m[j], metric
 * metrics[j] and result buffer out[j] (STORM_COUPLING_MATRICES * n_j * n_j doubles).  offsets has n_problems + 1
 * entries.  Bit-identical to n_problems calls of storm_sweep_coupling_matrices. */
int storm_sweep_coupling_matrices_batch(storm_sweep* sweep, size_t n_problems, storm_postproc* const* posts,
                                        const size_t* offsets, const size_t* mode_index, const uint64_t* ell,
                                        const double* freq, const int64_t* m, const storm_metric_desc* metrics,
                                        double* const* out);

/* ---- One scan over several GPUs of a node, inside the library ----
 *
 * The reference's rayon map over the scan frequencies (src/dynamic_interface.rs:125-127) is independent per frequency,
 * so a storm_multi runs it on several devices from ONE process: one storm_solver per listed device (the coefficient
 * planes are replicated), one host thread per device inside a call.  storm_multi_scan_and_optimize = ErasedSolver::
 * scan_and_optimize (src/dynamic_interface.rs:116-160): device k evaluates the contiguous slice
 * [k*ceil(n/n_dev), (k+1)*ceil(n/n_dev)) plus ONE halo point, so every adjacent pair is owned by exactly one device; the
 * sign-swap brackets (a few KB) are gathered in host memory in ascending order -- the reference's single global
 * filter_sign_swap order (src/bracket.rs:236-250); no collective library is involved inside one process.  Refinement
 * (src/dynamic_interface.rs:139-151): by default every device refines the brackets of its own slice in the call that
 * scanned it (one round of host threads, nothing between scan and refinement: refinement rounds are latency-bound
 * whatever the count); option "deal" = 1 deals the gathered brackets out cyclically by global bracket index instead, so
 * that a slice dense in modes (hundreds) does not serialise on the device that scanned it.
 * storm_multi_eigenvector_batch deals the modes cyclically.
 * Results are bit-identical to one storm_solver on one device either way (every evaluation runs the same kernels).
 * `devices` may list a device more than once (it then runs several slices; used to test the logic on one GPU).  Every
 * other option is forwarded to every solver.  One process per GPU over RCCL instead: storm_amd/distributed.py (same
 * split, the same two protocols: strong_scan_and_optimize(deal=False | True)). */
typedef struct storm_multi storm_multi;
int storm_multi_create(const storm_system_desc* desc, const int* devices, size_t n_dev, storm_multi** out);
void storm_multi_destroy(storm_multi* multi);
size_t storm_multi_devices(const storm_multi* multi);
int storm_multi_set_option(storm_multi* multi, const char* key, int64_t value);
int storm_multi_scan_and_optimize(storm_multi* multi, const double* freq, size_t n, uint64_t ulp, storm_bracket* out,
                                  size_t cap, size_t* n_out, int* had_invalid, double* det_out, uint64_t* index_out);
int storm_multi_eigenvector_batch(storm_multi* multi, const double* freq, size_t n, double* det_out, double* vec_out);
/* The split itself (pure index arithmetic, no device): the grid points [*lo, *hi) that part k of n_parts owns; the
 * part evaluates [*lo, min(*hi + 1, n)). */
void storm_shard_bounds(size_t n, size_t n_parts, size_t k, size_t* lo, size_t* hi);

/* ---- Memory cache (no counterpart in the reference) ----
 * Handles draw their device buffers, pinned staging buffers and streams from a process-wide cache and return them to
 * it on destroy, because hipFree / hipStreamDestroy synchronise the whole device and would serialise the concurrent
 * small scans of a rotation sweep (one handle per (ell, m, rotation rate) system), and because fresh memory can cost
 * ~30 ms per GB when the driver has to clear it first.  Default: every block is carved out of
 * slabs the library allocates as the need grows (4 GiB doubling to 32 GiB for blocks of 64 MiB and more, smaller ones for
 * smaller blocks; pinned host memory likewise) and keeps until storm_memory_cache_trim, below three quarters of the
 * device's memory in all.  storm_memory_cache_limit switches to separately allocated blocks of which the given number
 * of bytes may stay cached.  storm_memory_cache_stats counts what a trim would give back (listed blocks, slabs nothing is carved from).
 * A caller that used storm_det_batch_device with its own stream must synchronise that stream before destroying the
 * handle. */
int storm_memory_cache_trim(void);                                            /* give everything cached back to HIP */
int storm_memory_cache_limit(size_t device_bytes, size_t pinned_host_bytes);  /* what may stay cached from now on */
void storm_memory_cache_stats(size_t* device_bytes, size_t* pinned_host_bytes); /* currently cached (either may be NULL) */

/* Precision::Relative(p).as_ulp(), src/bracket.rs:30-40. */
uint64_t storm_precision_relative_to_ulp(double relative_precision);

/* Work counters since creation (or the last reset): discriminant evaluations (scan + refinement
 * + single calls), of which refinement; modes whose eigenvector was computed; kernel launches; modes that the
 * segment-parallel eigenvector path ("segments") sent through the sequential elimination. */
typedef struct {
  uint64_t det_evals, refine_evals, refine_rounds, eigenvectors, launches, segment_fallbacks;
} storm_stats;
int storm_solver_stats(storm_solver* solver, storm_stats* out, int reset);
int storm_multi_stats(storm_multi* multi, storm_stats* out, int reset);   /* summed over the devices (refine_rounds: max) */

/* Per-kernel device time, measured with HIP events recorded on the launch stream around every
 * launch while option "profile" is 1.  ms_out / launches_out have STORM_KERNEL_KINDS entries. */
enum { STORM_KERNEL_DET_FUSED = 0, STORM_KERNEL_BUILD_STEPS = 1, STORM_KERNEL_CHAIN = 2, STORM_KERNEL_EIGEN = 3,
       STORM_KERNEL_DET_QUAD = 4, STORM_KERNEL_DET_HEX = 5, STORM_KERNEL_POSTPROCESS = 6, STORM_KERNEL_SEGMENTS = 7,
       STORM_KERNEL_KINDS = 8 };
int storm_solver_kernel_times(storm_solver* solver, double* ms_out, uint64_t* launches_out, int reset);

/* Options.  Every "path" is bit-identical; only "arithmetic" = 1 changes results (to rounding).
 *   "arithmetic": 0 (default) parity build: separate multiply and add in the reference's order, results
 *                 bit-identical to the reference; 1 fast build: the same mathematics with FMA contraction
 *                 (v_fma_f64) and re-associated where that saves instructions (adjugate through 2x2 minors, stage
 *                 systems by LU, reciprocals by Newton steps, segment-parallel products, the running 4x4 product on
 *                 v_mfma_f64_4x4x4: DESIGN.md section 5), ~3x
 *                 the parity build's throughput.  Fast-mode contract (tests/test_hip_fast_mode.py):
 *                 |det_fast - det_ref| <= 1e-9 * max|det_ref| over the neighbouring grid points, identical
 *                 sign pattern / bracket indices on the BASELINE configs, roots within 1e-10 relative.
 *   "path": 0 auto, 1 force lane-per-frequency fused kernel, 2 force build-steps + ordered-product kernels,
 *           3 force the 4-lanes-per-frequency fused kernel, 4 force the 16/32-lanes-per-frequency fused kernel
 *   "hex_lanes": lanes per frequency of the path-4 kernel: 0 (default): a large batch goes out as ONE launch of column
 *                owners (4 lanes per frequency) for most of it and 16-lane groups for the rest (det_mix), a smaller one
 *                at 16 or 32 lanes, whichever packs the GPU's wave slots better; 4, 16, 32 force one (same bits in the
 *                parity build); -1..-99 (tuning knob) force det_mix with that percentage of column owners
 *   "hex_rounds": the path-4 kernel sends a batch that is larger than the GPU's wave slots in pieces of this many
 *                 wave rounds (default 1), so that kernels of other handles get dispatched between the pieces
 *                 instead of after the whole batch; 0 = one launch
 *   "eigen_layout": eigenvector kernels: 0 (default) one wave per mode below 1920 modes per launch and one lane per
 *                   mode from there on, 1 always wave per mode, 2 always lane per mode (same bits either way)
 *   "hex_min_n" / "quad_min_n": batch sizes from which "path" 0 picks those kernels (below: build + ordered product)
 *   "steps_bytes_max": cap of the step-matrix staging buffer in bytes (default 4 GiB)
 *   "fused_min_n": batch size from which "path" 0 picks the fused kernel
 *   "segments": FMA build only ("arithmetic" = 1; ignored in the parity build, whose association is the reference's):
 *               batches too small to fill the GPU (bracket-refinement rounds, single det calls, the eigenvectors of a
 *               scan's modes) cut the interval axis into segments evaluated in parallel.  Discriminants: the ordered
 *               product of the segments' transfer matrices (same algorithm, products associated by segment).
 *               Eigenvectors: the reference's pivoted elimination on the segment-boundary unknowns (multiple shooting),
 *               then y_{i+1} = L_i y_i inside every segment; a mode whose segment transfer matrices exceed 2^24 in
 *               magnitude (strongly evanescent zone: forward propagation would amplify rounding) is sent through the
 *               sequential elimination instead ("seg_growth_log2" moves that bound, default 24).  0 (default) automatic
 *               segment length, -1 off, > 0 forced length.
 *               Tolerances as for "arithmetic" = 1; in this mode det_out of the eigenvector calls is the discriminant of
 *               the segment-boundary system (agrees with the reference's in magnitude to rounding, not in sign).
 *   "scaled_scan": 1 = storm_scan_and_optimize, storm_scan_brackets and storm_optimize_brackets (and storm_multi_*, which
 *                  is built on them) bracket on the scaled discriminant of storm_det_scaled_batch: the sign-swap filter
 *                  takes the mantissa's sign, and every bracket is refined on values brought to one power of two of its
 *                  own (InverseQuadratic::optimize, src/bracket.rs:148-223, is invariant under a common scaling of its
 *                  three values), so a scan finds its modes where the reference's raw product has left the range of f64
 *                  (src/solver.rs:504,531,541; src/dynamic_interface.rs:130-137 drops such pairs and reports invalid
 *                  numbers).  Where the raw product is finite: the same bracket indices, roots equal to rounding
 *                  (tests/test_hip_scaled_det.py).  det_out, lo_f and hi_f are mantissa * 2^exponent saturated to
 *                  +-inf / +-0 -- the raw value where it exists; storm_optimize_brackets then takes only the SIGNS of
 *                  lo_f / hi_f and evaluates both ends again.  0 (default): the reference's behaviour.  FMA build: the
 *                  values come from the segment path (transfer matrices of 128-interval segments from the fast kernels,
 *                  then the scaled chain over the segments; a frequency whose segment overflows is redone through the
 *                  staged path): a 10^5 x 10^4 scan 24 ms against 17.5 ms raw.  Parity build: through the staged step
 *                  matrices, about twice the raw scan's time.
 *   "profile": 1 records HIP events around every kernel launch (see storm_solver_kernel_times) */
int storm_solver_set_option(storm_solver* solver, const char* key, int64_t value);

/* Counters and options of a sweep ("arithmetic", "hex_lanes", "hex_rounds", "eigen_layout", "steps_bytes_max",
 * "seg_growth_log2", "coupling_tile" as for a solver / post-processing handle; the staging cap defaults to 16 GiB for a
 * sweep).  "segments": 0 (default) = in the FMA build every eigenvector pass and every discriminant batch of at most
 * 3072 frequencies per class (the refinement rounds) goes through segment transfer matrices, with the tolerances of
 * "arithmetic" = 1 (a pass in which a transfer matrix outgrows the guard is redone sequentially and counted in
 * segment_fallbacks); -1 = never.  The parity build never takes that path.  A sweep runs
 * InverseQuadratic::optimize (src/bracket.rs:148-223) as a host loop over the brackets still open (one batched device
 * evaluation per round); "coupling_matrix" / "coupling_matrix_tile" as for a post-processing handle. */
int storm_sweep_stats(storm_sweep* sweep, storm_stats* out, int reset);
int storm_sweep_set_option(storm_sweep* sweep, const char* key, int64_t value);

size_t storm_solver_n_int(const storm_solver* solver);

const char* storm_last_error(void);
const char* storm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STORM_HIP_H */
