// storm_hip.hip -- host side of libstorm_hip: the C ABI of include/storm_hip.h.
//
// Owns the device-resident coefficient planes, picks a kernel path per batch, and runs the
// bracket refinement as lock-step rounds over all brackets (each round = one batched
// discriminant evaluation on the GPU).  No CPU fallback: without a HIP device every entry
// point fails with STORM_ERR_HIP.
#include "../../include/storm_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common_kernels.hpp"
#include "hostlogic/brackets.hpp"
#include "hostlogic/failure.hpp"
#include "hostlogic/memory_cache.hpp"
#include "hostlogic/multi.hpp"
#include "postprocess.hpp"
#include "types.hpp"

#define STORM_DECL_SCHEME(ns, n)                                  \
  namespace ns {                                                  \
  hipError_t launch_fused_scheme_##n(const storm::LaunchArgs&);   \
  hipError_t launch_quad_scheme_##n(const storm::LaunchArgs&);    \
  hipError_t launch_hex_scheme_##n(const storm::LaunchArgs&);     \
  hipError_t launch_build_scheme_##n(const storm::LaunchArgs&);   \
  hipError_t launch_seg_product_scheme_##n(const storm::LaunchArgs&); \
  hipError_t launch_seg_fill_scheme_##n(const storm::LaunchArgs&);    \
  hipError_t launch_hexinfo_scheme_##n(const storm::LaunchArgs&);     \
  }
#define STORM_DECL_ALL(ns)                                                                          \
  STORM_DECL_SCHEME(ns, 0) STORM_DECL_SCHEME(ns, 1) STORM_DECL_SCHEME(ns, 2) STORM_DECL_SCHEME(ns, 3) \
  STORM_DECL_SCHEME(ns, 4) STORM_DECL_SCHEME(ns, 5) STORM_DECL_SCHEME(ns, 6) STORM_DECL_SCHEME(ns, 7)
STORM_DECL_ALL(storm)
STORM_DECL_ALL(storm_fast)

namespace storm {

// [arithmetic: 0 parity, 1 fast][scheme]
#define STORM_TABLE(kind)                                                                                      \
  {{storm::launch_##kind##_scheme_0, storm::launch_##kind##_scheme_1, storm::launch_##kind##_scheme_2,           \
    storm::launch_##kind##_scheme_3, storm::launch_##kind##_scheme_4, storm::launch_##kind##_scheme_5,           \
    storm::launch_##kind##_scheme_6, storm::launch_##kind##_scheme_7},                                           \
   {storm_fast::launch_##kind##_scheme_0, storm_fast::launch_##kind##_scheme_1, storm_fast::launch_##kind##_scheme_2, \
    storm_fast::launch_##kind##_scheme_3, storm_fast::launch_##kind##_scheme_4, storm_fast::launch_##kind##_scheme_5, \
    storm_fast::launch_##kind##_scheme_6, storm_fast::launch_##kind##_scheme_7}}
static const launch_fn g_launch_fused[2][kNumSchemes] = STORM_TABLE(fused);
static const launch_fn g_launch_quad[2][kNumSchemes] = STORM_TABLE(quad);
static const launch_fn g_launch_hex[2][kNumSchemes] = STORM_TABLE(hex);
static const launch_fn g_launch_build[2][kNumSchemes] = STORM_TABLE(build);
static const launch_fn g_launch_seg_product[2][kNumSchemes] = STORM_TABLE(seg_product);
static const launch_fn g_launch_seg_fill[2][kNumSchemes] = STORM_TABLE(seg_fill);
static const launch_fn g_hexinfo[2][kNumSchemes] = STORM_TABLE(hexinfo);

static thread_local std::string g_error;

// (Failure / fail(): hostlogic/failure.hpp)
static void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) fail(STORM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// The process-wide cache of device blocks, pinned host blocks and streams (hostlogic/memory_cache.hpp: slab arenas, size
// classes, footprint limits -- host-only bookkeeping, also compiled under the CPU sanitizers) over the HIP runtime.
struct HipBackend {
  using Stream = hipStream_t;
  static void* alloc(bool pinned, size_t bytes) {
    void* p = nullptr;
    const hipError_t e = pinned ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
    if (e == hipSuccess) return p;
    (void)hipGetLastError();  // clear HIP's sticky error: the launch wrappers return hipGetLastError()
    return nullptr;
  }
  static void release(bool pinned, void* p) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
  [[noreturn]] static void alloc_failed(bool pinned, size_t bytes) {
    fail(STORM_ERR_HIP, std::string(pinned ? "hipHostMalloc" : "hipMalloc") + " of " + std::to_string(bytes) + " bytes failed");
  }
  static size_t device_total(int dev) {
    size_t bytes = 0;
    if (hipDeviceTotalMem(&bytes, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return bytes;
  }
  static int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
  static void set_device(int dev) { (void)hipSetDevice(dev); }
  static Stream stream_create() {
    hipStream_t st = nullptr;
    hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    return st;
  }
  static void stream_destroy(Stream st) { (void)hipStreamDestroy(st); }
};
using MemoryCache = BasicMemoryCache<HipBackend>;

// Growable device buffer (blocks come from / go back to the MemoryCache).
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;       // elements usable
  size_t granted = 0;   // bytes of the block
  int dev = 0;
  // `st`: the stream whose work may still be using the current block; it is synchronised before the block is recycled
  void reserve(size_t n, hipStream_t st = nullptr) {
    if (n <= cap) return;
    if (p) {
      hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
      MemoryCache::get().give(false, dev, p, granted);
    }
    p = nullptr; cap = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    p = static_cast<T*>(MemoryCache::get().take(false, dev, n * sizeof(T), granted));
    cap = granted / sizeof(T);
  }
  void release() { if (p) MemoryCache::get().give(false, dev, p, granted); p = nullptr; cap = 0; }
};

// Growable pinned host buffer (staging for large device-to-host results).
struct PinnedBuf {
  double* p = nullptr;
  size_t cap = 0, granted = 0;
  void reserve(size_t n) {  // callers have synchronised the stream that last wrote it
    if (n <= cap) return;
    if (p) MemoryCache::get().give(true, 0, p, granted);
    p = nullptr; cap = 0;
    p = static_cast<double*>(MemoryCache::get().take(true, 0, n * sizeof(double), granted));
    cap = granted / sizeof(double);
  }
  void release() { if (p) MemoryCache::get().give(true, 0, p, granted); p = nullptr; cap = 0; }
};

}  // namespace storm

using namespace storm;

// ---- scratch of the allocation-free entry point (storm_det_batch_device_noalloc) ----
// A file-scope device array instead of a DevBuf: callers under HIP graph capture or in loops that must not reach
// hipMalloc get the two scratch areas the discriminant kernels need -- the singular-stage flag word they atomicOr into
// and the segment transfer matrices of the segment-parallel path -- without any allocation.  Every GPU the library runs
// on holds its own copy until the process ends; the array's name is only a host-side handle, the device address comes
// from hipGetSymbolAddress (once per device, at the first storm_solver_create on that device) and is what the kernels
// receive as an argument.  All calls on one GPU share it: overlapping allocation-free calls on one GPU race.
constexpr size_t kScratchFlagBytes = 256;                    // [0, 256): flag words
constexpr size_t kScratchPartialBytes = size_t(16) << 20;    // then: segment transfer matrices, 128 B each
constexpr size_t kNoallocMaxN = size_t(1) << 24;             // frequencies per call (STORM_NOALLOC_MAX_N)
static_assert(kScratchPartialBytes == STORM_NOALLOC_SCRATCH_BYTES - 256 && kNoallocMaxN == STORM_NOALLOC_MAX_N, "storm_hip.h");
// (16 MiB: the remainder of a wave round, at most a quarter of 4 x wave slots frequencies x 32 segments x 128 B.)
// (no __attribute__((used)): it would keep the symbol local in the device code object, where hipGetSymbolAddress cannot
// find it; the host-side use below is what keeps the array alive and makes hipcc export it)
static __device__ __align__(256) unsigned char scratch[kScratchFlagBytes + kScratchPartialBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
// The current device's copy of `scratch`.
unsigned char* scratch_of_current_device() {
  int dev = 0;
  hip_check(hipGetDevice(&dev), "hipGetDevice");
  if (dev < 0 || dev >= kMaxDevices) fail(STORM_ERR_HIP, "device index out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    hip_check(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
    if (!p) fail(STORM_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    g_scratch_ptr[dev] = static_cast<unsigned char*>(p);
  }
  return g_scratch_ptr[dev];
}
}  // namespace

struct storm_solver {
  unsigned char* scratch = nullptr;  // this device's copy of the allocation-free entry point's scratch array
  int device = 0;
  int scheme = 0;
  bool ell0 = true, uniform = true;
  uint64_t ell = 0;  // Rotating1D::new(ell, m)
  int64_t m = 0;
  SystemDev sys{};
  DevBuf<double> tab;
  hipStream_t stream = nullptr;
  DevBuf<double> d_freq, d_det, d_steps, d_upper, d_vec;
  DevBuf<int> d_swaps, d_flags;
  // segment-parallel path (FMA build): segment transfer matrices, boundary values of the eigenvectors, max |T_s| per mode,
  // and the staging of modes that are sent through the sequential elimination after all
  DevBuf<double> d_partials, d_nodes, d_vec_fb, d_det_fb, d_freq_fb;
  DevBuf<unsigned long long> d_growth;
  PinnedBuf h_vec, h_flags, h_growth;
  // bracketing on the device (common_kernels.hip): bracket states, their grid indices, block counts / offsets of the
  // ordered compaction, proposed abscissae and their discriminants, live counters per round, error flags
  DevBuf<BracketDev> d_br;
  DevBuf<unsigned long long> d_bidx;
  DevBuf<unsigned> d_bcount, d_live;
  DevBuf<double> d_bx, d_bf;
  DevBuf<int> d_bflags;
  PinnedBuf h_br;
  int device_brackets = 1;                       // 1 (default): sign-swap filter and refinement state machine on the device
  int scaled_scan = 0;                           // 1: scans bracket on the scaled discriminant (mantissa, exponent): option "scaled_scan"
  int segments = 0;                              // 0 auto (FMA build only), -1 off, > 0 forced segment length
  double seg_growth_max = 16777216.0;            // 2^24: larger segment transfer matrices send a mode to the sequential path
  const double* h_freq_last = nullptr;           // the caller's frequencies of the eigenvector chunk in flight
  size_t classic_chunk = 1;                      // modes per pass of the sequential kernels (set by eigen_reserve)
  int arith = 0;                                 // 0 parity (un-fused), 1 fast (FMA contraction)
  int path = 0;                                  // 0 auto, 1 fused, 2 split, 3 quad, 4 hex
  size_t steps_bytes_max = size_t(4) << 30;
  size_t fused_min_n = size_t(1) << 62;          // auto: lane-per-frequency kernel from this batch size on (off: the
                                                 // 16-lane kernel is faster at every size measured)
  size_t quad_min_n = size_t(1) << 62;           // auto: quad kernel from this batch size on (off by default)
  size_t hex_min_n = 512;                        // auto: 16/32-lanes-per-frequency kernel from this batch size on
  int hex_lanes = 0;                             // lanes per frequency of that kernel: 0 = by wave-slot packing, 16, 32
  int hex_rounds = 1;                            // wave rounds per launch of that kernel, 0 = the whole batch in one launch
  int eigen_layout = 0;                          // eigenvector kernels: 0 = by mode count, 1 wave per mode, 2 lane per mode
  storm_stats stats{};
  // optional per-kernel timing with HIP events on the launch stream (option "profile")
  hipEvent_t sync_marker = nullptr;
  // `ready`: recorded on `stream` behind the create-time upload of the planes and the flag memset; a launch on a
  // caller's stream (storm_det_batch_device) waits for it.  `steps_done`: recorded behind the last kernel that read
  // d_steps, on `steps_user`; another stream waits for it before it overwrites (or the host before it recycles) d_steps.
  hipEvent_t ready = nullptr, steps_done = nullptr;
  hipStream_t steps_user = nullptr;
  bool profile = false;
  struct Timed { int kind; hipEvent_t start, stop; };
  std::vector<Timed> pending;
  std::vector<hipEvent_t> event_pool;
  double kernel_ms[STORM_KERNEL_KINDS] = {};
  uint64_t kernel_launches[STORM_KERNEL_KINDS] = {};
  std::mutex mu;
};

// Model arrays of the post-processing in HBM + per-batch scratch.
struct storm_postproc {
  int device = 0;
  PostModelDev model{};
  DevBuf<double> tab;
  // The model planes wait in pinned memory and are uploaded by the first call that needs them, on the stream that
  // call runs on (for storm_modes_batch: the solver's), so creating a handle costs no device work and no wait behind
  // other scans' kernels; the handle's own stream is taken only if an entry point needs one.
  PinnedBuf h_model;
  size_t tab_elems = 0;
  bool uploaded = false;
  hipStream_t upload_stream = nullptr;
  hipEvent_t ready = nullptr, sync_marker = nullptr;
  hipStream_t stream = nullptr;
  DevBuf<double> d_freq, d_vec, d_terms, d_scale, d_fields, d_nodes;
  DevBuf<long long> d_ids;
  DevBuf<double> d_weights, d_consts, d_out, d_partial;  // coupling integrals (d_partial: the pieces of a split walk)
  DevBuf<int> d_ell_index, d_cflags, d_order;
  DevBuf<long long> d_gather;                 // rows to pick out of a sweep's resident fields
  hipEvent_t ev0 = nullptr, ev1 = nullptr;    // around the coupling kernel
  std::vector<double> h_r, h_rho, h_rot, h_trap;  // host copies of the planes the coupling weights are built from
  PinnedBuf h_stage;
  std::vector<long long> h_ids;
  size_t bytes_max = size_t(1) << 30;  // cap of the per-batch scratch (modes are processed in chunks)
  double last_coupling_ms = 0.0;       // HIP-event duration of the last coupling kernel
  int arith = 0;                       // coupling integrals: 0 = the reference's association, 1 = FMA build (option "arithmetic")
  int cp_tile = 0;                     // modes per side of a workgroup's tile of pairs: 0 = automatic, 8, 16 (option "coupling_tile")
  int cp_matrix = 1;                   // FMA build: 1 = the matrix-pipe walk over the ordered mode list, 0 = the vector walk (option "coupling_matrix")
  int cp_matrix_tile = 0;              // its tile: 0 = automatic, 16, 32 (option "coupling_matrix_tile")
  int cp_split = 0;                    // pieces of the grid per tile: 0 = automatic (plan_coupling_matrix), k = forced (option "coupling_split")
  std::mutex mu;
};

namespace {

struct DeviceGuard {
  int prev = 0;
  explicit DeviceGuard(int dev) {
    hip_check(hipGetDevice(&prev), "hipGetDevice");
    if (prev != dev) hip_check(hipSetDevice(dev), "hipSetDevice");
    target = dev;
  }
  ~DeviceGuard() { if (prev != target) (void)hipSetDevice(prev); }
  int target;
};

template <class F>
int guarded(F&& f) {
  try {
    f();
    return STORM_OK;
  } catch (const Failure& e) {
    g_error = e.msg;
    return e.code;
  } catch (const std::bad_alloc&) {
    g_error = "out of host memory";
    return STORM_ERR_HIP;
  } catch (const std::exception& e) {
    g_error = e.what();
    return STORM_ERR_INVALID_ARGUMENT;
  }
}

// Frequency-independent planes: Rotating1D::eval (reference src/system/adiabatic.rs:78-114) per
// collocation point, plus dx*c1, (dx*lambda)^2 and m*rot for add_frequency (:116-152).
// Host arithmetic here is un-fused (this file is built with -ffp-contract=off).
size_t padded_stride(size_t n_int) { return (n_int + 63) / 64 * 64 + 64; }

size_t planes_elems(size_t n_int, size_t p) { return padded_stride(n_int) * p * kNumFields; }

// `t`: planes_elems(d.n_int, d.p) doubles.
void build_planes(const storm_system_desc& d, double* t, bool& uniform, double& rotm0) {
  const size_t n_int = d.n_int, p = d.p, stride = padded_stride(n_int), plane = stride * p;
  std::memset(t, 0, plane * kNumFields * sizeof(double));
  const double ell = double(d.ell), mm = double(d.m);
  const double lambda = ell * (ell + 1.0);
  uniform = true;
  rotm0 = n_int ? mm * d.rot[0] : 0.0;
  uint64_t rot_bits0;
  std::memcpy(&rot_bits0, &rotm0, 8);
  for (size_t i = 0; i < n_int; ++i)
    for (size_t j = 0; j < p; ++j) {
      const size_t src = i * p + j, dst = j * stride + i;
      const double dx = d.delta_over_x[src], u = d.u[src], vg = d.v_gamma[src], as = d.a_star[src];
      t[F_M00 * plane + dst] = dx * (vg - 1.0 - ell);
      t[F_M01 * plane + dst] = -dx * vg;
      t[F_M10 * plane + dst] = -dx * as;
      t[F_M11 * plane + dst] = dx * (as - u + 3.0 - ell);
      t[F_M22 * plane + dst] = dx * (3.0 - u - ell);
      t[F_M30 * plane + dst] = dx * (u * as);
      t[F_M31 * plane + dst] = dx * (u * vg);
      const double dl = dx * lambda;
      t[F_M32 * plane + dst] = dl;
      t[F_M33 * plane + dst] = dx * (-u + 2.0 - ell);
      t[F_D * plane + dst] = dx;
      t[F_DC1 * plane + dst] = dx * d.c1[src];
      t[F_DL2 * plane + dst] = dl * dl;
      const double rotm = mm * d.rot[src];
      t[F_ROTM * plane + dst] = rotm;
      uint64_t b;
      std::memcpy(&b, &rotm, 8);
      if (b != rot_bits0) uniform = false;
    }
  // padding: repeat the last interval (kernels may evaluate, but never apply, up to 31 intervals past the end)
  if (n_int)
    for (size_t fld = 0; fld < size_t(kNumFields); ++fld)
      for (size_t j = 0; j < p; ++j) {
        double* row = t + fld * plane + j * stride;
        for (size_t i = n_int; i < stride; ++i) row[i] = row[n_int - 1];
      }
}

// The interval-major copy behind the planes (types.hpp, aos_pitch), filled on the device by launch_aos_copy.
size_t aos_elems(size_t n_int, size_t p) { return padded_stride(n_int) * size_t(aos_pitch(int(p))); }

void count_launch(storm_solver* s, uint64_t k = 1) { s->stats.launches += k; }

// The pinned staging block also carried the planes to the device in storm_solver_create: before it can go back to
// the cache that upload (or an earlier result copy) must have finished.
void stage_reserve(storm_solver* s, size_t n) {
  if (n > s->h_vec.cap && s->h_vec.p) hip_check(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
  s->h_vec.reserve(n);
}

hipEvent_t take_event(storm_solver* s) {
  if (!s->event_pool.empty()) { hipEvent_t e = s->event_pool.back(); s->event_pool.pop_back(); return e; }
  hipEvent_t e;
  hip_check(hipEventCreate(&e), "hipEventCreate");
  return e;
}
// Brackets one launch with events on the stream it is launched on.
struct ScopedTimer {
  storm_solver* s; hipStream_t st; int kind; hipEvent_t a = nullptr, b = nullptr;
  ScopedTimer(storm_solver* s, hipStream_t st, int kind) : s(s), st(st), kind(kind) {
    if (!s->profile) return;
    a = take_event(s); b = take_event(s);
    hip_check(hipEventRecord(a, st), "hipEventRecord");
  }
  void stop() {
    if (!a) return;
    hip_check(hipEventRecord(b, st), "hipEventRecord");
    s->pending.push_back({kind, a, b});
    a = nullptr;
  }
};
// Call after the stream has been synchronised.
void resolve_timers(storm_solver* s) {
  for (auto& t : s->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      s->kernel_ms[t.kind] += ms;
      s->kernel_launches[t.kind] += 1;
    }
    s->event_pool.push_back(t.start);
    s->event_pool.push_back(t.stop);
  }
  s->pending.clear();
}

// Waits for everything submitted to `st` so far.  An event is recorded behind the last command first: with many
// handles in flight (one stream each, more streams than hardware queues) hipStreamSynchronize alone has to push its
// own marker through a queue that the hardware scheduler may have unmapped by then, which cost ~10 ms per call in
// tools/many_scans.cpp; with the event already queued behind the kernel the wait returns when the kernel does
// (64 scans on 16 threads: 140-175 ms -> 95 ms).
void sync_stream(hipStream_t st, hipEvent_t& marker) {
  if (!marker) hip_check(hipEventCreateWithFlags(&marker, hipEventDisableTiming), "hipEventCreate");
  hip_check(hipEventRecord(marker, st), "hipEventRecord");
  hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
}

void check_flags(storm_solver* s) {
  // the flag word lands in pinned memory (a copy to pageable memory takes the runtime's staged, host-synchronous path)
  s->h_flags.reserve(1);
  int* flags_ptr = reinterpret_cast<int*>(s->h_flags.p);
  hip_check(hipMemcpyAsync(flags_ptr, s->d_flags.p, sizeof(int), hipMemcpyDeviceToHost, s->stream), "flags D2H");
  sync_stream(s->stream, s->sync_marker);
  const int flags = *flags_ptr;
  resolve_timers(s);
  if (flags & 1) {
    hip_check(hipMemsetAsync(s->d_flags.p, 0, sizeof(int), s->stream), "flags reset");
    fail(STORM_ERR_SINGULAR, "singular collocation stage matrix (the reference would panic in try_inverse().unwrap())");
  }
}

// Is `p` host memory the device can DMA into directly (hipHostMalloc / hipHostRegister)?  Pageable pointers are unknown
// to the runtime: the query fails for them and the sticky error is cleared.
bool host_pointer_is_pinned(const void* p) {
  hipPointerAttribute_t attr{};
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeHost;
}

LaunchArgs make_args(storm_solver* s, const double* d_freq, size_t n, double* out, hipStream_t st, int* flags = nullptr) {
  LaunchArgs a;
  a.sys = s->sys; a.freq = d_freq; a.n = n; a.out = out; a.flags = flags ? flags : s->d_flags.p;
  a.ell0 = s->ell0; a.uniform = s->uniform; a.hex_lanes = s->hex_lanes; a.hex_rounds_per_launch = s->hex_rounds; a.stream = st;
  return a;
}

// d_steps is shared by every stream the handle is used with.  Before `st` writes it: wait (on the device) for the last
// reader if that ran on another stream; if the block is about to be recycled the host waits for that reader as well
// (DevBuf::reserve only synchronises the stream it is given).
void steps_acquire(storm_solver* s, hipStream_t st) {
  if (!s->steps_user || s->steps_user == st) return;
  hip_check(hipStreamWaitEvent(st, s->steps_done, 0), "hipStreamWaitEvent");
  hip_check(hipEventSynchronize(s->steps_done), "hipEventSynchronize");
}
void steps_release(storm_solver* s, hipStream_t st) {
  if (st == s->stream) { s->steps_user = nullptr; return; }  // the library stream is synchronised before every return
  if (!s->steps_done) hip_check(hipEventCreateWithFlags(&s->steps_done, hipEventDisableTiming), "hipEventCreate");
  hip_check(hipEventRecord(s->steps_done, st), "hipEventRecord");
  s->steps_user = st;
}

// Segment-parallel evaluation (scheme_kernels.hpp, "Segment-parallel kernels"): how the interval axis is cut for a batch
// of n frequencies.  n_seg == 0: not applicable.
struct SegPlan { int n_seg = 0, seg_len = 0; };
SegPlan plan_segments(const storm_solver* s, size_t n, size_t max_len, size_t min_segments = 0, bool serial_boundaries = false) {
  SegPlan p;
  const size_t n_int = s->sys.n_int;
  if (s->arith != 1 || s->segments < 0 || n == 0 || n_int < 64) return p;  // the parity build keeps the reference's association
  size_t len;
  if (s->segments > 0) {
    len = (size_t(s->segments) + 15) / 16 * 16;
  } else {
    const size_t target_groups = 4096;  // 16-lane groups in flight: one wave per SIMD
    const size_t want = std::max<size_t>({size_t(1), target_groups / n, min_segments});
    len = ((n_int + want - 1) / want + 15) / 16 * 16;
    if (serial_boundaries) {
      // Eigenvectors: the boundary system is eliminated by ONE wave per mode, ~0.9 us per boundary (eigen_forward +
      // eigen_backsub), while a segment costs seg_product + seg_fill ~0.33 us per interval of its length as long as the
      // groups fit the chip.  a n_int / len + b len is least at len = sqrt(a n_int / b) = sqrt(2.7 n_int): 160-176
      // intervals for 10^4.  Filling the chip (the rule above) asks for much shorter segments when there are few modes
      // (3 modes: 32 intervals, 316 boundaries, 0.29 ms; balanced: 0.13 ms), so take the longer of the two.
      const size_t balanced = (size_t(std::sqrt(2.7 * double(n_int))) + 15) / 16 * 16;
      len = std::max(len, balanced);
    }
    len = std::min(std::max<size_t>(len, 32), max_len);
  }
  size_t n_seg = (1 + (n_int - 1 + len - 1) / len + 3) / 4 * 4;  // the first interval is a segment of its own
  if (n * n_seg / 4 >= (size_t(1) << 31) || n_seg > (size_t(1) << 24)) return p;
  p.n_seg = int(n_seg); p.seg_len = int(len);
  return p;
}

// Discriminants through segment transfer matrices: one launch over every (frequency, segment) pair, then the ordered
// product + boundary elimination on n_seg "intervals".
void det_segments(storm_solver* s, const SegPlan& plan, const double* d_freq, size_t n, double* d_out, hipStream_t st, int* flags) {
  s->d_partials.reserve(n * size_t(plan.n_seg) * 16, st);
  LaunchArgs a = make_args(s, d_freq, n, s->d_partials.p, st, flags);
  a.n_seg = plan.n_seg; a.seg_len = plan.seg_len;
  hip_check(g_launch_seg_product[s->arith][s->scheme](a), "seg_product launch");
  SystemDev reduced = s->sys;
  reduced.n_int = size_t(plan.n_seg);
  hip_check(storm_fast::launch_chain(reduced, s->ell0, d_freq, n, s->d_partials.p, d_out, st), "chain launch");
}

// How many of n frequencies go to det_hex when the rest (less than a quarter wave round) is cheaper on the segment path
// (FMA build only; see det_device); `rest` receives the plan of the remainder (n_seg == 0: everything goes to det_hex).
size_t hex_whole_rounds(storm_solver* s, size_t n, SegPlan& rest) {
  rest = SegPlan{};
  const size_t n_int = s->sys.n_int;
  if (!(s->arith == 1 && s->path == 0 && s->hex_lanes == 0 && s->segments >= 0 && n_int >= 64)) return n;
  size_t info[2] = {0, 0};
  LaunchArgs q = make_args(s, nullptr, n, reinterpret_cast<double*>(info), s->stream, nullptr);
  hip_check(g_hexinfo[s->arith][s->scheme](q), "occupancy query");
  const size_t per_round = info[0];
  // (a remainder above a quarter round is cheaper inside det_hex: measured 20 000 frequencies 5.07 ms whole, 5.38 ms split)
  if (info[1] || !per_round || n <= per_round || n % per_round == 0 || n % per_round > per_round / 4) return n;
  // enough segments for the remainder's (frequency, segment) groups to fill ~85 % of the chip's 16-lane group slots (=
  // per_round): its launch then takes the time of one segment's chain, and the more segments the shorter that is (213
  // frequencies: 31 segments of 336 intervals 85 us, 51 of 208 ~50 us)
  const size_t rest_n = n % per_round;
  rest = plan_segments(s, rest_n, n_int, std::min<size_t>(255, std::max<size_t>(31, per_round * 85 / 100 / rest_n)));
  return rest.n_seg ? n - n % per_round : n;
}

// Core: discriminants of n device-resident frequencies into a device buffer, on stream st.
void det_device(storm_solver* s, const double* d_freq, size_t n, double* d_out, hipStream_t st, int* flags = nullptr) {
  if (n == 0) return;
  const size_t n_int = s->sys.n_int;
  const bool fused = s->path == 1 || (s->path == 0 && (n >= s->fused_min_n || n_int == 0));
  const bool quad = s->path == 3 || (s->path == 0 && !fused && n >= s->quad_min_n);
  const bool hex = s->path == 4 || (s->path == 0 && !fused && !quad && n >= s->hex_min_n);
  const SegPlan plan = (s->path == 0 && !fused && !quad && !hex) ? plan_segments(s, n, n_int) : SegPlan{};
  if (plan.n_seg) {
    // (d_partials is used on one stream at a time: a caller's stream takes the same hand-over as d_steps)
    steps_acquire(s, st);
    ScopedTimer t(s, st, STORM_KERNEL_SEGMENTS);
    det_segments(s, plan, d_freq, n, d_out, st, flags);
    t.stop();
    count_launch(s, 2);
    steps_release(s, st);
  } else if (hex) {
    // FMA build: all det_hex waves of a launch run equally long, so a batch of r.x wave rounds pays for the x: 12 501
    // frequencies are 3126 waves on 3072 slots and take 1.5 rounds' time.  The whole rounds go to det_hex, the remainder
    // (< one round) through the segment path, which has no such granularity (thousands of short groups).  This is what
    // keeps a strong-scaled slice (10^5 / 8 frequencies per GPU) at one round.  Not in the parity build (association).
    SegPlan rest_plan;
    const size_t n_hex = hex_whole_rounds(s, n, rest_plan);
    ScopedTimer t(s, st, STORM_KERNEL_DET_HEX);
    hip_check(g_launch_hex[s->arith][s->scheme](make_args(s, d_freq, n_hex, d_out, st, flags)), "det_hex launch");
    if (n_hex < n) {
      steps_acquire(s, st);
      det_segments(s, rest_plan, d_freq + n_hex, n - n_hex, d_out + n_hex, st, flags);
      steps_release(s, st);
      count_launch(s, 2);
    }
    t.stop();
    count_launch(s);
  } else if (quad) {
    ScopedTimer t(s, st, STORM_KERNEL_DET_QUAD);
    hip_check(g_launch_quad[s->arith][s->scheme](make_args(s, d_freq, n, d_out, st, flags)), "det_quad launch");
    t.stop();
    count_launch(s);
  } else if (fused) {
    ScopedTimer t(s, st, STORM_KERNEL_DET_FUSED);
    hip_check(g_launch_fused[s->arith][s->scheme](make_args(s, d_freq, n, d_out, st, flags)), "det_fused launch");
    t.stop();
    count_launch(s);
  } else {
    const size_t per_freq = n_int * 16 * sizeof(double);
    size_t chunk = per_freq ? s->steps_bytes_max / per_freq : n;
    if (chunk == 0) chunk = 1;
    if (chunk > n) chunk = n;
    if (chunk > 65535) chunk = 65535;  // grid.y limit
    steps_acquire(s, st);
    s->d_steps.reserve(chunk * n_int * 16, st);
    for (size_t off = 0; off < n; off += chunk) {
      const size_t m = std::min(chunk, n - off);
      ScopedTimer tb(s, st, STORM_KERNEL_BUILD_STEPS);
      hip_check(g_launch_build[s->arith][s->scheme](make_args(s, d_freq + off, m, s->d_steps.p, st, flags)), "build_steps launch");
      tb.stop();
      ScopedTimer tc(s, st, STORM_KERNEL_CHAIN);
      hip_check((s->arith ? storm_fast::launch_chain : storm::launch_chain)(s->sys, s->ell0, d_freq + off, m, s->d_steps.p, d_out + off, st, nullptr, nullptr), "chain launch");
      tc.stop();
      count_launch(s, 2);
    }
    steps_release(s, st);
  }
  s->stats.det_evals += n;
}

void det_host(storm_solver* s, const double* freq, size_t n, double* det_out) {
  if (n == 0) return;
  s->d_freq.reserve(n, s->stream);
  s->d_det.reserve(n, s->stream);
  hip_check(hipMemcpyAsync(s->d_freq.p, freq, n * sizeof(double), hipMemcpyHostToDevice, s->stream), "freq H2D");
  det_device(s, s->d_freq.p, n, s->d_det.p, s->stream);
  hip_check(hipMemcpyAsync(det_out, s->d_det.p, n * sizeof(double), hipMemcpyDeviceToHost, s->stream), "det D2H");
  check_flags(s);  // synchronises
}

// ---- bracketing (reference src/bracket.rs), lock-step over all brackets: hostlogic/brackets.hpp ----
void refine_all(storm_solver* s, std::vector<BracketState>& br, uint64_t ulp) {
  refine_brackets(br, ulp, [&](const double* xs, size_t n, double* fs) {
    det_host(s, xs, n, fs);
    s->stats.refine_evals += n;
    s->stats.refine_rounds += 1;
  });
}

// ---- the same on the device (common_kernels.hip, "Bracketing on the device") ----
constexpr size_t kMaxRounds = 4096;  // live counters per call (a bracket needs ~5-12 evaluations at 1e-8, < 70 at 1 ULP)

// Pinned staging of the device bracketing: [2 ints: flags][kMaxRounds + 1 live counters][block counts][brackets][indices]
struct BracketStage {
  int* flags; unsigned* live; unsigned* counts; BracketDev* br; unsigned long long* idx;
};
BracketStage bracket_stage(storm_solver* s, size_t n_blocks, size_t n_br) {
  size_t bytes = 16 + (kMaxRounds + 2) * sizeof(unsigned);
  bytes = (bytes + 15) / 16 * 16;
  const size_t off_counts = bytes;
  bytes += (n_blocks + 4) * sizeof(unsigned);
  bytes = (bytes + 15) / 16 * 16;
  const size_t off_br = bytes;
  bytes += n_br * sizeof(BracketDev);
  const size_t off_idx = bytes;
  bytes += n_br * sizeof(unsigned long long);
  bytes = std::max(bytes, size_t(256) << 10);  // room for ~2000 brackets from the start: regrowth is rare
  if ((bytes + 7) / 8 > s->h_br.cap && s->h_br.p)  // copies from / to the old block may still be in flight
    hip_check(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
  s->h_br.reserve((bytes + 7) / 8);
  char* base = reinterpret_cast<char*>(s->h_br.p);
  return BracketStage{reinterpret_cast<int*>(base), reinterpret_cast<unsigned*>(base + 16), reinterpret_cast<unsigned*>(base + off_counts),
                      reinterpret_cast<BracketDev*>(base + off_br), reinterpret_cast<unsigned long long*>(base + off_idx)};
}

void raise_bracket_flags(int flags) {
  if (flags & 8) fail(STORM_ERR_BRACKET, "bracket must satisfy 0 <= lower < upper < inf");
  if (flags & 4) fail(STORM_ERR_NON_FINITE, "non-finite discriminant during bracket refinement");
}

// InverseQuadratic::optimize on the n brackets in s->d_br, all of them advancing one evaluation per round; nothing but
// the live counters crosses PCIe between rounds.  `stage` must come from bracket_stage(s, ., >= n).
void refine_device(storm_solver* s, size_t n, uint64_t ulp, const BracketStage& stage) {
  if (ulp == 0) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be >= 1");
  if (!(ulp < (uint64_t(1) << 54))) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be < 2^54");
  if (n == 0) return;
  hipStream_t st = s->stream;
  s->d_bx.reserve(n, st); s->d_bf.reserve(n, st);
  s->d_live.reserve(kMaxRounds + 2, st);
  s->d_bflags.reserve(2, st);
  hip_check(hipMemsetAsync(s->d_live.p, 0, (kMaxRounds + 2) * sizeof(unsigned), st), "live memset");
  const uint64_t evals_before = s->stats.det_evals;
  // rounds between two looks at the live counters: a round costs ~30 us on the segment path (a wasted one is cheap),
  // ~0.5 ms on the ordered-product path (never waste one)
  const size_t batch = plan_segments(s, n, s->sys.n_int).n_seg ? 3 : 1;
  hip_check(storm::launch_refine_step(s->d_br.p, n, nullptr, s->d_bx.p, ulp, s->d_live.p, s->d_bflags.p, st), "refine_step launch");
  count_launch(s);
  size_t round = 0;  // live[r] = brackets that want evaluation r + 1
  for (;;) {
    // (no look at the counters before the first batch: a bracket that is converged before its first evaluation is rare)
    if (round + batch > kMaxRounds) fail(STORM_ERR_BRACKET, "bracket refinement did not converge");
    for (size_t b = 0; b < batch; ++b) {
      det_device(s, s->d_bx.p, n, s->d_bf.p, st);
      hip_check(storm::launch_refine_step(s->d_br.p, n, s->d_bf.p, s->d_bx.p, ulp, s->d_live.p + round + 1, s->d_bflags.p, st),
                "refine_step launch");
      count_launch(s);
      ++round;
    }
    hip_check(hipMemcpyAsync(stage.live, s->d_live.p, (round + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, st), "live D2H");
    hip_check(hipMemcpyAsync(stage.flags, s->d_bflags.p, sizeof(int), hipMemcpyDeviceToHost, st), "flags D2H");
    check_flags(s);  // synchronises; singular stage matrices
    raise_bracket_flags(*stage.flags);
    if (stage.live[round] == 0) break;
  }
  // counters as the host state machine keeps them: evaluations that a live bracket consumed, rounds that had one
  size_t rounds = 0;
  uint64_t evals = 0;
  for (size_t r = 0; r < round; ++r) { rounds += stage.live[r] > 0; evals += stage.live[r]; }
  s->stats.det_evals = evals_before + evals;
  s->stats.refine_evals += evals;
  s->stats.refine_rounds += rounds;
}

void download_brackets(storm_solver* s, size_t n, const BracketStage& stage, bool with_index) {
  if (n == 0) return;
  hip_check(hipMemcpyAsync(stage.br, s->d_br.p, n * sizeof(BracketDev), hipMemcpyDeviceToHost, s->stream), "brackets D2H");
  if (with_index)
    hip_check(hipMemcpyAsync(stage.idx, s->d_bidx.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream), "index D2H");
  sync_stream(s->stream, s->sync_marker);
}

// ---- eigenvectors: shared by storm_eigenvector_batch and storm_modes_batch ----
// Sizes the staging buffers; returns the number of modes processed per chunk.
size_t eigen_reserve(storm_solver* s, size_t n) {
  const size_t n_int = s->sys.n_int;
  const size_t len = (n_int + 1) * 4, ustride = n_int * 22 + 6;
  const size_t per_mode = (n_int * 16 + ustride + len) * sizeof(double);
  size_t chunk = s->steps_bytes_max / per_mode;
  if (chunk == 0) chunk = 1;
  if (chunk > n) chunk = n;
  if (chunk > 65535) chunk = 65535;
  s->d_freq.reserve(chunk, s->stream); s->d_det.reserve(chunk, s->stream);
  steps_acquire(s, s->stream);
  s->steps_user = nullptr;
  s->d_steps.reserve(chunk * n_int * 16 + 16, s->stream);
  s->d_upper.reserve((chunk + 64) * ustride, s->stream);  // lane-per-mode kernels pad the mode count to 64
  s->d_vec.reserve((chunk + 64) * len, s->stream);
  s->d_swaps.reserve(chunk * 2, s->stream);
  s->classic_chunk = chunk;
  return chunk;
}
// The sequential kernels: step matrices, pivoted elimination, back-substitution (bit-identical to the reference in the
// parity build).  Frequencies in d_freq; eigenvectors to d_vec, discriminants to d_det.
void eigen_classic(storm_solver* s, const double* d_freq, size_t m, double* d_vec, double* d_det) {
  ScopedTimer tb(s, s->stream, STORM_KERNEL_BUILD_STEPS);
  hip_check(g_launch_build[s->arith][s->scheme](make_args(s, d_freq, m, s->d_steps.p, s->stream)), "build_steps launch");
  tb.stop();
  ScopedTimer te(s, s->stream, STORM_KERNEL_EIGEN);
  hip_check((s->arith ? storm_fast::launch_eigen : storm::launch_eigen)(s->sys, s->ell0 ? 1 : 0, d_freq, m, s->d_steps.p, s->d_upper.p,
                                                                        s->d_swaps.p, d_det, d_vec, s->stream, nullptr, nullptr, s->eigen_layout), "eigen launch");
  te.stop();
  count_launch(s, 3);
}

// Segment-parallel eigenvectors (FMA build; see "segments" in storm_hip.h): transfer matrices of all (mode, segment)
// pairs, the reference's elimination + back-substitution on the n_seg-interval system of the segment boundaries, then
// the fill of every segment.  Modes whose transfer matrices grow beyond kSegGrowthMax are redone sequentially.
void eigen_segments(storm_solver* s, const SegPlan& plan, size_t m) {
  const size_t n_seg = size_t(plan.n_seg), len = (s->sys.n_int + 1) * 4;
  s->d_partials.reserve(m * n_seg * 16, s->stream);
  s->d_nodes.reserve((m + 64) * (n_seg + 1) * 4, s->stream);
  s->d_growth.reserve(m, s->stream);
  hip_check(hipMemsetAsync(s->d_growth.p, 0, m * sizeof(unsigned long long), s->stream), "growth memset");
  ScopedTimer ts(s, s->stream, STORM_KERNEL_SEGMENTS);
  LaunchArgs a = make_args(s, s->d_freq.p, m, s->d_partials.p, s->stream);
  a.n_seg = plan.n_seg; a.seg_len = plan.seg_len; a.growth = s->d_growth.p;
  hip_check(g_launch_seg_product[s->arith][s->scheme](a), "seg_product launch");
  ts.stop();
  SystemDev reduced = s->sys;
  reduced.n_int = n_seg;
  ScopedTimer te(s, s->stream, STORM_KERNEL_EIGEN);
  hip_check(storm_fast::launch_eigen(reduced, s->ell0 ? 1 : 0, s->d_freq.p, m, s->d_partials.p, s->d_upper.p, s->d_swaps.p,
                                     s->d_det.p, s->d_nodes.p, s->stream, nullptr, nullptr, s->eigen_layout), "eigen launch");
  te.stop();
  ScopedTimer tf(s, s->stream, STORM_KERNEL_SEGMENTS);
  a.nodes = s->d_nodes.p; a.vec = s->d_vec.p; a.growth = nullptr;
  hip_check(g_launch_seg_fill[s->arith][s->scheme](a), "seg_fill launch");
  tf.stop();
  count_launch(s, 4);
  // the guard: which modes have a segment whose transfer matrix is too large to propagate through?
  s->h_growth.reserve(m);
  hip_check(hipMemcpyAsync(s->h_growth.p, s->d_growth.p, m * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream), "growth D2H");
  sync_stream(s->stream, s->sync_marker);
  std::vector<size_t> redo;
  for (size_t k = 0; k < m; ++k)
    if (!(s->h_growth.p[k] <= s->seg_growth_max)) redo.push_back(k);  // (h_growth holds the bit patterns of |T|: read as doubles)
  if (redo.empty()) return;
  s->stats.segment_fallbacks += redo.size();
  std::vector<double> f(redo.size());
  // d_freq -> host is not needed: the caller's frequencies are still in the pinned upload area (see eigen_chunk)
  for (size_t q = 0; q < redo.size(); ++q) f[q] = s->h_freq_last[redo[q]];
  s->d_freq_fb.reserve(redo.size(), s->stream); s->d_vec_fb.reserve((redo.size() + 64) * len, s->stream);
  s->d_det_fb.reserve(redo.size(), s->stream);
  hip_check(hipMemcpyAsync(s->d_freq_fb.p, f.data(), f.size() * sizeof(double), hipMemcpyHostToDevice, s->stream), "freq H2D");
  hip_check(hipStreamSynchronize(s->stream), "hipStreamSynchronize");  // `f` is pageable and about to go out of scope
  for (size_t q0 = 0; q0 < redo.size(); q0 += s->classic_chunk) {
    const size_t k = std::min(s->classic_chunk, redo.size() - q0);
    eigen_classic(s, s->d_freq_fb.p + q0, k, s->d_vec_fb.p, s->d_det_fb.p);
    for (size_t q = 0; q < k; ++q) {
      hip_check(hipMemcpyAsync(s->d_vec.p + redo[q0 + q] * len, s->d_vec_fb.p + q * len, len * sizeof(double), hipMemcpyDeviceToDevice, s->stream), "vec D2D");
      hip_check(hipMemcpyAsync(s->d_det.p + redo[q0 + q], s->d_det_fb.p + q, sizeof(double), hipMemcpyDeviceToDevice, s->stream), "det D2D");
    }
  }
}

// Eigenvectors of m host frequencies into s->d_vec (and discriminants into s->d_det), left on the device.
void eigen_chunk(storm_solver* s, const double* freq, size_t m) {
  hip_check(hipMemcpyAsync(s->d_freq.p, freq, m * sizeof(double), hipMemcpyHostToDevice, s->stream), "freq H2D");
  const SegPlan plan = plan_segments(s, m, 256, 0, true);
  if (plan.n_seg) {
    s->h_freq_last = freq;
    eigen_segments(s, plan, m);
  } else {
    eigen_classic(s, s->d_freq.p, m, s->d_vec.p, s->d_det.p);
  }
  s->stats.eigenvectors += m;
}

// ---- post-processing ----
// src/gaunt.rs:1-17,32-37,46-51 (host code of this file is compiled -ffp-contract=off as well).
double gaunt_beta(uint64_t k, int64_t m) {
  const double kf = double(k), mf = double(m);
  return std::sqrt((kf * kf - mf * mf) / (4. * kf * kf - 1.));
}
double gaunt_q_kl1(uint64_t k, uint64_t l, int64_t m) {
  const double sign = (m % 2 == 0) ? 1.0 : -1.0;  // (-1)^(m % 2); (-1)^-1 = 1/(-1) = -1
  return sign * (k == l + 1 ? gaunt_beta(k, m) : (l == k + 1 ? gaunt_beta(l, m) : 0.));
}
double gaunt_q_kl1_h(uint64_t k, uint64_t l, int64_t m) {
  const double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return gaunt_q_kl1(k, l, m) * ((lk + ll) / 2. - 1.);
}
double gaunt_q_kl1_hd(uint64_t k, uint64_t l, int64_t m) {
  const double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return gaunt_q_kl1(k, l, m) * ((ll - lk) / 2. + 1.);
}
double gaunt_q_kl2(uint64_t k, uint64_t l, int64_t m) {  // src/gaunt.rs:19-30
  double v;
  if (k == l) { const double a = gaunt_beta(k + 1, m), b = gaunt_beta(k, m); v = a * a + b * b - 1. / 3.; }
  else if (k == l + 2) v = gaunt_beta(k, m) * gaunt_beta(l + 1, m);
  else if (k + 2 == l) v = gaunt_beta(k + 1, m) * gaunt_beta(l, m);
  else v = 0.;
  return (3. / 2.) * v;
}
double gaunt_q_kl2_h(uint64_t k, uint64_t l, int64_t m) {  // src/gaunt.rs:39-44
  const double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return gaunt_q_kl2(k, l, m) * ((lk + ll) / 2. - 3.);
}
double gaunt_q_kl2_hd(uint64_t k, uint64_t l, int64_t m) {  // src/gaunt.rs:53-58
  const double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return gaunt_q_kl2(k, l, m) * ((ll - lk) / 2. + 3.);
}
PostParams post_params(uint64_t ell, int64_t m) {  // src/postprocessing.rs:85-95
  if (ell > 1000000) fail(STORM_ERR_INVALID_ARGUMENT, "ell does not fit (the reference converts it to i32)");
  const uint64_t n1 = ell == 0 ? 0 : ell - 1;
  const uint64_t m_abs = m < 0 ? uint64_t(0) - uint64_t(m) : uint64_t(m);
  PostParams q;
  q.ell = int(ell);
  q.tn_zero = (m_abs == ell || ell == 1) ? 1 : 0;
  q.lambda = double(ell * (ell + 1));
  q.lambda_n1 = double(ell * n1);
  q.lambda_p1 = double((ell + 2) * (ell + 1));
  q.q_hd_n = gaunt_q_kl1_hd(ell, n1, m);
  q.q_hd_p = gaunt_q_kl1_hd(ell, ell + 1, m);
  q.q_h_n = gaunt_q_kl1_h(ell, n1, m);
  q.q_h_p = gaunt_q_kl1_h(ell, ell + 1, m);
  q.mf = double(m);
  q.ellf = double(ell);
  return q;
}
hipStream_t post_stream(storm_postproc* p) {
  if (!p->stream) p->stream = MemoryCache::get().take_stream(p->device);
  return p->stream;
}
// Makes the model planes available to work submitted on `st` after this call.
void ensure_model(storm_postproc* p, hipStream_t st) {
  if (!p->uploaded) {
    p->tab.reserve(p->tab_elems, st);
    hip_check(hipMemcpyAsync(p->tab.p, p->h_model.p, p->tab_elems * sizeof(double), hipMemcpyHostToDevice, st), "model H2D");
    hip_check(hipEventCreateWithFlags(&p->ready, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventRecord(p->ready, st), "hipEventRecord");
    p->model.tab = p->tab.p;
    p->upload_stream = st;
    p->uploaded = true;
  } else if (st != p->upload_stream) {
    hip_check(hipStreamWaitEvent(st, p->ready, 0), "hipStreamWaitEvent");
  }
}
size_t post_reserve(storm_postproc* p, size_t n, bool fields, size_t node_cap) {
  const size_t np = p->model.n_pts;
  const size_t per_mode = (np * (1 + 4 + (fields ? kPostFields : 0)) + node_cap + 8) * sizeof(double);
  size_t chunk = p->bytes_max / per_mode;
  if (chunk == 0) chunk = 1;
  if (chunk > n) chunk = n;
  if (chunk > 65535) chunk = 65535;
  p->d_terms.reserve(chunk * np, p->stream);
  p->d_scale.reserve(chunk, p->stream);
  p->d_ids.reserve(chunk * 4, p->stream);
  if (fields) p->d_fields.reserve(chunk * kPostFields * np, p->stream);
  if (node_cap) p->d_nodes.reserve(chunk * node_cap, p->stream);
  return chunk;
}
void post_chunk(storm_postproc* p, const PostParams& par, const double* d_freq, const double* d_vec, size_t k, bool fields,
                size_t node_cap, hipStream_t st) {
  PostArgs a;
  a.model = p->model; a.par = par; a.freq = d_freq; a.vec = d_vec; a.n = k;
  a.terms = p->d_terms.p; a.scale = p->d_scale.p; a.fields = fields ? p->d_fields.p : nullptr;
  a.ids = p->d_ids.p; a.nodes = node_cap ? p->d_nodes.p : nullptr; a.node_cap = node_cap; a.stream = st;
  hip_check(launch_postprocess(a), "postprocess launch");
}
// Copies the results of the last post_chunk to the caller (synchronises st).
void post_download(storm_postproc* p, size_t k, double* fields_out, storm_mode_id* ids_out, double* nodes_out, size_t node_cap,
                   hipStream_t st) {
  const size_t np = p->model.n_pts;
  const size_t nf = fields_out ? k * kPostFields * np : 0, nn = nodes_out ? k * node_cap : 0;
  p->h_stage.reserve(nf + nn + 1);
  p->h_ids.resize(k * 4);
  if (nf) hip_check(hipMemcpyAsync(p->h_stage.p, p->d_fields.p, nf * sizeof(double), hipMemcpyDeviceToHost, st), "fields D2H");
  if (nn) hip_check(hipMemcpyAsync(p->h_stage.p + nf, p->d_nodes.p, nn * sizeof(double), hipMemcpyDeviceToHost, st), "nodes D2H");
  hip_check(hipMemcpyAsync(p->h_ids.data(), p->d_ids.p, k * 4 * sizeof(long long), hipMemcpyDeviceToHost, st), "ids D2H");
  sync_stream(st, p->sync_marker);
  if (nf) std::memcpy(fields_out, p->h_stage.p, nf * sizeof(double));
  if (nn) std::memcpy(nodes_out, p->h_stage.p + nf, nn * sizeof(double));
  for (size_t i = 0; i < k; ++i)
    ids_out[i] = storm_mode_id{uint64_t(p->h_ids[i * 4]), uint64_t(p->h_ids[i * 4 + 1]), int64_t(p->h_ids[i * 4 + 2]),
                               uint64_t(p->h_ids[i * 4 + 3])};
}

}  // namespace

extern "C" {

const char* storm_last_error(void) { return g_error.c_str(); }
const char* storm_version(void) { return "storm_hip 0.2 (gfx950; arithmetic: parity [default] | fast)"; }

uint64_t storm_precision_relative_to_ulp(double rel) {
  const double q = std::floor(rel / std::numeric_limits<double>::epsilon());
  uint64_t u;
  if (!(q > 0.0)) u = 0;
  else if (q >= 18446744073709551615.0) u = UINT64_MAX;
  else u = uint64_t(q);
  return u < 1 ? 1 : u;
}

int storm_solver_create(const storm_system_desc* d, storm_solver** out) {
  if (out) *out = nullptr;
  storm_solver* s = nullptr;
  int rc = guarded([&] {
    if (!d || !out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (d->scheme < 0 || d->scheme >= kNumSchemes) fail(STORM_ERR_INVALID_ARGUMENT, "unknown difference scheme");
    if (d->p != size_t(scheme_points(d->scheme)))
      fail(STORM_ERR_INVALID_ARGUMENT, "p does not match the number of collocation points of the scheme");
    if (d->n_int > (size_t(1) << 40)) fail(STORM_ERR_INVALID_ARGUMENT, "n_int too large");
    if (d->n_int && (!d->u || !d->v_gamma || !d->a_star || !d->c1 || !d->rot || !d->delta_over_x))
      fail(STORM_ERR_INVALID_ARGUMENT, "null coefficient array");
    int ndev = 0;
    hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev <= 0) fail(STORM_ERR_HIP, "no HIP device available");
    s = new storm_solver();
    hip_check(hipGetDevice(&s->device), "hipGetDevice");
    s->scheme = d->scheme;
    double rotm0 = 0.0;
    const size_t soa_elems = planes_elems(d->n_int, d->p), tab_elems = soa_elems + aos_elems(d->n_int, d->p);
    s->h_vec.reserve(soa_elems);  // built in the (cached) pinned block the upload is staged from
    build_planes(*d, s->h_vec.p, s->uniform, rotm0);
    const double ell = double(d->ell);
    s->ell0 = (ell == 0.0);
    if (s->ell0) s->uniform = true;  // rotation does not enter for ell == 0
    s->tab.reserve(tab_elems, nullptr);
    s->stream = MemoryCache::get().take_stream(s->device);
    s->scratch = scratch_of_current_device();
    // Uploads are ordered on the solver's own (non-blocking) stream: a plain hipMemcpy from pageable
    // memory is only ordered against the NULL stream, so a kernel launched on s->stream right after
    // create could otherwise read the planes before the DMA has landed.
    // (staged through a cached pinned buffer: a copy from pageable memory goes through the runtime's own staging
    // path, which serialises concurrent creates)
    hip_check(hipMemcpyAsync(s->tab.p, s->h_vec.p, soa_elems * sizeof(double), hipMemcpyHostToDevice, s->stream),
              "planes H2D");
    hip_check(storm::launch_aos_copy(s->tab.p, 1, tab_elems, padded_stride(d->n_int), int(d->p), s->stream), "aos copy launch");
    s->d_flags.reserve(2);  // [0]: calls on the library stream (read back and reset by check_flags), [1]: caller streams
    hip_check(hipMemsetAsync(s->d_flags.p, 0, 2 * sizeof(int), s->stream), "flags memset");
    hip_check(hipEventCreateWithFlags(&s->ready, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventRecord(s->ready, s->stream), "hipEventRecord");
    // no wait: every later kernel of this handle runs on the same stream, behind the upload; the pinned staging
    // buffer belongs to the handle and is next written only after a synchronisation of that stream
    s->sys.tab = s->tab.p;
    s->sys.stride = padded_stride(d->n_int);
    s->sys.plane = s->sys.stride * d->p;
    s->sys.n_int = d->n_int;
    s->ell = d->ell; s->m = d->m;
    s->sys.ell = ell;
    s->sys.lambda = ell * (ell + 1.0);
    s->sys.inner_c1 = d->inner_c1;
    s->sys.inner_rotm = double(d->m) * d->inner_rot;
    s->sys.outer_u = d->outer_u;
    s->sys.rotm_uniform = rotm0;
    *out = s;
  });
  if (rc != STORM_OK && s) {
    storm_solver_destroy(s);
    if (out) *out = nullptr;
  }
  return rc;
}

void storm_solver_destroy(storm_solver* s) {
  if (!s) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(s->device);
  if (s->stream) { (void)hipStreamSynchronize(s->stream); MemoryCache::get().give_stream(s->device, s->stream); }
  for (auto& t : s->pending) { if (t.start) (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
  for (hipEvent_t e : s->event_pool) (void)hipEventDestroy(e);
  if (s->sync_marker) (void)hipEventDestroy(s->sync_marker);
  if (s->steps_done) { (void)hipEventSynchronize(s->steps_done); (void)hipEventDestroy(s->steps_done); }
  if (s->ready) (void)hipEventDestroy(s->ready);
  s->tab.release(); s->d_freq.release(); s->d_det.release(); s->d_steps.release();
  s->d_upper.release(); s->d_vec.release(); s->d_swaps.release(); s->d_flags.release(); s->h_vec.release(); s->h_flags.release();
  s->d_partials.release(); s->d_nodes.release(); s->d_vec_fb.release(); s->d_det_fb.release(); s->d_freq_fb.release();
  s->d_growth.release(); s->h_growth.release();
  s->d_br.release(); s->d_bidx.release(); s->d_bcount.release(); s->d_live.release(); s->d_bx.release(); s->d_bf.release();
  s->d_bflags.release(); s->h_br.release();
  (void)hipSetDevice(prev);
  delete s;
}

size_t storm_solver_n_int(const storm_solver* s) { return s ? s->sys.n_int : 0; }

int storm_solver_set_option(storm_solver* s, const char* key, int64_t value) {
  return guarded([&] {
    if (!s || !key) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    const std::string k(key);
    if (k == "path") {
      if (value < 0 || value > 4) fail(STORM_ERR_INVALID_ARGUMENT, "path must be 0..4");
      s->path = int(value);
    } else if (k == "steps_bytes_max") {
      if (value < 1) fail(STORM_ERR_INVALID_ARGUMENT, "steps_bytes_max must be positive");
      s->steps_bytes_max = size_t(value);
    } else if (k == "arithmetic") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "arithmetic must be 0 (parity) or 1 (fast)");
      s->arith = int(value);
    } else if (k == "profile") {
      s->profile = value != 0;
    } else if (k == "scaled_scan") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "scaled_scan must be 0 or 1");
      s->scaled_scan = int(value);
    } else if (k == "device_brackets") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "device_brackets must be 0 (host state machine) or 1");
      s->device_brackets = int(value);
    } else if (k == "segments") {
      if (value < -1 || value > (1 << 20)) fail(STORM_ERR_INVALID_ARGUMENT, "segments must be -1 (off), 0 (auto) or a segment length");
      s->segments = int(value);
    } else if (k == "seg_growth_log2") {
      if (value < 0 || value > 1000) fail(STORM_ERR_INVALID_ARGUMENT, "seg_growth_log2 must be 0..1000");
      s->seg_growth_max = std::ldexp(1.0, int(value));
    } else if (k == "hex_min_n") {
      if (value < 0) fail(STORM_ERR_INVALID_ARGUMENT, "hex_min_n must be >= 0");
      s->hex_min_n = size_t(value);
    } else if (k == "hex_lanes") {
      if (value != 0 && value != 4 && value != 16 && value != 32 && !(value < 0 && value > -100))
        fail(STORM_ERR_INVALID_ARGUMENT, "hex_lanes must be 0 (auto), 4, 16 or 32");
      s->hex_lanes = int(value);
    } else if (k == "hex_rounds") {
      if (value < 0 || value > 1024) fail(STORM_ERR_INVALID_ARGUMENT, "hex_rounds must be 0 (one launch) .. 1024");
      s->hex_rounds = int(value);
    } else if (k == "eigen_layout") {
      if (value < 0 || value > 2) fail(STORM_ERR_INVALID_ARGUMENT, "eigen_layout must be 0 (auto), 1 (wave per mode) or 2 (lane per mode)");
      s->eigen_layout = int(value);
    } else if (k == "quad_min_n") {
      if (value < 0) fail(STORM_ERR_INVALID_ARGUMENT, "quad_min_n must be >= 0");
      s->quad_min_n = size_t(value);
    } else if (k == "fused_min_n") {
      if (value < 0) fail(STORM_ERR_INVALID_ARGUMENT, "fused_min_n must be >= 0");
      s->fused_min_n = size_t(value);
    } else {
      fail(STORM_ERR_INVALID_ARGUMENT, "unknown option: " + k);
    }
  });
}

int storm_solver_stats(storm_solver* s, storm_stats* out, int reset) {
  return guarded([&] {
    if (!s || !out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    *out = s->stats;
    if (reset) s->stats = storm_stats{};
  });
}

int storm_solver_kernel_times(storm_solver* s, double* ms_out, uint64_t* launches_out, int reset) {
  return guarded([&] {
    if (!s || !ms_out || !launches_out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    for (int k = 0; k < STORM_KERNEL_KINDS; ++k) { ms_out[k] = s->kernel_ms[k]; launches_out[k] = s->kernel_launches[k]; }
    if (reset) for (int k = 0; k < STORM_KERNEL_KINDS; ++k) { s->kernel_ms[k] = 0; s->kernel_launches[k] = 0; }
  });
}

int storm_det_batch(storm_solver* s, const double* freq, size_t n, double* det_out) {
  return guarded([&] {
    if (!s || (n && (!freq || !det_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard g(s->device);
    det_host(s, freq, n, det_out);
  });
}

int storm_det_batch_device(storm_solver* s, const double* d_freq, size_t n, double* d_out, void* hip_stream) {
  return guarded([&] {
    if (!s || (n && (!d_freq || !d_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard g(s->device);
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream;
    if (st != s->stream) {
      // the planes were uploaded asynchronously on the library stream by storm_solver_create
      hip_check(hipStreamWaitEvent(st, s->ready, 0), "hipStreamWaitEvent");
      // a singular stage shows as NaN in d_out; its flag goes to a word of its own that no later call reports
      det_device(s, d_freq, n, d_out, st, s->d_flags.p + 1);
    } else {
      det_device(s, d_freq, n, d_out, st);
      check_flags(s);
    }
  });
}

// The discriminant as mantissa * 2^exponent for n host frequencies (caller holds the handle's lock, device current): step
// matrices staged in chunks of steps_bytes_max, chain_kernel<.., SCALED> over them.
static void det_scaled_staged(storm_solver* s, const double* freq, size_t n, double* mantissa_out, int64_t* exponent_out);

// FMA build: the same number through the segment path -- seg_product forms the transfer matrix of every (frequency,
// segment of 128 intervals) pair with the fast kernels' arithmetic (a segment's product stays far inside f64: the raw
// product of a whole grid needs ~1 bit per interval to overflow), then chain_kernel<.., SCALED> multiplies the segments'
// matrices with a renormalisation after every one.  A whole-grid scaled scan then costs about what the raw scan costs
// instead of staging 128 B per (frequency, interval).  A frequency whose value comes out non-finite (a segment that did
// overflow, or a genuinely non-finite discriminant) is evaluated again through the staged path, which renormalises every
// 16 intervals.  Not in the parity build (the segment path re-associates the ordered product).
static bool det_scaled_segments(storm_solver* s, const double* freq, size_t n, double* mantissa_out, int64_t* exponent_out) {
  const size_t n_int = s->sys.n_int;
  if (s->arith != 1 || s->segments < 0 || n_int < 64) return false;
  hipStream_t st = s->stream;
  const size_t len = s->segments > 0 ? (size_t(s->segments) + 15) / 16 * 16 : 128;
  const size_t n_seg = (1 + (n_int - 1 + len - 1) / len + 3) / 4 * 4;   // (the first interval is a segment of its own: plan_segments)
  if (n_seg > (size_t(1) << 24)) return false;
  const size_t per_freq = n_seg * 16 * sizeof(double);
  const size_t chunk = std::min<size_t>({n, std::max<size_t>(4, s->steps_bytes_max / per_freq / 4 * 4), ((size_t(1) << 31) - 1) * 4 / n_seg / 4 * 4});
  s->d_freq.reserve(chunk, st); s->d_det.reserve(chunk, st);
  s->d_bidx.reserve(chunk, st);  // (exponents: 8 bytes per frequency)
  steps_acquire(s, st);
  s->steps_user = nullptr;
  s->d_partials.reserve(chunk * n_seg * 16, st);
  SystemDev reduced = s->sys;
  reduced.n_int = n_seg;
  for (size_t off = 0; off < n; off += chunk) {
    const size_t m = std::min(chunk, n - off);
    hip_check(hipMemcpyAsync(s->d_freq.p, freq + off, m * sizeof(double), hipMemcpyHostToDevice, st), "freq H2D");
    LaunchArgs a = make_args(s, s->d_freq.p, m, s->d_partials.p, st);
    a.n_seg = int(n_seg); a.seg_len = int(len);
    hip_check(g_launch_seg_product[s->arith][s->scheme](a), "seg_product launch");
    hip_check(storm_fast::launch_chain_scaled(reduced, s->ell0, s->d_freq.p, m, s->d_partials.p, s->d_det.p,
                                              reinterpret_cast<long long*>(s->d_bidx.p), st, true), "chain launch");
    count_launch(s, 2);
    hip_check(hipMemcpyAsync(mantissa_out + off, s->d_det.p, m * sizeof(double), hipMemcpyDeviceToHost, st), "mantissa D2H");
    hip_check(hipMemcpyAsync(exponent_out + off, s->d_bidx.p, m * sizeof(int64_t), hipMemcpyDeviceToHost, st), "exponent D2H");
    hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
    // (the singular-stage flag is looked at by the staged pass below if one is needed, else here)
  }
  s->stats.det_evals += n;
  std::vector<size_t> again;
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(mantissa_out[i])) again.push_back(i);
  if (!again.empty()) {
    std::vector<double> f(again.size()), m2(again.size());
    std::vector<int64_t> e2(again.size());
    for (size_t k = 0; k < again.size(); ++k) f[k] = freq[again[k]];
    // a segment that overflowed may have raised the singular-stage flag on the way (inf - inf in a stage solve): the staged
    // pass decides
    hip_check(hipMemsetAsync(s->d_flags.p, 0, sizeof(int), st), "flags memset");
    det_scaled_staged(s, f.data(), f.size(), m2.data(), e2.data());
    s->stats.det_evals -= again.size();
    for (size_t k = 0; k < again.size(); ++k) { mantissa_out[again[k]] = m2[k]; exponent_out[again[k]] = e2[k]; }
  } else {
    check_flags(s);
  }
  return true;
}

static void det_scaled_locked(storm_solver* s, const double* freq, size_t n, double* mantissa_out, int64_t* exponent_out) {
  if (det_scaled_segments(s, freq, n, mantissa_out, exponent_out)) return;
  det_scaled_staged(s, freq, n, mantissa_out, exponent_out);
}

static void det_scaled_staged(storm_solver* s, const double* freq, size_t n, double* mantissa_out, int64_t* exponent_out) {
  hipStream_t st = s->stream;
  const size_t n_int = s->sys.n_int;
  const size_t per_freq = std::max<size_t>(1, n_int) * 16 * sizeof(double);
  size_t chunk = std::min<size_t>({n, std::max<size_t>(1, s->steps_bytes_max / per_freq), size_t(65535)});
  s->d_freq.reserve(chunk, st); s->d_det.reserve(chunk, st);
  s->d_bidx.reserve(chunk, st);  // (the exponents share the bracket-index buffer: 8 bytes per frequency)
  steps_acquire(s, st);
  s->steps_user = nullptr;
  s->d_steps.reserve(chunk * n_int * 16 + 16, st);
  for (size_t off = 0; off < n; off += chunk) {
    const size_t m = std::min(chunk, n - off);
    hip_check(hipMemcpyAsync(s->d_freq.p, freq + off, m * sizeof(double), hipMemcpyHostToDevice, st), "freq H2D");
    hip_check(g_launch_build[s->arith][s->scheme](make_args(s, s->d_freq.p, m, s->d_steps.p, st)), "build_steps launch");
    hip_check((s->arith ? storm_fast::launch_chain_scaled : storm::launch_chain_scaled)(
                  s->sys, s->ell0, s->d_freq.p, m, s->d_steps.p, s->d_det.p, reinterpret_cast<long long*>(s->d_bidx.p), st, false),
              "chain launch");
    count_launch(s, 2);
    hip_check(hipMemcpyAsync(mantissa_out + off, s->d_det.p, m * sizeof(double), hipMemcpyDeviceToHost, st), "mantissa D2H");
    hip_check(hipMemcpyAsync(exponent_out + off, s->d_bidx.p, m * sizeof(int64_t), hipMemcpyDeviceToHost, st), "exponent D2H");
    check_flags(s);  // synchronises
  }
  s->stats.det_evals += n;
}

int storm_det_scaled_batch(storm_solver* s, const double* freq, size_t n, double* mantissa_out, int64_t* exponent_out) {
  return guarded([&] {
    if (!s || (n && (!freq || !mantissa_out || !exponent_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard g(s->device);
    det_scaled_locked(s, freq, n, mantissa_out, exponent_out);
  });
}

int storm_det_batch_device_noalloc(storm_solver* s, const double* d_freq, size_t n, double* d_out, void* hip_stream) {
  return guarded([&] {
    if (!s || (n && (!d_freq || !d_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return;
    if (n > kNoallocMaxN) fail(STORM_ERR_CAPACITY, "more than STORM_NOALLOC_MAX_N frequencies in one allocation-free call");
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard g(s->device);
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream;
    int* flags = reinterpret_cast<int*>(s->scratch);
    double* partial = reinterpret_cast<double*>(s->scratch + kScratchFlagBytes);
    const size_t n_int = s->sys.n_int;
    // The kernel choice of storm_det_batch_device (det_device), options `path`, `fused_min_n`, `quad_min_n`, `hex_min_n`
    // and `segments` included, with two differences forced by the fixed scratch: the staged path (build_steps + chain,
    // `path` = 2 or a small batch without a segment plan) needs a buffer of 128 B per (frequency, interval) and is replaced
    // by det_hex, which computes the same ordered product; and a segment plan whose transfer matrices do not fit the
    // scratch is dropped in favour of det_hex as well (not an error).
    const bool fused = s->path == 1 || (s->path == 0 && (n >= s->fused_min_n || n_int == 0));
    const bool quad = s->path == 3 || (s->path == 0 && !fused && n >= s->quad_min_n);
    const bool hex = s->path == 4 || (s->path == 0 && !fused && !quad && n >= s->hex_min_n);
    const auto fits = [&](const SegPlan& p, size_t m) { return m * size_t(p.n_seg) * 16 * sizeof(double) <= kScratchPartialBytes; };
    SegPlan plan = (s->path == 0 && !fused && !quad && !hex) ? plan_segments(s, n, n_int) : SegPlan{};
    if (plan.n_seg && !fits(plan, n)) plan = SegPlan{};
    SegPlan rest_plan;
    size_t n_hex = n;
    if (!plan.n_seg && !fused && !quad && s->path != 2) {
      n_hex = hex_whole_rounds(s, n, rest_plan);
      if (rest_plan.n_seg && !fits(rest_plan, n - n_hex)) { rest_plan = SegPlan{}; n_hex = n; }
    }
    if (st != s->stream) hip_check(hipStreamWaitEvent(st, s->ready, 0), "hipStreamWaitEvent");
    hip_check(hipMemsetAsync(flags, 0, 2 * sizeof(int), st), "flags memset");  // the scratch keeps the last call's contents
    LaunchArgs a = make_args(s, d_freq, n, d_out, st, flags);
    if (plan.n_seg) {
      a.out = partial; a.n_seg = plan.n_seg; a.seg_len = plan.seg_len;
      hip_check(g_launch_seg_product[s->arith][s->scheme](a), "seg_product launch");
      SystemDev reduced = s->sys;
      reduced.n_int = size_t(plan.n_seg);
      hip_check(storm_fast::launch_chain(reduced, s->ell0, d_freq, n, partial, d_out, st), "chain launch");
      count_launch(s, 2);
    } else if (fused) {
      hip_check(g_launch_fused[s->arith][s->scheme](a), "det_fused launch");
      count_launch(s);
    } else if (quad) {
      hip_check(g_launch_quad[s->arith][s->scheme](a), "det_quad launch");
      count_launch(s);
    } else {
      a.n = n_hex;
      hip_check(g_launch_hex[s->arith][s->scheme](a), "det_hex launch");
      count_launch(s);
      if (n_hex < n) {  // the remainder of the last wave round, as in storm_det_batch_device
        LaunchArgs r = make_args(s, d_freq + n_hex, n - n_hex, partial, st, flags);
        r.n_seg = rest_plan.n_seg; r.seg_len = rest_plan.seg_len;
        hip_check(g_launch_seg_product[s->arith][s->scheme](r), "seg_product launch");
        SystemDev reduced = s->sys;
        reduced.n_int = size_t(rest_plan.n_seg);
        hip_check(storm_fast::launch_chain(reduced, s->ell0, d_freq + n_hex, n - n_hex, partial, d_out + n_hex, st), "chain launch");
        count_launch(s, 2);
      }
    }
    s->stats.det_evals += n;
    if (!hip_stream) hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
  });
}

// (filter_sign_swaps: hostlogic/brackets.hpp)

// scan_and_optimize with the filter, the compaction and the refinement on the device: the grid goes up once, the
// brackets come down once (plus the scan values if the caller wants them).
static void scan_on_device(storm_solver* s, const double* freq, size_t n, bool refine, uint64_t ulp, storm_bracket* out,
                           size_t cap, size_t* n_out, int* had_invalid, double* det_out, uint64_t* index_out) {
  if (refine && ulp == 0) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be >= 1");
  if (refine && !(ulp < (uint64_t(1) << 54))) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be < 2^54");
  if (n == 0) return;
  hipStream_t st = s->stream;
  s->d_freq.reserve(n, st); s->d_det.reserve(n, st);
  hip_check(hipMemcpyAsync(s->d_freq.p, freq, n * sizeof(double), hipMemcpyHostToDevice, st), "freq H2D");
  det_device(s, s->d_freq.p, n, s->d_det.p, st);
  if (det_out) hip_check(hipMemcpyAsync(det_out, s->d_det.p, n * sizeof(double), hipMemcpyDeviceToHost, st), "det D2H");
  // sign-swap filter, compaction and (if asked) refinement as ONE stream of launches: count per block -> exclusive sum on
  // the device -> emit in ascending order -> the total and the flags come to the host with one synchronisation (round 2:
  // counts to the host, sum there, offsets back, a second synchronisation after the emission)
  const size_t nb = swap_blocks(n);
  const size_t room = std::min(cap, n - 1 ? n - 1 : 1);   // brackets the device may write: the caller's, at most one per pair
  s->d_bcount.reserve(nb + 4, st);
  s->d_bflags.reserve(2, st);
  s->d_br.reserve(std::max<size_t>(room, 1), st); s->d_bidx.reserve(std::max<size_t>(room, 1), st);
  hip_check(hipMemsetAsync(s->d_bflags.p, 0, 2 * sizeof(int), st), "flags memset");
  hip_check(storm::launch_swap_count(s->d_det.p, n, s->d_bcount.p, s->d_bflags.p, st), "swap_count launch");
  hip_check(storm::launch_swap_offsets(s->d_bcount.p, nb, st), "swap_offsets launch");
  hip_check(storm::launch_swap_emit(s->d_freq.p, s->d_det.p, n, s->d_bcount.p, room, s->d_br.p, s->d_bidx.p, st), "swap_emit launch");
  count_launch(s, 3);
  // a scan finds a few dozen brackets: without refinement the first kEarly of them ride along with the total
  constexpr size_t kEarly = 256;
  BracketStage stage = bracket_stage(s, nb, kEarly);
  hip_check(hipMemcpyAsync(stage.counts, s->d_bcount.p + nb, sizeof(unsigned), hipMemcpyDeviceToHost, st), "total D2H");
  hip_check(hipMemcpyAsync(stage.flags, s->d_bflags.p, sizeof(int), hipMemcpyDeviceToHost, st), "flags D2H");
  const size_t early = refine ? 0 : std::min(room, kEarly);
  if (early) {
    hip_check(hipMemcpyAsync(stage.br, s->d_br.p, early * sizeof(BracketDev), hipMemcpyDeviceToHost, st), "brackets D2H");
    hip_check(hipMemcpyAsync(stage.idx, s->d_bidx.p, early * sizeof(unsigned long long), hipMemcpyDeviceToHost, st), "index D2H");
  }
  check_flags(s);  // synchronises; singular stage matrices
  const size_t total = stage.counts[0];
  if (had_invalid) *had_invalid = (*stage.flags & 2) ? 1 : 0;
  *n_out = total;
  if (total > cap) fail(STORM_ERR_CAPACITY, "bracket output buffer too small");
  if (total == 0) return;
  if (total > (size_t(1) << 31)) fail(STORM_ERR_CAPACITY, "too many brackets");
  if (refine || total > early) {
    stage = bracket_stage(s, nb, total);  // (may move the pinned block: nothing in it is needed any more)
    if (refine) refine_device(s, total, ulp, stage);
    download_brackets(s, total, stage, true);
  }
  for (size_t k = 0; k < total; ++k) {
    const BracketDev& b = stage.br[k];
    if (!std::isfinite(b.lo_x) || !std::isfinite(b.hi_x)) fail(STORM_ERR_BRACKET, "non-finite scan frequency");
    out[k] = storm_bracket{b.lo_x, b.lo_f, b.hi_x, b.hi_f, refine ? b.root : std::numeric_limits<double>::quiet_NaN(), b.evals};
    if (index_out) index_out[k] = stage.idx[k];
  }
}

// ---- Option "scaled_scan": bracketing on the scaled discriminant (north_star's "running log-determinant") ----
// The reference multiplies the raw step matrices (src/solver.rs:504,531,541) and can only notice afterwards that the product
// left the range of f64 (src/dynamic_interface.rs:130-137: the pair is dropped, the scan reports invalid numbers).  With
// this option a scan evaluates mantissa * 2^exponent instead (det_scaled_locked): the sign-swap filter needs the sign
// only, and InverseQuadratic::optimize is invariant under a common scaling of its three values, so every bracket works on
// values brought to ONE power of two of its own, E = the larger exponent of its two ends (exact scalings: where the raw
// product is finite the proposals are the raw run's up to the rounding of the closing elimination).
// Values handed back (det_out, lo_f, hi_f) are mantissa * 2^exponent saturated to +-inf / +-0: the raw value where it exists.
inline double saturating_ldexp(double m, int64_t e) {
  if (m == 0.0 || m != m) return m;
  if (e > 2000) return std::signbit(m) ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
  if (e < -2200) return std::signbit(m) ? -0.0 : 0.0;
  return std::ldexp(m, int(e));
}
// value on a bracket's scale: exponent differences beyond +-1000 are clamped (the sign survives, which is what keeps the
// bracket a bracket; the interpolation then degrades to a bisection-like step instead of meeting an infinity)
inline double on_scale(double m, int64_t e, int64_t scale) {
  const int64_t d = std::max<int64_t>(-1000, std::min<int64_t>(1000, e - scale));
  return std::ldexp(m, int(d));
}
static void refine_scaled(storm_solver* s, std::vector<BracketState>& br, const std::vector<int64_t>& scale, uint64_t ulp) {
  std::vector<double> mant;
  std::vector<int64_t> expo;
  refine_brackets(br, ulp, [&](const double* xs, size_t n, double* fs, const size_t* who) {
    mant.resize(n); expo.resize(n);
    det_scaled_locked(s, xs, n, mant.data(), expo.data());
    s->stats.det_evals -= n;   // (counted as refinement evaluations, like the raw path's)
    for (size_t q = 0; q < n; ++q) fs[q] = on_scale(mant[q], expo[q], scale[who[q]]);
    s->stats.refine_evals += n;
    s->stats.refine_rounds += 1;
  });
}
static void scan_scaled(storm_solver* s, const double* freq, size_t n, bool refine, uint64_t ulp, storm_bracket* out,
                        size_t cap, size_t* n_out, int* had_invalid, double* det_out, uint64_t* index_out) {
  if (n == 0) return;
  std::vector<double> mant(n);
  std::vector<int64_t> expo(n);
  det_scaled_locked(s, freq, n, mant.data(), expo.data());
  if (det_out)
    for (size_t i = 0; i < n; ++i) det_out[i] = saturating_ldexp(mant[i], expo[i]);
  std::vector<BracketState> br;
  std::vector<uint64_t> idx;
  bool invalid = false;
  filter_sign_swaps(freq, mant.data(), n, br, idx, invalid);   // (a NaN mantissa is a NaN discriminant: dropped and reported)
  if (had_invalid) *had_invalid = invalid ? 1 : 0;
  *n_out = br.size();
  if (br.size() > cap) fail(STORM_ERR_CAPACITY, "bracket output buffer too small");
  std::vector<int64_t> scale(br.size());
  for (size_t k = 0; k < br.size(); ++k) {
    if (!std::isfinite(br[k].lower.x) || !std::isfinite(br[k].upper.x)) fail(STORM_ERR_BRACKET, "non-finite scan frequency");
    const size_t i = size_t(idx[k]);
    scale[k] = std::max(expo[i], expo[i + 1]);
    br[k].lower.f = on_scale(mant[i], expo[i], scale[k]);
    br[k].upper.f = on_scale(mant[i + 1], expo[i + 1], scale[k]);
  }
  if (refine) refine_scaled(s, br, scale, ulp);
  for (size_t k = 0; k < br.size(); ++k) {
    out[k] = storm_bracket{br[k].lower.x, saturating_ldexp(br[k].lower.f, scale[k]), br[k].upper.x, saturating_ldexp(br[k].upper.f, scale[k]),
                           refine ? br[k].root : std::numeric_limits<double>::quiet_NaN(), br[k].evals};
    if (index_out) index_out[k] = idx[k];
  }
}
// storm_optimize_brackets under "scaled_scan": the end values may be infinite (saturated), so only their signs are taken
// from the caller and both ends are evaluated again (one batch) to put them on the bracket's scale.
static void optimize_scaled(storm_solver* s, storm_bracket* brackets, size_t n, uint64_t ulp) {
  std::vector<double> xs(2 * n), mant(2 * n);
  std::vector<int64_t> expo(2 * n);
  for (size_t k = 0; k < n; ++k) {
    const storm_bracket& b = brackets[k];
    if (!std::isfinite(b.lo_x) || !std::isfinite(b.hi_x)) fail(STORM_ERR_BRACKET, "bracket end points must be finite");
    if (b.lo_f != b.lo_f || b.hi_f != b.hi_f || sign_of(b.lo_f) == sign_of(b.hi_f)) fail(STORM_ERR_BRACKET, "bracket end values must differ in sign");
    xs[2 * k] = b.lo_x; xs[2 * k + 1] = b.hi_x;
  }
  det_scaled_locked(s, xs.data(), 2 * n, mant.data(), expo.data());
  std::vector<BracketState> br(n);
  std::vector<int64_t> scale(n);
  for (size_t k = 0; k < n; ++k) {
    if (sign_of(mant[2 * k]) != sign_of(brackets[k].lo_f) || sign_of(mant[2 * k + 1]) != sign_of(brackets[k].hi_f))
      fail(STORM_ERR_BRACKET, "bracket end values do not have the signs of the discriminant at the end points");
    scale[k] = std::max(expo[2 * k], expo[2 * k + 1]);
    br[k].lower = Pt{xs[2 * k], on_scale(mant[2 * k], expo[2 * k], scale[k])};
    br[k].upper = Pt{xs[2 * k + 1], on_scale(mant[2 * k + 1], expo[2 * k + 1], scale[k])};
  }
  refine_scaled(s, br, scale, ulp);
  for (size_t k = 0; k < n; ++k)
    brackets[k] = storm_bracket{br[k].lower.x, saturating_ldexp(br[k].lower.f, scale[k]), br[k].upper.x,
                                saturating_ldexp(br[k].upper.f, scale[k]), br[k].root, br[k].evals};
}

static void scan_impl(storm_solver* s, const double* freq, size_t n, bool refine, uint64_t ulp, storm_bracket* out,
                      size_t cap, size_t* n_out, int* had_invalid, double* det_out, uint64_t* index_out) {
  if (!s || !n_out || (n && !freq) || (cap && !out)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> lock(s->mu);
  DeviceGuard g(s->device);
  if (s->scaled_scan) {
    scan_scaled(s, freq, n, refine, ulp, out, cap, n_out, had_invalid, det_out, index_out);
    return;
  }
  if (s->device_brackets) {
    scan_on_device(s, freq, n, refine, ulp, out, cap, n_out, had_invalid, det_out, index_out);
    return;
  }
  std::vector<double> local;
  double* dets = det_out;
  if (!dets) { local.resize(n); dets = local.data(); }
  det_host(s, freq, n, dets);
  std::vector<BracketState> br;
  std::vector<uint64_t> idx;
  bool invalid = false;
  filter_sign_swaps(freq, dets, n, br, idx, invalid);
  if (had_invalid) *had_invalid = invalid ? 1 : 0;
  *n_out = br.size();
  if (br.size() > cap) fail(STORM_ERR_CAPACITY, "bracket output buffer too small");
  for (const BracketState& b : br)
    if (!std::isfinite(b.lower.x) || !std::isfinite(b.upper.x))
      fail(STORM_ERR_BRACKET, "non-finite scan frequency");
  if (refine) refine_all(s, br, ulp);
  for (size_t k = 0; k < br.size(); ++k) {
    out[k] = storm_bracket{br[k].lower.x, br[k].lower.f, br[k].upper.x, br[k].upper.f,
                           refine ? br[k].root : std::numeric_limits<double>::quiet_NaN(), br[k].evals};
    if (index_out) index_out[k] = idx[k];
  }
}

int storm_scan_and_optimize(storm_solver* s, const double* freq, size_t n, uint64_t ulp, storm_bracket* out,
                            size_t cap, size_t* n_out, int* had_invalid, double* det_out, uint64_t* index_out) {
  if (n_out) *n_out = 0;
  if (had_invalid) *had_invalid = 0;
  return guarded([&] { scan_impl(s, freq, n, true, ulp, out, cap, n_out, had_invalid, det_out, index_out); });
}

int storm_scan_brackets(storm_solver* s, const double* freq, size_t n, storm_bracket* out, size_t cap, size_t* n_out,
                        int* had_invalid, double* det_out, uint64_t* index_out) {
  if (n_out) *n_out = 0;
  if (had_invalid) *had_invalid = 0;
  return guarded([&] { scan_impl(s, freq, n, false, 1, out, cap, n_out, had_invalid, det_out, index_out); });
}

int storm_optimize_brackets(storm_solver* s, storm_bracket* brackets, size_t n, uint64_t ulp) {
  return guarded([&] {
    if (!s || (n && !brackets)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard g(s->device);
    if (s->scaled_scan) {
      optimize_scaled(s, brackets, n, ulp);
      return;
    }
    std::vector<BracketState> br(n);
    for (size_t k = 0; k < n; ++k) {
      const storm_bracket& b = brackets[k];
      // InverseQuadratic::optimize asserts finite end values of opposite sign (src/bracket.rs:159-166)
      if (!std::isfinite(b.lo_x) || !std::isfinite(b.hi_x) || !std::isfinite(b.lo_f) || !std::isfinite(b.hi_f))
        fail(STORM_ERR_BRACKET, "bracket end points must be finite");
      if (sign_of(b.lo_f) == sign_of(b.hi_f)) fail(STORM_ERR_BRACKET, "bracket end values must differ in sign");
      br[k].lower = Pt{b.lo_x, b.lo_f};
      br[k].upper = Pt{b.hi_x, b.hi_f};
    }
    if (s->device_brackets) {
      BracketStage stage = bracket_stage(s, 0, n);
      for (size_t k = 0; k < n; ++k) {
        BracketDev d{};
        d.lo_x = br[k].lower.x; d.lo_f = br[k].lower.f; d.hi_x = br[k].upper.x; d.hi_f = br[k].upper.f;
        stage.br[k] = d;
      }
      s->d_br.reserve(n, s->stream);
      s->d_bflags.reserve(2, s->stream);
      hip_check(hipMemsetAsync(s->d_bflags.p, 0, 2 * sizeof(int), s->stream), "flags memset");
      hip_check(hipMemcpyAsync(s->d_br.p, stage.br, n * sizeof(BracketDev), hipMemcpyHostToDevice, s->stream), "brackets H2D");
      refine_device(s, n, ulp, stage);
      download_brackets(s, n, stage, false);
      for (size_t k = 0; k < n; ++k)
        brackets[k] = storm_bracket{stage.br[k].lo_x, stage.br[k].lo_f, stage.br[k].hi_x, stage.br[k].hi_f, stage.br[k].root,
                                    stage.br[k].evals};
      return;
    }
    refine_all(s, br, ulp);
    for (size_t k = 0; k < n; ++k)
      brackets[k] = storm_bracket{br[k].lower.x, br[k].lower.f, br[k].upper.x, br[k].upper.f, br[k].root, br[k].evals};
  });
}

int storm_eigenvector_batch(storm_solver* s, const double* freq, size_t n, double* det_out, double* vec_out) {
  return guarded([&] {
    if (!s || (n && (!freq || !vec_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard g(s->device);
    const size_t len = (s->sys.n_int + 1) * 4;
    const size_t chunk = eigen_reserve(s, n);
    // A caller that keeps its result block in pinned (or registered) host memory gets the eigenvectors by DMA straight
    // into it; pageable memory goes through the library's pinned staging block and a host copy (7 MB for the 22 modes of
    // the bench headline: 0.3 ms of the step).
    const bool direct = host_pointer_is_pinned(vec_out);
    for (size_t off = 0; off < n; off += chunk) {
      const size_t m = std::min(chunk, n - off);
      eigen_chunk(s, freq + off, m);
      double* dst = vec_out + off * len;
      if (!direct) { stage_reserve(s, m * len); dst = s->h_vec.p; }
      hip_check(hipMemcpyAsync(dst, s->d_vec.p, m * len * sizeof(double), hipMemcpyDeviceToHost, s->stream), "vec D2H");
      if (det_out)
        hip_check(hipMemcpyAsync(det_out + off, s->d_det.p, m * sizeof(double), hipMemcpyDeviceToHost, s->stream), "det D2H");
      check_flags(s);
      if (!direct) std::memcpy(vec_out + off * len, s->h_vec.p, m * len * sizeof(double));
    }
  });
}

int storm_postproc_create(const storm_model_desc* d, storm_postproc** out) {
  return guarded([&] {
    if (!d || !out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const double* src[10] = {d->r_coord, d->m_coord, d->rho, d->p, d->v, d->u, d->gamma1, d->a_star, d->c1, d->rot};
    for (const double* q : src)
      if (!q) fail(STORM_ERR_INVALID_ARGUMENT, "null model array");
    const size_t np = d->n_pts;
    if (np < 2) fail(STORM_ERR_INVALID_ARGUMENT, "the model needs at least two points");
    if (d->r_coord[0] != 0.0) fail(STORM_ERR_INVALID_ARGUMENT, "r_coord[0] must be 0 (assert_eq! in Rotating1DPostprocessing::new)");
    std::unique_ptr<storm_postproc> p(new storm_postproc());
    hip_check(hipGetDevice(&p->device), "hipGetDevice (is a HIP device present?)");
    p->tab_elems = size_t(kPostPlanes) * np;
    p->h_model.reserve(p->tab_elems);
    double* tab = p->h_model.p;
    static_assert(PP_R == 0 && PP_ROT == 9 && PP_TRAP == 10, "plane order");
    for (int f = 0; f < 10; ++f) std::memcpy(tab + size_t(f) * np, src[f], np * sizeof(double));
    double* trap = tab + size_t(PP_TRAP) * np;  // src/postprocessing.rs:99-111
    const double* r = d->r_coord;
    trap[0] = 0.5 * (r[1] - r[0]);
    trap[np - 1] = 0.5 * (r[np - 1] - r[np - 2]);
    for (size_t i = 1; i + 1 < np; ++i) trap[i] = 0.5 * (r[i + 1] - r[i - 1]);
    p->model.tab = nullptr;
    p->model.n_pts = np;
    p->h_r.assign(d->r_coord, d->r_coord + np);
    p->h_rho.assign(d->rho, d->rho + np);
    p->h_rot.assign(d->rot, d->rot + np);
    p->h_trap.assign(trap, trap + np);
    *out = p.release();
  });
}

void storm_postproc_destroy(storm_postproc* p) {
  if (!p) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(p->device);
  if (p->ready) { (void)hipEventSynchronize(p->ready); (void)hipEventDestroy(p->ready); }
  if (p->sync_marker) (void)hipEventDestroy(p->sync_marker);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->stream) { (void)hipStreamSynchronize(p->stream); MemoryCache::get().give_stream(p->device, p->stream); }
  p->h_model.release();
  p->tab.release(); p->d_freq.release(); p->d_vec.release(); p->d_terms.release(); p->d_scale.release();
  p->d_fields.release(); p->d_nodes.release(); p->d_ids.release(); p->h_stage.release();
  p->d_weights.release(); p->d_consts.release(); p->d_out.release(); p->d_ell_index.release(); p->d_cflags.release(); p->d_order.release(); p->d_gather.release(); p->d_partial.release();
  (void)hipSetDevice(prev);
  delete p;
}

int storm_postproc_set_option(storm_postproc* p, const char* key, int64_t value) {
  return guarded([&] {
    if (!p || !key) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(p->mu);
    const std::string k(key);
    if (k == "arithmetic") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "arithmetic must be 0 (parity) or 1 (fast)");
      p->arith = int(value);
    } else if (k == "coupling_tile") {
      if (value != 0 && value != 8 && value != 16) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_tile must be 0 (auto), 8 or 16");
      p->cp_tile = int(value);
    } else if (k == "coupling_matrix") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_matrix must be 0 or 1");
      p->cp_matrix = int(value);
    } else if (k == "coupling_matrix_tile") {
      if (value != 0 && value != 16 && value != 32) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_matrix_tile must be 0 (auto), 16 or 32");
      p->cp_matrix_tile = int(value);
    } else if (k == "coupling_split") {
      if (value < 0 || value > 64) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_split must be 0 (automatic) or 1..64 pieces");
      p->cp_split = int(value);
    } else {
      fail(STORM_ERR_INVALID_ARGUMENT, "unknown option: " + k);
    }
  });
}

int storm_postprocess_batch(storm_postproc* p, uint64_t ell, int64_t m, const double* freq, size_t n, const double* vec,
                            double* fields_out, storm_mode_id* ids_out, double* nodes_out, size_t node_cap) {
  return guarded([&] {
    if (!p || (n && (!freq || !vec || !ids_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(p->mu);
    DeviceGuard g(p->device);
    const PostParams par = post_params(ell, m);
    hipStream_t st = post_stream(p);
    ensure_model(p, st);
    const size_t np = p->model.n_pts, len = np * 4;
    const size_t chunk = post_reserve(p, n, fields_out != nullptr, nodes_out ? node_cap : 0);
    p->d_freq.reserve(chunk, p->stream);
    p->d_vec.reserve(chunk * len, p->stream);
    for (size_t off = 0; off < n; off += chunk) {
      const size_t k = std::min(chunk, n - off);
      hip_check(hipMemcpyAsync(p->d_freq.p, freq + off, k * sizeof(double), hipMemcpyHostToDevice, st), "freq H2D");
      hip_check(hipMemcpyAsync(p->d_vec.p, vec + off * len, k * len * sizeof(double), hipMemcpyHostToDevice, st), "vec H2D");
      post_chunk(p, par, p->d_freq.p, p->d_vec.p, k, fields_out != nullptr, nodes_out ? node_cap : 0, st);
      post_download(p, k, fields_out ? fields_out + off * kPostFields * np : nullptr, ids_out + off,
                    nodes_out ? nodes_out + off * node_cap : nullptr, node_cap, st);
    }
  });
}

}  // extern "C"

namespace {
// Per-point weights of the coupling integrals: the mode-independent sub-expressions of src/perturbed.rs:88-108,157-252.
// `w`: (n_pts + 16) * kCouplingWeights doubles (16 rows of zero padding, never read since the kernel takes the weights
// through scalar loads; column CW_RX is left zero here and filled on the device).
void coupling_weights(const storm_postproc* p, int64_t m, const storm_metric_desc* metric, double* w) {
  const size_t np = p->model.n_pts;
  std::memset(w, 0, (np + 16) * kCouplingWeights * sizeof(double));
  const double mf = double(m);
  for (size_t i = 0; i < np; ++i) {
    double* u = w + i * kCouplingWeights;
    const double x = p->h_r[i], rho = p->h_rho[i], rot = p->h_rot[i];
    const double ea = metric->alpha[i], adea = metric->dalpha[i], eb = metric->beta[i], adeb = metric->dbeta[i];
    u[CW_TR2] = p->h_trap[i] * (x * x);
    u[CW_VOL] = p->h_trap[i] * (x * x) * rho;
    u[CW_VOL2] = u[CW_VOL] * 2.;
    u[CW_VOL2ROT] = u[CW_VOL] * 2. * rot;
    u[CW_C1] = 2. * mf * rot;
    u[CW_C2] = mf * mf * rot * rot;
    u[CW_MROT] = mf * rot;
    u[CW_RHO] = rho; u[CW_X] = x;
    u[CW_U1] = 1. + 2. * (ea + adea);
    u[CW_U2] = 1. + 2. * ea;
    u[CW_U3] = eb + adeb;
    u[CW_U4] = 1. + 2. * ea + adea;
    u[CW_U5] = 2. * eb + adeb;
    u[CW_EB] = eb;
    u[CW_TEA] = 3. * ea + adea;
    u[CW_ADTEA] = 3. * metric->dalpha[i] + metric->ddalpha[i];
    u[CW_TEB] = 3. * eb + adeb;
    u[CW_ADTEB] = 3. * metric->dbeta[i] + metric->ddbeta[i];
  }
}

// Gaunt factors per pair of distinct degrees (src/perturbed.rs:126-146): ell_index[k] = position of ell[k] among the
// distinct degrees, consts = [n_ell][n_ell][kCouplingConsts].  Returns n_ell.
size_t coupling_consts(size_t n, const uint64_t* ell, int64_t m, int* ell_index, std::vector<double>& consts) {
  std::vector<uint64_t> ells;
  for (size_t k = 0; k < n; ++k) {
    size_t at = 0;
    while (at < ells.size() && ells[at] != ell[k]) ++at;
    if (at == ells.size()) ells.push_back(ell[k]);
    ell_index[k] = int(at);
  }
  const size_t ne = ells.size();
  consts.assign(ne * ne * kCouplingConsts, 0.0);
  for (size_t a = 0; a < ne; ++a)
    for (size_t b = 0; b < ne; ++b) {
      double* q = consts.data() + (a * ne + b) * kCouplingConsts;
      const uint64_t le = ells[a], re = ells[b];
      q[CQ_Q2] = gaunt_q_kl2(le, re, m);
      q[CQ_Q2_HRD] = gaunt_q_kl2_hd(le, re, m);
      q[CQ_Q2_HLD] = gaunt_q_kl2_hd(re, le, m);
      q[CQ_Q2_H] = gaunt_q_kl2_h(le, re, m);
      if (le == re) {
        const uint64_t n1 = re == 0 ? 0 : re - 1;
        q[CQ_RT_P] = gaunt_q_kl1_hd(le, re + 1, m);
        q[CQ_RT_N] = gaunt_q_kl1_hd(le, n1, m);
        q[CQ_HT_P] = -gaunt_q_kl1_h(le, re + 1, m);
        q[CQ_HT_N] = -gaunt_q_kl1_h(le, n1, m);
        q[CQ_IPR] = 1.;
        q[CQ_IPH] = double(re * (re + 1));
      }
    }
  return ne;
}

// The kernel's shortcut for tiles whose Gaunt factors are all zero (coupling_kernels.hip, coupling_body) needs finite
// operands of moderate size: the weights are checked here, the modes on the device.
// The ordered mode list of the matrix-pipe walk (FMA build; CouplingArgs::order): the modes grouped by degree in order of
// first appearance, the caller's order kept inside a group, every group padded with -1 to a multiple of 16 positions.
inline void coupling_order(size_t n, const int* ell_index, size_t n_ell, std::vector<int>& order) {
  order.clear();
  for (size_t e = 0; e < n_ell; ++e) {
    for (size_t k = 0; k < n; ++k)
      if (size_t(ell_index[k]) == e) order.push_back(int(k));
    while (order.size() % 16) order.push_back(-1);
  }
}

// Tile of the coupling kernel when the handle's "coupling_tile" is 0: 16 (four waves share every staged chunk: half the
// L2 fetches) once the matrix has enough pairs to fill the card either way, 8 below (more, smaller workgroups).
inline int coupling_tile_auto(size_t max_modes) { return max_modes >= 192 ? 16 : 8; }

bool coupling_weights_tame(const double* w, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!(std::fabs(w[i]) < 0x1p200)) return false;
  return true;
}

// Is there a pair of degrees the kernel has a shorter walk for (different degrees: inner products and toroidal factors
// are zero; more than 2 apart: all ten)?
bool coupling_has_sparse_pair(const std::vector<double>& consts) {
  for (size_t at = 0; at + kCouplingConsts <= consts.size(); at += kCouplingConsts)
    if (consts[at + CQ_IPR] == 0.0 && consts[at + CQ_IPH] == 0.0) return true;
  return false;
}

// f(j) for j < n on up to `max_threads` host threads (the caller's included).  For host work that is bound by page
// faults or by one core: the result blocks of a batch of coupling problems (~90 MB into fresh pages), their weights.
// f must not throw.
template <class F>
void parallel_for(size_t n, size_t max_threads, F&& f) {
  const size_t n_thr = std::min(n, max_threads);
  if (n_thr <= 1) {
    for (size_t j = 0; j < n; ++j) f(j);
    return;
  }
  std::atomic<size_t> next{0};
  auto work = [&] { for (size_t j; (j = next.fetch_add(1)) < n;) f(j); };
  std::vector<std::thread> pool;
  for (size_t k = 1; k < n_thr; ++k) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

void check_metric(const storm_metric_desc* metric) {
  if (!metric) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
  const double* ms[6] = {metric->alpha, metric->dalpha, metric->ddalpha, metric->beta, metric->dbeta, metric->ddbeta};
  for (const double* q : ms)
    if (!q) fail(STORM_ERR_INVALID_ARGUMENT, "null metric array");
}

// The coupling integrals of n modes whose 13 fields `provide_fields` puts into p->d_fields ([n][13][n_pts]) on the
// given stream: an upload of host arrays (storm_coupling_matrices) or a device-side gather of fields that never left
// HBM (storm_sweep_coupling_matrices).  Caller holds no lock; arguments are validated here.
template <class ProvideFields>
void coupling_run(storm_postproc* p, size_t n, const uint64_t* ell, const double* freq, int64_t m,
                  const storm_metric_desc* metric, double* out, ProvideFields&& provide_fields) {
    if (!p || (n && (!ell || !freq || !metric || !out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return;
    const double* ms[6] = {metric->alpha, metric->dalpha, metric->ddalpha, metric->beta, metric->dbeta, metric->ddbeta};
    for (const double* q : ms)
      if (!q) fail(STORM_ERR_INVALID_ARGUMENT, "null metric array");
    if (n > 8 * size_t(65535)) fail(STORM_ERR_INVALID_ARGUMENT, "too many modes");
    std::lock_guard<std::mutex> lock(p->mu);
    DeviceGuard g(p->device);
    const size_t np = p->model.n_pts;
    std::vector<double> w((np + 16) * kCouplingWeights);
    coupling_weights(p, m, metric, w.data());
    std::vector<int> ell_index(n);
    std::vector<double> consts;
    const size_t ne = coupling_consts(n, ell, m, ell_index.data(), consts);
    const size_t nf = n * kPostFields * np;
    p->d_fields.reserve(nf, p->stream);
    p->d_freq.reserve(n, p->stream);
    p->d_weights.reserve(w.size(), p->stream);
    p->d_consts.reserve(consts.size(), p->stream);
    p->d_ell_index.reserve(n, p->stream);
    p->d_out.reserve(size_t(kCouplingMatrices) * n * n, p->stream);
    p->d_cflags.reserve(1 + n, p->stream);  // [0]: non-finite flag, [1..]: per-mode "tame" flags (kernel scratch)
    std::vector<int> order;
    if (p->arith == 1 && p->cp_matrix) coupling_order(n, ell_index.data(), ne, order);
    p->d_order.reserve(order.size() + 1, p->stream);
    hipStream_t st = post_stream(p);
    provide_fields(p->d_fields.p, st);
    hip_check(hipMemcpyAsync(p->d_freq.p, freq, n * sizeof(double), hipMemcpyHostToDevice, st), "freq H2D");
    hip_check(hipMemcpyAsync(p->d_weights.p, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, st), "weights H2D");
    hip_check(hipMemcpyAsync(p->d_consts.p, consts.data(), consts.size() * sizeof(double), hipMemcpyHostToDevice, st), "consts H2D");
    hip_check(hipMemcpyAsync(p->d_ell_index.p, ell_index.data(), n * sizeof(int), hipMemcpyHostToDevice, st), "ell H2D");
    hip_check(hipMemsetAsync(p->d_cflags.p, 0, sizeof(int), st), "flags memset");
    CouplingArgs a;
    a.n_pts = np; a.n_modes = n; a.fields = p->d_fields.p; a.freq = p->d_freq.p; a.ell_index = p->d_ell_index.p;
    a.consts = p->d_consts.p; a.n_ell = int(ne); a.weights = p->d_weights.p; a.out = p->d_out.p; a.flags = p->d_cflags.p;
    a.rows = nullptr; a.stream = st;
    int matrix_tile = p->cp_matrix_tile;
    if (!order.empty()) {
      hip_check(hipMemcpyAsync(p->d_order.p, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, st), "order H2D");
      a.order = p->d_order.p; a.n_pad = order.size();
      const storm::CouplingPlan plan = storm::plan_coupling_matrix(a.n_pad, np, p->cp_matrix_tile, p->cp_split);
      matrix_tile = plan.tile;
      a.n_split = plan.n_split;
      if (plan.n_split > 1) {
        p->d_partial.reserve(size_t(plan.n_split) * kCouplingMatrices * n * n, p->stream);
        a.partial = p->d_partial.p;
        a.partial_stride = size_t(kCouplingMatrices) * n * n;
      }
    }
    if (coupling_has_sparse_pair(consts)) {
      a.tame = p->d_cflags.p + 1;
      a.weights_tame = coupling_weights_tame(w.data(), np * kCouplingWeights) ? 1 : 0;
    }
    if (!p->ev0) hip_check(hipEventCreate(&p->ev0), "hipEventCreate");
    if (!p->ev1) hip_check(hipEventCreate(&p->ev1), "hipEventCreate");
    hip_check(hipEventRecord(p->ev0, st), "hipEventRecord");
    hip_check((p->arith ? storm_fast::launch_coupling : storm::launch_coupling)(a, p->cp_tile ? p->cp_tile : coupling_tile_auto(n), matrix_tile),
              "coupling launch");
    hip_check(hipEventRecord(p->ev1, st), "hipEventRecord");
    // results through pinned memory (a copy to pageable memory takes the runtime's staged, host-synchronous path)
    const size_t n_out = size_t(kCouplingMatrices) * n * n;
    p->h_stage.reserve(n_out + 1);
    int* flags_ptr = reinterpret_cast<int*>(p->h_stage.p + n_out);
    hip_check(hipMemcpyAsync(p->h_stage.p, p->d_out.p, n_out * sizeof(double), hipMemcpyDeviceToHost, st), "out D2H");
    hip_check(hipMemcpyAsync(flags_ptr, p->d_cflags.p, sizeof(int), hipMemcpyDeviceToHost, st), "flags D2H");
    sync_stream(st, p->sync_marker);
    std::memcpy(out, p->h_stage.p, n_out * sizeof(double));
    const int flags = *flags_ptr;
    float kernel_ms = 0.f;
    (void)hipEventElapsedTime(&kernel_ms, p->ev0, p->ev1);
    p->last_coupling_ms = double(kernel_ms);
    // The mode fields are by far the largest buffer of the handle (n x 13 x n_pts) and a rotation sweep has one handle
    // per rotation rate: hand the block back to the cache so the next handle's call reuses it instead of allocating.
    p->d_fields.release();
    if (flags & 1) fail(STORM_ERR_NON_FINITE, "Not finite number in a coupling integral (the reference asserts, src/perturbed.rs:165-260)");
}
}  // namespace

extern "C" {

int storm_coupling_matrices(storm_postproc* p, size_t n, const uint64_t* ell, const double* freq, const double* fields,
                            int64_t m, const storm_metric_desc* metric, double* out) {
  return guarded([&] {
    if (n && !fields) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    coupling_run(p, n, ell, freq, m, metric, out, [&](double* d_fields, hipStream_t st) {
      hip_check(hipMemcpyAsync(d_fields, fields, n * kPostFields * p->model.n_pts * sizeof(double), hipMemcpyHostToDevice, st),
                "fields H2D");
    });
  });
}

int storm_memory_cache_trim(void) {
  return guarded([&] { MemoryCache::get().trim(); });
}
int storm_memory_cache_limit(size_t device_bytes, size_t pinned_host_bytes) {
  return guarded([&] {
    MemoryCache::get().set_limit(false, device_bytes);
    MemoryCache::get().set_limit(true, pinned_host_bytes);
  });
}
void storm_memory_cache_stats(size_t* device_bytes, size_t* pinned_host_bytes) {
  if (device_bytes) *device_bytes = MemoryCache::get().cached(false);
  if (pinned_host_bytes) *pinned_host_bytes = MemoryCache::get().cached(true);
}

double storm_postproc_last_kernel_ms(const storm_postproc* p) { return p ? p->last_coupling_ms : 0.0; }

int storm_modes_batch(storm_solver* s, storm_postproc* p, const double* freq, size_t n, double* det_out, double* vec_out,
                      double* fields_out, storm_mode_id* ids_out, double* nodes_out, size_t node_cap) {
  return guarded([&] {
    if (!s || !p || (n && (!freq || !ids_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (s->sys.n_int + 1 != p->model.n_pts)
      fail(STORM_ERR_INVALID_ARGUMENT, "the solver grid is not the post-processing model's grid (n_int + 1 != n_pts)");
    if (s->device != p->device) fail(STORM_ERR_INVALID_ARGUMENT, "solver and post-processing handle live on different devices");
    if (n == 0) return;
    std::lock_guard<std::mutex> lock_s(s->mu);  // lock order: solver, then post-processing handle
    std::lock_guard<std::mutex> lock_p(p->mu);
    DeviceGuard g(s->device);
    const PostParams par = post_params(s->ell, s->m);
    ensure_model(p, s->stream);
    const size_t np = p->model.n_pts, len = np * 4;
    size_t chunk = std::min(eigen_reserve(s, n), post_reserve(p, n, fields_out != nullptr, nodes_out ? node_cap : 0));
    for (size_t off = 0; off < n; off += chunk) {
      const size_t k = std::min(chunk, n - off);
      eigen_chunk(s, freq + off, k);
      ScopedTimer tp(s, s->stream, STORM_KERNEL_POSTPROCESS);
      post_chunk(p, par, s->d_freq.p, s->d_vec.p, k, fields_out != nullptr, nodes_out ? node_cap : 0, s->stream);
      tp.stop();
      count_launch(s, fields_out ? 3 : 2);
      if (vec_out) {
        stage_reserve(s, k * len);
        hip_check(hipMemcpyAsync(s->h_vec.p, s->d_vec.p, k * len * sizeof(double), hipMemcpyDeviceToHost, s->stream), "vec D2H");
      }
      if (det_out)
        hip_check(hipMemcpyAsync(det_out + off, s->d_det.p, k * sizeof(double), hipMemcpyDeviceToHost, s->stream), "det D2H");
      post_download(p, k, fields_out ? fields_out + off * kPostFields * np : nullptr, ids_out + off,
                    nodes_out ? nodes_out + off * node_cap : nullptr, node_cap, s->stream);
      check_flags(s);  // synchronises the stream
      if (vec_out) std::memcpy(vec_out + off * len, s->h_vec.p, k * len * sizeof(double));
    }
  });
}

}  // extern "C"

// =====================================================================================================================
// storm_multi: one scan over several devices from one process; see include/storm_hip.h.
// =====================================================================================================================
struct storm_multi {
  std::vector<storm_solver*> solvers;
  std::vector<int> devices;
  std::mutex mu;
  bool deal = false;   // option "deal": 1 = brackets dealt cyclically over the devices for refinement (hostlogic/multi.hpp)
};

namespace {
// The single-device C ABI as the device side of hostlogic/multi.hpp (orchestration, gather, deal).
struct SolverOps {
  using Solver = storm_solver*;
  static int scan_brackets(Solver s, const double* freq, size_t n, storm_bracket* out, size_t cap, size_t* n_out, int* invalid,
                           double* det_out, uint64_t* index_out) {
    return storm_scan_brackets(s, freq, n, out, cap, n_out, invalid, det_out, index_out);
  }
  static int optimize_brackets(Solver s, storm_bracket* br, size_t n, uint64_t ulp) { return storm_optimize_brackets(s, br, n, ulp); }
  static int scan_and_optimize(Solver s, const double* freq, size_t n, uint64_t ulp, storm_bracket* out, size_t cap, size_t* n_out,
                               int* invalid, double* det_out, uint64_t* index_out) {
    return storm_scan_and_optimize(s, freq, n, ulp, out, cap, n_out, invalid, det_out, index_out);
  }
  static int eigenvector_batch(Solver s, const double* freq, size_t n, double* det_out, double* vec_out) {
    return storm_eigenvector_batch(s, freq, n, det_out, vec_out);
  }
  static size_t n_int(Solver s) { return s->sys.n_int; }
  static std::string last_error() { return g_error; }
};
}  // namespace

extern "C" {

void storm_shard_bounds(size_t n, size_t n_parts, size_t k, size_t* lo, size_t* hi) {
  size_t a, b;
  shard_bounds(n, n_parts, k, a, b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

int storm_multi_create(const storm_system_desc* desc, const int* devices, size_t n_dev, storm_multi** out) {
  if (out) *out = nullptr;
  storm_multi* q = nullptr;
  int rc = guarded([&] {
    if (!desc || !devices || !out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n_dev == 0 || n_dev > 1024) fail(STORM_ERR_INVALID_ARGUMENT, "a storm_multi needs 1..1024 devices");
    int ndev = 0;
    hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    for (size_t k = 0; k < n_dev; ++k)
      if (devices[k] < 0 || devices[k] >= ndev) fail(STORM_ERR_INVALID_ARGUMENT, "no such HIP device: " + std::to_string(devices[k]));
    q = new storm_multi();
    q->devices.assign(devices, devices + n_dev);
    q->solvers.assign(n_dev, nullptr);
    on_every_device(n_dev, [&](size_t k) {
      int prev = 0;
      if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(q->devices[k]) != hipSuccess) { g_error = "hipSetDevice failed"; return int(STORM_ERR_HIP); }
      const int r = storm_solver_create(desc, &q->solvers[k]);
      (void)hipSetDevice(prev);
      return r;
    }, SolverOps::last_error);
    *out = q;
  });
  if (rc != STORM_OK && q) { storm_multi_destroy(q); if (out) *out = nullptr; }
  return rc;
}

void storm_multi_destroy(storm_multi* q) {
  if (!q) return;
  for (storm_solver* s : q->solvers) storm_solver_destroy(s);
  delete q;
}

size_t storm_multi_devices(const storm_multi* q) { return q ? q->solvers.size() : 0; }

int storm_multi_set_option(storm_multi* q, const char* key, int64_t value) {
  return guarded([&] {
    if (!q) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    if (key && std::string(key) == "deal") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "deal must be 0 (refined where found) or 1 (dealt cyclically)");
      q->deal = value != 0;
      return;
    }
    for (storm_solver* s : q->solvers) {
      const int rc = storm_solver_set_option(s, key, value);
      if (rc != STORM_OK) fail(rc, g_error);
    }
  });
}

int storm_multi_stats(storm_multi* q, storm_stats* out, int reset) {
  return guarded([&] {
    if (!q || !out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    *out = storm_stats{};
    for (storm_solver* s : q->solvers) {
      storm_stats one{};
      const int rc = storm_solver_stats(s, &one, reset);
      if (rc != STORM_OK) fail(rc, g_error);
      out->det_evals += one.det_evals; out->refine_evals += one.refine_evals; out->eigenvectors += one.eigenvectors;
      out->launches += one.launches; out->segment_fallbacks += one.segment_fallbacks;
      out->refine_rounds = std::max(out->refine_rounds, one.refine_rounds);  // the devices' rounds run side by side
    }
  });
}

int storm_multi_scan_and_optimize(storm_multi* q, const double* freq, size_t n, uint64_t ulp, storm_bracket* out, size_t cap,
                                  size_t* n_out, int* had_invalid, double* det_out, uint64_t* index_out) {
  if (n_out) *n_out = 0;
  if (had_invalid) *had_invalid = 0;
  return guarded([&] {
    if (!q || !n_out || (n && !freq) || (cap && !out)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    multi_scan_and_optimize<SolverOps>(q->solvers, freq, n, ulp, out, cap, n_out, had_invalid, det_out, index_out, q->deal);
  });
}

int storm_multi_eigenvector_batch(storm_multi* q, const double* freq, size_t n, double* det_out, double* vec_out) {
  return guarded([&] {
    if (!q || (n && (!freq || !vec_out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    multi_eigenvector_batch<SolverOps>(q->solvers, freq, n, det_out, vec_out);
  });
}

}  // extern "C"

// =====================================================================================================================
// storm_sweep: several systems per launch.
//
// The reference's batch inputs sweep ~10^3 rotation rates x 8 (ell, m) systems with scans of only 100-300 frequencies
// each (scripts/generate-batch-input-deformed.py:19-35; every `scan` line builds one solver and calls
// ErasedSolver::scan_and_optimize, src/main.rs:815-831).  One such scan is latency-bound on the GPU (a handful of
// waves, ~10 dependent refinement rounds, one wave per eigenvector), so the set runs S systems in lock-step: one
// discriminant launch over every (system, frequency) pair, the bracket refinement rounds of all systems merged, and one
// eigenvector launch over every mode of every system.  Each (system, frequency) evaluation runs exactly the code a
// single-system handle runs, so results are bit-identical to S separate handles.
// =====================================================================================================================
struct storm_sweep {
  int device = 0, scheme = 0;
  size_t n_sys = 0, n_int = 0, p = 0;
  // Kernels are compiled per (ell == 0, uniform rotation) class, so systems are grouped by class internally:
  // internal position k holds the caller's system order[k]; class c = positions [cls_begin[c], cls_begin[c+1]).
  std::vector<size_t> order;
  size_t cls_begin[4] = {0, 0, 0, 0};
  std::vector<SystemDev> h_table;          // internal order
  std::vector<uint64_t> ell;               // caller order
  std::vector<int64_t> m;
  DevBuf<double> tab;                      // planes of all systems, internal order
  DevBuf<SystemDev> d_table;
  hipStream_t stream = nullptr;
  hipEvent_t sync_marker = nullptr;
  DevBuf<double> d_freq, d_det, d_steps, d_upper, d_vec;
  DevBuf<int> d_sys, d_swaps, d_flags;
  DevBuf<double> d_terms, d_scale, d_fields, d_nodes;  // post-processing scratch (storm_sweep_modes)
  DevBuf<long long> d_ids;
  DevBuf<PostSystem> d_post_table;                     // (model, ell, m) of every system, internal order
  // storm_sweep_modes_resident: the 13 fields of every mode of that call stay in d_fields (internal order);
  // resident_at[i] = row of the mode the caller passed i-th.  Any other modes call drops them.
  std::vector<long long> resident_at;
  DevBuf<double> d_cp;        // batched coupling integrals: weights, consts, frequencies, results of every problem
  DevBuf<double> d_cp_partial;  // ... and the sums of the pieces of a split walk, [piece][results of every problem]
  DevBuf<long long> d_cp_idx; // rows of the resident fields
  DevBuf<int> d_cp_ell, d_cp_flags, d_cp_order;
  DevBuf<CouplingArgs> d_cp_table;
  PinnedBuf h_cp;
  PinnedBuf h_in, h_out, h_flags;
  // segment-parallel eigenvectors (FMA build): transfer matrices, boundary vectors, growth guard, the systems with
  // n_int = number of segments for the boundary elimination
  DevBuf<double> d_partials, d_seg_nodes;
  DevBuf<unsigned long long> d_growth;
  DevBuf<SystemDev> d_table_red;
  std::vector<SystemDev> h_table_red;
  PinnedBuf h_growth;
  int segments = 0;           // option "segments": -1 = never, 0 = automatic (FMA build only)
  double seg_growth_max = 0x1p24;
  int arith = 0, hex_lanes = 0, hex_rounds = 1, eigen_layout = 0, cp_tile = 0, cp_matrix = 1, cp_matrix_tile = 0, cp_split = 0;
  // (A sweep refines its brackets with the host state machine, set_refine: it evaluates only the brackets still open.  A
  // device version over all bracket slots every round existed in round 2 behind an option and was removed in round 3: 64
  // systems x 300 frequencies, 1688 brackets, 7 rounds: 17.8 ms per scan on the host, 20.2 ms on the device -- the late rounds
  // have a handful of open brackets, and compacting them on the device would need the per-system, per-class padded
  // layout of the discriminant launch rebuilt every round, i.e. the counts on the host, i.e. the round trip it was
  // meant to save.  A single handle's brackets do run on the device: refine_device.)
  size_t steps_bytes_max = size_t(16) << 30;  // eigenvector staging per chunk (2 MB per mode at 6199 intervals)
  storm_stats stats{};
  std::mutex mu;
};

namespace {

constexpr int kSetClasses = 3;  // 0: ell == 0, 1: ell != 0 with bitwise-uniform m*rot, 2: general
constexpr size_t kSetPad = 4;   // det_hex: an aligned group of 4 frequencies (one wave at 16 lanes each) is one system's
// FMA build: class batches up to this size take the segment path (set_det).  The card holds ~12 288 sixteen-lane groups
// at once (one det_hex wave round): cutting chains only pays while a batch has fewer chains than that (measured: 7476
// frequencies in two segments each 3.1 ms, 5420 in two segments 2.3 ms, as whole chains in det_hex 1.85 / 1.45 ms).
constexpr size_t kSetSegMaxBatch = 3072, kSetSegGroups = 12288;

inline bool set_class_ell0(int c) { return c == 0; }
inline bool set_class_uniform(int c) { return c != 2; }
// segment-parallel evaluation (FMA build only: it re-associates the ordered product)
inline bool set_uses_segments(const storm_sweep* q) { return q->arith == 1 && q->segments >= 0 && q->n_int >= 64; }

void set_check_flags(storm_sweep* q) {
  q->h_flags.reserve(1);
  int* flags_ptr = reinterpret_cast<int*>(q->h_flags.p);
  hip_check(hipMemcpyAsync(flags_ptr, q->d_flags.p, sizeof(int), hipMemcpyDeviceToHost, q->stream), "flags D2H");
  sync_stream(q->stream, q->sync_marker);
  if (*flags_ptr & 1) {
    hip_check(hipMemsetAsync(q->d_flags.p, 0, sizeof(int), q->stream), "flags reset");
    fail(STORM_ERR_SINGULAR, "singular collocation stage matrix (the reference would panic in try_inverse().unwrap())");
  }
}

LaunchArgs set_args(storm_sweep* q, int cls, size_t first, size_t n, double* out) {
  LaunchArgs a;
  a.sys = q->h_table[q->cls_begin[cls]];
  a.freq = q->d_freq.p + first; a.n = n; a.out = out; a.flags = q->d_flags.p;
  a.ell0 = set_class_ell0(cls); a.uniform = set_class_uniform(cls); a.hex_lanes = q->hex_lanes; a.hex_rounds_per_launch = q->hex_rounds; a.stream = q->stream;
  a.table = q->d_table.p; a.sys_index = q->d_sys.p + first;
  return a;
}

void set_det_launch(storm_sweep* q, const std::vector<size_t>& poff);

// Discriminants of one frequency list per system.  x[k], cnt[k], f[k] belong to INTERNAL position k.
void set_det(storm_sweep* q, const std::vector<const double*>& x, const std::vector<size_t>& cnt,
             const std::vector<double*>& f) {
  const size_t S = q->n_sys;
  std::vector<size_t> poff(S + 1, 0);
  for (size_t k = 0; k < S; ++k) poff[k + 1] = poff[k] + (cnt[k] + kSetPad - 1) / kSetPad * kSetPad;
  const size_t total = poff[S];
  if (total == 0) return;
  if (total > (size_t(1) << 31)) fail(STORM_ERR_INVALID_ARGUMENT, "too many frequencies in one call");
  q->h_in.reserve(total + (total + 1) / 2);
  q->h_out.reserve(total);
  double* hx = q->h_in.p;
  int* hs = reinterpret_cast<int*>(q->h_in.p + total);
  for (size_t k = 0; k < S; ++k) {
    if (!cnt[k]) continue;
    std::memcpy(hx + poff[k], x[k], cnt[k] * sizeof(double));
    for (size_t i = poff[k] + cnt[k]; i < poff[k + 1]; ++i) hx[i] = x[k][cnt[k] - 1];  // padding repeats the last one
    for (size_t i = poff[k]; i < poff[k + 1]; ++i) hs[i] = int(k);
  }
  q->d_freq.reserve(total, q->stream); q->d_det.reserve(total, q->stream); q->d_sys.reserve(total, q->stream);
  hip_check(hipMemcpyAsync(q->d_freq.p, hx, total * sizeof(double), hipMemcpyHostToDevice, q->stream), "freq H2D");
  hip_check(hipMemcpyAsync(q->d_sys.p, hs, total * sizeof(int), hipMemcpyHostToDevice, q->stream), "system index H2D");
  set_det_launch(q, poff);
  hip_check(hipMemcpyAsync(q->h_out.p, q->d_det.p, total * sizeof(double), hipMemcpyDeviceToHost, q->stream), "det D2H");
  set_check_flags(q);  // synchronises
  for (size_t k = 0; k < S; ++k) {
    if (cnt[k]) std::memcpy(f[k], q->h_out.p + poff[k], cnt[k] * sizeof(double));
    q->stats.det_evals += cnt[k];
  }
}

// The discriminant launches of one evaluation of d_freq (frequencies) / d_sys (system positions) into d_det: per class of
// systems the slots [poff[cls_begin[c]], poff[cls_begin[c + 1]]), every system's share padded to kSetPad.
void set_det_launch(storm_sweep* q, const std::vector<size_t>& poff) {
  for (int c = 0; c < kSetClasses; ++c) {
    const size_t first = poff[q->cls_begin[c]], n = poff[q->cls_begin[c + 1]] - first;
    if (!n) continue;
    // FMA build, batches that leave the card idle as one chain per frequency (refinement rounds: a few thousand
    // brackets at most): segment transfer matrices in one launch, then the ordered product + boundary elimination over
    // the segments with every frequency's own boundary rows (the single-handle version is det_segments)
    if (set_uses_segments(q) && n <= kSetSegMaxBatch) {
      const size_t n_int = q->n_int;
      const size_t want = std::max<size_t>(2, kSetSegGroups / n);
      size_t seg_len = ((n_int + want - 1) / want + 15) / 16 * 16;
      seg_len = std::max<size_t>(seg_len, 32);
      const size_t n_seg = (1 + (n_int - 1 + seg_len - 1) / seg_len + 3) / 4 * 4;  // the first interval is a segment of its own
      q->d_partials.reserve(n * n_seg * 16, q->stream);
      LaunchArgs a = set_args(q, c, first, n, q->d_partials.p);
      a.n_seg = int(n_seg); a.seg_len = int(seg_len);
      hip_check(g_launch_seg_product[q->arith][q->scheme](a), "seg_product launch");
      SystemDev reduced = q->h_table[q->cls_begin[c]];
      reduced.n_int = n_seg;
      hip_check(storm_fast::launch_chain(reduced, set_class_ell0(c), q->d_freq.p + first, n, q->d_partials.p, q->d_det.p + first,
                                         q->stream, q->d_table.p, q->d_sys.p + first), "chain launch");
      q->stats.launches += 2;
      continue;
    }
    hip_check(g_launch_hex[q->arith][q->scheme](set_args(q, c, first, n, q->d_det.p + first)), "det_hex launch");
    q->stats.launches += 1;
  }
}

struct SetBracket : BracketState { size_t pos = 0; };  // pos: internal position of the system

// refine_all over the brackets of every system (sorted by system position): one merged evaluation per round.
void set_refine(storm_sweep* q, std::vector<SetBracket>& br, uint64_t ulp) {
  if (ulp == 0) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be >= 1");
  if (!(ulp < (uint64_t(1) << 54))) fail(STORM_ERR_INVALID_ARGUMENT, "ulp precision must be < 2^54");
  const size_t S = q->n_sys;
  std::vector<double> xs, fs;
  std::vector<size_t> who, cnt(S);
  std::vector<const double*> xp(S);
  std::vector<double*> fp(S);
  for (;;) {
    xs.clear(); who.clear();
    std::fill(cnt.begin(), cnt.end(), size_t(0));
    for (size_t k = 0; k < br.size(); ++k) {
      SetBracket& b = br[k];
      if (b.done) continue;
      double x;
      if (!b.has_prev) {
        x = secant(b.lower, b.upper);
      } else {
        x = inverse_quadratic(b.lower, b.upper, b.previous);
        if (x <= b.lower.x || x >= b.upper.x) x = secant(b.lower, b.upper);
      }
      if (clamp_proposal(b.upper.x, b.lower.x, x, ulp)) {
        b.done = true; b.root = x;
        continue;
      }
      b.pending = x;
      xs.push_back(x); who.push_back(k);
      cnt[b.pos] += 1;
    }
    if (xs.empty()) break;
    fs.resize(xs.size());
    size_t at = 0;
    for (size_t k = 0; k < S; ++k) { xp[k] = xs.data() + at; fp[k] = fs.data() + at; at += cnt[k]; }
    set_det(q, xp, cnt, fp);
    q->stats.refine_evals += xs.size();
    q->stats.refine_rounds += 1;
    for (size_t i = 0; i < who.size(); ++i) {
      SetBracket& b = br[who[i]];
      const Pt next{xs[i], fs[i]};
      if (!std::isfinite(next.x) || !std::isfinite(next.f))
        fail(STORM_ERR_NON_FINITE, "non-finite discriminant during bracket refinement");
      b.evals += 1;
      if (sign_of(b.upper.f) == sign_of(next.f)) { b.previous = b.upper; b.upper = next; }
      else { b.previous = b.lower; b.lower = next; }
      b.has_prev = true;
    }
  }
}

size_t set_eigen_reserve(storm_sweep* q, size_t n) {
  const size_t n_int = q->n_int, len = (n_int + 1) * 4, ustride = n_int * 22 + 6;
  const size_t per_mode = (n_int * 16 + ustride + len) * sizeof(double);
  size_t chunk = q->steps_bytes_max / per_mode;
  if (chunk == 0) chunk = 1;
  if (chunk > n) chunk = n;
  if (chunk > 65535) chunk = 65535;  // grid.y of build_steps
  q->d_freq.reserve(chunk, q->stream); q->d_det.reserve(chunk, q->stream); q->d_sys.reserve(chunk, q->stream);
  q->d_vec.reserve((chunk + 64) * len, q->stream);
  q->d_swaps.reserve(chunk * 2, q->stream);
  // (step matrices and factors of the sequential elimination, 2 MB per mode: reserved where that path is taken)
  return chunk;
}

// Segment-parallel eigenvectors of one chunk (FMA build; the single-handle version is eigen_segments): transfer matrices
// of every (mode, segment) pair, the reference's elimination + back-substitution on the system of the segment boundaries
// (every mode with its own system's boundary rows: the table with n_int = n_seg), then the fill of every segment.  No
// step matrices are staged (2 MB per mode in the sequential path; here 128 B per (mode, segment)).  Returns false if a
// transfer matrix grew past the guard: the caller then takes the sequential path for the chunk.
bool set_eigen_segments(storm_sweep* q, const int* pos, size_t m) {
  const size_t n_int = q->n_int, len = (n_int + 1) * 4, S = q->n_sys;
  const size_t want = std::max<size_t>(1, 4096 / m);
  size_t seg_len = ((n_int + want - 1) / want + 15) / 16 * 16;
  seg_len = std::min<size_t>(std::max<size_t>(seg_len, 32), 256);
  const size_t n_seg = (1 + (n_int - 1 + seg_len - 1) / seg_len + 3) / 4 * 4;  // the first interval is a segment of its own
  if (m * n_seg / 4 >= (size_t(1) << 31)) return false;
  const size_t ustride = n_seg * 22 + 6, nstride = (n_seg + 1) * 4;
  q->d_partials.reserve(m * n_seg * 16, q->stream);
  q->d_seg_nodes.reserve((m + 64) * nstride, q->stream);
  q->d_growth.reserve(m, q->stream);
  q->d_upper.reserve((m + 64) * ustride, q->stream);
  q->d_table_red.reserve(S, q->stream);
  q->h_table_red = q->h_table;
  for (SystemDev& t : q->h_table_red) t.n_int = n_seg;
  hip_check(hipMemcpyAsync(q->d_table_red.p, q->h_table_red.data(), S * sizeof(SystemDev), hipMemcpyHostToDevice, q->stream), "table H2D");
  hip_check(hipMemsetAsync(q->d_growth.p, 0, m * sizeof(unsigned long long), q->stream), "growth memset");
  size_t first = 0;
  std::vector<std::pair<size_t, size_t>> cls(kSetClasses, {0, 0});
  for (int c = 0; c < kSetClasses; ++c) {
    size_t last = first;
    while (last < m && size_t(pos[last]) < q->cls_begin[c + 1]) ++last;
    cls[c] = {first, last - first};
    first = last;
  }
  for (int c = 0; c < kSetClasses; ++c) {
    const size_t at = cls[c].first, n = cls[c].second;
    if (!n) continue;
    LaunchArgs a = set_args(q, c, at, n, q->d_partials.p + at * n_seg * 16);
    a.n_seg = int(n_seg); a.seg_len = int(seg_len); a.growth = q->d_growth.p + at;
    hip_check(g_launch_seg_product[q->arith][q->scheme](a), "seg_product launch");
    q->stats.launches += 1;
  }
  if (storm::eigen_uses_lanes(m, q->eigen_layout)) {
    hip_check(storm_fast::launch_eigen(q->h_table_red[0], 2, q->d_freq.p, m, q->d_partials.p, q->d_upper.p, q->d_swaps.p, q->d_det.p,
                                       q->d_seg_nodes.p, q->stream, q->d_table_red.p, q->d_sys.p, 2), "eigen launch");
    q->stats.launches += 2;
  } else {
    for (int c = 0; c < kSetClasses; ++c) {
      const size_t at = cls[c].first, n = cls[c].second;
      if (!n) continue;
      hip_check(storm_fast::launch_eigen(q->h_table_red[q->cls_begin[c]], set_class_ell0(c) ? 1 : 0, q->d_freq.p + at, n,
                                         q->d_partials.p + at * n_seg * 16, q->d_upper.p + at * ustride, q->d_swaps.p + at * 2,
                                         q->d_det.p + at, q->d_seg_nodes.p + at * nstride, q->stream, q->d_table_red.p,
                                         q->d_sys.p + at, 1), "eigen launch");
      q->stats.launches += 2;
    }
  }
  for (int c = 0; c < kSetClasses; ++c) {
    const size_t at = cls[c].first, n = cls[c].second;
    if (!n) continue;
    LaunchArgs a = set_args(q, c, at, n, nullptr);
    a.n_seg = int(n_seg); a.seg_len = int(seg_len);
    a.nodes = q->d_seg_nodes.p + at * nstride; a.vec = q->d_vec.p + at * len;
    hip_check(g_launch_seg_fill[q->arith][q->scheme](a), "seg_fill launch");
    q->stats.launches += 1;
  }
  q->h_growth.reserve(m);
  hip_check(hipMemcpyAsync(q->h_growth.p, q->d_growth.p, m * sizeof(unsigned long long), hipMemcpyDeviceToHost, q->stream), "growth D2H");
  sync_stream(q->stream, q->sync_marker);
  for (size_t k = 0; k < m; ++k)
    if (!(q->h_growth.p[k] <= q->seg_growth_max)) return false;  // (the bit patterns of |T|, read as doubles)
  return true;
}

// Modes of one chunk, sorted by internal system position: step matrices + elimination + back-substitution, one
// launch of each per class; eigenvectors and discriminants stay in q->d_vec / q->d_det.
void set_eigen_chunk(storm_sweep* q, const double* freq, const int* pos, size_t m) {
  q->h_in.reserve(m + (m + 1) / 2);
  std::memcpy(q->h_in.p, freq, m * sizeof(double));
  int* hs = reinterpret_cast<int*>(q->h_in.p + m);
  std::memcpy(hs, pos, m * sizeof(int));
  hip_check(hipMemcpyAsync(q->d_freq.p, q->h_in.p, m * sizeof(double), hipMemcpyHostToDevice, q->stream), "freq H2D");
  hip_check(hipMemcpyAsync(q->d_sys.p, hs, m * sizeof(int), hipMemcpyHostToDevice, q->stream), "system index H2D");
  const size_t n_int = q->n_int, len = (n_int + 1) * 4, ustride = n_int * 22 + 6;
  if (set_uses_segments(q)) {
    if (set_eigen_segments(q, pos, m)) { q->stats.eigenvectors += m; return; }
    q->stats.segment_fallbacks += m;  // a transfer matrix too large to propagate through: the whole chunk sequentially
  }
  q->d_steps.reserve(m * n_int * 16 + 16, q->stream);
  q->d_upper.reserve((m + 64) * ustride, q->stream);  // lane-per-mode kernels pad the mode count to 64
  auto eigen = q->arith ? storm_fast::launch_eigen : storm::launch_eigen;
  // with the lane-per-mode kernels the whole chunk is eliminated in one launch whatever the systems' classes
  const bool lanes = storm::eigen_uses_lanes(m, q->eigen_layout);
  size_t first = 0;
  for (int c = 0; c < kSetClasses; ++c) {
    size_t last = first;
    while (last < m && size_t(pos[last]) < q->cls_begin[c + 1]) ++last;
    const size_t n = last - first;
    if (n) {
      double* steps = q->d_steps.p + first * n_int * 16;
      hip_check(g_launch_build[q->arith][q->scheme](set_args(q, c, first, n, steps)), "build_steps launch");
      q->stats.launches += 1;
      if (!lanes) {
        hip_check(eigen(q->h_table[q->cls_begin[c]], set_class_ell0(c) ? 1 : 0, q->d_freq.p + first, n, steps,
                        q->d_upper.p + first * ustride, q->d_swaps.p + first * 2, q->d_det.p + first, q->d_vec.p + first * len,
                        q->stream, q->d_table.p, q->d_sys.p + first, 1),
                  "eigen launch");
        q->stats.launches += 2;
      }
    }
    first = last;
  }
  if (lanes) {
    hip_check(eigen(q->h_table[0], 2, q->d_freq.p, m, q->d_steps.p, q->d_upper.p, q->d_swaps.p, q->d_det.p, q->d_vec.p, q->stream,
                    q->d_table.p, q->d_sys.p, 2),
              "eigen launch");
    q->stats.launches += 2;
  }
  q->stats.eigenvectors += m;
}

// Caller-order mode lists -> internal order.  src[j] = index (in the caller's concatenated arrays) of internal mode j.
void set_mode_order(const storm_sweep* q, const size_t* offsets, std::vector<size_t>& src, std::vector<int>& pos) {
  src.clear(); pos.clear();
  for (size_t k = 0; k < q->n_sys; ++k) {
    const size_t u = q->order[k];
    if (offsets[u + 1] < offsets[u]) fail(STORM_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    for (size_t i = offsets[u]; i < offsets[u + 1]; ++i) { src.push_back(i); pos.push_back(int(k)); }
  }
}

void set_validate_offsets(const storm_sweep* q, const size_t* offsets) {
  if (!offsets) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
  if (offsets[0] != 0) fail(STORM_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  for (size_t u = 0; u < q->n_sys; ++u)
    if (offsets[u + 1] < offsets[u]) fail(STORM_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
}

}  // namespace

extern "C" {

int storm_sweep_create(const storm_system_desc* descs, size_t n_sys, storm_sweep** out) {
  if (out) *out = nullptr;
  storm_sweep* q = nullptr;
  int rc = guarded([&] {
    if (!out || (n_sys && !descs)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n_sys == 0) fail(STORM_ERR_INVALID_ARGUMENT, "a sweep needs at least one system");
    if (n_sys > (size_t(1) << 24)) fail(STORM_ERR_INVALID_ARGUMENT, "too many systems");
    const storm_system_desc& d0 = descs[0];
    if (d0.scheme < 0 || d0.scheme >= kNumSchemes) fail(STORM_ERR_INVALID_ARGUMENT, "unknown difference scheme");
    if (d0.p != size_t(scheme_points(d0.scheme)))
      fail(STORM_ERR_INVALID_ARGUMENT, "p does not match the number of collocation points of the scheme");
    if (d0.n_int > (size_t(1) << 40)) fail(STORM_ERR_INVALID_ARGUMENT, "n_int too large");
    for (size_t u = 0; u < n_sys; ++u) {
      const storm_system_desc& d = descs[u];
      if (d.scheme != d0.scheme || d.n_int != d0.n_int || d.p != d0.p)
        fail(STORM_ERR_INVALID_ARGUMENT, "all systems of a sweep must share the scheme and the number of grid intervals");
      if (d.n_int && (!d.u || !d.v_gamma || !d.a_star || !d.c1 || !d.rot || !d.delta_over_x))
        fail(STORM_ERR_INVALID_ARGUMENT, "null coefficient array");
    }
    int ndev = 0;
    hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev <= 0) fail(STORM_ERR_HIP, "no HIP device available");
    q = new storm_sweep();
    hip_check(hipGetDevice(&q->device), "hipGetDevice");
    q->scheme = d0.scheme; q->n_sys = n_sys; q->n_int = d0.n_int; q->p = d0.p;
    const size_t elems = planes_elems(q->n_int, q->p);  // (no interval-major copy: see DESIGN.md section 12)
    // The kernel class of a system (ell == 0 / bitwise-uniform m*rot / general) decides its internal position; it follows
    // from the descriptor alone (same test as build_planes).
    std::vector<char> uniform(n_sys, 1);
    std::vector<double> rotm0(n_sys, 0.0);
    for (size_t u = 0; u < n_sys; ++u) {
      const storm_system_desc& d = descs[u];
      const double mm = double(d.m);
      const size_t cnt = d.n_int * d.p;
      rotm0[u] = cnt ? mm * d.rot[0] : 0.0;
      uint64_t first;
      std::memcpy(&first, &rotm0[u], 8);
      for (size_t k = 0; k < cnt && uniform[u]; ++k) {
        const double v = mm * d.rot[k];
        uint64_t b;
        std::memcpy(&b, &v, 8);
        if (b != first) uniform[u] = 0;
      }
    }
    std::vector<int> cls(n_sys);
    for (size_t u = 0; u < n_sys; ++u) cls[u] = descs[u].ell == 0 ? 0 : (uniform[u] ? 1 : 2);
    for (int c = 0; c < kSetClasses; ++c) {
      q->cls_begin[c] = q->order.size();
      for (size_t u = 0; u < n_sys; ++u) if (cls[u] == c) q->order.push_back(u);
    }
    q->cls_begin[kSetClasses] = n_sys;
    q->stream = MemoryCache::get().take_stream(q->device);
    q->tab.reserve(elems * n_sys, nullptr);
    q->d_table.reserve(n_sys, nullptr);
    q->h_table.resize(n_sys);
    q->ell.resize(n_sys); q->m.resize(n_sys);
    // Planes are built (a few host threads) straight into a pinned block and uploaded from there, a slab of systems at
    // a time so that a sweep of thousands of systems does not pin gigabytes of host memory.
    const size_t slab = std::max<size_t>(1, std::min<size_t>(n_sys, (size_t(256) << 20) / (elems * sizeof(double))));
    q->h_in.reserve(elems * slab);
    for (size_t k0 = 0; k0 < n_sys; k0 += slab) {
      const size_t k1 = std::min(n_sys, k0 + slab);
      if (k0) sync_stream(q->stream, q->sync_marker);  // the previous slab has left the pinned block
      {
        const size_t hw = std::max<size_t>(1, std::thread::hardware_concurrency());
        const size_t n_thr = std::min<size_t>({k1 - k0, hw, size_t(16)});
        std::atomic<size_t> next{k0};
        auto work = [&] {
          for (size_t k; (k = next.fetch_add(1)) < k1;) {
            bool uni = true;
            double r0 = 0.0;
            build_planes(descs[q->order[k]], q->h_in.p + (k - k0) * elems, uni, r0);
          }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_thr; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
      }
      hip_check(hipMemcpyAsync(q->tab.p + k0 * elems, q->h_in.p, (k1 - k0) * elems * sizeof(double), hipMemcpyHostToDevice,
                               q->stream), "planes H2D");
    }
    for (size_t k = 0; k < n_sys; ++k) {
      const size_t u = q->order[k];
      const storm_system_desc& d = descs[u];
      SystemDev& sd = q->h_table[k];
      const double ell = double(d.ell);
      sd.tab = q->tab.p + k * elems;
      sd.stride = padded_stride(d.n_int);
      sd.plane = sd.stride * d.p;
      sd.n_int = d.n_int;
      sd.ell = ell;
      sd.lambda = ell * (ell + 1.0);
      sd.inner_c1 = d.inner_c1;
      sd.inner_rotm = double(d.m) * d.inner_rot;
      sd.outer_u = d.outer_u;
      sd.rotm_uniform = rotm0[u];
      q->ell[u] = d.ell; q->m[u] = d.m;
    }
    // the table is copied from pageable memory: hipMemcpyAsync stages it before returning
    hip_check(hipMemcpyAsync(q->d_table.p, q->h_table.data(), n_sys * sizeof(SystemDev), hipMemcpyHostToDevice, q->stream),
              "system table H2D");
    q->d_flags.reserve(1);
    hip_check(hipMemsetAsync(q->d_flags.p, 0, sizeof(int), q->stream), "flags memset");
    sync_stream(q->stream, q->sync_marker);  // h_in is reused by the first evaluation
    *out = q;
  });
  if (rc != STORM_OK && q) {
    storm_sweep_destroy(q);
    if (out) *out = nullptr;
  }
  return rc;
}

void storm_sweep_destroy(storm_sweep* q) {
  if (!q) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(q->device);
  if (q->stream) { (void)hipStreamSynchronize(q->stream); MemoryCache::get().give_stream(q->device, q->stream); }
  if (q->sync_marker) (void)hipEventDestroy(q->sync_marker);
  q->tab.release(); q->d_table.release(); q->d_freq.release(); q->d_det.release(); q->d_steps.release(); q->d_upper.release();
  q->d_vec.release(); q->d_sys.release(); q->d_swaps.release(); q->d_flags.release(); q->d_terms.release(); q->d_scale.release();
  q->d_fields.release(); q->d_nodes.release(); q->d_ids.release(); q->d_post_table.release(); q->d_cp.release(); q->d_cp_idx.release(); q->d_cp_ell.release(); q->d_cp_flags.release(); q->d_cp_order.release(); q->d_cp_partial.release();
  q->d_cp_table.release(); q->h_cp.release(); q->h_in.release(); q->h_out.release(); q->h_flags.release();
  q->d_partials.release(); q->d_seg_nodes.release(); q->d_growth.release(); q->d_table_red.release(); q->h_growth.release();
  (void)hipSetDevice(prev);
  delete q;
}

size_t storm_sweep_size(const storm_sweep* q) { return q ? q->n_sys : 0; }
size_t storm_sweep_n_int(const storm_sweep* q) { return q ? q->n_int : 0; }

int storm_sweep_set_option(storm_sweep* q, const char* key, int64_t value) {
  return guarded([&] {
    if (!q || !key) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    const std::string k(key);
    if (k == "arithmetic") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "arithmetic must be 0 (parity) or 1 (fast)");
      q->arith = int(value);
    } else if (k == "hex_lanes") {
      if (value != 0 && value != 4 && value != 16 && value != 32) fail(STORM_ERR_INVALID_ARGUMENT, "hex_lanes must be 0 (auto), 4, 16 or 32");
      q->hex_lanes = int(value);
    } else if (k == "coupling_tile") {
      if (value != 0 && value != 8 && value != 16) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_tile must be 0 (auto), 8 or 16");
      q->cp_tile = int(value);
    } else if (k == "coupling_matrix") {
      if (value != 0 && value != 1) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_matrix must be 0 or 1");
      q->cp_matrix = int(value);
    } else if (k == "coupling_matrix_tile") {
      if (value != 0 && value != 16 && value != 32) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_matrix_tile must be 0 (auto), 16 or 32");
      q->cp_matrix_tile = int(value);
    } else if (k == "coupling_split") {
      if (value < 0 || value > 64) fail(STORM_ERR_INVALID_ARGUMENT, "coupling_split must be 0 (automatic) or 1..64 pieces");
      q->cp_split = int(value);
    } else if (k == "segments") {
      if (value != 0 && value != -1) fail(STORM_ERR_INVALID_ARGUMENT, "segments must be 0 (automatic, FMA build only) or -1 (never)");
      q->segments = int(value);
    } else if (k == "seg_growth_log2") {
      if (value < 0 || value > 1000) fail(STORM_ERR_INVALID_ARGUMENT, "seg_growth_log2 must be 0 .. 1000");
      q->seg_growth_max = std::ldexp(1.0, int(value));
    } else if (k == "hex_rounds") {
      if (value < 0 || value > 1024) fail(STORM_ERR_INVALID_ARGUMENT, "hex_rounds must be 0 (one launch) .. 1024");
      q->hex_rounds = int(value);
    } else if (k == "eigen_layout") {
      if (value < 0 || value > 2) fail(STORM_ERR_INVALID_ARGUMENT, "eigen_layout must be 0 (auto), 1 (wave per mode) or 2 (lane per mode)");
      q->eigen_layout = int(value);
    } else if (k == "steps_bytes_max") {
      if (value < 1) fail(STORM_ERR_INVALID_ARGUMENT, "steps_bytes_max must be positive");
      q->steps_bytes_max = size_t(value);
    } else {
      fail(STORM_ERR_INVALID_ARGUMENT, "unknown option: " + k);
    }
  });
}

int storm_sweep_stats(storm_sweep* q, storm_stats* out, int reset) {
  return guarded([&] {
    if (!q || !out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    *out = q->stats;
    if (reset) q->stats = storm_stats{};
  });
}

int storm_sweep_det(storm_sweep* q, const size_t* offsets, const double* freq, double* det_out) {
  return guarded([&] {
    if (!q) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    set_validate_offsets(q, offsets);
    if (offsets[q->n_sys] && (!freq || !det_out)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    DeviceGuard g(q->device);
    const size_t S = q->n_sys;
    std::vector<const double*> x(S);
    std::vector<double*> f(S);
    std::vector<size_t> cnt(S);
    for (size_t k = 0; k < S; ++k) {
      const size_t u = q->order[k];
      x[k] = freq + offsets[u]; f[k] = det_out + offsets[u]; cnt[k] = offsets[u + 1] - offsets[u];
    }
    set_det(q, x, cnt, f);
  });
}

int storm_sweep_scan_and_optimize(storm_sweep* q, const size_t* freq_offsets, const double* freq, uint64_t ulp,
                                       storm_bracket* out, size_t cap, size_t* out_offsets, int* had_invalid, double* det_out,
                                       uint64_t* index_out) {
  return guarded([&] {
    if (!q || !out_offsets || (cap && !out)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    set_validate_offsets(q, freq_offsets);
    const size_t S = q->n_sys, total = freq_offsets[S];
    for (size_t u = 0; u <= S; ++u) out_offsets[u] = 0;
    if (had_invalid) for (size_t u = 0; u < S; ++u) had_invalid[u] = 0;
    if (total && !freq) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(q->mu);
    DeviceGuard g(q->device);
    std::vector<double> local;
    double* dets = det_out;
    if (!dets) { local.resize(total); dets = local.data(); }
    std::vector<const double*> x(S);
    std::vector<double*> f(S);
    std::vector<size_t> cnt(S);
    for (size_t k = 0; k < S; ++k) {
      const size_t u = q->order[k];
      x[k] = freq + freq_offsets[u]; f[k] = dets + freq_offsets[u]; cnt[k] = freq_offsets[u + 1] - freq_offsets[u];
    }
    set_det(q, x, cnt, f);
    // FilterSignSwap + finite filter per system (src/bracket.rs:236-250, src/dynamic_interface.rs:128-138), gathered in
    // internal order (what the merged refinement rounds need); idx = grid index inside the system's own scan
    std::vector<SetBracket> br;
    std::vector<uint64_t> idx;
    std::vector<size_t> n_br(S, 0);  // by caller index
    for (size_t k = 0; k < S; ++k) {
      const size_t u = q->order[k];
      const double* xs = x[k];
      const double* fs = f[k];
      bool invalid = false;
      for (size_t i = 0; i + 1 < cnt[k]; ++i) {
        const double a = fs[i], b = fs[i + 1];
        if (sign_of(a) != sign_of(b)) {
          if (std::isfinite(a) && std::isfinite(b)) {
            if (!std::isfinite(xs[i]) || !std::isfinite(xs[i + 1])) fail(STORM_ERR_BRACKET, "non-finite scan frequency");
            SetBracket st;
            st.lower = Pt{xs[i], a};
            st.upper = Pt{xs[i + 1], b};
            st.pos = k;
            br.push_back(st);
            idx.push_back(i);
            n_br[u] += 1;
          } else {
            invalid = true;
          }
        }
      }
      if (had_invalid) had_invalid[u] = invalid ? 1 : 0;
    }
    for (size_t u = 0; u < S; ++u) out_offsets[u + 1] = out_offsets[u] + n_br[u];
    if (br.size() > cap) fail(STORM_ERR_CAPACITY, "bracket output buffer too small");
    set_refine(q, br, ulp);
    std::vector<size_t> at(out_offsets, out_offsets + S);
    for (size_t j = 0; j < br.size(); ++j) {
      const size_t u = q->order[br[j].pos], dst = at[u]++;
      out[dst] = storm_bracket{br[j].lower.x, br[j].lower.f, br[j].upper.x, br[j].upper.f, br[j].root, br[j].evals};
      if (index_out) index_out[dst] = idx[j];
    }
  });
}

static int sweep_modes_impl(storm_sweep* q, storm_postproc* const* posts, const size_t* offsets, const double* freq,
                            double* det_out, double* vec_out, double* fields_out, storm_mode_id* ids_out, double* nodes_out,
                            size_t node_cap, bool resident) {
  return guarded([&] {
    if (!q) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    set_validate_offsets(q, offsets);
    const size_t S = q->n_sys, n = offsets[S];
    if (n == 0) return;
    if (!freq) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    const bool post = posts != nullptr;
    if (post && !ids_out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (!post && (fields_out || ids_out || nodes_out))
      fail(STORM_ERR_INVALID_ARGUMENT, "fields, ids and nodes need the post-processing handles");
    if (!post && !vec_out) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    const size_t np = q->n_int + 1, len = np * 4;
    if (post)
      for (size_t u = 0; u < S; ++u) {
        if (offsets[u + 1] == offsets[u]) continue;
        if (!posts[u]) fail(STORM_ERR_INVALID_ARGUMENT, "null post-processing handle");
        if (posts[u]->model.n_pts != np)
          fail(STORM_ERR_INVALID_ARGUMENT, "the solver grid is not the post-processing model's grid (n_int + 1 != n_pts)");
        if (posts[u]->device != q->device) fail(STORM_ERR_INVALID_ARGUMENT, "sweep and post-processing handle live on different devices");
      }
    std::lock_guard<std::mutex> lock(q->mu);
    DeviceGuard g(q->device);
    std::vector<size_t> src;
    std::vector<int> pos;
    set_mode_order(q, offsets, src, pos);
    std::vector<double> fr(n);
    for (size_t j = 0; j < n; ++j) fr[j] = freq[src[j]];
    const bool want_fields = fields_out != nullptr || resident;
    const size_t ncap = nodes_out ? node_cap : 0;
    q->resident_at.clear();  // d_fields is about to be reused
    if (resident && !post) fail(STORM_ERR_INVALID_ARGUMENT, "resident fields need the post-processing handles");
    size_t chunk = set_eigen_reserve(q, n);
    if (post) {
      const size_t per_mode = (np * (1 + (want_fields ? kPostFields : 0)) + ncap + 8) * sizeof(double);
      chunk = std::min(chunk, std::max<size_t>(1, q->steps_bytes_max / per_mode));
      q->d_terms.reserve(chunk * np, q->stream);
      q->d_scale.reserve(chunk, q->stream);
      q->d_ids.reserve(chunk * 4, q->stream);
      if (want_fields) q->d_fields.reserve(chunk * kPostFields * np, q->stream);
      if (ncap) q->d_nodes.reserve(chunk * ncap, q->stream);
      if (resident && chunk < n)
        fail(STORM_ERR_CAPACITY, "the fields of all modes do not fit the staging cap in one pass: raise steps_bytes_max or pass fewer modes");
    }
    std::vector<PostSystem> table;  // outlives the copy below (the stream is synchronised before this call returns)
    if (post) {
      // one (model, ell, m) entry per system; the kernels pick the entry of each mode (its system's internal position)
      table.resize(S);
      for (size_t p = 0; p < S; ++p) {
        const size_t u = q->order[p];
        table[p] = PostSystem{};
        if (offsets[u + 1] == offsets[u]) continue;
        storm_postproc* pp = posts[u];
        {
          std::lock_guard<std::mutex> lock_p(pp->mu);
          ensure_model(pp, q->stream);
        }
        table[p].model = pp->model;
        table[p].par = post_params(q->ell[u], q->m[u]);
      }
      q->d_post_table.reserve(S, q->stream);
      hip_check(hipMemcpyAsync(q->d_post_table.p, table.data(), S * sizeof(PostSystem), hipMemcpyHostToDevice, q->stream),
                "post-processing table H2D");
    }
    std::vector<long long> h_ids;
    for (size_t off = 0; off < n; off += chunk) {
      const size_t k = std::min(chunk, n - off);
      set_eigen_chunk(q, fr.data() + off, pos.data() + off, k);
      if (post) {
        PostArgs pa;
        pa.model.tab = nullptr; pa.model.n_pts = np; pa.par = PostParams{};
        pa.systems = q->d_post_table.p; pa.mode_sys = q->d_sys.p;
        pa.freq = q->d_freq.p; pa.vec = q->d_vec.p; pa.n = k;
        pa.terms = q->d_terms.p; pa.scale = q->d_scale.p;
        pa.fields = want_fields ? q->d_fields.p : nullptr;
        pa.ids = q->d_ids.p;
        pa.nodes = ncap ? q->d_nodes.p : nullptr;
        pa.node_cap = ncap; pa.stream = q->stream;
        hip_check(launch_postprocess(pa), "postprocess launch");
        q->stats.launches += want_fields ? 3 : 2;
      }
      // results to the host: staged in pinned memory in internal order, scattered to the caller's order
      const size_t nv = vec_out ? k * len : 0, nf = fields_out ? k * kPostFields * np : 0, nn = ncap ? k * ncap : 0;
      q->h_out.reserve(nv + nf + nn + k + 1);
      double* hv = q->h_out.p; double* hf = hv + nv; double* hn = hf + nf; double* hd = hn + nn;
      if (nv) hip_check(hipMemcpyAsync(hv, q->d_vec.p, nv * sizeof(double), hipMemcpyDeviceToHost, q->stream), "vec D2H");
      if (nf) hip_check(hipMemcpyAsync(hf, q->d_fields.p, nf * sizeof(double), hipMemcpyDeviceToHost, q->stream), "fields D2H");
      if (nn) hip_check(hipMemcpyAsync(hn, q->d_nodes.p, nn * sizeof(double), hipMemcpyDeviceToHost, q->stream), "nodes D2H");
      hip_check(hipMemcpyAsync(hd, q->d_det.p, k * sizeof(double), hipMemcpyDeviceToHost, q->stream), "det D2H");
      if (post) {
        h_ids.resize(k * 4);
        hip_check(hipMemcpyAsync(h_ids.data(), q->d_ids.p, k * 4 * sizeof(long long), hipMemcpyDeviceToHost, q->stream), "ids D2H");
      }
      set_check_flags(q);  // synchronises
      for (size_t j = 0; j < k; ++j) {
        const size_t dst = src[off + j];
        if (vec_out) std::memcpy(vec_out + dst * len, hv + j * len, len * sizeof(double));
        if (fields_out) std::memcpy(fields_out + dst * kPostFields * np, hf + j * kPostFields * np, kPostFields * np * sizeof(double));
        if (ncap) std::memcpy(nodes_out + dst * ncap, hn + j * ncap, ncap * sizeof(double));
        if (det_out) det_out[dst] = hd[j];
        if (post)
          ids_out[dst] = storm_mode_id{uint64_t(h_ids[j * 4]), uint64_t(h_ids[j * 4 + 1]), int64_t(h_ids[j * 4 + 2]),
                                       uint64_t(h_ids[j * 4 + 3])};
      }
    }
    if (resident) {
      q->resident_at.assign(n, 0);
      for (size_t j = 0; j < n; ++j) q->resident_at[src[j]] = (long long)j;
    }
    // The staging of a large pass (2 MB per mode: step matrices, factors, eigenvectors) goes back to the cache now (the
    // stream is idle: every chunk ends with a synchronisation), not when the sweep is destroyed: with several sweeps
    // in flight (tools/storm_batch runs four groups at a time) they then share one or two sets instead of holding one
    // each through their coupling phase -- and fresh device memory costs ~30 ms per GB when the driver has to clear it.
    for (DevBuf<double>* b : {&q->d_steps, &q->d_upper, &q->d_vec, &q->d_terms})
      if (b->granted >= MemoryCache::kLarge) b->release();
  });
}

int storm_sweep_modes(storm_sweep* q, storm_postproc* const* posts, const size_t* offsets, const double* freq, double* det_out,
                      double* vec_out, double* fields_out, storm_mode_id* ids_out, double* nodes_out, size_t node_cap) {
  return sweep_modes_impl(q, posts, offsets, freq, det_out, vec_out, fields_out, ids_out, nodes_out, node_cap, false);
}

int storm_sweep_modes_resident(storm_sweep* q, storm_postproc* const* posts, const size_t* offsets, const double* freq,
                               storm_mode_id* ids_out, double* nodes_out, size_t node_cap) {
  return sweep_modes_impl(q, posts, offsets, freq, nullptr, nullptr, nullptr, ids_out, nodes_out, node_cap, true);
}

int storm_sweep_coupling_matrices(storm_sweep* q, storm_postproc* p, size_t n, const size_t* mode_index, const uint64_t* ell,
                                  const double* freq, int64_t m, const storm_metric_desc* metric, double* out) {
  return guarded([&] {
    if (!q || !p || (n && !mode_index)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return;
    if (q->device != p->device) fail(STORM_ERR_INVALID_ARGUMENT, "sweep and post-processing handle live on different devices");
    if (p->model.n_pts != q->n_int + 1)
      fail(STORM_ERR_INVALID_ARGUMENT, "the solver grid is not the post-processing model's grid (n_int + 1 != n_pts)");
    std::vector<long long> rows(n);
    const double* resident_fields = nullptr;
    {
      // Only the row lookup is under the sweep's lock: integrals of different blocks may run concurrently from several
      // host threads (each on its own post-processing handle and stream).  The caller must not start another
      // storm_sweep_modes* call on this sweep while they run (it would overwrite the resident fields).
      std::lock_guard<std::mutex> lock(q->mu);
      if (q->resident_at.empty()) fail(STORM_ERR_INVALID_ARGUMENT, "no resident fields: call storm_sweep_modes_resident first");
      for (size_t k = 0; k < n; ++k) {
        if (mode_index[k] >= q->resident_at.size()) fail(STORM_ERR_INVALID_ARGUMENT, "mode index out of range");
        rows[k] = q->resident_at[mode_index[k]];
      }
      resident_fields = q->d_fields.p;
    }
    const size_t row_elems = size_t(kPostFields) * (q->n_int + 1);
    coupling_run(p, n, ell, freq, m, metric, out, [&](double* d_fields, hipStream_t st) {
      // the resident fields were completed by a call that synchronised the sweep's stream: safe to read from `st`
      p->d_gather.reserve(n, st);
      hip_check(hipMemcpyAsync(p->d_gather.p, rows.data(), n * sizeof(long long), hipMemcpyHostToDevice, st), "gather index H2D");
      hip_check(launch_gather_rows(resident_fields, p->d_gather.p, n, row_elems, d_fields, st), "gather launch");
    });
  });
}

int storm_sweep_coupling_matrices_batch(storm_sweep* q, size_t n_problems, storm_postproc* const* posts, const size_t* offsets,
                                        const size_t* mode_index, const uint64_t* ell, const double* freq, const int64_t* m,
                                        const storm_metric_desc* metrics, double* const* out) {
  return guarded([&] {
    if (!q || (n_problems && (!posts || !offsets || !m || !metrics || !out))) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    if (n_problems == 0) return;
    if (n_problems > 65535) fail(STORM_ERR_INVALID_ARGUMENT, "too many problems in one call");
    if (offsets[0] != 0) fail(STORM_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
    const size_t total = offsets[n_problems];
    if (total && (!mode_index || !ell || !freq)) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
    const size_t np = q->n_int + 1;
    size_t max_modes = 0, out_total = 0;
    for (size_t j = 0; j < n_problems; ++j) {
      if (offsets[j + 1] < offsets[j]) fail(STORM_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
      const size_t n = offsets[j + 1] - offsets[j];
      if (!n) continue;
      if (!posts[j] || !out[j]) fail(STORM_ERR_INVALID_ARGUMENT, "null argument");
      if (posts[j]->device != q->device) fail(STORM_ERR_INVALID_ARGUMENT, "sweep and post-processing handle live on different devices");
      if (posts[j]->model.n_pts != np)
        fail(STORM_ERR_INVALID_ARGUMENT, "the solver grid is not the post-processing model's grid (n_int + 1 != n_pts)");
      check_metric(&metrics[j]);
      if (n > 8 * size_t(65535)) fail(STORM_ERR_INVALID_ARGUMENT, "too many modes");
      max_modes = std::max(max_modes, n);
      out_total += size_t(kCouplingMatrices) * n * n;
    }
    if (max_modes == 0) return;
    std::lock_guard<std::mutex> lock(q->mu);
    DeviceGuard g(q->device);
    if (q->resident_at.empty()) fail(STORM_ERR_INVALID_ARGUMENT, "no resident fields: call storm_sweep_modes_resident first");
    // host image: [weights of every problem | consts of every problem | frequencies], rows and ell indices separately
    const size_t w_elems = (np + 16) * kCouplingWeights;
    std::vector<std::vector<double>> consts(n_problems);
    std::vector<int> ell_index(total);
    std::vector<size_t> n_ell(n_problems, 0), c_off(n_problems, 0);
    size_t c_total = 0;
    for (size_t j = 0; j < n_problems; ++j) {
      const size_t n = offsets[j + 1] - offsets[j];
      if (!n) continue;
      n_ell[j] = coupling_consts(n, ell + offsets[j], m[j], ell_index.data() + offsets[j], consts[j]);
      c_off[j] = c_total;
      c_total += consts[j].size();
    }
    const size_t in_elems = n_problems * w_elems + c_total + total;
    q->h_cp.reserve(std::max(in_elems, out_total) + 1);
    double* hw = q->h_cp.p;
    double* hc = hw + n_problems * w_elems;
    double* hf = hc + c_total;
    parallel_for(n_problems, 4, [&](size_t j) {
      if (offsets[j + 1] == offsets[j]) return;
      coupling_weights(posts[j], m[j], &metrics[j], hw + j * w_elems);
      std::memcpy(hc + c_off[j], consts[j].data(), consts[j].size() * sizeof(double));
    });
    if (total) std::memcpy(hf, freq, total * sizeof(double));
    std::vector<long long> rows(total);
    for (size_t k = 0; k < total; ++k) {
      if (mode_index[k] >= q->resident_at.size()) fail(STORM_ERR_INVALID_ARGUMENT, "mode index out of range");
      rows[k] = q->resident_at[mode_index[k]];
    }
    hipStream_t st = q->stream;
    q->d_cp.reserve(in_elems + out_total, st);
    q->d_cp_idx.reserve(total, st);
    q->d_cp_ell.reserve(total, st);
    q->d_cp_flags.reserve(1 + total, st);  // [0]: non-finite flag, [1..]: per-mode "tame" flags (kernel scratch)
    q->d_cp_table.reserve(n_problems, st);
    double* dw = q->d_cp.p;
    double* dc = dw + n_problems * w_elems;
    double* df = dc + c_total;
    double* dout = q->d_cp.p + in_elems;
    std::vector<CouplingArgs> table(n_problems);
    size_t out_at = 0;
    bool with_tame = false;
    // FMA build: every problem's ordered mode list (matrix-pipe walk), back to back in one device array
    std::vector<int> orders, one;
    std::vector<size_t> ord_off(n_problems + 1, 0);
    size_t max_pad = 0;
    if (q->arith == 1 && q->cp_matrix) {
      for (size_t j = 0; j < n_problems; ++j) {
        const size_t n = offsets[j + 1] - offsets[j];
        coupling_order(n, ell_index.data() + offsets[j], n_ell[j], one);
        orders.insert(orders.end(), one.begin(), one.end());
        ord_off[j + 1] = orders.size();
        max_pad = std::max(max_pad, one.size());
      }
      q->d_cp_order.reserve(orders.size() + 1, st);
      hip_check(hipMemcpyAsync(q->d_cp_order.p, orders.data(), orders.size() * sizeof(int), hipMemcpyHostToDevice, st), "order H2D");
    }
    // the matrix-pipe walk's tile and the pieces the grid is cut into per tile (a launch ends waiting for its last
    // workgroups: many short ones instead of a few long ones); the pieces' sums go to d_cp_partial, [piece][all outputs]
    storm::CouplingPlan plan{q->cp_matrix_tile, 1};
    if (max_pad) {
      plan = storm::plan_coupling_batch(max_pad, n_problems, np, q->cp_matrix_tile, q->cp_split);
      if (plan.n_split > 1) q->d_cp_partial.reserve(size_t(plan.n_split) * out_total, st);
    }
    for (size_t j = 0; j < n_problems; ++j) {
      const size_t n = offsets[j + 1] - offsets[j];
      CouplingArgs& a = table[j];
      if (max_pad) {
        a.order = q->d_cp_order.p + ord_off[j]; a.n_pad = ord_off[j + 1] - ord_off[j];
        a.n_split = plan.n_split;
        if (plan.n_split > 1) { a.partial = q->d_cp_partial.p + out_at; a.partial_stride = out_total; }
      }
      a.n_pts = np; a.n_modes = n; a.fields = q->d_fields.p; a.rows = q->d_cp_idx.p + offsets[j];
      a.freq = df + offsets[j]; a.ell_index = q->d_cp_ell.p + offsets[j];
      a.consts = dc + c_off[j]; a.n_ell = int(n_ell[j]); a.weights = dw + j * w_elems;
      a.out = dout + out_at; a.flags = q->d_cp_flags.p; a.stream = st;
      if (n && coupling_has_sparse_pair(consts[j])) {
        a.tame = q->d_cp_flags.p + 1 + offsets[j];
        a.weights_tame = coupling_weights_tame(hw + j * w_elems, np * kCouplingWeights) ? 1 : 0;
        with_tame = true;
      }
      out_at += size_t(kCouplingMatrices) * n * n;
    }
    hip_check(hipMemcpyAsync(q->d_cp.p, q->h_cp.p, in_elems * sizeof(double), hipMemcpyHostToDevice, st), "coupling inputs H2D");
    hip_check(hipMemcpyAsync(q->d_cp_idx.p, rows.data(), total * sizeof(long long), hipMemcpyHostToDevice, st), "rows H2D");
    hip_check(hipMemcpyAsync(q->d_cp_ell.p, ell_index.data(), total * sizeof(int), hipMemcpyHostToDevice, st), "ell H2D");
    hip_check(hipMemcpyAsync(q->d_cp_table.p, table.data(), n_problems * sizeof(CouplingArgs), hipMemcpyHostToDevice, st), "table H2D");
    hip_check(hipMemsetAsync(q->d_cp_flags.p, 0, sizeof(int), st), "flags memset");
    hip_check((q->arith ? storm_fast::launch_coupling_batch : storm::launch_coupling_batch)(q->d_cp_table.p, n_problems,
                                                                                             max_modes, np, st, with_tame,
                                                                                             q->cp_tile ? q->cp_tile : coupling_tile_auto(max_modes), max_pad, plan.tile,
                                                                                             plan.n_split, q->d_cp_partial.p, dout, out_total, q->d_cp_flags.p),
              "coupling launch");
    q->stats.launches += 1;
    // (the pageable sources above are staged by the runtime before the calls return; the pinned block is reused for the
    // results only after the stream has been synchronised)
    sync_stream(st, q->sync_marker);
    q->h_flags.reserve(1);
    int* flags_ptr = reinterpret_cast<int*>(q->h_flags.p);
    hip_check(hipMemcpyAsync(q->h_cp.p, dout, out_total * sizeof(double), hipMemcpyDeviceToHost, st), "out D2H");
    hip_check(hipMemcpyAsync(flags_ptr, q->d_cp_flags.p, sizeof(int), hipMemcpyDeviceToHost, st), "flags D2H");
    sync_stream(st, q->sync_marker);
    std::vector<size_t> out_off(n_problems + 1, 0);
    for (size_t j = 0; j < n_problems; ++j) {
      const size_t n = offsets[j + 1] - offsets[j];
      out_off[j + 1] = out_off[j] + size_t(kCouplingMatrices) * n * n;
    }
    // (the callers' blocks are usually fresh pages: one thread touches them at ~3 GB/s, four at ~10)
    parallel_for(n_problems, 4, [&](size_t j) {
      const size_t cnt = out_off[j + 1] - out_off[j];
      if (cnt) std::memcpy(out[j], q->h_cp.p + out_off[j], cnt * sizeof(double));
    });
    if (*flags_ptr & 1)
      fail(STORM_ERR_NON_FINITE, "Not finite number in a coupling integral (the reference asserts, src/perturbed.rs:165-260)");
  });
}

int storm_sweep_eigenvectors(storm_sweep* q, const size_t* offsets, const double* freq, double* det_out,
                                       double* vec_out) {
  return storm_sweep_modes(q, nullptr, offsets, freq, det_out, vec_out, nullptr, nullptr, nullptr, 0);
}

}  // extern "C"
