// mfma64_microbench.hip -- does v_mfma_f64_4x4x4 (four independent 4x4x4 FP64 products per instruction) run beside the FP64
// vector pipe on gfx950, or through it?  Measures: VALU v_fma_f64 alone, MFMA alone, both interleaved in one wave, the
// latency of a dependent MFMA chain (D fed back as B: the ordered product P <- S_t P), and checks the operand layout.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma64_microbench.hip -o tools/mfma64_microbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// NV = vector FMAs per inner step (independent), NM = MFMAs per inner step (independent accumulators)
template <int NV, int NM>
__global__ void mix_kernel(double* out, int iters, double a, double b) {
  double x[NV > 0 ? NV : 1];
  double acc[NM > 0 ? NM : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) x[i] = out[threadIdx.x] + i;
#pragma unroll
  for (int i = 0; i < NM; ++i) acc[i] = out[threadIdx.x] * 0.5 + i;
  const double ma = a * 1e-3, mb = b * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < (NV > NM ? NV : NM); ++i) {
        if (i < NM) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, acc[i], 0, 0, 0);
        if (i < NV) x[i] = __builtin_fma(x[i], a, b);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < NM; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// dependent chain: P <- S * P with S constant per lane (A operand), P in the B/D layout
__global__ void chain_kernel(double* out, int iters, double a) {
  double p = out[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p = __builtin_amdgcn_mfma_f64_4x4x4f64(a, p, 0.0, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = p;
}

// layout probe: one wave, lane l supplies A-operand a[l], B-operand b[l]; returns d[l]
__global__ void layout_kernel(const double* a, const double* b, double* d) {
  d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}

template <typename F>
static float time_ms(F&& launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.gcnArchName, cus);
  const int threads = 256;
  double* out;
  CHECK(hipMalloc(&out, size_t(cus) * 8 * threads * sizeof(double)));
  CHECK(hipMemset(out, 0, size_t(cus) * 8 * threads * sizeof(double)));

  // ---- layout: which (i,k) of A, (k,j) of B and (i,j) of D does lane l hold?
  {
    std::vector<double> A(64), B(64), D(64);
    // hypothesis: block = l / 16; A: i = l % 4, k = (l / 4) % 4; B: k = (l / 4) % 4, j = l % 4; D: i = (l / 4) % 4, j = l % 4
    double Am[4][4][4], Bm[4][4][4];
    for (int blk = 0; blk < 4; ++blk)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          Am[blk][i][j] = 1.0 + blk * 100 + i * 10 + j + 0.25 * ((i * 7 + j * 3 + blk) % 5);
          Bm[blk][i][j] = 2.0 - blk * 3 + i * 0.5 - j * 1.5 + 0.125 * ((i * 5 + j * 11 + blk) % 7);
        }
    for (int l = 0; l < 64; ++l) {
      const int blk = l / 16, lo = l % 4, hi = (l / 4) % 4;
      A[l] = Am[blk][lo][hi];
      B[l] = Bm[blk][hi][lo];
    }
    double *dA, *dB, *dD;
    CHECK(hipMalloc(&dA, 512)); CHECK(hipMalloc(&dB, 512)); CHECK(hipMalloc(&dD, 512));
    CHECK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dD);
    CHECK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    int ok_ij = 0, ok_ji = 0;
    for (int l = 0; l < 64; ++l) {
      const int blk = l / 16, lo = l % 4, hi = (l / 4) % 4;
      double ref_hi_lo = 0, ref_lo_hi = 0;
      for (int k = 0; k < 4; ++k) {
        ref_hi_lo = std::fma(Am[blk][hi][k], Bm[blk][k][lo], ref_hi_lo);
        ref_lo_hi = std::fma(Am[blk][lo][k], Bm[blk][k][hi], ref_lo_hi);
      }
      ok_ij += D[l] == ref_hi_lo;
      ok_ji += D[l] == ref_lo_hi;
    }
    printf("layout: A lane l = A[l%%4][(l/4)%%4], B lane l = B[(l/4)%%4][l%%4], block l/16: D lane l = D[(l/4)%%4][l%%4] on %d/64 lanes, "
           "= D[l%%4][(l/4)%%4] on %d/64 lanes (k ascending, fused)\n", ok_ij, ok_ji);
    if (ok_ij != 64 && ok_ji != 64) {
      printf("  lane: got\n");
      for (int l = 0; l < 16; ++l) printf("  %2d: %.6f\n", l, D[l]);
    }
  }

  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {   // waves per SIMD
    const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
    auto run = [&](auto kern, int nv, int nm, const char* name) {
      const float ms = time_ms([&] { kern<<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); });
      const double waves = double(blocks) * 4;
      const double vflop = waves * 64 * 2.0 * nv * 4 * iters, mflop = waves * 512.0 * nm * 4 * iters;
      printf("  %d waves/SIMD %-18s %8.3f ms  valu %6.2f TFLOP/s  mfma %6.2f TFLOP/s  sum %6.2f\n", wps, name, ms,
             vflop / ms / 1e9, mflop / ms / 1e9, (vflop + mflop) / ms / 1e9);
    };
    run(mix_kernel<8, 0>, 8, 0, "valu8");
    run(mix_kernel<0, 4>, 0, 4, "mfma4");
    run(mix_kernel<8, 2>, 8, 2, "valu8+mfma2");
    run(mix_kernel<8, 1>, 8, 1, "valu8+mfma1");
    run(mix_kernel<8, 4>, 8, 4, "valu8+mfma4");
    run(mix_kernel<4, 4>, 4, 4, "valu4+mfma4");
  }
  {
    const float ms = time_ms([&] { chain_kernel<<<1, 64>>>(out, 20000, 0.25); });
    printf("dependent MFMA chain (D -> B): %.1f ns per step\n", ms * 1e6 / (20000.0 * 8));
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
