// fp64_microbench.hip -- measures what the FP64 vector pipe of one MI355X actually delivers:
// dependent-chain latency, single-wave issue rate, and chip-wide v_fma_f64 / v_mul+v_add rates.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_microbench.hip -o tools/fp64_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP, int MODE>  // MODE 0: fma, 1: mul then add (un-fused), 2: add only, 3: v_rcp_f64
__global__ void valu_kernel(double* out, int iters, double a, double b) {
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = out[threadIdx.x] + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) x[i] = __builtin_fma(x[i], a, b);
        else if (MODE == 1) { double t = x[i] * a; asm volatile("" : "+v"(t)); x[i] = t + b; }
        else if (MODE == 3) x[i] = __builtin_amdgcn_rcp(x[i]);
        else x[i] = x[i] + b;
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
__device__ __forceinline__ double quad_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, K * 0x55, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, K * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ void dpp_chain_kernel(double* out, int iters, double a) {
  double x = out[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x = quad_bcast<1>(x) + a;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void shfl_chain_kernel(double* out, int iters, double a) {
  double x = out[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x = __shfl(x, 3, 8) + a;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void div_chain_kernel(double* out, int iters, double a) {
  double x = out[threadIdx.x] + 2.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x = a / x;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <class F>
float time_ms(F&& launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();  // warm
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
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* d;
  CHECK(hipMalloc(&d, sizeof(double) * 256 * 4096 * 8));
  CHECK(hipMemset(d, 0, sizeof(double) * 256 * 4096 * 8));
  const int iters = 20000;
  // --- single wave, dependent chain latency (ILP 1) and issue rate (ILP 8)
  {
    float ms;
    ms = time_ms([&] { valu_kernel<1, 2><<<1, 64>>>(d, iters, 1.0000001, 1e-9); });
    printf("1 wave  add  ILP1 : %.2f ns/op\n", ms * 1e6 / (iters * 8.0));
    ms = time_ms([&] { valu_kernel<1, 0><<<1, 64>>>(d, iters, 1.0000001, 1e-9); });
    printf("1 wave  fma  ILP1 : %.2f ns/op\n", ms * 1e6 / (iters * 8.0));
    ms = time_ms([&] { valu_kernel<8, 0><<<1, 64>>>(d, iters, 1.0000001, 1e-9); });
    printf("1 wave  fma  ILP8 : %.2f ns/op\n", ms * 1e6 / (iters * 64.0));
    ms = time_ms([&] { valu_kernel<8, 2><<<1, 64>>>(d, iters, 1.0000001, 1e-9); });
    printf("1 wave  add  ILP8 : %.2f ns/op\n", ms * 1e6 / (iters * 64.0));
    ms = time_ms([&] { valu_kernel<8, 3><<<1, 64>>>(d, iters, 1.0000001, 1e-9); });
    printf("1 wave  rcp  ILP8 : %.2f ns/op (v_rcp_f64; against fma ILP8: cycles per wave instruction = 4 x the ratio)\n", ms * 1e6 / (iters * 64.0));
    ms = time_ms([&] { dpp_chain_kernel<<<1, 64>>>(d, iters, 1e-9); });
    printf("1 wave  dpp+add chain : %.2f ns/step\n", ms * 1e6 / (iters * 8.0));
    ms = time_ms([&] { shfl_chain_kernel<<<1, 64>>>(d, iters, 1e-9); });
    printf("1 wave  shfl(bpermute)+add chain : %.2f ns/step\n", ms * 1e6 / (iters * 8.0));
    ms = time_ms([&] { div_chain_kernel<<<1, 64>>>(d, iters / 4, 3.0); });
    printf("1 wave  div chain : %.2f ns/op\n", ms * 1e6 / (iters / 4 * 8.0));
  }
  // --- chip-wide throughput at 1, 2, 4, 8 waves per SIMD
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = p.multiProcessorCount * wps;  // 256 threads = 4 waves = 1 per SIMD
    const double ops = double(blocks) * 256 * iters * 8.0 * 8;
    float ms = time_ms([&] { valu_kernel<8, 0><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9); });
    printf("%d waves/SIMD fma    : %.1f TFLOP/s (fma = 2 flop)\n", wps, 2 * ops / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { valu_kernel<8, 1><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9); });
    printf("%d waves/SIMD mul+add: %.1f TFLOP/s (2 instr = 2 flop)\n", wps, 2 * ops / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { valu_kernel<8, 3><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9); });
    printf("%d waves/SIMD rcp    : %.2f T v_rcp_f64 lanes/s\n", wps, ops / (ms * 1e-3) / 1e12);
  }
  hipFree(d);
  return 0;
}
