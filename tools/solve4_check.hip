// solve4_check -- runs the FMA build's 4x4 device routines (mat4.hpp: fast_recip, m4_inverse_fast, lu4_factor/lu4_solve,
// m4_solve_n) on matrices from a file, one lane per problem, so tests/test_hip_solve4.py can compare them with numpy.
// The kernels of libstorm_hip.so inline these routines; nothing here is part of the library.
//
//   solve4_check.bin <in> <out>
//   in : int64 n, then n x (16 doubles A column-major, 16 doubles B column-major)
//   out: n x (16 doubles X = A^-1 B by m4_solve_n, 16 doubles scale*A^-1 by m4_inverse_fast with scale = -2,
//             1 double: 1.0 if lu4_factor accepted A, 0.0 if the adjugate fallback ran, 1 double: ok flag, 1 double: fast_recip(A[0]))
#define STORM_NS storm_fast
#define STORM_FAST_BUILD 1
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../storm_amd/csrc/mat4.hpp"

using namespace storm_fast;

constexpr int kOut = 35;

__global__ void check_kernel(const double* __restrict__ in, double* __restrict__ out, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  M4 a, b;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    a.v[e] = in[t * 32 + e];
    b.v[e] = in[t * 32 + 16 + e];
  }
  bool ok = true;
  M4 x[1] = {b};
  m4_solve_n<1>(a, x, ok);
  bool ok_inv = true;
  const M4 inv = m4_inverse_fast(a, ok_inv, -2.0);
  Lu4 f;
  const bool accepted = lu4_factor(a, f);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    out[t * kOut + e] = x[0].v[e];
    out[t * kOut + 16 + e] = inv.v[e];
  }
  out[t * kOut + 32] = accepted ? 1.0 : 0.0;
  out[t * kOut + 33] = (ok && ok_inv) ? 1.0 : 0.0;
  out[t * kOut + 34] = fast_recip(a.v[0]);
}

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s <in> <out>\n", argv[0]);
    return 1;
  }
  std::FILE* fi = std::fopen(argv[1], "rb");
  if (!fi) return 1;
  int64_t n = 0;
  if (std::fread(&n, 8, 1, fi) != 1 || n <= 0 || n > (1 << 24)) return 1;
  std::vector<double> in((size_t)n * 32), out((size_t)n * kOut);
  if (std::fread(in.data(), 8, in.size(), fi) != in.size()) return 1;
  std::fclose(fi);
  double *d_in = nullptr, *d_out = nullptr;
  CHECK(hipMalloc(&d_in, in.size() * 8));
  CHECK(hipMalloc(&d_out, out.size() * 8));
  CHECK(hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice));
  check_kernel<<<(unsigned)((n + 63) / 64), 64>>>(d_in, d_out, (long long)n);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost));
  std::FILE* fo = std::fopen(argv[2], "wb");
  if (!fo) return 1;
  std::fwrite(out.data(), 8, out.size(), fo);
  std::fclose(fo);
  CHECK(hipFree(d_in));
  CHECK(hipFree(d_out));
  return 0;
}
