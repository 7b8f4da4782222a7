// Device check of storm_amd/csrc/shared_div.hpp: shared_div(a, prepare(x)) == a / x bit for bit.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/shared_div_check.bin tools/shared_div_check.hip
// run:   tools/shared_div_check.bin [pairs in millions, default 2048]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../storm_amd/csrc/shared_div.hpp"

__device__ uint64_t mix(uint64_t z) {  // splitmix64
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__device__ double make(uint64_t bits, int mode) {
  // mode 0: any finite-or-not bit pattern; 1: moderate exponents; 2: mantissa edge cases with moderate exponents
  if (mode == 0) return __longlong_as_double((long long)bits);
  uint64_t sign = bits >> 63, man = bits & 0xfffffffffffffULL;
  uint64_t ex = 1023 - 420 + (bits >> 52) % 841;  // straddles the +-400 fast-path limits
  if (mode == 2) {
    const uint64_t edges[8] = {0, 1, 2, 0xfffffffffffffULL, 0xffffffffffffeULL, 0x8000000000000ULL, 0x7ffffffffffffULL, 0x8000000000001ULL};
    man = edges[(bits >> 40) & 7];
  }
  return __longlong_as_double((long long)((sign << 63) | (ex << 52) | man));
}
__global__ void check(uint64_t seed, unsigned long long* bad, unsigned long long* fast, double* example) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nbad = 0, nfast = 0;
  for (int it = 0; it < 256; ++it) {
    const uint64_t h = mix(seed + id * 256 + it);
    const int mode = (h & 15) == 0 ? 0 : ((h & 15) < 4 ? 2 : 1);
    const double x = make(mix(h), mode);
    const storm::SharedDenominator d = storm::shared_div_prepare(x);
    double batch[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) batch[k] = make(mix(h + 77 * (k + 1)), (k < 4 || (h & 64)) ? mode : 1);
    storm::shared_div6(batch, d);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double a = make(mix(h + 77 * (k + 1)), mode);
      const double ref = a / x, got = (h & 32) ? batch[k] : storm::shared_div(a, d);
      const long long rb = __double_as_longlong(ref), gb = __double_as_longlong(got);
      const bool both_nan = ref != ref && got != got;
      if (rb != gb && !both_nan) {
        if (nbad == 0 && atomicAdd(bad, 0ULL) == 0) { example[0] = a; example[1] = x; example[2] = ref; example[3] = got; }
        ++nbad;
      }
      if (d.ok && storm::shared_div_moderate(a)) ++nfast;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  atomicAdd(fast, nfast);
}
int main(int argc, char** argv) {
  const unsigned long long millions = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2048;
  unsigned long long *d_bad, *d_fast, h[2] = {0, 0};
  double *d_ex, ex[4] = {0, 0, 0, 0};
  if (hipMalloc(&d_bad, 8) != hipSuccess || hipMalloc(&d_fast, 8) != hipSuccess || hipMalloc(&d_ex, 32) != hipSuccess) return 2;
  (void)hipMemset(d_bad, 0, 8); (void)hipMemset(d_fast, 0, 8); (void)hipMemset(d_ex, 0, 32);
  const unsigned long long per_launch = 4096ULL * 256 * 256 * 4;  // pairs per launch
  unsigned long long done = 0;
  for (uint64_t s = 1; done < millions * 1000000ULL; ++s, done += per_launch) {
    check<<<4096, 256>>>(s * 0x100000000ULL, d_bad, d_fast, d_ex);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  }
  (void)hipMemcpy(&h[0], d_bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&h[1], d_fast, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(ex, d_ex, 32, hipMemcpyDeviceToHost);
  printf("pairs %llu  fast-path %llu  mismatches %llu\n", done, h[1], h[0]);
  if (h[0]) printf("example: a=%a x=%a  a/x=%a  shared=%a\n", ex[0], ex[1], ex[2], ex[3]);
  return h[0] ? 1 : 0;
}
