#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ float ag_sqrt_f32_rn(float x) {
  const float y = __builtin_amdgcn_sqrtf(x);
  const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
  const float rm = __fmaf_rn(-ym, y, x), rp = __fmaf_rn(-yp, y, x);
  float r = rm <= 0.0f ? ym : y;
  r = rp > 0.0f ? yp : r;
  return r;
}
__global__ void k(unsigned lo, unsigned n, unsigned long long* bad, unsigned* first) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += gridDim.x * blockDim.x) {
    float x = __uint_as_float(lo + i);
    float a = ag_sqrt_f32_rn(x);
    float b = (float)sqrt((double)x);   // correctly rounded: double sqrt then round (innocuous double rounding for sqrt)
    if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) { atomicAdd(bad, 1ull); atomicMin(first, lo + i); }
  }
}
int main() {
  unsigned long long* bad; unsigned* first; hipMalloc(&bad, 8); hipMalloc(&first, 4);
  hipMemset(bad, 0, 8); hipMemset(first, 0xff, 4);
  // every float in [2^-96, 1] and zero, a band below, and some negatives
  unsigned lo = 0x0f800000u /* 2^-96 */, hi = 0x3f800000u;
  k<<<4096, 256>>>(lo, hi - lo + 1, bad, first);
  unsigned long long hb; unsigned hf; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
  printf("[2^-96,1]: mismatches %llu first %08x\n", hb, hf);
  hipMemset(bad, 0, 8); hipMemset(first, 0xff, 4);
  k<<<4096, 256>>>(0u, lo, bad, first);
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
  printf("[0,2^-96): mismatches %llu first %08x\n", hb, hf);
  return 0;
}
