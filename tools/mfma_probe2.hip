// Hardware probe 2 (not product code): f64 MFMA rate vs accumulator count and waves/SIMD; MFMA+VALU co-issue.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma64(double* out, int iters, double a0, double b0) {
  d4 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// MFMA + independent VALU fma chain in the same wave
template <int NFMA>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, double a0, double b0) {
  d4 c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  double v[8]; for (int i = 0; i < 8; ++i) v[i] = i;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NFMA; ++t) v[(j * NFMA + t) & 7] = fma(a, v[(j * NFMA + t) & 7], b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; CK(hipMalloc(&out, 8 * 256 * 8192));
  const int cus = 256; int iters = 10000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = cus * wps;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma64<1>, blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("waves/SIMD=%d acc=1: %.2f TF\n", wps, (double)blocks * 4 * iters * 1 * 2048.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma64<2>, blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("waves/SIMD=%d acc=2: %.2f TF\n", wps, (double)blocks * 4 * iters * 2 * 2048.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma64<4>, blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("waves/SIMD=%d acc=4: %.2f TF\n", wps, (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma64<8>, blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("waves/SIMD=%d acc=8: %.2f TF\n", wps, (double)blocks * 4 * iters * 8 * 2048.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma64<16>, blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("waves/SIMD=%d acc=16: %.2f TF\n", wps, (double)blocks * 4 * iters * 16 * 2048.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mix<1>, blocks, 256, 0, 0, out, iters, 0.999, 0.5); });
    printf("waves/SIMD=%d mix 1 fma/mfma: mfma %.2f TF + valu %.2f TF, %.3f ms\n", wps, (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9, (double)blocks * 256 * iters * 4 * 1 * 2.0 / ms / 1e9, ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_mix<4>, blocks, 256, 0, 0, out, iters, 0.999, 0.5); });
    printf("waves/SIMD=%d mix 4 fma/mfma: mfma %.2f TF + valu %.2f TF, %.3f ms\n", wps, (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9, (double)blocks * 256 * iters * 4 * 4 * 2.0 / ms / 1e9, ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_mix<8>, blocks, 256, 0, 0, out, iters, 0.999, 0.5); });
    printf("waves/SIMD=%d mix 8 fma/mfma: mfma %.2f TF + valu %.2f TF, %.3f ms\n", wps, (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9, (double)blocks * 256 * iters * 4 * 8 * 2.0 / ms / 1e9, ms);
  }
  return 0;
}
