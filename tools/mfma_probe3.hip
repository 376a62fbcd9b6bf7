// Hardware probe 3 (not product code): in-kernel clock under f64 MFMA vs f64 VALU load; VALU f64 peak search.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

__global__ __launch_bounds__(256) void k_mfma64(double* out, unsigned long long* st, int iters, double a0, double b0) {
  d4 c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NCH>
__global__ __launch_bounds__(256) void k_fma64(double* out, unsigned long long* st, int iters, double a0, double b0) {
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  double c[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) c[i] = i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) c[j] = fma(a, c[j], b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}
// scalar-operand form: c[j] = fma(s_k, v, c[j]) with s uniform (SGPR pair)
template <int NCH>
__global__ __launch_bounds__(256) void k_fma64s(double* out, const double* __restrict__ sc, int iters, double a0) {
  double v = a0 + threadIdx.x * 1e-9;
  double c[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) c[i] = i;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) c[j] = fma(sc[(i * NCH + j) & 1023], v, c[j]);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static double clk(std::vector<unsigned long long>& st, int blocks) {
  std::vector<double> r; for (int i = 0; i < blocks; ++i) r.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0);
  std::sort(r.begin(), r.end()); return r[r.size() / 2];
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; CK(hipMalloc(&out, 8 * 256 * 8192));
  double* sc; CK(hipMalloc(&sc, 8 * 1024)); { std::vector<double> h(1024, 0.999); CK(hipMemcpy(sc, h.data(), 8192, hipMemcpyHostToDevice)); }
  unsigned long long* st; CK(hipMalloc(&st, 16 * 8192));
  std::vector<unsigned long long> h(2 * 8192);
  const int cus = 256;
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = cus * wps; float ms;
    if (wps <= 4) {
      int iters = 40000;
      ms = timeit([&] { hipLaunchKernelGGL(k_mfma64, blocks, 256, 0, 0, out, st, iters, 1.0, 0.5); });
      CK(hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost));
      printf("MFMA f64 waves/SIMD=%d: %.2f TF, %.2f ms, in-kernel clock %.0f MHz\n", wps, (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9, ms, clk(h, blocks));
    }
    int iters = 20000;
    ms = timeit([&] { hipLaunchKernelGGL(k_fma64<8>, blocks, 256, 0, 0, out, st, iters, 0.999, 0.5); });
    CK(hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost));
    printf("VALU f64 8ch waves/SIMD=%d: %.2f TF, %.2f ms, clock %.0f MHz\n", wps, (double)blocks * 256 * iters * 8 * 2.0 / ms / 1e9, ms, clk(h, blocks));
    ms = timeit([&] { hipLaunchKernelGGL(k_fma64<16>, blocks, 256, 0, 0, out, st, iters, 0.999, 0.5); });
    CK(hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost));
    printf("VALU f64 16ch waves/SIMD=%d: %.2f TF, %.2f ms, clock %.0f MHz\n", wps, (double)blocks * 256 * iters * 16 * 2.0 / ms / 1e9, ms, clk(h, blocks));
    ms = timeit([&] { hipLaunchKernelGGL(k_fma64s<16>, blocks, 256, 0, 0, out, sc, iters, 0.5); });
    printf("VALU f64 sgpr-operand 16ch waves/SIMD=%d: %.2f TF, %.2f ms\n", wps, (double)blocks * 256 * iters * 16 * 2.0 / ms / 1e9, ms);
  }
  return 0;
}
