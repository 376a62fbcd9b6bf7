// Hardware probe (not product code): measures v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 /
// v_fma_f64 issue rates and fp64 exp() throughput on gfx950, and verifies the f64 MFMA operand and
// C/D lane maps with exact integer data.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

__global__ __launch_bounds__(256) void k_mfma64(double* out, int iters, double a0, double b0) {
  d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void k_mfma32(float* out, int iters, float a0, float b0) {
  f4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  float a = a0 + threadIdx.x * 1e-6f, b = b0 + threadIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
  }
  f4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void k_fma64(double* out, int iters, double a0, double b0) {
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
  for (int i = 0; i < iters; ++i) {
    c0 = fma(a, c0, b); c1 = fma(a, c1, b); c2 = fma(a, c2, b); c3 = fma(a, c3, b);
    c4 = fma(a, c4, b); c5 = fma(a, c5, b); c6 = fma(a, c6, b); c7 = fma(a, c7, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ __launch_bounds__(256) void k_exp64(double* out, int iters, double a0) {
  double x = a0 - threadIdx.x * 1e-3, s = 0;
  for (int i = 0; i < iters; ++i) { s += exp(x); x -= 1e-6; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// layout check: D = A(16x4) * B(4x16), A[i][k] = i*10+k+1, B[k][j] = (k+1)*100 + j (asymmetric)
__global__ void k_layout(double* D) {
  int l = threadIdx.x;
  double a = (double)((l & 15) * 10 + (l >> 4) + 1);
  double b = (double)(((l >> 4) + 1) * 100 + (l & 15));
  d4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
__global__ void k_layout32(float* D) {
  int l = threadIdx.x;
  float a = (float)((l & 15) * 10 + (l >> 4) + 1);
  float b = (float)(((l >> 4) + 1) * 100 + (l & 15));
  f4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  // layout
  { double* d; CK(hipMalloc(&d, 256 * 8)); hipLaunchKernelGGL(k_layout, 1, 64, 0, 0, d); std::vector<double> h(256); CK(hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double r = 0; for (int k = 0; k < 4; ++k) r += (i * 10 + k + 1) * ((k + 1) * 100 + j); if (r != h[i * 16 + j]) ++bad; }
    printf("f64 16x16x4 layout check: %d mismatches\n", bad); hipFree(d); }
  { float* d; CK(hipMalloc(&d, 256 * 4)); hipLaunchKernelGGL(k_layout32, 1, 64, 0, 0, d); std::vector<float> h(256); CK(hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float r = 0; for (int k = 0; k < 4; ++k) r += (i * 10 + k + 1) * ((k + 1) * 100 + j); if (r != h[i * 16 + j]) ++bad; }
    printf("f32 16x16x4 layout check: %d mismatches\n", bad); hipFree(d); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double* out; CK(hipMalloc(&out, 8 * 256 * 4096));
  int cus = p.multiProcessorCount;
  for (int wpc = 1; wpc <= 2; ++wpc) {   // blocks per CU (256 threads = 1 wave / SIMD each)
    int blocks = cus * wpc; int iters = 20000; float ms;
    hipLaunchKernelGGL(k_mfma64, blocks, 256, 0, 0, out, 100, 1.0, 0.5); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma64, blocks, 256, 0, 0, out, iters, 1.0, 0.5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    double fl = (double)blocks * 4 * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4 blocks/CU=%d: %.3f ms, %.2f TFLOP/s, cycles/MFMA/SIMD at 2.4GHz = %.1f\n", wpc, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * wpc));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma32, blocks, 256, 0, 0, (float*)out, iters, 1.0f, 0.5f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mfma_f32_16x16x4 blocks/CU=%d: %.3f ms, %.2f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma64, blocks * 4, 256, 0, 0, out, iters, 0.999, 0.5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("v_fma_f64 blocks/CU=%d: %.3f ms, %.2f TFLOP/s\n", wpc * 4, ms, (double)blocks * 4 * 256 * iters * 8 * 2.0 / ms / 1e9);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_exp64, blocks * 4, 256, 0, 0, out, 2000, -0.5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("exp(f64) blocks/CU=%d: %.3f ms, %.2f Gexp/s\n", wpc * 4, ms, (double)blocks * 4 * 256 * 2000 / ms / 1e6);
  }
  return 0;
}
