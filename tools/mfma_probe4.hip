// Hardware probe 4 (not product code): v_mfma_f64_4x4x4_4b_f64 rate and lane layout (incl. cbsz/abid broadcast).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template <int NACC, int CBSZ>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters, double a0, double b0) {
  double c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = 0;
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; j += 4) {
      c[j + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j + 0], CBSZ, 0, 0);
      c[j + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j + 1], CBSZ, CBSZ ? 1 : 0, 0);
      c[j + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j + 2], CBSZ, CBSZ ? 2 : 0, 0);
      c[j + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j + 3], CBSZ, CBSZ ? 3 : 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// layout discovery: a = lane id encoded, b = lane id encoded -> find which (a-lane, b-lane) pairs feed each output lane
__global__ void k_layout(double* D, int cbsz, int abid) {
  const int l = threadIdx.x;
  // run 64 x 64 one-hot experiments would be slow; instead use the bilinear trick: a_l = 2^(l%8)... use two passes
  // pass 1: a = 1 at all lanes, b = one-hot -> which b lanes reach output lane (D1[lb][l])
  for (int lb = 0; lb < 64; ++lb) {
    double a = 1.0, b = (l == lb) ? 1.0 : 0.0, c = 0.0;
    if (cbsz == 0) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    else if (abid == 0) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 0, 0);
    else if (abid == 1) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 1, 0);
    else if (abid == 2) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 2, 0);
    else c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 3, 0);
    D[lb * 64 + l] = c;
  }
  for (int la = 0; la < 64; ++la) {
    double a = (l == la) ? 1.0 : 0.0, b = 1.0, c = 0.0;
    if (cbsz == 0) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    else if (abid == 0) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 0, 0);
    else if (abid == 1) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 1, 0);
    else if (abid == 2) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 2, 0);
    else c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 3, 0);
    D[4096 + la * 64 + l] = c;
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; CK(hipMalloc(&out, 8 * 256 * 8192));
  const int cus = 256; int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = cus * wps; float ms;
    ms = timeit([&] { hipLaunchKernelGGL((k_rate<4, 0>), blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("4x4x4 f64 waves/SIMD=%d acc=4 cbsz=0: %.2f TF (%.1f cyc/inst at 2.4GHz)\n", wps, (double)blocks * 4 * iters * 4 * 512.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * wps));
    ms = timeit([&] { hipLaunchKernelGGL((k_rate<8, 0>), blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("4x4x4 f64 waves/SIMD=%d acc=8 cbsz=0: %.2f TF\n", wps, (double)blocks * 4 * iters * 8 * 512.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k_rate<8, 2>), blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("4x4x4 f64 waves/SIMD=%d acc=8 cbsz=2 (A broadcast): %.2f TF\n", wps, (double)blocks * 4 * iters * 8 * 512.0 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k_rate<16, 2>), blocks, 256, 0, 0, out, iters, 1.0, 0.5); });
    printf("4x4x4 f64 waves/SIMD=%d acc=16 cbsz=2 (A broadcast): %.2f TF\n", wps, (double)blocks * 4 * iters * 16 * 512.0 / ms / 1e9);
  }
  double* D; CK(hipMalloc(&D, 8 * 8192)); std::vector<double> h(8192);
  for (int mode = 0; mode < 3; ++mode) {
    int cbsz = mode == 0 ? 0 : 2, abid = mode == 2 ? 1 : 0;
    hipLaunchKernelGGL(k_layout, 1, 64, 0, 0, D, cbsz, abid); CK(hipMemcpy(h.data(), D, 8 * 8192, hipMemcpyDeviceToHost));
    printf("layout cbsz=%d abid=%d: for output lanes 0,1,4,5,16,17,20,63: contributing B lanes / A lanes\n", cbsz, abid);
    int outs[8] = {0, 1, 4, 5, 16, 17, 20, 63};
    for (int oi = 0; oi < 8; ++oi) { int l = outs[oi];
      printf("  out lane %2d: B lanes:", l); for (int lb = 0; lb < 64; ++lb) if (h[lb * 64 + l] != 0) printf(" %d", lb);
      printf(" | A lanes:"); for (int la = 0; la < 64; ++la) if (h[4096 + la * 64 + l] != 0) printf(" %d", la); printf("\n"); }
  }
  return 0;
}
