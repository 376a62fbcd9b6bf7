// micro-benchmark of k_bl_gradcoarse on zero operands (config H's shapes).  Build after `make -C safe-bayesian-optimization_amd/csrc`:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I include -I safe-bayesian-optimization_amd/csrc -c tools/dev/gradcoarse_bench.hip -o /tmp/gcb.o
//   (cd safe-bayesian-optimization_amd/csrc && hipcc --offload-arch=gfx950 -o ../../tools/dev/gradcoarse_bench /tmp/gcb.o api.o model.o posterior.o sets.o comm.o fit.o plant.o tensor.o guard.o -L/opt/rocm/lib -lrccl)
#include "bilinear.hip"
#include <cstdio>
int main() {
  using namespace sbo;
  const int q = 2, r0 = 22, cnt0 = 4096, nlines = 4096;
  BlDims dm{};
  dm.q = q; dm.r0u = r0; dm.cnt0 = cnt0; dm.nlines = nlines;
  for (int o = 0; o < q; ++o) dm.r0[o] = r0;
  double *S0, *Vb, *xn0, *xn1, *tmax; unsigned long long* gkey;
  hipMalloc(&S0, sizeof(double) * q * r0 * cnt0);
  hipMalloc(&Vb, sizeof(double) * q * 3 * r0 * nlines);
  hipMalloc(&xn0, sizeof(double) * cnt0);
  hipMalloc(&xn1, sizeof(double) * nlines);
  const int ntx = cnt0 / 128, nty = nlines / 64, nt = ntx * nty;
  hipMalloc(&tmax, sizeof(double) * q * 2 * nt);
  hipMalloc(&gkey, 64);
  hipMemset(S0, 0, sizeof(double) * q * r0 * cnt0);
  hipMemset(Vb, 0, sizeof(double) * q * 3 * r0 * nlines);
  hipMemset(xn0, 0, sizeof(double) * cnt0);
  hipMemset(xn1, 0, sizeof(double) * nlines);
  hipMemset(gkey, 0, 64);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a, 0);
    for (int k = 0; k < 10; ++k)
      hipLaunchKernelGGL(k_bl_gradcoarse, dim3(nt, q), dim3(128), 0, 0, dm, (const double*)S0, (const double*)Vb, (const double*)xn0, (const double*)xn1, ntx, tmax, gkey);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("k_bl_gradcoarse: %.1f us per launch\n", ms * 100.0f);
  }
  return 0;
}
