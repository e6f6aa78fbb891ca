// microbenchmark: fp64 FMA dependent-chain latency / throughput, ds_bpermute and LDS round trips (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template <int CH>
__global__ void k_fma(double *out, int iters, double a, double b) {
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_shfl(double *out, int iters) {
  double x = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x = __shfl_up(x, 1, 8) + 1.0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_lds(double *out, int iters) {
  __shared__ double s[64];
  double x = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[threadIdx.x] = x;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      x = s[(threadIdx.x + 1) & 63] + 1.0;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <class F> double timeit(F f) {
  f(); hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  f(); hipDeviceSynchronize();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
  double *out; hipMalloc(&out, 8 << 20);
  const int iters = 20000;
  const double clk = 2.1e9;  // rough
  for (int waves = 1; waves <= 8; waves *= 2) {
    double t1 = timeit([&] { hipLaunchKernelGGL(k_fma<1>, dim3(256 * 4), dim3(64 * waves / 4 > 64 ? 64 : 64), 0, 0, out, iters, 1.0000001, 1e-9); });
    (void)t1;
  }
  // one wave per SIMD (1024 blocks of 64): dependent chain
  double t = timeit([&] { hipLaunchKernelGGL(k_fma<1>, dim3(1024), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf("fma dep chain, 1 wave/SIMD: %.2f cycles per fma\n", t * clk / (iters * 16.0));
  t = timeit([&] { hipLaunchKernelGGL(k_fma<2>, dim3(1024), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf("fma 2 chains, 1 wave/SIMD: %.2f cycles per fma\n", t * clk / (iters * 32.0));
  t = timeit([&] { hipLaunchKernelGGL(k_fma<4>, dim3(1024), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf("fma 4 chains, 1 wave/SIMD: %.2f cycles per fma\n", t * clk / (iters * 64.0));
  t = timeit([&] { hipLaunchKernelGGL(k_fma<8>, dim3(1024), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf("fma 8 chains, 1 wave/SIMD: %.2f cycles per fma\n", t * clk / (iters * 128.0));
  t = timeit([&] { hipLaunchKernelGGL(k_fma<1>, dim3(2048), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf("fma dep chain, 2 waves/SIMD: %.2f cycles per fma per wave\n", t * clk / (iters * 16.0));
  t = timeit([&] { hipLaunchKernelGGL(k_fma<1>, dim3(4096), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf("fma dep chain, 4 waves/SIMD: %.2f cycles per fma per wave\n", t * clk / (iters * 16.0));
  t = timeit([&] { hipLaunchKernelGGL(k_shfl, dim3(1024), dim3(64), 0, 0, out, iters / 10); });
  printf("shfl_up(double)+add dep chain: %.1f cycles per round\n", t * clk / (iters / 10 * 16.0));
  t = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(1024), dim3(64), 0, 0, out, iters / 10); });
  printf("lds write->read dep chain: %.1f cycles per round\n", t * clk / (iters / 10 * 16.0));
  return 0;
}
