// Latency / issue microbenchmarks for the instruction classes of the Riccati sweep (one wave, gfx950).
// Build:  hipcc --offload-arch=gfx950 -O3 -o lat lat.hip ; run on the GPU box: ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__device__ __forceinline__ double rl(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
template <int MODE>
__global__ void __launch_bounds__(64, 2) k(double* out, long long* cyc, long long* rt, double a, double b) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 v4 = {a, b, a, b};
  double x = a + threadIdx.x, y = b, z = a * 3 + threadIdx.x, w = b - threadIdx.x;
  const int lane = threadIdx.x;
  long long r0 = wall_clock64();
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) x = __builtin_fma(x, y, a);                                    // dependent fp64 FMA chain
      if (MODE == 1) { x = __builtin_fma(x, y, a); z = __builtin_fma(z, y, a); w = __builtin_fma(w, y, a); y = __builtin_fma(y, b, a); }  // 4 independent chains
      if (MODE == 2) x = __shfl(x, (lane + 8) & 63, 64) + a;                          // bpermute round trip + add
      if (MODE == 3) { double d = __builtin_amdgcn_mov_dpp(x, 0x153, 0xf, 0xf, false); x = __builtin_amdgcn_update_dpp(d, x, 0x15b, 0xf, 0xc, false) + a; }
      if (MODE == 4) x = rl(x, 9) + a;                                                // readlane broadcast + add
      if (MODE == 5) x = __builtin_amdgcn_rcp(x) + a;                                 // v_rcp_f64 + add
      if (MODE == 6) { int lo = __double2loint(x), hi = __double2hiint(x);           // quad_perm DPP (2 x 32 bit) + add
        x = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true)) + a; }
      if (MODE == 8) { v4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, v4, 0, 0, 0); }                                  // dependent accumulate chain
      if (MODE == 9) { v4 = __builtin_amdgcn_mfma_f64_16x16x4f64(v4[0], y, v4, 0, 0, 0); }                              // result feeds the A operand too
      if (MODE == 10) { v4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, v4, 0, 0, 0); x = v4[1] * a; }                   // MFMA -> VALU -> MFMA
      if (MODE == 7) { x = __shfl(x, (lane + 8) & 63, 64) + a; z = __shfl(z, (lane + 8) & 63, 64) + a; w = __shfl(w, (lane + 8) & 63, 64) + a; }
    }
  }
  long long t1 = clock64();
  long long r1 = wall_clock64();
  out[threadIdx.x + 64 * blockIdx.x] = x + y + z + w + v4[0] + v4[1] + v4[2] + v4[3];
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}
template <int MODE>
void run(const char* name, int blocks) {
  double* out; long long *cyc, *rt;
  hipMalloc(&out, 64 * 8 * blocks); hipMalloc(&cyc, 8 * blocks); hipMalloc(&rt, 8 * blocks);
  for (int rep = 0; rep < 2; ++rep) k<MODE><<<blocks, 64>>>(out, cyc, rt, 1.0000001, 0.9999999);
  hipDeviceSynchronize();
  long long c, r; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&r, rt, 8, hipMemcpyDeviceToHost);
  int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
  printf("%-44s blocks=%5d  %7.1f clk64/iter  %8.2f ns/iter (wall clock %d kHz)\n", name, blocks, (double)c / N, (double)r / N * 1e6 / wcr, wcr);
  hipFree(out); hipFree(cyc); hipFree(rt);
}
int main() {
  for (int blocks : {1, 1024}) {
    run<0>("dependent v_fma_f64", blocks);
    run<1>("4 independent v_fma_f64", blocks);
    run<2>("ds_bpermute f64 (2 x b32) + add", blocks);
    run<7>("3 independent bpermute f64 + add", blocks);
    run<3>("2 x v_mov_b64_dpp row_newbcast + add", blocks);
    run<6>("2 x v_mov_b32_dpp quad_perm + add", blocks);
    run<4>("2 x v_readlane + add", blocks);
    run<5>("v_rcp_f64 + add", blocks);
    run<8>("mfma_f64_16x16x4 accumulate chain", blocks);
    run<9>("mfma_f64_16x16x4, D -> A and C", blocks);
    run<10>("mfma_f64_16x16x4 -> v_mul_f64 -> A", blocks);
  }
  return 0;
}
