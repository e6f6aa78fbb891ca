// agx_big_k2.hpp -- K2 for large models (16 < nv <= 32), blocked factorisation on the fp64 matrix cores.
//
// Same recursion as k_riccati_mfma (agx_big_k1.hpp; mim_solvers SolverCSQP backwardPass / computeDirection as called from
// agimus_controller/ocp_base_croco.py:172, acceleration-input form of DESIGN.md section 4), one 256-thread workgroup per
// instance.  What changed is how  Kw = Qww^-1 Qwx  is formed.  k_riccati_mfma eliminates [Qww | a quarter of the right-hand
// sides] on every wave, a row per lane, 30 pivots whose multipliers travel through v_readlane: 945 column updates of three
// instructions per wave and node, four times over (the elimination of Qww itself is redundant on every wave), on a
// chain of 51 nodes.  Here
//   * wave 0 inverts Qww (padded to 32 x 32 with a unit diagonal) as a 2 x 2 block matrix of 16 x 16 tiles held in the
//     accumulator layout of v_mfma_f64_16x16x4 (lane 16 g + j: column j of rows g + 4 r):
//         inv11 = A11^-1 (in-place Gauss-Jordan inside the tile: 16 pivots, ~40 instructions each, no LDS)
//         W = inv11 A12,  S = A22 - A12' W,  invS = S^-1 (16 pivots),
//         B21 = -invS W',  B11 = inv11 - W B21,  B22 = invS
//     -- the five tile products are 20 MFMAs with the accumulator-layout registers as operands
//     (mfma(X[r], Y[r]) summed over r is X' Y, see agx_riccati_mx.hpp);
//   * every wave then forms its 16 columns of  Kw = Qww^-1 Qwx  on the matrix cores (16 MFMAs, operands from LDS),
//   * V = Qxx - Qxw Kw as before.
// The vector part (vp = vx + V f, kw, the gradient of the value function) is spread over four lanes per row.
#pragma once

namespace agx {

// acc += X' Y for two 16 x 16 tiles in the accumulator layout (NEG: acc -= X' Y)
template <bool NEG = false>
__device__ __forceinline__ agx_v4d tile_xty(const agx_v4d &X, const agx_v4d &Y, agx_v4d acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? flip_sign(X[r]) : X[r], Y[r], acc, 0, 0, 0);
  return acc;
}

// One pivot of the in-place Gauss-Jordan inversion of a 16 x 16 tile in the accumulator layout (g = lane >> 4, j = lane & 15;
// x[r] = X[g + 4 r][j]):  p = 1 / a_kk;  a_kj <- a_kj p;  a_ij <- a_ij - a_ik a_kj (i, j != k);  a_ik <- -a_ik p;  a_kk <- p.
template <int K>
__device__ __forceinline__ void tile_inv_pivot(agx_v4d &x, const int g, const int j, bool &bad) {
  constexpr int rk = K >> 2, gk = K & 3;
  const double piv = readlane_f64(x[rk], 16 * gk + K);
  bad = bad || !(piv > 0.0);  // the matrix is not positive definite (or not finite): the checker's LLT fails at the same place
  const double rp = chain_rcp(piv);
  const double rowk = __shfl(x[rk], 16 * gk + j, 64);  // the pivot row at my column
  const bool colk = (j == K);
  const double m = colk ? rp : rowk * rp;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double ck = row_bcast<K>(x[r]);  // column k at my rows
    const double x0 = colk ? 0.0 : x[r];
    double nv = __builtin_fma(-ck, m, x0);
    if (r == rk) nv = (g == gk) ? m : nv;
    x[r] = nv;
  }
}
template <int K0, int K1>
__device__ __forceinline__ void tile_inv_range(agx_v4d &x, const int g, const int j, bool &bad) {
  if constexpr (K0 < K1) {
    tile_inv_pivot<K0>(x, g, j, bad);
    tile_inv_range<K0 + 1, K1>(x, g, j, bad);
  }
}
// inverse of a symmetric positive definite tile (no pivoting), in place; bad: some pivot was not positive
__device__ __forceinline__ void tile_inverse(agx_v4d &x, const int g, const int j, bool &bad) { tile_inv_range<0, 16>(x, g, j, bad); }

// (Tried and discarded, round 3: the rank-one update of a pivot as ONE MFMA -- for a symmetric tile both operands are the
// pivot row's own registers -- with the next pivot's reciprocal formed ahead of the update.  Fewer instructions, but the
// dependent MFMA -> v_readlane -> MFMA chain is longer than the DPP / ds_bpermute one: inversion phase 3.7 -> 5.0 us per
// node, and treating the tile as exactly symmetric cost the 31-link chain its parity.)
// Development builds (-DAGX_BLK_STAMP): thread 0 of workgroup 0 sums the time between phase boundaries (100 MHz ticks)
#ifdef AGX_BLK_STAMP
#define AGX_BLK_T(i)                                                                     \
  do {                                                                                   \
    if (tid == 0 && b == 0) { const long long now_ = wall_clock64(); ph_[i] += now_ - last_; last_ = now_; } \
  } while (0)
#else
#define AGX_BLK_T(i) do { } while (0)
#endif

template <int NV>
__global__ void __launch_bounds__(256, 2) k_riccati_blk(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                        const double *__restrict__ qts, double *__restrict__ Kws,
                                                        double *__restrict__ kws, double *__restrict__ dxs,
                                                        double *__restrict__ wss, DevState *__restrict__ st, int forward,
                                                        int gains_pass) {
  static_assert(NV >= 16 && NV <= 32 && NV % 2 == 0, "Qww: one 16 x 16 tile (nv = 16) or 2 x 2; the value function: 2 x 2 or 4 x 4");
  constexpr int NX = 2 * NV, NWT = (NV + 15) / 16, NW = 16 * NWT;  // tiles / padded size of Qww
  constexpr int NXT = (NX + 15) / 16, NTW = NXT * NXT / 4;         // tiles per side of the value function, tiles per wave of its update
  constexpr int LV = NX + 1, LQ = NW + 1;
  typedef QT<NV> Q;
  // V: the value function of node t + 1 as the update left it (not symmetrised, no dreg) until the element-wise pass of node t
  // replaces it by  Qxx = Hxx + Phi' (sym V + dreg) Phi,  the start value of the update's accumulators
  // Qxw row NX: qw (the gradient is one more right-hand side; its solution is column NX of Kl).  Qww is inverted in place,
  // Qw0 keeps the matrix for the refinement step.
  __shared__ double V[NX][LV], Qxw[NX + 1][LQ], Qww[NW][LQ], Qw0[NW][LQ], Kl[NV][LV];
  __shared__ double vx[NX], vp[NX], fl[NX], qx[NX], dxl[NX], wl[NV];
  __shared__ int s_bad;  // Qww of some node not positive definite (the checker's LLT fails: the direction is discarded, k_sqp_head)
  constexpr int NT = (NX + 1 + 15) / 16;  // right-hand-side tiles of 16 columns: 4 (nv = 30: the gradient takes a spare column), 5 (nv = 32)
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x, nt = 256, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  // The wave that inverts Qww: the two workgroups a CU holds (512 instances on 256 CUs: blocks i and i + 256 under the
  // round-robin placement) use different SIMDs for it, so that the two serial inversions do not queue behind each other
#ifndef AGX_BLK_INV_WAVE
  const int inv_wave = (blockIdx.x >> 8) & 3;
#else
  const int inv_wave = AGX_BLK_INV_WAVE;
#endif
  DevState &S = st[b];
  if (!gains_pass && (S.done || S.admm_conv)) return;
  // gains_pass: as in k_riccati_mfma
  if (gains_pass) {
    const bool run = gains_pass == 1 || (gains_pass == 4 && !S.done) || (gains_pass == 2 && S.gains_iter != S.dir_iter);
    __syncthreads();  // everyone has read the state before it is written
    if (threadIdx.x == 0) { S.ls_acc = run ? 1 : 0; if (run && gains_pass != 1) S.gains_iter = S.dir_iter; }
    if (!run) return;
  }
  const bool grad = !gains_pass;
  const double dreg = (gains_pass && gains_pass != 4) ? (S.solved ? S.dreg : S.gains_dreg) : S.dreg;
  const double *qb = qts + (long long)b * (T + 1) * Q::SIZE;
  double *Kw = Kws + (long long)b * T * NV * NX, *kw = kws + (long long)b * T * NV;
  // Tile entries travel global -> registers one node ahead (the sweep is a dependent chain: an un-prefetched HBM read per
  // phase would sit on it): this thread's entries [r][c] of the six Hessian blocks, coalesced rows.
  constexpr int NE = (NV * NV + 255) / 256;
  double pw[NE], pq[NE], pv[NE], pqq[NE], pqv[NE], pvv[NE], pf = 0.0, pgw = 0.0, pgq = 0.0, pgv = 0.0;
  auto fetch = [&](int t) {
    const double *tl = qb + (long long)t * Q::SIZE;
#pragma unroll
    for (int n = 0; n < NE; ++n) {
      const int e = tid + 256 * n, r = e / NV, c = e % NV, rc = r * Q::LD + c;
      const bool in = e < NV * NV;
      pw[n] = in ? tl[Q::Hww + rc] : 0.0; pq[n] = in ? tl[Q::Hqw + rc] : 0.0; pv[n] = in ? tl[Q::Hvw + rc] : 0.0;
      pqq[n] = in ? tl[Q::Hqq + rc] : 0.0; pqv[n] = in ? tl[Q::Hqv + rc] : 0.0; pvv[n] = in ? tl[Q::Hvv + rc] : 0.0;
    }
    if (grad) {
      if (tid < NX) pf = tl[Q::f + tid];
      if (tid < NV) { pgw = tl[Q::gw + tid]; pgq = tl[Q::gx + tid]; pgv = tl[Q::gx + NV + tid]; }
    }
  };
  {  // value function of the terminal node; unit pad of Qww
    const double *tt = qb + (long long)T * Q::SIZE;
    for (int e = tid; e < NV * NV; e += nt) {
      const int r = e / NV, c = e % NV;
      const double hqq = tt[Q::Hqq + r * Q::LD + c], hqv = tt[Q::Hqv + r * Q::LD + c], hvv = tt[Q::Hvv + r * Q::LD + c];
      V[r][c] = hqq;
      V[r][NV + c] = hqv;
      V[NV + c][r] = hqv;
      V[NV + r][NV + c] = hvv;
    }
    for (int i = tid; i < NX; i += nt) { vx[i] = grad ? tt[Q::gx + i] : 0.0; vp[i] = 0.0; }
    for (int e = tid; e < NW * LQ; e += nt) {
      const int r = e / LQ, c = e % LQ;
      Qww[r][c] = (r == c && r >= NV) ? 1.0 : 0.0;
      Qw0[r][c] = (r == c && r >= NV) ? 1.0 : 0.0;
    }
    if (tid < LQ) Qxw[NX][tid] = 0.0;
    if (tid == 0) s_bad = 0;
  }
  fetch(T - 1);
  if (tid < NX) fl[tid] = pf;
  __syncthreads();
#ifdef AGX_BLK_STAMP
  long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = wall_clock64();
#endif
  for (int t = T - 1; t >= 0; --t) {
    const double h = dts[t], h2 = h * h;
    // ---- phase 1 (V read only): vp = vx + (V + dreg) f on four lanes per row; this thread's entries of sym V + dreg
    if (grad) {
      const int row = tid >> 2, part = tid & 3;
      double s = 0.0;
      if (row < NX)
        for (int c = part; c < NX; c += 4) s += V[row][c] * fl[c];
      s += dpp_xor1(s);
      s += dpp_xor2(s);
      if (row < NX && part == 0) vp[row] = vx[row] + s + dreg * fl[row];
    }
    double sqq[NE], sqv[NE], svq[NE], svv[NE];
#pragma unroll
    for (int n = 0; n < NE; ++n) {
      const int e = tid + 256 * n, r = e / NV, c = e % NV;
      if (e < NV * NV) {
        const double dg = (r == c) ? dreg : 0.0;
        sqq[n] = 0.5 * (V[r][c] + V[c][r]) + dg;
        sqv[n] = 0.5 * (V[r][NV + c] + V[NV + c][r]);
        svq[n] = 0.5 * (V[NV + r][c] + V[c][NV + r]);
        svv[n] = 0.5 * (V[NV + r][NV + c] + V[NV + c][NV + r]) + dg;
      } else {
        sqq[n] = 0.0; sqv[n] = 0.0; svq[n] = 0.0; svv[n] = 0.0;
      }
    }
    __syncthreads();
    AGX_BLK_T(1);
    // ---- phase 2: Qww, Qxw, and Qxx over V (element-wise thanks to the (Phi, G) structure of the acceleration-input QP)
#pragma unroll
    for (int n = 0; n < NE; ++n) {
      const int e = tid + 256 * n, r = e / NV, c = e % NV;
      if (e < NV * NV) {
        const double Vqq = sqq[n], Vqv = sqv[n], Vvq = svq[n], Vvv = svv[n];
        const double Yq = h2 * Vqq + h * Vvq, Yv = h2 * Vqv + h * Vvv;    // (G' V) blocks, rows = acceleration index
        const double YqT = h2 * Vqq + h * Vqv, YvT = h2 * Vvq + h * Vvv;  // their transposes at [r][c]
        const double qww = pw[n] + h2 * Yq + h * Yv;
        Qww[r][c] = qww;
        Qw0[r][c] = qww;
        Qxw[r][c] = pq[n] + YqT;
        Qxw[NV + r][c] = pv[n] + h * YqT + YvT;
        const double mqv = pqv[n] + h * Vqq + Vqv;
        V[r][c] = pqq[n] + Vqq;
        V[r][NV + c] = mqv;
        V[NV + c][r] = mqv;
        V[NV + r][NV + c] = pvv[n] + h2 * Vqq + h * (Vqv + Vvq) + Vvv;
      }
    }
    if (grad && tid < NV) {
      const double vpq = vp[tid], vpv = vp[NV + tid];
      Qxw[NX][tid] = pgw + h2 * vpq + h * vpv;  // qw
      qx[tid] = pgq + vpq;
      qx[NV + tid] = pgv + h * vpq + vpv;
    }
    if (t > 0) fetch(t - 1);  // next node's tile entries are on their way during the factorisation
    __syncthreads();
    AGX_BLK_T(2);
    // ---- phase 3: Qww^-1 by blocks, in place (the pad rows / columns keep their unit diagonal)
    if (wave == inv_wave) {
      agx_v4d A11;
#pragma unroll
      for (int r = 0; r < 4; ++r) A11[r] = Qww[l4 + 4 * r][l15];
      bool bad = false;
      tile_inverse(A11, l4, l15, bad);  // inv11
      if constexpr (NWT == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Qww[l4 + 4 * r][l15] = A11[r];
      } else {
        agx_v4d A12, A22;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          A12[r] = Qww[l4 + 4 * r][16 + l15];
          A22[r] = Qww[16 + l4 + 4 * r][16 + l15];
        }
        const agx_v4d zero = {0.0, 0.0, 0.0, 0.0};
        const agx_v4d W = tile_xty(A11, A12, zero);              // inv11 A12
        const agx_v4d Wt = tile_xty(A12, A11, zero);             // A12' inv11 = W'
        agx_v4d Sc = tile_xty<true>(A12, W, A22);                // A22 - A12' W
        tile_inverse(Sc, l4, l15, bad);                          // invS
        const agx_v4d B21 = tile_xty<true>(Sc, Wt, zero);        // -invS W'
        const agx_v4d B11 = tile_xty<true>(Wt, B21, A11);        // inv11 - W B21
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Qww[l4 + 4 * r][l15] = B11[r];
          Qww[16 + l4 + 4 * r][l15] = B21[r];
          Qww[l15][16 + l4 + 4 * r] = B21[r];
          Qww[16 + l4 + 4 * r][16 + l15] = Sc[r];
        }
      }
      if (bad && lane == 0) s_bad = 1;
    }
    __syncthreads();
    AGX_BLK_T(3);
    // ---- phase 4: [Kw | kw] = Qww^-1 [Qwx | qw], 16 columns per wave and pass, on the matrix cores: the product with the inverse,
    // then one step of iterative refinement with the matrix itself (x += Qww^-1 (b - Qww x)).  The product alone is not
    // enough: its error is cond(Qww) eps and Qww = M' Luu M + ... squares the condition of the mass matrix (a serial chain of
    // 31 links: 2.5e8); the refined solution has the backward error of an elimination, which is what the line search and the
    // checker's LLT see.
    for (int tj = wave; tj < (grad ? NT : (NX + 15) / 16); tj += 4) {
      const agx_v4d zero = {0.0, 0.0, 0.0, 0.0};
      agx_v4d kk[NWT], rr[NWT];
      const int xr = 16 * tj + l15, xrc = xr <= NX ? xr : 0;
#pragma unroll
      for (int ti = 0; ti < NWT; ++ti) kk[ti] = zero;
#pragma unroll
      for (int ks = 0; ks < 4 * NWT; ++ks) {
        const int k = 4 * ks + l4;
        const double bv = (xr <= NX && k < NV) ? Qxw[xrc][k] : 0.0;  // [Qwx | qw][k][xr]
#pragma unroll
        for (int ti = 0; ti < NWT; ++ti) kk[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Qww[16 * ti + l15][k], bv, kk[ti], 0, 0, 0);
      }
      // residual in the accumulator layout (rows l4 + 4 q of each row tile); the accumulator registers of x are the B operand,
      // k-index l4 + 4 q: the A operand takes the same columns of the matrix
#pragma unroll
      for (int ti = 0; ti < NWT; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ra = 16 * ti + l4 + 4 * q;
          rr[ti][q] = (xr <= NX && ra < NV) ? Qxw[xrc][ra < NV ? ra : 0] : 0.0;
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ti = 0; ti < NWT; ++ti)
#pragma unroll
          for (int tk = 0; tk < NWT; ++tk)
            rr[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Qw0[16 * ti + l15][16 * tk + l4 + 4 * q], kk[tk][q], rr[ti], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ti = 0; ti < NWT; ++ti)
#pragma unroll
          for (int tk = 0; tk < NWT; ++tk)
            kk[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Qww[16 * ti + l15][16 * tk + l4 + 4 * q], rr[tk][q], kk[ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < NWT; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ra = 16 * ti + l4 + 4 * q;
          if (xr <= NX && ra < NV) Kl[ra][xr] = kk[ti][q];
        }
    }
    __syncthreads();
    AGX_BLK_T(4);
    // ---- phase 5: V <- Qxx - Qxw Kw: wave w owns rows 16 w .. 16 w + 15 of the 64 x 64 result
    // (tile n of wave w: the (w + 4 n)-th of the NXT x NXT tiles -- column band w of the 4 x 4 tiles, one tile each of the 2 x 2)
    agx_v4d acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int ti = (wave + 4 * n) / NXT, tj = (wave + 4 * n) % NXT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * ti + l4 + 4 * q, j = 16 * tj + l15;
        acc[n][q] = (i < NX && j < NX) ? V[i < NX ? i : 0][j < NX ? j : 0] : 0.0;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4 * NWT; ++ks) {
      const int k = 4 * ks + l4;
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const int ti = (wave + 4 * n) / NXT, tj = (wave + 4 * n) % NXT;
        const int i = 16 * ti + l15, j = 16 * tj + l15;
        const double av = (k < NV && i < NX) ? -Qxw[i < NX ? i : 0][k < NV ? k : 0] : 0.0;
        const double bv = (k < NV && j < NX) ? Kl[k < NV ? k : 0][j < NX ? j : 0] : 0.0;
        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[n], 0, 0, 0);
      }
    }
    double vxn = 0.0;
    if (grad) {  // gradient of the value function of node t, four lanes per row
      const int row = tid >> 2, part = tid & 3;
      double s = 0.0;
      if (row < NX)
        for (int k = part; k < NV; k += 4) s += Qxw[row][k] * Kl[k][NX];
      s += dpp_xor1(s);
      s += dpp_xor2(s);
      vxn = (row < NX) ? qx[row] - s : 0.0;
    }
    // gains of this node to HBM, whole rows
    for (int e = tid; e < NV * NX; e += nt) Kw[(long long)t * NV * NX + e] = Kl[e / NX][e % NX];
    if (grad && tid < NV) kw[(long long)t * NV + tid] = Kl[tid][NX];
    __syncthreads();  // every read of Qxx / Qxw / Kl is done
    AGX_BLK_T(5);
    // ---- phase 6: the value function of node t (as computed: the next node symmetrises on reading), the next node's gap
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int ti = (wave + 4 * n) / NXT, tj = (wave + 4 * n) % NXT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * ti + l4 + 4 * q, j = 16 * tj + l15;
        if (i < NX && j < NX) V[i][j] = acc[n][q];
      }
    }
    if (grad) {
      if ((tid & 3) == 0 && (tid >> 2) < NX) vx[tid >> 2] = vxn;
      if (tid < NX) fl[tid] = pf;
    }
    __syncthreads();
    AGX_BLK_T(6);
  }
#ifdef AGX_BLK_STAMP
  if (tid == 0 && b == 0)
    printf("k_riccati_blk phases (10 ns ticks over %d nodes): p1 %lld p2 %lld p3 %lld p4 %lld p5 %lld p6 %lld\n", T, ph_[1], ph_[2], ph_[3], ph_[4], ph_[5], ph_[6]);
#endif
  if (!gains_pass) {
    const int bad = s_bad;  // (written before the last barriers of the loop)
    if (tid == 0) {
      S.dir_fail = bad;
      if (bad) atomicOr(&S.flags, 1);
    }
    if (bad) return;  // no forward pass on gains of an indefinite problem
  }
  if (gains_pass || !forward) return;
  // ---- forward pass: w = -kw - Kw dx (8 lanes per row, columns strided over them), then the state update
  double *dx = dxs + (long long)b * (T + 1) * NX, *ws = wss + (long long)b * T * NV;
  if (tid < NX) { dxl[tid] = 0.0; dx[tid] = 0.0; }
  __threadfence_block();
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const double *tl = qb + (long long)t * Q::SIZE;
    const double h = dts[t], h2 = h * h;
    {
      const int r = tid >> 3, p = tid & 7;
      double s = 0.0;
      if (r < NV) {
        const double *kr = Kw + ((long long)t * NV + r) * NX;
        for (int c = p; c < NX; c += 8) s += kr[c] * dxl[c];
      }
      s += dpp_xor1(s); s += dpp_xor2(s); s += dpp_xor4(s);
      if (r < NV && p == 0) {
        const double wv = -(kw[(long long)t * NV + r] + s);
        wl[r] = wv;
        ws[(long long)t * NV + r] = wv;
      }
    }
    __syncthreads();
    double nq = 0.0, nv2 = 0.0;
    if (tid < NV) {
      nq = dxl[tid] + h * dxl[NV + tid] + h2 * wl[tid] + tl[Q::f + tid];
      nv2 = dxl[NV + tid] + h * wl[tid] + tl[Q::f + NV + tid];
    }
    __syncthreads();
    if (tid < NV) {
      dxl[tid] = nq; dxl[NV + tid] = nv2;
      dx[(long long)(t + 1) * NX + tid] = nq;
      dx[(long long)(t + 1) * NX + NV + tid] = nv2;
    }
    __syncthreads();
  }
}

}  // namespace agx
