// agx_riccati_mx.hpp -- K2 for NV <= 7 on the matrix cores: the Riccati backward sweep of one instance
// in the operand layout of v_mfma_f64_16x16x4_f64 (one wave per instance, as k_riccati).
//
// The sweep of agx_kernels.hpp (riccati_body) eliminates the acceleration block with 7 rank-1
// Gauss-Jordan updates over ALL nine blocks of the node matrix on an 8 x 8 lane grid; a wave issues
// one double-precision instruction every ~2.5 ns, so its ~400 instructions per node are the time
// of the kernel (scripts/microbench/lat.hip).  Here only the 8 x (8 + 16) acceleration rows are
// eliminated lane-wise; the Schur complement, the product V f and -- in the exit sweep -- the
// sigma terms and the u-space gains  K = M Kw - taux  are MFMA instructions (one issue slot each).
//
// Extended state index X (16): 0..6 q | 7 gradient slot | 8..14 v | 15 pad.  Acceleration index w (8).
// A 16-column tile lives in registers as the MFMA accumulator does: lane l = 16 g + j holds column j
// of rows g + 4 r (register r).  In that layout a tile is at the same time
//   * the B operand  B[k = row][n = col]  of k-step r, and
//   * the A operand  A[m = col][k = row]  of its TRANSPOSE,
// so  Vnew = Mxx - Mwx' Kw  takes the saved Mwx registers (A) and the eliminated ones (B) as they are.
// The gradient rides as column 7 of every tile (vx in V, qw in the acceleration rows, gx in Hxx):
//   V[:,7] += V f   is  D = V' F + V  with F = f e7'  (4 k-steps; the same k-steps symmetrise V),
// Phi' V Phi, G' V Phi and G' V G are lane-local combinations plus a rotation of the 16-lane DPP row by 8
// (q columns <-> v columns).  Row 7 of V collects qw' Kw (unused); every other pad stays exactly zero.
//
// Same mathematics as riccati_body (agx_kernels.hpp) -- mim_solvers SolverCSQP backwardPass /
// computeDirection as called from agimus_controller/ocp_base_croco.py:172; results agree to round-off.
#pragma once

namespace agx {

typedef double mx4 __attribute__((ext_vector_type(4)));

// Nodes of tile elements in flight per wave (13 doubles per lane and node).  The sweep is bound by the
// latency of its tile loads: 1024 waves x depth x 3 KB in flight over ~3 us of loaded latency is the
// bandwidth it can draw (depth 2: ~1.7 TB/s, the rate of the lane-grid kernel too).
#ifndef AGX_MX_DEPTH
#define AGX_MX_DEPTH 4
#endif
constexpr int kMxDepth = AGX_MX_DEPTH;


// lane K of my 16-lane row (row_newbcast on the double-precision ALU, K compile time)
template <int K>
__device__ __forceinline__ double row_bcast(double x) { return __builtin_amdgcn_mov_dpp(x, 0x150 + K, 0xf, 0xf, false); }
// lane (j + 8) & 15 of my row: the q columns see the v columns and vice versa
__device__ __forceinline__ double row_ror8(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x128, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0x128, 0xf, 0xf, false));
}
// Value of lane 16 G + (my column) for every lane: the 16 lanes of row group G broadcast to the four groups with gfx950's
// v_permlane16_swap / v_permlane32_swap (VALU) instead of ds_bpermute through the LDS crossbar: swap(x, x) leaves
// [x0 x0 x2 x2] / [x1 x1 x3 x3] (16-lane rows), the second swap copies the half that holds group G over the other one.
// Tried for the pivot row of the elimination (7 fetches per node, -DAGX_MX_PERMLANE) and NOT used: four swaps per double, each
// behind the two wait states the ISA wants after a VALU write of its operands, cost the chain more than the two
// ds_bpermute pairs they replace, whose latency overlaps the reciprocal (measured: sweeps + 6 %, batch-1 step + 10 %).
template <int G>
__device__ __forceinline__ int bcast_group_b32(int x) {
  const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const int y = (G & 1) ? a[1] : a[0];
  const auto b = __builtin_amdgcn_permlane32_swap(y, y, false, false);
  return (G & 2) ? b[1] : b[0];
}
template <int G>
__device__ __forceinline__ double bcast_group(double x) {
  return __hiloint2double(bcast_group_b32<G>(__double2hiint(x)), bcast_group_b32<G>(__double2loint(x)));
}
// 1 / x to full precision with a dependent chain of 1 + 3 operations instead of 1 + 4 (fast_rcp): with e = 1 - x y0,
// y0 (1 + e)(1 + e^2) = y0 (1 + e + e^2 + e^3), e^2 formed next to the first correction.  The pivots are on the
// critical path of the sweep 7 times per node.
__device__ __forceinline__ double chain_rcp(double x) {
  const double y0 = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y0, 1.0);
  const double y1 = __builtin_fma(y0, e, y0), e2 = e * e;
  return __builtin_fma(y1, e2, y1);
}
__device__ __forceinline__ double flip_sign(double x) { return __hiloint2double(__double2hiint(x) ^ (int)0x80000000, __double2loint(x)); }

template <int NV, bool GAINS>
__device__ __forceinline__ void riccati_mx_body(const int b, const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                const double *__restrict__ qts, const double *__restrict__ auxs,
                                                double *__restrict__ Kws, double *__restrict__ kws, double *__restrict__ dxs,
                                                double *__restrict__ wss, double *__restrict__ Kout, DevState *__restrict__ st,
                                                int forward, int gmode, int iter, double *__restrict__ dus = nullptr,
                                                double *__restrict__ nodestat = nullptr) {
  static_assert(NV <= 7, "16-column tiles: 7 joints + the gradient slot per half");
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  constexpr int NX = 2 * NV, TS = Q::SIZE, LD = Q::LD;
  constexpr int Z = Q::cost + 1;  // zero of every tile: the cost line is 8 doubles, only [0] is ever written (buffers are cleared on allocation)
  const DevOcp &o = *op;
  const int T = o.T, lane = threadIdx.x;
  DevState &S = st[b];
  // which instances sweep: as riccati_body
  if (!GAINS && (S.done || S.admm_conv)) return;
  if (GAINS && (gmode == 1 || gmode == 4) && S.done) return;  // gmode 4: see riccati_body
  if (GAINS && gmode == 2 && S.gains_iter == S.dir_iter) return;
  const double dreg = (GAINS && gmode != 1 && gmode != 4) ? (S.solved ? S.dreg : S.gains_dreg) : S.dreg;
  if (GAINS && gmode != 0 && lane == 0) S.gains_iter = (gmode == 1) ? iter : S.dir_iter;
  const double sig = GAINS ? kSigma : 0.0;
  const double *qb = qts + (long long)b * (T + 1) * TS;
  const double *ab = auxs + (long long)b * (T + 1) * A::SIZE;

  const int j = lane & 15, g = lane >> 4;
  auto xreal = [](int X) { return (X & 7) < NV; };
  auto xoff = [](int X) { return X < 8 ? X : NV + (X - 8); };
  // ---- per-lane element offsets inside a node's tile
  int oHxx[4], oHwx[2], oHww[2], oF[4], oTx[2], oMt[2];
  double mTx[2], mMt[2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    int off = Z;
    if (xreal(i)) {
      if (j == 7) off = GAINS ? Z : Q::gx + xoff(i);
      else if (xreal(j)) {
        const int ri = i & 7, cj = j & 7;
        off = (i < 8) ? ((j < 8) ? Q::Hqq + ri * LD + cj : Q::Hqv + ri * LD + cj) : ((j < 8) ? Q::Hqv + cj * LD + ri : Q::Hvv + ri * LD + cj);
      }
    }
    oHxx[r] = off;
    oF[r] = (!GAINS && j == 7 && xreal(i)) ? Q::f + xoff(i) : Z;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int w = g + 4 * s;
    int ox = Z, ow = Z, otx = -1, omt = -1;  // aux operands: -1 = structural zero (the aux tile has no zero element: loaded value x 0)
    if (w < NV) {
      if (j == 7) ox = GAINS ? Z : Q::gw + w;
      else if (xreal(j)) ox = (j < 8) ? Q::Hqw + j * LD + w : Q::Hvw + (j - 8) * LD + w;
      if (j < NV) ow = Q::Hww + w * LD + j;
      if (xreal(j) && j != 7) otx = (j < 8) ? A::tq + w * A::LD + j : A::tv + w * A::LD + (j - 8);
      if (j < NV) omt = A::M + w * A::LD + j;
    }
    oHwx[s] = ox; oHww[s] = ow;
    mTx[s] = otx >= 0 ? 1.0 : 0.0; mMt[s] = omt >= 0 ? 1.0 : 0.0;
    oTx[s] = otx >= 0 ? otx : 0; oMt[s] = omt >= 0 ? omt : 0;
  }
  // ---- per-lane constants
  const double mhi = (j >= 8 && j < 8 + NV) ? 1.0 : 0.0;  // v columns take h x (q column of the same row)
  const double mlo = (j < NV) ? 1.0 : 0.0;
  const double m15 = (g == 3) ? 0.0 : 1.0;                 // register 3 of group 3 is row 15: stays zero
  double dg[4];                                            // (dreg + sigma) on the real diagonal
#pragma unroll
  for (int r = 0; r < 4; ++r) dg[r] = (xreal(g + 4 * r) && j == g + 4 * r) ? dreg + sig : 0.0;
  const double cm = (j == 7) ? 1.0 : 0.5;  // the gradient column is not mirrored (row 7 is not a copy of it)
  double ih[4];                          // I / 2 on the real diagonal, as the B operand of k-step r
#pragma unroll
  for (int r = 0; r < 4; ++r) ih[r] = (xreal(g + 4 * r) && j == g + 4 * r) ? 0.5 : 0.0;
  double nz[4], ez[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { ez[q] = (g == q) ? 1.0 : 0.0; nz[q] = 1.0 - ez[q]; }
  // gains of the direction sweep -> Kw [NV][NX] | kw [NV] (the forward pass and the step kernels read those)
  long long stK[2];
  bool stV[2];
  double *stP[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int w = g + 4 * s;
    stV[s] = (w < NV) && (j == 7 || xreal(j));
    const int wc = w < NV ? w : 0;
    if (!GAINS) {
      stP[s] = (j == 7) ? kws + (long long)b * T * NV + wc : Kws + (long long)b * T * NV * NX + wc * NX + (xreal(j) ? xoff(j) : 0);
      stK[s] = (j == 7) ? NV : NV * NX;
    } else {
      stV[s] = (w < NV) && xreal(j) && j != 7;
      stP[s] = Kout + (long long)b * T * NV * NX + wc * NX + (xreal(j) && j != 7 ? xoff(j) : 0);
      stK[s] = NV * NX;
    }
  }

  struct Tile { double hxx[4], hwx[2], hww[2], fb[4], tx[2], mt[2]; };
  __shared__ double s_dt[kMaxHorizon];
  stage_dts(s_dt, dts, T);
  auto load_tile = [&](Tile &z, int t) {
    const double *tl = qb + (long long)t * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) z.hxx[r] = tl[oHxx[r]];
#pragma unroll
    for (int s = 0; s < 2; ++s) { z.hwx[s] = tl[oHwx[s]]; z.hww[s] = tl[oHww[s]]; }
    if (!GAINS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) z.fb[r] = tl[oF[r]];
    } else {
      const double *al = ab + (long long)t * A::SIZE;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        z.tx[s] = al[oTx[s]];
        z.mt[s] = al[oMt[s]];
      }
    }
  };

  // value function of node t+1: the terminal tile (+ regularisation)
  mx4 V;
  {
    const double *tt = qb + (long long)T * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) V[r] = tt[oHxx[r]] + dg[r];
  }
  bool bad_pivot = false;

  auto step = [&](Tile &z, int t) {
    const double h = s_dt[t], h2 = h * h;
    double Hxx[4] = {z.hxx[0], z.hxx[1], z.hxx[2], z.hxx[3]};
    double Hwx[2] = {z.hwx[0], z.hwx[1]}, Hww[2] = {z.hww[0], z.hww[1]};
    double Tx[2] = {0.0, 0.0}, Mt[2] = {0.0, 0.0};
    if (GAINS) {
      // sigma [taux M]' [taux M] on the matrix cores (independent of the recursion: issued ahead of it)
      Tx[0] = z.tx[0] * mTx[0]; Tx[1] = z.tx[1] * mTx[1]; Mt[0] = z.mt[0] * mMt[0]; Mt[1] = z.mt[1] * mMt[1];
      const double sT0 = sig * Tx[0], sT1 = sig * Tx[1], sM0 = sig * Mt[0], sM1 = sig * Mt[1];
      mx4 hx = {Hxx[0], Hxx[1], Hxx[2], Hxx[3]};
      hx = __builtin_amdgcn_mfma_f64_16x16x4f64(sT0, Tx[0], hx, 0, 0, 0);
      hx = __builtin_amdgcn_mfma_f64_16x16x4f64(sT1, Tx[1], hx, 0, 0, 0);
      mx4 hw = {Hwx[0], Hwx[1], 0.0, 0.0};
      hw = __builtin_amdgcn_mfma_f64_16x16x4f64(sM0, Tx[0], hw, 0, 0, 0);
      hw = __builtin_amdgcn_mfma_f64_16x16x4f64(sM1, Tx[1], hw, 0, 0, 0);
      mx4 hu = {Hww[0], Hww[1], 0.0, 0.0};
      hu = __builtin_amdgcn_mfma_f64_16x16x4f64(sM0, Mt[0], hu, 0, 0, 0);
      hu = __builtin_amdgcn_mfma_f64_16x16x4f64(sM1, Mt[1], hu, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Hxx[r] = hx[r];
      Hwx[0] = hw[0]; Hwx[1] = hw[1]; Hww[0] = hu[0]; Hww[1] = hu[1];
    }
    // V <- (V + V') / 2, and in the direction sweep V[:,7] += V f  (vp = vx + V f):  D = V' (I/2 + f e7') + V/2.
    // The Schur complement below takes Mwx' for Mxw, which is exact only for a symmetric V: round-off
    // asymmetry a would propagate as (Phi - G K)' a (Phi + G K) and grow ~1.5 x per node; the k-steps
    // that form V f transpose V on the way, so the symmetric part costs the direction sweep nothing.
    {
      mx4 acc = {V[0] * cm, V[1] * cm, V[2] * cm, V[3] * cm};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V[0], GAINS ? ih[0] : z.fb[0] + ih[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V[1], GAINS ? ih[1] : z.fb[1] + ih[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V[2], GAINS ? ih[2] : z.fb[2] + ih[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V[3], GAINS ? ih[3] : z.fb[3] + ih[3], acc, 0, 0, 0);
      V = acc;
    }
    const double hhi = h * mhi, hlo = h * mlo, h3 = h * m15;
    // Y = G' V (acceleration rows): h^2 V[q rows] + h V[v rows]
    double Ww[2], Wx[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const double Y = h2 * V[s] + h * V[s + 2];
      const double Yr = row_ror8(Y);
      Wx[s] = (Hwx[s] + Y) + hhi * Yr;       // [Yq | h Yq + Yv]
      Ww[s] = (Hww[s] + h2 * Y) + hlo * Yr;  // h^2 Yq + h Yv
    }
    const double Ax0 = Wx[0], Ax1 = Wx[1];  // Mwx before the elimination: the A operand of the Schur complement
    // Mxx = Hxx + Phi' V Phi
    mx4 C;
    {
      double W1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) W1[r] = V[r] + hhi * row_ror8(V[r]);
      C[0] = (Hxx[0] + dg[0]) + W1[0];
      C[1] = (Hxx[1] + dg[1]) + W1[1];
      C[2] = ((Hxx[2] + dg[2]) + W1[2]) + h * W1[0];
      C[3] = ((Hxx[3] + dg[3]) + W1[3]) + h3 * W1[1];
    }
    // ---- Gauss-Jordan over the acceleration rows [Mww | Mwx]
    double rpr[2] = {0.0, 0.0};
    auto pivot = [&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      if constexpr (k < NV) {
        constexpr int s = k >> 2, gk = k & 3;
        const double piv = readlane_f64(Ww[s], 16 * gk + k);
#ifdef AGX_MX_OLD_CHAIN
        const double rp = fast_rcp(piv);
#else
        const double rp = chain_rcp(piv);
#endif
#ifdef AGX_MX_PERMLANE  // measured slower (round 3): direction sweep 0.209 -> 0.222 ms, batch-1 step 0.325 -> 0.358 ms
        const double rW = bcast_group<gk>(Ww[s]), rX = bcast_group<gk>(Wx[s]);
#else
        const double rW = __shfl(Ww[s], 16 * gk + j, 64), rX = __shfl(Wx[s], 16 * gk + j, 64);  // pivot row: lane (gk, j) to every group
#endif
        // column k of my rows, with the pivot row's own entry zeroed (it is left untouched): off the reciprocal's chain
        const double c0 = row_bcast<k>(Ww[0]) * (s == 0 ? nz[gk] : 1.0), c1 = row_bcast<k>(Ww[1]) * (s == 1 ? nz[gk] : 1.0);
        const double f0 = c0 * rp, f1 = c1 * rp;
        Ww[0] -= f0 * rW; Wx[0] -= f0 * rX;
        Ww[1] -= f1 * rW; Wx[1] -= f1 * rX;
        rpr[s] += rp * ez[gk];
      }
    };
    pivot(std::integral_constant<int, 0>()); pivot(std::integral_constant<int, 1>()); pivot(std::integral_constant<int, 2>());
    pivot(std::integral_constant<int, 3>()); pivot(std::integral_constant<int, 4>()); pivot(std::integral_constant<int, 5>());
    pivot(std::integral_constant<int, 6>());
    bad_pivot = bad_pivot || (g < NV && !(rpr[0] > 0.0)) || (g + 4 < NV && !(rpr[1] > 0.0));
    // Kw = D^-1 [Mwq | kw | Mwv]  (rows beyond NV: rpr = 0)
    const double K0 = Wx[0] * rpr[0], K1 = Wx[1] * rpr[1];
    // V of node t = Mxx - Mwx' Kw
    V = __builtin_amdgcn_mfma_f64_16x16x4f64(Ax0, flip_sign(K0), C, 0, 0, 0);
    V = __builtin_amdgcn_mfma_f64_16x16x4f64(Ax1, flip_sign(K1), V, 0, 0, 0);
    if (!GAINS) {
      if (stV[0]) stP[0][(long long)t * stK[0]] = K0;
      if (stV[1]) stP[1][(long long)t * stK[1]] = K1;
    } else {
      // u-space gains  K = M Kw - taux  (column 7 of Kw is zero in this sweep)
      mx4 ku = {flip_sign(Tx[0]), flip_sign(Tx[1]), 0.0, 0.0};
      ku = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[0], K0, ku, 0, 0, 0);
      ku = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[1], K1, ku, 0, 0, 0);
      if (stV[0]) stP[0][(long long)t * stK[0]] = ku[0];
      if (stV[1]) stP[1][(long long)t * stK[1]] = ku[1];
    }
    // Refill this register set, (a) after the last use of its old contents and (b) unconditionally (node 0 again at
    // the end).  Either a live old value or a branch makes the loaded values reach the next pass through register
    // copies at the loop latch, and the s_waitcnt vmcnt(0) in front of those copies drains the prefetch queue.
    prefetch_group_begin();
    load_tile(z, t >= kMxDepth ? t - kMxDepth : 0);
    prefetch_group_end();
  };

  int t = T - 1;
  // the nodes that do not fill a group of kMxDepth first, one at a time; the pipelined loop then runs whole groups
  for (int r = T % kMxDepth; r > 0; --r, --t) {
    Tile z;
    load_tile(z, t);
    step(z, t);
  }
  if (t >= 0) {
    Tile tl[kMxDepth];
#pragma unroll
    for (int i = 0; i < kMxDepth; ++i) load_tile(tl[i], t - i);
    prefetch_queue_settle(tl);
    for (; t >= 0; t -= kMxDepth) {
#pragma unroll
      for (int i = 0; i < kMxDepth; ++i) step(tl[i], t - i);
    }
  }
  if (!GAINS) {
    // gmode 3: the LQR pass next to the ADMM factorisation of the same instance -- that sweep owns dir_fail (as it
    // does when it runs afterwards); the sticky "a direction was discarded" bit is raised by both
    const bool any_bad = __any(bad_pivot);
    if (lane == 0) {
      if (gmode != 3) S.dir_fail = any_bad ? 1 : 0;
      if (any_bad) atomicOr(&S.flags, 1);
    }
  }
  if (GAINS || !forward) return;
  if (forward == 2) {  // K3 rides along (see riccati_forward)
    FwdKkt kk;
    kk.ab = ab; kk.preg = S.preg; kk.dreg = dreg; kk.du = dus + (long long)b * T * NV; kk.ns = nodestat + (long long)b * (T + 1) * 4;
    riccati_forward<NV, true>(b, T, dts, qb, Kws + (long long)b * T * NV * NX, kws + (long long)b * T * NV, dxs, wss, s_dt, kk);
  } else {
    riccati_forward<NV>(b, T, dts, qb, Kws + (long long)b * T * NV * NX, kws + (long long)b * T * NV, dxs, wss, s_dt);
  }
}

template <int NV, bool GAINS>
__global__ void __launch_bounds__(64, 2) k_riccati_mx(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                   const double *__restrict__ qts, const double *__restrict__ auxs,
                                                   double *__restrict__ Kws, double *__restrict__ kws, double *__restrict__ dxs,
                                                   double *__restrict__ wss, double *__restrict__ Kout, DevState *__restrict__ st,
                                                   int forward, int gmode, double *__restrict__ dus, double *__restrict__ nodestat) {
  riccati_mx_body<NV, GAINS>(blockIdx.x, op, dts, qts, auxs, Kws, kws, dxs, wss, Kout, st, forward, gmode, 0, dus, nodestat);
}

// direction sweep (even workgroups) and speculative exit sweep (odd) of one SQP iteration in one launch, as k_riccati_pair
template <int NV>
__global__ void __launch_bounds__(64, 2) k_riccati_mx_pair(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                        const double *__restrict__ qts, const double *__restrict__ auxs,
                                                        double *__restrict__ Kws, double *__restrict__ kws, double *__restrict__ dxs,
                                                        double *__restrict__ wss, double *__restrict__ Kout, DevState *__restrict__ st,
                                                        int iter, int forward, double *__restrict__ dus, double *__restrict__ nodestat) {
  // Workgroups go to the 8 XCDs round robin (blockIdx.x % 8): the two sweeps of an instance take block ids 8 apart, so that both
  // stream the instance's tiles through the same L2 (grid: 16 workgroups per 8 instances, rounded up)
  const int grp = blockIdx.x >> 4, w16 = blockIdx.x & 15;
  const int b = grp * 8 + (w16 & 7);
  if (b >= op->B) return;
  if (w16 >> 3)
    riccati_mx_body<NV, true>(b, op, dts, qts, auxs, Kws, kws, dxs, wss, Kout, st, 0, 1, iter);
  else
    riccati_mx_body<NV, false>(b, op, dts, qts, auxs, Kws, kws, dxs, wss, Kout, st, forward, 0, iter, dus, nodestat);
}

}  // namespace agx
