// agx_riccati_mx2.hpp -- EXACT TWO-LEVEL Riccati sweep for small batches (nv <= 7, MFMA operand layout).
//
// One wave per instance walks 101 dependent nodes in agx_riccati_mx.hpp: at batch 1 ... 256 that chain is the time of
// an MPC step (231 of 325 us) while most of the chip idles.  Here the horizon is cut into S
// segments that are swept IN PARALLEL, and the value functions at the segment boundaries are recovered exactly:
//
//   launch 1 (k_riccati_mx2_elem, B x S waves)
//     last segment:   the ordinary sweep from the terminal node -> gains, V at its first node;
//     other segments: the ordinary sweep from a ZERO terminal value function -> J, eta (column 7), and next to it,
//                     off the critical chain, the transition under those gains and its Gramian
//                        A <- A (Phi - G Kw0),  beta <- beta + A (f - G kw0),  Cm <- Cm + (A G) Mww^-1 (A G)'
//                     (G'A' rides through the elimination as 16 more right-hand sides).  For ANY value function (P, p)
//                     at the end of the segment the one at its start is then
//                        P_a = J + A' P (I + Cm P)^-1 A,     p_a = eta + A' (I + P Cm)^-1 (p + P beta)
//                     -- one Riccati step of a macro stage (the matrix-inversion-lemma form of composing the stages).
//   launch 2 (k_riccati_mx2_sweep, B x (S - 1) waves)
//     segment s applies that step for the segments behind it (S - 2 - s steps of ~1.2 us: a 14-pivot elimination of
//     I + P Cm and five chains of four MFMAs), then runs the ORDINARY sweep over its own nodes from the recovered
//     value function: the gains come out of the same recursion as in the one-wave sweep, only the boundary value
//     functions take another (exact) route.  numpy prototype on the Panda tiles: boundary P, p and all gains to 4e-15
//     relative, cond(I + P Cm) <= 55, no pivoting needed (smallest relative pivot 0.19).
//   launch 3 (k_riccati_mx2_fwd, B x S waves): the forward pass, segmented with the transitions  x_end = A x_start + beta  that
//     the sweeps of launches 1 / 2 accumulate under the TRUE gains: a walk over the boundaries, then the segment's nodes.
// Chain per sweep at T = 100, S = 10: 10 x 1.4 + 8 x 1.2 + 10 node times instead of 101.  2.4 x the arithmetic: used
// below a batch threshold only (agimus_hip.hip: mx2_segments).
//
// The per-node arithmetic is the step of riccati_mx_body, reproduced here with the extra recursions (a copy, so that
// the code object of the one-wave sweep -- the kernel of every large-batch number -- does not change).
// Reference: mim_solvers SolverCSQP backwardPass as called from agimus_controller/ocp_base_croco.py:172.
#pragma once

namespace agx {

constexpr int kMx2MaxSeg = 16;
__host__ __device__ __forceinline__ int mx2_bound(int s, int S, int T) { return (int)((long long)s * T / S); }

// raw register image of a 16 x 16 tile (4 doubles per lane)
__device__ __forceinline__ void mx2_store(double *__restrict__ p, const mx4 &v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r * 64 + threadIdx.x] = v[r];
}
__device__ __forceinline__ mx4 mx2_load(const double *__restrict__ p) {
  mx4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = p[r * 64 + threadIdx.x];
  return v;
}

// Nodes t_hi - 1 ... t_lo of instance b.  vinit 0: V = 0 (element sweep), 1: V = terminal tile (+ regularisation),
// 2: V as given.  CLOOP: the transition A / At (and, ELEM, the Gramian Cm) under the sweep's gains is accumulated.
template <int NV, bool GAINS, bool CLOOP, bool ELEM>
__device__ __forceinline__ void riccati_mx_seg(const int b, const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                               const double *__restrict__ qts, const double *__restrict__ auxs,
                                               double *__restrict__ Kws, double *__restrict__ kws, double *__restrict__ Kout,
                                               const double dreg, const int t_lo, const int t_hi, const int vinit, mx4 &V, mx4 &Am,
                                               mx4 &At, mx4 &Cm, bool &bad_pivot) {
  static_assert(NV <= 7, "16-column tiles: 7 joints + the gradient slot per half");
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  constexpr int NX = 2 * NV, TS = Q::SIZE, LD = Q::LD;
  constexpr int Z = Q::cost + 1;  // zero of every tile: the cost line is 8 doubles, only [0] is ever written (buffers are cleared on allocation)
  const DevOcp &o = *op;
  const int T = o.T, lane = threadIdx.x;
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


  if (vinit == 1) {
    const double *tt = qb + (long long)T * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) V[r] = tt[oHxx[r]] + dg[r];
  } else if (vinit == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) V[r] = 0.0;
  }
  if constexpr (CLOOP) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { Am[r] = 2.0 * ih[r]; At[r] = 2.0 * ih[r]; Cm[r] = 0.0; }  // identity on the real indices
  }

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
    // ---- Gauss-Jordan over the acceleration rows [Mww | Mwx]  (CLOOP / ELEM: | G' A' as 16 more right-hand sides)
    double rpr[2] = {0.0, 0.0};
    double GAt[2] = {0.0, 0.0}, Wz[2] = {0.0, 0.0};
    if constexpr (CLOOP) {
#pragma unroll
      for (int s = 0; s < 2; ++s) { GAt[s] = h2 * At[s] + h * At[s + 2]; Wz[s] = GAt[s]; }
    }
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
        double rZ = 0.0;
        if constexpr (ELEM) rZ = __shfl(Wz[s], 16 * gk + j, 64);  // the extra right-hand sides G' A' ride along
        // column k of my rows, with the pivot row's own entry zeroed (it is left untouched): off the reciprocal's chain
        const double c0 = row_bcast<k>(Ww[0]) * (s == 0 ? nz[gk] : 1.0), c1 = row_bcast<k>(Ww[1]) * (s == 1 ? nz[gk] : 1.0);
        const double f0 = c0 * rp, f1 = c1 * rp;
        Ww[0] -= f0 * rW; Wx[0] -= f0 * rX;
        Ww[1] -= f1 * rW; Wx[1] -= f1 * rX;
        if constexpr (ELEM) { Wz[0] -= f0 * rZ; Wz[1] -= f1 * rZ; }
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
    if constexpr (CLOOP) {
      // transition of the segment under these gains, x_end = A x_t + beta (beta = column 7 of A), both as A (rows x_end,
      // columns x_t) and as its transpose At (the products below need the summed index first):
      //   A  <- A (Phi - G Kw) + (A f - A G kw) e7'  =  A Phi  +  At' F  -  (G'At)' Kw     (F: f in column 7)
      //   At <- (Phi - G Kw)' At                      =  Phi' At  -  Kw' (G'At)
      // and, ELEM, the Gramian  Cm <- Cm + (A G) Mww^-1 (A G)'  =  Cm + (G'At)' Z,  Z = Mww^-1 G'At out of the elimination.
      const double nK0 = flip_sign(K0), nK1 = flip_sign(K1);
      mx4 An;
#pragma unroll
      for (int r = 0; r < 4; ++r) An[r] = Am[r] + hhi * row_ror8(Am[r]);
      if (!GAINS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) An = __builtin_amdgcn_mfma_f64_16x16x4f64(At[r], z.fb[r], An, 0, 0, 0);
      }
      An = __builtin_amdgcn_mfma_f64_16x16x4f64(GAt[0], nK0, An, 0, 0, 0);
      An = __builtin_amdgcn_mfma_f64_16x16x4f64(GAt[1], nK1, An, 0, 0, 0);
      mx4 Atn = {At[0], At[1], At[2] + h * At[0], At[3] + h3 * At[1]};
      Atn = __builtin_amdgcn_mfma_f64_16x16x4f64(K0, flip_sign(GAt[0]), Atn, 0, 0, 0);
      Atn = __builtin_amdgcn_mfma_f64_16x16x4f64(K1, flip_sign(GAt[1]), Atn, 0, 0, 0);
      if constexpr (ELEM) {
        const double Z0 = Wz[0] * rpr[0], Z1 = Wz[1] * rpr[1];
        Cm = __builtin_amdgcn_mfma_f64_16x16x4f64(GAt[0], Z0, Cm, 0, 0, 0);
        Cm = __builtin_amdgcn_mfma_f64_16x16x4f64(GAt[1], Z1, Cm, 0, 0, 0);
      }
      Am = An; At = Atn;
    }
    if constexpr (ELEM) {
      // gains of the zero-terminal problem: not the solver's, nothing is stored
    } else if (!GAINS) {
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
    load_tile(z, t - kMxDepth >= t_lo ? t - kMxDepth : t_lo);
    prefetch_group_end();
  };


  int t = t_hi - 1;
  for (int r = (t_hi - t_lo) % kMxDepth; r > 0; --r, --t) {
    Tile z;
    load_tile(z, t);
    step(z, t);
  }
  if (t >= t_lo) {
    Tile tl[kMxDepth];
#pragma unroll
    for (int i = 0; i < kMxDepth; ++i) load_tile(tl[i], t - i);
    prefetch_queue_settle(tl);
    for (; t >= t_lo; t -= kMxDepth) {
#pragma unroll
      for (int i = 0; i < kMxDepth; ++i) step(tl[i], t - i);
    }
  }
}

// One step of the boundary chain: V (value function at the END of a segment: P with p in column 7) -> the value function
// at its START, from the segment's element J (eta in column 7), A (rows x_end, columns x_start, beta in column 7), Cm.
template <int NV>
__device__ __forceinline__ void mx2_boundary_step(mx4 &V, const mx4 &J, const mx4 &A, const mx4 &Cm) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  auto xreal = [](int X) { return (X & 7) < NV; };
  double ih[4], one[4], nz[4], ez[4], rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    ih[r] = (xreal(i) && j == i) ? 0.5 : 0.0;
    one[r] = (j == i) ? 1.0 : 0.0;          // identity on all 16 slots: the pad rows of N are unit rows
    rowok[r] = xreal(i) ? 1.0 : 0.0;        // rows 7 and 15 carry nothing
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { ez[q] = (g == q) ? 1.0 : 0.0; nz[q] = 1.0 - ez[q]; }
  const double cm = (j == 7) ? 1.0 : 0.5, c7 = (j == 7) ? 1.0 : 0.0, n7 = 1.0 - c7;
  // P <- (P + P') / 2 on the real block, column 7 (p) kept, rows 7 / 15 cleared
  mx4 P = {V[0] * cm, V[1] * cm, V[2] * cm, V[3] * cm};
#pragma unroll
  for (int r = 0; r < 4; ++r) P = __builtin_amdgcn_mfma_f64_16x16x4f64(V[r], ih[r], P, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) P[r] *= rowok[r];
  // column 7 <- p + P beta   (beta = column 7 of A; P symmetric, its column 7 holds p: masked out of the operand)
  {
    mx4 acc = P;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[r] * n7, A[r] * c7, acc, 0, 0, 0);
    P = acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) P[r] *= rowok[r];
  }
  // N = I + P Cm  (P without its gradient column as the transposed operand)
  mx4 N = {one[0], one[1], one[2], one[3]};
#pragma unroll
  for (int r = 0; r < 4; ++r) N = __builtin_amdgcn_mfma_f64_16x16x4f64(P[r] * n7, Cm[r], N, 0, 0, 0);
  // Gauss-Jordan on [N | P] over the 14 real indices, no pivoting (I + P Cm with P, Cm positive semi-definite)
  double rpr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rpr[r] = 1.0 - rowok[r];  // pad rows: unit pivot
  auto pivot = [&](auto Kc) {
    constexpr int k = decltype(Kc)::value;
    if constexpr ((k & 7) < NV) {
      constexpr int rk = k >> 2, gk = k & 3;
      const double piv = readlane_f64(N[rk], 16 * gk + k);
      const double rp = chain_rcp(piv);
      const double rN = __shfl(N[rk], 16 * gk + j, 64), rP = __shfl(P[rk], 16 * gk + j, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double c = row_bcast<k>(N[r]) * (r == rk ? nz[gk] : 1.0);
        const double f = c * rp;
        N[r] -= f * rN;
        P[r] -= f * rP;
      }
      rpr[rk] += rp * ez[gk];
    }
  };
  pivot(std::integral_constant<int, 0>()); pivot(std::integral_constant<int, 1>()); pivot(std::integral_constant<int, 2>());
  pivot(std::integral_constant<int, 3>()); pivot(std::integral_constant<int, 4>()); pivot(std::integral_constant<int, 5>());
  pivot(std::integral_constant<int, 6>()); pivot(std::integral_constant<int, 8>()); pivot(std::integral_constant<int, 9>());
  pivot(std::integral_constant<int, 10>()); pivot(std::integral_constant<int, 11>()); pivot(std::integral_constant<int, 12>());
  pivot(std::integral_constant<int, 13>()); pivot(std::integral_constant<int, 14>());
  mx4 Sm;  // S = N^-1 [P | p + P beta]: symmetric real block, column 7 = s_hat
#pragma unroll
  for (int r = 0; r < 4; ++r) Sm[r] = P[r] * rpr[r] * rowok[r];
  // U = S A0 (A0 = A without its beta column), plus s_hat in column 7:  V_start = J + A0' (U + s_hat e7')
  mx4 U = {Sm[0] * c7, Sm[1] * c7, Sm[2] * c7, Sm[3] * c7};
#pragma unroll
  for (int r = 0; r < 4; ++r) U = __builtin_amdgcn_mfma_f64_16x16x4f64(Sm[r] * n7, A[r] * n7, U, 0, 0, 0);
  mx4 Vn = J;
#pragma unroll
  for (int r = 0; r < 4; ++r) Vn = __builtin_amdgcn_mfma_f64_16x16x4f64(A[r] * n7, U[r], Vn, 0, 0, 0);
  V = Vn;
}

// which instances a sweep of this launch takes (as riccati_mx_body); *dreg_out: the regularisation it runs with
__device__ __forceinline__ bool mx2_takes(const DevState &S, bool gains, int gmode, double *dreg_out) {
  if (!gains && (S.done || S.admm_conv)) return false;
  if (gains && (gmode == 1 || gmode == 4) && S.done) return false;
  *dreg_out = S.dreg;
  return true;
}

// launch 1: elements of the segments 0 ... S - 2 and the ordinary sweep of the last one.  pair: odd workgroups take the sigma
// (gains) sweep of the same instance, as k_riccati_mx_pair.  elem [2][B][S][3][256], bnd [2][B][S][256] doubles.
template <int NV>
__global__ void __launch_bounds__(64, 2) k_riccati_mx2_elem(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                         const double *__restrict__ qts, const double *__restrict__ auxs,
                                                         double *__restrict__ Kws, double *__restrict__ kws, double *__restrict__ Kout,
                                                         DevState *__restrict__ st, double *__restrict__ elem, double *__restrict__ bnd,
                                                         double *__restrict__ cl, int S, int pair, int gains_only, int gmode, int iter) {
  const int T = op->T, B = op->B;
  const int unit = pair ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  const bool gains = gains_only || (pair && (blockIdx.x & 1));
  const int b = unit / S, s = unit % S;
  DevState &St = st[b];
  double dreg;
  if (!mx2_takes(St, gains, gmode, &dreg)) return;
  const int t_lo = mx2_bound(s, S, T), t_hi = mx2_bound(s + 1, S, T);
  double *el = elem + (((size_t)(gains ? 1 : 0) * B + b) * S + s) * 768;
  double *bn = bnd + (((size_t)(gains ? 1 : 0) * B + b) * S + s) * 256;
  mx4 V, Am, At, Cm;
  bool bad = false;
  if (s == S - 1) {
    if (gains && gmode != 0 && threadIdx.x == 0) St.gains_iter = (gmode == 1) ? iter : St.dir_iter;
    if (gains) riccati_mx_seg<NV, true, false, false>(b, op, dts, qts, auxs, Kws, kws, Kout, dreg, t_lo, t_hi, 1, V, Am, At, Cm, bad);
    else {  // + the transition x_T = A x_start + beta under the gains: the forward pass is segmented too
      riccati_mx_seg<NV, false, true, false>(b, op, dts, qts, auxs, Kws, kws, Kout, dreg, t_lo, t_hi, 1, V, Am, At, Cm, bad);
      mx2_store(cl + ((size_t)b * S + s) * 256, Am);
    }
    mx2_store(bn, V);
    if (!gains) {  // this wave owns the flag; the sweeps of launch 2 only raise it
      const bool any_bad = __any(bad);
      if (threadIdx.x == 0) { St.dir_fail = any_bad ? 1 : 0; if (any_bad) atomicOr(&St.flags, 1); }
    }
  } else {
    if (gains) riccati_mx_seg<NV, true, true, true>(b, op, dts, qts, auxs, nullptr, nullptr, nullptr, dreg, t_lo, t_hi, 0, V, Am, At, Cm, bad);
    else riccati_mx_seg<NV, false, true, true>(b, op, dts, qts, auxs, nullptr, nullptr, nullptr, dreg, t_lo, t_hi, 0, V, Am, At, Cm, bad);
    mx2_store(el, V);
    mx2_store(el + 256, Am);
    mx2_store(el + 512, Cm);
  }
}

// launch 2: segment s < S - 1 recovers the value function at its end through the boundary chain, then sweeps its nodes
template <int NV>
__global__ void __launch_bounds__(64, 2) k_riccati_mx2_sweep(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                          const double *__restrict__ qts, const double *__restrict__ auxs,
                                                          double *__restrict__ Kws, double *__restrict__ kws, double *__restrict__ Kout,
                                                          DevState *__restrict__ st, const double *__restrict__ elem,
                                                          const double *__restrict__ bnd, double *__restrict__ cl, int S, int pair,
                                                          int gains_only, int gmode) {
  const int T = op->T, B = op->B;
  const int unit = pair ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  const bool gains = gains_only || (pair && (blockIdx.x & 1));
  const int b = unit / (S - 1), s = unit % (S - 1);
  DevState &St = st[b];
  double dreg;
  if (!mx2_takes(St, gains, gmode, &dreg)) return;
  const size_t base = ((size_t)(gains ? 1 : 0) * B + b) * S;
  mx4 V = mx2_load(bnd + (base + (S - 1)) * 256);
  for (int s2 = S - 2; s2 > s; --s2) {
    const double *el = elem + (base + s2) * 768;
    const mx4 J = mx2_load(el), A = mx2_load(el + 256), Cm = mx2_load(el + 512);
    mx2_boundary_step<NV>(V, J, A, Cm);
  }
  mx4 Am, At, Cm2;
  bool bad = false;
  const int t_lo = mx2_bound(s, S, T), t_hi = mx2_bound(s + 1, S, T);
  if (gains) riccati_mx_seg<NV, true, false, false>(b, op, dts, qts, auxs, Kws, kws, Kout, dreg, t_lo, t_hi, 2, V, Am, At, Cm2, bad);
  else {
    riccati_mx_seg<NV, false, true, false>(b, op, dts, qts, auxs, Kws, kws, Kout, dreg, t_lo, t_hi, 2, V, Am, At, Cm2, bad);
    mx2_store(cl + ((size_t)b * S + s) * 256, Am);
    if (__any(bad) && threadIdx.x == 0) { St.dir_fail = 1; atomicOr(&St.flags, 1); }
  }
}

// launch 3: the forward pass, segments in parallel.  Segment s first walks the boundary states x_{b(s'+1)} = A_s' x_{b(s')} + beta_s'
// (s' < s; the transitions the sweeps left in `cl`), then its own nodes as riccati_forward does: lane (r, c) of the 8 x 8 grid
// holds Kw[r][c], 2 FMAs + a row sum + one transposing ds_bpermute per node.  The state at a segment's first node is the one
// of the boundary walk (the segment before it does not write it).
template <int NV>
__global__ void __launch_bounds__(64, 2) k_riccati_mx2_fwd(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                        const double *__restrict__ qts, const double *__restrict__ Kws,
                                                        const double *__restrict__ kws, double *__restrict__ dxs,
                                                        double *__restrict__ wss, const DevState *__restrict__ st,
                                                        const double *__restrict__ cl, int S) {
  constexpr int NX = 2 * NV, TS = QT<NV>::SIZE;
  typedef QT<NV> Q;
  __shared__ double s_dt[kMaxHorizon];
  const int T = op->T, b = blockIdx.x / S, s = blockIdx.x % S, lane = threadIdx.x;
  const DevState &St = st[b];
  if (St.done || St.admm_conv) return;
  stage_dts(s_dt, dts, T);
  // ---- boundary walk in the tile layout: lane (g, j) holds x_j (every group a copy), x_7 = 1 picks up beta
  const int j = lane & 15;
  double x = (j == 7) ? 1.0 : 0.0;
  for (int s2 = 0; s2 < s; ++s2) {
    const mx4 A = mx2_load(cl + ((size_t)b * S + s2) * 256);
    double y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double p = A[r] * x;  // row g + 4 r, my column
      p += dpp_xor1(p); p += dpp_xor2(p); p += dpp_xor4(p); p += row_ror8(p);  // sum over the 16 lanes of the row
      y[r] = p;
    }
    // y_i sits in group i & 3, register i >> 2: back to "lane j holds x_j"
    const int src = 16 * (j & 3);
    const double c0 = __shfl(y[0], src, 64), c1 = __shfl(y[1], src, 64), c2 = __shfl(y[2], src, 64), c3 = __shfl(y[3], src, 64);
    const int rj = j >> 2;
    x = rj == 0 ? c0 : (rj == 1 ? c1 : (rj == 2 ? c2 : c3));
    if (j == 7) x = 1.0;
    if (j == 15 || ((j & 7) >= NV && j != 7)) x = 0.0;
  }
  // ---- the segment's nodes on the 8 x 8 lane grid
  const int t_lo = mx2_bound(s, S, T), t_hi = mx2_bound(s + 1, S, T);
  const int r = lane >> 3, c = lane & 7;
  const bool in = (r < NV) && (c < NV);
  const int rr = r < NV ? r : 0, cc = c < NV ? c : 0;
  const double inm = in ? 1.0 : 0.0;
  const double *qb = qts + (long long)b * (T + 1) * TS;
  const double *Kw = Kws + (long long)b * T * NV * NX, *kw = kws + (long long)b * T * NV;
  double *dx = dxs + (long long)b * (T + 1) * NX, *ws = wss + (long long)b * T * NV;
  double dq_r = __shfl(x, rr, 64), dv_r = __shfl(x, 8 + rr, 64), dq_c = __shfl(x, cc, 64), dv_c = __shfl(x, 8 + cc, 64);
  if (c == 0 && r < NV) { dx[(long long)t_lo * NX + r] = dq_r; dx[(long long)t_lo * NX + NV + r] = dv_r; }
  auto row_sum = [](double p) { p += dpp_xor1(p); p += dpp_xor2(p); p += dpp_xor4(p); return p; };
  struct Gain { double kq, kv, kw, fq, fv; };
  constexpr int DEPTH = 4;
  auto load_gain = [&](Gain &g, int t) {
    const double *kr = Kw + ((long long)t * NV + rr) * NX;
    g.kq = kr[cc];
    g.kv = kr[NV + cc];
    g.kw = kw[(long long)t * NV + rr];
    g.fq = qb[(long long)t * TS + Q::f + rr];
    g.fv = qb[(long long)t * TS + Q::f + NV + rr];
  };
  const bool last_seg = s == S - 1;
  auto fstep = [&](Gain &g, int t) {
    const double h = s_dt[t], h2 = h * h;
    double p = (g.kq * inm) * dq_c + (g.kv * inm) * dv_c;
    const double kwv = g.kw, fqc = g.fq, fvc = g.fv;
    p = row_sum(p);
    const double wv = -(kwv + p);
    const double nq = dq_r + h * dv_r + h2 * wv + fqc;
    const double nv2 = dv_r + h * wv + fvc;
    dq_r = nq; dv_r = nv2;
    dq_c = __shfl(nq, 8 * cc, 64);
    dv_c = __shfl(nv2, 8 * cc, 64);
    if (c == 0 && r < NV) {
      ws[(long long)t * NV + r] = wv;
      if (last_seg || t + 1 < t_hi) {  // the first node of the next segment belongs to that segment's boundary walk
        dx[(long long)(t + 1) * NX + r] = nq;
        dx[(long long)(t + 1) * NX + NV + r] = nv2;
      }
    }
    prefetch_group_begin();
    load_gain(g, t + DEPTH < t_hi ? t + DEPTH : t_hi - 1);
    prefetch_group_end();
  };
  int t = t_lo;
  for (int rem = (t_hi - t_lo) % DEPTH; rem > 0; --rem, ++t) {
    Gain g1;
    load_gain(g1, t);
    fstep(g1, t);
  }
  if (t < t_hi) {
    Gain g[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) load_gain(g[i], t + i);
    prefetch_queue_settle(g);
    for (; t < t_hi; t += DEPTH) {
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) fstep(g[i], t + i);
    }
  }
}

}  // namespace agx
