// agx_k1_lanes.hpp -- K1 for serial chains with NV <= 8: EIGHT LANES PER NODE, one lane per joint.
//
// The one-lane-per-node kernel (k_calc_qp) keeps ~600 doubles of per-joint state live and spills
// to scratch at one wave per SIMD.  Here a node's joints sit in 8 adjacent lanes of a wave:
//   * recursions along the chain become 3-step scans over the 8-lane group (SE3 prefix product,
//     prefix sums of velocity / acceleration, suffix sums of composite inertia / force / Coriolis);
//   * everything that is "all joints to all joints" (CRBA, RNEA-derivative matrices, J'WJ, the
//     M / taux products of the QP transformation) goes through a per-node LDS tile: every lane
//     publishes its joint's vectors and computes ONE COLUMN of each matrix;
//   * per-lane state is ~100 doubles, no scratch, several waves per SIMD.
// Same mathematics as agx_device.hpp (see the derivation there); results agree to round-off.
#pragma once

#include "agx_device.hpp"

namespace agx {

__device__ __forceinline__ double g_up(double x, int off) { return __shfl_up(x, off, 8); }
__device__ __forceinline__ double g_dn(double x, int off) { return __shfl_down(x, off, 8); }
__device__ __forceinline__ double g_bc(double x, int src) { return __shfl(x, src, 8); }

// inclusive prefix / suffix sums of N doubles over the 8-lane group
template <int N>
__device__ __forceinline__ void g_prefix_sum(double *x, int l8) {
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    double y[N];
#pragma unroll
    for (int e = 0; e < N; ++e) y[e] = g_up(x[e], off);
    if (l8 >= off) {
#pragma unroll
      for (int e = 0; e < N; ++e) x[e] += y[e];
    }
  }
}
template <int N>
__device__ __forceinline__ void g_suffix_sum(double *x, int l8) {
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    double y[N];
#pragma unroll
    for (int e = 0; e < N; ++e) y[e] = g_dn(x[e], off);
    if (l8 + off < 8) {
#pragma unroll
      for (int e = 0; e < N; ++e) x[e] += y[e];
    }
  }
}
__device__ __forceinline__ double g_sum(double x) {
  x += __shfl_xor(x, 1, 8);
  x += __shfl_xor(x, 2, 8);
  x += __shfl_xor(x, 4, 8);
  return x;
}

// per-node LDS tile (doubles).  Phase 1 (dynamics): S, m6, Sd, psi, Dt.  Phase 2 (after the
// derivative matrices exist) reuses the same storage for tq, tv and the frame Jacobian J.
struct LjNode {
  union {
    struct { double S[8][6], m6[8][6], Sd[8][6], psi[8][6]; } p1;
    struct { double tq[8][8], tv[8][8], J[8][6]; } p2;
  } u;
  double Dt[8][4];
  double M[8][8];
  double vec[3][8];  // rhs / lu / D
  double pad[2];     // node stride 314 doubles: spreads the 8 nodes of a wave over the banks
};

#ifndef AGX_K1_WAVES
#define AGX_K1_WAVES 2
#endif
template <int NV, bool TERM>
__global__ void __launch_bounds__(128, AGX_K1_WAVES) k_calc_qp_lj(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                    const double *__restrict__ dts, const double *__restrict__ xs,
                                                    const double *__restrict__ us, RefView rv, double *__restrict__ qts,
                                                    double *__restrict__ auxs, const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  __shared__ LjNode lds[16];
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  const int l8 = threadIdx.x & 7;
  LjNode &L = lds[threadIdx.x >> 3];
  const long long n_nodes = TERM ? (long long)o.B : (long long)o.B * T;
  const long long node = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const bool node_ok = node < n_nodes;
  const long long nid = node_ok ? node : 0;  // out-of-range groups shadow node 0 and store nothing
  const int b = TERM ? (int)nid : (int)(nid / T), t = TERM ? T : (int)(nid % T);
  const bool act = node_ok && !st[b].done;
  const bool jl = l8 < NV;       // lane carries a joint
  const int j = jl ? l8 : NV - 1;
  const double preg = st[b].preg;
  const double dt = TERM ? 0.0 : dts[t];
  const double *xp = xs + ((long long)b * (T + 1) + t) * NX;
  const double qj = xp[j], vj = jl ? xp[NV + j] : 0.0;
  const double uj = TERM ? 0.0 : us[((long long)b * T + t) * NV + j];
  double *qt = qts + ((long long)b * (T + 1) + t) * Q::SIZE;
  double *ax = auxs + ((long long)b * (T + 1) + t) * A::SIZE;
  const DevRows &rows = o.rows[TERM ? 1 : 0];
  const bool wr = act && jl;  // this lane stores

  // ---- kinematics: local placement, then SE3 prefix product along the chain
  double R[9], p[3];
  {
    const double *ax3 = m.axis[j];
    double s, c;
    sincos(qj, &s, &c);
    const double omc = 1.0 - c;
    double Rq[9];
    Rq[0] = c + omc * ax3[0] * ax3[0];
    Rq[1] = omc * ax3[0] * ax3[1] - s * ax3[2];
    Rq[2] = omc * ax3[0] * ax3[2] + s * ax3[1];
    Rq[3] = omc * ax3[1] * ax3[0] + s * ax3[2];
    Rq[4] = c + omc * ax3[1] * ax3[1];
    Rq[5] = omc * ax3[1] * ax3[2] - s * ax3[0];
    Rq[6] = omc * ax3[2] * ax3[0] - s * ax3[1];
    Rq[7] = omc * ax3[2] * ax3[1] + s * ax3[0];
    Rq[8] = c + omc * ax3[2] * ax3[2];
    mm3(m.placement[j], Rq, R);
    p[0] = m.placement[j][9]; p[1] = m.placement[j][10]; p[2] = m.placement[j][11];
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    double Rp[9], pp[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rp[e] = g_up(R[e], off);
#pragma unroll
    for (int e = 0; e < 3; ++e) pp[e] = g_up(p[e], off);
    if (l8 >= off) {
      double tt[3];
      mv3(Rp, p, tt);
      p[0] = pp[0] + tt[0]; p[1] = pp[1] + tt[1]; p[2] = pp[2] + tt[2];
      mm3(Rp, R, R);
    }
  }
  double S[6];
  {
    double z[3];
    mv3(R, m.axis[j], z);
    cross3(p, z, S);
    S[3] = z[0]; S[4] = z[1]; S[5] = z[2];
    if (!jl) {
#pragma unroll
      for (int e = 0; e < 6; ++e) S[e] = 0.0;
    }
  }
  // body inertia in the world frame
  double Ib[10];
  {
    double cw[3];
    mv3(R, m.com[j], cw);
    cw[0] += p[0]; cw[1] += p[1]; cw[2] += p[2];
    const double ms = jl ? m.mass[j] : 0.0;
    double Tm[9], Iw[9];
    mm3(R, m.inertia[j], Tm);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) Iw[3 * a + c] = Tm[3 * a] * R[3 * c] + Tm[3 * a + 1] * R[3 * c + 1] + Tm[3 * a + 2] * R[3 * c + 2];
    const double cc = dot3(cw, cw), on = jl ? 1.0 : 0.0;
    Ib[0] = ms;
    Ib[1] = ms * cw[0]; Ib[2] = ms * cw[1]; Ib[3] = ms * cw[2];
    Ib[4] = on * Iw[0] + ms * (cc - cw[0] * cw[0]);
    Ib[5] = on * 0.5 * (Iw[1] + Iw[3]) - ms * cw[0] * cw[1];
    Ib[6] = on * 0.5 * (Iw[2] + Iw[6]) - ms * cw[0] * cw[2];
    Ib[7] = on * Iw[4] + ms * (cc - cw[1] * cw[1]);
    Ib[8] = on * 0.5 * (Iw[5] + Iw[7]) - ms * cw[1] * cw[2];
    Ib[9] = on * Iw[8] + ms * (cc - cw[2] * cw[2]);
  }
  double Ic[10];
#pragma unroll
  for (int e = 0; e < 10; ++e) Ic[e] = Ib[e];
  g_suffix_sum<10>(Ic, l8);
  double m6[6];
  iapply(Ic, S, m6);

  // ---- TERM nodes carry costs only; running nodes: dynamics
  double v6[6], Sd[6], h6[6], tqc[NV], tvc[NV], Mc[NV], qdd = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) { tqc[i] = 0.0; tvc[i] = 0.0; Mc[i] = 0.0; }
  if (!TERM) {
#pragma unroll
    for (int e = 0; e < 6; ++e) v6[e] = S[e] * vj;
    g_prefix_sum<6>(v6, l8);
    mcross(v6, S, Sd);
    double a0[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) a0[e] = Sd[e] * vj;
    g_prefix_sum<6>(a0, l8);
    a0[0] -= m.gravity[0]; a0[1] -= m.gravity[1]; a0[2] -= m.gravity[2];
    iapply(Ib, v6, h6);
    double fb[6], g6[6], x6[6];
    iapply(Ib, a0, g6);
    fcross(v6, h6, x6);
#pragma unroll
    for (int e = 0; e < 6; ++e) fb[e] = g6[e] + x6[e];
    g_suffix_sum<6>(fb, l8);
    const double nle = dot6(S, fb);
    // publish S, m6 ; column j of M
#pragma unroll
    for (int e = 0; e < 6; ++e) { L.u.p1.S[l8][e] = S[e]; L.u.p1.m6[l8][e] = m6[e]; }
    L.vec[0][l8] = uj - nle;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double val;
      if (i >= l8) val = dot6(S, L.u.p1.m6[i]);
      else val = dot6(L.u.p1.S[i], m6);
      if (i == l8) val += m.armature[j];
      Mc[i] = val;
      L.M[i][l8] = val;
      if (wr) ax[A::M + i * NV + l8] = val;
    }
    __syncthreads();
    // every lane factorises M (LDL', reciprocal pivots) and solves for qdd
    {
      double Lf[NV][NV], dk[NV], dinv[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int c = 0; c <= i; ++c) Lf[i][c] = L.M[i][c];
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        double dd = Lf[c][c];
#pragma unroll
        for (int k = 0; k < c; ++k) dd -= Lf[c][k] * Lf[c][k] * dk[k];
        dk[c] = dd;
        dinv[c] = 1.0 / dd;
#pragma unroll
        for (int i = c + 1; i < NV; ++i) {
          double sacc = Lf[i][c];
#pragma unroll
          for (int k = 0; k < c; ++k) sacc -= Lf[i][k] * Lf[c][k] * dk[k];
          Lf[i][c] = sacc * dinv[c];
        }
      }
      double y[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double sacc = L.vec[0][i];
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= Lf[i][k] * y[k];
        y[i] = sacc;
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) y[i] *= dinv[i];
#pragma unroll
      for (int i = NV - 1; i >= 0; --i) {
        double sacc = y[i];
#pragma unroll
        for (int k = i + 1; k < NV; ++k) sacc -= Lf[k][i] * y[k];
        y[i] = sacc;
      }
      qdd = 0.0;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i == l8) qdd = y[i];
    }
    // gap f = xnext - xs[t+1]
    if (wr) {
      const double *xn = xp + NX;
      qt[Q::f + l8] = qj + dt * vj + dt * dt * qdd - xn[l8];
      qt[Q::f + NV + l8] = vj + dt * qdd - xn[NV + l8];
    }
    // ---- pass B: accelerations with qdd, psi, composite force / momentum / E
    double a6[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) a6[e] = S[e] * qdd + Sd[e] * vj;
    g_prefix_sum<6>(a6, l8);
    a6[0] -= m.gravity[0]; a6[1] -= m.gravity[1]; a6[2] -= m.gravity[2];
    double psi[6];
    {
      double t1[6], t2[6];
      mcross(a6, S, t1);
      mcross(v6, Sd, t2);
#pragma unroll
      for (int e = 0; e < 6; ++e) psi[e] = t1[e] + t2[e];
    }
    double cmp[18];  // fC (6) | f0C (3) | EC (9)
    {
      double gg[6], xx[6];
      iapply(Ib, a6, gg);
      fcross(v6, h6, xx);
#pragma unroll
      for (int e = 0; e < 6; ++e) cmp[e] = gg[e] + xx[e];
      cmp[6] = h6[0]; cmp[7] = h6[1]; cmp[8] = h6[2];
      const double *vl = v6, *w = v6 + 3, *hh = Ib + 1;
      const double Io[9] = {Ib[4], Ib[5], Ib[6], Ib[5], Ib[7], Ib[8], Ib[6], Ib[8], Ib[9]};
      double WI[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        WI[0 + c] = w[1] * Io[6 + c] - w[2] * Io[3 + c];
        WI[3 + c] = w[2] * Io[0 + c] - w[0] * Io[6 + c];
        WI[6 + c] = w[0] * Io[3 + c] - w[1] * Io[0 + c];
      }
      const double vh = dot3(vl, hh);
      double *E = cmp + 9;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) E[3 * r + c] = WI[3 * r + c] + WI[3 * c + r] - vl[r] * hh[c] - hh[r] * vl[c] + (r == c ? 2.0 * vh : 0.0);
      const double *n0 = h6 + 3;
      E[1] += n0[2]; E[2] -= n0[1];
      E[3] -= n0[2]; E[5] += n0[0];
      E[6] += n0[1]; E[7] -= n0[0];
    }
    g_suffix_sum<18>(cmp, l8);
    const double *fC = cmp, *f0C = cmp + 6, *EC = cmp + 9;
    double Dt[3], colv[6], colq[6];
    {
      double tt[3];
      cross3(f0C, S, tt);
      const double *sa = S + 3;
      Dt[0] = 2.0 * tt[0] + EC[0] * sa[0] + EC[3] * sa[1] + EC[6] * sa[2];
      Dt[1] = 2.0 * tt[1] + EC[1] * sa[0] + EC[4] * sa[1] + EC[7] * sa[2];
      Dt[2] = 2.0 * tt[2] + EC[2] * sa[0] + EC[5] * sa[1] + EC[8] * sa[2];
      double IcSd[6], IcPs[6], sxf[6], u1[3], u2[3], e1[3], e2[3];
      iapply(Ic, Sd, IcSd);
      iapply(Ic, psi, IcPs);
      fcross(S, fC, sxf);
      cross3(f0C, S + 3, u1);
      cross3(f0C, Sd + 3, u2);
      mv3(EC, S + 3, e1);
      mv3(EC, Sd + 3, e2);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        colv[e] = 2.0 * IcSd[e] - 2.0 * u1[e];
        colv[3 + e] = 2.0 * IcSd[3 + e] + e1[e];
        colq[e] = sxf[e] - 2.0 * u2[e] + IcPs[e];
        colq[3 + e] = sxf[3 + e] + e2[e] + IcPs[3 + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) { L.u.p1.Sd[l8][e] = Sd[e]; L.u.p1.psi[l8][e] = psi[e]; }
    L.Dt[l8][0] = Dt[0]; L.Dt[l8][1] = Dt[1]; L.Dt[l8][2] = Dt[2];
    __syncthreads();
    // column l8 of dtau/dq (tqc) and dtau/dqdot (tvc)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double dvv, dqq;
      if (i >= l8) {
        const double *mi = L.u.p1.m6[i], *Di = L.Dt[i];
        dvv = 2.0 * dot6(mi, Sd) + dot3(Di, S + 3);
        dqq = dot3(Di, Sd + 3) + dot6(mi, psi);
      } else {
        const double *Si = L.u.p1.S[i];
        dvv = dot6(Si, colv);
        dqq = dot6(Si, colq);
      }
      tvc[i] = jl ? dvv : 0.0;
      tqc[i] = jl ? dqq : 0.0;
    }
    __syncthreads();  // phase 1 storage is dead from here on
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      L.u.p2.tq[i][l8] = tqc[i];
      L.u.p2.tv[i][l8] = tvc[i];
      if (wr) { ax[A::tq + i * NV + l8] = tqc[i]; ax[A::tv + i * NV + l8] = tvc[i]; }
    }
  } else {
    __syncthreads();
    if (wr) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { ax[A::M + i * NV + l8] = 0.0; ax[A::tq + i * NV + l8] = 0.0; ax[A::tv + i * NV + l8] = 0.0; }
      qt[Q::f + l8] = 0.0;
      qt[Q::f + NV + l8] = 0.0;
    }
  }

  // ---- cost rows: lane j owns component j of state / control terms and column j of J'WJ
  double cost = 0.0, Lq = 0.0, Lv = 0.0, Lu = 0.0, Lvv = 0.0, Luu = 0.0, Lqqc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) Lqqc[i] = 0.0;
  const double *ref = ref_at(rv, b, t, T);
  const int *frames = frames_at(rv, b, t, T);
  for (int r = 0; r < rows.n; ++r) {
    if (!rows.active[r]) continue;
    const double *tile = ref + rows.off[r];
    const double wi = tile[0];
    const double *rr = tile + 1;
    const double *aw = rr + rows.nref[r];
    const int kind = rows.kind[r];
    if (kind == AGX_RES_STATE) {
      const double rq = qj - rr[j], rvv = vj - rr[NV + j];
      const double wq = jl ? wi * aw[j] : 0.0, wv = jl ? wi * aw[NV + j] : 0.0;
      cost += 0.5 * (wq * rq * rq + wv * rvv * rvv);
      Lq += wq * rq;
      Lv += wv * rvv;
      Lvv += wv;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i == l8) Lqqc[i] += wq;
    } else if (kind == AGX_RES_CONTROL) {
      if (!TERM) {
        const double ru = uj - rr[j];
        const double wu = jl ? wi * aw[j] : 0.0;
        cost += 0.5 * wu * ru * ru;
        Lu += wu * ru;
        Luu += wu;
      }
    } else if (kind == AGX_RES_FRAME_PLACEMENT || kind == AGX_RES_FRAME_TRANSLATION || kind == AGX_RES_FRAME_ROTATION) {
      int frame = frames ? frames[r] : -1;
      if (frame < 0) frame = rows.frame[r];
      const int jf = m.frame_parent[frame];
      double RF[9], pF[3];
      if (jf >= 0) {
        double Rp[9], pp[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) Rp[e] = g_bc(R[e], jf);
#pragma unroll
        for (int e = 0; e < 3; ++e) pp[e] = g_bc(p[e], jf);
        mm3(Rp, m.frame_placement[frame], RF);
        double tt[3];
        mv3(Rp, &m.frame_placement[frame][9], tt);
        pF[0] = pp[0] + tt[0]; pF[1] = pp[1] + tt[1]; pF[2] = pp[2] + tt[2];
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) RF[e] = m.frame_placement[frame][e];
#pragma unroll
        for (int e = 0; e < 3; ++e) pF[e] = m.frame_placement[frame][9 + e];
      }
      const bool on = jl && (l8 <= jf);
      double res[6], Jc[6];
      int nr;
      double dl[3], tz[3], lin[3], ang[3];
      dl[0] = pF[0] - p[0]; dl[1] = pF[1] - p[1]; dl[2] = pF[2] - p[2];
      cross3(S + 3, dl, tz);  // z x (pF - pj): world linear velocity of the frame per unit joint rate
      if (kind == AGX_RES_FRAME_PLACEMENT) {
        nr = 6;
        double Rrel[9], d3[3], prel[3], TL[9], TR[9];
        mtm3(rr, RF, Rrel);
        d3[0] = pF[0] - rr[9]; d3[1] = pF[1] - rr[10]; d3[2] = pF[2] - rr[11];
        mtv3(rr, d3, prel);
        log6<true>(Rrel, prel, res, TL, TR);
        mtv3(RF, tz, lin);
        mtv3(RF, S + 3, ang);
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          Jc[e] = TL[3 * e] * lin[0] + TL[3 * e + 1] * lin[1] + TL[3 * e + 2] * lin[2] + TR[3 * e] * ang[0] + TR[3 * e + 1] * ang[1] + TR[3 * e + 2] * ang[2];
          Jc[3 + e] = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
        }
      } else if (kind == AGX_RES_FRAME_TRANSLATION) {
        nr = 3;
        res[0] = pF[0] - rr[0]; res[1] = pF[1] - rr[1]; res[2] = pF[2] - rr[2];
        res[3] = res[4] = res[5] = 0.0;
        Jc[0] = tz[0]; Jc[1] = tz[1]; Jc[2] = tz[2];
        Jc[3] = Jc[4] = Jc[5] = 0.0;
      } else {
        nr = 3;
        double Rrel[9], TL[9];
        mtm3(rr, RF, Rrel);
        log3(Rrel, res);
        res[3] = res[4] = res[5] = 0.0;
        jlog3(res, TL);
        mtv3(RF, S + 3, ang);
#pragma unroll
        for (int e = 0; e < 3; ++e) Jc[e] = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
        Jc[3] = Jc[4] = Jc[5] = 0.0;
      }
      double we[6], a = 0.0;
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        we[e] = (e < nr) ? wi * aw[e] : 0.0;
        a += 0.5 * we[e] * res[e] * res[e];
        if (!on) Jc[e] = 0.0;
      }
      if (l8 == 0) cost += a;
      __syncthreads();  // previous users of the J tile are done
#pragma unroll
      for (int e = 0; e < 6; ++e) L.u.p2.J[l8][e] = Jc[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 6; ++e) Lq += we[e] * res[e] * Jc[e];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 6; ++e) acc += we[e] * L.u.p2.J[i][e] * Jc[e];
        Lqqc[i] += acc;
      }
    }
  }
  const double sc = TERM ? 1.0 : dt;
  cost = g_sum(cost) * sc;
  if (act && l8 == 0) qt[Q::cost] = cost;
  // ---- QP transformation, column l8 of every block
  const double lu = sc * Lu, D = sc * Luu + preg;
  L.vec[1][l8] = jl ? lu : 0.0;
  L.vec[2][l8] = jl ? D : 0.0;
  __syncthreads();
  double gw = 0.0, gq = sc * Lq, gv = sc * Lv;
  double DMc[NV], Dtq[NV], Dtv[NV];
#pragma unroll
  for (int l = 0; l < NV; ++l) {
    const double lul = L.vec[1][l], Dl = L.vec[2][l];
    gw += Mc[l] * lul;
    gq += tqc[l] * lul;
    gv += tvc[l] * lul;
    DMc[l] = Dl * Mc[l];
    Dtq[l] = Dl * tqc[l];
    Dtv[l] = Dl * tvc[l];
  }
  if (wr) {
    qt[Q::gw + l8] = TERM ? 0.0 : gw;
    qt[Q::gx + l8] = gq;
    qt[Q::gx + NV + l8] = gv;
    ax[A::Lvv + l8] = sc * Lvv;
    ax[A::Luu + l8] = sc * Luu;
    ax[A::Lu + l8] = lu;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double hww = 0.0, hqw = 0.0, hvw = 0.0, hqq = sc * Lqqc[i], hqv = 0.0, hvv = (i == l8) ? sc * Lvv : 0.0;
    if (!TERM) {
#pragma unroll
      for (int l = 0; l < NV; ++l) {
        const double Mil = L.M[i][l], tqli = L.u.p2.tq[l][i], tvli = L.u.p2.tv[l][i];
        hww += Mil * DMc[l];
        hqw += tqli * DMc[l];
        hvw += tvli * DMc[l];
        hqq += tqli * Dtq[l];
        hqv += tqli * Dtv[l];
        hvv += tvli * Dtv[l];
      }
    }
    if (wr) {
      qt[Q::Hww + i * NV + l8] = hww;
      qt[Q::Hqw + i * NV + l8] = hqw;
      qt[Q::Hvw + i * NV + l8] = hvw;
      qt[Q::Hqq + i * NV + l8] = hqq;
      qt[Q::Hqv + i * NV + l8] = hqv;
      qt[Q::Hvv + i * NV + l8] = hvv;
      ax[A::Lqq + i * NV + l8] = sc * Lqqc[i];
    }
  }
}

}  // namespace agx
