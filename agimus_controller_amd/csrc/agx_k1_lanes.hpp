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

#include <type_traits>

#include "agx_device.hpp"

namespace agx {


// Shifts inside the 8-lane group as DPP row shifts (VALU, no LDS traffic): a row is 16 lanes = two
// groups; values that cross from one group into the next land only in lanes the callers mask out
// (l8 < off for shift-up, l8 + off >= 8 for shift-down).  The LDS pipe of the CU is the scarce
// resource of this kernel -- ds_bpermute-based __shfl would put ~400 more instructions on it.
template <int OFF>
__device__ __forceinline__ double g_up(double x) { return dpp_mov<0x110 + OFF>(x); }  // row_shr:OFF -> lane i reads lane i-OFF
template <int OFF>
__device__ __forceinline__ double g_dn(double x) { return dpp_mov<0x100 + OFF>(x); }  // row_shl:OFF -> lane i reads lane i+OFF
__device__ __forceinline__ double g_bc(double x, int src) { return __shfl(x, src, 8); }

// inclusive prefix / suffix sums of N doubles over the 8-lane group
template <int N, int OFF>
__device__ __forceinline__ void g_prefix_step(double *x, int l8) {
  double y[N];
#pragma unroll
  for (int e = 0; e < N; ++e) y[e] = g_up<OFF>(x[e]);
  if (l8 >= OFF) {
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] += y[e];
  }
}
template <int N>
__device__ __forceinline__ void g_prefix_sum(double *x, int l8) {
  g_prefix_step<N, 1>(x, l8);
  g_prefix_step<N, 2>(x, l8);
  g_prefix_step<N, 4>(x, l8);
}
template <int N, int OFF>
__device__ __forceinline__ void g_suffix_step(double *x, int l8) {
  double y[N];
#pragma unroll
  for (int e = 0; e < N; ++e) y[e] = g_dn<OFF>(x[e]);
  if (l8 + OFF < 8) {
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] += y[e];
  }
}
template <int N>
__device__ __forceinline__ void g_suffix_sum(double *x, int l8) {
  g_suffix_step<N, 1>(x, l8);
  g_suffix_step<N, 2>(x, l8);
  g_suffix_step<N, 4>(x, l8);
}
// Two block rows at once as 16-byte stores: neighbouring lanes (columns c, c+1) swap one value so
// that the even lane owns [i][c..c+1] and the odd lane [i+1][c-1..c]: half as many store
// instructions, each lane writes an aligned double2, every row is still a whole 64-byte line.
__device__ __forceinline__ void store_row_pair(double *blk, int i, int l8, double a, double b) {
  const bool even = !(l8 & 1);
  const double y = dpp_mov<0xB1>(even ? b : a);  // quad_perm [1,0,3,2]: swap with the neighbour lane
  const double2 v = even ? make_double2(a, y) : make_double2(y, b);
  *reinterpret_cast<double2 *>(blk + (even ? i : i + 1) * 8 + (l8 & 6)) = v;
}

__device__ __forceinline__ double g_sum(double x) {
  x += __shfl_xor(x, 1, 8);
  x += __shfl_xor(x, 2, 8);
  x += __shfl_xor(x, 4, 8);
  return x;
}

// per-node LDS tile (doubles).  Phase 1 (dynamics): S, m6, Sd, psi, Dt.  Phase 2 (after the
// derivative matrices exist) reuses the same storage for tq, tv and the frame Jacobian J.
constexpr int kLjRef = 72;  // reference-tile doubles staged in LDS per node (larger tiles use the one-lane-per-node kernel)
// Per-node LDS tile.  The kernel runs in three phases that reuse the same storage:
//   c  (costs):      the node's reference tile, the frames its cost rows use, the frame Jacobian J
//   d1 (dynamics):   S, m6 = Ic S, Dt of every joint (the all-to-all operands of CRBA / RNEA derivatives)
//   d2 (transform):  the full dtau/dq, dtau/dqdot matrices
struct LjNode {
  union {
    struct { double ref[kLjRef], frm[4][14], J[8][6]; } c;  // frm: two frame rows, the two geometry frames of a collision row
    struct { double S[8][6], m6[8][6], Dt[8][4]; } d1;
    struct { double tq[8][8], tv[8][8]; } d2;
  } u;
  double M[8][8];
  double vec[3][8];  // rhs / lu / D
  int fpar[4];       // parent joints of the staged frames | their row indices
  int cpar[4];       // collision row: parent joints of the two geometry frames, row index, unused
};
// joint constants of the model, staged once per wave: placement 12 | axis 3 | com 3 | inertia 9 | mass | armature
struct LjModel {
  double j[8][30];
};

// Second argument of __launch_bounds__: workgroups (= waves here) per CU the compiler must leave room for.  It has
// no effect on this kernel up to 16 (measured: 252 VGPRs for every value): the 19 KB of LDS per wave (8 nodes x 2.1 KB
// + the staged joint constants) already cap residency at 8 waves per CU = 2 per SIMD, so the register allocator is
// free to use the whole file.  A third wave per SIMD needs the per-node LDS tile below 1.4 KB first.
#ifndef AGX_K1_WAVES
#define AGX_K1_WAVES 2
#endif
// COLL: the row table may hold ONE ResidualDistanceCollision cost row (closest points of the pair evaluated redundantly by
// the 8 lanes of the node, lane j its own column of the distance gradient); a template flag because the narrow-phase code
// would otherwise cost the collision-free kernel registers.
template <int NV, bool TERM, bool COLL = false>
__device__ __forceinline__ void calc_qp_lj_body(const long long blk, LjNode *lds, LjModel &lmod, const DevModel *__restrict__ mp,
                                                const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                const double *__restrict__ xs, const double *__restrict__ us, const RefView &rv,
                                                double *__restrict__ qts, double *__restrict__ auxs,
                                                const DevState *__restrict__ st, const int phase) {
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  const int l8 = threadIdx.x & 7;
  LjNode &L = lds[threadIdx.x >> 3];
  const long long n_nodes = TERM ? (long long)o.B : (long long)o.B * T;
  const long long node = (blk * blockDim.x + threadIdx.x) >> 3;
  const bool node_ok = node < n_nodes;
  const long long nid = node_ok ? node : 0;  // out-of-range groups shadow node 0 and store nothing
  const int b = TERM ? (int)nid : (int)(nid / T), t = TERM ? T : (int)(nid % T);
  const bool act = node_ok && k1_active(st[b], phase);  // phase 1: the trial points of the instances in the line search
  if (!__any(act)) return;  // the whole wave (= workgroup) belongs to instances this pass skips
  const bool jl = l8 < NV;       // lane carries a joint
  const int j = jl ? l8 : NV - 1;
  const double preg = k1_preg(st[b], phase);
  const double dt = TERM ? 0.0 : dts[t];
  const double *xp = xs + ((long long)b * (T + 1) + t) * NX;
  const double qj = xp[j], vj = jl ? xp[NV + j] : 0.0;
  const double uj = TERM ? 0.0 : us[((long long)b * T + t) * NV + j];
  double *qt = qts + ((long long)b * (T + 1) + t) * Q::SIZE;
  double *ax = auxs + ((long long)b * (T + 1) + t) * A::SIZE;
  const DevRows &rows = o.rows[TERM ? 1 : 0];
  const bool wr = act && jl;  // this lane stores

  // ---- prologue: every global read of the kernel is issued here, back to back, and parked in LDS
  // (one memory round trip instead of one per use; nothing is loaded after the first store)
  const double *gref = ref_at(rv, b, t, T);
  const int *gframes = frames_at(rv, b, t, T);
  {
    if (threadIdx.x < 8) {
      const int jj = threadIdx.x < NV ? threadIdx.x : NV - 1;
      double *d = lmod.j[threadIdx.x];
#pragma unroll
      for (int e = 0; e < 12; ++e) d[e] = m.placement[jj][e];
#pragma unroll
      for (int e = 0; e < 3; ++e) { d[12 + e] = m.axis[jj][e]; d[15 + e] = m.com[jj][e]; }
#pragma unroll
      for (int e = 0; e < 9; ++e) d[18 + e] = m.inertia[jj][e];
      d[27] = m.mass[jj];
      d[28] = m.armature[jj];
    }
    for (int e = l8; e < o.stride; e += 8) L.u.c.ref[e] = gref[e];  // host guarantees stride <= kLjRef
    // frames of the (up to two) frame-based cost rows
    int slot = 0;
    for (int r = 0; r < rows.n && slot < 2; ++r) {
      const int kind = rows.kind[r];
      if (kind != AGX_RES_FRAME_PLACEMENT && kind != AGX_RES_FRAME_TRANSLATION && kind != AGX_RES_FRAME_ROTATION) continue;
      int frame = gframes ? gframes[r] : -1;
      if (frame < 0) frame = rows.frame[r];
      for (int e = l8; e < 12; e += 8) L.u.c.frm[slot][e] = m.frame_placement[frame][e];
      if (l8 == 0) { L.fpar[slot] = m.frame_parent[frame]; L.fpar[2 + slot] = r; }
      ++slot;
    }
    if constexpr (COLL) {
      for (int r = 0; r < rows.n; ++r) {
        if (rows.kind[r] != AGX_RES_COLLISION) continue;
        for (int e = l8; e < 12; e += 8) { L.u.c.frm[2][e] = m.frame_placement[rows.frame[r]][e]; L.u.c.frm[3][e] = m.frame_placement[rows.frame_b[r]][e]; }
        if (l8 == 0) { L.cpar[0] = m.frame_parent[rows.frame[r]]; L.cpar[1] = m.frame_parent[rows.frame_b[r]]; L.cpar[2] = r; }
        break;
      }
    }
  }
  const double grav[3] = {m.gravity[0], m.gravity[1], m.gravity[2]};
  __syncthreads();
  const double *mj = lmod.j[l8];

  // ---- kinematics: local placement, then SE3 prefix product along the chain
  double R[9], p[3];
  {
    const double *ax3 = mj + 12;
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
    mm3(mj, Rq, R);
    p[0] = mj[9]; p[1] = mj[10]; p[2] = mj[11];
  }
  auto se3_step = [&](auto OFFc) {
    constexpr int OFF = decltype(OFFc)::value;
    double Rp[9], pp[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rp[e] = g_up<OFF>(R[e]);
#pragma unroll
    for (int e = 0; e < 3; ++e) pp[e] = g_up<OFF>(p[e]);
    if (l8 >= OFF) {
      double tt[3];
      mv3(Rp, p, tt);
      p[0] = pp[0] + tt[0]; p[1] = pp[1] + tt[1]; p[2] = pp[2] + tt[2];
      mm3(Rp, R, R);
    }
  };
  se3_step(std::integral_constant<int, 1>());
  se3_step(std::integral_constant<int, 2>());
  se3_step(std::integral_constant<int, 4>());
  double S[6];
  {
    double z[3];
    mv3(R, mj + 12, z);
    cross3(p, z, S);
    S[3] = z[0]; S[4] = z[1]; S[5] = z[2];
    if (!jl) {
#pragma unroll
      for (int e = 0; e < 6; ++e) S[e] = 0.0;
    }
  }
  // ---- cost rows: lane j owns component j of state / control terms and column j of J'WJ
  double cost = 0.0, Lq = 0.0, Lv = 0.0, Lu = 0.0, Lvv = 0.0, Luu = 0.0, Lqqc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) Lqqc[i] = 0.0;
  int fslot = 0;
  for (int r = 0; r < rows.n; ++r) {
    const int kind0 = rows.kind[r];
    const bool is_frame = kind0 == AGX_RES_FRAME_PLACEMENT || kind0 == AGX_RES_FRAME_TRANSLATION || kind0 == AGX_RES_FRAME_ROTATION;
    const int my_slot = fslot;
    if (is_frame) ++fslot;
    if (!rows.active[r]) continue;
    const double *tile = L.u.c.ref + rows.off[r];
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
      // staged frame data (first two frame rows), otherwise straight from the model
      const double *fpl = L.u.c.frm[my_slot];  // host guarantees at most two frame rows
      const int jf = L.fpar[my_slot];
      double RF[9], pF[3];
      if (jf >= 0) {
        double Rp[9], pp[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) Rp[e] = g_bc(R[e], jf);
#pragma unroll
        for (int e = 0; e < 3; ++e) pp[e] = g_bc(p[e], jf);
        mm3(Rp, fpl, RF);
        double tt[3];
        mv3(Rp, fpl + 9, tt);
        pF[0] = pp[0] + tt[0]; pF[1] = pp[1] + tt[1]; pF[2] = pp[2] + tt[2];
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) RF[e] = fpl[e];
#pragma unroll
        for (int e = 0; e < 3; ++e) pF[e] = fpl[9 + e];
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
      wave_lds_sync();  // previous users of the J tile are done
#pragma unroll
      for (int e = 0; e < 6; ++e) L.u.c.J[l8][e] = Jc[e];
      wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 6; ++e) Lq += we[e] * res[e] * Jc[e];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 6; ++e) acc += we[e] * L.u.c.J[i][e] * Jc[e];
        Lqqc[i] += acc;
      }
    }
  }
  if constexpr (COLL) {
    // colmpc.ResidualDistanceCollision (ocp_croco_generic.py:524-533) with a scalar activation: one row per problem here
    for (int r = 0; r < rows.n; ++r) {
      if (rows.kind[r] != AGX_RES_COLLISION) continue;
      if (!rows.active[r]) break;
      const double *tile = L.u.c.ref + rows.off[r];
      const double wi = tile[0], aw0 = tile[1 + rows.nref[r]];
      double Rg[2][9], pg[2][3];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const double *fpl = L.u.c.frm[2 + g];
        const int jf = L.cpar[g];
        if (jf >= 0) {
          double Rp[9], pp[3], tt[3];
#pragma unroll
          for (int e = 0; e < 9; ++e) Rp[e] = g_bc(R[e], jf);
#pragma unroll
          for (int e = 0; e < 3; ++e) pp[e] = g_bc(p[e], jf);
          mm3(Rp, fpl, Rg[g]);
          mv3(Rp, fpl + 9, tt);
          pg[g][0] = pp[0] + tt[0]; pg[g][1] = pp[1] + tt[1]; pg[g][2] = pp[2] + tt[2];
        } else {
#pragma unroll
          for (int e = 0; e < 9; ++e) Rg[g][e] = fpl[e];
#pragma unroll
          for (int e = 0; e < 3; ++e) pg[g][e] = fpl[9 + e];
        }
      }
      double ca[3], cb[3], nn[3];
      const double d = collision_distance_placed(m, rows.frame[r], rows.frame_b[r], Rg[0], pg[0], Rg[1], pg[1], ca, cb, nn);
      double a, ar, arr;
      activation1(rows.act[r], rows.alpha[r], aw0, d, a, ar, arr);
      if (l8 == 0) cost += wi * a;
      const bool ona = jl && (l8 <= L.cpar[0]), onb = jl && (l8 <= L.cpar[1]);
      double da[3], db[3], ta[3], tb[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) { da[e] = ca[e] - p[e]; db[e] = cb[e] - p[e]; }
      cross3(S + 3, da, ta);
      cross3(S + 3, db, tb);
      const double gj = (ona ? dot3(nn, ta) : 0.0) - (onb ? dot3(nn, tb) : 0.0);
      Lq += wi * ar * gj;
      wave_lds_sync();  // previous users of the J tile are done
      L.u.c.J[l8][0] = gj;
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < NV; ++i) Lqqc[i] += wi * arr * L.u.c.J[i][0] * gj;
      break;
    }
  }
  wave_lds_sync();  // the cost phase's LDS (reference tile, frames, J) is dead from here on

  // body inertia in the world frame
  double Ib[10];
  {
    double cw[3];
    mv3(R, mj + 15, cw);
    cw[0] += p[0]; cw[1] += p[1]; cw[2] += p[2];
    const double ms = jl ? mj[27] : 0.0;
    double Tm[9], Iw[9];
    mm3(R, mj + 18, Tm);
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
  double v6[6], Sd[6], h6[6], tqc[NV], tvc[NV], Mc[NV], qdd = 0.0, gapq = 0.0, gapv = 0.0;
  // successor state for the gap, fetched before anything is stored
  const double xnq = TERM ? 0.0 : xp[NX + j], xnv = TERM ? 0.0 : xp[NX + NV + j];
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
    a0[0] -= grav[0]; a0[1] -= grav[1]; a0[2] -= grav[2];
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
    for (int e = 0; e < 6; ++e) { L.u.d1.S[l8][e] = S[e]; L.u.d1.m6[l8][e] = m6[e]; }
    L.vec[0][l8] = uj - nle;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double val;
      if (i >= l8) val = dot6(S, L.u.d1.m6[i]);
      else val = dot6(L.u.d1.S[i], m6);
      if (i == l8) val += mj[28];
      Mc[i] = val;
      L.M[i][l8] = val;
    }
    wave_lds_sync();
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
    // gap f = xnext - xs[t+1]  (stored at the end: no load may queue behind a store)
    gapq = qj + dt * vj + dt * dt * qdd - xnq;
    gapv = vj + dt * qdd - xnv;
    // ---- pass B: accelerations with qdd, psi, composite force / momentum / E
    double a6[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) a6[e] = S[e] * qdd + Sd[e] * vj;
    g_prefix_sum<6>(a6, l8);
    a6[0] -= grav[0]; a6[1] -= grav[1]; a6[2] -= grav[2];
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
    L.u.d1.Dt[l8][0] = Dt[0]; L.u.d1.Dt[l8][1] = Dt[1]; L.u.d1.Dt[l8][2] = Dt[2];
    wave_lds_sync();
    // column l8 of dtau/dq (tqc) and dtau/dqdot (tvc)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double dvv, dqq;
      if (i >= l8) {
        const double *mi = L.u.d1.m6[i], *Di = L.u.d1.Dt[i];
        dvv = 2.0 * dot6(mi, Sd) + dot3(Di, S + 3);
        dqq = dot3(Di, Sd + 3) + dot6(mi, psi);
      } else {
        const double *Si = L.u.d1.S[i];
        dvv = dot6(Si, colv);
        dqq = dot6(Si, colq);
      }
      tvc[i] = jl ? dvv : 0.0;
      tqc[i] = jl ? dqq : 0.0;
    }
    wave_lds_sync();  // phase 1 storage is dead from here on
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      L.u.d2.tq[i][l8] = tqc[i];
      L.u.d2.tv[i][l8] = tvc[i];
    }
  } else {
    wave_lds_sync();
  }

  const double sc = TERM ? 1.0 : dt;
  cost = g_sum(cost) * sc;
  if (act && l8 == 0) qt[Q::cost] = cost;
  // ---- QP transformation, column l8 of every block
  const double lu = sc * Lu, D = sc * Luu + preg;
  L.vec[1][l8] = jl ? lu : 0.0;
  L.vec[2][l8] = jl ? D : 0.0;
  wave_lds_sync();
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
  // all 8 lanes store (lanes >= NV write the zero padding): every block row is one whole 64-byte line
  auto store_block = [&](double *blk, const double *col) {
#pragma unroll
    for (int i = 0; i + 1 < NV; i += 2) store_row_pair(blk, i, l8, jl ? col[i] : 0.0, jl ? col[i + 1] : 0.0);
    if (NV & 1) blk[(NV - 1) * 8 + l8] = jl ? col[NV - 1] : 0.0;
  };
  if (act) {
    store_block(ax + A::M, Mc);
    store_block(ax + A::tq, tqc);
    store_block(ax + A::tv, tvc);
  }
  if (wr) {
    qt[Q::f + l8] = gapq;
    qt[Q::f + NV + l8] = gapv;
    qt[Q::gw + l8] = TERM ? 0.0 : gw;
    qt[Q::gx + l8] = gq;
    qt[Q::gx + NV + l8] = gv;
    ax[A::Lvv + l8] = sc * Lvv;
    ax[A::Luu + l8] = sc * Luu;
    ax[A::Lu + l8] = lu;
  }
  double cww[NV], cqw[NV], cvw[NV], cqq[NV], cqv[NV], cvv[NV], clq[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double hww = 0.0, hqw = 0.0, hvw = 0.0, hqq = sc * Lqqc[i], hqv = 0.0, hvv = (i == l8) ? sc * Lvv : 0.0;
    if (!TERM) {
#pragma unroll
      for (int l = 0; l < NV; ++l) {
        const double Mil = L.M[i][l], tqli = L.u.d2.tq[l][i], tvli = L.u.d2.tv[l][i];
        hww += Mil * DMc[l];
        hqw += tqli * DMc[l];
        hvw += tvli * DMc[l];
        hqq += tqli * Dtq[l];
        hqv += tqli * Dtv[l];
        hvv += tvli * Dtv[l];
      }
    }
    cww[i] = hww; cqw[i] = hqw; cvw[i] = hvw; cqq[i] = hqq; cqv[i] = hqv; cvv[i] = hvv; clq[i] = sc * Lqqc[i];
  }
  if (act) {
    store_block(qt + Q::Hww, cww);
    store_block(qt + Q::Hqw, cqw);
    store_block(qt + Q::Hvw, cvw);
    store_block(qt + Q::Hqq, cqq);
    store_block(qt + Q::Hqv, cqv);
    store_block(qt + Q::Hvv, cvv);
    store_block(ax + A::Lqq, clq);
  }
}

// Separate launches (timing, terminal-only / running-only callers)
template <int NV, bool TERM, bool COLL = false>
__global__ void __launch_bounds__(64, AGX_K1_WAVES) k_calc_qp_lj(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                    const double *__restrict__ dts, const double *__restrict__ xs,
                                                    const double *__restrict__ us, RefView rv, double *__restrict__ qts,
                                                    double *__restrict__ auxs, const DevState *__restrict__ st, int phase) {
  __shared__ LjNode lds[8];  // one wave per workgroup: 8 nodes
  __shared__ LjModel lmod;
  calc_qp_lj_body<NV, TERM, COLL>(blockIdx.x, lds, lmod, mp, op, dts, xs, us, rv, qts, auxs, st, phase);
}

// The derivative pass of one SQP iteration in ONE launch: the first n_run workgroups take the running
// nodes, the rest the terminal nodes (wave-uniform branch, shared LDS declarations): the short
// terminal launch and its dispatch gap disappear behind the tail of the running nodes.
template <int NV, bool COLL = false>
__global__ void __launch_bounds__(64, AGX_K1_WAVES) k_calc_qp_lj_all(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                        const double *__restrict__ dts, const double *__restrict__ xs,
                                                        const double *__restrict__ us, RefView rv, double *__restrict__ qts,
                                                        double *__restrict__ auxs, const DevState *__restrict__ st, int n_run,
                                                        int phase) {
  __shared__ LjNode lds[8];
  __shared__ LjModel lmod;
  if ((int)blockIdx.x < n_run)
    calc_qp_lj_body<NV, false, COLL>(blockIdx.x, lds, lmod, mp, op, dts, xs, us, rv, qts, auxs, st, phase);
  else
    calc_qp_lj_body<NV, true, COLL>((long long)blockIdx.x - n_run, lds, lmod, mp, op, dts, xs, us, rv, qts, auxs, st, phase);
}

// Constraint values, Jacobian rows and the l1 violation of every node (k_con_eval, agx_admm.hpp) with 8 lanes per node for the
// row kinds of the shipped problems: Control (ConstraintModelControlLimit), State, collision distance.  Lane j holds joint
// j: the SE3 prefix product of k_calc_qp_lj gives the world placements, the closest points of a pair are evaluated
// redundantly by the 8 lanes, lane j writes its column of the distance gradient.  The one-lane kernel (512 VGPRs, private
// arrays) took 90 us per launch at B = 256, T = 200; problems with other constraint kinds still use it.
// cg [B][T+1][AGX_MAX_NC], cjac [B][T+1][AGX_MAX_DENSE][24] (d/dq | d/dv | d/du, 8 each).
template <int NV>
__global__ void __launch_bounds__(64) k_con_eval_lj(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                    const double *__restrict__ xs, const double *__restrict__ us,
                                                    double *__restrict__ cg, double *__restrict__ cjac,
                                                    double *__restrict__ nodestat, const DevState *__restrict__ st, int phase) {
  constexpr int NX = 2 * NV;
  __shared__ double s_mod[8][16];  // placement 12 | axis 3 of every joint
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T, l8 = threadIdx.x & 7;
  const long long n_nodes = (long long)o.B * (T + 1);
  const long long node_raw = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const bool ok = node_raw < n_nodes;
  const long long node = ok ? node_raw : n_nodes - 1;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  const bool act = ok && k1_active(st[b], phase);
  if (!__any(act)) return;
  if (threadIdx.x < 8) {
    const int jj = threadIdx.x < NV ? threadIdx.x : NV - 1;
#pragma unroll
    for (int e = 0; e < 12; ++e) s_mod[threadIdx.x][e] = m.placement[jj][e];
#pragma unroll
    for (int e = 0; e < 3; ++e) s_mod[threadIdx.x][12 + e] = m.axis[jj][e];
  }
  __syncthreads();
  const DevCons &c = o.cons[t == T ? 1 : 0];
  const bool jl = l8 < NV;
  const int j = jl ? l8 : NV - 1;
  const double *xp = xs + node * NX;
  const double qj = xp[j], vj = xp[NV + j];
  const double uj = (t < T) ? us[((long long)b * T + t) * NV + j] : 0.0;
  // kinematics: local placement, then the SE3 prefix product along the chain (as calc_qp_lj_body)
  double R[9], p[3], z[3];
  if (c.ncoll > 0) {
    const double *mj = s_mod[l8];
    const double *ax3 = mj + 12;
    double sn, cs;
    sincos(qj, &sn, &cs);
    const double omc = 1.0 - cs;
    double Rq[9];
    Rq[0] = cs + omc * ax3[0] * ax3[0];
    Rq[1] = omc * ax3[0] * ax3[1] - sn * ax3[2];
    Rq[2] = omc * ax3[0] * ax3[2] + sn * ax3[1];
    Rq[3] = omc * ax3[1] * ax3[0] + sn * ax3[2];
    Rq[4] = cs + omc * ax3[1] * ax3[1];
    Rq[5] = omc * ax3[1] * ax3[2] - sn * ax3[0];
    Rq[6] = omc * ax3[2] * ax3[0] - sn * ax3[1];
    Rq[7] = omc * ax3[2] * ax3[1] + sn * ax3[0];
    Rq[8] = cs + omc * ax3[2] * ax3[2];
    mm3(mj, Rq, R);
    p[0] = mj[9]; p[1] = mj[10]; p[2] = mj[11];
    auto se3_step = [&](auto OFFc) {
      constexpr int OFF = decltype(OFFc)::value;
      double Rp[9], pp[3];
#pragma unroll
      for (int e = 0; e < 9; ++e) Rp[e] = g_up<OFF>(R[e]);
#pragma unroll
      for (int e = 0; e < 3; ++e) pp[e] = g_up<OFF>(p[e]);
      if (l8 >= OFF) {
        double tt[3];
        mv3(Rp, p, tt);
        p[0] = pp[0] + tt[0]; p[1] = pp[1] + tt[1]; p[2] = pp[2] + tt[2];
        mm3(Rp, R, R);
      }
    };
    se3_step(std::integral_constant<int, 1>());
    se3_step(std::integral_constant<int, 2>());
    se3_step(std::integral_constant<int, 4>());
    mv3(R, ax3, z);  // joint axis in the world
  }
  double v = 0.0;
  for (int r = 0; r < c.n; ++r) {
    const int kind = c.kind[r], off = c.off[r];
    if (kind == AGX_RES_CONTROL) {
      const double g = uj - c.ref[r][j];
      if (jl) {
        if (act) cg[node * AGX_MAX_NC + off + j] = g;
        v += fmax(c.lb[off + j] - g, 0.0) + fmax(g - c.ub[off + j], 0.0);
      }
    } else if (kind == AGX_RES_STATE) {
      const double gq = qj - c.ref[r][j], gv = vj - c.ref[r][NV + j];
      if (jl) {
        if (act) { cg[node * AGX_MAX_NC + off + j] = gq; cg[node * AGX_MAX_NC + off + NV + j] = gv; }
        v += fmax(c.lb[off + j] - gq, 0.0) + fmax(gq - c.ub[off + j], 0.0) + fmax(c.lb[off + NV + j] - gv, 0.0) + fmax(gv - c.ub[off + NV + j], 0.0);
      }
    } else {  // collision distance (the host selects this kernel only for these three kinds)
      double Rg[2][9], pg[2][3];
      int jp[2];
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        const int frame = gi == 0 ? c.frame[r] : c.frame_b[r];
        const double *fpl = m.frame_placement[frame];
        const int jf = m.frame_parent[frame];
        jp[gi] = jf;
        if (jf >= 0) {
          double Rp[9], pp[3], tt[3];
#pragma unroll
          for (int e = 0; e < 9; ++e) Rp[e] = g_bc(R[e], jf);
#pragma unroll
          for (int e = 0; e < 3; ++e) pp[e] = g_bc(p[e], jf);
          mm3(Rp, fpl, Rg[gi]);
          mv3(Rp, fpl + 9, tt);
          pg[gi][0] = pp[0] + tt[0]; pg[gi][1] = pp[1] + tt[1]; pg[gi][2] = pp[2] + tt[2];
        } else {
#pragma unroll
          for (int e = 0; e < 9; ++e) Rg[gi][e] = fpl[e];
#pragma unroll
          for (int e = 0; e < 3; ++e) pg[gi][e] = fpl[9 + e];
        }
      }
      double ca[3], cb[3], nn[3];
      const double d = collision_distance_placed(m, c.frame[r], c.frame_b[r], Rg[0], pg[0], Rg[1], pg[1], ca, cb, nn);
      const bool ona = jl && (l8 <= jp[0]), onb = jl && (l8 <= jp[1]);  // serial chain: joints up to the parent move the frame
      double da[3], db[3], ta[3], tb[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) { da[e] = ca[e] - p[e]; db[e] = cb[e] - p[e]; }
      cross3(z, da, ta);
      cross3(z, db, tb);
      const double gj = (ona ? dot3(nn, ta) : 0.0) - (onb ? dot3(nn, tb) : 0.0);
      if (act) {
        double *row = cjac + (node * AGX_MAX_DENSE + c.coll_slot[r]) * 24;
        row[l8] = jl ? gj : 0.0; row[8 + l8] = 0.0; row[16 + l8] = 0.0;
        if (l8 == 0) cg[node * AGX_MAX_NC + off] = d;
      }
      if (l8 == 0) v += fmax(c.lb[off] - d, 0.0) + fmax(d - c.ub[off], 0.0);
    }
  }
  v += dpp_xor4(v); v += dpp_xor2(v); v += dpp_xor1(v);
  if (act && l8 == 0) nodestat[node * 4 + 3] = v;
}

}  // namespace agx
