// agx_big_k1.hpp -- node evaluation for large models (8 < nv <= 32, trees): ONE WORKGROUP PER NODE.
//
// The register-resident kernels (agx_k1_lanes.hpp) give a node 8 lanes; a 30-joint tree does not fit
// that shape, and one lane per node needs ~58 KB of private arrays per lane (scratch).  Here a node
// owns a 256-thread workgroup and ~52 KB of LDS, nothing lives in scratch:
//   * per-joint quantities (placements, S, v, a, inertias, forces) are LDS vectors; recursions along the
//     tree become masked sums over the ancestor / descendant bit sets of the model (no level
//     synchronisation: v_i = sum_{j in anc(i)} S_j qd_j, f^C_i = sum_{j in desc(i)} f_j, ...),
//     one thread per (joint, component);
//   * all-pairs quantities (CRBA, the RNEA-derivative matrices) are one thread per matrix entry;
//   * M qdd = u - nle is solved by one wave, a row per lane, Gauss-Jordan with v_readlane broadcasts;
//   * the dense contractions of the acceleration-input transformation,
//       [tq tv M]' diag(D) [tq tv M]   (six nv x nv x nv products)   and   J' W J  of the cost rows,
//     run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, nv padded to 32: wave w owns output tile
//     (w >> 1, w & 1) of every block, operands straight from LDS);
//   * tiles leave as whole 128-byte row segments.
// Same mathematics as agx_device.hpp (bias_and_inertia, rnea_derivatives, node_costs); same QP / aux
// tiles as the one-lane kernel it replaces.  k_ls_trial_wg is the value-only variant for the line search.
//
// What this stands in for upstream: calc + calcDiff of IntegratedActionModelEuler(
// DifferentialActionModelFreeFwdDynamics(CostModelSum)), agimus_controller/ocp/ocp_croco_generic.py:688-711,743-745.
//
// (included at the end of agx_kernels.hpp)
#pragma once

namespace agx {

constexpr int kWgRef = 256;  // reference-tile doubles staged per node (agx_ocp_create refuses larger tiles for nv > 8)
constexpr int kWgJ = 16;     // scalar residual rows with a dense gradient in q per node: FramePlacement 6,
                             // FrameTranslation / FrameRotation 3, collision 1 (agx_ocp_create checks the total)

typedef double agx_v4d __attribute__((ext_vector_type(4)));

// Development builds (-DAGX_WG_PROFILE): thread 0 of workgroup 777 leaves a cycle stamp at every phase boundary of
// wg_node (scripts/time_wg_phases.py prints the differences).
#ifdef AGX_WG_PROFILE
__device__ long long g_wg_ts[64];
__device__ int g_wg_n;
#define AGX_WG_STAMP()                                                                       \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 777 && DIFF && !TERM) { g_wg_ts[g_wg_n] = wall_clock64(); g_wg_n += 1; } \
  } while (0)
#else
#define AGX_WG_STAMP() do { } while (0)
#endif

// workgroups (= waves per SIMD) of the derivative pass a CU is asked to hold: 2 leaves the kernel its ~200 VGPRs (1.48 ms
// per launch at B = 512, T = 50), 3 (what the 52 KB of LDS allow) caps them at 168 and spills ~30 values that live
// across the solve / cost-row region (268 B per lane, no arrays): 1.24 ms
#ifndef AGX_WG_MINWAVES
#define AGX_WG_MINWAVES 3
#endif

template <int NV>
struct WgNode {
  static constexpr int LDM = 32;
  alignas(16) double M[NV][LDM], tq[NV][LDM], tv[NV][LDM];  // columns >= NV stay zero (MFMA operands); copied out as double2
  double S[NV][6], Sd[NV][6], v[NV][6], m6[NV][6];
  double Ib[NV][10], Ic[NV][10];
  double x[2 * NV], u[NV], xn[2 * NV];
  double nle[32], rhs[32], qdd[32], D[32], lu[32], Lq[32], Lv[32], Lvv[32], Luu[32], dqq[32], cpart[32], fq[32], fv[32];
  // Working storage, reused phase by phase (doubles, offsets in comments):
  //   kinematics   Rl [0, 360)                     | Rw [540, 810) pw [810, 900) ref [900, 1156) J [1156, 1668) wJ [1668, 1684)
  //   bias forces  a0 f fc [0, 540)                |  ... cost rows and M qdd = rhs run side by side on Rw .. wJ ...
  //   derivatives  a psi [0, 360)  pre [360, 900)  (-> Dt colv colq)      cmp [900, 1440)   (ref, J are dead by then)
  union {
    struct {
      union {
        double Rl[NV][12];                                     // local placements (dead once the world placements exist)
        struct { double a[NV][6], f[NV][6], fc[NV][6]; } d1;   // bias forces
      };
      double Rw[NV][9], pw[NV][3], ref[kWgRef], J[kWgJ][LDM], wJ[kWgJ];
    } c;
    struct {
      double a[NV][6], psi[NV][6];
      union {
        double pre[NV][18];  // fC (6) | f0 (3) | E (9) of the body; dead once the subtree sums exist
        struct { double Dt[NV][3], colv[NV][6], colq[NV][6]; } col;
      };
      double cmp[NV][18];  // subtree sums of pre
    } d3;
  } w;
  unsigned anc[32], desc[32];
  int par[32];
};

// x = M^-1 rhs for the SPD joint-space inertia, one wave: lane i keeps row i of [M | rhs] in registers,
// nv Gauss-Jordan pivots, the pivot row travels through v_readlane (no LDS, no barrier).
template <int NV>
__device__ __forceinline__ double wave_spd_solve(const double (*M)[32], const double *rhs, int lane) {
  const int i = lane < NV ? lane : NV - 1;
  double a[NV + 1];
#pragma unroll
  for (int j = 0; j < NV; ++j) a[j] = M[i][j];
  a[NV] = rhs[i];
  double d = 1.0;  // the row's own pivot (final once column i has been eliminated everywhere else)
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double piv = readlane_f64(a[k], k);
    const double rp = fast_rcp(piv);
    const double f = (lane == k) ? 0.0 : a[k] * rp;  // the pivot row itself stays
    if (i == k) d = piv;
#pragma unroll
    for (int j = k + 1; j <= NV; ++j) a[j] -= f * readlane_f64(a[j], k);
  }
  return a[NV] / d;
}

template <int NV>
__device__ __forceinline__ void wg_frame_world(const WgNode<NV> &L, const DevModel &m, int frame, double *R, double *p, int *joint) {
  const int par = m.frame_parent[frame];
  *joint = par;
  const double *fp = m.frame_placement[frame];
  if (par >= 0) {
    const double *Rp = L.w.c.Rw[par], *pp = L.w.c.pw[par];
    mm3(Rp, fp, R);
    double t[3];
    mv3(Rp, fp + 9, t);
    p[0] = pp[0] + t[0]; p[1] = pp[1] + t[1]; p[2] = pp[2] + t[2];
  } else {
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = fp[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) p[e] = fp[9 + e];
  }
}

// ---- fp64 matrix-core helpers: wave w of the 4-wave workgroup owns tile (ti, tj) = (w >> 1, w & 1) of a 32 x 32
// result.  v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md): lane l supplies A[m = l & 15][k = l >> 4] and
// B[k = l >> 4][n = l & 15]; result register q holds D[m = (l >> 4) + 4 q][n = l & 15].
template <int NV>
__device__ __forceinline__ double wg_op(const double (*X)[32], int k, int col) { return k < NV ? X[k][col] : 0.0; }

// acc[m][n] += sum_k fa(m, k) fb(k, n) over KSTEPS k-steps of 4
template <int KSTEPS, class FA, class FB>
__device__ __forceinline__ agx_v4d wg_mma(int ti, int tj, int lane, FA fa, FB fb, agx_v4d acc) {
  const int m = 16 * ti + (lane & 15), n = 16 * tj + (lane & 15), l4 = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(m, 4 * ks + l4), fb(4 * ks + l4, n), acc, 0, 0, 0);
  return acc;
}
// Masked sums over the tree on the matrix cores: out[i][n] = sum_{k in set(i)} fb(k, n), set(i) = bits of mask[i]
// (ancestors: recursions root -> leaf such as v_i = sum S_k qd_k; descendants: subtree sums such as composite
// inertias / forces).  The 0 / 1 operand is exact; no level synchronisation, no data-dependent loops.
template <class FB>
__device__ __forceinline__ agx_v4d wg_mask_mma(const unsigned *mask, int ti, int tj, int lane, FB fb) {
  const unsigned mrow = mask[16 * ti + (lane & 15)];
  const agx_v4d zero = {0.0, 0.0, 0.0, 0.0};
  return wg_mma<8>(ti, tj, lane, [&](int, int k) { return ((mrow >> k) & 1u) ? 1.0 : 0.0; }, fb, zero);
}
// rows (lane >> 4) + 4 q, column lane & 15 of tile (ti, tj) -> f(row, col, value)
template <class F>
__device__ __forceinline__ void wg_scatter(const agx_v4d &acc, int ti, int tj, int lane, F f) {
#pragma unroll
  for (int q = 0; q < 4; ++q) f(16 * ti + (lane >> 4) + 4 * q, 16 * tj + (lane & 15), acc[q]);
}

// Tile (ti, tj) of a 32-column block to HBM as 16-byte stores: neighbouring lanes (columns c, c + 1) swap one value so
// that the even lane owns [row a][c .. c + 1] and the odd lane [row b][c - 1 .. c] of a pair of result rows (a, b):
// half as many store instructions, every row segment still a whole 128-byte line (columns nv..31 carry zeros).
__device__ __forceinline__ void wg_store_tile(double *__restrict__ blk, const agx_v4d &acc, int ti, int tj, int lane, int nv) {
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool even = !(l15 & 1);
#pragma unroll
  for (int qp = 0; qp < 2; ++qp) {
    const double a = acc[2 * qp], b = acc[2 * qp + 1];
    const double y = dpp_mov<0xB1>(even ? b : a);  // quad_perm [1,0,3,2]: swap with the neighbour lane
    const double2 v = even ? make_double2(a, y) : make_double2(y, b);
    const int row = 16 * ti + l4 + 4 * (2 * qp + (even ? 0 : 1));
    if (row < nv) *reinterpret_cast<double2 *>(blk + row * 32 + 16 * tj + (l15 & 14)) = v;
  }
}

// Inputs of one node evaluation.  LS: the point is (xs + alpha dx, us + alpha du).
struct WgIn {
  const double *x, *dx, *u, *du, *xn, *dxn;  // node state / control / successor state (and the direction, may be null)
  double alpha, dt, preg, mu_dyn;
  const double *ref;  // the node's reference tile (stride doubles)
  int stride;
  const int *frames;
  bool kin_only = false;  // stop after the kinematics (k_con_eval_wg)
};

// Cost rows of one node, evaluated by ONE wave (lane j: component j of the state / control rows, column j of every
// dense row; lanes >= nv shadow the last joint and store nothing).  Runs next to the wave that solves M qdd = rhs.
template <int NV, bool TERM, bool DIFF>
__device__ __forceinline__ void wg_costs(WgNode<NV> &L, const DevModel &m, const DevRows &rows, const WgIn &in, int lane, double sc) {
  int nJ = 0;
  const int j = lane < NV ? lane : NV - 1;
  const bool jl = lane < NV;
  const double qj = L.x[j], vj = L.x[NV + j], uj = TERM ? 0.0 : L.u[j];
  double cost = 0.0, Lq = 0.0, Lv = 0.0, Lu = 0.0, Lvv = 0.0, Luu = 0.0, dqq = 0.0;
  for (int r = 0; r < rows.n; ++r) {
    if (!rows.active[r]) continue;
    const double *tile = L.w.c.ref + rows.off[r];
    const double wi = tile[0];
    const double *rr = tile + 1;
    const double *aw = rr + rows.nref[r];
    const int kind = rows.kind[r];
    const bool quad = rows.act[r] == AGX_ACT_WEIGHTED_QUAD;
    // sum over the lanes of the wave (Exp / QuadExp activations are functions of |r|^2 of the whole row)
    auto wave_sum = [](double v) {
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) v += __shfl_xor(v, sft, 64);
      return v;
    };
    if (kind == AGX_RES_STATE && !quad) {
      const double rq = qj - rr[j], rvv = vj - rr[NV + j];
      const ActVec A = activation_vec(rows.act[r], rows.alpha[r], wave_sum(jl ? rq * rq + rvv * rvv : 0.0));
      const bool real = j < rows.nvu;  // pad joints: see DevRows::nvu
      if (lane == 0) cost += wi * A.a;
      Lq += wi * A.c1 * rq; Lv += wi * A.c1 * rvv;
      Lvv += real ? wi * (A.c2 + A.c3 * rvv * rvv) : 1.0;
      dqq += real ? wi * (A.c2 + A.c3 * rq * rq) : 1.0;
    } else if (kind == AGX_RES_STATE) {
      const double rq = qj - rr[j], rvv = vj - rr[NV + j];
      const double wq = wi * aw[j], wv = wi * aw[NV + j];
      cost += 0.5 * (wq * rq * rq + wv * rvv * rvv);
      Lq += wq * rq; Lv += wv * rvv; Lvv += wv; dqq += wq;
    } else if (kind == AGX_RES_CONTROL && !quad) {
      if (!TERM) {
        const double ru = uj - rr[j];
        const ActVec A = activation_vec(rows.act[r], rows.alpha[r], wave_sum(jl ? ru * ru : 0.0));
        if (lane == 0) cost += wi * A.a;
        Lu += wi * A.c1 * ru;
        Luu += (j < rows.nvu) ? wi * (A.c2 + A.c3 * ru * ru) : 1.0;
      }
    } else if (kind == AGX_RES_CONTROL) {
      if (!TERM) {
        const double ru = uj - rr[j], wu = wi * aw[j];
        cost += 0.5 * wu * ru * ru;
        Lu += wu * ru; Luu += wu;
      }
    } else if (kind == AGX_RES_FRAME_PLACEMENT || kind == AGX_RES_FRAME_TRANSLATION || kind == AGX_RES_FRAME_ROTATION) {
      int frame = in.frames ? in.frames[r] : -1;
      if (frame < 0) frame = rows.frame[r];
      double RF[9], pF[3];
      int jf;
      wg_frame_world<NV>(L, m, frame, RF, pF, &jf);
      const bool on = (jf >= 0) && ((L.anc[jf >= 0 ? jf : 0] >> j) & 1u);
      const double *Sj = L.S[j], *pj = L.w.c.pw[j];
      double res[6], Jc[6], dl[3], tz[3], lin[3], ang[3];
      int nr;
      dl[0] = pF[0] - pj[0]; dl[1] = pF[1] - pj[1]; dl[2] = pF[2] - pj[2];
      cross3(Sj + 3, dl, tz);  // z x (pF - pj)
      if (kind == AGX_RES_FRAME_PLACEMENT) {
        nr = 6;
        double Rrel[9], d3[3], prel[3], TL[9], TR[9];
        mtm3(rr, RF, Rrel);
        d3[0] = pF[0] - rr[9]; d3[1] = pF[1] - rr[10]; d3[2] = pF[2] - rr[11];
        mtv3(rr, d3, prel);
        log6<DIFF>(Rrel, prel, res, TL, TR);
        if (DIFF) {
          mtv3(RF, tz, lin);
          mtv3(RF, Sj + 3, ang);
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            Jc[e] = TL[3 * e] * lin[0] + TL[3 * e + 1] * lin[1] + TL[3 * e + 2] * lin[2] + TR[3 * e] * ang[0] + TR[3 * e + 1] * ang[1] + TR[3 * e + 2] * ang[2];
            Jc[3 + e] = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
          }
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
        Jc[3] = Jc[4] = Jc[5] = 0.0;
        if (DIFF) {
          jlog3(res, TL);
          mtv3(RF, Sj + 3, ang);
#pragma unroll
          for (int e = 0; e < 3; ++e) Jc[e] = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
        }
      }
      double a = 0.0;
      ActVec A = {0.0, 0.0, 0.0, 0.0};
      if (!quad) {
        double n2 = 0.0;
#pragma unroll
        for (int e = 0; e < 6; ++e)
          if (e < nr) n2 += res[e] * res[e];
        A = activation_vec(rows.act[r], rows.alpha[r], n2);
        a = wi * A.a;
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        // weights of the Gauss-Newton Hessian (J' diag(we) J) and of the gradient (J' ge) of the row's activation
        const double we = (e < nr) ? (quad ? wi * aw[e] : wi * (A.c2 + A.c3 * res[e] * res[e])) : 0.0;
        const double ge = (e < nr) ? (quad ? we * res[e] : wi * A.c1 * res[e]) : 0.0;
        if (quad) a += 0.5 * we * res[e] * res[e];
        if (DIFF && e < nr) {
          const double jc = on ? Jc[e] : 0.0;
          Lq += ge * jc;
          if (jl) L.w.c.J[nJ + e][j] = jc;
          if (lane == 0) L.w.c.wJ[nJ + e] = we;
        }
      }
      if (lane == 0) cost += a;
      nJ += nr;
    } else if (kind == AGX_RES_COLLISION) {
      // colmpc.ResidualDistanceCollision (ocp_croco_generic.py:524-533) with a scalar activation
      double Ra[9], pa[3], Rb[9], pb[3], ca[3], cb[3], n[3];
      int ja, jb;
      wg_frame_world<NV>(L, m, rows.frame[r], Ra, pa, &ja);
      wg_frame_world<NV>(L, m, rows.frame_b[r], Rb, pb, &jb);
      const double d = collision_distance_placed(m, rows.frame[r], rows.frame_b[r], Ra, pa, Rb, pb, ca, cb, n);
      double a, ar, arr;
      activation1(rows.act[r], rows.alpha[r], aw[0], d, a, ar, arr);
      if (lane == 0) cost += wi * a;
      if (DIFF) {
        const bool ona = (ja >= 0) && ((L.anc[ja >= 0 ? ja : 0] >> j) & 1u);
        const bool onb = (jb >= 0) && ((L.anc[jb >= 0 ? jb : 0] >> j) & 1u);
        const double *Sj = L.S[j], *pj = L.w.c.pw[j];
        double da[3], db[3], ta[3], tb[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) { da[e] = ca[e] - pj[e]; db[e] = cb[e] - pj[e]; }
        cross3(Sj + 3, da, ta);
        cross3(Sj + 3, db, tb);
        const double g = (ona ? dot3(n, ta) : 0.0) - (onb ? dot3(n, tb) : 0.0);
        Lq += wi * ar * g;
        if (jl) L.w.c.J[nJ][j] = g;
        if (lane == 0) L.w.c.wJ[nJ] = wi * arr;
      }
      nJ += 1;
    }
  }
  if (jl) {
    L.cpart[j] = cost;
    if (DIFF) { L.Lq[j] = sc * Lq; L.Lv[j] = sc * Lv; L.Lvv[j] = sc * Lvv; L.Luu[j] = sc * Luu; L.lu[j] = sc * Lu; L.dqq[j] = dqq; L.D[j] = sc * Luu + in.preg; }
  }
}

// DIFF = true: QP tile + aux tile (K1).  DIFF = false: returns cost + mu_dyn |gap|_1 on every thread (line search).
template <int NV, bool TERM, bool DIFF>
__device__ __forceinline__ double wg_node(WgNode<NV> &L, const DevModel &m, const DevRows &rows, const WgIn &in, double *__restrict__ qt,
                                          double *__restrict__ ax) {
  constexpr int NX = 2 * NV, NT = 256, LDM = 32;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  static_assert(Q::LD == LDM && A::LD == LDM, "tile row stride of large models");
  static_assert(sizeof(WgNode<NV>) <= (NV <= 30 ? 54608 : 65536), "three workgroups per CU (160 KiB of LDS) up to 30 joints, two above");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti = wave >> 1, tj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const double dt = in.dt, sc = TERM ? 1.0 : dt;
  const agx_v4d zero4 = {0.0, 0.0, 0.0, 0.0};

  AGX_WG_STAMP();
  // ---- phase 0: inputs and model index sets into LDS, accumulators cleared
  for (int e = tid; e < NX; e += NT) {
    L.x[e] = in.x[e] + (in.dx ? in.alpha * in.dx[e] : 0.0);
    if (!TERM) L.xn[e] = in.xn[e] + (in.dxn ? in.alpha * in.dxn[e] : 0.0);
  }
  if (!TERM)
    for (int e = tid; e < NV; e += NT) L.u[e] = in.u[e] + (in.du ? in.alpha * in.du[e] : 0.0);
  for (int e = tid; e < in.stride; e += NT) L.w.c.ref[e] = in.ref[e];
  if (tid < 32) {
    L.anc[tid] = tid < NV ? m.anc[tid] : 0u;
    L.desc[tid] = tid < NV ? m.desc[tid] : 0u;
    L.par[tid] = tid < NV ? m.parent[tid] : -1;
    L.nle[tid] = 0.0; L.rhs[tid] = 0.0; L.qdd[tid] = 0.0; L.D[tid] = 0.0; L.lu[tid] = 0.0; L.Lq[tid] = 0.0; L.Lv[tid] = 0.0;
    L.Lvv[tid] = 0.0; L.Luu[tid] = 0.0; L.dqq[tid] = 0.0; L.cpart[tid] = 0.0; L.fq[tid] = 0.0; L.fv[tid] = 0.0;
  }
  if (DIFF) {
    for (int e = tid; e < NV * LDM; e += NT) { (&L.M[0][0])[e] = 0.0; (&L.tq[0][0])[e] = 0.0; (&L.tv[0][0])[e] = 0.0; }
    for (int e = tid; e < kWgJ * LDM; e += NT) (&L.w.c.J[0][0])[e] = 0.0;
    if (tid < kWgJ) L.w.c.wJ[tid] = 0.0;
  }
  int nJ_total = 0;  // dense residual rows of the node (uniform: a function of the row table)
  for (int r = 0; r < rows.n; ++r) {
    if (!rows.active[r]) continue;
    const int kind = rows.kind[r];
    nJ_total += kind == AGX_RES_FRAME_PLACEMENT ? 6 : ((kind == AGX_RES_FRAME_TRANSLATION || kind == AGX_RES_FRAME_ROTATION) ? 3 : (kind == AGX_RES_COLLISION ? 1 : 0));
  }
  __syncthreads();
  AGX_WG_STAMP();

  // ---- kinematics: local placements ...
  if (tid < NV) {
    const int i = tid;
    const double *ax3 = m.axis[i];
    double s, c;
    sincos(L.x[i], &s, &c);
    const double omc = 1.0 - c;
    double Rq[9], Rl[9];
    Rq[0] = c + omc * ax3[0] * ax3[0];
    Rq[1] = omc * ax3[0] * ax3[1] - s * ax3[2];
    Rq[2] = omc * ax3[0] * ax3[2] + s * ax3[1];
    Rq[3] = omc * ax3[1] * ax3[0] + s * ax3[2];
    Rq[4] = c + omc * ax3[1] * ax3[1];
    Rq[5] = omc * ax3[1] * ax3[2] - s * ax3[0];
    Rq[6] = omc * ax3[2] * ax3[0] - s * ax3[1];
    Rq[7] = omc * ax3[2] * ax3[1] + s * ax3[0];
    Rq[8] = c + omc * ax3[2] * ax3[2];
    mm3(m.placement[i], Rq, Rl);
#pragma unroll
    for (int e = 0; e < 9; ++e) L.w.c.Rl[i][e] = Rl[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) L.w.c.Rl[i][9 + e] = m.placement[i][9 + e];
  }
  __syncthreads();
  AGX_WG_STAMP();
  // ... then every (joint, column) composes its own path to the root: thread 4 i + c carries column c of the
  // rotation (c < 3) or the translation (c = 3) of joint i:  y <- R_j y (+ p_j) for j = parent, grandparent, ...
  if (tid < 4 * NV) {
    const int i = tid >> 2, c = tid & 3;
    double y0, y1, y2;
    if (c < 3) { y0 = L.w.c.Rl[i][c]; y1 = L.w.c.Rl[i][3 + c]; y2 = L.w.c.Rl[i][6 + c]; }
    else { y0 = L.w.c.Rl[i][9]; y1 = L.w.c.Rl[i][10]; y2 = L.w.c.Rl[i][11]; }
    for (int j = L.par[i]; j >= 0; j = L.par[j]) {
      const double *Tj = L.w.c.Rl[j];
      const double z0 = Tj[0] * y0 + Tj[1] * y1 + Tj[2] * y2, z1 = Tj[3] * y0 + Tj[4] * y1 + Tj[5] * y2, z2 = Tj[6] * y0 + Tj[7] * y1 + Tj[8] * y2;
      y0 = z0 + (c == 3 ? Tj[9] : 0.0); y1 = z1 + (c == 3 ? Tj[10] : 0.0); y2 = z2 + (c == 3 ? Tj[11] : 0.0);
    }
    if (c < 3) { L.w.c.Rw[i][c] = y0; L.w.c.Rw[i][3 + c] = y1; L.w.c.Rw[i][6 + c] = y2; }
    else { L.w.c.pw[i][0] = y0; L.w.c.pw[i][1] = y1; L.w.c.pw[i][2] = y2; }
  }
  __syncthreads();
  AGX_WG_STAMP();
  if (tid < NV) {
    const int i = tid;
    const double *R = L.w.c.Rw[i], *p = L.w.c.pw[i];
    double z[3], S[6];
    mv3(R, m.axis[i], z);
    cross3(p, z, S);
    S[3] = z[0]; S[4] = z[1]; S[5] = z[2];
#pragma unroll
    for (int e = 0; e < 6; ++e) L.S[i][e] = S[e];
    if (!TERM) {
      // body inertia about the world origin: {m, m c, Ixx, Ixy, Ixz, Iyy, Iyz, Izz}
      double cw[3], Tm[9], Iw[9];
      mv3(R, m.com[i], cw);
      cw[0] += p[0]; cw[1] += p[1]; cw[2] += p[2];
      const double ms = m.mass[i];
      mm3(R, m.inertia[i], Tm);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2) Iw[3 * a + c2] = Tm[3 * a] * R[3 * c2] + Tm[3 * a + 1] * R[3 * c2 + 1] + Tm[3 * a + 2] * R[3 * c2 + 2];
      const double cc = dot3(cw, cw);
      double *I = L.Ib[i];
      I[0] = ms;
      I[1] = ms * cw[0]; I[2] = ms * cw[1]; I[3] = ms * cw[2];
      I[4] = Iw[0] + ms * (cc - cw[0] * cw[0]);
      I[5] = 0.5 * (Iw[1] + Iw[3]) - ms * cw[0] * cw[1];
      I[6] = 0.5 * (Iw[2] + Iw[6]) - ms * cw[0] * cw[2];
      I[7] = Iw[4] + ms * (cc - cw[1] * cw[1]);
      I[8] = 0.5 * (Iw[5] + Iw[7]) - ms * cw[1] * cw[2];
      I[9] = Iw[8] + ms * (cc - cw[2] * cw[2]);
    }
  }
  __syncthreads();
  AGX_WG_STAMP();
  if (in.kin_only) return 0.0;  // constraint evaluation (k_con_eval_wg): world placements and joint axes are in LDS

  if (TERM) {
    if (wave == 1) wg_costs<NV, TERM, DIFF>(L, m, rows, in, lane, sc);
    __syncthreads();
  } else {
    // ---- bias forces and joint-space inertia; sums along the tree on the matrix cores (tiles (0, 0) and (1, 0): 6 columns)
    if (tj == 0) {  // v_i = sum over the path root..i of S_k qd_k
      const agx_v4d acc = wg_mask_mma(L.anc, ti, 0, lane, [&](int k, int n) { return (k < NV && n < 6) ? L.S[k][n] * L.x[NV + k] : 0.0; });
      wg_scatter(acc, ti, 0, lane, [&](int r, int n, double val) { if (r < NV && n < 6) L.v[r][n] = val; });
    }
    __syncthreads();
    AGX_WG_STAMP();
    if (tid < NV) {
      double Sd[6];
      mcross(L.v[tid], L.S[tid], Sd);
#pragma unroll
      for (int e = 0; e < 6; ++e) L.Sd[tid][e] = Sd[e];
    }
    __syncthreads();
    AGX_WG_STAMP();
    if (tj == 0) {  // bias acceleration (qdd = 0), gravity as a base acceleration
      const agx_v4d acc = wg_mask_mma(L.anc, ti, 0, lane, [&](int k, int n) { return (k < NV && n < 6) ? L.Sd[k][n] * L.x[NV + k] : 0.0; });
      wg_scatter(acc, ti, 0, lane, [&](int r, int n, double val) { if (r < NV && n < 6) L.w.c.d1.a[r][n] = val - (n < 3 ? m.gravity[n] : 0.0); });
    }
    __syncthreads();
    AGX_WG_STAMP();
    if (tid < NV) {
      double h6[6], g6[6], x6[6];
      iapply(L.Ib[tid], L.v[tid], h6);
      iapply(L.Ib[tid], L.w.c.d1.a[tid], g6);
      fcross(L.v[tid], h6, x6);
#pragma unroll
      for (int e = 0; e < 6; ++e) L.w.c.d1.f[tid][e] = g6[e] + x6[e];
    }
    __syncthreads();
    AGX_WG_STAMP();
    if (tj == 0) {  // subtree sums: composite force (columns 0..5) and composite inertia (6..15)
      const agx_v4d acc = wg_mask_mma(L.desc, ti, 0, lane, [&](int k, int n) { return k < NV ? (n < 6 ? L.w.c.d1.f[k][n] : L.Ib[k][n - 6]) : 0.0; });
      wg_scatter(acc, ti, 0, lane, [&](int r, int n, double val) {
        if (r < NV) { if (n < 6) L.w.c.d1.fc[r][n] = val; else L.Ic[r][n - 6] = val; }
      });
    }
    __syncthreads();
    AGX_WG_STAMP();
    if (tid < NV) {
      double m6[6];
      iapply(L.Ic[tid], L.S[tid], m6);
#pragma unroll
      for (int e = 0; e < 6; ++e) L.m6[tid][e] = m6[e];
      const double nle = dot6(L.S[tid], L.w.c.d1.fc[tid]);
      L.nle[tid] = nle;
      L.rhs[tid] = L.u[tid] - nle;
    }
    __syncthreads();
    AGX_WG_STAMP();
    {
      // CRBA on the matrix cores: G = m6 S' (G[r][c] = (Ic_r S_r) . S_c) and its transpose;
      // M[r][c] = G[r][c] for c on the path to r, G[c][r] for r on the path to c, 0 on different branches
      const agx_v4d g1 = wg_mma<2>(ti, tj, lane, [&](int r, int k) { return (r < NV && k < 6) ? L.m6[r][k] : 0.0; },
                                   [&](int k, int c) { return (c < NV && k < 6) ? L.S[c][k] : 0.0; }, zero4);
      const agx_v4d g2 = wg_mma<2>(ti, tj, lane, [&](int r, int k) { return (r < NV && k < 6) ? L.S[r][k] : 0.0; },
                                   [&](int k, int c) { return (c < NV && k < 6) ? L.m6[c][k] : 0.0; }, zero4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * ti + l4 + 4 * q, c = 16 * tj + l15;
        if (r < NV && c < NV) {
          double val = ((L.anc[r] >> c) & 1u) ? g1[q] : (((L.anc[c] >> r) & 1u) ? g2[q] : 0.0);
          if (r == c) val += m.armature[r];
          L.M[r][c] = val;
        }
      }
    }
    __syncthreads();
    AGX_WG_STAMP();
    // ---- wave 0 solves M qdd = u - nle, wave 1 evaluates the cost rows meanwhile
    if (wave == 0) {
      const double qdd = wave_spd_solve<NV>(L.M, L.rhs, lane);
      if (lane < NV) {
        L.qdd[lane] = qdd;
        const double qj = L.x[lane], vj = L.x[NV + lane];
        L.fq[lane] = qj + dt * vj + dt * dt * qdd - L.xn[lane];
        L.fv[lane] = vj + dt * qdd - L.xn[NV + lane];
      }
    } else if (wave == 1) {
      wg_costs<NV, TERM, DIFF>(L, m, rows, in, lane, sc);
    }
    __syncthreads();
  }
  AGX_WG_STAMP();
  // node cost: fixed summation order
  double cost_tot = 0.0;
  for (int e = 0; e < NV; ++e) cost_tot += L.cpart[e];
  cost_tot *= sc;
  if (!DIFF) {
    double g = 0.0;
    if (!TERM)
      for (int e = 0; e < NV; ++e) g += fabs(L.fq[e]) + fabs(L.fv[e]);
    return cost_tot + in.mu_dyn * g;
  }

  // Lqq = J' W J + diag(state weights) on the matrix cores (J of the frame / collision rows)
  agx_v4d acc_lqq = zero4;
  for (int ks = 0; 4 * ks < nJ_total; ++ks) {
    const int k = 4 * ks + l4;
    acc_lqq = __builtin_amdgcn_mfma_f64_16x16x4f64(L.w.c.J[k][16 * ti + l15], L.w.c.wJ[k] * L.w.c.J[k][16 * tj + l15], acc_lqq, 0, 0, 0);
  }
  if (ti == tj) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (l4 + 4 * r == l15) acc_lqq[r] += L.dqq[16 * ti + l15];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc_lqq[r] *= sc;

  if (!TERM) {
    // ---- RNEA derivatives (derivation: agx_device.hpp, rnea_derivatives)
    if (tj == 0) {  // accelerations with qdd
      const agx_v4d acc = wg_mask_mma(L.anc, ti, 0, lane, [&](int k, int n) { return (k < NV && n < 6) ? L.S[k][n] * L.qdd[k] + L.Sd[k][n] * L.x[NV + k] : 0.0; });
      wg_scatter(acc, ti, 0, lane, [&](int r, int n, double val) { if (r < NV && n < 6) L.w.d3.a[r][n] = val - (n < 3 ? m.gravity[n] : 0.0); });
    }
    __syncthreads();
    AGX_WG_STAMP();
    if (tid < NV) {
      const int i = tid;
      const double *vi = L.v[i], *Si = L.S[i], *Sdi = L.Sd[i], *ai = L.w.d3.a[i], *I = L.Ib[i];
      double t1[6], t2[6], h6[6];
      mcross(ai, Si, t1);
      mcross(vi, Sdi, t2);
#pragma unroll
      for (int e = 0; e < 6; ++e) L.w.d3.psi[i][e] = t1[e] + t2[e];
      double gg[6], xx[6];
      iapply(I, vi, h6);
      iapply(I, ai, gg);
      fcross(vi, h6, xx);
      double *pre = L.w.d3.pre[i];
#pragma unroll
      for (int e = 0; e < 6; ++e) pre[e] = gg[e] + xx[e];
      pre[6] = h6[0]; pre[7] = h6[1]; pre[8] = h6[2];
      // E = W Io + (W Io)' - v hh' - hh v' + 2 (v.hh) 1 - [n0]x
      const double *vl = vi, *w = vi + 3, *hh = I + 1;
      const double Io[9] = {I[4], I[5], I[6], I[5], I[7], I[8], I[6], I[8], I[9]};
      double WI[9], E[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        WI[0 + c] = w[1] * Io[6 + c] - w[2] * Io[3 + c];
        WI[3 + c] = w[2] * Io[0 + c] - w[0] * Io[6 + c];
        WI[6 + c] = w[0] * Io[3 + c] - w[1] * Io[0 + c];
      }
      const double vh = dot3(vl, hh);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) E[3 * r + c] = WI[3 * r + c] + WI[3 * c + r] - vl[r] * hh[c] - hh[r] * vl[c] + (r == c ? 2.0 * vh : 0.0);
      const double *n0 = h6 + 3;
      E[1] += n0[2]; E[2] -= n0[1];
      E[3] -= n0[2]; E[5] += n0[0];
      E[6] += n0[1]; E[7] -= n0[0];
#pragma unroll
      for (int e = 0; e < 9; ++e) pre[9 + e] = E[e];
    }
    __syncthreads();
    AGX_WG_STAMP();
    {  // subtree sums of fC | f0 | E: 18 columns, all four tiles
      const agx_v4d acc = wg_mask_mma(L.desc, ti, tj, lane, [&](int k, int n) { return (k < NV && n < 18) ? L.w.d3.pre[k][n] : 0.0; });
      wg_scatter(acc, ti, tj, lane, [&](int r, int n, double val) { if (r < NV && n < 18) L.w.d3.cmp[r][n] = val; });
    }
    __syncthreads();
    AGX_WG_STAMP();
    if (tid < NV) {
      const int i = tid;
      const double *Si = L.S[i], *Sdi = L.Sd[i], *psi = L.w.d3.psi[i], *Ic = L.Ic[i];
      const double *fC = L.w.d3.cmp[i], *f0C = fC + 6, *EC = fC + 9;
      double tt[3], Dt[3];
      cross3(f0C, Si, tt);
      const double *sa = Si + 3;
      Dt[0] = 2.0 * tt[0] + EC[0] * sa[0] + EC[3] * sa[1] + EC[6] * sa[2];
      Dt[1] = 2.0 * tt[1] + EC[1] * sa[0] + EC[4] * sa[1] + EC[7] * sa[2];
      Dt[2] = 2.0 * tt[2] + EC[2] * sa[0] + EC[5] * sa[1] + EC[8] * sa[2];
      double IcSd[6], IcPs[6], sxf[6], u1[3], u2[3], e1[3], e2[3];
      iapply(Ic, Sdi, IcSd);
      iapply(Ic, psi, IcPs);
      fcross(Si, fC, sxf);
      cross3(f0C, Si + 3, u1);
      cross3(f0C, Sdi + 3, u2);
      mv3(EC, Si + 3, e1);
      mv3(EC, Sdi + 3, e2);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        L.w.d3.col.Dt[i][e] = Dt[e];
        L.w.d3.col.colv[i][e] = 2.0 * IcSd[e] - 2.0 * u1[e];
        L.w.d3.col.colv[i][3 + e] = 2.0 * IcSd[3 + e] + e1[e];
        L.w.d3.col.colq[i][e] = sxf[e] - 2.0 * u2[e] + IcPs[e];
        L.w.d3.col.colq[i][3 + e] = sxf[3 + e] + e2[e] + IcPs[3 + e];
      }
    }
    __syncthreads();
    AGX_WG_STAMP();
    {
      // dtau/dq, dtau/dqdot on the matrix cores.  Entry (r, c):
      //   c on the path root..r:      dv = [m6_r | Dt_r] . [2 Sd_c | S_c,ang],  dq = [m6_r | Dt_r] . [psi_c | Sd_c,ang]   (k = 9)
      //   r a strict ancestor of c:   dv = S_r . colv_c,                        dq = S_r . colq_c                        (k = 6)
      auto fa1 = [&](int r, int k) { return r < NV ? (k < 6 ? L.m6[r][k] : (k < 9 ? L.w.d3.col.Dt[r][k - 6] : 0.0)) : 0.0; };
      auto fa2 = [&](int r, int k) { return (r < NV && k < 6) ? L.S[r][k] : 0.0; };
      const agx_v4d v1 = wg_mma<3>(ti, tj, lane, fa1, [&](int k, int c) { return c < NV ? (k < 6 ? 2.0 * L.Sd[c][k] : (k < 9 ? L.S[c][k - 3] : 0.0)) : 0.0; }, zero4);
      const agx_v4d q1 = wg_mma<3>(ti, tj, lane, fa1, [&](int k, int c) { return c < NV ? (k < 6 ? L.w.d3.psi[c][k] : (k < 9 ? L.Sd[c][k - 3] : 0.0)) : 0.0; }, zero4);
      const agx_v4d v2 = wg_mma<2>(ti, tj, lane, fa2, [&](int k, int c) { return (c < NV && k < 6) ? L.w.d3.col.colv[c][k] : 0.0; }, zero4);
      const agx_v4d q2 = wg_mma<2>(ti, tj, lane, fa2, [&](int k, int c) { return (c < NV && k < 6) ? L.w.d3.col.colq[c][k] : 0.0; }, zero4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * ti + l4 + 4 * q, c = 16 * tj + l15;
        if (r < NV && c < NV) {
          const bool path = (L.anc[r] >> c) & 1u, above = (L.anc[c] >> r) & 1u;
          L.tv[r][c] = path ? v1[q] : (above ? v2[q] : 0.0);
          L.tq[r][c] = path ? q1[q] : (above ? q2[q] : 0.0);
        }
      }
    }
    __syncthreads();
    AGX_WG_STAMP();
  }

  // ---- acceleration-input transformation on the matrix cores and the stores
  agx_v4d hww = zero4, hqw = zero4, hvw = zero4, hqq = acc_lqq, hqv = zero4, hvv = zero4;
  if (ti == tj) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (l4 + 4 * r == l15) hvv[r] = L.Lvv[16 * ti + l15];
  }
  wg_store_tile(ax + A::Lqq, acc_lqq, ti, tj, lane, NV);
  if (!TERM) {
    const int ca = 16 * ti + l15, cb = 16 * tj + l15;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k = 4 * ks + l4;
      const double d = L.D[k];
      const double aM = wg_op<NV>(L.M, k, ca), aq = wg_op<NV>(L.tq, k, ca), av = wg_op<NV>(L.tv, k, ca);
      const double bM = d * wg_op<NV>(L.M, k, cb), bq = d * wg_op<NV>(L.tq, k, cb), bv = d * wg_op<NV>(L.tv, k, cb);
      hww = __builtin_amdgcn_mfma_f64_16x16x4f64(aM, bM, hww, 0, 0, 0);
      hqw = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bM, hqw, 0, 0, 0);
      hvw = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bM, hvw, 0, 0, 0);
      hqq = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bq, hqq, 0, 0, 0);
      hqv = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bv, hqv, 0, 0, 0);
      hvv = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, hvv, 0, 0, 0);
    }
  }
  AGX_WG_STAMP();
  wg_store_tile(qt + Q::Hww, hww, ti, tj, lane, NV);
  wg_store_tile(qt + Q::Hqw, hqw, ti, tj, lane, NV);
  wg_store_tile(qt + Q::Hvw, hvw, ti, tj, lane, NV);
  wg_store_tile(qt + Q::Hqq, hqq, ti, tj, lane, NV);
  wg_store_tile(qt + Q::Hqv, hqv, ti, tj, lane, NV);
  wg_store_tile(qt + Q::Hvv, hvv, ti, tj, lane, NV);
  // aux blocks M | tq | tv straight from LDS (zeros at the terminal node), whole rows
  for (int e = 2 * tid; e < NV * LDM; e += 2 * NT) {
    *reinterpret_cast<double2 *>(ax + A::M + e) = *reinterpret_cast<const double2 *>(&L.M[0][0] + e);
    *reinterpret_cast<double2 *>(ax + A::tq + e) = *reinterpret_cast<const double2 *>(&L.tq[0][0] + e);
    *reinterpret_cast<double2 *>(ax + A::tv + e) = *reinterpret_cast<const double2 *>(&L.tv[0][0] + e);
  }
  // gradients: gw = M lu, gx = Lx + taux' lu
  if (tid < 3 * NV) {
    const int which = tid / NV, i = tid % NV;
    double s = which == 0 ? 0.0 : (which == 1 ? L.Lq[i] : L.Lv[i]);
    if (!TERM) {
      if (which == 0) for (int l = 0; l < NV; ++l) s += L.M[i][l] * L.lu[l];
      else if (which == 1) for (int l = 0; l < NV; ++l) s += L.tq[l][i] * L.lu[l];
      else for (int l = 0; l < NV; ++l) s += L.tv[l][i] * L.lu[l];
    }
    qt[(which == 0 ? Q::gw : (which == 1 ? Q::gx : Q::gx + NV)) + i] = s;
  } else if (tid >= 128 && tid < 128 + NV) {
    const int i = tid - 128;
    qt[Q::f + i] = L.fq[i];
    qt[Q::f + NV + i] = L.fv[i];
    ax[A::Lvv + i] = L.Lvv[i];
    ax[A::Luu + i] = L.Luu[i];
    ax[A::Lu + i] = L.lu[i];
  } else if (tid == 255) {
    qt[Q::cost] = cost_tot;
  }
  AGX_WG_STAMP();
  return cost_tot;
}

// K1 for large models: running nodes first (unit = b T + t), then the terminal nodes, one workgroup each.
// Measured and discarded in round 3 (B = 512, T = 50, same box): the node's inputs loaded ahead of the status check (the input
// phase is 6 us of the 34 us a workgroup spends on a node, AGX_WG_PROFILE stamps) -- 1.22 ms either way: with three
// workgroups per CU that latency is covered by the other two; and a persistent variant (768 workgroups walking the units,
// the next unit's inputs loaded a node ahead) -- 2.7 ms: the loop state pushed the scratch of the 168-register build from
// 268 to 932 B per lane.
template <int NV>
__global__ void __launch_bounds__(256, AGX_WG_MINWAVES) k_calc_qp_wg(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                    const double *__restrict__ dts, const double *__restrict__ xs,
                                                    const double *__restrict__ us, RefView rv, double *__restrict__ qts,
                                                    double *__restrict__ auxs, const DevState *__restrict__ st, int phase) {
  constexpr int NX = 2 * NV;
  __shared__ WgNode<NV> L;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long unit = blockIdx.x, n_run = (long long)o.B * T;
  const bool term = unit >= n_run;
  const int b = term ? (int)(unit - n_run) : (int)(unit / T), t = term ? T : (int)(unit % T);
  if (!k1_active(st[b], phase)) return;  // phase 1: the trial points of the instances in the line search (k_sqp_head / k_sqp_accept)
  const long long node = (long long)b * (T + 1) + t;
  WgIn in;
  in.x = xs + node * NX; in.dx = nullptr; in.xn = in.x + NX; in.dxn = nullptr;
  in.u = us + ((long long)b * T + t) * NV; in.du = nullptr;
  in.alpha = 0.0; in.preg = k1_preg(st[b], phase); in.mu_dyn = o.mu_dyn;
  in.ref = ref_at(rv, b, t, T); in.stride = o.stride; in.frames = frames_at(rv, b, t, T);
  double *qt = qts + node * QT<NV>::SIZE, *ax = auxs + node * AUX<NV>::SIZE;
  if (term) { in.dt = 0.0; wg_node<NV, true, true>(L, *mp, o.rows[1], in, qt, ax); }
  else { in.dt = dts[t]; wg_node<NV, false, true>(L, *mp, o.rows[0], in, qt, ax); }
}

// Constraint values, Jacobians and the l1 violation of every node for large models (k_con_eval of agx_admm.hpp): control-limit
// rows (g = u - ref, identity Jacobian on u), state bounds (g = x - ref, identity on x), collision-distance rows (colmpc.ResidualDistanceCollision: g = d(q), Jacobian
// row on q as in wg_costs) and frame translation / rotation / placement rows (3 / 3 / 6 Jacobian rows on q).  One workgroup per node: the kinematics of wg_node, then wave 0 evaluates the rows, lane j its
// column.  cg [B][T+1][AGX_MAX_NC], cjac [B][T+1][AGX_MAX_DENSE][32] (d / dq only: no supported row depends on v; u rows are I).
template <int NV>
__global__ void __launch_bounds__(256, AGX_WG_MINWAVES) k_con_eval_wg(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                                      const double *__restrict__ xs, const double *__restrict__ us,
                                                                      double *__restrict__ cg, double *__restrict__ cjac,
                                                                      double *__restrict__ nodestat, const DevState *__restrict__ st,
                                                                      int phase) {
  constexpr int NX = 2 * NV;
  __shared__ WgNode<NV> L;
  __shared__ DevRows none;
  const DevOcp &o = *op;
  const DevModel &m = *mp;
  const int T = o.T, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long node = blockIdx.x;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  if (!k1_active(st[b], phase)) return;
  const DevCons &c = o.cons[t == T ? 1 : 0];
  if (threadIdx.x == 0) none.n = 0;
  if (c.ncoll > 0) {
    WgIn in;
    in.x = xs + node * NX; in.dx = nullptr; in.xn = in.x; in.dxn = nullptr;
    in.u = us + ((long long)b * T + (t < T ? t : T - 1)) * NV; in.du = nullptr; in.alpha = 0.0; in.preg = 0.0; in.mu_dyn = 0.0; in.dt = 0.0;
    in.ref = nullptr; in.stride = 0; in.frames = nullptr; in.kin_only = true;
    __syncthreads();
    wg_node<NV, true, false>(L, m, none, in, nullptr, nullptr);  // TERM: no successor state / control is read
  }
  if (wave != 0) return;
  const int j = lane < NV ? lane : NV - 1;
  const bool jl = lane < NV;
  double v = 0.0;
  for (int r = 0; r < c.n; ++r) {
    const int off = c.off[r];
    if (c.kind[r] == AGX_RES_CONTROL) {
      const double g = (t < T) ? us[((long long)b * T + t) * NV + j] - c.ref[r][j] : 0.0;
      if (jl) {
        cg[node * AGX_MAX_NC + off + j] = g;
        v += fmax(c.lb[off + j] - g, 0.0) + fmax(g - c.ub[off + j], 0.0);
      }
    } else if (c.kind[r] == AGX_RES_STATE) {
      const double gq = xs[node * NX + j] - c.ref[r][j], gv = xs[node * NX + NV + j] - c.ref[r][NV + j];
      if (jl) {
        cg[node * AGX_MAX_NC + off + j] = gq;
        cg[node * AGX_MAX_NC + off + NV + j] = gv;
        v += fmax(c.lb[off + j] - gq, 0.0) + fmax(gq - c.ub[off + j], 0.0) + fmax(c.lb[off + NV + j] - gv, 0.0) + fmax(gv - c.ub[off + NV + j], 0.0);
      }
    } else if (c.kind[r] == AGX_RES_COLLISION) {
      double Ra[9], pa[3], Rb[9], pb[3], ca[3], cb[3], n[3];
      int ja, jb;
      wg_frame_world<NV>(L, m, c.frame[r], Ra, pa, &ja);
      wg_frame_world<NV>(L, m, c.frame_b[r], Rb, pb, &jb);
      const double d = collision_distance_placed(m, c.frame[r], c.frame_b[r], Ra, pa, Rb, pb, ca, cb, n);
      const bool ona = (ja >= 0) && ((L.anc[ja >= 0 ? ja : 0] >> j) & 1u);
      const bool onb = (jb >= 0) && ((L.anc[jb >= 0 ? jb : 0] >> j) & 1u);
      const double *Sj = L.S[j], *pj = L.w.c.pw[j];
      double da[3], db[3], ta[3], tb[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) { da[e] = ca[e] - pj[e]; db[e] = cb[e] - pj[e]; }
      cross3(Sj + 3, da, ta);
      cross3(Sj + 3, db, tb);
      const double gq = (ona ? dot3(n, ta) : 0.0) - (onb ? dot3(n, tb) : 0.0);
      if (lane < 32) cjac[(node * AGX_MAX_DENSE + c.coll_slot[r]) * 32 + lane] = jl ? gq : 0.0;
      if (lane == 0) {
        cg[node * AGX_MAX_NC + off] = d;
        v += fmax(c.lb[off] - d, 0.0) + fmax(d - c.ub[off], 0.0);
      }
    } else if (c.kind[r] == AGX_RES_FRAME_TRANSLATION || c.kind[r] == AGX_RES_FRAME_ROTATION || c.kind[r] == AGX_RES_FRAME_PLACEMENT) {
      // the residuals of the cost rows as constraints (constraints_eval, agx_device.hpp): translation p(q) - pref with the
      // LOCAL_WORLD_ALIGNED linear frame Jacobian, rotation log3(Rref' R) with Jlog3 x LOCAL angular Jacobian, placement
      // log6(Mref^-1 M) with Jlog6 x LOCAL Jacobian; lane j: column j of the nr Jacobian rows
      const int kind = c.kind[r], nr = c.nr[r];
      double RF[9], pF[3];
      int jf;
      wg_frame_world<NV>(L, m, c.frame[r], RF, pF, &jf);
      const double *rr = c.ref[r];
      double res[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, TL[9], TR[9];
      if (kind == AGX_RES_FRAME_TRANSLATION) {
#pragma unroll
        for (int e = 0; e < 3; ++e) res[e] = pF[e] - rr[e];
      } else if (kind == AGX_RES_FRAME_ROTATION) {
        double Rrel[9];
        mtm3(rr, RF, Rrel);
        log3(Rrel, res);
        jlog3(res, TL);
      } else {
        double Rrel[9], d3[3], prel[3];
        mtm3(rr, RF, Rrel);
        d3[0] = pF[0] - rr[9]; d3[1] = pF[1] - rr[10]; d3[2] = pF[2] - rr[11];
        mtv3(rr, d3, prel);
        log6<true>(Rrel, prel, res, TL, TR);
      }
      const bool on = (jf >= 0) && ((L.anc[jf >= 0 ? jf : 0] >> j) & 1u);
      const double *Sj = L.S[j], *pj = L.w.c.pw[j];
      double d3[3], tz[3], lin[3], ang[3], out[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int e = 0; e < 3; ++e) d3[e] = pF[e] - pj[e];
      cross3(Sj + 3, d3, tz);  // z x (pF - pj)
      if (kind == AGX_RES_FRAME_TRANSLATION) {
#pragma unroll
        for (int e = 0; e < 3; ++e) out[e] = tz[e];
      } else {
        mtv3(RF, tz, lin);
        mtv3(RF, Sj + 3, ang);
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const double bot = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
          if (kind == AGX_RES_FRAME_ROTATION) out[e] = bot;
          else {
            out[e] = TL[3 * e] * lin[0] + TL[3 * e + 1] * lin[1] + TL[3 * e + 2] * lin[2] + TR[3 * e] * ang[0] + TR[3 * e + 1] * ang[1] + TR[3 * e + 2] * ang[2];
            out[3 + e] = bot;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 6; ++e)
        if (e < nr) {
          if (lane < 32) cjac[(node * AGX_MAX_DENSE + c.coll_slot[r] + e) * 32 + lane] = (jl && on) ? out[e] : 0.0;
          if (lane == 0) {
            cg[node * AGX_MAX_NC + off + e] = res[e];
            v += fmax(c.lb[off + e] - res[e], 0.0) + fmax(res[e] - c.ub[off + e], 0.0);
          }
        }
    }
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) v += __shfl_xor(v, sft, 64);
  if (lane == 0) nodestat[node * 4 + 3] = v;
}

// One semi-implicit Euler step (OCPBaseCroco.integrate, ocp_base_croco.py:184-189): forward dynamics only, one
// workgroup per state, no cost rows.  tlist == null: n packed states x [n][nx], u [n][nu] -> xnext [n][nx].
// tlist != null (warm-start shift, warm_start_shift_previous_solution.py:98-104: nodes with dt_i > dt_0 are
// integrated over dt_0): workgroup (b, k) takes node t = tlist[k] of the horizon buffers and writes the staging copy.
template <int NV>
__global__ void __launch_bounds__(256) k_integrate_wg(const DevModel *__restrict__ mp, double dt, const double *__restrict__ x,
                                                      const double *__restrict__ u, double *__restrict__ xnext,
                                                      const int *__restrict__ tlist, int nlist, int T) {
  constexpr int NX = 2 * NV;
  __shared__ WgNode<NV> L;
  __shared__ DevRows none;
  long long ix = blockIdx.x, iu = blockIdx.x;
  if (tlist) {
    const int b = blockIdx.x / nlist, t = tlist[blockIdx.x % nlist];
    ix = (long long)b * (T + 1) + t;
    iu = (long long)b * T + t;
  }
  if (threadIdx.x == 0) none.n = 0;
  __syncthreads();
  WgIn in;
  in.x = x + ix * NX; in.dx = nullptr; in.xn = in.x; in.dxn = nullptr;  // "gap" against x itself: fq = dt v + dt^2 a, fv = dt a
  in.u = u + iu * NV; in.du = nullptr; in.alpha = 0.0; in.preg = 0.0; in.mu_dyn = 0.0; in.dt = dt;
  in.ref = nullptr; in.stride = 0; in.frames = nullptr;
  wg_node<NV, false, false>(L, *mp, none, in, nullptr, nullptr);
  if (threadIdx.x < NV) {
    const int j = threadIdx.x;
    xnext[ix * NX + j] = L.x[j] + L.fq[j];
    xnext[ix * NX + NV + j] = L.x[NV + j] + L.fv[j];
  }
}

// Warm-start shift without dynamics (the nodes with dt_i == dt_0; warm_start_shift_previous_solution.py:93-97):
// xs[i] <- xs[i+1], us[i] <- us[i+1] (the last control is kept) into the staging halves of the buffers; nodes with
// dt_i != dt_0 keep their control, their state comes from k_integrate_wg.  k_shift_commit copies the staging back.
__global__ void k_shift_copy(const double *__restrict__ dts, double *__restrict__ xs, double *__restrict__ us, int B, int T, int NX, int NU) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nxs = (long long)B * (T + 1) * NX, nus = (long long)B * T * NU;
  const double dt0 = dts[0];
  if (i < nxs) {
    const int t = (int)((i / NX) % (T + 1));
    if (t < T && dts[t] == dt0) xs[nxs + i] = xs[i + NX];
  }
  if (i < nus) {
    const int t = (int)((i / NU) % T);
    us[nus + i] = (dts[t] == dt0 && t < T - 1) ? us[i + NU] : us[i];
  }
}

}  // namespace agx

namespace agx {

// ---------------------------------------------------------------------------
// K2 for large models, matrix-core version.  One 256-thread workgroup per instance; per node
//   * Qww, Qxw = Qwx', qw, qx from the QP tile and the value function of node t+1: element-wise thanks to the
//     (Phi, G) structure of the acceleration-input QP (as in the register kernel), coalesced tile reads;
//   * Kw = Qww^-1 Qwx, kw = Qww^-1 qw: every wave keeps the rows of [Qww | a quarter of the right-hand sides] in
//     registers, a row per lane, 30 Gauss-Jordan pivots with v_readlane broadcasts: no barrier inside the elimination;
//   * V = Qxx - Qxw Kw  (60 x 30 x 60) on v_mfma_f64_16x16x4_f64, wave w owns row band w of the 4 x 4 tiles;
//     the accumulators start at Qxx, built in the result layout.
// (k_riccati_big above: Gauss-Jordan on the whole 90 x 91 matrix in LDS, two barriers per pivot, kept as AGX_RICCATI_MFMA=0.)
// ---------------------------------------------------------------------------
template <int NV>
__global__ void __launch_bounds__(256) k_riccati_mfma(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                      const double *__restrict__ qts, double *__restrict__ Kws,
                                                      double *__restrict__ kws, double *__restrict__ dxs,
                                                      double *__restrict__ wss, DevState *__restrict__ st, int forward,
                                                      int gains_pass) {
  static_assert(NV > 16 && NV <= 32 && NV % 2 == 0, "tiling below: 2 nv <= 64 result rows, nv / 2 right-hand sides per wave");
  constexpr int NX = 2 * NV, LV = NX + 1, LQ = NV + 1, NR = NV / 2 + 1;  // right-hand sides per wave: NV / 2 columns of Qwx (+ qw on wave 3)
  typedef QT<NV> Q;
  __shared__ double V[NX][LV], Qxw[NX][LQ], Qww[NV][LQ], Kl[NV][LV];
  __shared__ double vx[NX], vp[NX], fl[NX], qx[NX], qw[NV], kl[NV], dxl[NX], wl[NV];
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x, nt = 256, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  DevState &S = st[b];
  if (!gains_pass && (S.done || S.admm_conv)) return;
  // gains_pass selects the instances of the sigma sweep like gmode of riccati_body: 1 everyone (agx_ocp_direction, timing),
  // 2 the fix-up on exit (instances whose last direction has no gains yet), 4 the unfinished instances before the line search of
  // an iteration the loop may end with (current regularisation).  ls_acc tells k_gains_to_u_* which instances were swept.
  if (gains_pass) {
    const bool run = gains_pass == 1 || (gains_pass == 4 && !S.done) || (gains_pass == 2 && S.gains_iter != S.dir_iter);
    __syncthreads();  // everyone has read the state before it is written
    if (threadIdx.x == 0) { S.ls_acc = run ? 1 : 0; if (run && gains_pass != 1) S.gains_iter = S.dir_iter; }
    if (!run) return;
  }
  const double dreg = (gains_pass && gains_pass != 4) ? (S.solved ? S.dreg : S.gains_dreg) : S.dreg;
  const double *qb = qts + (long long)b * (T + 1) * Q::SIZE;
  double *Kw = Kws + (long long)b * T * NV * NX, *kw = kws + (long long)b * T * NV;
  {  // value function of the terminal node
    const double *tt = qb + (long long)T * Q::SIZE;
    for (int e = tid; e < NV * NV; e += nt) {
      const int r = e / NV, c = e % NV;
      const double hqq = tt[Q::Hqq + r * Q::LD + c], hqv = tt[Q::Hqv + r * Q::LD + c], hvv = tt[Q::Hvv + r * Q::LD + c];
      V[r][c] = hqq + (r == c ? dreg : 0.0);
      V[r][NV + c] = hqv;
      V[NV + c][r] = hqv;
      V[NV + r][NV + c] = hvv + (r == c ? dreg : 0.0);
    }
    for (int i = tid; i < NX; i += nt) vx[i] = gains_pass ? 0.0 : tt[Q::gx + i];
  }
  __syncthreads();
  // Tile entries travel global -> registers one node ahead (the sweep is a dependent chain of ~10 us steps: an
  // un-prefetched HBM read per phase would sit on it): this thread's entries of Hww | Hqw | Hvw for the element-wise
  // build (coalesced rows) and of Hqq | Hqv | Hvv in the MFMA result layout for the start value of the accumulators.
  constexpr int NE = (NV * NV + 255) / 256;
  double pw[NE], pq[NE], pv[NE], px[4][4], pf = 0.0, pgw = 0.0, pgq = 0.0, pgv = 0.0;
  // result-layout bookkeeping of this lane: entry (i, j) = (16 wave + l4 + 4 q, 16 tj + l15) of the 2nv x 2nv matrix
  auto fetch = [&](int t) {
    const double *tl = qb + (long long)t * Q::SIZE;
#pragma unroll
    for (int n = 0; n < NE; ++n) {
      const int e = tid + 256 * n, r = e / NV, c = e % NV, rc = r * Q::LD + c;
      const bool in = e < NV * NV;
      pw[n] = in ? tl[Q::Hww + rc] : 0.0; pq[n] = in ? tl[Q::Hqw + rc] : 0.0; pv[n] = in ? tl[Q::Hvw + rc] : 0.0;
    }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * wave + l4 + 4 * q, j = 16 * tj + l15;
        const int bi = i >= NV, bj = j >= NV, r = i - bi * NV, c = j - bj * NV;
        // (q, q): Hqq[r][c]   (q, v): Hqv[r][c]   (v, q): Hqv[c][r]   (v, v): Hvv[r][c]
        const int off = (bi && !bj) ? Q::Hqv + c * Q::LD + r : ((bi ? Q::Hvv : (bj ? Q::Hqv : Q::Hqq)) + r * Q::LD + c);
        px[tj][q] = (i < NX && j < NX) ? tl[off] : 0.0;
      }
    if (!gains_pass) {
      if (tid < NX) pf = tl[Q::f + tid];
      if (tid < NV) { pgw = tl[Q::gw + tid]; pgq = tl[Q::gx + tid]; pgv = tl[Q::gx + NV + tid]; }
    }
  };
  fetch(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    const double h = dts[t], h2 = h * h;
    if (tid < NX) fl[tid] = gains_pass ? 0.0 : pf;
    const double gw_t = pgw, gq_t = pgq, gv_t = pgv;
    __syncthreads();
    if (tid < NX) {  // vp = vx + V f
      double s = vx[tid];
      for (int j = 0; j < NX; ++j) s += V[tid][j] * fl[j];
      vp[tid] = s;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NE; ++n) {
      const int e = tid + 256 * n, r = e / NV, c = e % NV;
      if (e < NV * NV) {
        const double Vqq = V[r][c], Vqv = V[r][NV + c], Vvq = V[NV + r][c], Vvv = V[NV + r][NV + c];
        const double Yq = h2 * Vqq + h * Vvq, Yv = h2 * Vqv + h * Vvv;    // (G' V) blocks, rows = acceleration index
        const double YqT = h2 * Vqq + h * Vqv, YvT = h2 * Vvq + h * Vvv;  // their transposes at [r][c]
        Qww[r][c] = pw[n] + h2 * Yq + h * Yv;
        Qxw[r][c] = pq[n] + YqT;
        Qxw[NV + r][c] = pv[n] + h * YqT + YvT;
      }
    }
    if (tid < NV) {
      const double vpq = vp[tid], vpv = vp[NV + tid];
      qw[tid] = gains_pass ? 0.0 : gw_t + h2 * vpq + h * vpv;
      qx[tid] = gains_pass ? 0.0 : gq_t + vpq;
      qx[NV + tid] = gains_pass ? 0.0 : gv_t + h * vpq + vpv;
    }
    // start value of the accumulators: Qxx = Hxx + Phi' V Phi in the result layout, branch free:
    //   value = H + a0 V[r][c] + a1 V[r][nv + c] + a2 V[nv + r][c] + a3 V[nv + r][nv + c],
    //   (a0..a3) = (1, 0, 0, 0) on (q, q), (h, 1, 0, 0) on (q, v), (h, 0, 1, 0) on (v, q), (h^2, h, h, 1) on (v, v)
    agx_v4d acc[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * wave + l4 + 4 * q, j = 16 * tj + l15;
        const bool in = i < NX && j < NX;
        const int bi = i >= NV, bj = j >= NV, r = in ? i - bi * NV : 0, c = in ? j - bj * NV : 0;
        const double a0 = bi ? (bj ? h2 : h) : (bj ? h : 1.0), a1 = bj ? (bi ? h : 1.0) : 0.0, a2 = bi ? (bj ? h : 1.0) : 0.0, a3 = (bi && bj) ? 1.0 : 0.0;
        const double val = px[tj][q] + a0 * V[r][c] + a1 * V[r][NV + c] + a2 * V[NV + r][c] + a3 * V[NV + r][NV + c];
        acc[tj][q] = in ? val : 0.0;
      }
    }
    if (t > 0) fetch(t - 1);  // next node's tile entries are on their way during the elimination
    __syncthreads();
    {
      // ---- Kw = Qww^-1 Qwx, kw = Qww^-1 qw: lane r = row r of [Qww | right-hand sides of this wave]
      const int r = lane < NV ? lane : NV - 1;
      double a[NV], y[NR];
#pragma unroll
      for (int j = 0; j < NV; ++j) a[j] = Qww[r][j];
      const int c0 = (NV / 2) * wave;  // x columns c0 .. c0 + NV / 2 - 1
#pragma unroll
      for (int j = 0; j < NV / 2; ++j) y[j] = Qxw[c0 + j][r];
      y[NV / 2] = qw[r];  // every wave carries it (only wave 3 stores it)
      double d = 1.0;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const double piv = readlane_f64(a[k], k);
        const double rp = fast_rcp(piv);
        const double f = (lane == k) ? 0.0 : a[k] * rp;
        if (r == k) d = rp;
#pragma unroll
        for (int j = k + 1; j < NV; ++j) a[j] -= f * readlane_f64(a[j], k);
#pragma unroll
        for (int j = 0; j < NR; ++j) y[j] -= f * readlane_f64(y[j], k);
      }
      if (lane < NV) {
#pragma unroll
        for (int j = 0; j < NV / 2; ++j) Kl[r][c0 + j] = y[j] * d;
        if (wave == 3) kl[r] = y[NV / 2] * d;
      }
    }
    __syncthreads();
    // ---- V <- Qxx - Qxw Kw: wave w owns rows 16 w .. 16 w + 15 of the 64 x 64 result
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k = 4 * ks + l4, i = 16 * wave + l15;
      const double av = (k < NV && i < NX) ? -Qxw[i][k] : 0.0;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int j = 16 * tj + l15;
        const double bv = (k < NV && j < NX) ? Kl[k][j] : 0.0;
        acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[tj], 0, 0, 0);
      }
    }
    double vxn = 0.0;
    if (tid < NX) {  // gradient of the value function of node t
      double s = qx[tid];
      for (int k = 0; k < NV; ++k) s -= Qxw[tid][k] * kl[k];
      vxn = s;
    }
    __syncthreads();  // every read of the old V / Qxw is done
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * wave + l4 + 4 * q, j = 16 * tj + l15;
        if (i < NX && j < NX) V[i][j] = acc[tj][q];
      }
    if (tid < NX) vx[tid] = vxn;
    // gains of this node to HBM, whole rows
    for (int e = tid; e < NV * NX; e += nt) Kw[(long long)t * NV * NX + e] = Kl[e / NX][e % NX];
    if (tid < NV) kw[(long long)t * NV + tid] = kl[tid];
    __syncthreads();
    for (int e = tid; e < NX * NX; e += nt) {  // symmetrise in place: one thread per unordered pair
      const int i = e / NX, j = e % NX;
      if (i > j) continue;
      const double sv = 0.5 * (V[i][j] + V[j][i]) + ((i == j) ? dreg : 0.0);
      V[i][j] = sv;
      V[j][i] = sv;
    }
    __syncthreads();
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
