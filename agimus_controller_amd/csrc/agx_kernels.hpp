// agx_kernels.hpp -- gfx950 kernels of the SQP solve path.
//
//   K1  k_calc_qp / k_calc_qp_term       node-parallel calc + calcDiff -> QP tile in acceleration-input form
//       k_calc_diff / k_calc_diff_term   same, canonical Fx|Fu|L* tile (test / introspection entry point)
//   K2  k_riccati                        Riccati backward + linear forward (one wave per instance)
//   K4  k_step                           du, KKT, convergence test, merit line search (one workgroup per instance)
//   exit: k_sigma_tile, k_riccati (gains pass), k_gains_to_u -> the gains the solver reports
//   plus warm-start shift, reference generators and small batch utilities.
//
// Replaces mim_solvers::SolverCSQP::solve as called at
// agimus_controller/agimus_controller/ocp_base_croco.py:172 (unconstrained branch).
#pragma once

#include <type_traits>

#include "agx_device.hpp"

// Per-instance solver state, device resident.
struct DevState {
  double kkt, cost, merit, gap;  // as agx_status
  double preg, dreg;             // crocoddyl regularisation (reg_min 1e-9)
  int iter, qp_iters, solved, flags;
  int done;                      // 1: instance finished (solved, or regularisation saturated)
  int gains_iter;  // SQP iteration whose tiles the reported gains (Kout) were swept from (-1: none)
  int dir_iter;    // SQP iteration of the last direction this instance computed
  double gains_preg, gains_dreg; // regularisation the last direction was computed with
  // constrained problems (ADMM, agx_admm.hpp)
  double rho_sparse;             // persists across solves (SolverCSQP reset_rho = false); 0 = not initialised
  double con;                    // l1 norm of the constraint violation at the last evaluation
  int admm_conv, admm_iter;      // QP converged in this SQP iteration / ADMM iterations done
  int ls_acc;                    // large models: the sigma sweep in front of k_gains_to_u_* took this instance (agx_big.hpp)
  int admm_refactor;             // ADMM: the Hessian part of the augmented tiles changed (first iteration / new rho)
  int dir_fail;                  // the last backward sweep met a non-positive / non-finite pivot (Quu not positive definite)
  // line search by derivative passes at the trial points (nv <= 7: k_sqp_head / k_sqp_accept)
  int searching;                 // the instance is inside the line search of the current SQP iteration
  int ls_n;                      // index of the trial in flight: step length 2^-ls_n
  int tiles_ok;                  // QP / aux tiles (and the constraint data) belong to the current (xs, us): the derivative pass skips it
  int pad_ls;
  double preg_trial;             // control regularisation the NEXT iteration runs with if the trial in flight is accepted (baked into its tiles)
};
// which instances a derivative pass (K1, k_con_eval) works on: phase 0 = start of an SQP iteration (everyone whose tiles are
// stale), phase 1 = the trial points of the instances that are searching (their tiles are overwritten in place)
__device__ __forceinline__ bool k1_active(const DevState &S, int phase) { return phase ? (S.searching != 0) : (!S.done && !S.tiles_ok); }
__device__ __forceinline__ double k1_preg(const DevState &S, int phase) { return phase ? S.preg_trial : S.preg; }

// Addressing of the reference tiles (host tile or a window of the resident trajectory).
struct RefView {
  const double *base;
  long long bstride;  // doubles between instances
  long long tstride;  // doubles between nodes
  long long term_off; // extra offset of the terminal node's tile
  const int *frames;  // [B][T+1][AGX_MAX_ROWS] or null
};
__device__ __forceinline__ const double *ref_at(const RefView &rv, int b, int t, int T) {
  return rv.base + (long long)b * rv.bstride + (long long)t * rv.tstride + (t == T ? rv.term_off : 0);
}
__device__ __forceinline__ const int *frames_at(const RefView &rv, int b, int t, int T) {
  return rv.frames ? rv.frames + ((long long)b * (T + 1) + t) * AGX_MAX_ROWS : nullptr;
}

namespace agx {

constexpr double kSigma = 1e-6;   // SolverCSQP proximal weight
constexpr double kRegMin = 1e-9;  // crocoddyl reg_min
constexpr double kRegMax = 1e9;

// ---------------------------------------------------------------------------
// K1: derivative pass over the running nodes.  unit = b*T + t, one lane each.
// ---------------------------------------------------------------------------
template <int NV, bool CHAIN, bool GEN = false>
__global__ void __launch_bounds__(64) k_calc_diff(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                  const double *__restrict__ dts, const double *__restrict__ xs,
                                                  const double *__restrict__ us, RefView rv,
                                                  double *__restrict__ tiles, const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV, NU = NV;
  typedef TileOff<NV> TO;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (unit >= (long long)o.B * T) return;
  const int b = (int)(unit / T), t = (int)(unit % T);
  if (st && st[b].done) return;
  const double dt = dts[t];
  double x[NX], u[NU];
  const double *xp = xs + ((long long)b * (T + 1) + t) * NX;
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = xp[i];
  const double *up = us + ((long long)b * T + t) * NU;
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = up[i];
  double *tile = tiles + ((long long)b * (T + 1) + t) * TO::SIZE;

  Kin<NV> k;
  kinematics<NV, CHAIN>(m, x, k);
  Dyn<NV> d;
  double nle[NV], M[NV][NV], Minv[NV][NV], qdd[NV];
  bias_and_inertia<NV, CHAIN>(m, k, x + NV, d, nle, M);
  spd_inverse<NV>(M, Minv);
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) a += Minv[i][j] * (u[j] - nle[j]);
    qdd[i] = a;
  }
  // gap f = xnext - xs[t+1]
  {
    const double *xn = xp + NX;
AGX_UNROLL_NV
    for (int i = 0; i < NV; ++i) {
      tile[TO::f + i] = x[i] + dt * x[NV + i] + dt * dt * qdd[i] - xn[i];
      tile[TO::f + NV + i] = x[NV + i] + dt * qdd[i] - xn[NV + i];
    }
  }
  {
    double dq[NV][NV], dv[NV][NV];
    rnea_derivatives<NV, CHAIN>(m, k, d, x + NV, qdd, dq, dv);
    // da/dq = -Minv dtau/dq ; da/dv = -Minv dtau/dv ; da/du = Minv
    const double dt2 = dt * dt;
AGX_UNROLL_NV
    for (int i = 0; i < NV; ++i) {
AGX_UNROLL_NV
      for (int j = 0; j < NV; ++j) {
        double aq = 0.0, av = 0.0;
AGX_UNROLL_NV
        for (int l = 0; l < NV; ++l) {
          aq -= Minv[i][l] * dq[l][j];
          av -= Minv[i][l] * dv[l][j];
        }
        // Fx = I + [[dt^2 aq, dt I + dt^2 av],[dt aq, dt av]]
        tile[TO::Fx + i * NX + j] = (i == j ? 1.0 : 0.0) + dt2 * aq;
        tile[TO::Fx + i * NX + NV + j] = (i == j ? dt : 0.0) + dt2 * av;
        tile[TO::Fx + (NV + i) * NX + j] = dt * aq;
        tile[TO::Fx + (NV + i) * NX + NV + j] = (i == j ? 1.0 : 0.0) + dt * av;
        tile[TO::Fu + i * NU + j] = dt2 * Minv[i][j];
        tile[TO::Fu + (NV + i) * NU + j] = dt * Minv[i][j];
      }
    }
  }
  CostAcc<NV> c;
  node_costs<NV, CHAIN, false, true>(m, o.rows[0], k, x, u, ref_at(rv, b, t, T), frames_at(rv, b, t, T), c);
  CostGen<NV> g;
  if constexpr (GEN) node_costs_general<NV, CHAIN, false, true>(m, o.rows[0], k, x, u, ref_at(rv, b, t, T), frames_at(rv, b, t, T), c, g);
  tile[TO::cost] = dt * c.cost;
#pragma unroll
  for (int i = 0; i < NX * NU; ++i) tile[TO::Lxu + i] = 0.0;
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    tile[TO::Lx + i] = dt * c.Lq[i];
    tile[TO::Lx + NV + i] = dt * c.Lv[i];
    tile[TO::Lu + i] = dt * c.Lu[i];
AGX_UNROLL_NV
    for (int j = 0; j < NV; ++j) {
      tile[TO::Lxx + i * NX + j] = dt * c.Lqq[i][j];
      tile[TO::Lxx + i * NX + NV + j] = GEN ? dt * g.Lqv[i][j] : 0.0;
      tile[TO::Lxx + (NV + i) * NX + j] = GEN ? dt * g.Lqv[j][i] : 0.0;
      tile[TO::Lxx + (NV + i) * NX + NV + j] = ((i == j) ? dt * c.Lvv[i] : 0.0) + (GEN ? dt * g.Lvvd[i][j] : 0.0);
      tile[TO::Luu + i * NU + j] = (i == j) ? dt * c.Luu[i] : 0.0;
      if (GEN) tile[TO::Lxu + i * NU + j] = dt * g.Lqu[i][j];
    }
  }
}

// terminal nodes: cost only, xnext = x (dt = 0, cost not scaled)
template <int NV, bool CHAIN, bool GEN = false>
__global__ void __launch_bounds__(64) k_calc_diff_term(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                       const double *__restrict__ xs, RefView rv,
                                                       double *__restrict__ tiles, const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV, NU = NV;
  typedef TileOff<NV> TO;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= o.B) return;
  if (st && st[b].done) return;
  double x[NX];
  const double *xp = xs + ((long long)b * (T + 1) + T) * NX;
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = xp[i];
  double *tile = tiles + ((long long)b * (T + 1) + T) * TO::SIZE;
  Kin<NV> k;
  kinematics<NV, CHAIN>(m, x, k);
  CostAcc<NV> c;
  node_costs<NV, CHAIN, true, true>(m, o.rows[1], k, x, nullptr, ref_at(rv, b, T, T), frames_at(rv, b, T, T), c);
  CostGen<NV> g;
  if constexpr (GEN) node_costs_general<NV, CHAIN, true, true>(m, o.rows[1], k, x, nullptr, ref_at(rv, b, T, T), frames_at(rv, b, T, T), c, g);
  tile[TO::cost] = c.cost;
AGX_UNROLL_NV
  for (int i = 0; i < NX; ++i) {
AGX_UNROLL_NV
    for (int j = 0; j < NX; ++j) tile[TO::Fx + i * NX + j] = (i == j) ? 1.0 : 0.0;
    tile[TO::f + i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < NX * NU; ++i) { tile[TO::Fu + i] = 0.0; tile[TO::Lxu + i] = 0.0; }
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    tile[TO::Lx + i] = c.Lq[i];
    tile[TO::Lx + NV + i] = c.Lv[i];
    tile[TO::Lu + i] = 0.0;
AGX_UNROLL_NV
    for (int j = 0; j < NV; ++j) {
      tile[TO::Lxx + i * NX + j] = c.Lqq[i][j];
      tile[TO::Lxx + i * NX + NV + j] = GEN ? g.Lqv[i][j] : 0.0;
      tile[TO::Lxx + (NV + i) * NX + j] = GEN ? g.Lqv[j][i] : 0.0;
      tile[TO::Lxx + (NV + i) * NX + NV + j] = ((i == j) ? c.Lvv[i] : 0.0) + (GEN ? g.Lvvd[i][j] : 0.0);
      tile[TO::Luu + i * NU + j] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------
// Production path: QP tiles in acceleration-input form.
//
// With a = Minv (u - nle) the Euler node reads  x+ = Phi x + G a,  Phi = [[I, hI],[0, I]],
// G = [h^2 I; h I].  Linearised:  dx+ = Phi dx + G w + f  with  w = da = -Minv taux dx + Minv du,
// i.e.  du = M w + taux dx  (M = mass matrix + armature, taux = [dtau/dq dtau/dv] from RNEA).
// Substituting du into the node's quadratic model gives a QP in (dx, w) whose dynamics are the
// same trivially structured (Phi, G) at every node, so the sequential Riccati recursion only
// needs O(n^2) block combinations per node plus one nu x nu Cholesky and the Schur complement;
// every dense product involving M and taux is done here, in the node-parallel kernel:
//   Hww = M D M,  Hxw = taux' D M,  Hxx = Lxx + taux' D taux,  gw = M Lu,  gx = Lx + taux' Lu
// (D = Luu + preg: crocoddyl's control regularisation lives in u-space and is folded in).
// The gains are mapped back with  K = M Kw - taux,  k = M kw.
//
// QP tile (QT):  Hqq | Hqv | Hvv | Hqw | Hvw | Hww (NV x NV each) | gx (NX) | gw (NV) | f (NX) | cost
// aux tile (AUX): M | tauq | tauv | Lqq (NV x NV each) | Lvv | Luu | Lu (NV each)
// ---------------------------------------------------------------------------
// Every NV x NV block is stored with a row stride of LD = 8 doubles: a row is one aligned 64-byte
// line, so the column-per-lane kernels write whole lines and the 8 x 8 lane grid of the Riccati
// kernel reads element [r][c] of a block at offset 8 r + c = its own lane id (one 512-byte load).
template <int NV>
struct QT {
  static constexpr int NX = 2 * NV, LD = (NV <= 8 ? 8 : 32), B2 = NV * LD;  // row stride: one (nv <= 8) or four 64-byte lines
  static constexpr int Hqq = 0, Hqv = B2, Hvv = 2 * B2, Hqw = 3 * B2, Hvw = 4 * B2, Hww = 5 * B2, gx = 6 * B2, gw = gx + 2 * LD,
                       f = gw + LD, cost = f + 2 * LD, SIZE = cost + 8;
};
template <int NV>
struct AUX {
  static constexpr int LD = (NV <= 8 ? 8 : 32), B2 = NV * LD;
  static constexpr int M = 0, tq = B2, tv = 2 * B2, Lqq = 3 * B2, Lvv = 4 * B2, Luu = Lvv + LD, Lu = Luu + LD, SIZE = Lu + LD;
};

template <int NV, bool CHAIN, bool GEN = false>
__device__ __forceinline__ void calc_qp_body(const long long unit, const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                             const double *__restrict__ dts, const double *__restrict__ xs,
                                             const double *__restrict__ us, const RefView &rv, double *__restrict__ qts,
                                             double *__restrict__ auxs, const DevState *__restrict__ st,
                                             double *__restrict__ auxg = nullptr, int phase = 0) {
  constexpr int NX = 2 * NV, NU = NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  if (unit >= (long long)o.B * T) return;
  const int b = (int)(unit / T), t = (int)(unit % T);
  if (!k1_active(st[b], phase)) return;
  const double preg = k1_preg(st[b], phase);
  const double dt = dts[t];
  double x[NX], u[NU];
  const double *xp = xs + ((long long)b * (T + 1) + t) * NX;
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = xp[i];
  const double *up = us + ((long long)b * T + t) * NU;
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = up[i];
  double *qt = qts + ((long long)b * (T + 1) + t) * Q::SIZE;
  double *ax = auxs + ((long long)b * (T + 1) + t) * A::SIZE;

  Kin<NV> k;
  kinematics<NV, CHAIN>(m, x, k);
  Dyn<NV> d;
  double nle[NV], M[NV][NV], L[NV][NV], Minv[NV][NV], qdd[NV];
  bias_and_inertia<NV, CHAIN>(m, k, x + NV, d, nle, M);
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < NV; ++j) { L[i][j] = M[i][j]; ax[A::M + i * A::LD + j] = M[i][j]; }
  static_assert(NV <= 8, "large models use the workgroup-per-node kernel (agx_big_k1.hpp)");
  spd_inverse<NV>(L, Minv);
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) a += Minv[i][j] * (u[j] - nle[j]);
    qdd[i] = a;
  }
  {
    const double *xn = xp + NX;
AGX_UNROLL_NV
    for (int i = 0; i < NV; ++i) {
      qt[Q::f + i] = x[i] + dt * x[NV + i] + dt * dt * qdd[i] - xn[i];
      qt[Q::f + NV + i] = x[NV + i] + dt * qdd[i] - xn[NV + i];
    }
  }
  double tq[NV][NV], tv[NV][NV];
  rnea_derivatives<NV, CHAIN>(m, k, d, x + NV, qdd, tq, tv);
  CostAcc<NV> c;
  node_costs<NV, CHAIN, false, true>(m, o.rows[0], k, x, u, ref_at(rv, b, t, T), frames_at(rv, b, t, T), c);
  CostGen<NV> g;
  if constexpr (GEN) node_costs_general<NV, CHAIN, false, true>(m, o.rows[0], k, x, u, ref_at(rv, b, t, T), frames_at(rv, b, t, T), c, g);
  qt[Q::cost] = dt * c.cost;
  double D[NV], lu[NV];
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    D[i] = dt * c.Luu[i] + preg;
    lu[i] = dt * c.Lu[i];
    ax[A::Lvv + i] = dt * c.Lvv[i];
    ax[A::Luu + i] = dt * c.Luu[i];
    ax[A::Lu + i] = lu[i];
  }
  // DM = D M (row scaling), then the five transformed blocks
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    double gwi = 0.0, gq = dt * c.Lq[i], gv = dt * c.Lv[i];
AGX_UNROLL_NV
    for (int l = 0; l < NV; ++l) {
      gwi += M[i][l] * lu[l];
      gq += tq[l][i] * lu[l];
      gv += tv[l][i] * lu[l];
    }
    qt[Q::gw + i] = gwi;
    qt[Q::gx + i] = gq;
    qt[Q::gx + NV + i] = gv;
AGX_UNROLL_NV
    for (int j = 0; j < NV; ++j) {
      double hww = 0.0, hqw = 0.0, hvw = 0.0, hqq = dt * c.Lqq[i][j], hqv = 0.0, hvv = (i == j) ? dt * c.Lvv[i] : 0.0;
AGX_UNROLL_NV
      for (int l = 0; l < NV; ++l) {
        const double dm = D[l] * M[l][j];
        hww += M[i][l] * dm;
        hqw += tq[l][i] * dm;
        hvw += tv[l][i] * dm;
        const double dtq = D[l] * tq[l][j], dtv = D[l] * tv[l][j];
        hqq += tq[l][i] * dtq;
        hqv += tq[l][i] * dtv;
        hvv += tv[l][i] * dtv;
      }
      if constexpr (GEN) {
        // general rows (agx_general.hpp): Lxu = [Lqu; 0], dense Lqv / Lvv:
        //   Hxx += Lxu taux + taux' Lxu',  Hxw += Lxu M
        double lt_ij = 0.0, lt_ji = 0.0, ltv = 0.0, lm = 0.0;
AGX_UNROLL_NV
        for (int l = 0; l < NV; ++l) {
          lt_ij += g.Lqu[i][l] * tq[l][j];
          lt_ji += g.Lqu[j][l] * tq[l][i];
          ltv += g.Lqu[i][l] * tv[l][j];
          lm += g.Lqu[i][l] * M[l][j];
        }
        hqq += dt * (lt_ij + lt_ji);
        hqv += dt * (g.Lqv[i][j] + ltv);
        hvv += dt * g.Lvvd[i][j];
        hqw += dt * lm;
        double *ag = auxg + ((long long)b * (T + 1) + t) * (3 * A::B2);
        ag[i * A::LD + j] = dt * g.Lqv[i][j];
        ag[A::B2 + i * A::LD + j] = dt * g.Lvvd[i][j];
        ag[2 * A::B2 + i * A::LD + j] = dt * g.Lqu[i][j];
      }
      qt[Q::Hww + i * Q::LD + j] = hww;
      qt[Q::Hqw + i * Q::LD + j] = hqw;
      qt[Q::Hvw + i * Q::LD + j] = hvw;
      qt[Q::Hqq + i * Q::LD + j] = hqq;
      qt[Q::Hqv + i * Q::LD + j] = hqv;
      qt[Q::Hvv + i * Q::LD + j] = hvv;
      ax[A::tq + i * A::LD + j] = tq[i][j];
      ax[A::tv + i * A::LD + j] = tv[i][j];
      ax[A::Lqq + i * A::LD + j] = dt * c.Lqq[i][j];
    }
  }
}

template <int NV, bool CHAIN, bool GEN = false>
__global__ void __launch_bounds__(64) k_calc_qp(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                const double *__restrict__ dts, const double *__restrict__ xs,
                                                const double *__restrict__ us, RefView rv, double *__restrict__ qts,
                                                double *__restrict__ auxs, const DevState *__restrict__ st,
                                                double *__restrict__ auxg = nullptr, int phase = 0) {
  calc_qp_body<NV, CHAIN, GEN>((long long)blockIdx.x * blockDim.x + threadIdx.x, mp, op, dts, xs, us, rv, qts, auxs, st, auxg, phase);
}

template <int NV, bool CHAIN, bool GEN = false>
__device__ __forceinline__ void calc_qp_term_body(const int b, const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                  const double *__restrict__ xs, const RefView &rv, double *__restrict__ qts,
                                                  double *__restrict__ auxs, const DevState *__restrict__ st,
                                                  double *__restrict__ auxg = nullptr, int phase = 0) {
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  if (b >= o.B) return;
  if (!k1_active(st[b], phase)) return;
  double x[NX];
  const double *xp = xs + ((long long)b * (T + 1) + T) * NX;
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = xp[i];
  double *qt = qts + ((long long)b * (T + 1) + T) * Q::SIZE;
  double *ax = auxs + ((long long)b * (T + 1) + T) * A::SIZE;
  Kin<NV> k;
  kinematics<NV, CHAIN>(m, x, k);
  CostAcc<NV> c;
  node_costs<NV, CHAIN, true, true>(m, o.rows[1], k, x, nullptr, ref_at(rv, b, T, T), frames_at(rv, b, T, T), c);
  CostGen<NV> g;
  if constexpr (GEN) node_costs_general<NV, CHAIN, true, true>(m, o.rows[1], k, x, nullptr, ref_at(rv, b, T, T), frames_at(rv, b, T, T), c, g);
  qt[Q::cost] = c.cost;
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    qt[Q::gx + i] = c.Lq[i];
    qt[Q::gx + NV + i] = c.Lv[i];
    qt[Q::gw + i] = 0.0;
    qt[Q::f + i] = 0.0;
    qt[Q::f + NV + i] = 0.0;
    ax[A::Lvv + i] = c.Lvv[i];
    ax[A::Luu + i] = 0.0;
    ax[A::Lu + i] = 0.0;
AGX_UNROLL_NV
    for (int j = 0; j < NV; ++j) {
      qt[Q::Hqq + i * Q::LD + j] = c.Lqq[i][j];
      qt[Q::Hqv + i * Q::LD + j] = GEN ? g.Lqv[i][j] : 0.0;
      qt[Q::Hvv + i * Q::LD + j] = ((i == j) ? c.Lvv[i] : 0.0) + (GEN ? g.Lvvd[i][j] : 0.0);
      if constexpr (GEN) {
        double *ag = auxg + ((long long)b * (T + 1) + T) * (3 * A::B2);
        ag[i * A::LD + j] = g.Lqv[i][j];
        ag[A::B2 + i * A::LD + j] = g.Lvvd[i][j];
        ag[2 * A::B2 + i * A::LD + j] = 0.0;
      }
      qt[Q::Hqw + i * Q::LD + j] = 0.0;
      qt[Q::Hvw + i * Q::LD + j] = 0.0;
      qt[Q::Hww + i * Q::LD + j] = 0.0;
      ax[A::M + i * A::LD + j] = 0.0;
      ax[A::tq + i * A::LD + j] = 0.0;
      ax[A::tv + i * A::LD + j] = 0.0;
      ax[A::Lqq + i * A::LD + j] = c.Lqq[i][j];
    }
  }
}

template <int NV, bool CHAIN, bool GEN = false>
__global__ void __launch_bounds__(64) k_calc_qp_term(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                     const double *__restrict__ xs, RefView rv, double *__restrict__ qts,
                                                     double *__restrict__ auxs, const DevState *__restrict__ st,
                                                     double *__restrict__ auxg = nullptr, int phase = 0) {
  calc_qp_term_body<NV, CHAIN, GEN>(blockIdx.x * blockDim.x + threadIdx.x, mp, op, xs, rv, qts, auxs, st, auxg, phase);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------
// K2: Riccati backward + linear forward in (dx, w) coordinates, one wave per instance,
// REGISTER RESIDENT: no LDS, no barriers.
//
// The 64 lanes form an 8 x 8 grid, lane = 8 r + c.  Lane (r, c) holds element [r][c] of every
// NV x NV block of the symmetric 3 x 3 block matrix
//        [ Qww Qwq Qwv ]          [ qw ]
//    M = [ Qqw Qqq Qqv ] ,   g =  [ qq ]   (g_*[r], replicated over c)
//        [ Qvw Qvq Qvv ]          [ qv ]
// Phase A builds M and g from the node's QP tile and the value function of node t+1; thanks to
// the (Phi, G) structure every entry is a local combination of the lane's own V elements
// (Vqq, Vqv, Vvq = Vqv', Vvv at [r][c]) -- only V f needs a row reduction (3 xor shuffles).
// Phase B eliminates the NV acceleration variables by Gauss-Jordan pivots inside the ww block,
// applied to the whole matrix: afterwards the x-x blocks ARE the Schur complement
// Qxx - Qxw Qww^-1 Qwx (= the new value Hessian, bitwise symmetric by construction), the w-x
// blocks divided by the pivots are the gains Kw, g_x is the new value gradient and g_w / pivot
// the feed-forward kw.  Per pivot: 6 cross-lane fetches (column k of the lane's row, row k of
// the lane's column), 12 FMAs.
// ---------------------------------------------------------------------------
// value held by lane (my row, column K) of the 8 x 8 lane grid, K a compile-time constant.  gfx950's
// double-precision ALU takes a DPP operand for the row_newbcast control only (v_mov_b64_dpp): one
// broadcast of lane K over the whole 16-lane DPP row (two grid rows), then lane 8 + K over the upper
// grid row through the bank mask -- two 64-bit moves instead of four 32-bit ones plus the zero fill.
template <int K>
__device__ __forceinline__ double grid_col(double x) {
  double d = __builtin_amdgcn_mov_dpp(x, 0x150 + K, 0xf, 0xf, false);
  return __builtin_amdgcn_update_dpp(d, x, 0x150 + 8 + K, 0xf, 0xc, false);
}
// In front of a software-pipelined loop that keeps several nodes of loads in flight: make the loads of the prologue
// land once.  The compiler's s_waitcnt at the loop head is the minimum over the paths that reach it, and the prologue
// (whose loads it reorders, and sinks towards the loop -- neither s_waitcnt nor a memory clobber holds back a load
// through a __restrict__ pointer) would otherwise pin it near vmcnt(0) on every pass.  An empty asm that "modifies"
// every prefetched register is a use the loads cannot move past.
template <class SET, int N>
__device__ __forceinline__ void prefetch_queue_settle(SET (&sets)[N]) {
  static_assert(sizeof(SET) % sizeof(double) == 0, "register sets of doubles");
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double *p = reinterpret_cast<double *>(&sets[i]);
#pragma unroll
    for (int e = 0; e < (int)(sizeof(SET) / sizeof(double)); ++e) asm volatile("" : "+v"(p[e]));
  }
}
// Around the reload of one register set at the end of a step: keeps the loads of the step together and in program
// order.  Left to itself the scheduler regroups the loads of the unrolled steps by base pointer; vmcnt counts in
// order, so the first use of the group issued last then waits for every load in flight.
__device__ __forceinline__ void prefetch_group_begin() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void prefetch_group_end() { __builtin_amdgcn_sched_barrier(0); }
// Step lengths of the horizon for the time-serial sweeps, staged in LDS once per wave.  Read per node from global
// memory they sit at the end of every group of prefetch loads: the scalar path would put an s_waitcnt lgkmcnt(0)
// on every node, and on the vector path the compiler gathers the loads of the unrolled steps at the top of the
// loop, where waiting for them (vmcnt counts in order) drains the tile prefetch queue behind them.
constexpr int kMaxHorizon = 512;  // nodes per instance the sweeps (and k_step) take; checked on the host
__device__ __forceinline__ void stage_dts(double *__restrict__ s_dt, const double *__restrict__ dts, int T) {
  for (int i = threadIdx.x & 63; i < T; i += 64) s_dt[i] = dts[i];
  wave_lds_sync();
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

// GAINS = true is the exit pass: CSQP's proximal (sigma) backward sweep whose gains the solver reports.
// sigma enters every Hessian block exactly like an extra control weight (D -> D + sigma) plus sigma I
// on Hxx; the corrections  sigma [taux M]' [taux M]  are formed on the fly from the aux tile, the
// gradient recursion is skipped (it does not influence the gains) and the gains are mapped to
// u-space in registers:  K = M Kw - taux  -> Kout.  No forward pass.
// Linear forward pass of one instance with the gains (Kw, kw) of the backward sweep, on the same
// 8 x 8 lane grid as the backward sweep: lane (r, c) holds Kw[r][c] (q and v halves), the state lives
// twice -- indexed by the lane's column (operand of the products) and by its row (the update).
// Per node: two FMAs, a DPP row reduction (w = -kw - Kw dx), the lane-local state update and one
// transposing broadcast of the new state (row r's value to every lane of column r).
// KKT = true folds K3 (k_node_kkt) into the pass: du = M w + taux dx of every node, and the instance totals of the
// KKT residual / cost / gap shares (same identities as k_node_kkt).  The pass is a dependent chain with idle issue
// slots, so the node-parallel kernel's 2 KB per node of aux-tile reads hide behind it.  The totals go to the
// nodestat entry of node 0 (the entries of the other nodes stay zero from allocation: k_step's reduction over the
// nodes then returns them unchanged); cost and gap are summed node by node, in horizon order.
struct FwdKkt {
  const double *ab;   // aux tiles of the instance
  double preg, dreg;  // regularisation of the direction
  double *du;         // [T][NV] of the instance
  double *ns;         // nodestat of the instance's node 0
};
template <int NV, bool KKT = false>
__device__ __forceinline__ void riccati_forward(const int b, const int T, const double *__restrict__ dts, const double *__restrict__ qb,
                                                const double *__restrict__ Kw, const double *__restrict__ kw,
                                                double *__restrict__ dxs, double *__restrict__ wss, const double *s_dt,
                                                const FwdKkt kk = FwdKkt()) {
  constexpr int NX = 2 * NV, TS = QT<NV>::SIZE;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const int lane = threadIdx.x;
  const int r = lane >> 3, c = lane & 7;
  const bool in = (r < NV) && (c < NV);
  const int rr = r < NV ? r : 0, cc = c < NV ? c : 0;
  const double inm = in ? 1.0 : 0.0, rowm = r < NV ? 1.0 : 0.0;
  double *dx = dxs + (long long)b * (T + 1) * NX, *ws = wss + (long long)b * T * NV;
  if (lane < NV) { dx[lane] = 0.0; dx[NV + lane] = 0.0; }
  __threadfence_block();  // the gains written by the backward sweep are read back by other lanes below
  double dq_r = 0.0, dv_r = 0.0, dq_c = 0.0, dv_c = 0.0;
  double kkt_run = 0.0, gap_run = 0.0, cost_run = 0.0;
  int vz = 0;
  if (KKT) asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  struct GainPlain { double kq, kv, kw, fq, fv; };
  struct GainKkt { double kq, kv, kw, fq, fv, am, atq, atv, alqq, aluu, alvv, cost; };
  typedef typename std::conditional<KKT, GainKkt, GainPlain>::type Gain;
#ifndef AGX_FWD_DEPTH
#define AGX_FWD_DEPTH 8
#endif
#ifndef AGX_FWDK_DEPTH
#define AGX_FWDK_DEPTH 4
#endif
  constexpr int FWD_DEPTH = KKT ? AGX_FWDK_DEPTH : AGX_FWD_DEPTH;  // nodes of gains in flight (12 doubles per node with KKT, 5 without)
  auto load_gain = [&](Gain &g, int t) {
    const double *kr = Kw + ((long long)t * NV + rr) * NX;
    g.kq = kr[cc];  // lanes outside the block load element [0] of their row / column and are masked at the use: a
    g.kv = kr[NV + cc];  // conditional load is a branch, and the join behind it puts register copies on the loop latch
    g.kw = kw[(long long)t * NV + rr];
    g.fq = qb[(long long)t * TS + Q::f + rr];
    g.fv = qb[(long long)t * TS + Q::f + NV + rr];
    if constexpr (KKT) {
      const double *ax = kk.ab + (long long)t * A::SIZE;
      g.am = ax[A::M + rr * A::LD + cc]; g.atq = ax[A::tq + rr * A::LD + cc]; g.atv = ax[A::tv + rr * A::LD + cc];
      g.alqq = ax[A::Lqq + rr * A::LD + cc];
      g.aluu = ax[A::Luu + rr]; g.alvv = ax[A::Lvv + rr];
      g.cost = qb[(long long)t * TS + Q::cost + vz];  // vz: a zero the compiler cannot see -- a wave-uniform address would go
                                                      // through the scalar cache, and its s_waitcnt lgkmcnt(0) onto the chain
    }
  };
  auto row_sum = [](double p) { p += dpp_xor1(p); p += dpp_xor2(p); p += dpp_xor4(p); return p; };
  // the state-dependent shares of node t's KKT residual: |Lqq dq + dreg dq|, |(Lvv + dreg) dv|  (zero at t = 0: dx_0 = 0)
  auto state_share = [&](double alqq, double alvv) {
    const double hq = row_sum(alqq * dq_c * inm);
    return fmax(fabs(hq + kk.dreg * dq_r), fabs((alvv + kk.dreg) * dv_r));
  };
  auto fstep = [&](Gain &g, int t) {
    const double h = s_dt[t], h2 = h * h;
    double p = (g.kq * inm) * dq_c + (g.kv * inm) * dv_c;
    const double kwv = g.kw, fqc = g.fq, fvc = g.fv;
    p = row_sum(p);  // on every lane of the row
    const double wv = -(kwv + p);
    if constexpr (KKT) {
      const double w_c = __shfl(wv, 8 * cc, 64);
      const double du = row_sum((g.am * w_c + g.atq * dq_c + g.atv * dv_c) * inm);  // du = M w + taux dx, row r
      double share = fmax(fmax(fabs(fqc), fabs(fvc)), fabs((g.aluu + kk.preg) * du));
      share = fmax(share, state_share(g.alqq, g.alvv));
      kkt_run = fmax(kkt_run, share * rowm);
      gap_run += (fabs(fqc) + fabs(fvc)) * rowm;
      cost_run += g.cost;
      if (c == 0 && r < NV) kk.du[(long long)t * NV + r] = du;
    }
    const double nq = dq_r + h * dv_r + h2 * wv + fqc;
    const double nv2 = dv_r + h * wv + fvc;
    dq_r = nq; dv_r = nv2;
    dq_c = __shfl(nq, 8 * cc, 64);   // row cc's value (lane (cc, 0)) to every lane of column cc
    dv_c = __shfl(nv2, 8 * cc, 64);
    if (c == 0 && r < NV) {
      ws[(long long)t * NV + r] = wv;
      dx[(long long)(t + 1) * NX + r] = nq;
      dx[(long long)(t + 1) * NX + NV + r] = nv2;
    }
    // refill this register set FWD_DEPTH nodes ahead: after the last use of its old contents and unconditionally
    // (the last node again at the end) -- see the note on the loop latch in agx_riccati_mx.hpp
    prefetch_group_begin();
    load_gain(g, t + FWD_DEPTH < T ? t + FWD_DEPTH : T - 1);
    prefetch_group_end();
  };
  int t = 0;
  // the nodes that do not fill a group first, one at a time; the pipelined loop then runs whole groups
  for (int rem = T % FWD_DEPTH; rem > 0; --rem, ++t) {
    Gain g1;
    load_gain(g1, t);
    fstep(g1, t);
  }
  if (t < T) {
    Gain g[FWD_DEPTH];
#pragma unroll
    for (int i = 0; i < FWD_DEPTH; ++i) load_gain(g[i], t + i);
    prefetch_queue_settle(g);
    for (; t < T; t += FWD_DEPTH) {
#pragma unroll
      for (int i = 0; i < FWD_DEPTH; ++i) fstep(g[i], t + i);
    }
  }
  if constexpr (KKT) {
    // terminal node: only the state-dependent shares and its cost
    const double *ax = kk.ab + (long long)T * A::SIZE;
    kkt_run = fmax(kkt_run, state_share(ax[A::Lqq + rr * A::LD + cc], ax[A::Lvv + rr]) * rowm);
    cost_run += qb[(long long)T * TS + Q::cost];
    kkt_run = wave_max(kkt_run);
    double gap = 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) gap += readlane_f64(gap_run, 8 * i);  // rows in order (every lane of a row holds the row's sum)
    if (lane == 0) { kk.ns[0] = kkt_run; kk.ns[1] = cost_run; kk.ns[2] = gap; kk.ns[3] = 0.0; }
  }
}

// Factors of one node kept for gradient-only sweeps (ADMM iterations that do not change rho,
// agx_admm.hpp): the elimination multipliers of the 7 pivots, the pivot reciprocals and V' f.
template <int NV>
struct FT {
  static constexpr int FW = 0, FQ = 56, FV = 112, RP = 168, PQ = 176, PV = 184, SIZE = 192;
};

template <int NV, bool GAINS, bool STORE = false>
__device__ __forceinline__ void riccati_body(const int b, const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                             const double *__restrict__ qts, const double *__restrict__ auxs,
                                             double *__restrict__ Kws, double *__restrict__ kws,
                                             double *__restrict__ dxs, double *__restrict__ wss,
                                             double *__restrict__ dus, double *__restrict__ Kout,
                                             DevState *__restrict__ st, int forward, int gmode, int iter,
                                             double *__restrict__ facs = nullptr) {
  (void)dus;
  constexpr int gains_pass = GAINS ? 1 : 0;
  typedef AUX<NV> A;
  static_assert(NV <= 7, "the register-resident Riccati kernel maps an NV x NV block plus a gradient column onto an 8 x 8 lane grid");
  constexpr int NX = 2 * NV, TS = QT<NV>::SIZE;
  typedef QT<NV> Q;
  const DevOcp &o = *op;
  const int T = o.T, lane = threadIdx.x;
  DevState &S = st[b];
  bool bad_pivot = false;  // per lane: the reciprocal pivot of its grid row was not positive at some node
  // direction sweep: live instances.  Gains sweep, by gmode:
  //   0  every instance (agx_ocp_direction, timing), with the regularisation of its last direction;
  //   1  speculative, launched next to the direction sweep of SQP iteration `iter`: live instances, same dreg;
  //   2  fix-up on exit: only instances whose last direction (dir_iter) has no sweep yet.
  if (!gains_pass && (S.done || S.admm_conv)) return;
  // gmode 4: as 1 (unfinished instances, current regularisation) but launched on its own after the head of the step, which
  // has set dir_iter: the sweep of an iteration the loop may end with before the line search overwrites its tiles
  if (gains_pass && (gmode == 1 || gmode == 4) && S.done) return;
  if (gains_pass && gmode == 2 && S.gains_iter == S.dir_iter) return;
  const double dreg = (gains_pass && gmode != 1 && gmode != 4) ? (S.solved ? S.dreg : S.gains_dreg) : S.dreg;
  if (gains_pass && gmode != 0 && lane == 0) S.gains_iter = (gmode == 1) ? iter : S.dir_iter;
  const double *qb = qts + (long long)b * (T + 1) * TS;
  const double *ab = auxs + (long long)b * (T + 1) * A::SIZE;
  double *Kw = Kws + (long long)b * T * NV * NX, *kw = kws + (long long)b * T * NV;
  double *Ko = GAINS ? Kout + (long long)b * T * NV * NX : nullptr;
  const double sig = GAINS ? kSigma : 0.0;
  const int r = lane >> 3, c = lane & 7;
  const bool in = (r < NV) && (c < NV);
  const int rc = in ? r * Q::LD + c : 0, cr = in ? c * Q::LD + r : 0;  // [r][c] and [c][r] of a block
  const int rr = r < NV ? r : 0, cc = c < NV ? c : 0;
  const double diag = (r == c) ? 1.0 : 0.0;
  const int col_lane = c;
  // value function of node t+1 (starts as the terminal cost + dreg)
  double Vqq, Vqv, Vvq, Vvv, vxq, vxv;
  {
    const double *tt = qb + (long long)T * TS;
    Vqq = in ? tt[Q::Hqq + rc] + (dreg + sig) * diag : 0.0;
    Vqv = in ? tt[Q::Hqv + rc] : 0.0;
    Vvq = in ? tt[Q::Hqv + cr] : 0.0;
    Vvv = in ? tt[Q::Hvv + rc] + (dreg + sig) * diag : 0.0;
    vxq = (r < NV) ? tt[Q::gx + rr] : 0.0;
    vxv = (r < NV) ? tt[Q::gx + NV + rr] : 0.0;
  }
  // tile elements of a node, prefetched kGridDepth nodes ahead into as many register sets
  constexpr int kGridDepth = 4;
  struct Tile {
    double hqq, hqv, hvq, hvv, hqw, hvw, hwq, hwv, hww, gq, gv, gwr, fq, fv;
  };
  // GAINS: the aux blocks M | tq | tv of a node (contiguous, 3 * NV * LD doubles) travel
  // global -> three registers (one node ahead) -> LDS (double buffered); every lane then reads
  // the columns it needs as LDS broadcasts instead of holding 6 * NV prefetched values.
  constexpr int AXN = 3 * A::B2;
  __shared__ double s_aux[GAINS ? 2 : 1][GAINS ? AXN : 1];
  double axr0 = 0.0, axr1 = 0.0, axr2 = 0.0;
  auto aux_fetch = [&](int t) {
    const double *al = ab + (long long)t * A::SIZE + A::M;
    axr0 = (lane < AXN) ? al[lane] : 0.0;
    axr1 = (64 + lane < AXN) ? al[64 + lane] : 0.0;
    axr2 = (128 + lane < AXN) ? al[128 + lane] : 0.0;
  };
  auto aux_put = [&](int t) {
    double *sa = s_aux[GAINS ? (t & 1) : 0];
    if (lane < AXN) sa[lane] = axr0;
    if (64 + lane < AXN) sa[64 + lane] = axr1;
    if (128 + lane < AXN) sa[128 + lane] = axr2;
  };
  auto load_tile = [&](Tile &z, int t) {
    const double *tl = qb + (long long)t * TS;
    z.hqq = tl[Q::Hqq + rc]; z.hqv = tl[Q::Hqv + rc]; z.hvq = tl[Q::Hqv + cr]; z.hvv = tl[Q::Hvv + rc];
    z.hqw = tl[Q::Hqw + rc]; z.hvw = tl[Q::Hvw + rc]; z.hwq = tl[Q::Hqw + cr]; z.hwv = tl[Q::Hvw + cr];
    z.hww = tl[Q::Hww + rc];
    if (!GAINS) {
      z.gq = tl[Q::gx + rr]; z.gv = tl[Q::gx + NV + rr]; z.gwr = tl[Q::gw + rr];
      z.fq = tl[Q::f + cc]; z.fv = tl[Q::f + NV + cc];
    }
  };
  __shared__ double s_dt[kMaxHorizon];
  stage_dts(s_dt, dts, T);
  auto step = [&](Tile &z, int t) {
    const double h = s_dt[t], h2 = h * h;
    // current tile -> working copies (the register set is refilled at the end of the step)
    double Hqq_ = z.hqq, Hqv_ = z.hqv, Hvq_ = z.hvq, Hvv_ = z.hvv, Hqw_ = z.hqw, Hvw_ = z.hvw, Hwq_ = z.hwq, Hwv_ = z.hwv, Hww_ = z.hww;
    const double gq_ = GAINS ? 0.0 : z.gq, gv_ = GAINS ? 0.0 : z.gv, gw_ = GAINS ? 0.0 : z.gwr;
    const double fq_ = (c < NV && !GAINS) ? z.fq : 0.0, fv_ = (c < NV && !GAINS) ? z.fv : 0.0;
    double Mr_[NV], tq_rc = 0.0, tv_rc = 0.0;
    if (GAINS) {
      // aux pipeline: registers hold node t-1 (fetched during the previous step) -> LDS; fetch node t-2
      wave_lds_sync();
      if (t >= 1) aux_put(t - 1);
      if (t >= 2) aux_fetch(t - 2);
      wave_lds_sync();
      const double *sa = s_aux[GAINS ? (t & 1) : 0];
      // sigma [taux M]' [taux M] at [r][c] (and at [c][r] for the transposed blocks), sigma I on Hxx
      double xww = 0.0, xqw = 0.0, xwq = 0.0, xvw = 0.0, xwv = 0.0, xqq = 0.0, xqv = 0.0, xvq = 0.0, xvv = 0.0;
AGX_UNROLL_NV
      for (int l = 0; l < NV; ++l) {
        const double aMr = sa[A::M + l * A::LD + rr], aMc = sa[A::M + l * A::LD + cc];
        const double aqr = sa[A::tq + l * A::LD + rr], aqc = sa[A::tq + l * A::LD + cc];
        const double avr = sa[A::tv + l * A::LD + rr], avc = sa[A::tv + l * A::LD + cc];
        xww += aMr * aMc;
        xqw += aqr * aMc; xwq += aMr * aqc;
        xvw += avr * aMc; xwv += aMr * avc;
        xqq += aqr * aqc; xqv += aqr * avc; xvq += avr * aqc; xvv += avr * avc;
        Mr_[l] = aMr;
      }
      Hww_ += sig * xww; Hqw_ += sig * xqw; Hwq_ += sig * xwq; Hvw_ += sig * xvw; Hwv_ += sig * xwv;
      Hqq_ += sig * (xqq + diag); Hqv_ += sig * xqv; Hvq_ += sig * xvq; Hvv_ += sig * (xvv + diag);
      tq_rc = sa[A::tq + rc]; tv_rc = sa[A::tv + rc];
    }
    // ---- phase A
    // The gradient rides in the spare grid column c = 7 of the w column block (NV <= 7):
    // lane (r, 7) keeps qw[r] in Mww, qx[r] in Mqw / Mvw, so the pivots below update it like any
    // other column and no separate broadcast of the pivot row's gradient is needed.
    const bool gcol = !GAINS && (c == 7) && (r < NV);
    double vpq = 0.0, vpv = 0.0;
    if (!GAINS) {
      // vp = vx + V f  (row reduction over c; lanes outside the block contribute zeros)
      double pq = Vqq * fq_ + Vqv * fv_, pv = Vvq * fq_ + Vvv * fv_;
      pq += dpp_xor1(pq); pv += dpp_xor1(pv);
      pq += dpp_xor2(pq); pv += dpp_xor2(pv);
      pq += dpp_xor4(pq); pv += dpp_xor4(pv);
      vpq = vxq + pq; vpv = vxv + pv;
      if constexpr (STORE) {
        double *ft = facs + ((long long)b * T + t) * FT<NV>::SIZE;
        if (c == 0) { ft[FT<NV>::PQ + r] = pq; ft[FT<NV>::PV + r] = pv; }
      }
    }
    // Y = G' V  (rows indexed by the acceleration variable):  Yq = h^2 Vqq + h Vvq, Yv = h^2 Vqv + h Vvv
    const double Yq = h2 * Vqq + h * Vvq, Yv = h2 * Vqv + h * Vvv;
    // Y' at [r][c]: Yq'[r][c] = Yq[c][r] = h^2 Vqq + h Vqv ; Yv'[r][c] = Yv[c][r] = h^2 Vvq + h Vvv
    const double YqT = h2 * Vqq + h * Vqv, YvT = h2 * Vvq + h * Vvv;
    double Mww = Hww_ + h2 * Yq + h * Yv;            // Hww + G' V G
    double Mwq = Hwq_ + Yq, Mwv = Hwv_ + h * Yq + Yv;  // (G' V Phi)
    double Mqw = Hqw_ + YqT, Mvw = Hvw_ + h * YqT + YvT;
    double Mqq = Hqq_ + Vqq;
    double Mqv = Hqv_ + h * Vqq + Vqv;
    double Mvq = Hvq_ + h * Vqq + Vvq;
    double Mvv = Hvv_ + h2 * Vqq + h * (Vqv + Vvq) + Vvv;
    if (!in) { Mww = diag; Mwq = 0.0; Mwv = 0.0; Mqw = 0.0; Mvw = 0.0; Mqq = 0.0; Mqv = 0.0; Mvq = 0.0; Mvv = 0.0; }
    if (gcol) {
      Mww = gw_ + h2 * vpq + h * vpv;  // qw[r]
      Mqw = gq_ + vpq;                 // qx = gx + Phi' vp
      Mvw = gv_ + h * vpq + vpv;
    }
    // ---- phase B: Gauss-Jordan pivots k = 0..NV-1 in the ww block
    double rp_row = 1.0;
    auto pivot = [&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      if (k >= NV) return;
      // row k of my column (three column blocks: bpermute), column k of my row (three row blocks: DPP)
      const double rw = __shfl(Mww, 8 * k + col_lane, 64), rq = __shfl(Mwq, 8 * k + col_lane, 64), rv2 = __shfl(Mwv, 8 * k + col_lane, 64);
      const double cw = grid_col<k>(Mww), cq = grid_col<k>(Mqw), cv = grid_col<k>(Mvw);
      const double piv = readlane_f64(Mww, 9 * k);  // wave-uniform: scalar broadcast
      const double rp = fast_rcp(piv);
      const double fw = (r == k) ? 0.0 : cw * rp;  // the pivot row itself is left untouched
      const double fqx = cq * rp, fvx = cv * rp;
      if constexpr (STORE) {
        double *ft = facs + ((long long)b * T + t) * FT<NV>::SIZE;
        if (c == 0) { ft[FT<NV>::FW + k * 8 + r] = fw; ft[FT<NV>::FQ + k * 8 + r] = fqx; ft[FT<NV>::FV + k * 8 + r] = fvx; }
      }
      Mww -= fw * rw; Mwq -= fw * rq; Mwv -= fw * rv2;
      Mqw -= fqx * rw; Mqq -= fqx * rq; Mqv -= fqx * rv2;
      Mvw -= fvx * rw; Mvq -= fvx * rq; Mvv -= fvx * rv2;
      if (r == k) rp_row = rp;
    };
    pivot(std::integral_constant<int, 0>()); pivot(std::integral_constant<int, 1>()); pivot(std::integral_constant<int, 2>());
    pivot(std::integral_constant<int, 3>()); pivot(std::integral_constant<int, 4>()); pivot(std::integral_constant<int, 5>());
    pivot(std::integral_constant<int, 6>());
    bad_pivot = bad_pivot || !(rp_row > 0.0);  // lane (r, *) keeps the reciprocal of pivot r (1.0 outside the block)
    if constexpr (STORE) {
      double *ft = facs + ((long long)b * T + t) * FT<NV>::SIZE;
      if (c == 0) ft[FT<NV>::RP + r] = rp_row;
    }
    // gains of this node: Kw = D^-1 [Mwq Mwv], kw = D^-1 qw
    if (!GAINS) {
      if (in) {
        Kw[(long long)t * NV * NX + r * NX + c] = Mwq * rp_row;
        Kw[(long long)t * NV * NX + r * NX + NV + c] = Mwv * rp_row;
      }
      if (gcol) kw[(long long)t * NV + r] = Mww * rp_row;
    } else {
      // u-space gains  K = M Kw - taux : row r of M against column c of Kw
      const double kq = Mwq * rp_row, kv = Mwv * rp_row;
      double Kq = -tq_rc, Kv = -tv_rc;
AGX_UNROLL_NV
      for (int l = 0; l < NV; ++l) {
        Kq += Mr_[l] * __shfl(kq, 8 * l + col_lane, 64);
        Kv += Mr_[l] * __shfl(kv, 8 * l + col_lane, 64);
      }
      if (in) {
        Ko[(long long)t * NV * NX + r * NX + c] = Kq;
        Ko[(long long)t * NV * NX + r * NX + NV + c] = Kv;
      }
    }
    // gradient of the value function of node t (lanes of the gradient column)
    vxq = Mqw; vxv = Mvw;
    // value function of node t
    Vqq = Mqq + dreg * diag; Vqv = Mqv; Vvq = Mvq; Vvv = Mvv + dreg * diag;  // (sigma is part of H at every node)
    if (!in) { Vqq = 0.0; Vqv = 0.0; Vvq = 0.0; Vvv = 0.0; }
    // refill this register set: after the last use of its old contents and unconditionally (node 0 again at the
    // end), so that no register copy -- and no s_waitcnt vmcnt(0) -- sits on the loop latch (see agx_riccati_mx.hpp)
    prefetch_group_begin();
    load_tile(z, t >= kGridDepth ? t - kGridDepth : 0);
    prefetch_group_end();
  };
  if (GAINS) { aux_fetch(T - 1); aux_put(T - 1); if (T >= 2) aux_fetch(T - 2); }
  int t = T - 1;
  for (int rem = T % kGridDepth; rem > 0; --rem, --t) {  // the nodes that do not fill a group, one at a time
    Tile z;
    load_tile(z, t);
    step(z, t);
  }
  if (t >= 0) {
    Tile tl[kGridDepth];
#pragma unroll
    for (int i = 0; i < kGridDepth; ++i) load_tile(tl[i], t - i);
    prefetch_queue_settle(tl);
    for (; t >= 0; t -= kGridDepth) {
#pragma unroll
      for (int i = 0; i < kGridDepth; ++i) step(tl[i], t - i);
      // The factorisation of a constrained QP (the sweeps that keep their factors: STORE) stops at a breakdown: the direction
      // is discarded whatever the remaining nodes give (k_sqp_head), the ADMM loop does not start (k_admm_reduce), and an
      // instance that sits inside the non-convex zone of a QuadExp cost breaks down in every SQP iteration of every MPC
      // step (the quorum-1.0 step is eight such iterations of three instances).
      if (STORE && !GAINS && __any(bad_pivot)) break;
    }
  }
  if (!GAINS) {
    // Quu of some node not positive definite (SolverCSQP: the LLT of the backward pass fails and the
    // direction is discarded): the step kernel rejects the step and raises the regularisation, the ADMM
    // loop stops for this instance
    const bool any_bad = __any(bad_pivot);
    if (lane == 0) {
      S.dir_fail = any_bad ? 1 : 0;
      if (any_bad) atomicOr(&S.flags, 1);  // atomic: the LQR pass of the same instance may raise it at the same time
    }
    if (STORE && any_bad) return;  // no forward pass on gains that were not computed
  }
  if (GAINS || !forward) return;
  riccati_forward<NV>(b, T, dts, qb, Kw, kw, dxs, wss, s_dt);
}

template <int NV, bool GAINS>
__global__ void __launch_bounds__(64) k_riccati(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                const double *__restrict__ qts, const double *__restrict__ auxs,
                                                double *__restrict__ Kws, double *__restrict__ kws,
                                                double *__restrict__ dxs, double *__restrict__ wss,
                                                double *__restrict__ dus, double *__restrict__ Kout,
                                                DevState *__restrict__ st, int forward, int gmode) {
  riccati_body<NV, GAINS>(blockIdx.x, op, dts, qts, auxs, Kws, kws, dxs, wss, dus, Kout, st, forward, gmode, 0);
}

// Direction sweep and speculative gains sweep of one SQP iteration in ONE launch: both are latency
// bound at one wave per SIMD, so the second wave rides along almost for free.  Even workgroups run
// the direction, odd ones the sigma sweep of the same instance; they share only read-only inputs.
template <int NV>
__global__ void __launch_bounds__(64) k_riccati_pair(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                     const double *__restrict__ qts, const double *__restrict__ auxs,
                                                     double *__restrict__ Kws, double *__restrict__ kws,
                                                     double *__restrict__ dxs, double *__restrict__ wss,
                                                     double *__restrict__ dus, double *__restrict__ Kout,
                                                     DevState *__restrict__ st, int iter) {
  const int b = blockIdx.x >> 1;
  if (blockIdx.x & 1)
    riccati_body<NV, true>(b, op, dts, qts, auxs, Kws, kws, dxs, wss, dus, Kout, st, 0, 1, iter);
  else
    riccati_body<NV, false>(b, op, dts, qts, auxs, Kws, kws, dxs, wss, dus, Kout, st, 1, 0, iter);
}

}  // namespace agx
#include "agx_riccati_mx.hpp"  // the same sweep in the MFMA operand layout (default for NV <= 7)
#include "agx_riccati_mx2.hpp" // ... cut into segments swept in parallel, exact boundary value functions (small batches)
namespace agx {

// ---------------------------------------------------------------------------
// K3: per-node quantities of the direction, node parallel with 8 lanes per node:
//   du = M w + taux dx  and the node's share of the KKT residual (through the QP-optimality
//   identity, mim_solvers checkKKTConditions), of the cost and of the gap norm:
//     Lu + Fu' lam' = -(Lxu' dx + (Luu + preg) du),   Lx + Fx' lam' - lam = -(Lxx dx + Lxu du + dreg dx)
// Lane j holds component j of w / dx and column j of every matrix row (one 64-byte line per row
// and node); the row sums run over the group as a DPP butterfly.  nodestat[node] = {kkt, cost, gap, 0}.
// ---------------------------------------------------------------------------
template <int NV>
__global__ void __launch_bounds__(256) k_node_kkt(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                  const double *__restrict__ auxs, const double *__restrict__ dxs,
                                                  const double *__restrict__ wss, double *__restrict__ dus,
                                                  double *__restrict__ nodestat, const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T;
  const int l8 = threadIdx.x & 7;
  const long long n_nodes = (long long)o.B * (T + 1);
  const long long node = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const bool ok = node < n_nodes;
  const long long nid = ok ? node : 0;
  const int b = (int)(nid / (T + 1)), t = (int)(nid % (T + 1));
  const DevState &S = st[b];
  const bool act = ok && !S.done;
  if (!__any(act)) return;  // wave of finished instances (no workgroup barrier below)
  const double preg = S.preg, dreg = S.dreg;
  const bool jl = l8 < NV;
  const int jj = jl ? l8 : 0;
  const double *qt = qts + nid * Q::SIZE;
  const double *ax = auxs + nid * A::SIZE;
  const double *dx = dxs + nid * NX;
  const double dq = jl ? dx[jj] : 0.0, dv = jl ? dx[NV + jj] : 0.0;
  double kkt = 0.0, gap = 0.0;
  if (t < T) {
    const double wj = jl ? wss[((long long)b * T + t) * NV + jj] : 0.0;
    const double fq = jl ? qt[Q::f + jj] : 0.0, fv = jl ? qt[Q::f + NV + jj] : 0.0;
    kkt = fmax(fabs(fq), fabs(fv));
    gap = fabs(fq) + fabs(fv);
    double pr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      pr[i] = (i < NV) ? ax[A::M + i * A::LD + l8] * wj + ax[A::tq + i * A::LD + l8] * dq + ax[A::tv + i * A::LD + l8] * dv : 0.0;
    const double du = transpose_reduce8(pr, l8);
    if (jl) {
      if (act) dus[((long long)b * T + t) * NV + l8] = du;
      kkt = fmax(kkt, fabs((ax[A::Luu + l8] + preg) * du));
    }
  }
  if (t > 0) {
    double pr[8];
AGX_UNROLL_NV
    for (int i = 0; i < 8; ++i) pr[i] = (i < NV) ? ax[A::Lqq + i * A::LD + l8] * dq : 0.0;
    const double hq = transpose_reduce8(pr, l8);
    if (jl) kkt = fmax(kkt, fmax(fabs(hq + dreg * dq), fabs((ax[A::Lvv + l8] + dreg) * dv)));
  }
  // group max / sum (fixed order: deterministic)
  kkt = fmax(kkt, dpp_xor4(kkt)); kkt = fmax(kkt, dpp_xor2(kkt)); kkt = fmax(kkt, dpp_xor1(kkt));
  gap += dpp_xor4(gap); gap += dpp_xor2(gap); gap += dpp_xor1(gap);
  if (act && l8 == 0) {
    double *ns = nodestat + nid * 4;
    ns[0] = kkt; ns[1] = qt[Q::cost]; ns[2] = gap;
    if (!o.has_con) ns[3] = 0.0;  // constrained problems: the violation share of k_con_eval (which may run before this kernel) stays
  }
}

// ---------------------------------------------------------------------------
// K4: the step of one SQP iteration (SolverCSQP::solve after computeDirection).
//
// THE LINE-SEARCH TRIAL IS THE NEXT DERIVATIVE PASS (k_sqp_head / k_sqp_accept, every model size).  Upstream a trial evaluates
// problem.calc at (xs + alpha dx, us + alpha du) and the accepted point is evaluated again, with derivatives, at the start of
// the next iteration.  Here k_sqp_head writes the trial iterate, the node-parallel derivative kernel (K1, and k_con_eval of
// constrained problems) runs AT THE TRIAL POINT, over the tiles of the current iterate, which nobody needs any more
// (direction, KKT shares and the speculative gains sweep have consumed them), and k_sqp_accept sums cost, gaps and
// violation out of the new tiles: merit_try < merit (or the filter test) -> the trial iterate becomes the iterate and its
// tiles are already there (tiles_ok: the next iteration's derivative pass skips the instance); otherwise the next step
// length goes through the same two launches (the host learns from a counter that somebody is still searching).  With
// alpha = 1 accepted -- every step of a warm-started MPC loop -- an SQP iteration costs ONE node evaluation instead of two,
// and no kernel holds a one-lane-per-node evaluation of a node any more (the step kernel of constrained problems used to:
// 512 VGPRs, 450 spilled registers; large models ran ten value-only trial launches per iteration).
// ---------------------------------------------------------------------------
// crocoddyl / mim_solvers regularisation schedule on the step length (th_stepdec 0.5, th_stepinc 0.01, factor 10)
__device__ __forceinline__ void reg_schedule(double used, double &pr, double &dr, bool &stop) {
  if (used > 0.5) { pr = fmax(pr / 10.0, kRegMin); dr = fmax(dr / 10.0, kRegMin); }
  stop = false;
  if (used <= 0.01) {
    pr = fmin(pr * 10.0, kRegMax);
    dr = fmin(dr * 10.0, kRegMax);
    if (pr == kRegMax) stop = true;
  }
}
// end of an SQP iteration of one instance (one thread): step of length `used` accepted (ok) or every trial rejected
__device__ __forceinline__ void sqp_iteration_end(DevState &S, bool ok, double used, int iter, int max_iter, int *n_done) {
  if (!ok) S.flags |= 2;
  double pr = S.preg, dr = S.dreg;
  // the gains on exit belong to the point this direction was computed at, with the regularisation
  // that was in force then: keep it in gains_preg / gains_dreg
  S.gains_preg = pr;
  S.gains_dreg = dr;
  bool stop;
  reg_schedule(used, pr, dr, stop);
  S.preg = pr;
  S.dreg = dr;
  S.dir_fail = 0;
  S.searching = 0;
  if (stop) {
    S.done = 1;
    S.iter = iter + 1;
    atomicAdd(n_done, 1);
  } else if (iter + 1 == max_iter) {
    S.iter = max_iter;
  } else if (!ok) {
    atomicAdd(n_done + 4, 1);  // never reset: the next iteration needs a derivative pass at the unchanged iterate (the host follows the growth)
  }
}
// trial iterate of step length alpha into the staging halves of xs / us (behind the B instances of the live buffers)
template <int NV>
__device__ __forceinline__ void write_trial_iterate(const DevOcp &o, int b, double alpha, const double *__restrict__ xs,
                                                    const double *__restrict__ us, const double *__restrict__ dxs,
                                                    const double *__restrict__ dus, double *__restrict__ xs_t, double *__restrict__ us_t) {
  constexpr int NX = 2 * NV, NU = NV;
  const int T = o.T;
  const long long ox = (long long)b * (T + 1) * NX, ou = (long long)b * T * NU;
  for (int e = threadIdx.x; e < (T + 1) * NX; e += blockDim.x) xs_t[ox + e] = xs[ox + e] + alpha * dxs[ox + e];
  for (int e = threadIdx.x; e < T * NU; e += blockDim.x) us_t[ou + e] = us[ou + e] + alpha * dus[ou + e];
}

// Hand-off to the host without a launch of its own (small batches, where every launch is ~ 4 us of a 200 us step): every
// workgroup of a kernel arrives at a counter when it is done, the last one stores the counters the host waits for, then the
// sequence stamp, into mapped pinned memory (as k_publish3).  hw.seq == nullptr: nothing -- large batches publish with a
// one-thread kernel (a thousand workgroups arriving at one counter cost the B = 1024 step more than the launch: measured).
// One thread per workgroup calls this, after its last write to the counters.
struct HostWords {
  unsigned long long *done, *handed, *stale, *seq;  // mapped host words (handed / stale may be null)
  unsigned long long stamp;
};
__device__ __forceinline__ void arrive_and_publish(int *__restrict__ counts, const HostWords &hw) {
  if (!hw.seq) return;
  __threadfence();
  const int a = atomicAdd(counts + 5, 1);
  if (a != (int)gridDim.x - 1) return;
  counts[5] = 0;  // the next kernel that arrives here starts from zero (kernels of one stream do not overlap)
  __threadfence();
  __hip_atomic_store(hw.done, (unsigned long long)(unsigned)__hip_atomic_load(counts + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (hw.handed) __hip_atomic_store(hw.handed, (unsigned long long)(unsigned)__hip_atomic_load(counts + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (hw.stale) __hip_atomic_store(hw.stale, (unsigned long long)(unsigned)__hip_atomic_load(counts + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(hw.seq, hw.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Head of the step: instance totals of the per-node KKT / cost / gap / violation shares (k_node_kkt, k_con_eval) in a
// fixed summation order -> KKT test; a converged instance finishes; the others enter the line search at alpha = 1 (trial
// iterate written here) -- unless their direction came out of a failed factorisation: that one is never accepted (the CPU
// restatement gets there through NaNs in the trial merit), all ten step lengths count as rejected at once.
// mode bit0: run the line search; without it only the totals (test hook).  mode bit2: timing mode (nothing committed).
// n_done[0]: finished instances.
template <int NV>
__global__ void __launch_bounds__(128) k_sqp_head(const DevOcp *__restrict__ op, const double *__restrict__ xs, const double *__restrict__ us,
                                                  const double *__restrict__ dxs, const double *__restrict__ dus,
                                                  double *__restrict__ xs_t, double *__restrict__ us_t,
                                                  const double *__restrict__ nodestat, DevState *__restrict__ st, int iter, int max_iter,
                                                  int mode, int *__restrict__ n_done, HostWords hw) {
  __shared__ double red[8];
  __shared__ int flag;
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x;
  DevState &S = st[b];
  if (S.done) {  // (uniform over the workgroup)
    if (tid == 0) arrive_and_publish(n_done, hw);
    return;
  }
  double kkt = 0.0, csum = 0.0, gsum = 0.0, vsum = 0.0;
  for (int t = tid; t <= T; t += blockDim.x) {
    const double *ns = nodestat + ((long long)b * (T + 1) + t) * 4;
    kkt = fmax(kkt, ns[0]);
    csum += ns[1];
    gsum += ns[2];
    if (o.has_con) vsum += ns[3];
  }
  kkt = wave_max(kkt);
  csum = wave_sum(csum);
  gsum = wave_sum(gsum);
  vsum = wave_sum(vsum);
  if ((tid & 63) == 0) { red[tid >> 6] = kkt; red[2 + (tid >> 6)] = csum; red[4 + (tid >> 6)] = gsum; red[6 + (tid >> 6)] = vsum; }
  __syncthreads();
  if (tid == 0) {
    double kk = fmax(red[0], red[1]);
    const double cc = red[2] + red[3], gg = red[4] + red[5], vv = red[6] + red[7];
    kk = fmax(kk, vv);  // checkKKTConditions: KKT = max(KKT, constraint_norm)
    const int dir_fail = S.dir_fail;
    if (dir_fail) kk = __builtin_nan("");  // discarded direction: its KKT residual is undefined (never "converged")
    S.kkt = kk; S.cost = cc; S.gap = gg; S.con = vv; S.merit = cc + o.mu_dyn * gg + o.mu_con * vv;
    S.qp_iters = o.has_con ? S.admm_iter : 1;
    S.admm_conv = 0;  // the next SQP iteration's plain LQR pass runs for this instance again
    if (!(mode & 4)) S.dir_iter = iter;
    if (!(kk == kk)) S.flags |= 1;
    const bool conv = (kk <= o.tol) && (mode & 1) && !(mode & 4);
    int what = 0;  // 0: nothing more, 1: line search starts
    if (conv) { S.solved = 1; S.done = 1; S.iter = iter; atomicAdd(n_done, 1); }
    else if ((mode & 1) && !(mode & 4)) {
      S.tiles_ok = 0;
      if (dir_fail) { S.flags |= 4; sqp_iteration_end(S, false, 1.0 / 512.0, iter, max_iter, n_done); }
      else {
        what = 1;
        S.searching = 1;
        S.ls_n = 0;
        double pr = S.preg, dr = S.dreg;
        bool stop;
        reg_schedule(1.0, pr, dr, stop);
        S.preg_trial = pr;
      }
    }
    flag = what;
    arrive_and_publish(n_done, hw);  // (the host wants the finished count; the trial iterate below is for the next kernel of the stream)
  }
  __syncthreads();
  if (flag) write_trial_iterate<NV>(o, b, 1.0, xs, us, dxs, dus, xs_t, us_t);
}

// After the derivative pass at the trial points: totals of the trial (cost and dynamics gaps out of the new tiles, violation
// shares of k_con_eval), merit test (the reference default: accept the first alpha = 2^-n with merit_try < merit) or
// the solver's filter of size 1 (rejected only if no better in cost AND gaps AND constraints), then
//   accepted: (xs, us) <- trial iterate, tiles_ok, regularisation schedule, end of the iteration;
//   rejected: next step length -> new trial iterate, counted in n_done[3]; after the tenth the iteration ends
//             with the step rejected and the tiles stale.
template <int NV>
__global__ void __launch_bounds__(128) k_sqp_accept(const DevOcp *__restrict__ op, double *__restrict__ xs, double *__restrict__ us,
                                                    const double *__restrict__ dxs, const double *__restrict__ dus,
                                                    double *__restrict__ xs_t, double *__restrict__ us_t,
                                                    const double *__restrict__ qts, const double *__restrict__ nodestat,
                                                    DevState *__restrict__ st, int iter, int max_iter, int *__restrict__ n_done,
                                                    HostWords hw) {
  constexpr int NX = 2 * NV, NU = NV;
  typedef QT<NV> Q;
  __shared__ double red[6];
  __shared__ int flag;
  __shared__ double s_alpha;
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x;
  DevState &S = st[b];
  if (!S.searching) {
    if (tid == 0) arrive_and_publish(n_done, hw);
    return;
  }
  double pc = 0.0, pg = 0.0, pv = 0.0;
  for (int t = tid; t <= T; t += blockDim.x) {
    const double *qt = qts + ((long long)b * (T + 1) + t) * Q::SIZE;
    pc += qt[Q::cost];
    double g = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) g += fabs(qt[Q::f + i]);  // the terminal tile carries zeros there
    pg += g;
    if (o.has_con) pv += nodestat[((long long)b * (T + 1) + t) * 4 + 3];
  }
  pc = wave_sum(pc); pg = wave_sum(pg); pv = wave_sum(pv);
  if ((tid & 63) == 0) { red[tid >> 6] = pc; red[2 + (tid >> 6)] = pg; red[4 + (tid >> 6)] = pv; }
  __syncthreads();
  if (tid == 0) {
    const double tc = red[0] + red[1], tg = red[2] + red[3], tv = red[4] + red[5];
    bool ok;
    if (o.use_filter) ok = !((S.cost <= tc) && (S.gap <= tg) && (S.con <= tv));
    else ok = S.merit > tc + o.mu_dyn * tg + o.mu_con * tv;
    const int n = S.ls_n;
    const double alpha = ldexp(1.0, -n);
    int what;  // 1: accepted, 2: next trial, 0: all ten rejected
    if (ok) {
      what = 1;
      S.tiles_ok = 1;
      sqp_iteration_end(S, true, alpha, iter, max_iter, n_done);
    } else if (n + 1 < 10) {
      what = 2;
      S.flags |= 4;  // a step length was rejected in this solve (the line search backtracked)
      S.ls_n = n + 1;
      double pr = S.preg, dr = S.dreg;
      bool stop;
      reg_schedule(0.5 * alpha, pr, dr, stop);
      S.preg_trial = pr;
      atomicAdd(n_done + 3, 1);  // never reset: the host follows its growth (trials handed on to the next round)
      s_alpha = 0.5 * alpha;
    } else {
      what = 0;
      sqp_iteration_end(S, false, alpha, iter, max_iter, n_done);
    }
    flag = what;
    arrive_and_publish(n_done, hw);  // (counters only: the iterate below is consumed by later kernels of the stream)
  }
  __syncthreads();
  if (flag == 1) {
    const long long ox = (long long)b * (T + 1) * NX, ou = (long long)b * T * NU;
    for (int e = tid; e < (T + 1) * NX; e += blockDim.x) xs[ox + e] = xs_t[ox + e];
    for (int e = tid; e < T * NU; e += blockDim.x) us[ou + e] = us_t[ou + e];
  } else if (flag == 2) {
    write_trial_iterate<NV>(o, b, s_alpha, xs, us, dxs, dus, xs_t, us_t);
  }
}

// ---------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------
__global__ void k_reset_state(DevState *st, int B, int *n_done) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) *n_done = 0;
  if (b >= B) return;
  DevState s;
  s.rho_sparse = st[b].rho_sparse; s.con = 0.0; s.admm_conv = 0; s.admm_iter = 0; s.ls_acc = 0; s.admm_refactor = 1;
  s.kkt = 0.0; s.cost = 0.0; s.merit = 0.0; s.gap = 0.0;
  s.preg = kRegMin; s.dreg = kRegMin;
  s.iter = 0; s.qp_iters = 0; s.solved = 0; s.flags = 0; s.done = 0; s.gains_iter = -1; s.dir_iter = -1; s.dir_fail = 0;
  s.searching = 0; s.ls_n = 0; s.tiles_ok = 0; s.pad_ls = 0; s.preg_trial = kRegMin;
  s.gains_preg = kRegMin; s.gains_dreg = kRegMin;
  st[b] = s;
}

// three words under one stamp: finished instances, line-search trials handed on, iterations ended with stale tiles
__global__ void k_publish3(const int *__restrict__ d_counts, unsigned long long *host_done, unsigned long long *host_handed,
                           unsigned long long *host_stale, unsigned long long *host_seq, unsigned long long seq) {
  __hip_atomic_store(host_done, (unsigned long long)(unsigned)d_counts[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_handed, (unsigned long long)(unsigned)d_counts[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_stale, (unsigned long long)(unsigned)d_counts[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// stream -> host hand-off through mapped pinned memory: value first, then the sequence stamp
__global__ void k_publish(const int *__restrict__ d_value, unsigned long long *host_value, unsigned long long *host_seq,
                          unsigned long long seq) {
  __hip_atomic_store(host_value, (unsigned long long)(unsigned)*d_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// first-node results of every instance, packed for one device-to-host transfer (agx_ocp_first_packed)
__global__ void k_pack_first(const double *__restrict__ us, const double *__restrict__ Kout, const double *__restrict__ xs,
                             const DevState *__restrict__ st, double *__restrict__ out, int B, int T, int NX, int NU,
                             int last_max_iter) {
  const int NK = NU * NX, FS = NU + NK + NX + 8;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * FS) return;
  const int b = (int)(i / FS), e = (int)(i % FS);
  double v;
  if (e < NU) v = us[(long long)b * T * NU + e];
  else if (e < NU + NK) v = Kout[(long long)b * T * NK + (e - NU)];
  else if (e < NU + NK + NX) v = xs[((long long)b * (T + 1) + 1) * NX + (e - NU - NK)];
  else {
    const DevState &S = st[b];
    switch (e - NU - NK - NX) {
      case 0: v = S.kkt; break;
      case 1: v = S.cost; break;
      case 2: v = S.merit; break;
      case 3: v = S.gap; break;
      case 4: v = S.done ? S.iter : last_max_iter; break;
      case 5: v = S.qp_iters; break;
      case 6: v = S.solved; break;
      default: v = S.flags; break;
    }
  }
  out[i] = v;
}

// a freshly constructed solver: rho back to its base value (the multipliers are zeroed by the host)
__global__ void k_reset_rho(DevState *st, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) st[b].rho_sparse = 0.0;
}

// xs[b][0] <- x0[b]   (SolverCSQP pins xs_[0] = problem.x0)
__global__ void k_pin_x0(double *xs, const double *x0, int B, int T, int NX) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * NX) return;
  const int b = i / NX, e = i % NX;
  xs[(long long)b * (T + 1) * NX + e] = x0[i];
}
// x0[b] <- xs[b][1]
__global__ void k_x0_from_pred(double *x0, const double *xs, int B, int T, int NX) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * NX) return;
  const int b = i / NX, e = i % NX;
  x0[i] = xs[((long long)b * (T + 1) + 1) * NX + e];
}

// WarmStartShiftPreviousSolution.shift (warm_start_shift_previous_solution.py:85-109):
// one lane per instance walks the horizon in order (the update is sequential in i
// because xs[i] <- xs[i+1] reads the not-yet-shifted neighbour).
template <int NV, bool CHAIN>
__global__ void k_shift(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op, const double *__restrict__ dts,
                        double *__restrict__ xs, double *__restrict__ us) {
  constexpr int NX = 2 * NV, NU = NV;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  // lanes over (instance, node): every node only reads node i and i+1 of the OLD solution,
  // so stage through registers and write after a barrier-free two-phase scheme per block.
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = unit < (long long)o.B * T;
  const int b = valid ? (int)(unit / T) : 0, i = valid ? (int)(unit % T) : 0;
  double xo[NX], uo[NU];
  const double dt0 = dts[0];
  if (valid) {
    double *X = xs + (long long)b * (T + 1) * NX, *U = us + (long long)b * T * NU;
    if (dts[i] == dt0) {
AGX_UNROLL_NV
      for (int e = 0; e < NX; ++e) xo[e] = X[(long long)(i + 1) * NX + e];
      const int iu = (i < T - 1) ? i + 1 : i;
AGX_UNROLL_NV
      for (int e = 0; e < NU; ++e) uo[e] = U[(long long)iu * NU + e];
    } else {
      double x[NX], u[NU], c;
AGX_UNROLL_NV
      for (int e = 0; e < NX; ++e) x[e] = X[(long long)i * NX + e];
#pragma unroll
      for (int e = 0; e < NU; ++e) { u[e] = U[(long long)i * NU + e]; uo[e] = u[e]; }
      DevRows none;
      none.n = 0;
      node_calc_running<NV, CHAIN>(m, none, dt0, x, u, nullptr, nullptr, xo, &c);
    }
  }
  // all reads of a block's nodes happen before its writes only within the block; a
  // node's source i+1 may belong to the next block, so the shifted copy goes to a
  // scratch buffer (xs_out/us_out) in the caller -- see k_shift_commit.
  if (valid) {
    double *Xo = xs + (long long)o.B * (T + 1) * NX;  // scratch region appended by the host allocator
    double *Uo = us + (long long)o.B * T * NU;
AGX_UNROLL_NV
    for (int e = 0; e < NX; ++e) Xo[((long long)b * (T + 1) + i) * NX + e] = xo[e];
AGX_UNROLL_NV
    for (int e = 0; e < NU; ++e) Uo[((long long)b * T + i) * NU + e] = uo[e];
  }
}
// The prologue of one resident MPC step in ONE launch (agx_ocp_mpc_step), one workgroup per instance:
//   x0 <- xs[1] of the previous solution (from_pred), the warm-start shift above (same arithmetic as
//   k_shift / k_shift_commit; the shifted nodes are staged in LDS because node i reads node i + 1),
//   xs[0] <- x0 (k_pin_x0) and the solver-state reset (k_reset_state).
// Five launches of 1-10 us each otherwise: 6 % of the latency of a batch-1 step.
template <int NV, bool CHAIN>
__global__ void __launch_bounds__(128) k_mpc_prologue(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                      const double *__restrict__ dts, double *__restrict__ xs,
                                                      double *__restrict__ us, double *__restrict__ x0s,
                                                      DevState *__restrict__ st, int *__restrict__ n_done, int from_pred) {
  constexpr int NX = 2 * NV, NU = NV, ND = NX + NU;
  extern __shared__ double sh_nodes[];  // [T][NX + NU]
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x;
  double *X = xs + (long long)b * (T + 1) * NX, *U = us + (long long)b * T * NU;
  double *x0 = x0s + (long long)b * NX;
  double x1 = 0.0;
  if (tid < NX) x1 = from_pred ? X[NX + tid] : x0[tid];
  const double dt0 = dts[0];
#pragma unroll 1
  for (int i = tid; i < T; i += blockDim.x) {
    double xo[NX], uo[NU];
    if (dts[i] == dt0) {
AGX_UNROLL_NV
      for (int e = 0; e < NX; ++e) xo[e] = X[(long long)(i + 1) * NX + e];
      const int iu = (i < T - 1) ? i + 1 : i;
AGX_UNROLL_NV
      for (int e = 0; e < NU; ++e) uo[e] = U[(long long)iu * NU + e];
    } else {
      double x[NX], u[NU], c;
AGX_UNROLL_NV
      for (int e = 0; e < NX; ++e) x[e] = X[(long long)i * NX + e];
#pragma unroll
      for (int e = 0; e < NU; ++e) { u[e] = U[(long long)i * NU + e]; uo[e] = u[e]; }
      DevRows none;
      none.n = 0;
      node_calc_running<NV, CHAIN>(m, none, dt0, x, u, nullptr, nullptr, xo, &c);
    }
    double *d = sh_nodes + (long long)i * ND;
AGX_UNROLL_NV
    for (int e = 0; e < NX; ++e) d[e] = xo[e];
AGX_UNROLL_NV
    for (int e = 0; e < NU; ++e) d[NX + e] = uo[e];
  }
  __syncthreads();
  for (int k = tid; k < T * NX; k += blockDim.x) X[k] = sh_nodes[(k / NX) * ND + (k % NX)];
  for (int k = tid; k < T * NU; k += blockDim.x) U[k] = sh_nodes[(k / NU) * ND + NX + (k % NU)];
  __syncthreads();
  if (tid < NX) {
    if (from_pred) x0[tid] = x1;
    X[tid] = x1;
  }
  if (tid == 0) {
    if (b == 0) *n_done = 0;
    DevState s;
    s.rho_sparse = st[b].rho_sparse; s.con = 0.0; s.admm_conv = 0; s.admm_iter = 0; s.ls_acc = 0; s.admm_refactor = 1;
    s.kkt = 0.0; s.cost = 0.0; s.merit = 0.0; s.gap = 0.0;
    s.preg = kRegMin; s.dreg = kRegMin;
    s.iter = 0; s.qp_iters = 0; s.solved = 0; s.flags = 0; s.done = 0; s.gains_iter = -1; s.dir_iter = -1; s.dir_fail = 0;
  s.searching = 0; s.ls_n = 0; s.tiles_ok = 0; s.pad_ls = 0; s.preg_trial = kRegMin;
    s.gains_preg = kRegMin; s.gains_dreg = kRegMin;
    st[b] = s;
  }
}
__global__ void k_shift_commit(double *xs, double *us, int B, int T, int NX, int NU) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nxs = (long long)B * (T + 1) * NX, nus = (long long)B * T * NU;
  if (i < nxs) {
    const long long node = (i / NX) % (T + 1);
    if (node < T) xs[i] = xs[nxs + i];
  }
  if (i < nus) us[i] = us[nus + i];
}

template <int NV, bool CHAIN>
__global__ void k_integrate(const DevModel *__restrict__ mp, double dt, int n, const double *__restrict__ x,
                            const double *__restrict__ u, double *__restrict__ xnext) {
  constexpr int NX = 2 * NV, NU = NV;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double xl[NX], ul[NU], xn[NX], c;
AGX_UNROLL_NV
  for (int e = 0; e < NX; ++e) xl[e] = x[(long long)i * NX + e];
AGX_UNROLL_NV
  for (int e = 0; e < NU; ++e) ul[e] = u[(long long)i * NU + e];
  DevRows none;
  none.n = 0;
  node_calc_running<NV, CHAIN>(*mp, none, dt, xl, ul, nullptr, nullptr, xn, &c);
AGX_UNROLL_NV
  for (int e = 0; e < NX; ++e) xnext[(long long)i * NX + e] = xn[e];
}

// The consumer of an MPC step (linear feedback controller behind AgimusController.send_control_msg,
// agimus_controller_ros/agimus_controller.py:418-426): between two MPC steps the robot is driven at
// the control rate with  u = us[0] + K[0] (x0 - x_measured).  Here the "robot" is the model itself:
// n_sub semi-implicit Euler steps of dt_sub per instance, optionally with a constant torque
// disturbance, starting at x0; the end state becomes the next measured state x0.
template <int NV, bool CHAIN>
__global__ void k_feedback_rollout(const DevModel *__restrict__ mp, const double *__restrict__ us, const double *__restrict__ Kout,
                                   double *__restrict__ x0, const double *__restrict__ disturbance, int B, int T, int n_sub,
                                   double dt_sub) {
  constexpr int NX = 2 * NV, NU = NV;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xr[NX], x[NX], u0[NU], K[NU][NX], tau_d[NU];
#pragma unroll
  for (int e = 0; e < NX; ++e) { xr[e] = x0[(long long)b * NX + e]; x[e] = xr[e]; }
AGX_UNROLL_NV
  for (int i = 0; i < NU; ++i) {
    u0[i] = us[(long long)b * T * NU + i];
    tau_d[i] = disturbance ? disturbance[(long long)b * NU + i] : 0.0;
AGX_UNROLL_NV
    for (int e = 0; e < NX; ++e) K[i][e] = Kout[(long long)b * T * NU * NX + i * NX + e];
  }
  DevRows none;
  none.n = 0;
  for (int s = 0; s < n_sub; ++s) {
    double u[NU], xn[NX], c;
AGX_UNROLL_NV
    for (int i = 0; i < NU; ++i) {
      double acc = u0[i] + tau_d[i];
#pragma unroll
      for (int e = 0; e < NX; ++e) acc += K[i][e] * (xr[e] - x[e]);
      u[i] = acc;
    }
    node_calc_running<NV, CHAIN>(*mp, none, dt_sub, x, u, nullptr, nullptr, xn, &c);
#pragma unroll
    for (int e = 0; e < NX; ++e) x[e] = xn[e];
  }
AGX_UNROLL_NV
  for (int e = 0; e < NX; ++e) x0[(long long)b * NX + e] = x[e];
}

template <int NV, bool CHAIN>
__global__ void k_rnea(const DevModel *__restrict__ mp, int n, const double *__restrict__ q, const double *__restrict__ v,
                       const double *__restrict__ a, double *__restrict__ tau) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double ql[NV], vl[NV], al[NV], tl[NV];
AGX_UNROLL_NV
  for (int e = 0; e < NV; ++e) { ql[e] = q[(long long)i * NV + e]; vl[e] = v[(long long)i * NV + e]; al[e] = a[(long long)i * NV + e]; }
  Kin<NV> k;
  kinematics<NV, CHAIN>(*mp, ql, k);
  rnea<NV, CHAIN>(*mp, k, vl, al, tl);
AGX_UNROLL_NV
  for (int e = 0; e < NV; ++e) tau[(long long)i * NV + e] = tl[e];
}

template <int NV, bool CHAIN>
__global__ void k_frame(const DevModel *__restrict__ mp, int n, int frame, const double *__restrict__ q, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double ql[NV];
AGX_UNROLL_NV
  for (int e = 0; e < NV; ++e) ql[e] = q[(long long)i * NV + e];
  Kin<NV> k;
  kinematics<NV, CHAIN>(*mp, ql, k);
  double R[9], p[3];
  int jf;
  frame_world<NV>(*mp, k, frame, R, p, &jf);
#pragma unroll
  for (int e = 0; e < 9; ++e) out[(long long)i * 12 + e] = R[e];
#pragma unroll
  for (int e = 0; e < 3; ++e) out[(long long)i * 12 + 9 + e] = p[e];
}

// Frame Jacobian 6 x nv (pinocchio getFrameJacobian): rows linear | angular, LOCAL_WORLD_ALIGNED
// (local = 0: axes of the world, origin at the frame) or LOCAL (local = 1: the frame's own axes); J row-major [6][NV].
template <int NV, bool CHAIN>
AGX_DEV void frame_jacobian_rows(const DevModel &m, const Kin<NV> &k, const double *R, const double *p, const int jf, const int local,
                                 double *J) {
AGX_UNROLL_NV
  for (int j = 0; j < NV; ++j) {
    const bool on = (jf >= 0) && (CHAIN ? (j <= jf) : ((m.anc[jf >= 0 ? jf : 0] >> j) & 1u));
    double d[3], lin[3], ang[3] = {k.S[j][3], k.S[j][4], k.S[j][5]};
    d[0] = p[0] - k.p[j][0]; d[1] = p[1] - k.p[j][1]; d[2] = p[2] - k.p[j][2];
    cross3(ang, d, lin);
    if (local) {
      double l2[3], a2[3];
      mtv3(R, lin, l2);
      mtv3(R, ang, a2);
      lin[0] = l2[0]; lin[1] = l2[1]; lin[2] = l2[2];
      ang[0] = a2[0]; ang[1] = a2[1]; ang[2] = a2[2];
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      J[e * NV + j] = on ? lin[e] : 0.0;
      J[(3 + e) * NV + j] = on ? ang[e] : 0.0;
    }
  }
}
template <int NV, bool CHAIN>
__global__ void k_frame_jacobian(const DevModel *__restrict__ mp, int n, int frame, int local, const double *__restrict__ q,
                                 double *__restrict__ out) {
  const DevModel &m = *mp;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double ql[NV];
AGX_UNROLL_NV
  for (int e = 0; e < NV; ++e) ql[e] = q[(long long)i * NV + e];
  Kin<NV> k;
  kinematics<NV, CHAIN>(m, ql, k);
  double R[9], p[3];
  int jf;
  frame_world<NV>(m, k, frame, R, p, &jf);
  double J[6 * NV];
  frame_jacobian_rows<NV, CHAIN>(m, k, R, p, jf, local, J);
#pragma unroll
  for (int e = 0; e < 6 * NV; ++e) out[(long long)i * 6 * NV + e] = J[e];
}

// ---------------------------------------------------------------------------
// SinusWaveCartesianSpace on the device (trajectories/sine_wave_cartesian_space.py:62-111 upstream): the end effector
// `frame` of instance b follows  p0 + A s(t) sin(w t)  with its initial orientation.  One lane per instance walks the
// points in order (the inverse kinematics of a point starts from the solution of the previous one, exactly as the
// class keeps `ik_q`): Newton steps  q <- q - J' (J J')^-1 log6(des^-1 cur)  with the LOCAL Jacobian until the error
// norm is below `precision`, then  dq = J' (J J')^-1 v_des  with the LOCAL_WORLD_ALIGNED Jacobian; accelerations are
// zero.  All six components of the pose error are used (the class's default mask).  fail[b] = 1 + index of the first
// point whose iteration did not converge within it_max steps (0: none).
// ---------------------------------------------------------------------------
struct CartSineParams {
  const double *q0, *amp, *puls;  // [B][nv], [B][3], [B][3]
  double dt, scale, precision;
  int n_points, frame, it_max;
  double *q, *dq;                 // [B][n_points][nv]
  double *pose;                   // [B][n_points][12]: the DESIRED end-effector pose (R row major | p), the reference the points carry
  int *fail;                      // [B]
};
// x <- A^-1 x for a symmetric positive definite 6 x 6 (J J'): elimination without pivoting
AGX_DEV void solve_spd6(double *A, double *x) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double rp = 1.0 / A[7 * k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const double f = A[6 * i + k] * rp;
#pragma unroll
      for (int j = k + 1; j < 6; ++j) A[6 * i + j] -= f * A[6 * k + j];
      x[i] -= f * x[k];
    }
  }
#pragma unroll
  for (int k = 5; k >= 0; --k) {
    double s = x[k];
#pragma unroll
    for (int j = k + 1; j < 6; ++j) s -= A[6 * k + j] * x[j];
    x[k] = s / A[7 * k];
  }
}
// y = J' (J J')^-1 r  (J row-major [6][NV])
template <int NV>
AGX_DEV void pinv_apply(const double *J, const double *r, double *y) {
  double A[36], x[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    x[i] = r[i];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double s = 0.0;
AGX_UNROLL_NV
      for (int c = 0; c < NV; ++c) s += J[i * NV + c] * J[j * NV + c];
      A[6 * i + j] = s;
    }
  }
  solve_spd6(A, x);
AGX_UNROLL_NV
  for (int c = 0; c < NV; ++c) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += J[i * NV + c] * x[i];
    y[c] = s;
  }
}
template <int NV, bool CHAIN>
__global__ void __launch_bounds__(64) k_cartesian_sine_ik(const DevModel *__restrict__ mp, int B, CartSineParams cp) {
  const DevModel &m = *mp;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double q[NV];
AGX_UNROLL_NV
  for (int e = 0; e < NV; ++e) q[e] = cp.q0[(long long)b * NV + e];
  double amp[3], w[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) { amp[e] = cp.amp[3 * b + e]; w[e] = cp.puls[3 * b + e]; }
  Kin<NV> k;
  double R0[9], p0[3], R[9], p[3], J[6 * NV];
  int jf;
  kinematics<NV, CHAIN>(m, q, k);
  frame_world<NV>(m, k, cp.frame, R0, p0, &jf);
  int failed = 0;
  for (int i = 0; i < cp.n_points; ++i) {
    const double t = i * cp.dt;
    const double s = fmin(fmax(t / cp.scale, 0.0), 1.0);
    const double quint = ((6.0 * s - 15.0) * s + 10.0) * s * s * s;
    const double dquint = (0.0 < t && t < cp.scale) ? ((30.0 * s - 60.0) * s + 30.0) * s * s / cp.scale : 0.0;
    double des_p[3], des_v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      double sn, cs;
      sincos(w[e] * t, &sn, &cs);
      des_p[e] = p0[e] + amp[e] * quint * sn;
      des_v[e] = amp[e] * (dquint * sn + quint * w[e] * cs);
    }
    for (int it = 0;; ++it) {
      kinematics<NV, CHAIN>(m, q, k);
      frame_world<NV>(m, k, cp.frame, R, p, &jf);
      double Rrel[9], d[3], prel[3], err[6];
      mtm3(R0, R, Rrel);  // des^-1 * cur (the desired orientation is the initial one)
      d[0] = p[0] - des_p[0]; d[1] = p[1] - des_p[1]; d[2] = p[2] - des_p[2];
      mtv3(R0, d, prel);
      log6<false>(Rrel, prel, err, nullptr, nullptr);
      if (sqrt(dot6(err, err)) < cp.precision) break;
      if (it > cp.it_max) { if (!failed) failed = i + 1; break; }  // upstream: `if i > it_max: break` (sine_wave_cartesian_space.py:80)
      frame_jacobian_rows<NV, CHAIN>(m, k, R, p, jf, 1, J);
      double dq[NV];
      pinv_apply<NV>(J, err, dq);
AGX_UNROLL_NV
      for (int e = 0; e < NV; ++e) q[e] -= dq[e];
    }
    frame_jacobian_rows<NV, CHAIN>(m, k, R, p, jf, 0, J);
    double dq[NV];
    pinv_apply<NV>(J, des_v, dq);
    const long long o = ((long long)b * cp.n_points + i) * NV;
AGX_UNROLL_NV
    for (int e = 0; e < NV; ++e) { cp.q[o + e] = q[e]; cp.dq[o + e] = dq[e]; }
    // the point's end-effector reference is the desired pose ee_des_pos (sine_wave_cartesian_space.py:126-133), not FK(q_ik)
    double *po = cp.pose + ((long long)b * cp.n_points + i) * 12;
#pragma unroll
    for (int e = 0; e < 9; ++e) po[e] = R0[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) po[9 + e] = des_p[e];
  }
  cp.fail[b] = failed;
}

// residual vector of one running row at the resident solution (debug data,
// ocp_croco_generic.py:840-853): out [B][T][nr]
template <int NV, bool CHAIN>
__global__ void k_residuals(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op, const double *__restrict__ xs,
                            const double *__restrict__ us, RefView rv, int row, double *__restrict__ out) {
  constexpr int NX = 2 * NV, NU = NV;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (unit >= (long long)o.B * T) return;
  const int b = (int)(unit / T), t = (int)(unit % T);
  const DevRows &rows = o.rows[0];
  const int nr = rows.nr[row], kind = rows.kind[row];
  const double *x = xs + ((long long)b * (T + 1) + t) * NX, *u = us + ((long long)b * T + t) * NU;
  const double *rr = ref_at(rv, b, t, T) + rows.off[row] + 1;
  double *dst = out + unit * nr;
  if (kind == AGX_RES_STATE) {
    for (int i = 0; i < NX; ++i) dst[i] = x[i] - rr[i];
  } else if (kind == AGX_RES_CONTROL) {
    for (int i = 0; i < NU; ++i) dst[i] = u[i] - rr[i];
  } else if (kind == AGX_RES_FRAME_PLACEMENT || kind == AGX_RES_FRAME_TRANSLATION || kind == AGX_RES_FRAME_ROTATION) {
    double ql[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) ql[e] = x[e];
    Kin<NV> k;
    kinematics<NV, CHAIN>(m, ql, k);
    const int *fr = frames_at(rv, b, t, T);
    int frame = fr ? fr[row] : -1;
    if (frame < 0) frame = rows.frame[row];
    double RF[9], pF[3];
    int jf;
    frame_world<NV>(m, k, frame, RF, pF, &jf);
    if (kind == AGX_RES_FRAME_PLACEMENT) {
      double Rrel[9], d[3], prel[3], res[6];
      mtm3(rr, RF, Rrel);
      d[0] = pF[0] - rr[9]; d[1] = pF[1] - rr[10]; d[2] = pF[2] - rr[11];
      mtv3(rr, d, prel);
      log6<false>(Rrel, prel, res, nullptr, nullptr);
      for (int e = 0; e < 6; ++e) dst[e] = res[e];
    } else if (kind == AGX_RES_FRAME_TRANSLATION) {
      for (int e = 0; e < 3; ++e) dst[e] = pF[e] - rr[e];
    } else {
      double Rrel[9], res[3];
      mtm3(rr, RF, Rrel);
      log3(Rrel, res);
      for (int e = 0; e < 3; ++e) dst[e] = res[e];
    }
  } else if (kind == AGX_RES_COLLISION) {
    double ql[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) ql[e] = x[e];
    Kin<NV> k;
    kinematics<NV, CHAIN>(m, ql, k);
    double ca[3], cb[3], n[3];
    int ja, jb;
    dst[0] = collision_distance<NV>(m, k, rows.frame[row], rows.frame_b[row], ca, cb, n, &ja, &jb);
  } else if (kind == AGX_RES_FRAME_VELOCITY) {
    double xl[NX];
    for (int e = 0; e < NX; ++e) xl[e] = x[e];
    Kin<NV> k;
    kinematics<NV, CHAIN>(m, xl, k);
    const int *fr = frames_at(rv, b, t, T);
    int frame = fr ? fr[row] : -1;
    if (frame < 0) frame = rows.frame[row];
    double vel[6];
    frame_velocity<NV, CHAIN, false>(m, k, frame, rows.frame_b[row], xl + NV, vel, nullptr, nullptr);
    for (int e = 0; e < 6; ++e) dst[e] = vel[e] - rr[e];
  } else if (kind == AGX_RES_CONTROL_GRAV) {
    double xl[NX], g0[NV], zero[NV], M0[NV][NV];
    for (int e = 0; e < NX; ++e) xl[e] = x[e];
    for (int e = 0; e < NV; ++e) zero[e] = 0.0;
    Kin<NV> k;
    kinematics<NV, CHAIN>(m, xl, k);
    Dyn<NV> d0;
    bias_and_inertia<NV, CHAIN>(m, k, zero, d0, g0, M0);
    for (int i = 0; i < NU; ++i) dst[i] = u[i] - g0[i];
  } else {
    for (int i = 0; i < nr; ++i) dst[i] = 0.0;
  }
}

// ---------------------------------------------------------------------------
// Device-resident reference trajectory: sine wave in configuration space
// (trajectories/sine_wave_configuration_space.py:41-72 with the quintic ramp of
// trajectories/quintic_trajectory.py:34-40).  One lane per (instance, sample).
// Every sample stores the running-layout tile followed by the terminal-layout
// tile (2*stride doubles) and the raw point [q v a u pose] in pts.
// ---------------------------------------------------------------------------
struct SineParams {
  const double *q0, *amp, *puls, *scale, *t0;  // [B][nv] (t0: [B])
  // generic trajectory (trajectories/generic_trajectory.py:37-70): samples given by the caller,
  // [B][n_points][nv] each; when set they replace the sine formula
  const double *gq, *gdq, *gddq;
  const double *gpose;  // optional [B][n_points][12]: end-effector reference of every sample (otherwise the pose of the sample's q)
  double w_q[AGX_MAX_NV], w_qdot[AGX_MAX_NV], w_effort[AGX_MAX_NV], w_pose[6];
  double dt;
  int n_points, frame;
};

template <int NV, bool CHAIN>
__global__ void k_sine_fill(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op, SineParams sp,
                            double *__restrict__ traj, double *__restrict__ pts) {
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (unit >= (long long)o.B * sp.n_points) return;
  const int b = (int)(unit / sp.n_points), kk = (int)(unit % sp.n_points);
  const double t = sp.t0[b] + kk * sp.dt;
  double q[NV], dq[NV], ddq[NV], u[NV];
  if (sp.gq) {
AGX_UNROLL_NV
    for (int i = 0; i < NV; ++i) { q[i] = sp.gq[unit * NV + i]; dq[i] = sp.gdq[unit * NV + i]; ddq[i] = sp.gddq[unit * NV + i]; }
  } else
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    const double sd = sp.scale[(long long)b * NV + i], w = sp.puls[(long long)b * NV + i], A = sp.amp[(long long)b * NV + i];
    double p5, v5, a5;
    if (t <= 0.0) { p5 = 0.0; v5 = 0.0; a5 = 0.0; }
    else if (t >= sd) { p5 = 1.0; v5 = 0.0; a5 = 0.0; }
    else {
      const double s = t / sd, s2 = s * s, s3 = s2 * s;
      p5 = 10.0 * s3 - 15.0 * s3 * s + 6.0 * s3 * s2;
      v5 = (30.0 * s2 - 60.0 * s3 + 30.0 * s3 * s) / sd;
      a5 = (60.0 * s - 180.0 * s2 + 120.0 * s3) / (sd * sd);
    }
    double sw, cw;
    sincos(w * t, &sw, &cw);
    q[i] = sp.q0[(long long)b * NV + i] + A * p5 * sw;
    dq[i] = A * (v5 * sw + p5 * w * cw);
    ddq[i] = A * (a5 * sw + 2.0 * v5 * w * cw - p5 * w * w * sw);
  }
  Kin<NV> k;
  kinematics<NV, CHAIN>(m, q, k);
  rnea<NV, CHAIN>(m, k, dq, ddq, u);
  double RF[9], pF[3];
  int jf;
  frame_world<NV>(m, k, sp.frame, RF, pF, &jf);
  if (sp.gpose) {
#pragma unroll
    for (int e = 0; e < 9; ++e) RF[e] = sp.gpose[unit * 12 + e];
#pragma unroll
    for (int e = 0; e < 3; ++e) pF[e] = sp.gpose[unit * 12 + 9 + e];
  }
  double *pt = pts + unit * (4 * NV + 12);
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) { pt[i] = q[i]; pt[NV + i] = dq[i]; pt[2 * NV + i] = ddq[i]; pt[3 * NV + i] = u[i]; }
AGX_UNROLL_NV
  for (int e = 0; e < 9; ++e) pt[4 * NV + e] = RF[e];
AGX_UNROLL_NV
  for (int e = 0; e < 3; ++e) pt[4 * NV + 9 + e] = pF[e];
  for (int layout = 0; layout < 2; ++layout) {
    const DevRows &rows = o.rows[layout];
    double *tile = traj + unit * 2 * o.stride + layout * o.stride;
    for (int r = 0; r < rows.n; ++r) {
      double *tr = tile + rows.off[r];
      tr[0] = rows.weight[r];
      double *rr = tr + 1, *aw = rr + rows.nref[r];
      const int kind = rows.kind[r];
      if (kind == AGX_RES_STATE) {
        for (int i = 0; i < NV; ++i) { rr[i] = q[i]; rr[NV + i] = dq[i]; aw[i] = sp.w_q[i]; aw[NV + i] = sp.w_qdot[i]; }
      } else if (kind == AGX_RES_CONTROL) {
        for (int i = 0; i < NV; ++i) { rr[i] = u[i]; aw[i] = sp.w_effort[i]; }
      } else if (kind == AGX_RES_FRAME_PLACEMENT) {
        for (int e = 0; e < 9; ++e) rr[e] = RF[e];
        for (int e = 0; e < 3; ++e) rr[9 + e] = pF[e];
        for (int e = 0; e < 6; ++e) aw[e] = sp.w_pose[e];
      } else if (kind == AGX_RES_FRAME_TRANSLATION) {
        for (int e = 0; e < 3; ++e) { rr[e] = pF[e]; aw[e] = sp.w_pose[e]; }
      } else if (kind == AGX_RES_FRAME_ROTATION) {
        for (int e = 0; e < 9; ++e) rr[e] = RF[e];
        for (int e = 0; e < 3; ++e) aw[e] = sp.w_pose[3 + e];
      } else {
        for (int e = 0; e < rows.nref[r] + rows.nr[r]; ++e) rr[e] = 0.0;
      }
    }
  }
}

// Horizon window with non-uniform sample indexes (TrajectoryBuffer.horizon_indexes, trajectory.py:181-231:
// node t looks at sample k0 + hidx[t], e.g. [0,1,2,4,6,9,...] for dt factors 1,2,3): gathered into the
// handle's own tile [B][T+1][stride]; the terminal node takes the terminal layout of its sample.
__global__ void k_gather_window(const double *__restrict__ traj, double *__restrict__ ref, const int *__restrict__ hidx, int B, int T,
                                int stride, int n_points, int k0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per_b = (long long)(T + 1) * stride;
  if (i >= (long long)B * per_b) return;
  const int b = (int)(i / per_b);
  const int t = (int)((i % per_b) / stride), e = (int)(i % stride);
  const double *src = traj + ((long long)b * n_points + k0 + hidx[t]) * 2 * stride + (t == T ? stride : 0);
  ref[i] = src[e];
}

// WarmStartReference.generate on the device (warm_start_reference.py:33-96):
// xs[t] = ref state of sample k0+t (xs[0] = x0), us[t] = ref effort of sample k0+t
// (us[0] = RNEA at the measured state would need its acceleration: the reference
// passes initial_state.robot_acceleration; the resident trajectory uses sample k0's).
__global__ void k_ws_from_ref(double *xs, double *us, double *x0, const double *pts, int B, int T, int NV, int n_points, int k0, int set_x0,
                              const int *__restrict__ hidx) {
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (unit >= (long long)B * (T + 1)) return;
  const int b = (int)(unit / (T + 1)), t = (int)(unit % (T + 1));
  const double *pt = pts + ((long long)b * n_points + k0 + (hidx ? hidx[t] : t)) * (4 * NV + 12);
  const int NX = 2 * NV;
  for (int i = 0; i < NX; ++i) xs[unit * NX + i] = pt[i];
  if (t < T)
    for (int i = 0; i < NV; ++i) us[((long long)b * T + t) * NV + i] = pt[3 * NV + i];
  if (t == 0 && set_x0)
    for (int i = 0; i < NX; ++i) x0[(long long)b * NX + i] = pt[i];
}

}  // namespace agx

#include "agx_k1_lanes.hpp"
#include "agx_admm.hpp"
#include "agx_big.hpp"
#include "agx_big_k1.hpp"
#include "agx_big_k2.hpp"
