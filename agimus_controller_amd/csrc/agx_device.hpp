// agx_device.hpp -- device-side rigid-body dynamics and shooting-node evaluation
// for gfx950 (MI355X).  Hand-derived world-frame analytical derivatives; the CPU
// checker under oracle/ uses link-local recursions + automatic differentiation
// instead, so the two share no derivation.
//
// What this replaces (third-party, reached through Python bindings upstream):
//   pinocchio computeAllTerms / Cholesky Minv / computeRNEADerivatives / frame
//   Jacobians / log6 / Jlog6, and crocoddyl DifferentialActionModelFreeFwdDynamics
//   + IntegratedActionModelEuler + CostModelSum (ocp_croco_generic.py:688-711,743-745).
//
// Spatial vectors are [linear(3); angular(3)], expressed in the WORLD frame.
#pragma once

#ifdef AGX_HOST_BUILD
// The analytic-derivative CPU leg of bench.py's cpu_baseline (agx_analytic.cpp under oracle/, test / bench infrastructure)
// compiles the per-node arithmetic of this header for the host: plain C++ below, the cross-lane helpers are left out.
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "../../include/agimus_hip.h"

#define AGX_MAX_FRAMES 72

struct DevModel {
  int nv, nframes, is_chain, pad;
  int parent[AGX_MAX_NV];
  unsigned anc[AGX_MAX_NV];  // bit j: joint j is i itself or an ancestor of i
  unsigned desc[AGX_MAX_NV]; // bit j: joint j is i itself or a descendant of i (subtree of i)
  double placement[AGX_MAX_NV][12];
  double axis[AGX_MAX_NV][3];
  double mass[AGX_MAX_NV];
  double com[AGX_MAX_NV][3];
  double inertia[AGX_MAX_NV][9];
  double armature[AGX_MAX_NV];
  double gravity[3];
  int frame_parent[AGX_MAX_FRAMES];
  double frame_placement[AGX_MAX_FRAMES][12];
  double frame_radius[AGX_MAX_FRAMES], frame_halflen[AGX_MAX_FRAMES];  // collision geometry carried by frames
  double frame_box[AGX_MAX_FRAMES][3];                                 // box half extents (all 0: capsule / sphere)
};

struct DevRows {
  int n;
  int kind[AGX_MAX_ROWS], act[AGX_MAX_ROWS], active[AGX_MAX_ROWS], frame[AGX_MAX_ROWS], frame_b[AGX_MAX_ROWS];
  int off[AGX_MAX_ROWS], nref[AGX_MAX_ROWS], nr[AGX_MAX_ROWS];
  double alpha[AGX_MAX_ROWS], weight[AGX_MAX_ROWS];
  int general;  // some active row is a ControlGrav / FrameVelocity residual (agx_general.hpp)
  int nvu;      // joints of the caller's model: the pad joints of a model below the compiled capacity take no part in an Exp / QuadExp
                // activation of a State / Control row (their second derivative is set to 1, as the unit weights of the quadratic rows)
};

// ConstraintListItem rows of one node type:  lb <= g(x, u) <= ub, g stacked over the rows.
#define AGX_MAX_CONS 4
#define AGX_MAX_DENSE 8  // constraint components with a dense Jacobian in q per node type (collision 1, translation / rotation 3, placement 6)
// constraint components per node: 32 for the 7-joint capacity (its constraint kernels keep them in registers), state bounds +
// control limits of a 32-joint model + dense rows for the workgroup path (2 x 32 + 32 + 8)
#if (defined(AGX_GROUP) && AGX_GROUP == 0) || defined(AGX_ONLY_NV7)
#define AGX_MAX_NC 32
#else
#define AGX_MAX_NC 104
#endif
struct DevCons {
  int n, nc, ncoll, pad;  // rows, components, Jacobian slots in use (ncoll: historically the collision rows)
  int kind[AGX_MAX_CONS], frame[AGX_MAX_CONS], frame_b[AGX_MAX_CONS], off[AGX_MAX_CONS], nr[AGX_MAX_CONS];
  int coll_slot[AGX_MAX_CONS];  // first Jacobian slot (of AGX_MAX_DENSE) of a row whose components have dense gradients in q
  double ref[AGX_MAX_CONS][2 * AGX_MAX_NV];
  double lb[AGX_MAX_NC], ub[AGX_MAX_NC];
};

struct DevOcp {
  int T, B, stride, pad;
  DevRows rows[2];  // 0 running, 1 terminal
  double tol, mu_dyn, mu_con;
  DevCons cons[2];
  int max_qp, has_con;
  double eps_abs, eps_rel;
  int use_filter, pad2;  // SolverCSQP.use_filter_line_search (ocp_param_base.py:64)
};

#define AGX_DEV __device__ __forceinline__
#define AGX_HD __host__ __device__ __forceinline__
// loops over the joints: fully unrolled for the register-resident sizes (nv <= 8), rolled for
// large models (nv = 30: per-lane arrays live in scratch, the code must stay small)
#define AGX_UNROLL_NV _Pragma("clang loop unroll_count(NV <= 8 ? 64 : 1)")

namespace agx {

#ifndef AGX_HOST_BUILD
// ------------------------------------------------------------------ DPP helpers (8-lane groups)
// Cross-lane moves on the VALU (no LDS traffic).  A DPP row is 16 lanes = two 8-lane groups.
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, false); }
// value of a compile-time-known lane, through the scalar unit (v_readlane) instead of the LDS crossbar
__device__ __forceinline__ double readlane_f64(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
// LDS hand-off between lanes of ONE wave: DS operations of a wave execute in order, so only the
// compiler has to be kept from reordering / caching across the hand-off (no s_barrier, and no
// wait for the global stores a workgroup barrier would drag in).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double dpp_xor1(double x) { return dpp_mov<0xB1>(x); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ double dpp_xor2(double x) { return dpp_mov<0x4E>(x); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ double dpp_xor4(double x) {                            // lanes 0-3 <-> 4-7 of each group
  const int lo = __double2loint(x), hi = __double2hiint(x);
  int l2 = __builtin_amdgcn_mov_dpp(lo, 0x104, 0xf, 0xf, true);  // row_shl:4 (banks 0, 2 keep it)
  l2 = dpp_i32<0x114, 0xa>(l2, lo);                               // row_shr:4 into banks 1, 3
  int h2 = __builtin_amdgcn_mov_dpp(hi, 0x104, 0xf, 0xf, true);
  h2 = dpp_i32<0x114, 0xa>(h2, hi);
  return __hiloint2double(h2, l2);
}
// Every lane of an 8-lane group holds 8 partial sums p[0..7] (one per row); afterwards lane l
// holds the group total of row l.  Butterfly with 4 + 2 + 1 exchanges instead of 8 x 3.
__device__ __forceinline__ double transpose_reduce8(const double *p, int l8) {
  const bool b4 = l8 & 4, b2 = l8 & 2, b1 = l8 & 1;
  double k4[4], k2[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) k4[k] = (b4 ? p[k + 4] : p[k]) + dpp_xor4(b4 ? p[k] : p[k + 4]);
#pragma unroll
  for (int k = 0; k < 2; ++k) k2[k] = (b2 ? k4[k + 2] : k4[k]) + dpp_xor2(b2 ? k4[k] : k4[k + 2]);
  return (b1 ? k2[1] : k2[0]) + dpp_xor1(b1 ? k2[0] : k2[1]);
}

#endif  // !AGX_HOST_BUILD

// ------------------------------------------------------------------ 3-vectors
AGX_DEV void cross3(const double *a, const double *b, double *c) {
  double c0 = a[1] * b[2] - a[2] * b[1];
  double c1 = a[2] * b[0] - a[0] * b[2];
  double c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
AGX_DEV double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
AGX_DEV double dot6(const double *a, const double *b) { return dot3(a, b) + dot3(a + 3, b + 3); }
AGX_DEV void mv3(const double *R, const double *x, double *y) {
  double y0 = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  double y1 = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  double y2 = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
AGX_DEV void mtv3(const double *R, const double *x, double *y) {
  double y0 = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  double y1 = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  double y2 = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
AGX_DEV void mm3(const double *A, const double *B, double *C) {
  double t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = t[i];
}
// C = A^T B
AGX_DEV void mtm3(const double *A, const double *B, double *C) {
  double t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = t[i];
}

// ------------------------------------------------------------ spatial algebra
// motion cross product  m1 x m2
AGX_DEV void mcross(const double *a, const double *b, double *c) {
  double t1[3], t2[3], t3[3];
  cross3(a + 3, b, t1);
  cross3(a, b + 3, t2);
  cross3(a + 3, b + 3, t3);
  c[0] = t1[0] + t2[0]; c[1] = t1[1] + t2[1]; c[2] = t1[2] + t2[2];
  c[3] = t3[0]; c[4] = t3[1]; c[5] = t3[2];
}
// force cross product  m x* f
AGX_DEV void fcross(const double *m, const double *f, double *c) {
  double t1[3], t2[3], t3[3];
  cross3(m + 3, f, t1);
  cross3(m + 3, f + 3, t2);
  cross3(m, f, t3);
  c[0] = t1[0]; c[1] = t1[1]; c[2] = t1[2];
  c[3] = t2[0] + t3[0]; c[4] = t2[1] + t3[1]; c[5] = t2[2] + t3[2];
}
// rigid inertia about the world origin: I = {m, h[3], Ixx,Ixy,Ixz,Iyy,Iyz,Izz}
AGX_DEV void iapply(const double *I, const double *mot, double *f) {
  const double m = I[0];
  const double *h = I + 1;
  double hxw[3], hxv[3];
  cross3(h, mot + 3, hxw);
  cross3(h, mot, hxv);
  const double *w = mot + 3;
  f[0] = m * mot[0] - hxw[0];
  f[1] = m * mot[1] - hxw[1];
  f[2] = m * mot[2] - hxw[2];
  f[3] = hxv[0] + I[4] * w[0] + I[5] * w[1] + I[6] * w[2];
  f[4] = hxv[1] + I[5] * w[0] + I[7] * w[1] + I[8] * w[2];
  f[5] = hxv[2] + I[6] * w[0] + I[8] * w[1] + I[9] * w[2];
}

// ------------------------------------------------------------------ kinematics
template <int NV>
struct Kin {
  double R[NV][9];  // world rotation of joint frame
  double p[NV][3];  // world position of joint origin
  double S[NV][6];  // world joint axis (p x z ; z)
};

template <int NV, bool CHAIN>
AGX_DEV int parent_of(const DevModel &m, int i) { return CHAIN ? i - 1 : m.parent[i]; }
template <int NV, bool CHAIN>
AGX_DEV bool is_anc(const DevModel &m, int i, int j) {  // j ancestor-or-self of i
  return CHAIN ? (j <= i) : ((m.anc[i] >> j) & 1u);
}

template <int NV, bool CHAIN>
AGX_DEV void kinematics(const DevModel &m, const double *q, Kin<NV> &k) {
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    const double *ax = m.axis[i];
    double s, c;
    sincos(q[i], &s, &c);
    const double omc = 1.0 - c;
    double Rq[9];
    Rq[0] = c + omc * ax[0] * ax[0];
    Rq[1] = omc * ax[0] * ax[1] - s * ax[2];
    Rq[2] = omc * ax[0] * ax[2] + s * ax[1];
    Rq[3] = omc * ax[1] * ax[0] + s * ax[2];
    Rq[4] = c + omc * ax[1] * ax[1];
    Rq[5] = omc * ax[1] * ax[2] - s * ax[0];
    Rq[6] = omc * ax[2] * ax[0] - s * ax[1];
    Rq[7] = omc * ax[2] * ax[1] + s * ax[0];
    Rq[8] = c + omc * ax[2] * ax[2];
    double Rl[9];
    mm3(m.placement[i], Rq, Rl);
    const int par = parent_of<NV, CHAIN>(m, i);
    if (par >= 0) {
      mm3(k.R[par], Rl, k.R[i]);
      double t[3];
      mv3(k.R[par], &m.placement[i][9], t);
      k.p[i][0] = k.p[par][0] + t[0]; k.p[i][1] = k.p[par][1] + t[1]; k.p[i][2] = k.p[par][2] + t[2];
    } else {
#pragma unroll
      for (int e = 0; e < 9; ++e) k.R[i][e] = Rl[e];
      k.p[i][0] = m.placement[i][9]; k.p[i][1] = m.placement[i][10]; k.p[i][2] = m.placement[i][11];
    }
    double z[3];
    mv3(k.R[i], ax, z);
    cross3(k.p[i], z, k.S[i]);
    k.S[i][3] = z[0]; k.S[i][4] = z[1]; k.S[i][5] = z[2];
  }
}

// world inertia of the body carried by joint i
template <int NV>
AGX_DEV void body_inertia(const DevModel &m, const Kin<NV> &k, int i, double *I) {
  double c[3];
  mv3(k.R[i], m.com[i], c);
  c[0] += k.p[i][0]; c[1] += k.p[i][1]; c[2] += k.p[i][2];
  const double ms = m.mass[i];
  double T[9], Iw[9];
  mm3(k.R[i], m.inertia[i], T);
  // Iw = T R^T
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) Iw[3 * a + b] = T[3 * a] * k.R[i][3 * b] + T[3 * a + 1] * k.R[i][3 * b + 1] + T[3 * a + 2] * k.R[i][3 * b + 2];
  const double cc = dot3(c, c);
  I[0] = ms;
  I[1] = ms * c[0]; I[2] = ms * c[1]; I[3] = ms * c[2];
  I[4] = Iw[0] + ms * (cc - c[0] * c[0]);
  I[5] = 0.5 * (Iw[1] + Iw[3]) - ms * c[0] * c[1];
  I[6] = 0.5 * (Iw[2] + Iw[6]) - ms * c[0] * c[2];
  I[7] = Iw[4] + ms * (cc - c[1] * c[1]);
  I[8] = 0.5 * (Iw[5] + Iw[7]) - ms * c[1] * c[2];
  I[9] = Iw[8] + ms * (cc - c[2] * c[2]);
}

// State carried between the two dynamics passes.
template <int NV>
struct Dyn {
  double v[NV][6];    // spatial velocity
  double Sd[NV][6];   // dS/dt = v x S
  double Ib[NV][10];  // body inertia (world)
  double Ic[NV][10];  // composite inertia (after crba)
  double a0[NV][6];   // bias acceleration (qdd = 0), includes gravity
};

// Pass A: bias torques nle = C v + g (RNEA with qdd = 0) and joint-space inertia
// M (CRBA), both in the world frame.  M gets the armature on its diagonal.
template <int NV, bool CHAIN>
AGX_DEV void bias_and_inertia(const DevModel &m, const Kin<NV> &k, const double *qd, Dyn<NV> &d, double *nle, double (*M)[NV]) {
  double f[NV][6];
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    const int par = parent_of<NV, CHAIN>(m, i);
#pragma unroll
    for (int e = 0; e < 6; ++e) d.v[i][e] = (par >= 0 ? d.v[par][e] : 0.0) + k.S[i][e] * qd[i];
    mcross(d.v[i], k.S[i], d.Sd[i]);
#pragma unroll
    for (int e = 0; e < 6; ++e) d.a0[i][e] = (par >= 0 ? d.a0[par][e] : (e < 3 ? -m.gravity[e] : 0.0)) + d.Sd[i][e] * qd[i];
    body_inertia<NV>(m, k, i, d.Ib[i]);
#pragma unroll
    for (int e = 0; e < 10; ++e) d.Ic[i][e] = d.Ib[i][e];
    double h[6], g[6], x[6];
    iapply(d.Ib[i], d.v[i], h);
    iapply(d.Ib[i], d.a0[i], g);
    fcross(d.v[i], h, x);
#pragma unroll
    for (int e = 0; e < 6; ++e) f[i][e] = g[e] + x[e];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; --i) {
    nle[i] = dot6(k.S[i], f[i]);
    double m6[6];
    iapply(d.Ic[i], k.S[i], m6);
AGX_UNROLL_NV
    for (int j = 0; j < NV; ++j) {
      if (j <= i || !CHAIN) {
        if (j == i) {
          M[i][i] = dot6(k.S[i], m6) + m.armature[i];
        } else if (is_anc<NV, CHAIN>(m, i, j)) {
          const double val = dot6(k.S[j], m6);
          M[i][j] = val;
          M[j][i] = val;
        } else if (!CHAIN && !is_anc<NV, CHAIN>(m, j, i)) {
          M[i][j] = 0.0;  // different branches
        }
      }
    }
    const int par = parent_of<NV, CHAIN>(m, i);
    if (par >= 0) {
#pragma unroll
      for (int e = 0; e < 6; ++e) f[par][e] += f[i][e];
#pragma unroll
      for (int e = 0; e < 10; ++e) d.Ic[par][e] += d.Ic[i][e];
    }
  }
}

// In-place lower Cholesky of a dense NV x NV SPD matrix (registers), then
// explicit inverse Minv = L^-T L^-1.
// x = A^-1 b for a symmetric positive definite A (destroyed): Cholesky + two triangular solves.
// Used by the large-model derivative pass, where forming the inverse (n^3) is not worth it.
template <int NV>
AGX_DEV void spd_solve(double (*A)[NV], double *b) {
  // right-looking (outer product) Cholesky: after column j is final, the trailing rows are updated
  // with independent FMAs (no dot-product chains through scratch memory)
  for (int j = 0; j < NV; ++j) {
    const double l = sqrt(A[j][j]), il = 1.0 / l;
    A[j][j] = l;
    for (int i = j + 1; i < NV; ++i) A[i][j] *= il;
    for (int i = j + 1; i < NV; ++i) {
      const double lij = A[i][j];
#pragma unroll 4
      for (int kk = j + 1; kk <= i; ++kk) A[i][kk] -= lij * A[kk][j];
    }
  }
  for (int i = 0; i < NV; ++i) {
    double s = b[i];
    for (int kk = 0; kk < i; ++kk) s -= A[i][kk] * b[kk];
    b[i] = s / A[i][i];
  }
  for (int i = NV - 1; i >= 0; --i) {
    double s = b[i];
    for (int kk = i + 1; kk < NV; ++kk) s -= A[kk][i] * b[kk];
    b[i] = s / A[i][i];
  }
}

template <int NV>
AGX_DEV void spd_inverse(double (*A)[NV], double (*Ainv)[NV]) {
  double Li[NV][NV];  // L^-1 (lower)
AGX_UNROLL_NV
  for (int j = 0; j < NV; ++j) {
    double dd = A[j][j];
#pragma unroll
    for (int kk = 0; kk < j; ++kk) dd -= A[j][kk] * A[j][kk];
    const double l = sqrt(dd);
    const double il = 1.0 / l;
    A[j][j] = l;
AGX_UNROLL_NV
    for (int i = j + 1; i < NV; ++i) {
      double s = A[i][j];
#pragma unroll
      for (int kk = 0; kk < j; ++kk) s -= A[i][kk] * A[j][kk];
      A[i][j] = s * il;
    }
  }
  // invert L: Li[i][j] for j <= i
AGX_UNROLL_NV
  for (int j = 0; j < NV; ++j) {
    Li[j][j] = 1.0 / A[j][j];
AGX_UNROLL_NV
    for (int i = j + 1; i < NV; ++i) {
      double s = 0.0;
#pragma unroll
      for (int kk = j; kk < i; ++kk) s -= A[i][kk] * Li[kk][j];
      Li[i][j] = s / A[i][i];
    }
  }
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int kk = i; kk < NV; ++kk) s += Li[kk][i] * Li[kk][j];
      Ainv[i][j] = s;
      Ainv[j][i] = s;
    }
}

// Pass B: partial derivatives of RNEA at (q, qd, qdd) in the world frame.
//   dq[i][j] = d tau_i / d q_j ,  dv[i][j] = d tau_i / d qd_j
// Derivation ("RNEA derivatives"): with Sd = v x S, psi = a x S + v x Sd,
// composite inertia Ic, composite force fC and the composite Coriolis-like matrix
// D = 2B whose only non-zero blocks are  D_lin,ang = -2 [f0]x  and  D_ang,ang = E:
//   j on the path root..i :  dv = 2 (Ic_i S_i).Sd_j + (D_i^T S_i).S_j
//                            dq = (D_i^T S_i).Sd_j + (Ic_i S_i).psi_j
//   i strict ancestor of j:  dv = S_i.(2 Ic_j Sd_j + D_j S_j)
//                            dq = S_i.(S_j x* fC_j + D_j Sd_j + Ic_j psi_j)
template <int NV, bool CHAIN>
AGX_DEV void rnea_derivatives(const DevModel &m, const Kin<NV> &k, const Dyn<NV> &d, const double *qd, const double *qdd,
                              double (*dq)[NV], double (*dv)[NV]) {
  double a[NV][6], psi[NV][6], fC[NV][6], f0C[NV][3], EC[NV][9];
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    const int par = parent_of<NV, CHAIN>(m, i);
#pragma unroll
    for (int e = 0; e < 6; ++e)
      a[i][e] = (par >= 0 ? a[par][e] : (e < 3 ? -m.gravity[e] : 0.0)) + k.S[i][e] * qdd[i] + d.Sd[i][e] * qd[i];
    double t1[6], t2[6];
    mcross(a[i], k.S[i], t1);
    mcross(d.v[i], d.Sd[i], t2);
#pragma unroll
    for (int e = 0; e < 6; ++e) psi[i][e] = t1[e] + t2[e];
    double h[6], g[6], x[6];
    iapply(d.Ib[i], d.v[i], h);
    iapply(d.Ib[i], a[i], g);
    fcross(d.v[i], h, x);
#pragma unroll
    for (int e = 0; e < 6; ++e) fC[i][e] = g[e] + x[e];
    f0C[i][0] = h[0]; f0C[i][1] = h[1]; f0C[i][2] = h[2];
    // E = W Io + (W Io)^T - v hh^T - hh v^T + 2 (v.hh) 1 - [n0]x
    const double *I = d.Ib[i];
    const double *vl = d.v[i], *w = d.v[i] + 3, *hh = I + 1;
    const double Io[9] = {I[4], I[5], I[6], I[5], I[7], I[8], I[6], I[8], I[9]};
    double WI[9];
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
      for (int c = 0; c < 3; ++c)
        EC[i][3 * r + c] = WI[3 * r + c] + WI[3 * c + r] - vl[r] * hh[c] - hh[r] * vl[c] + (r == c ? 2.0 * vh : 0.0);
    const double *n0 = h + 3;
    EC[i][1] += n0[2]; EC[i][2] -= n0[1];
    EC[i][3] -= n0[2]; EC[i][5] += n0[0];
    EC[i][6] += n0[1]; EC[i][7] -= n0[0];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; --i) {
    double m6[6];
    iapply(d.Ic[i], k.S[i], m6);
    // Dt_ang = 2 f0C x S_lin + E^T S_ang
    double Dt[3], t[3];
    cross3(f0C[i], k.S[i], t);
    const double *sa = k.S[i] + 3;
    Dt[0] = 2.0 * t[0] + EC[i][0] * sa[0] + EC[i][3] * sa[1] + EC[i][6] * sa[2];
    Dt[1] = 2.0 * t[1] + EC[i][1] * sa[0] + EC[i][4] * sa[1] + EC[i][7] * sa[2];
    Dt[2] = 2.0 * t[2] + EC[i][2] * sa[0] + EC[i][5] * sa[1] + EC[i][8] * sa[2];
    // column vectors of joint i
    double IcSd[6], IcPs[6], colv[6], colq[6], sxf[6], u1[3], u2[3], e1[3], e2[3];
    iapply(d.Ic[i], d.Sd[i], IcSd);
    iapply(d.Ic[i], psi[i], IcPs);
    fcross(k.S[i], fC[i], sxf);
    cross3(f0C[i], k.S[i] + 3, u1);   // f0 x S_ang
    cross3(f0C[i], d.Sd[i] + 3, u2);  // f0 x Sd_ang
    mv3(EC[i], k.S[i] + 3, e1);
    mv3(EC[i], d.Sd[i] + 3, e2);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      colv[e] = 2.0 * IcSd[e] - 2.0 * u1[e];
      colv[3 + e] = 2.0 * IcSd[3 + e] + e1[e];
      colq[e] = sxf[e] - 2.0 * u2[e] + IcPs[e];
      colq[3 + e] = sxf[3 + e] + e2[e] + IcPs[3 + e];
    }
AGX_UNROLL_NV
    for (int j = 0; j < NV; ++j) {
      if (j == i) {
        dv[i][i] = 2.0 * dot6(m6, d.Sd[i]) + dot3(Dt, k.S[i] + 3);
        dq[i][i] = dot3(Dt, d.Sd[i] + 3) + dot6(m6, psi[i]);
      } else if ((CHAIN && j < i) || (!CHAIN && is_anc<NV, CHAIN>(m, i, j))) {
        // j strict ancestor of i: row i, column j  and  row j, column i
        dv[i][j] = 2.0 * dot6(m6, d.Sd[j]) + dot3(Dt, k.S[j] + 3);
        dq[i][j] = dot3(Dt, d.Sd[j] + 3) + dot6(m6, psi[j]);
        dv[j][i] = dot6(k.S[j], colv);
        dq[j][i] = dot6(k.S[j], colq);
      } else if (!CHAIN && !is_anc<NV, CHAIN>(m, j, i)) {
        dv[i][j] = 0.0;
        dq[i][j] = 0.0;
      }
    }
    const int par = parent_of<NV, CHAIN>(m, i);
    if (par >= 0) {
#pragma unroll
      for (int e = 0; e < 6; ++e) fC[par][e] += fC[i][e];
#pragma unroll
      for (int e = 0; e < 3; ++e) f0C[par][e] += f0C[i][e];
#pragma unroll
      for (int e = 0; e < 9; ++e) EC[par][e] += EC[i][e];
    }
  }
}

// Plain RNEA (warm start / reference generators): tau = M(q) qdd + nle(q, qd), no armature.
template <int NV, bool CHAIN>
AGX_DEV void rnea(const DevModel &m, const Kin<NV> &k, const double *qd, const double *qdd, double *tau) {
  double v[NV][6], a[NV][6], f[NV][6];
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    const int par = parent_of<NV, CHAIN>(m, i);
    double Sd[6], I[10];
#pragma unroll
    for (int e = 0; e < 6; ++e) v[i][e] = (par >= 0 ? v[par][e] : 0.0) + k.S[i][e] * qd[i];
    mcross(v[i], k.S[i], Sd);
#pragma unroll
    for (int e = 0; e < 6; ++e)
      a[i][e] = (par >= 0 ? a[par][e] : (e < 3 ? -m.gravity[e] : 0.0)) + k.S[i][e] * qdd[i] + Sd[e] * qd[i];
    body_inertia<NV>(m, k, i, I);
    double h[6], g[6], x[6];
    iapply(I, v[i], h);
    iapply(I, a[i], g);
    fcross(v[i], h, x);
#pragma unroll
    for (int e = 0; e < 6; ++e) f[i][e] = g[e] + x[e];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; --i) {
    tau[i] = dot6(k.S[i], f[i]);
    const int par = parent_of<NV, CHAIN>(m, i);
    if (par >= 0) {
#pragma unroll
      for (int e = 0; e < 6; ++e) f[par][e] += f[i][e];
    }
  }
}

// ------------------------------------------------------------------ log maps
// pinocchio::log3 including its explicit branch near pi (spatial/log.hxx); the
// branch decides the sign of the axis at theta = pi, which the reference's golden
// solution depends on (Panda tool frame at q = 0 is exactly pi from identity).
AGX_DEV void log3(const double *R, double *w) {
  double tr = R[0] + R[4] + R[8];
  tr = fmin(3.0, fmax(-1.0, tr));
  const double ct = 0.5 * (tr - 1.0);
  const double theta = acos(ct);
  if (theta >= 3.14159265358979323846 - 1e-2) {
    const double cphi = -ct;
    const double beta = theta * theta / (1.0 + cphi);
    const double v0 = (R[0] + cphi) * beta, v1 = (R[4] + cphi) * beta, v2 = (R[8] + cphi) * beta;
    w[0] = (R[7] > R[5] ? 1.0 : -1.0) * (v0 > 0.0 ? sqrt(v0) : 0.0);
    w[1] = (R[2] > R[6] ? 1.0 : -1.0) * (v1 > 0.0 ? sqrt(v1) : 0.0);
    w[2] = (R[3] > R[1] ? 1.0 : -1.0) * (v2 > 0.0 ? sqrt(v2) : 0.0);
  } else {
    const double t = 0.5 * (theta > 1e-8 ? theta / sin(theta) : 1.0);
    w[0] = t * (R[7] - R[5]);
    w[1] = t * (R[2] - R[6]);
    w[2] = t * (R[3] - R[1]);
  }
}
AGX_DEV void jlog3(const double *w, double *J) {
  const double t2 = dot3(w, w), t = sqrt(t2);
  double alpha, diag;
  if (t < 1e-4) {
    alpha = 1.0 / 12.0 + t2 / 720.0;
    diag = 0.5 * (2.0 - t2 / 6.0);
  } else {
    double st, ct;
    sincos(t, &st, &ct);
    const double s1 = st / (1.0 - ct);
    alpha = 1.0 / t2 - s1 / (2.0 * t);
    diag = 0.5 * t * s1;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) J[3 * i + j] = alpha * w[i] * w[j] + (i == j ? diag : 0.0);
  J[1] -= 0.5 * w[2]; J[2] += 0.5 * w[1];
  J[3] += 0.5 * w[2]; J[5] -= 0.5 * w[0];
  J[6] -= 0.5 * w[1]; J[7] += 0.5 * w[0];
}
// r = log6(R, p) and (optionally) the blocks of Jlog6: TL (= BR) and TR; BL = 0.
template <bool JAC>
AGX_DEV void log6(const double *R, const double *p, double *r, double *TL, double *TR) {
  double w[3];
  log3(R, w);
  const double t2 = dot3(w, w);
  double alpha, beta, bdot = 1.0 / 360.0;
  const double t = sqrt(t2);
  if (t2 < 1e-12) {
    alpha = 1.0 - t2 / 12.0;
    beta = 1.0 / 12.0 + t2 / 720.0;
  } else {
    double st, ct;
    sincos(t, &st, &ct);
    const double i22 = 1.0 / (2.0 * (1.0 - ct));
    alpha = t * st * i22;
    beta = 1.0 / t2 - st / t * i22;
    if (t >= 1e-4) {
      const double tinv = 1.0 / t, t2inv = tinv * tinv;
      bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * i22;
    }
  }
  double wxp[3];
  cross3(w, p, wxp);
  const double wp = dot3(w, p);
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    r[e] = alpha * p[e] - 0.5 * wxp[e] + beta * wp * w[e];
    r[3 + e] = w[e];
  }
  if (JAC) {
    jlog3(w, TL);
    // pinocchio::Jlog6 uses the Taylor beta below 1e-4 (not 1e-6): keep its thresholds
    double betaJ = beta;
    if (t < 1e-4) betaJ = 1.0 / 12.0 + t2 / 720.0;
    double v3[3], C[9];
#pragma unroll
    for (int e = 0; e < 3; ++e) v3[e] = (bdot * wp) * w[e] - (t2 * bdot + 2.0 * betaJ) * p[e];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) C[3 * i + j] = v3[i] * w[j] + betaJ * w[i] * p[j] + (i == j ? wp * betaJ : 0.0);
    C[1] -= 0.5 * p[2]; C[2] += 0.5 * p[1];
    C[3] += 0.5 * p[2]; C[5] -= 0.5 * p[0];
    C[6] -= 0.5 * p[1]; C[7] += 0.5 * p[0];
    mm3(C, TL, TR);
  }
}

// world placement of an operational frame
template <int NV>
AGX_DEV void frame_world(const DevModel &m, const Kin<NV> &k, int frame, double *R, double *p, int *joint) {
  const int par = m.frame_parent[frame];
  *joint = par;
  if (par >= 0) {
    // dynamic joint index: select with uniform compares so that Kin stays in registers
    double Rp[9], pp[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rp[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 3; ++e) pp[e] = 0.0;
AGX_UNROLL_NV
    for (int i = 0; i < NV; ++i)
      if (i == par) {
#pragma unroll
        for (int e = 0; e < 9; ++e) Rp[e] = k.R[i][e];
#pragma unroll
        for (int e = 0; e < 3; ++e) pp[e] = k.p[i][e];
      }
    mm3(Rp, m.frame_placement[frame], R);
    double t[3];
    mv3(Rp, &m.frame_placement[frame][9], t);
    p[0] = pp[0] + t[0]; p[1] = pp[1] + t[1]; p[2] = pp[2] + t[2];
  } else {
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = m.frame_placement[frame][e];
#pragma unroll
    for (int e = 0; e < 3; ++e) p[e] = m.frame_placement[frame][9 + e];
  }
}

// ------------------------------------------------------------ shooting node
// Tile layout (AGX_TILE_DOUBLES): Fx | Fu | f | Lx | Lu | Lxx | Lxu | Luu | cost
template <int NV>
struct TileOff {
  static constexpr int NX = 2 * NV, NU = NV;
  static constexpr int Fx = 0, Fu = Fx + NX * NX, f = Fu + NX * NU, Lx = f + NX, Lu = Lx + NX, Lxx = Lu + NU,
                       Lxu = Lxx + NX * NX, Luu = Lxu + NX * NU, cost = Luu + NU * NU, SIZE = cost + 1;
};

// Cost accumulators of one node.  Every supported residual depends on q only
// through the frame placement, on v and u only diagonally, so the Hessian is
// {dense qq block, diagonal vv, diagonal uu} and Lxu = 0.
// Activation of a scalar residual: value, first and second derivative.
// WeightedQuad a = w r^2 / 2; colmpc QuadExp a = exp(-r^2 / alpha), Exp a = exp(-|r| / alpha)
// (ocp_croco_generic.py:98-143; the colmpc forms restated from recall).
AGX_DEV void activation1(int act, double alpha, double w, double r, double &a, double &ar, double &arr) {
  if (act == AGX_ACT_QUAD_EXP) {
    a = exp(-r * r / alpha);
    ar = -2.0 * r * a / alpha;
    arr = (-2.0 / alpha + 4.0 * r * r / (alpha * alpha)) * a;
  } else if (act == AGX_ACT_EXP) {
    a = exp(-fabs(r) / alpha);
    ar = (r > 0.0 ? -1.0 : (r < 0.0 ? 1.0 : 0.0)) * a / alpha;
    arr = a / (alpha * alpha);
  } else {
    a = 0.5 * w * r * r;
    ar = w * r;
    arr = w;
  }
}

// The same activations on a vector residual (ocp_croco_generic.py:118-131 builds them with residual.nr components): the
// value is a function of |r|^2, the second derivative is kept diagonal as crocoddyl's ActivationDataAbstract stores it:
//   QuadExp  a = exp(-|r|^2 / alpha),  a_r = -2 a / alpha r_j,         a_rr = (-2 / alpha + 4 r_j^2 / alpha^2) a
//   Exp      a = exp(-|r| / alpha),    a_r = -a / (alpha |r|) r_j,     a_rr = a / alpha^2
// returned as a and the coefficients of  a_r = c1 r_j,  a_rr = c2 + c3 r_j^2  (recalled forms: parity unpinned, as activation1).
struct ActVec { double a, c1, c2, c3; };
AGX_DEV ActVec activation_vec(int act, double alpha, double n2) {
  ActVec A;
  if (act == AGX_ACT_QUAD_EXP) {
    A.a = exp(-n2 / alpha);
    A.c1 = -2.0 * A.a / alpha;
    A.c2 = -2.0 * A.a / alpha;
    A.c3 = 4.0 * A.a / (alpha * alpha);
  } else {
    const double n = sqrt(n2);
    A.a = exp(-n / alpha);
    A.c1 = n > 0.0 ? -A.a / (alpha * n) : 0.0;
    A.c2 = A.a / (alpha * alpha);
    A.c3 = 0.0;
  }
  return A;
}

// Closest points of two segments (Ericson 5.1.9; capsule / capsule narrow phase behind
// colmpc.ResidualDistanceCollision): parameters s, t in [0, 1].
AGX_DEV double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
AGX_DEV void closest_seg_seg(const double *a0, const double *a1, const double *b0, const double *b1, double &s, double &t) {
  const double eps = 1e-14;
  double d1[3], d2[3], r[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { d1[k] = a1[k] - a0[k]; d2[k] = b1[k] - b0[k]; r[k] = a0[k] - b0[k]; }
  const double a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r);
  if (a <= eps && e <= eps) { s = 0.0; t = 0.0; return; }
  if (a <= eps) { s = 0.0; t = clamp01(f / e); return; }
  const double c = dot3(d1, r);
  if (e <= eps) { t = 0.0; s = clamp01(-c / a); return; }
  const double b = dot3(d1, d2);
  const double denom = a * e - b * b;
  s = (denom > eps * a * e) ? clamp01((b * f - c * e) / denom) : 0.0;
  t = (b * s + f) / e;
  if (t < 0.0) { t = 0.0; s = clamp01(-c / a); }
  else if (t > 1.0) { t = 1.0; s = clamp01((b - c) / a); }
}

// Parameter s in [-h, h] of the point of the segment c + s d closest to the box |x_i| <= b_i (all in
// the box frame).  f(s) = dist^2 is convex and piecewise quadratic, f' monotone: bisection on f'.
AGX_HD double seg_box_param(const double *c, const double *d, double h, const double *b) {
  auto fp = [&](double s) {
    double g = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double x = c[i] + s * d[i], e = fabs(x) - b[i];
      if (e > 0.0) g += (x > 0.0 ? e : -e) * d[i];
    }
    return g;
  };
  if (!(h > 0.0)) return 0.0;
  double lo = -h, hi = h;
  if (fp(lo) >= 0.0) return lo;
  if (fp(hi) <= 0.0) return hi;
  for (int it = 0; it < 60; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (fp(mid) < 0.0) lo = mid; else hi = mid;
  }
  return 0.5 * (lo + hi);
}

AGX_HD bool frame_is_box(const DevModel &m, int f) { return m.frame_box[f][0] > 0.0; }
AGX_HD bool frame_has_geometry(const DevModel &m, int f) { return m.frame_radius[f] > 0.0 || frame_is_box(m, f); }

// Signed distance of two geometry frames and the witness points (on the capsule segments / on the
// box):  d = sgn |ca - cb| - ra - rb,  n = sgn (ca - cb) / |ca - cb|;  d'(q) = n' (Ja(ca) - Jb(cb)).
// Capsule / sphere pairs: segment-segment closest points.  Box against capsule / sphere: closest
// point of the segment to the box; when that point lies inside the box (sgn = -1) the box witness is
// its projection on the nearest face (a simple penetration model, not coal's EPA depth).
// core: world placements (Ra, pa), (Rb, pb) of the two geometry frames given
AGX_DEV double collision_distance_placed(const DevModel &m, int fa, int fb, const double *Ra, const double *pa, const double *Rb,
                                         const double *pb, double *ca, double *cb, double *n) {
  const double ha = m.frame_halflen[fa], hb = m.frame_halflen[fb];
  double sgn = 1.0;
  if (frame_is_box(m, fa) || frame_is_box(m, fb)) {
    const bool bb = frame_is_box(m, fb);  // which of the two is the box
    const double *Rx = bb ? Rb : Ra, *px = bb ? pb : pa, *Rc = bb ? Ra : Rb, *pc = bb ? pa : pb;
    const double *half = m.frame_box[bb ? fb : fa];
    const double hc = bb ? ha : hb;
    double rel[3], cl[3], dl[3], zc[3] = {Rc[2], Rc[5], Rc[8]};
#pragma unroll
    for (int e = 0; e < 3; ++e) rel[e] = pc[e] - px[e];
    mtv3(Rx, rel, cl);
    mtv3(Rx, zc, dl);
    const double s = seg_box_param(cl, dl, hc, half);
    double x[3], y[3];
    bool inside = true;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      x[e] = cl[e] + s * dl[e];
      y[e] = x[e] < -half[e] ? -half[e] : (x[e] > half[e] ? half[e] : x[e]);
      inside = inside && (y[e] == x[e]);
    }
    if (inside) {
      int best = 0;
      double depth = half[0] - fabs(x[0]);
#pragma unroll
      for (int e = 1; e < 3; ++e)
        if (half[e] - fabs(x[e]) < depth) { depth = half[e] - fabs(x[e]); best = e; }
#pragma unroll
      for (int e = 0; e < 3; ++e)  // (no dynamic index: the witness arrays stay in registers)
        if (e == best) y[e] = x[e] < 0.0 ? -half[e] : half[e];
      sgn = -1.0;
    }
    double wy[3];
    mv3(Rx, y, wy);
#pragma unroll
    for (int e = 0; e < 3; ++e) {  // (selects, not pointers into the witness arrays: those would live in scratch)
      const double seg = pc[e] + s * zc[e], box = px[e] + wy[e];
      ca[e] = bb ? seg : box;
      cb[e] = bb ? box : seg;
    }
  } else {
    double a0[3], a1[3], b0[3], b1[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      a0[e] = pa[e] - ha * Ra[3 * e + 2]; a1[e] = pa[e] + ha * Ra[3 * e + 2];
      b0[e] = pb[e] - hb * Rb[3 * e + 2]; b1[e] = pb[e] + hb * Rb[3 * e + 2];
    }
    double sa, sb;
    closest_seg_seg(a0, a1, b0, b1, sa, sb);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      ca[e] = pa[e] + ((2.0 * sa - 1.0) * ha) * Ra[3 * e + 2];
      cb[e] = pb[e] + ((2.0 * sb - 1.0) * hb) * Rb[3 * e + 2];
    }
  }
  double d2 = 0.0;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    n[e] = ca[e] - cb[e];
    d2 += n[e] * n[e];
  }
  const double dn = sqrt(d2), inv = dn > 0.0 ? sgn / dn : 0.0;
#pragma unroll
  for (int e = 0; e < 3; ++e) n[e] *= inv;
  return sgn * dn - (m.frame_radius[fa] + m.frame_radius[fb]);
}
template <int NV>
AGX_DEV double collision_distance(const DevModel &m, const Kin<NV> &k, int fa, int fb, double *ca, double *cb, double *n,
                                  int *ja, int *jb) {
  double Ra[9], pa[3], Rb[9], pb[3];
  frame_world<NV>(m, k, fa, Ra, pa, ja);
  frame_world<NV>(m, k, fb, Rb, pb, jb);
  return collision_distance_placed(m, fa, fb, Ra, pa, Rb, pb, ca, cb, n);
}

template <int NV>
struct CostAcc {
  double cost;
  double Lq[NV], Lv[NV], Lu[NV];
  double Lqq[NV][NV];
  double Lvv[NV], Luu[NV];
};

// Evaluates the cost rows of one node.  DIFF = false: value only (line search).
template <int NV, bool CHAIN, bool TERM, bool DIFF>
AGX_DEV void node_costs(const DevModel &m, const DevRows &rows, const Kin<NV> &k, const double *x, const double *u,
                        const double *ref, const int *frames, CostAcc<NV> &c) {
  c.cost = 0.0;
  if (DIFF) {
AGX_UNROLL_NV
    for (int i = 0; i < NV; ++i) {
      c.Lq[i] = 0.0; c.Lv[i] = 0.0; c.Lu[i] = 0.0; c.Lvv[i] = 0.0; c.Luu[i] = 0.0;
#pragma unroll
      for (int j = 0; j < NV; ++j) c.Lqq[i][j] = 0.0;
    }
  }
  for (int r = 0; r < rows.n; ++r) {
    if (!rows.active[r]) continue;
    const double *tile = ref + rows.off[r];
    const double wi = tile[0];
    const double *rr = tile + 1;
    const double *aw = rr + rows.nref[r];
    const int kind = rows.kind[r];
    const bool quad = rows.act[r] == AGX_ACT_WEIGHTED_QUAD;
    if (kind == AGX_RES_STATE && !quad) {
      double n2 = 0.0;
AGX_UNROLL_NV
      for (int i = 0; i < NV; ++i) {
        const double rq = x[i] - rr[i], rv = x[NV + i] - rr[NV + i];
        n2 += rq * rq + rv * rv;
      }
      const ActVec A = activation_vec(rows.act[r], rows.alpha[r], n2);
      c.cost += wi * A.a;
      if (DIFF) {
AGX_UNROLL_NV
        for (int i = 0; i < NV; ++i) {
          const double rq = x[i] - rr[i], rv = x[NV + i] - rr[NV + i];
          const bool real = i < rows.nvu;
          c.Lq[i] += wi * A.c1 * rq;
          c.Lv[i] += wi * A.c1 * rv;
          c.Lqq[i][i] += real ? wi * (A.c2 + A.c3 * rq * rq) : 1.0;
          c.Lvv[i] += real ? wi * (A.c2 + A.c3 * rv * rv) : 1.0;
        }
      }
    } else if (kind == AGX_RES_STATE) {
      double a = 0.0;
AGX_UNROLL_NV
      for (int i = 0; i < NV; ++i) {
        const double rq = x[i] - rr[i], rv = x[NV + i] - rr[NV + i];
        a += 0.5 * (aw[i] * rq * rq + aw[NV + i] * rv * rv);
        if (DIFF) {
          c.Lq[i] += wi * aw[i] * rq;
          c.Lv[i] += wi * aw[NV + i] * rv;
          c.Lqq[i][i] += wi * aw[i];
          c.Lvv[i] += wi * aw[NV + i];
        }
      }
      c.cost += wi * a;
    } else if (kind == AGX_RES_CONTROL && !quad) {
      if (!TERM) {
        double n2 = 0.0;
AGX_UNROLL_NV
        for (int i = 0; i < NV; ++i) n2 += (u[i] - rr[i]) * (u[i] - rr[i]);
        const ActVec A = activation_vec(rows.act[r], rows.alpha[r], n2);
        c.cost += wi * A.a;
        if (DIFF) {
AGX_UNROLL_NV
          for (int i = 0; i < NV; ++i) {
            const double ru = u[i] - rr[i];
            c.Lu[i] += wi * A.c1 * ru;
            c.Luu[i] += (i < rows.nvu) ? wi * (A.c2 + A.c3 * ru * ru) : 1.0;
          }
        }
      }
    } else if (kind == AGX_RES_CONTROL) {
      if (!TERM) {
        double a = 0.0;
AGX_UNROLL_NV
        for (int i = 0; i < NV; ++i) {
          const double ru = u[i] - rr[i];
          a += 0.5 * aw[i] * ru * ru;
          if (DIFF) {
            c.Lu[i] += wi * aw[i] * ru;
            c.Luu[i] += wi * aw[i];
          }
        }
        c.cost += wi * a;
      }
    } else if (kind == AGX_RES_FRAME_PLACEMENT || kind == AGX_RES_FRAME_TRANSLATION || kind == AGX_RES_FRAME_ROTATION) {
      int frame = frames ? frames[r] : -1;
      if (frame < 0) frame = rows.frame[r];
      double RF[9], pF[3];
      int jf;
      frame_world<NV>(m, k, frame, RF, pF, &jf);
      double res[6], J[6][NV];  // residual and its Jacobian wrt q
      int nr;
      if (kind == AGX_RES_FRAME_PLACEMENT) {
        nr = 6;
        double Rrel[9], d[3], prel[3], TL[9], TR[9];
        mtm3(rr, RF, Rrel);
        d[0] = pF[0] - rr[9]; d[1] = pF[1] - rr[10]; d[2] = pF[2] - rr[11];
        mtv3(rr, d, prel);
        log6<DIFF>(Rrel, prel, res, TL, TR);
        if (DIFF) {
AGX_UNROLL_NV
          for (int j = 0; j < NV; ++j) {
            const bool on = (jf >= 0) && (CHAIN ? (j <= jf) : ((m.anc[jf >= 0 ? jf : 0] >> j) & 1u));
            double lin[3], ang[3], dl[3], t[3];
            dl[0] = pF[0] - k.p[j][0]; dl[1] = pF[1] - k.p[j][1]; dl[2] = pF[2] - k.p[j][2];
            cross3(k.S[j] + 3, dl, t);  // z x (pF - pj)
            mtv3(RF, t, lin);           // LOCAL frame Jacobian column
            mtv3(RF, k.S[j] + 3, ang);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const double top = TL[3 * e] * lin[0] + TL[3 * e + 1] * lin[1] + TL[3 * e + 2] * lin[2] +
                                 TR[3 * e] * ang[0] + TR[3 * e + 1] * ang[1] + TR[3 * e + 2] * ang[2];
              const double bot = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
              J[e][j] = on ? top : 0.0;
              J[3 + e][j] = on ? bot : 0.0;
            }
          }
        }
      } else if (kind == AGX_RES_FRAME_TRANSLATION) {
        nr = 3;
        res[0] = pF[0] - rr[0]; res[1] = pF[1] - rr[1]; res[2] = pF[2] - rr[2];
        res[3] = res[4] = res[5] = 0.0;
        if (DIFF) {
AGX_UNROLL_NV
          for (int j = 0; j < NV; ++j) {
            const bool on = (jf >= 0) && (CHAIN ? (j <= jf) : ((m.anc[jf >= 0 ? jf : 0] >> j) & 1u));
            double dl[3], t[3];
            dl[0] = pF[0] - k.p[j][0]; dl[1] = pF[1] - k.p[j][1]; dl[2] = pF[2] - k.p[j][2];
            cross3(k.S[j] + 3, dl, t);
#pragma unroll
            for (int e = 0; e < 3; ++e) { J[e][j] = on ? t[e] : 0.0; J[3 + e][j] = 0.0; }
          }
        }
      } else {
        nr = 3;
        double Rrel[9], TL[9];
        mtm3(rr, RF, Rrel);
        log3(Rrel, res);
        res[3] = res[4] = res[5] = 0.0;
        if (DIFF) {
          jlog3(res, TL);
AGX_UNROLL_NV
          for (int j = 0; j < NV; ++j) {
            const bool on = (jf >= 0) && (CHAIN ? (j <= jf) : ((m.anc[jf >= 0 ? jf : 0] >> j) & 1u));
            double ang[3];
            mtv3(RF, k.S[j] + 3, ang);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const double bot = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
              J[e][j] = on ? bot : 0.0;
              J[3 + e][j] = 0.0;
            }
          }
        }
      }
      // weights of the gradient (J' ge) and of the Gauss-Newton Hessian (J' diag(we) J) of the row's activation
      double a = 0.0, we[6], ge[6];
      if (quad) {
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          if (e < nr) a += 0.5 * aw[e] * res[e] * res[e];
          we[e] = (e < nr) ? wi * aw[e] : 0.0;
          ge[e] = we[e] * res[e];
        }
      } else {
        double n2 = 0.0;
#pragma unroll
        for (int e = 0; e < 6; ++e)
          if (e < nr) n2 += res[e] * res[e];
        const ActVec A = activation_vec(rows.act[r], rows.alpha[r], n2);
        a = A.a;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          we[e] = (e < nr) ? wi * (A.c2 + A.c3 * res[e] * res[e]) : 0.0;
          ge[e] = (e < nr) ? wi * A.c1 * res[e] : 0.0;
        }
      }
      c.cost += wi * a;
      if (DIFF) {
        // J' ge and J' W J with the six residual rows innermost: one update per Hessian element
        // (large models keep Lqq in scratch: a read-modify-write per row and element serialises)
AGX_UNROLL_NV
        for (int i = 0; i < NV; ++i) {
          double wj[6], gi = 0.0;
#pragma unroll
          for (int e = 0; e < 6; ++e) { wj[e] = we[e] * J[e][i]; gi += ge[e] * J[e][i]; }
          c.Lq[i] += gi;
AGX_UNROLL_NV
          for (int j = 0; j <= i; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < 6; ++e) acc += wj[e] * J[e][j];
            c.Lqq[i][j] += acc;
          }
        }
      }
    } else if (kind == AGX_RES_COLLISION) {
      // colmpc.ResidualDistanceCollision (ocp_croco_generic.py:524-533) with a scalar activation
      double ca[3], cb[3], n[3];
      int ja, jb;
      const double d = collision_distance<NV>(m, k, rows.frame[r], rows.frame_b[r], ca, cb, n, &ja, &jb);
      double a, ar, arr;
      activation1(rows.act[r], rows.alpha[r], aw[0], d, a, ar, arr);
      c.cost += wi * a;
      if (DIFF) {
        double g[NV];
AGX_UNROLL_NV
        for (int j = 0; j < NV; ++j) {
          const bool ona = (ja >= 0) && (CHAIN ? (j <= ja) : ((m.anc[ja >= 0 ? ja : 0] >> j) & 1u));
          const bool onb = (jb >= 0) && (CHAIN ? (j <= jb) : ((m.anc[jb >= 0 ? jb : 0] >> j) & 1u));
          double da[3], db[3], ta[3], tb[3];
#pragma unroll
          for (int e = 0; e < 3; ++e) { da[e] = ca[e] - k.p[j][e]; db[e] = cb[e] - k.p[j][e]; }
          cross3(k.S[j] + 3, da, ta);
          cross3(k.S[j] + 3, db, tb);
          g[j] = (ona ? dot3(n, ta) : 0.0) - (onb ? dot3(n, tb) : 0.0);
        }
AGX_UNROLL_NV
        for (int i = 0; i < NV; ++i) {
          c.Lq[i] += wi * ar * g[i];
#pragma unroll
          for (int j = 0; j <= i; ++j) c.Lqq[i][j] += wi * arr * g[i] * g[j];
        }
      }
    }
  }
  if (DIFF) {
AGX_UNROLL_NV
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = i + 1; j < NV; ++j) c.Lqq[i][j] = c.Lqq[j][i];
  }
}

}  // namespace agx
#include "agx_general.hpp"
namespace agx {

// Constraints of one node (ConstraintModelManager of the node's differential model): g stacked over
// the rows; JAC also returns the Jacobian rows [d/dq (8) | d/dv (8) | d/du (8)] of every component of
// the rows with dense Jacobians (slot coll_slot[r] + e); State / Control rows have identity Jacobians.
template <int NV, bool CHAIN, bool JAC>
AGX_DEV void constraints_eval(const DevModel &m, const DevCons &c, const double *x, const double *u, double *g,
                              double (*cj)[24]) {
  Kin<NV> k;
  if (c.ncoll > 0) kinematics<NV, CHAIN>(m, x, k);
  if (JAC)
    for (int s = 0; s < c.ncoll; ++s)
      for (int j = 0; j < 24; ++j) cj[s][j] = 0.0;
  for (int r = 0; r < c.n; ++r) {
    const int kind = c.kind[r], off = c.off[r];
    if (kind == AGX_RES_STATE) {
#pragma unroll
      for (int i = 0; i < 2 * NV; ++i) g[off + i] = x[i] - c.ref[r][i];
    } else if (kind == AGX_RES_CONTROL) {
#pragma unroll
      for (int i = 0; i < NV; ++i) g[off + i] = u[i] - c.ref[r][i];
    } else if (kind == AGX_RES_FRAME_VELOCITY) {
      // v_frame(q, qd) - vref in the row's reference frame (frame_b: WORLD / LOCAL / LOCAL_WORLD_ALIGNED)
      double vel[6], Rq[6][NV], Rv[6][NV];
      frame_velocity<NV, CHAIN, JAC>(m, k, c.frame[r], c.frame_b[r], x + NV, vel, Rq, Rv);
#pragma unroll
      for (int e = 0; e < 6; ++e) g[off + e] = vel[e] - c.ref[r][e];
      if (JAC) {
        for (int e = 0; e < 6; ++e)
AGX_UNROLL_NV
          for (int j = 0; j < NV; ++j) { cj[c.coll_slot[r] + e][j] = Rq[e][j]; cj[c.coll_slot[r] + e][8 + j] = Rv[e][j]; }
      }
    } else if (kind == AGX_RES_CONTROL_GRAV) {
      // u - g(q): d/du = I, d/dq = -dg/dq (RNEA derivative at zero velocity and acceleration)
      Dyn<NV> d0;
      double g0[NV], M0[NV][NV], zero[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) zero[i] = 0.0;
      bias_and_inertia<NV, CHAIN>(m, k, zero, d0, g0, M0);
#pragma unroll
      for (int i = 0; i < NV; ++i) g[off + i] = u[i] - g0[i];
      if (JAC) {
        double Gq[NV][NV], Gv[NV][NV];
        rnea_derivatives<NV, CHAIN>(m, k, d0, zero, zero, Gq, Gv);
        for (int i = 0; i < NV; ++i) {
AGX_UNROLL_NV
          for (int j = 0; j < NV; ++j) cj[c.coll_slot[r] + i][j] = -Gq[i][j];
          cj[c.coll_slot[r] + i][16 + i] = 1.0;
        }
      }
    } else if (kind == AGX_RES_FRAME_TRANSLATION || kind == AGX_RES_FRAME_ROTATION || kind == AGX_RES_FRAME_PLACEMENT) {
      // the residuals of the cost rows (node_costs) as constraints: translation p(q) - pref with the
      // LOCAL_WORLD_ALIGNED linear frame Jacobian, rotation log3(Rref' R) with Jlog3 * LOCAL angular
      // Jacobian, placement log6(Mref^-1 M) with Jlog6 * LOCAL Jacobian
      double RF[9], pF[3];
      int jf;
      frame_world<NV>(m, k, c.frame[r], RF, pF, &jf);
      const double *rr = c.ref[r];
      double res[6], TL[9], TR[9];
      if (kind == AGX_RES_FRAME_TRANSLATION) {
#pragma unroll
        for (int e = 0; e < 3; ++e) res[e] = pF[e] - rr[e];
      } else if (kind == AGX_RES_FRAME_ROTATION) {
        double Rrel[9];
        mtm3(rr, RF, Rrel);
        log3(Rrel, res);
        if (JAC) jlog3(res, TL);
      } else {
        double Rrel[9], d[3], prel[3];
        mtm3(rr, RF, Rrel);
        d[0] = pF[0] - rr[9]; d[1] = pF[1] - rr[10]; d[2] = pF[2] - rr[11];
        mtv3(rr, d, prel);
        log6<JAC>(Rrel, prel, res, TL, TR);
      }
      for (int e = 0; e < c.nr[r]; ++e) g[off + e] = res[e];
      if (JAC) {
AGX_UNROLL_NV
        for (int j = 0; j < NV; ++j) {
          const bool on = (jf >= 0) && (CHAIN ? (j <= jf) : ((m.anc[jf >= 0 ? jf : 0] >> j) & 1u));
          double d[3], t[3], lin[3], ang[3], out[6];
#pragma unroll
          for (int e = 0; e < 3; ++e) d[e] = pF[e] - k.p[j][e];
          cross3(k.S[j] + 3, d, t);  // z x (pF - pj)
          if (kind == AGX_RES_FRAME_TRANSLATION) {
#pragma unroll
            for (int e = 0; e < 3; ++e) out[e] = t[e];
          } else {
            mtv3(RF, t, lin);
            mtv3(RF, k.S[j] + 3, ang);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const double bot = TL[3 * e] * ang[0] + TL[3 * e + 1] * ang[1] + TL[3 * e + 2] * ang[2];
              if (kind == AGX_RES_FRAME_ROTATION) out[e] = bot;
              else {
                out[e] = TL[3 * e] * lin[0] + TL[3 * e + 1] * lin[1] + TL[3 * e + 2] * lin[2] +
                         TR[3 * e] * ang[0] + TR[3 * e + 1] * ang[1] + TR[3 * e + 2] * ang[2];
                out[3 + e] = bot;
              }
            }
          }
          for (int e = 0; e < c.nr[r]; ++e) cj[c.coll_slot[r] + e][j] = on ? out[e] : 0.0;
        }
      }
    } else if (kind == AGX_RES_COLLISION) {
      double ca[3], cb[3], n[3];
      int ja, jb;
      g[off] = collision_distance<NV>(m, k, c.frame[r], c.frame_b[r], ca, cb, n, &ja, &jb);
      if (JAC) {
        double *gj = cj[c.coll_slot[r]];
AGX_UNROLL_NV
        for (int j = 0; j < NV; ++j) {
          const bool ona = (ja >= 0) && (CHAIN ? (j <= ja) : ((m.anc[ja >= 0 ? ja : 0] >> j) & 1u));
          const bool onb = (jb >= 0) && (CHAIN ? (j <= jb) : ((m.anc[jb >= 0 ? jb : 0] >> j) & 1u));
          double da[3], db[3], ta[3], tb[3];
#pragma unroll
          for (int e = 0; e < 3; ++e) { da[e] = ca[e] - k.p[j][e]; db[e] = cb[e] - k.p[j][e]; }
          cross3(k.S[j] + 3, da, ta);
          cross3(k.S[j] + 3, db, tb);
          gj[j] = (ona ? dot3(n, ta) : 0.0) - (onb ? dot3(n, tb) : 0.0);
        }
      }
    }
  }
}
// constraint kinds whose components are scalar rows with dense Jacobians [Gq | Gv | Gu] (Jacobian slots)
AGX_HD bool cons_has_dense_rows(int kind) {
  return kind == AGX_RES_COLLISION || kind == AGX_RES_FRAME_TRANSLATION || kind == AGX_RES_FRAME_ROTATION || kind == AGX_RES_FRAME_PLACEMENT ||
         kind == AGX_RES_FRAME_VELOCITY || kind == AGX_RES_CONTROL_GRAV;
}
// l1 norm of the violation of lb <= g <= ub (SolverCSQP::calc / tryStep constraint_norm)
AGX_DEV double violation_l1(const DevCons &c, const double *g) {
  double v = 0.0;
  for (int k = 0; k < c.nc; ++k) v += fmax(c.lb[k] - g[k], 0.0) + fmax(g[k] - c.ub[k], 0.0);
  return v;
}
// (Until round 2 a `constraint_violation(m, c, x, u)` wrapper lived here, `noinline` because the constrained step kernel
// evaluated trial merits wrongly with it inlined -- DESIGN.md section 8.  The line search now takes the violation of a trial
// point from k_con_eval run at that point; nothing evaluates constraints lane by lane inside a step kernel any more.)
// ADMM penalty of one constraint component (SolverCSQP::apply_rho_update)
AGX_DEV double admm_rho(double lb, double ub, double rho_sparse) {
  if (lb == -INFINITY && ub == INFINITY) return 1e-6;
  if (fabs(lb - ub) <= 1e-6) return 1e3 * rho_sparse;
  return rho_sparse;
}

// calc of a running node: forward dynamics + semi-implicit Euler + cost
// (crocoddyl IntegratedActionModelEuler::calc).
template <int NV, bool CHAIN, bool GEN = false>
AGX_DEV void node_calc_running(const DevModel &m, const DevRows &rows, double dt, const double *x, const double *u,
                               const double *ref, const int *frames, double *xnext, double *cost) {
  Kin<NV> k;
  kinematics<NV, CHAIN>(m, x, k);
  Dyn<NV> d;
  double nle[NV], M[NV][NV], Minv[NV][NV];
  bias_and_inertia<NV, CHAIN>(m, k, x + NV, d, nle, M);
  spd_inverse<NV>(M, Minv);
AGX_UNROLL_NV
  for (int i = 0; i < NV; ++i) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) a += Minv[i][j] * (u[j] - nle[j]);
    xnext[NV + i] = x[NV + i] + dt * a;
    xnext[i] = x[i] + dt * x[NV + i] + dt * dt * a;
  }
  CostAcc<NV> c;
  node_costs<NV, CHAIN, false, false>(m, rows, k, x, u, ref, frames, c);
  if constexpr (GEN) {
    CostGen<NV> g;
    node_costs_general<NV, CHAIN, false, false>(m, rows, k, x, u, ref, frames, c, g);
  }
  *cost = dt * c.cost;
}

template <int NV, bool CHAIN, bool GEN = false>
AGX_DEV void node_calc_terminal(const DevModel &m, const DevRows &rows, const double *x, const double *ref,
                                const int *frames, double *cost) {
  Kin<NV> k;
  kinematics<NV, CHAIN>(m, x, k);
  CostAcc<NV> c;
  node_costs<NV, CHAIN, true, false>(m, rows, k, x, nullptr, ref, frames, c);
  if constexpr (GEN) {
    CostGen<NV> g;
    node_costs_general<NV, CHAIN, true, false>(m, rows, k, x, nullptr, ref, frames, c, g);
  }
  *cost = c.cost;
}

}  // namespace agx
