// agimus_controller_amd -- constrained QP direction: the ADMM loop of mim_solvers::SolverCSQP
// (computeDirection / backwardPass / forwardPass / update_lagrangian_parameters / update_rho_vec),
// restated from recall (parity with the reference binaries unpinned, checked against
// the CPU restatement under oracle/).  Correctness-first layout: every ADMM iteration is
//
//   k_admm_tile    node parallel   QP tile + sigma / rho / y / z / prox-centre terms -> augmented tile
//   k_riccati      one wave / instance (the production kernel, unchanged) on the augmented tile
//   k_admm_update  node parallel   du, C d, z / y update, residual norms, the node's KKT share
//   k_admm_reduce  one block / instance: norms, rho adaptation, convergence
//
// Supported constraint kinds: Control (ConstraintModelControlLimit) and State (identity Jacobians, lane
// local), and every other residual as scalar rows with dense Jacobians [Gq | Gv | Gu]: collision
// distance, FrameTranslation / FrameRotation / FramePlacement, FrameVelocity, ControlGrav.
// In the acceleration-input coordinates of the QP tiles (du = M w + taux dx) a constraint row with
// Jacobians (Gx, Gu) has the row  c = [Gx + Gu taux | Gu M]  on (dx, w).
//
// (included at the end of agx_kernels.hpp)
#pragma once

namespace agx {

constexpr double kAlphaRelax = 1.6;  // SolverCSQP alpha
constexpr double kRhoMin = 1e-6, kRhoMax = 1e3, kAdaptiveRhoTol = 5.0;
constexpr int kRhoInterval = 25;

// g, collision Jacobians and the l1 violation of every node at the current (xs, us).
// One lane per node.  cg [B][T+1][AGX_MAX_NC], cjac [B][T+1][AGX_MAX_DENSE][24] (d/dq | d/dv | d/du, 8 each).
template <int NV, bool CHAIN>
__global__ void __launch_bounds__(64) k_con_eval(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                 const double *__restrict__ xs, const double *__restrict__ us,
                                                 double *__restrict__ cg, double *__restrict__ cjac,
                                                 double *__restrict__ nodestat, const DevState *__restrict__ st, int phase) {
  constexpr int NX = 2 * NV, NU = NV;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long node = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= (long long)o.B * (T + 1)) return;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  if (!k1_active(st[b], phase)) return;  // phase 1: at the trial point of a searching instance, results in place
  const DevCons &c = o.cons[t == T ? 1 : 0];
  double x[NX], u[NU], g[AGX_MAX_NC], cj[AGX_MAX_DENSE][24];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = xs[node * NX + i];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = (t < T) ? us[((long long)b * T + t) * NU + i] : 0.0;
  for (int k = 0; k < AGX_MAX_NC; ++k) g[k] = 0.0;
  constraints_eval<NV, CHAIN, true>(m, c, x, u, g, cj);
  for (int k = 0; k < c.nc; ++k) cg[node * AGX_MAX_NC + k] = g[k];
  for (int r = 0; r < c.ncoll; ++r)
    for (int j = 0; j < 24; ++j) cjac[(node * AGX_MAX_DENSE + r) * 24 + j] = ((j & 7) < NV) ? cj[r][j] : 0.0;
  nodestat[node * 4 + 3] = violation_l1(c, g);
}

// Start of the ADMM loop of one SQP iteration (reset_params + equality-QP initial guess): prox centre
// cx <- dx of the plain LQR pass (du was written by k_node_kkt), z <- 0, y kept.
template <int NV>
__global__ void __launch_bounds__(256) k_admm_init(const DevOcp *__restrict__ op, const double *__restrict__ dxs,
                                                   double *__restrict__ cxs, double *__restrict__ zs, DevState *__restrict__ st,
                                                   int *__restrict__ n_conv) {
  constexpr int NX = 2 * NV;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long node = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= (long long)o.B * (T + 1)) return;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  DevState &S = st[b];
  if (S.done) {
    if (t == 0) atomicAdd(n_conv, 1);  // finished instances count as converged: the host waits for B
    return;
  }
  for (int i = 0; i < NX; ++i) cxs[node * NX + i] = dxs[node * NX + i];
  for (int k = 0; k < AGX_MAX_NC; ++k) zs[node * AGX_MAX_NC + k] = 0.0;
  if (t == 0) {
    S.admm_conv = 0;
    S.admm_refactor = 1;
    S.admm_iter = o.max_qp;
    if (!(S.rho_sparse > 0.0)) S.rho_sparse = 1e-1;  // rho_sparse_base of a fresh solver
  }
}

// Pre-factorised flow: the per-instance part of k_admm_init, ahead of the plain LQR pass (the augmented Hessians
// and their factorisation do not depend on it)
__global__ void k_admm_pre(const DevOcp *__restrict__ op, DevState *__restrict__ st) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= op->B) return;
  DevState &S = st[b];
  if (S.done) return;
  S.admm_conv = 0;
  S.admm_refactor = 1;
  S.admm_iter = op->max_qp;
  if (!(S.rho_sparse > 0.0)) S.rho_sparse = 1e-1;  // rho_sparse_base of a fresh solver
}

// Augmented QP tile of one node: 8 lanes per node, lane j owns column j of every block.
//   H  += [taux M]' diag(sigma + rho_u) [taux M] + sigma I_x + rho_x (state rows) + rho g g' (collision)
//   g  += [taux M]' (h_u - sigma du_c) - sigma dx_c + h_x + h g,      h = y - rho z
template <int NV>
__global__ void __launch_bounds__(128) k_admm_tile(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                   double *__restrict__ qt2s, const double *__restrict__ auxs,
                                                   const double *__restrict__ cxs, const double *__restrict__ dus,
                                                   const double *__restrict__ cjac, const double *__restrict__ ys,
                                                   const double *__restrict__ zs, const DevState *__restrict__ st, int grad_only) {
  constexpr int NX = 2 * NV, LD = 8, B2 = NV * LD;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  __shared__ double lds[16][3 * B2 + 2];
  const DevOcp &o = *op;
  const int T = o.T;
  const int l8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const long long n_nodes = (long long)o.B * (T + 1);
  const long long node = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const bool ok = node < n_nodes;
  const long long unit = ok ? node : 0;
  const int b = (int)(unit / (T + 1)), t = (int)(unit % (T + 1));
  const DevState &S = st[b];
  const bool live = ok && !S.done && !S.admm_conv;
  const double *qt = qts + unit * Q::SIZE;
  double *q2 = qt2s + unit * Q::SIZE;
  const double *ax = auxs + unit * A::SIZE;
  const double *cx = cxs + unit * NX;
  const double *y = ys + unit * AGX_MAX_NC, *z = zs + unit * AGX_MAX_NC;
  const DevCons &c = o.cons[t == T ? 1 : 0];
  const double sig = kSigma, rs = S.rho_sparse;
  const bool jl = l8 < NV, wr = live && jl;
  const int j = jl ? l8 : 0;
  // runs only where the Hessian part changed (first ADMM iteration of the SQP iteration, or new rho);
  // otherwise k_admm_update has already written the gradient
  // grad_only: the Hessian part of these tiles is already there (and factorised: admm_direction, pre-factorised flow);
  // only the gradient, which needs the plain LQR pass, is (re)written
  const bool full = S.admm_refactor != 0 && !grad_only, grad = S.admm_refactor != 0;
  if (!__syncthreads_or(live && grad)) return;
  // stage M | tq | tv (contiguous in the aux tile)
  double *sh = lds[grp];
  for (int e = l8; e < 3 * B2; e += 8) sh[e] = ax[A::M + e];
  __syncthreads();
  // The six Hessian blocks of the augmented tile are formed in registers (lane j: column j of every block, entry [i][j] of a
  // row i is the group's 64-byte line) and stored ONCE: base tile + diagonal terms + rank-one terms of the dense rows +
  // [taux M]' diag(sigma + rho_u) [taux M].  (Until round 3 the base tile was copied first and every term was a
  // read-modify-write of global memory: 0.31 ms per launch at B = 256, T = 200.)  The rest of the tile is copied.
  double a_qq[NV], a_qv[NV], a_vv[NV], a_qw[NV], a_vw[NV], a_ww[NV];
  if (full) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      a_qq[i] = qt[Q::Hqq + i * Q::LD + j]; a_qv[i] = qt[Q::Hqv + i * Q::LD + j]; a_vv[i] = qt[Q::Hvv + i * Q::LD + j];
      a_qw[i] = qt[Q::Hqw + i * Q::LD + j]; a_vw[i] = qt[Q::Hvw + i * Q::LD + j]; a_ww[i] = qt[Q::Hww + i * Q::LD + j];
    }
    if (live)
      for (int e = Q::gx + l8; e < Q::SIZE; e += 8) q2[e] = qt[e];
  }
  // per-row weights on u (control rows) and the state / collision terms
  double wu[NV], hu[NV];
#pragma unroll
  for (int l = 0; l < NV; ++l) { wu[l] = sig; hu[l] = 0.0; }
  double add_qq_diag = sig, add_vv_diag = sig, gq = -sig * cx[j], gv = -sig * cx[NV + j];
  for (int r = 0; r < c.n; ++r) {
    const int off = c.off[r];
    if (c.kind[r] == AGX_RES_CONTROL) {
#pragma unroll
      for (int l = 0; l < NV; ++l) {
        const double rho = admm_rho(c.lb[off + l], c.ub[off + l], rs);
        wu[l] += rho;
        hu[l] += y[off + l] - rho * z[off + l];
      }
    } else if (c.kind[r] == AGX_RES_STATE) {
      const double rq = admm_rho(c.lb[off + j], c.ub[off + j], rs), rv = admm_rho(c.lb[off + NV + j], c.ub[off + NV + j], rs);
      add_qq_diag += rq; add_vv_diag += rv;
      gq += y[off + j] - rq * z[off + j];
      gv += y[off + NV + j] - rv * z[off + NV + j];
    }
  }
  if (full) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (i == j) { a_qq[i] += add_qq_diag; a_vv[i] += add_vv_diag; }
  }
  // rows with dense Jacobians (collision distance, frame residuals, ControlGrav): every component is a
  // scalar row G = [Gq | Gv | Gu]; in tile coordinates  c = [Gq + Gu taux | Gu M]  on (dx, w): rank one
  // on every block; the gradient terms of the u part ride in hu[] with those of the control rows
  const double *Mm = sh, *tq = sh + B2, *tv = sh + 2 * B2;
  for (int r = 0; r < c.n; ++r) {
    if (!cons_has_dense_rows(c.kind[r])) continue;
    for (int e = 0; e < c.nr[r]; ++e) {
      const int off = c.off[r] + e;
      const double *gj = cjac + (unit * AGX_MAX_DENSE + c.coll_slot[r] + e) * 24;
      const double rho = admm_rho(c.lb[off], c.ub[off], rs);
      const double h = y[off] - rho * z[off];
      gq += h * gj[j];
      gv += h * gj[8 + j];
      double cq = gj[j], cv = gj[8 + j], cw = 0.0;
      if (t < T) {
#pragma unroll
        for (int l = 0; l < NV; ++l) {
          const double gu = gj[16 + l];
          hu[l] += h * gu;
          cw += Mm[l * LD + j] * gu; cq += tq[l * LD + j] * gu; cv += tv[l * LD + j] * gu;
        }
      }
      if (full) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const double cqi = __shfl(cq, i, 8), cvi = __shfl(cv, i, 8), cwi = __shfl(cw, i, 8);
          a_qq[i] += rho * cqi * cq;
          a_qv[i] += rho * cqi * cv;
          a_vv[i] += rho * cvi * cv;
          if (t < T) {
            a_ww[i] += rho * cwi * cw;
            a_qw[i] += rho * cqi * cw;
            a_vw[i] += rho * cvi * cw;
          }
        }
      }
    }
  }
  double gwv = 0.0;
  if (t < T) {
    const double *du = dus + ((long long)b * T + t) * NV;
    double Mc[NV], tqc[NV], tvc[NV];
#pragma unroll
    for (int l = 0; l < NV; ++l) {
      Mc[l] = Mm[l * LD + j]; tqc[l] = tq[l * LD + j]; tvc[l] = tv[l * LD + j];
      const double e = hu[l] - sig * du[l];
      gwv += Mc[l] * e; gq += tqc[l] * e; gv += tvc[l] * e;
    }
    if (full)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double hww = 0.0, hqw = 0.0, hvw = 0.0, hqq = 0.0, hqv = 0.0, hvv = 0.0;
#pragma unroll
      for (int l = 0; l < NV; ++l) {
        const double Mli = Mm[l * LD + i] * wu[l], tqli = tq[l * LD + i] * wu[l], tvli = tv[l * LD + i] * wu[l];
        hww += Mli * Mc[l]; hqw += tqli * Mc[l]; hvw += tvli * Mc[l];
        hqq += tqli * tqc[l]; hqv += tqli * tvc[l]; hvv += tvli * tvc[l];
      }
      a_ww[i] += hww; a_qw[i] += hqw; a_vw[i] += hvw;
      a_qq[i] += hqq; a_qv[i] += hqv; a_vv[i] += hvv;
    }
  }
  if (wr && full) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      q2[Q::Hqq + i * Q::LD + j] = a_qq[i]; q2[Q::Hqv + i * Q::LD + j] = a_qv[i]; q2[Q::Hvv + i * Q::LD + j] = a_vv[i];
      q2[Q::Hqw + i * Q::LD + j] = a_qw[i]; q2[Q::Hvw + i * Q::LD + j] = a_vw[i]; q2[Q::Hww + i * Q::LD + j] = a_ww[i];
    }
  }
  if (wr && grad) {
    if (t < T) q2[Q::gw + j] = qt[Q::gw + j] + gwv;
    q2[Q::gx + j] = qt[Q::gx + j] + gq;
    q2[Q::gx + NV + j] = qt[Q::gx + NV + j] + gv;
  }
}

// ---------------------------------------------------------------------------
// Riccati sweep of one ADMM iteration.  When the Hessian part of the instance's augmented tiles
// changed (S.admm_refactor) the full Gauss-Jordan sweep runs and leaves its factors behind
// (FT: multipliers of the 7 pivots, pivot reciprocals, V' f of every node); otherwise only the
// gradient recursion is redone with those factors -- the reference makes the same distinction
// (backwardPass / backwardPass_without_rho_update).
// ---------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void riccati_vec_body(const int b, const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                 const double *__restrict__ qts, const double *__restrict__ Kws,
                                                 double *__restrict__ kws, double *__restrict__ dxs, double *__restrict__ wss,
                                                 const double *__restrict__ facs) {
  constexpr int NX = 2 * NV, TS = QT<NV>::SIZE;
  typedef QT<NV> Q;
  typedef FT<NV> F;
  const DevOcp &o = *op;
  const int T = o.T, lane = threadIdx.x;
  const int r = lane >> 3, c = lane & 7;
  const int rr = r < NV ? r : 0;
  const double *qb = qts + (long long)b * (T + 1) * TS;
  const double *Kw = Kws + (long long)b * T * NV * NX;
  double *kw = kws + (long long)b * T * NV;
  const double *fb = facs + (long long)b * T * F::SIZE;
  // Lanes c = 0, 1, 2 of grid row r carry entry r of the three gradient blocks (w | q | v); each has
  // its own factor column, so a pivot is one v_readlane pair and ONE FMA per lane.
  const int sel = c < 3 ? c : 0;
  const int goff = sel == 0 ? Q::gw : (sel == 1 ? Q::gx : Q::gx + NV);
  const int foff = sel == 0 ? F::FW : (sel == 1 ? F::FQ : F::FV);
  const int poff = sel == 2 ? F::PV : F::PQ;
  double v = qb[(long long)T * TS + (sel == 2 ? Q::gx + NV : Q::gx) + rr];  // value-function gradient (c = 1: q, c = 2: v)
  struct Node { double g, rp, p, f[NV]; };
  __shared__ double s_dt[kMaxHorizon];  // step lengths (see stage_dts)
  stage_dts(s_dt, dts, T);
  auto load_node = [&](Node &z, int t) {
    const double *tl = qb + (long long)t * TS;
    const double *ft = fb + (long long)t * F::SIZE;
    z.g = tl[goff + rr];
    z.rp = ft[F::RP + rr]; z.p = ft[poff + rr];
#pragma unroll
    for (int k = 0; k < NV; ++k) z.f[k] = ft[foff + k * 8 + rr];
  };
  auto step = [&](Node &z, int t) {
    const double h = s_dt[t], h2 = h * h;
    const double vp = v + z.p;
    const double vpq = dpp_mov<0x55>(vp), vpv = dpp_mov<0xAA>(vp);  // quad broadcast of lanes c = 1 / c = 2
    const double ca = sel == 0 ? h2 : (sel == 1 ? 1.0 : h), cb = sel == 0 ? h : (sel == 1 ? 0.0 : 1.0);
    double g = z.g + ca * vpq + cb * vpv;  // gw + h2 vpq + h vpv | gq + vpq | gv + h vpq + vpv
    const double rp = z.rp;
    double f[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) f[k] = z.f[k];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double gk = readlane_f64(g, 8 * k);  // gW of row k (its own factor fw[k] is 0 there)
      g -= f[k] * gk;
    }
    if (c == 0 && r < NV) kw[(long long)t * NV + r] = g * rp;
    v = g;
    prefetch_group_begin();
    load_node(z, t >= 4 ? t - 4 : 0);  // unconditional, after the last use of the old contents (see agx_riccati_mx.hpp)
    prefetch_group_end();
  };
  int t = T - 1;
  for (int rem = T % 4; rem > 0; --rem, --t) {
    Node z;
    load_node(z, t);
    step(z, t);
  }
  if (t >= 0) {
    Node n[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) load_node(n[i], t - i);
    prefetch_queue_settle(n);
    for (; t >= 0; t -= 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) step(n[i], t - i);
    }
  }
  riccati_forward<NV>(b, T, dts, qb, Kw, kw, dxs, wss, s_dt);
}

// ---------------------------------------------------------------------------
// Gradient-only sweeps, parallel in time.  Between two factorisations the backward gradient recursion and the forward
// pass are AFFINE recursions whose linear parts are fixed: with the closed-loop transition  Abar_t = Phi - G Kw_t
//     forward    dx_{t+1} = Abar_t dx_t + (f_t - G kw_t)
//     backward   v_t      = Abar_t' v_{t+1} + c_t                    (c_t: what the node's gradient contributes)
// so the horizon is cut into kSeg segments, one wave each:
//   1. every wave sweeps its segment from a zero boundary value               -> the segment's offset
//   2. one wave chains the boundaries: value_out = P_s(') value_in + offset   (P_s = product of the segment's Abar_t,
//      computed once per factorisation by k_seg_products)
//   3. every wave sweeps its segment again from its true boundary value       -> kw_t (backward), dx_t, w_t (forward)
// Two passes over T / kSeg nodes + kSeg boundary steps instead of one pass over T nodes.  Measured at B = 256, T = 200
// (config 3): 95 -> 68 us per sweep, not the 3 x the node counts promise: with 8 waves per instance every SIMD holds two
// of them and the recursion is issue / cache-latency bound (0.8 us per node and wave against 0.3 us alone); 4 segments and
// deeper prefetch (8 nodes, spills) were slower.  AGX_ADMM_SEGMENTS=0 selects the one-wave sweep.
// ---------------------------------------------------------------------------
constexpr int kSeg = 8;
#ifndef AGX_VEC_DEPTH
#define AGX_VEC_DEPTH 4
#endif
constexpr int kVecDepth = AGX_VEC_DEPTH;  // nodes of factors / gains in flight per wave in the segment passes
__host__ __device__ inline int seg_len(int T) { return (T + kSeg - 1) / kSeg; }

// P_s = Abar_{b-1} ... Abar_a of segment s = [a, b) on the 8 x 8 lane grid (blocks qq | qv | vq | vv, element [r][c] on
// lane 8 r + c), from the gains Kw of the last factorisation.  One wave per (instance, segment).
template <int NV>
__global__ void __launch_bounds__(64) k_seg_products(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                     const double *__restrict__ Kws, double *__restrict__ segP,
                                                     const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV;
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x / kSeg, sg = blockIdx.x % kSeg, lane = threadIdx.x;
  const DevState &S = st[b];
  if (S.done || S.admm_conv || !S.admm_refactor || S.dir_fail) return;  // (dir_fail: the factorisation these products belong to broke down)
  const int L = seg_len(T), ta = sg * L, tb = min(T, ta + L);
  const int r = lane >> 3, c = lane & 7;
  const bool in = (r < NV) && (c < NV);
  const double *Kw = Kws + (long long)b * T * NV * NX;
  double Pqq = (in && r == c) ? 1.0 : 0.0, Pqv = 0.0, Pvq = 0.0, Pvv = Pqq;
  for (int t = ta; t < tb; ++t) {
    const double h = dts[t], h2 = h * h;
    const double *kr = Kw + ((long long)t * NV + (r < NV ? r : 0)) * NX;
    const double Kq = in ? kr[c] : 0.0, Kv = in ? kr[NV + c] : 0.0;
    double Xq = 0.0, Xv = 0.0;
    auto acc = [&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      if (k >= NV) return;
      const double kq = grid_col<k>(Kq), kv = grid_col<k>(Kv);  // K[r][k]
      const double pqq = __shfl(Pqq, 8 * k + c, 64), pqv = __shfl(Pqv, 8 * k + c, 64), pvq = __shfl(Pvq, 8 * k + c, 64), pvv = __shfl(Pvv, 8 * k + c, 64);  // P[k][c]
      Xq += kq * pqq + kv * pvq;
      Xv += kq * pqv + kv * pvv;
    };
    acc(std::integral_constant<int, 0>()); acc(std::integral_constant<int, 1>()); acc(std::integral_constant<int, 2>());
    acc(std::integral_constant<int, 3>()); acc(std::integral_constant<int, 4>()); acc(std::integral_constant<int, 5>());
    acc(std::integral_constant<int, 6>());
    const double nqq = Pqq + h * Pvq - h2 * Xq, nqv = Pqv + h * Pvv - h2 * Xv;
    Pvq = Pvq - h * Xq; Pvv = Pvv - h * Xv;
    Pqq = nqq; Pqv = nqv;
    if (!in) { Pqq = 0.0; Pqv = 0.0; Pvq = 0.0; Pvv = 0.0; }
  }
  double *P = segP + ((long long)b * kSeg + sg) * 256;
  P[lane] = Pqq; P[64 + lane] = Pqv; P[128 + lane] = Pvq; P[192 + lane] = Pvv;
}

// backward gradient recursion over the nodes t_hi-1 .. t_lo of one segment (see riccati_vec_body); v: the value gradient
// on the lanes c = 1 (q) and c = 2 (v) of grid row r, in: boundary value, out: value at t_lo
template <int NV, bool STORE_KW>
__device__ __forceinline__ double vec_backward_seg(const int t_lo, const int t_hi, double v, const double *s_dt,
                                                   const double *__restrict__ qb, const double *__restrict__ fb, double *__restrict__ kw) {
  constexpr int TS = QT<NV>::SIZE;
  typedef QT<NV> Q;
  typedef FT<NV> F;
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
  const int rr = r < NV ? r : 0;
  const int sel = c < 3 ? c : 0;
  const int goff = sel == 0 ? Q::gw : (sel == 1 ? Q::gx : Q::gx + NV);
  const int foff = sel == 0 ? F::FW : (sel == 1 ? F::FQ : F::FV);
  const int poff = sel == 2 ? F::PV : F::PQ;
  struct Node { double g, rp, p, f[NV]; };
  auto load_node = [&](Node &z, int t) {
    const double *tl = qb + (long long)t * TS;
    const double *ft = fb + (long long)t * F::SIZE;
    z.g = tl[goff + rr];
    z.rp = ft[F::RP + rr]; z.p = ft[poff + rr];
#pragma unroll
    for (int k = 0; k < NV; ++k) z.f[k] = ft[foff + k * 8 + rr];
  };
  constexpr int D = kVecDepth;
  auto step = [&](Node &z, int t, bool reload) {
    const double h = s_dt[t], h2 = h * h;
    const double vp = v + z.p;
    const double vpq = dpp_mov<0x55>(vp), vpv = dpp_mov<0xAA>(vp);  // quad broadcast of lanes c = 1 / c = 2
    const double ca = sel == 0 ? h2 : (sel == 1 ? 1.0 : h), cb = sel == 0 ? h : (sel == 1 ? 0.0 : 1.0);
    double g = z.g + ca * vpq + cb * vpv;
    const double rp = z.rp;
    double f[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) f[k] = z.f[k];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double gk = readlane_f64(g, 8 * k);
      g -= f[k] * gk;
    }
    if (STORE_KW && c == 0 && r < NV) kw[(long long)t * NV + r] = g * rp;
    v = g;
    if (reload) {
      prefetch_group_begin();
      load_node(z, t - D >= t_lo ? t - D : t_lo);  // unconditional, after the last use of the old contents (see agx_riccati_mx.hpp)
      prefetch_group_end();
    }
  };
  if (t_hi <= t_lo) return v;
  // whole groups of D nodes in the pipelined loop; the nodes that do not fill a group come LAST: the reloads of the final
  // group have fetched them already (a segment is 25 nodes: in front of the loop the odd node cost one exposed load latency)
  const int rem = (t_hi - t_lo) % D;
  int t = t_hi - 1;
  Node n[D];
#pragma unroll
  for (int i = 0; i < D; ++i) load_node(n[i], t - i >= t_lo ? t - i : t_lo);
  prefetch_queue_settle(n);
  for (; t - (D - 1) >= t_lo; t -= D) {
#pragma unroll
    for (int i = 0; i < D; ++i) step(n[i], t - i, true);
  }
#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (i < rem) step(n[i], t - i, false);
  return v;
}

// forward pass over the nodes t_lo .. t_hi-1 of one segment (see riccati_forward); the state enters / leaves indexed by the
// lane's grid row (dq_r, dv_r)
template <int NV, bool STORE>
__device__ __forceinline__ void forward_seg(const int t_lo, const int t_hi, double &dq_r, double &dv_r, const double *s_dt,
                                            const double *__restrict__ qb, const double *__restrict__ Kw, const double *__restrict__ kw,
                                            double *__restrict__ dx, double *__restrict__ ws) {
  constexpr int NX = 2 * NV, TS = QT<NV>::SIZE;
  typedef QT<NV> Q;
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
  const bool in = (r < NV) && (c < NV);
  const double inm = in ? 1.0 : 0.0;
  const int rr = r < NV ? r : 0, cc = c < NV ? c : 0;
  double dq_c = __shfl(dq_r, 8 * cc, 64), dv_c = __shfl(dv_r, 8 * cc, 64);
  struct Gain { double kq, kv, kw, fq, fv; };
  constexpr int DEPTH = kVecDepth;
  auto load_gain = [&](Gain &g, int t) {
    const double *kr = Kw + ((long long)t * NV + rr) * NX;
    g.kq = kr[cc];  // masked at the use (see riccati_forward)
    g.kv = kr[NV + cc];
    g.kw = kw[(long long)t * NV + rr];
    g.fq = qb[(long long)t * TS + Q::f + rr];
    g.fv = qb[(long long)t * TS + Q::f + NV + rr];
  };
  auto fstep = [&](Gain &g, int t, bool reload) {
    const double h = s_dt[t], h2 = h * h;
    double p = (g.kq * inm) * dq_c + (g.kv * inm) * dv_c;
    const double kwv = g.kw, fqc = g.fq, fvc = g.fv;
    p += dpp_xor1(p); p += dpp_xor2(p); p += dpp_xor4(p);
    const double wv = -(kwv + p);
    const double nq = dq_r + h * dv_r + h2 * wv + fqc;
    const double nv2 = dv_r + h * wv + fvc;
    dq_r = nq; dv_r = nv2;
    dq_c = __shfl(nq, 8 * cc, 64);
    dv_c = __shfl(nv2, 8 * cc, 64);
    if (STORE && c == 0 && r < NV) {
      ws[(long long)t * NV + r] = wv;
      dx[(long long)(t + 1) * NX + r] = nq;
      dx[(long long)(t + 1) * NX + NV + r] = nv2;
    }
    if (reload) {
      prefetch_group_begin();
      load_gain(g, t + DEPTH < t_hi ? t + DEPTH : t_hi - 1);  // unconditional, at the end (see riccati_forward)
      prefetch_group_end();
    }
  };
  if (t_hi <= t_lo) return;
  const int rem = (t_hi - t_lo) % DEPTH;  // the nodes that do not fill a group come last (see vec_backward_seg)
  int t = t_lo;
  Gain g[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) load_gain(g[i], t + i < t_hi ? t + i : t_hi - 1);
  prefetch_queue_settle(g);
  for (; t + (DEPTH - 1) < t_hi; t += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) fstep(g[i], t + i, true);
  }
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i)
    if (i < rem) fstep(g[i], t + i, false);
}

template <int NV>
__device__ __forceinline__ void riccati_vec_segments(const int b, const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                     const double *__restrict__ qts, const double *__restrict__ Kws,
                                                     double *__restrict__ kws, double *__restrict__ dxs, double *__restrict__ wss,
                                                     const double *__restrict__ facs, const double *__restrict__ segP) {
  constexpr int NX = 2 * NV, TS = QT<NV>::SIZE;
  typedef QT<NV> Q;
  typedef FT<NV> F;
  __shared__ double s_off[kSeg][2][8], s_in[kSeg][2][8];  // segment offsets / boundary values (q | v), indexed by joint
  __shared__ double s_dt[kMaxHorizon];                     // step lengths (see stage_dts)
  const DevOcp &o = *op;
  const int T = o.T, lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < T; i += blockDim.x) s_dt[i] = dts[i];
  __syncthreads();
  const int r = lane >> 3, c = lane & 7;
  const int rr = r < NV ? r : 0, cc = c < NV ? c : 0;
  const int L = seg_len(T), ta = min(T, sg * L), tb = min(T, ta + L);
  const double *qb = qts + (long long)b * (T + 1) * TS;
  const double *Kw = Kws + (long long)b * T * NV * NX;
  double *kw = kws + (long long)b * T * NV;
  const double *fb = facs + (long long)b * T * F::SIZE;
  const int sel = c < 3 ? c : 0;
  // the wave that chains the boundaries fetches every segment's product now: the loads are in flight during pass 1
  double pp[kSeg][4];
  if (sg == 0) {
#pragma unroll
    for (int s2 = 0; s2 < kSeg; ++s2) {
      const double *Ps = segP + ((long long)b * kSeg + s2) * 256;
      pp[s2][0] = Ps[lane]; pp[s2][1] = Ps[64 + lane]; pp[s2][2] = Ps[128 + lane]; pp[s2][3] = Ps[192 + lane];
    }
  }
  // ---- backward 1: offsets from a zero boundary value
  {
    const double v = vec_backward_seg<NV, false>(ta, tb, 0.0, s_dt, qb, fb, kw);
    if ((c == 1 || c == 2) && r < 8) s_off[sg][c - 1][r] = (r < NV) ? v : 0.0;
  }
  __syncthreads();
  // ---- backward 2: boundary values, last segment first:  v_start(s) = P_s' v_in(s) + offset(s)
  if (sg == 0) {
    double vq_r = (r < NV) ? qb[(long long)T * TS + Q::gx + rr] : 0.0, vv_r = (r < NV) ? qb[(long long)T * TS + Q::gx + NV + rr] : 0.0;  // by grid row
#pragma unroll
    for (int s2 = kSeg - 1; s2 >= 0; --s2) {
      if (c == 0 && r < 8) { s_in[s2][0][r] = vq_r; s_in[s2][1][r] = vv_r; }
      const double pqq = pp[s2][0], pqv = pp[s2][1], pvq = pp[s2][2], pvv = pp[s2][3];
      double nq = pqq * vq_r + pvq * vv_r, nv2 = pqv * vq_r + pvv * vv_r;  // column sums over r: (P' v)[c]
      nq += __shfl_xor(nq, 8, 64); nv2 += __shfl_xor(nv2, 8, 64);
      nq += __shfl_xor(nq, 16, 64); nv2 += __shfl_xor(nv2, 16, 64);
      nq += __shfl_xor(nq, 32, 64); nv2 += __shfl_xor(nv2, 32, 64);
      // back to "indexed by grid row": lane (r, *) takes column r's value, plus the segment's offset
      vq_r = __shfl(nq, rr, 64) + s_off[s2][0][rr];
      vv_r = __shfl(nv2, rr, 64) + s_off[s2][1][rr];
      if (r >= NV) { vq_r = 0.0; vv_r = 0.0; }
    }
  }
  __syncthreads();
  // ---- backward 3: the segment again from its true boundary value, gains feed-forward kw stored
  {
    const double v0 = s_in[sg][sel == 2 ? 1 : 0][rr];
    vec_backward_seg<NV, true>(ta, tb, v0, s_dt, qb, fb, kw);
  }
  __threadfence_block();  // kw of this segment is read back by other lanes of the wave below
  // ---- forward 1: offsets from a zero state
  {
    double dq = 0.0, dv = 0.0;
    forward_seg<NV, false>(ta, tb, dq, dv, s_dt, qb, Kw, kw, nullptr, nullptr);
    __syncthreads();  // s_off is free again (everyone is past backward 2 / 3 reads of it)
    if (c == 0 && r < 8) { s_off[sg][0][r] = dq; s_off[sg][1][r] = dv; }
  }
  __syncthreads();
  // ---- forward 2: boundary states, first segment first:  dx_in(s + 1) = P_s dx_in(s) + offset(s)
  if (sg == 0) {
    double dq_r = 0.0, dv_r = 0.0;  // dx_0 = 0
#pragma unroll
    for (int s2 = 0; s2 < kSeg; ++s2) {
      if (c == 0 && r < 8) { s_in[s2][0][r] = dq_r; s_in[s2][1][r] = dv_r; }
      const double dq_c = __shfl(dq_r, 8 * cc, 64), dv_c = __shfl(dv_r, 8 * cc, 64);
      double nq = pp[s2][0] * dq_c + pp[s2][1] * dv_c, nv2 = pp[s2][2] * dq_c + pp[s2][3] * dv_c;  // row sums over c
      nq += dpp_xor1(nq); nv2 += dpp_xor1(nv2);
      nq += dpp_xor2(nq); nv2 += dpp_xor2(nv2);
      nq += dpp_xor4(nq); nv2 += dpp_xor4(nv2);
      dq_r = nq + s_off[s2][0][rr];
      dv_r = nv2 + s_off[s2][1][rr];
      if (r >= NV) { dq_r = 0.0; dv_r = 0.0; }
    }
  }
  __syncthreads();
  // ---- forward 3: the segment again from its true state
  {
    double dq = s_in[sg][0][rr], dv = s_in[sg][1][rr];
    double *dx = dxs + (long long)b * (T + 1) * NX, *ws = wss + (long long)b * T * NV;
    if (sg == 0 && lane < NV) { dx[lane] = 0.0; dx[NV + lane] = 0.0; }
    forward_seg<NV, true>(ta, tb, dq, dv, s_dt, qb, Kw, kw, dx, ws);
  }
}

// One workgroup of kSeg waves per instance: a factorisation sweep runs on wave 0 alone (the Riccati recursion itself is
// not affine in the value function), the gradient-only sweeps on all of them.
template <int NV>
__global__ void __launch_bounds__(64 * kSeg) k_riccati_admm(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                            const double *__restrict__ qts, const double *__restrict__ auxs,
                                                            double *__restrict__ Kws, double *__restrict__ kws,
                                                            double *__restrict__ dxs, double *__restrict__ wss,
                                                            double *__restrict__ dus, double *__restrict__ Kout,
                                                            DevState *__restrict__ st, double *__restrict__ facs,
                                                            const double *__restrict__ segP, int force_vec) {
  const int b = blockIdx.x;
  const DevState &S = st[b];
  if (S.done || S.admm_conv) return;
  // the factorisation of this iteration (k_riccati_lqr_prefactor, launched before this kernel) broke down: k_admm_reduce stops
  // the instance and k_sqp_head discards the direction -- no gradient sweep on factors that do not exist
  if (force_vec && S.dir_fail) return;
  if (S.admm_refactor && !force_vec) {  // force_vec: the factors of this Hessian exist already (k_riccati_lqr_prefactor)
    if (threadIdx.x < 64) riccati_body<NV, false, true>(b, op, dts, qts, auxs, Kws, kws, dxs, wss, dus, Kout, st, 1, 0, 0, facs);
  } else if (segP) {
    riccati_vec_segments<NV>(b, op, dts, qts, Kws, kws, dxs, wss, facs, segP);
  } else if (threadIdx.x < 64) {
    riccati_vec_body<NV>(b, op, dts, qts, Kws, kws, dxs, wss, facs);
  }
}

// The plain LQR pass of a constrained SQP iteration (equality-QP initial guess) and the factorisation of the
// augmented Hessians of its first ADMM iteration in ONE launch: neither needs the other (only the augmented GRADIENT
// needs the LQR pass), and at the batch sizes of the constrained workloads (B = 256: a quarter of the SIMDs) the two
// waves of an instance run side by side.  Even workgroups: LQR sweep + forward pass on the base tiles, gains to a
// scratch buffer; odd ones: Gauss-Jordan sweep on the augmented tiles leaving factors and gains, no forward pass.
// The first ADMM iteration is then a gradient-only sweep like the others.
template <int NV>
__global__ void __launch_bounds__(64, 2) k_riccati_lqr_prefactor(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                                 const double *__restrict__ qts, const double *__restrict__ qt2s,
                                                                 const double *__restrict__ auxs, double *__restrict__ Kws,
                                                                 double *__restrict__ kws, double *__restrict__ Kws_lqr,
                                                                 double *__restrict__ kws_lqr, double *__restrict__ dxs,
                                                                 double *__restrict__ wss, double *__restrict__ dus,
                                                                 double *__restrict__ Kout, DevState *__restrict__ st,
                                                                 double *__restrict__ facs) {
  const int b = blockIdx.x >> 1;
  if (blockIdx.x & 1)
    riccati_body<NV, false, true>(b, op, dts, qt2s, auxs, Kws, kws, dxs, wss, dus, Kout, st, 0, 0, 0, facs);
  else
    riccati_mx_body<NV, false>(b, op, dts, qts, auxs, Kws_lqr, kws_lqr, dxs, wss, Kout, st, 1, 3, 0);
}

// After the Riccati sweep on the augmented tiles: du, the multiplier update and the node's shares of
// the ADMM residual norms and of the KKT residual.  8 lanes per node, lane j = component j.
//   z_rel = alpha C d + (1 - alpha) z;  z = clip(z_rel + y / rho, lb - g, ub - g);  y += rho (z_rel - z)
// KKT share through the optimality identity of the augmented QP (constraints):
//   Lu + Fu' lam' + Gu' y = -[(Luu + preg) du + sigma (du - du_c) + Gu' (rho C d + h - y)]
//   Lx + Fx' lam' - lam + Gx' y = -[(Lxx + dreg) dx + sigma (dx - dx_c) + Gx' (rho C d + h - y)],  h = y_old - rho z_old
// (device function: called by k_admm_update for the whole batch and by k_admm_loop for the nodes of its instance; out4: the
// node's primal / dual residual norms and their scales, zero for lanes without a node)
template <int NV>
__device__ __forceinline__ void admm_update_node(const DevOcp *op, const long long node, const bool ok_in, const double *qts,
                                                 const double *auxs, const double *dxs, const double *wss, double *dus,
                                                 double *cxs, const double *cg, const double *cjac, double *ys, double *zs,
                                                 double *nodestat, double *admmstat, double *qt2s, const DevState *st,
                                                 double *out4, const double *auxg = nullptr) {
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T;
  const int l8 = threadIdx.x & 7;
  const long long n_nodes = (long long)o.B * (T + 1);
  const bool ok = ok_in && node < n_nodes;
  out4[0] = 0.0; out4[1] = 0.0; out4[2] = 0.0; out4[3] = 0.0;
  const long long nid = ok ? node : 0;
  const int b = (int)(nid / (T + 1)), t = (int)(nid % (T + 1));
  const DevState &S = st[b];
  // dir_fail: the sweep of this iteration broke down (k_admm_reduce stops the instance): multipliers stay
  const bool act = ok && !S.done && !S.admm_conv && !S.dir_fail;
  if (!__any(act)) return;
  const double preg = S.preg, dreg = S.dreg, rs = S.rho_sparse, sig = kSigma;
  const bool jl = l8 < NV;
  const int jj = jl ? l8 : 0;
  const double *qt = qts + nid * Q::SIZE;
  const double *ax = auxs + nid * A::SIZE;
  const double *dx = dxs + nid * NX;
  double *cx = cxs + nid * NX;
  double *y = ys + nid * AGX_MAX_NC, *z = zs + nid * AGX_MAX_NC;
  const double *g = cg + nid * AGX_MAX_NC;
  const DevCons &c = o.cons[t == T ? 1 : 0];
  const double dq = jl ? dx[jj] : 0.0, dv = jl ? dx[NV + jj] : 0.0;
  const double cq = jl ? cx[jj] : 0.0, cv = jl ? cx[NV + jj] : 0.0;
  double du = 0.0, duc = 0.0, kkt = 0.0, gap = 0.0;
  double Mr[8], tqr[8], tvr[8];  // M[i][l8], taux[i][l8]: operands of du here and of the next gradient below
  if (t < T) {
    const double wj = jl ? wss[((long long)b * T + t) * NV + jj] : 0.0;
    const double fq = jl ? qt[Q::f + jj] : 0.0, fv = jl ? qt[Q::f + NV + jj] : 0.0;
    kkt = fmax(fabs(fq), fabs(fv));
    gap = fabs(fq) + fabs(fv);
    double pr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      Mr[i] = (i < NV) ? ax[A::M + i * A::LD + l8] : 0.0;
      tqr[i] = (i < NV) ? ax[A::tq + i * A::LD + l8] : 0.0;
      tvr[i] = (i < NV) ? ax[A::tv + i * A::LD + l8] : 0.0;
      pr[i] = Mr[i] * wj + tqr[i] * dq + tvr[i] * dv;
    }
    du = transpose_reduce8(pr, l8);
    duc = jl ? dus[((long long)b * T + t) * NV + jj] : 0.0;
  }
  // constraint rows: lane-local for State / Control components, group-wide for the collision scalar
  double primal = 0.0, primal_rel = 0.0;
  double dual_q = 0.0, dual_v = 0.0, dual_u = 0.0, drel_q = 0.0, drel_v = 0.0, drel_u = 0.0;  // (G' rho dz)_j, (G' y)_j
  double e_q = 0.0, e_v = 0.0, e_u = 0.0;                                                   // (G' (rho C d + h - y))_j
  double hn_q = 0.0, hn_v = 0.0, hn_u = 0.0;  // (G' (y - rho z))_j with the updated y, z: the next iteration's gradient terms
  auto comp = [&](int k, double Cd, double &dual, double &drel, double &e, double jac) {
    const double rho = admm_rho(c.lb[k], c.ub[k], rs);
    const double z0 = z[k], y0 = y[k];
    const double zrel = kAlphaRelax * Cd + (1.0 - kAlphaRelax) * z0;
    double zn = zrel + y0 / rho;
    zn = fmin(fmax(zn, c.lb[k] - g[k]), c.ub[k] - g[k]);
    const double yn = y0 + rho * (zrel - zn);
    primal = fmax(primal, fabs(Cd - zn));
    primal_rel = fmax(primal_rel, fmax(fabs(Cd), fabs(zn)));
    dual += jac * rho * (zn - z0);
    drel += jac * yn;
    e += jac * (rho * Cd + (y0 - rho * z0) - yn);
    return (double2){zn, yn};
  };
  for (int r = 0; r < c.n; ++r) {
    const int off = c.off[r];
    if (c.kind[r] == AGX_RES_CONTROL) {
      if (jl && t < T) {
        const double2 zy = comp(off + jj, du, dual_u, drel_u, e_u, 1.0);
        hn_u += zy.y - admm_rho(c.lb[off + jj], c.ub[off + jj], rs) * zy.x;
        if (act) { z[off + jj] = zy.x; y[off + jj] = zy.y; }
      }
    } else if (c.kind[r] == AGX_RES_STATE) {
      if (jl) {
        const double2 a = comp(off + jj, dq, dual_q, drel_q, e_q, 1.0);
        const double2 bq = comp(off + NV + jj, dv, dual_v, drel_v, e_v, 1.0);
        hn_q += a.y - admm_rho(c.lb[off + jj], c.ub[off + jj], rs) * a.x;
        hn_v += bq.y - admm_rho(c.lb[off + NV + jj], c.ub[off + NV + jj], rs) * bq.x;
        if (act) { z[off + jj] = a.x; y[off + jj] = a.y; z[off + NV + jj] = bq.x; y[off + NV + jj] = bq.y; }
      }
    } else if (cons_has_dense_rows(c.kind[r])) {
      for (int e = 0; e < c.nr[r]; ++e) {
        const double *row = cjac + (nid * AGX_MAX_DENSE + c.coll_slot[r] + e) * 24;
        const double gqj = jl ? row[jj] : 0.0, gvj = jl ? row[8 + jj] : 0.0, guj = (jl && t < T) ? row[16 + jj] : 0.0;
        double Cd = gqj * dq + gvj * dv + guj * du;
        Cd += dpp_xor4(Cd); Cd += dpp_xor2(Cd); Cd += dpp_xor1(Cd);
        // identical on every lane of the group; the three accumulator sets take the row's q / v / u entries
        const double2 zy = comp(off + e, Cd, dual_q, drel_q, e_q, gqj);
        const double rho = admm_rho(c.lb[off + e], c.ub[off + e], rs);
        const double z0 = z[off + e], y0 = y[off + e];
        const double hn = zy.y - rho * zy.x, de = rho * Cd + (y0 - rho * z0) - zy.y, dz = rho * (zy.x - z0);
        dual_v += gvj * dz; drel_v += gvj * zy.y; e_v += gvj * de;
        dual_u += guj * dz; drel_u += guj * zy.y; e_u += guj * de;
        hn_q += hn * gqj; hn_v += hn * gvj; hn_u += hn * guj;
        if (act && l8 == 0) { z[off + e] = zy.x; y[off + e] = zy.y; }
      }
    }
  }
  // general cost rows (ControlGrav / FrameVelocity, agx_general.hpp): the blocks Lqv | Lvvd | Lqu of the node's Hessian in the
  // optimality identities (as k_node_kkt_gen);  row i on lane i, the operands of the other lanes through the 8-lane shuffle
  double gen_u = 0.0, gen_q = 0.0, gen_v = 0.0;
  if (auxg) {
    const double *ag = auxg + nid * (3 * A::B2);
    const double *Lqv = ag, *Lvvd = ag + A::B2, *Lqu = ag + 2 * A::B2;
#pragma unroll
    for (int l = 0; l < NV; ++l) {
      const double dq_l = __shfl(dq, l, 8), dv_l = __shfl(dv, l, 8), du_l = __shfl(du, l, 8);
      gen_u += Lqu[l * A::LD + jj] * dq_l;
      gen_q += Lqv[jj * A::LD + l] * dv_l + Lqu[jj * A::LD + l] * du_l;
      gen_v += Lqv[l * A::LD + jj] * dq_l + Lvvd[jj * A::LD + l] * dv_l;
    }
  }
  // KKT shares
  if (t < T && jl) kkt = fmax(kkt, fabs((ax[A::Luu + l8] + preg) * du + gen_u + sig * (du - duc) + e_u));
  if (t > 0) {
    double pr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pr[i] = (i < NV) ? ax[A::Lqq + i * A::LD + l8] * dq : 0.0;
    const double hq = transpose_reduce8(pr, l8);
    if (jl) {
      kkt = fmax(kkt, fabs(hq + gen_q + dreg * dq + sig * (dq - cq) + e_q));
      kkt = fmax(kkt, fabs((ax[A::Lvv + l8] + dreg) * dv + gen_v + sig * (dv - cv) + e_v));
    }
  }
  double dual = fmax(fabs(dual_q), fmax(fabs(dual_v), fabs(dual_u)));
  double drel = fmax(fabs(drel_q), fmax(fabs(drel_v), fabs(drel_u)));
  // group reductions (fixed order)
  kkt = fmax(kkt, dpp_xor4(kkt)); kkt = fmax(kkt, dpp_xor2(kkt)); kkt = fmax(kkt, dpp_xor1(kkt));
  gap += dpp_xor4(gap); gap += dpp_xor2(gap); gap += dpp_xor1(gap);
  primal = fmax(primal, dpp_xor4(primal)); primal = fmax(primal, dpp_xor2(primal)); primal = fmax(primal, dpp_xor1(primal));
  primal_rel = fmax(primal_rel, dpp_xor4(primal_rel)); primal_rel = fmax(primal_rel, dpp_xor2(primal_rel)); primal_rel = fmax(primal_rel, dpp_xor1(primal_rel));
  dual = fmax(dual, dpp_xor4(dual)); dual = fmax(dual, dpp_xor2(dual)); dual = fmax(dual, dpp_xor1(dual));
  drel = fmax(drel, dpp_xor4(drel)); drel = fmax(drel, dpp_xor2(drel)); drel = fmax(drel, dpp_xor1(drel));
  // Gradient of the next ADMM iteration's augmented tile (what k_admm_tile computes from y, z, cx, du),
  // valid while rho stays: after a rho update k_admm_tile rebuilds Hessian and gradient.
  //   g = g0 + [taux M]' (h_u - sigma du) - sigma dx + h_x
  double gwn = 0.0, gqn = hn_q - sig * dq, gvn = hn_v - sig * dv;
  if (t < T) {
    const double e_own = hn_u - sig * du;
#pragma unroll
    for (int l = 0; l < NV; ++l) {
      const double e = __shfl(e_own, l, 8);
      gwn += Mr[l] * e; gqn += tqr[l] * e; gvn += tvr[l] * e;
    }
  }
  if (act) {
    if (jl) {
      double *q2 = qt2s + nid * Q::SIZE;
      if (t < T) q2[Q::gw + l8] = qt[Q::gw + l8] + gwn;
      q2[Q::gx + l8] = qt[Q::gx + l8] + gqn;
      q2[Q::gx + NV + l8] = qt[Q::gx + NV + l8] + gvn;
      if (t < T) dus[((long long)b * T + t) * NV + l8] = du;
      cx[l8] = dq; cx[NV + l8] = dv;  // prox centre of the next iteration
    }
    if (l8 == 0) {
      double *ns = nodestat + nid * 4;
      ns[0] = kkt; ns[1] = qt[Q::cost]; ns[2] = gap;
      double *as = admmstat + nid * 4;
      as[0] = primal; as[1] = dual; as[2] = primal_rel; as[3] = drel;
    }
    out4[0] = primal; out4[1] = dual; out4[2] = primal_rel; out4[3] = drel;
  }
}

template <int NV>
__global__ void __launch_bounds__(256) k_admm_update(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                     const double *__restrict__ auxs, const double *__restrict__ dxs,
                                                     const double *__restrict__ wss, double *__restrict__ dus,
                                                     double *__restrict__ cxs, const double *__restrict__ cg,
                                                     const double *__restrict__ cjac, double *__restrict__ ys,
                                                     double *__restrict__ zs, double *__restrict__ nodestat,
                                                     double *__restrict__ admmstat, double *__restrict__ qt2s,
                                                     const DevState *__restrict__ st, const double *__restrict__ auxg) {
  double out4[4];
  admm_update_node<NV>(op, ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3, true, qts, auxs, dxs, wss, dus, cxs, cg, cjac, ys, zs,
                       nodestat, admmstat, qt2s, st, out4, auxg);
}


// One thread per instance: rho adaptation (update_rho_vec) and the convergence test from the residual norms of iteration `iter`.
// Returns 1 when the QP has converged (or ran into max_qp), 2 when rho changed (the next sweep factorises again), else 0.
__device__ __forceinline__ int admm_reduce_decide(const DevOcp &o, DevState &S, const double np_, const double nd, const double npr,
                                                  const double ndr, const int iter, int *__restrict__ n_conv) {
  // update_rho_vec (std::max / std::min semantics for the 0/0 cases)
  const double scale = sqrt((np_ * ndr) / (nd * npr));
  double est = scale * S.rho_sparse;
  est = (est < kRhoMin) ? kRhoMin : est;
  est = (kRhoMax < est) ? kRhoMax : est;
  int refactor = 0;
  if (iter % kRhoInterval == 0 && iter > 1)
    if (est > S.rho_sparse * kAdaptiveRhoTol || est < S.rho_sparse / kAdaptiveRhoTol) { S.rho_sparse = est; refactor = 1; }
  S.admm_refactor = refactor;  // unchanged rho: the next sweep only redoes the gradient recursion
  const bool conv = (np_ <= o.eps_abs + o.eps_rel * npr) && (nd <= o.eps_abs + o.eps_rel * ndr);
  if (conv || iter == o.max_qp) {
    S.admm_conv = 1;
    S.admm_iter = conv ? iter : o.max_qp;
    atomicAdd(n_conv, 1);
    return 1;
  }
  return refactor ? 2 : 0;
}

// Per instance: residual norms over the nodes, rho adaptation (update_rho_vec), convergence.
__global__ void __launch_bounds__(128) k_admm_reduce(const DevOcp *__restrict__ op, const double *__restrict__ admmstat,
                                                     DevState *__restrict__ st, int iter, int *__restrict__ n_conv) {
  __shared__ double red[4][2];
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x;
  DevState &S = st[b];
  if (S.done || S.admm_conv) return;
  if (S.dir_fail) {  // the factorisation of this instance's augmented problem broke down: nothing to iterate on
    if (tid == 0) { S.admm_conv = 1; S.admm_iter = iter; atomicAdd(n_conv, 1); }
    return;
  }
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int t = tid; t <= T; t += blockDim.x) {
    const double *as = admmstat + ((long long)b * (T + 1) + t) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmax(v[k], as[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_max(v[k]);
  if ((tid & 63) == 0)
    for (int k = 0; k < 4; ++k) red[k][tid >> 6] = v[k];
  __syncthreads();
  if (tid != 0) return;
  admm_reduce_decide(o, S, fmax(red[0][0], red[0][1]), fmax(red[1][0], red[1][1]), fmax(red[2][0], red[2][1]), fmax(red[3][0], red[3][1]), iter, n_conv);
}

// Several ADMM iterations of one instance in ONE launch (iterations first .. last, all of them gradient-only sweeps: the
// host runs the iterations that may factorise again -- the first of an SQP iteration and those after a rho check, every
// kRhoInterval -- through k_riccati_admm / k_admm_update / k_admm_reduce): the segment-parallel sweep, the node update on
// the same workgroup (eight lanes per node, 64 kSeg / 8 nodes per pass) and the norms / convergence test without leaving
// the CU.  Three dependent launches per iteration (4.5 us each from dispatch to completion) and the host's poll every
// four iterations go away; an instance leaves the loop at its own convergence.  Quorum < 1 keeps the host's schedule
// (chunks of four iterations: which instances are cut must not depend on how the workgroups happen to progress).
template <int NV>
__global__ void __launch_bounds__(64 * kSeg) k_admm_loop(const DevOcp *op, const double *dts, const double *qts, double *qt2s,
                                                         const double *auxs, const double *Kws, double *kws, double *dxs,
                                                         double *wss, double *dus, double *cxs, const double *cg,
                                                         const double *cjac, double *ys, double *zs, double *nodestat,
                                                         double *admmstat, DevState *st, const double *facs, const double *segP,
                                                         int first, int last, int *n_conv, const double *auxg) {
  __shared__ double s_red[4][kSeg];
  __shared__ int s_stop;
  const DevOcp &o = *op;
  const int b = blockIdx.x, T = o.T, tid = threadIdx.x;
  DevState &S = st[b];
  if (S.done || S.admm_conv || S.dir_fail || S.admm_refactor) return;  // (refactor: left to the host's factorising iteration)
  for (int iter = first; iter <= last; ++iter) {
    riccati_vec_segments<NV>(b, op, dts, qt2s, Kws, kws, dxs, wss, facs, segP);
    __threadfence_block();
    __syncthreads();  // dx, w of every segment are visible to the whole workgroup
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base <= T; base += 8 * kSeg) {
      const int t = base + (tid >> 3);
      double out4[4];
      admm_update_node<NV>(op, (long long)b * (T + 1) + (t <= T ? t : T), t <= T, qts, auxs, dxs, wss, dus, cxs, cg, cjac, ys, zs, nodestat,
                           admmstat, qt2s, st, out4, auxg);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmax(v[k], out4[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_max(v[k]);
    if ((tid & 63) == 0)
      for (int k = 0; k < 4; ++k) s_red[k][tid >> 6] = v[k];
    __threadfence_block();
    __syncthreads();  // the next gradient (qt2s), multipliers and prox centre are written; norms are in LDS
    if (tid == 0) {
      double m[4];
      for (int k = 0; k < 4; ++k) {
        m[k] = s_red[k][0];
        for (int w = 1; w < kSeg; ++w) m[k] = fmax(m[k], s_red[k][w]);
      }
      s_stop = admm_reduce_decide(o, S, m[0], m[1], m[2], m[3], iter, n_conv);
    }
    __syncthreads();
    if (s_stop) break;
  }
}

// The host left the ADMM loop at iteration `iter` with a quorum of converged QPs (agx_ocp_set_quorum): the others
// go on with the iterate they have, and report the iterations they ran.
__global__ void k_admm_cap(DevState *__restrict__ st, int B, int iter) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  DevState &S = st[b];
  if (!S.done && !S.admm_conv) S.admm_iter = iter;
}

// K = M Kw - taux: one lane per (node, column of K); the gains of the last ADMM backward pass
template <int NV>
__global__ void __launch_bounds__(256) k_gains_to_u(const DevOcp *__restrict__ op, const double *__restrict__ auxs,
                                                    const double *__restrict__ Kws, double *__restrict__ Kout,
                                                    const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long node = unit >> 4;  // 16 lanes per node, NX <= 16 of them active
  const int j = (int)(unit & 15);
  if (node >= (long long)o.B * T || j >= NX) return;
  const int b = (int)(node / T), t = (int)(node % T);
  if (st[b].done) return;
  const double *ax = auxs + ((long long)b * (T + 1) + t) * A::SIZE;
  const double *Kw = Kws + node * NV * NX;
  double *K = Kout + node * NV * NX;
  const double *tx = (j < NV) ? ax + A::tq + j : ax + A::tv + (j - NV);
  double kc[NV];
#pragma unroll
  for (int l = 0; l < NV; ++l) kc[l] = Kw[l * NX + j];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double acc = -tx[i * A::LD];
#pragma unroll
    for (int l = 0; l < NV; ++l) acc += ax[A::M + i * A::LD + l] * kc[l];
    K[i * NX + j] = acc;
  }
}

}  // namespace agx
