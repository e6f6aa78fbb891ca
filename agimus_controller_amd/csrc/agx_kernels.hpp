// agx_kernels.hpp -- gfx950 kernels of the SQP solve path.
//
//   K1  k_calc_diff / k_calc_diff_term   node-parallel calc + calcDiff (one lane per node)
//   K2  k_direction                      Riccati backward + linear forward + KKT (one wave per instance)
//   K4  k_linesearch                     merit line search, calc only (one workgroup per instance)
//   plus warm-start shift, reference generators and small batch utilities.
//
// Replaces mim_solvers::SolverCSQP::solve as called at
// agimus_controller/agimus_controller/ocp_base_croco.py:172 (unconstrained branch).
#pragma once

#include "agx_device.hpp"

// Per-instance solver state, device resident.
struct DevState {
  double kkt, cost, merit, gap;  // as agx_status
  double preg, dreg;             // crocoddyl regularisation (reg_min 1e-9)
  int iter, qp_iters, solved, flags;
  int done;                      // 1: instance finished (solved, or regularisation saturated)
  int pad;
};

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
template <int NV, bool CHAIN>
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
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) a += Minv[i][j] * (u[j] - nle[j]);
    qdd[i] = a;
  }
  // gap f = xnext - xs[t+1]
  {
    const double *xn = xp + NX;
#pragma unroll
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
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        double aq = 0.0, av = 0.0;
#pragma unroll
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
  tile[TO::cost] = dt * c.cost;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    tile[TO::Lx + i] = dt * c.Lq[i];
    tile[TO::Lx + NV + i] = dt * c.Lv[i];
    tile[TO::Lu + i] = dt * c.Lu[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      tile[TO::Lxx + i * NX + j] = dt * c.Lqq[i][j];
      tile[TO::Lxx + i * NX + NV + j] = 0.0;
      tile[TO::Lxx + (NV + i) * NX + j] = 0.0;
      tile[TO::Lxx + (NV + i) * NX + NV + j] = (i == j) ? dt * c.Lvv[i] : 0.0;
      tile[TO::Luu + i * NU + j] = (i == j) ? dt * c.Luu[i] : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < NX * NU; ++i) tile[TO::Lxu + i] = 0.0;
}

// terminal nodes: cost only, xnext = x (dt = 0, cost not scaled)
template <int NV, bool CHAIN>
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
  tile[TO::cost] = c.cost;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
#pragma unroll
    for (int j = 0; j < NX; ++j) tile[TO::Fx + i * NX + j] = (i == j) ? 1.0 : 0.0;
    tile[TO::f + i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < NX * NU; ++i) { tile[TO::Fu + i] = 0.0; tile[TO::Lxu + i] = 0.0; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    tile[TO::Lx + i] = c.Lq[i];
    tile[TO::Lx + NV + i] = c.Lv[i];
    tile[TO::Lu + i] = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      tile[TO::Lxx + i * NX + j] = c.Lqq[i][j];
      tile[TO::Lxx + i * NX + NV + j] = 0.0;
      tile[TO::Lxx + (NV + i) * NX + j] = 0.0;
      tile[TO::Lxx + (NV + i) * NX + NV + j] = (i == j) ? c.Lvv[i] : 0.0;
      tile[TO::Luu + i * NU + j] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------
// K2: QP direction for one instance per wave.
//   plain pass  (preg, dreg):  K, k -> forward dx, du, KKT          [every iteration]
//   sigma pass  (ADMM form):   K_out, the gains the solver reports   [on exit only]
// LDS holds V, [Fx|Fu], V[Fx|Fu], Q and the gains of the current node.
// ---------------------------------------------------------------------------
template <int NV>
struct DirLds {
  static constexpr int NX = 2 * NV, NU = NV, NXU = 3 * NV;
  double A[NX * NXU];
  double W[NX * NXU];
  double Q[NXU * NXU];
  double V[NX * NX];
  double Kl[NU * NX];
  double q[NXU];
  double Vx[NX], Vp[NX], f[NX], kl[NU];
  double dx[NX], du[NU], dxn[NX];
  double H[NXU * NXU];  // forward pass: [Lxx Lxu; Lxu^T Luu]
};

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

// one backward sweep; SIGMA selects the proximal form.  Gains go to Kdst/kdst.
template <int NV, bool SIGMA>
__device__ void riccati_backward(DirLds<NV> &s, const double *__restrict__ tiles_b, int T, double preg, double dreg,
                                 const double *__restrict__ cx, const double *__restrict__ cu,
                                 double *__restrict__ Kdst, double *__restrict__ kdst, double *cost_sum, double *gap_sum) {
  constexpr int NX = 2 * NV, NU = NV, NXU = 3 * NV;
  typedef TileOff<NV> TO;
  const int lane = threadIdx.x;
  const double sig = SIGMA ? kSigma : 0.0;
  // terminal value function
  {
    const double *tt = tiles_b + (long long)T * TO::SIZE;
    for (int e = lane; e < NX * NX; e += 64) s.V[e] = tt[TO::Lxx + e] + ((e / NX == e % NX) ? (sig + dreg) : 0.0);
    if (lane < NX) s.Vx[lane] = tt[TO::Lx + lane] - (SIGMA ? sig * cx[(long long)T * NX + lane] : 0.0);
    if (cost_sum && lane == 0) *cost_sum += tt[TO::cost];
  }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const double *tl = tiles_b + (long long)t * TO::SIZE;
    for (int e = lane; e < NX * NX; e += 64) s.A[(e / NX) * NXU + (e % NX)] = tl[TO::Fx + e];
    for (int e = lane; e < NX * NU; e += 64) s.A[(e / NU) * NXU + NX + (e % NU)] = tl[TO::Fu + e];
    if (lane < NX) {
      const double fv = tl[TO::f + lane];
      s.f[lane] = fv;
      if (gap_sum) *gap_sum += fabs(fv);
    }
    if (cost_sum && lane == 0) *cost_sum += tl[TO::cost];
    __syncthreads();
    // Vp = Vx + V f ;  W = V A
    if (lane < NX) {
      double acc = s.Vx[lane];
#pragma unroll
      for (int j = 0; j < NX; ++j) acc += s.V[lane * NX + j] * s.f[j];
      s.Vp[lane] = acc;
    }
    for (int e = lane; e < NX * NXU; e += 64) {
      const int i = e / NXU, j = e % NXU;
      double acc = 0.0;
#pragma unroll
      for (int l = 0; l < NX; ++l) acc += s.V[i * NX + l] * s.A[l * NXU + j];
      s.W[e] = acc;
    }
    __syncthreads();
    // Q = L + A^T W (upper triangle, mirrored) ; q = l + A^T Vp
    for (int e = lane; e < NXU * NXU; e += 64) {
      const int a = e / NXU, bq = e % NXU;
      if (bq < a) continue;
      double acc;
      if (bq < NX) acc = tl[TO::Lxx + a * NX + bq];
      else if (a < NX) acc = tl[TO::Lxu + a * NU + (bq - NX)];
      else acc = tl[TO::Luu + (a - NX) * NU + (bq - NX)];
#pragma unroll
      for (int l = 0; l < NX; ++l) acc += s.A[l * NXU + a] * s.W[l * NXU + bq];
      if (a == bq) acc += (a < NX) ? sig : (sig + preg);
      s.Q[a * NXU + bq] = acc;
      s.Q[bq * NXU + a] = acc;
    }
    if (lane < NXU) {
      double acc = (lane < NX) ? tl[TO::Lx + lane] : tl[TO::Lu + lane - NX];
#pragma unroll
      for (int l = 0; l < NX; ++l) acc += s.A[l * NXU + lane] * s.Vp[l];
      if (SIGMA) acc -= sig * ((lane < NX) ? cx[(long long)t * NX + lane] : cu[(long long)t * NU + lane - NX]);
      s.q[lane] = acc;
    }
    __syncthreads();
    // Cholesky of Quu in registers (every lane, broadcast LDS reads), then lane j
    // solves column j of Qux, lane NX solves Qu.
    {
      double L[NU][NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i][j] = s.Q[(NX + i) * NXU + NX + j];
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        double dd = L[j][j];
#pragma unroll
        for (int l = 0; l < j; ++l) dd -= L[j][l] * L[j][l];
        const double ll = sqrt(dd), il = 1.0 / ll;
        L[j][j] = il;  // store the reciprocal of the pivot
#pragma unroll
        for (int i = j + 1; i < NU; ++i) {
          double sacc = L[i][j];
#pragma unroll
          for (int l = 0; l < j; ++l) sacc -= L[i][l] * L[j][l];
          L[i][j] = sacc * il;
        }
      }
      if (lane <= NX) {
        double rhs[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) rhs[i] = (lane < NX) ? s.Q[(NX + i) * NXU + lane] : s.q[NX + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double sacc = rhs[i];
#pragma unroll
          for (int l = 0; l < i; ++l) sacc -= L[i][l] * rhs[l];
          rhs[i] = sacc * L[i][i];
        }
#pragma unroll
        for (int i = NU - 1; i >= 0; --i) {
          double sacc = rhs[i];
#pragma unroll
          for (int l = i + 1; l < NU; ++l) sacc -= L[l][i] * rhs[l];
          rhs[i] = sacc * L[i][i];
        }
        if (lane < NX) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            s.Kl[i * NX + lane] = rhs[i];
            Kdst[(long long)t * NU * NX + i * NX + lane] = rhs[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            s.kl[i] = rhs[i];
            if (kdst) kdst[(long long)t * NU + i] = rhs[i];
          }
        }
      }
    }
    __syncthreads();
    // V = sym(Qxx - Qxu K) + dreg ; Vx = Qx - K^T Qu
    for (int e = lane; e < NX * NX; e += 64) {
      const int a = e / NX, bq = e % NX;
      if (bq < a) continue;
      double v1 = s.Q[a * NXU + bq], v2 = s.Q[bq * NXU + a];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        v1 -= s.Q[a * NXU + NX + i] * s.Kl[i * NX + bq];
        v2 -= s.Q[bq * NXU + NX + i] * s.Kl[i * NX + a];
      }
      const double v = 0.5 * (v1 + v2) + (a == bq ? dreg : 0.0);
      s.V[a * NX + bq] = v;
      s.V[bq * NX + a] = v;
    }
    if (lane < NX) {
      double acc = s.q[lane];
#pragma unroll
      for (int i = 0; i < NU; ++i) acc -= s.Kl[i * NX + lane] * s.q[NX + i];
      s.Vx[lane] = acc;
    }
    __syncthreads();
  }
}

// mode bit0: plain pass + forward + KKT;  bit1: force the sigma pass (last iteration / timeout)
template <int NV>
__global__ void __launch_bounds__(64) k_direction(const DevOcp *__restrict__ op, const double *__restrict__ tiles,
                                                  double *__restrict__ Kws, double *__restrict__ kws,
                                                  double *__restrict__ Kout, double *__restrict__ dxs,
                                                  double *__restrict__ dus, DevState *__restrict__ st, int iter, int mode,
                                                  int *__restrict__ n_done) {
  constexpr int NX = 2 * NV, NU = NV, NXU = 3 * NV;
  typedef TileOff<NV> TO;
  __shared__ DirLds<NV> s;
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, lane = threadIdx.x;
  DevState &S = st[b];
  if (S.done) return;
  const double *tiles_b = tiles + (long long)b * (T + 1) * TO::SIZE;
  double *Kw = Kws + (long long)b * T * NU * NX, *kw = kws + (long long)b * T * NU;
  double *dx = dxs + (long long)b * (T + 1) * NX, *du = dus + (long long)b * T * NU;
  const double preg = S.preg, dreg = S.dreg;
  bool converged = false;
  if (mode & 1) {
    double cost = 0.0, gap = 0.0;
    riccati_backward<NV, false>(s, tiles_b, T, preg, dreg, nullptr, nullptr, Kw, kw, &cost, &gap);
    gap = wave_sum(gap);
    cost = __shfl(cost, 0, 64);
    // forward pass, dx_0 = 0
    if (lane < NX) { s.dx[lane] = 0.0; dx[lane] = 0.0; }
    double kkt = 0.0;
    for (int t = 0; t < T; ++t) {
      const double *tl = tiles_b + (long long)t * TO::SIZE;
      for (int e = lane; e < NX * NX; e += 64) {
        s.A[(e / NX) * NXU + (e % NX)] = tl[TO::Fx + e];
        s.H[(e / NX) * NXU + (e % NX)] = tl[TO::Lxx + e];
      }
      for (int e = lane; e < NX * NU; e += 64) {
        s.A[(e / NU) * NXU + NX + (e % NU)] = tl[TO::Fu + e];
        const double l = tl[TO::Lxu + e];
        s.H[(e / NU) * NXU + NX + (e % NU)] = l;
        s.H[(NX + (e % NU)) * NXU + (e / NU)] = l;
        s.Kl[e] = Kw[(long long)t * NU * NX + e];
      }
      for (int e = lane; e < NU * NU; e += 64) s.H[(NX + e / NU) * NXU + NX + (e % NU)] = tl[TO::Luu + e];
      if (lane < NX) {
        const double fv = tl[TO::f + lane];
        s.f[lane] = fv;
        kkt = fmax(kkt, fabs(fv));
      }
      if (lane < NU) s.kl[lane] = kw[(long long)t * NU + lane];
      __syncthreads();
      if (lane < NU) {
        double acc = -s.kl[lane];
#pragma unroll
        for (int j = 0; j < NX; ++j) acc -= s.Kl[lane * NX + j] * s.dx[j];
        s.du[lane] = acc;
        du[(long long)t * NU + lane] = acc;
      }
      __syncthreads();
      if (lane < NX) {
        double acc = s.f[lane];
#pragma unroll
        for (int j = 0; j < NX; ++j) acc += s.A[lane * NXU + j] * s.dx[j];
#pragma unroll
        for (int j = 0; j < NU; ++j) acc += s.A[lane * NXU + NX + j] * s.du[j];
        s.dxn[lane] = acc;
        dx[(long long)(t + 1) * NX + lane] = acc;
      }
      // stationarity through the QP optimality identity (incl. the regularisation terms):
      //   Lx + Fx' lam' - lam = -(Lxx dx + Lxu du + dreg dx),  Lu + Fu' lam' = -(Lxu' dx + (Luu + preg) du)
      if (lane < NXU) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) acc += s.H[lane * NXU + j] * s.dx[j];
#pragma unroll
        for (int j = 0; j < NU; ++j) acc += s.H[lane * NXU + NX + j] * s.du[j];
        if (lane < NX) {
          acc += dreg * s.dx[lane];
          if (t > 0) kkt = fmax(kkt, fabs(acc));
        } else {
          acc += preg * s.du[lane - NX];
          kkt = fmax(kkt, fabs(acc));
        }
      }
      __syncthreads();
      if (lane < NX) s.dx[lane] = s.dxn[lane];
      __syncthreads();
    }
    {
      const double *tt = tiles_b + (long long)T * TO::SIZE;
      if (lane < NX) {
        double acc = dreg * s.dx[lane];
#pragma unroll
        for (int j = 0; j < NX; ++j) acc += tt[TO::Lxx + lane * NX + j] * s.dx[j];
        kkt = fmax(kkt, fabs(acc));
      }
    }
    kkt = wave_max(kkt);
    converged = (kkt <= o.tol);
    if (lane == 0) {
      S.kkt = kkt;
      S.cost = cost;
      S.gap = gap;
      S.merit = cost + o.mu_dyn * gap;
      S.qp_iters = 1;
      if (!(kkt == kkt)) S.flags |= 1;
    }
  }
  if (converged || (mode & 2)) {
    __syncthreads();
    riccati_backward<NV, true>(s, tiles_b, T, preg, dreg, dx, du, Kout + (long long)b * T * NU * NX, nullptr, nullptr, nullptr);
  }
  if (converged && lane == 0) {
    S.solved = 1;
    S.done = 1;
    S.iter = iter;
    atomicAdd(n_done, 1);
  }
}

// ---------------------------------------------------------------------------
// K4: merit line search, one workgroup per instance, lanes over
// nodes.  alpha = 2^-n, n = 0..9, accept the first merit_try < merit.
// ---------------------------------------------------------------------------
template <int NV, bool CHAIN>
__global__ void __launch_bounds__(128) k_linesearch(const DevModel *__restrict__ mp, const DevOcp *__restrict__ op,
                                                    const double *__restrict__ dts, double *__restrict__ xs,
                                                    double *__restrict__ us, RefView rv, const double *__restrict__ dxs,
                                                    const double *__restrict__ dus, DevState *__restrict__ st, int iter,
                                                    int max_iter, int *__restrict__ n_done) {
  constexpr int NX = 2 * NV, NU = NV;
  __shared__ double red[4];
  __shared__ int accept;
  const DevModel &m = *mp;
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x;
  DevState &S = st[b];
  if (S.done) return;
  double *X = xs + (long long)b * (T + 1) * NX, *U = us + (long long)b * T * NU;
  const double *DX = dxs + (long long)b * (T + 1) * NX, *DU = dus + (long long)b * T * NU;
  const double merit = S.merit;
  double alpha = 1.0, used = 1.0;
  bool ok = false;
  // this kernel supports T + 1 <= blockDim * NPT nodes
  constexpr int NPT = 4;
  for (int n = 0; n < 10; ++n, alpha *= 0.5) {
    used = alpha;
    double part = 0.0;
    for (int r = 0; r < NPT; ++r) {
      const int t = tid + r * blockDim.x;
      if (t > T) break;
      double x[NX], u[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = X[(long long)t * NX + i] + alpha * DX[(long long)t * NX + i];
      if (t < T) {
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = U[(long long)t * NU + i] + alpha * DU[(long long)t * NU + i];
        double xn[NX], c;
        node_calc_running<NV, CHAIN>(m, o.rows[0], dts[t], x, u, ref_at(rv, b, t, T), frames_at(rv, b, t, T), xn, &c);
        double g = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i)
          g += fabs(xn[i] - (X[(long long)(t + 1) * NX + i] + alpha * DX[(long long)(t + 1) * NX + i]));
        part += c + o.mu_dyn * g;
      } else {
        double c;
        node_calc_terminal<NV, CHAIN>(m, o.rows[1], x, ref_at(rv, b, T, T), frames_at(rv, b, T, T), &c);
        part += c;
      }
    }
    part = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
      accept = (merit > tot) ? 1 : 0;
    }
    __syncthreads();
    ok = accept != 0;
    if (ok) break;
    __syncthreads();
  }
  if (ok) {
    for (int r = 0; r < NPT; ++r) {
      const int t = tid + r * blockDim.x;
      if (t > T) break;
#pragma unroll
      for (int i = 0; i < NX; ++i) X[(long long)t * NX + i] += used * DX[(long long)t * NX + i];
      if (t < T) {
#pragma unroll
        for (int i = 0; i < NU; ++i) U[(long long)t * NU + i] += used * DU[(long long)t * NU + i];
      }
    }
  }
  if (tid == 0) {
    if (!ok) S.flags |= 2;
    double preg = S.preg, dreg = S.dreg;
    // crocoddyl/mim_solvers regularisation schedule (th_stepdec 0.5, th_stepinc 0.01, factor 10)
    if (used > 0.5) { preg = fmax(preg / 10.0, kRegMin); dreg = fmax(dreg / 10.0, kRegMin); }
    bool stop = false;
    if (used <= 0.01) {
      preg = fmin(preg * 10.0, kRegMax);
      dreg = fmin(dreg * 10.0, kRegMax);
      if (preg == kRegMax) stop = true;
    }
    S.preg = preg;
    S.dreg = dreg;
    if (stop) {
      S.done = 1;
      S.iter = iter + 1;
      atomicAdd(n_done, 1);
    } else if (iter + 1 == max_iter) {
      S.iter = max_iter;
    }
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
  s.kkt = 0.0; s.cost = 0.0; s.merit = 0.0; s.gap = 0.0;
  s.preg = kRegMin; s.dreg = kRegMin;
  s.iter = 0; s.qp_iters = 0; s.solved = 0; s.flags = 0; s.done = 0; s.pad = 0;
  st[b] = s;
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
#pragma unroll
      for (int e = 0; e < NX; ++e) xo[e] = X[(long long)(i + 1) * NX + e];
      const int iu = (i < T - 1) ? i + 1 : i;
#pragma unroll
      for (int e = 0; e < NU; ++e) uo[e] = U[(long long)iu * NU + e];
    } else {
      double x[NX], u[NU], c;
#pragma unroll
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
#pragma unroll
    for (int e = 0; e < NX; ++e) Xo[((long long)b * (T + 1) + i) * NX + e] = xo[e];
#pragma unroll
    for (int e = 0; e < NU; ++e) Uo[((long long)b * T + i) * NU + e] = uo[e];
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
#pragma unroll
  for (int e = 0; e < NX; ++e) xl[e] = x[(long long)i * NX + e];
#pragma unroll
  for (int e = 0; e < NU; ++e) ul[e] = u[(long long)i * NU + e];
  DevRows none;
  none.n = 0;
  node_calc_running<NV, CHAIN>(*mp, none, dt, xl, ul, nullptr, nullptr, xn, &c);
#pragma unroll
  for (int e = 0; e < NX; ++e) xnext[(long long)i * NX + e] = xn[e];
}

template <int NV, bool CHAIN>
__global__ void k_rnea(const DevModel *__restrict__ mp, int n, const double *__restrict__ q, const double *__restrict__ v,
                       const double *__restrict__ a, double *__restrict__ tau) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double ql[NV], vl[NV], al[NV], tl[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) { ql[e] = q[(long long)i * NV + e]; vl[e] = v[(long long)i * NV + e]; al[e] = a[(long long)i * NV + e]; }
  Kin<NV> k;
  kinematics<NV, CHAIN>(*mp, ql, k);
  rnea<NV, CHAIN>(*mp, k, vl, al, tl);
#pragma unroll
  for (int e = 0; e < NV; ++e) tau[(long long)i * NV + e] = tl[e];
}

template <int NV, bool CHAIN>
__global__ void k_frame(const DevModel *__restrict__ mp, int n, int frame, const double *__restrict__ q, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double ql[NV];
#pragma unroll
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
#pragma unroll
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
  double *pt = pts + unit * (4 * NV + 12);
#pragma unroll
  for (int i = 0; i < NV; ++i) { pt[i] = q[i]; pt[NV + i] = dq[i]; pt[2 * NV + i] = ddq[i]; pt[3 * NV + i] = u[i]; }
#pragma unroll
  for (int e = 0; e < 9; ++e) pt[4 * NV + e] = RF[e];
#pragma unroll
  for (int e = 0; e < 3; ++e) pt[4 * NV + 9 + e] = pF[e];
  for (int layout = 0; layout < 2; ++layout) {
    const DevRows &rows = o.rows[layout];
    double *tile = traj + unit * 2 * o.stride + layout * o.stride;
    for (int r = 0; r < rows.n; ++r) {
      double *tr = tile + rows.off[r];
      tr[0] = 1.0;
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

// WarmStartReference.generate on the device (warm_start_reference.py:33-96):
// xs[t] = ref state of sample k0+t (xs[0] = x0), us[t] = ref effort of sample k0+t
// (us[0] = RNEA at the measured state would need its acceleration: the reference
// passes initial_state.robot_acceleration; the resident trajectory uses sample k0's).
__global__ void k_ws_from_ref(double *xs, double *us, double *x0, const double *pts, int B, int T, int NV, int n_points, int k0, int set_x0) {
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (unit >= (long long)B * (T + 1)) return;
  const int b = (int)(unit / (T + 1)), t = (int)(unit % (T + 1));
  const double *pt = pts + ((long long)b * n_points + k0 + t) * (4 * NV + 12);
  const int NX = 2 * NV;
  for (int i = 0; i < NX; ++i) xs[unit * NX + i] = pt[i];
  if (t < T)
    for (int i = 0; i < NV; ++i) us[((long long)b * T + t) * NV + i] = pt[3 * NV + i];
  if (t == 0 && set_x0)
    for (int i = 0; i < NX; ++i) x0[(long long)b * NX + i] = pt[i];
}

}  // namespace agx
