// agimus_controller_amd -- large models (nv > 7, e.g. the 30-DoF humanoid of BASELINE configs[4]).
//
// The register-resident kernels map an nv x nv block onto an 8 x 8 lane grid; beyond that the same
// algebra runs out of LDS: one 256-thread workgroup per instance keeps the 3nv x (3nv + 1) elimination
// matrix (w | q | v blocks + gradient column) and the value function of node t+1 in LDS
// (nv = 30: 66 KB + 29 KB) and eliminates the nv acceleration variables with nv Gauss-Jordan pivots,
// two barriers each.  Same tiles (row stride 32), same acceleration-input QP, same outputs as the
// nv <= 7 path.  The derivative pass and the line-search trials of these sizes are the
// workgroup-per-node kernels of agx_big_k1.hpp (LDS + fp64 MFMA, no scratch).
//
// (included at the end of agx_kernels.hpp)
#pragma once

namespace agx {

// K2 for large nv.  gains_pass: backward sweep only, every instance, on the sigma-augmented tiles
// (k_sigma_tile_big), gradient ignored.
template <int NV>
__global__ void __launch_bounds__(256) k_riccati_big(const DevOcp *__restrict__ op, const double *__restrict__ dts,
                                                     const double *__restrict__ qts, double *__restrict__ Kws,
                                                     double *__restrict__ kws, double *__restrict__ dxs,
                                                     double *__restrict__ wss, DevState *__restrict__ st, int forward,
                                                     int gains_pass) {
  constexpr int NX = 2 * NV, R = 3 * NV, GC = 3 * NV, CS = 3 * NV + 2;
  typedef QT<NV> Q;
  __shared__ double Mx[R * CS];
  // the value Hessian of node t+1 lives in the x-x block of Mx (rows / columns NV..3NV): the build below
  // reads exactly the four entries it overwrites, so no second 2nv x 2nv array is needed (69 KB of LDS
  // in total at nv = 30: two workgroups per CU)
#define AGX_VV(i, j) Mx[(NV + (i)) * CS + NV + (j)]
  __shared__ double vx[NX], vp[NX], fl[NX], rpv[NV], dxl[NX], wl[NV];
  __shared__ double rowbuf[2][3 * NV + 8], colbuf[2][3 * NV + 8];
  const DevOcp &o = *op;
  const int T = o.T, b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
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
  // value function of the terminal node
  {
    const double *tt = qb + (long long)T * Q::SIZE;
    for (int e = tid; e < NX * NX; e += nt) {
      const int i = e / NX, j = e % NX;
      const int ib = i < NV ? i : i - NV, jb = j < NV ? j : j - NV;
      double v;
      if (i < NV && j < NV) v = tt[Q::Hqq + ib * Q::LD + jb];
      else if (i < NV) v = tt[Q::Hqv + ib * Q::LD + jb];
      else if (j < NV) v = tt[Q::Hqv + jb * Q::LD + ib];
      else v = tt[Q::Hvv + ib * Q::LD + jb];
      AGX_VV(i, j) = v + ((i == j) ? dreg : 0.0);
    }
    for (int i = tid; i < NX; i += nt) vx[i] = gains_pass ? 0.0 : tt[Q::gx + i];
  }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const double *tl = qb + (long long)t * Q::SIZE;
    const double h = dts[t], h2 = h * h;
    // vp = vx + V f
    for (int i = tid; i < NX; i += nt) fl[i] = gains_pass ? 0.0 : tl[Q::f + i];
    __syncthreads();
    for (int i = tid; i < NX; i += nt) {
      double s = vx[i];
      for (int j = 0; j < NX; ++j) s += AGX_VV(i, j) * fl[j];
      vp[i] = s;
    }
    __syncthreads();
    // elimination matrix: rows / columns  w (0..NV) | q | v, gradient in column GC
    for (int e = tid; e < NV * NV; e += nt) {
      const int r = e / NV, c = e % NV;
      const int rc = r * Q::LD + c, cr = c * Q::LD + r;
      const double Vqq = AGX_VV(r, c), Vqv = AGX_VV(r, NV + c), Vvq = AGX_VV(NV + r, c), Vvv = AGX_VV(NV + r, NV + c);
      const double Yq = h2 * Vqq + h * Vvq, Yv = h2 * Vqv + h * Vvv;
      const double YqT = h2 * Vqq + h * Vqv, YvT = h2 * Vvq + h * Vvv;
      Mx[r * CS + c] = tl[Q::Hww + rc] + h2 * Yq + h * Yv;
      Mx[r * CS + NV + c] = tl[Q::Hqw + cr] + Yq;
      Mx[r * CS + 2 * NV + c] = tl[Q::Hvw + cr] + h * Yq + Yv;
      Mx[(NV + r) * CS + c] = tl[Q::Hqw + rc] + YqT;
      Mx[(2 * NV + r) * CS + c] = tl[Q::Hvw + rc] + h * YqT + YvT;
      Mx[(NV + r) * CS + NV + c] = tl[Q::Hqq + rc] + Vqq;
      Mx[(NV + r) * CS + 2 * NV + c] = tl[Q::Hqv + rc] + h * Vqq + Vqv;
      Mx[(2 * NV + r) * CS + NV + c] = tl[Q::Hqv + cr] + h * Vqq + Vvq;
      Mx[(2 * NV + r) * CS + 2 * NV + c] = tl[Q::Hvv + rc] + h2 * Vqq + h * (Vqv + Vvq) + Vvv;
    }
    for (int r = tid; r < NV; r += nt) {
      const double vpq = vp[r], vpv = vp[NV + r];
      Mx[r * CS + GC] = gains_pass ? 0.0 : tl[Q::gw + r] + h2 * vpq + h * vpv;
      Mx[(NV + r) * CS + GC] = gains_pass ? 0.0 : tl[Q::gx + r] + vpq;
      Mx[(2 * NV + r) * CS + GC] = gains_pass ? 0.0 : tl[Q::gx + NV + r] + h * vpq + vpv;
    }
    __syncthreads();
    // Gauss-Jordan pivots on the w block, register blocked: thread (br, bc) keeps the RB x CB block
    // rows RB*br.., columns CB*bc.. of the elimination matrix in registers for all nv pivots; per pivot
    // the owners of row k / column k publish them through (double-buffered) LDS vectors, everybody
    // reads RB + CB values and does RB*CB FMAs.  The pivot row itself stays as it is (factor 0).
    {
      constexpr int NBC = 16, CB = (GC + NBC) / NBC, NBR = 256 / NBC - 1, RB = (R + NBR - 1) / NBR;
      static_assert(CB * NBC > GC && RB * NBR >= R, "block grid must cover the elimination matrix");
      const int br = tid / NBC, bc = tid % NBC;
      const bool owner = br < NBR;
      double m[RB][CB];
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          const int r = RB * br + i, c = CB * bc + j;
          m[i][j] = (owner && r < R && c <= GC) ? Mx[r * CS + c] : 0.0;
        }
      for (int k = 0; k < NV; ++k) {
        double *rowk = rowbuf[k & 1], *colk = colbuf[k & 1];
        if (owner && k / RB == br) {
#pragma unroll
          for (int i = 0; i < RB; ++i)
            if (i == k % RB) {
#pragma unroll
              for (int j = 0; j < CB; ++j) rowk[CB * bc + j] = m[i][j];
            }
        }
        if (owner && k / CB == bc) {
#pragma unroll
          for (int j = 0; j < CB; ++j)
            if (j == k % CB) {
#pragma unroll
              for (int i = 0; i < RB; ++i) colk[RB * br + i] = m[i][j];
            }
        }
        __syncthreads();
        const double rp = 1.0 / rowk[k];
        if (tid == 0) rpv[k] = rp;
        double fi[RB], rj[CB];
#pragma unroll
        for (int i = 0; i < RB; ++i) fi[i] = (RB * br + i == k) ? 0.0 : colk[RB * br + i] * rp;
#pragma unroll
        for (int j = 0; j < CB; ++j) rj[j] = rowk[CB * bc + j];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) m[i][j] -= fi[i] * rj[j];
      }
      __syncthreads();
      // back to LDS: what the rest of the step reads (x columns and the gradient column of every row)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          const int r = RB * br + i, c = CB * bc + j;
          if (owner && r < R && c >= NV && c <= GC) Mx[r * CS + c] = m[i][j];
        }
      __syncthreads();
    }
    // gains of this node and the value function of node t
    for (int e = tid; e < NV * NX; e += nt) {
      const int r = e / NX, c = e % NX;
      Kw[(long long)t * NV * NX + e] = Mx[r * CS + NV + c] * rpv[r];
    }
    for (int r = tid; r < NV; r += nt) kw[(long long)t * NV + r] = Mx[r * CS + GC] * rpv[r];
    for (int e = tid; e < NX * NX; e += nt) {  // symmetrise in place: one thread per unordered pair
      const int i = e / NX, j = e % NX;
      if (i > j) continue;
      const double sv = 0.5 * (AGX_VV(i, j) + AGX_VV(j, i)) + ((i == j) ? dreg : 0.0);
      AGX_VV(i, j) = sv;
      AGX_VV(j, i) = sv;
    }
    for (int i = tid; i < NX; i += nt) vx[i] = Mx[(NV + i) * CS + GC];
    __syncthreads();
  }
  if (gains_pass || !forward) return;
  // forward pass
  double *dx = dxs + (long long)b * (T + 1) * NX, *ws = wss + (long long)b * T * NV;
  for (int i = tid; i < NX; i += nt) { dxl[i] = 0.0; dx[i] = 0.0; }
  __threadfence_block();
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const double *tl = qb + (long long)t * Q::SIZE;
    const double h = dts[t], h2 = h * h;
    for (int r = tid; r < NV; r += nt) {
      double s = -kw[(long long)t * NV + r];
      const double *kr = Kw + ((long long)t * NV + r) * NX;
      for (int c = 0; c < NX; ++c) s -= kr[c] * dxl[c];
      wl[r] = s;
      ws[(long long)t * NV + r] = s;
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

#undef AGX_VV

// K3 for large nv (see k_node_kkt for the identities): 32 lanes per node, lane l = column l of every matrix row (one
// coalesced 256-byte row load per matrix and row), the row sums  du_i = sum_l M[i][l] w_l + tq[i][l] dq_l + tv[i][l] dv_l
// and  (Lqq dq)_i  as 32-lane butterflies on the VALU (DPP inside the 16-lane rows, v_permlane16_swap across them).
// One lane per node with serial loops (round 1) read its 31 KB at 3 TB/s with 0.4 waves per SIMD.
__device__ __forceinline__ double sum32(double p) {
  p += dpp_xor1(p); p += dpp_xor2(p); p += dpp_xor4(p);
  p += dpp_mov<0x128>(p);  // row_ror:8
  const unsigned lo = __double2loint(p), hi = __double2hiint(p);
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(c.x, a.x) + __hiloint2double(c.y, a.y);
}
__device__ __forceinline__ double max32(double p) {
  p = fmax(p, dpp_xor1(p)); p = fmax(p, dpp_xor2(p)); p = fmax(p, dpp_xor4(p));
  p = fmax(p, dpp_mov<0x128>(p));
  const unsigned lo = __double2loint(p), hi = __double2hiint(p);
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return fmax(__hiloint2double(c.x, a.x), __hiloint2double(c.y, a.y));
}
template <int NV>
__global__ void __launch_bounds__(64) k_node_kkt_big(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                     const double *__restrict__ auxs, const double *__restrict__ dxs,
                                                     const double *__restrict__ wss, double *__restrict__ dus,
                                                     double *__restrict__ nodestat, const DevState *__restrict__ st) {
  static_assert(NV <= 32, "a matrix row per 32 lanes");
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T, l = threadIdx.x & 31;
  const long long n_nodes = (long long)o.B * (T + 1);
  const long long node_raw = (long long)blockIdx.x * 2 + (threadIdx.x >> 5);
  const bool ok = node_raw < n_nodes;
  const long long node = ok ? node_raw : n_nodes - 1;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  const DevState &S = st[b];
  const bool act = ok && !S.done;
  if (!__any(act)) return;  // both nodes of the wave finished
  const double preg = S.preg, dreg = S.dreg;
  const double *qt = qts + node * Q::SIZE;
  const double *ax = auxs + node * A::SIZE;
  const double *dx = dxs + node * NX;
  const bool in = l < NV;
  const int lc = in ? l : 0;
  const double m = in ? 1.0 : 0.0;
  const double dq = dx[lc] * m, dv = dx[NV + lc] * m;
  double kkt = 0.0, gap = 0.0;
  if (t < T) {  // uniform over the node's 32 lanes
    const double fq = qt[Q::f + lc] * m, fv = qt[Q::f + NV + lc] * m;
    kkt = fmax(fabs(fq), fabs(fv));
    gap = fabs(fq) + fabs(fv);
  }
  // the butterflies run on whole waves: both nodes walk the rows together, a terminal node with zero operands
  const double w = (t < T) ? wss[((long long)b * T + t) * NV + lc] * m : 0.0;
  const double tm = (t < T) ? 1.0 : 0.0, sm = (t > 0) ? 1.0 : 0.0;
  double du = 0.0, hq = 0.0;
#pragma unroll 2
  for (int i = 0; i < NV; ++i) {
    const double p = (ax[A::M + i * A::LD + lc] * w + ax[A::tq + i * A::LD + lc] * dq + ax[A::tv + i * A::LD + lc] * dv) * tm;
    const double q = ax[A::Lqq + i * A::LD + lc] * dq;
    const double s = sum32(p), h = sum32(q);
    if (l == i) { du = s; hq = h; }
  }
  if (in) {
    if (t < T) {
      if (act) dus[((long long)b * T + t) * NV + l] = du;
      kkt = fmax(kkt, fabs((ax[A::Luu + l] + preg) * du));
    }
    kkt = fmax(kkt, sm * fmax(fabs(hq + dreg * dq), fabs((ax[A::Lvv + l] + dreg) * dv)));
  }
  kkt = max32(kkt);
  gap = sum32(gap);
  if (act && l == 0) {
    double *ns = nodestat + node * 4;
    ns[0] = kkt; ns[1] = qt[Q::cost]; ns[2] = gap;
    if (!o.has_con) ns[3] = 0.0;  // constrained problems: the violation share of the constraint evaluation (which may run before this kernel) stays
  }
}

// ---------------------------------------------------------------------------
// Constraints for large models (8 < nv <= 32) through the ADMM loop of agx_admm.hpp: ConstraintModelControlLimit
// (`ocp_croco_generic.py:624-640`: lower / upper bounds on u; identity Jacobian on u, i.e. the rows [taux | M] in the (dx, w)
// coordinates of the QP tiles) and collision-distance rows (`ocp_traj_tracking_collision_avoidance.yaml:48-56`; one Jacobian row
// on q each, k_con_eval_wg in agx_big_k1.hpp).  Correctness-first: every ADMM iteration factorises again (k_riccati_blk on the
// augmented tile; the stored-factor gradient sweeps of the 7-joint path have no counterpart here), the node kernels follow
// k_admm_tile / k_admm_update.  Other constraint kinds are refused for nv > 7.
// ---------------------------------------------------------------------------
// Augmented QP tile of one node (k_admm_tile for large models; instances whose rho changed, or at the first ADMM iteration):
//   H += [taux M]' diag(sigma + rho_u) [taux M] + sigma I_x,    g += [taux M]' (h_u - sigma du_c) - sigma dx_c,   h = y - rho z
// one 256-thread workgroup per node, M | tq | tv staged in LDS, thread (i, j) forms element [i][j] of the six blocks.
template <int NV>
__global__ void __launch_bounds__(256) k_admm_tile_big(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                       double *__restrict__ qt2s, const double *__restrict__ auxs,
                                                       const double *__restrict__ cxs, const double *__restrict__ dus,
                                                       const double *__restrict__ cjac, const double *__restrict__ ys,
                                                       const double *__restrict__ zs, const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  __shared__ double sM[NV * NV], sq[NV * NV], sv[NV * NV], wu[32], eu[32];
  __shared__ double sg[AGX_MAX_DENSE][32], srho[AGX_MAX_DENSE], sh[AGX_MAX_DENSE];  // rows on q: Jacobian row, rho, h = y - rho z
  __shared__ double sxr[2][32], sxh[2][32];  // state bounds: rho and h of the q | v component of every joint
  const DevOcp &o = *op;
  const int T = o.T, tid = threadIdx.x, nt = blockDim.x;
  const long long node = blockIdx.x;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  const DevState &S = st[b];
  if (S.done || S.admm_conv || !S.admm_refactor) return;  // uniform over the workgroup
  const double *qt = qts + node * Q::SIZE;
  double *q2 = qt2s + node * Q::SIZE;
  const double *ax = auxs + node * A::SIZE;
  const double *cx = cxs + node * NX;
  const DevCons &c = o.cons[t == T ? 1 : 0];
  const double sig = kSigma, rs = S.rho_sparse;
  if (t < T) {
    for (int e = tid; e < NV * NV; e += nt) {
      const int i = e / NV, j = e % NV;
      sM[e] = ax[A::M + i * A::LD + j]; sq[e] = ax[A::tq + i * A::LD + j]; sv[e] = ax[A::tv + i * A::LD + j];
    }
    if (tid < NV) {
      double w = sig, h = 0.0;
      for (int r = 0; r < c.n; ++r) {
        if (c.kind[r] != AGX_RES_CONTROL) continue;
        const int k = c.off[r] + tid;
        const double rho = admm_rho(c.lb[k], c.ub[k], rs);
        w += rho;
        h += ys[node * AGX_MAX_NC + k] - rho * zs[node * AGX_MAX_NC + k];
      }
      wu[tid] = w;
      eu[tid] = h - sig * dus[((long long)b * T + t) * NV + tid];
    }
  }
  if (tid < NV) {
    double rq = 0.0, rv = 0.0, hq = 0.0, hv = 0.0;
    for (int r = 0; r < c.n; ++r) {
      if (c.kind[r] != AGX_RES_STATE) continue;
      const int kq = c.off[r] + tid, kv = kq + NV;
      const double a = admm_rho(c.lb[kq], c.ub[kq], rs), bb = admm_rho(c.lb[kv], c.ub[kv], rs);
      rq += a; rv += bb;
      hq += ys[node * AGX_MAX_NC + kq] - a * zs[node * AGX_MAX_NC + kq];
      hv += ys[node * AGX_MAX_NC + kv] - bb * zs[node * AGX_MAX_NC + kv];
    }
    sxr[0][tid] = rq; sxr[1][tid] = rv; sxh[0][tid] = hq; sxh[1][tid] = hv;
  }
  int nd = 0;  // dense rows of this node type (uniform)
  for (int r = 0; r < c.n; ++r)
    if (c.kind[r] != AGX_RES_CONTROL && c.kind[r] != AGX_RES_STATE)  // rows with Jacobian rows on q: collision distance (1), frame translation / rotation (3), placement (6)
      for (int e = 0; e < c.nr[r]; ++e) {
        const int k = c.off[r] + e;
        if (tid < 32) sg[nd][tid] = tid < NV ? cjac[(node * AGX_MAX_DENSE + c.coll_slot[r] + e) * 32 + tid] : 0.0;
        if (tid == 32) {
          const double rho = admm_rho(c.lb[k], c.ub[k], rs);
          srho[nd] = rho;
          sh[nd] = ys[node * AGX_MAX_NC + k] - rho * zs[node * AGX_MAX_NC + k];
        }
        ++nd;
      }
  __syncthreads();
  for (int e = tid; e < NV * NV; e += nt) {
    const int i = e / NV, j = e % NV;
    double hww = 0.0, hqw = 0.0, hvw = 0.0, hqq = 0.0, hqv = 0.0, hvv = 0.0;
    if (t < T)
      for (int l = 0; l < NV; ++l) {
        const double d = wu[l];
        const double Mli = sM[l * NV + i], tqli = sq[l * NV + i], tvli = sv[l * NV + i];
        const double Mlj = d * sM[l * NV + j], tqlj = d * sq[l * NV + j], tvlj = d * sv[l * NV + j];
        hww += Mli * Mlj; hqw += tqli * Mlj; hvw += tvli * Mlj;
        hqq += tqli * tqlj; hqv += tqli * tvlj; hvv += tvli * tvlj;
      }
    for (int s = 0; s < nd; ++s) hqq += srho[s] * sg[s][i] * sg[s][j];  // rho g g' of the rows on q
    const double d = (i == j) ? sig : 0.0;
    const int o2 = i * Q::LD + j;
    q2[Q::Hqq + o2] = qt[Q::Hqq + o2] + hqq + d + ((i == j) ? sxr[0][i] : 0.0);
    q2[Q::Hqv + o2] = qt[Q::Hqv + o2] + hqv;
    q2[Q::Hvv + o2] = qt[Q::Hvv + o2] + hvv + d + ((i == j) ? sxr[1][i] : 0.0);
    if (t < T) {
      q2[Q::Hww + o2] = qt[Q::Hww + o2] + hww;
      q2[Q::Hqw + o2] = qt[Q::Hqw + o2] + hqw;
      q2[Q::Hvw + o2] = qt[Q::Hvw + o2] + hvw;
    }
  }
  if (tid < NV) {  // gradient, gap, cost
    const int i = tid;
    double gw = 0.0, gq = -sig * cx[i] + sxh[0][i], gv = -sig * cx[NV + i] + sxh[1][i];
    if (t < T)
      for (int l = 0; l < NV; ++l) { gw += sM[l * NV + i] * eu[l]; gq += sq[l * NV + i] * eu[l]; gv += sv[l * NV + i] * eu[l]; }
    for (int s = 0; s < nd; ++s) gq += sh[s] * sg[s][i];
    if (t < T) q2[Q::gw + i] = qt[Q::gw + i] + gw;
    q2[Q::gx + i] = qt[Q::gx + i] + gq;
    q2[Q::gx + NV + i] = qt[Q::gx + NV + i] + gv;
    q2[Q::f + i] = qt[Q::f + i];
    q2[Q::f + NV + i] = qt[Q::f + NV + i];
  }
  if (tid == 255) q2[Q::cost] = qt[Q::cost];
}

// After the sweep on the augmented tiles (k_admm_update for large models, control-limit rows): du, z / y update, the node's
// shares of the ADMM residual norms and of the KKT residual, the gradient of the next iteration's augmented tile.
// 32 lanes per node, lane l = component l / column l of every matrix row (as k_node_kkt_big).
template <int NV>
__global__ void __launch_bounds__(64) k_admm_update_big(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                        const double *__restrict__ auxs, const double *__restrict__ dxs,
                                                        const double *__restrict__ wss, double *__restrict__ dus,
                                                        double *__restrict__ cxs, const double *__restrict__ cg,
                                                        const double *__restrict__ cjac, double *__restrict__ ys, double *__restrict__ zs,
                                                        double *__restrict__ nodestat, double *__restrict__ admmstat,
                                                        double *__restrict__ qt2s, const DevState *__restrict__ st) {
  static_assert(NV <= 32, "a matrix row per 32 lanes");
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T, l = threadIdx.x & 31, half = threadIdx.x & 32;
  const long long n_nodes = (long long)o.B * (T + 1);
  const long long node_raw = (long long)blockIdx.x * 2 + (threadIdx.x >> 5);
  const bool ok = node_raw < n_nodes;
  const long long node = ok ? node_raw : n_nodes - 1;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  const DevState &S = st[b];
  const bool act = ok && !S.done && !S.admm_conv && !S.dir_fail;
  if (!__any(act)) return;
  const double preg = S.preg, dreg = S.dreg, rs = S.rho_sparse, sig = kSigma;
  const double *qt = qts + node * Q::SIZE;
  const double *ax = auxs + node * A::SIZE;
  const double *dx = dxs + node * NX;
  double *cx = cxs + node * NX;
  double *y = ys + node * AGX_MAX_NC, *z = zs + node * AGX_MAX_NC;
  const double *g = cg + node * AGX_MAX_NC;
  const DevCons &c = o.cons[t == T ? 1 : 0];
  const bool in = l < NV;
  const int lc = in ? l : 0;
  const double m = in ? 1.0 : 0.0;
  const double dq = dx[lc] * m, dv = dx[NV + lc] * m, cq = cx[lc] * m, cv = cx[NV + lc] * m;
  double kkt = 0.0, gap = 0.0;
  if (t < T) {
    const double fq = qt[Q::f + lc] * m, fv = qt[Q::f + NV + lc] * m;
    kkt = fmax(fabs(fq), fabs(fv));
    gap = fabs(fq) + fabs(fv);
  }
  const double w = (t < T) ? wss[((long long)b * T + t) * NV + lc] * m : 0.0;
  const double tm = (t < T) ? 1.0 : 0.0, sm = (t > 0) ? 1.0 : 0.0;
  double du = 0.0, hq = 0.0;
#pragma unroll 2
  for (int i = 0; i < NV; ++i) {
    const double p = (ax[A::M + i * A::LD + lc] * w + ax[A::tq + i * A::LD + lc] * dq + ax[A::tv + i * A::LD + lc] * dv) * tm;
    const double q = ax[A::Lqq + i * A::LD + lc] * dq;
    const double s = sum32(p), h = sum32(q);
    if (l == i) { du = s; hq = h; }
  }
  const double duc = (t < T && in) ? dus[((long long)b * T + t) * NV + l] : 0.0;
  // control-limit rows: component l on lane l, C d = du
  double primal = 0.0, primal_rel = 0.0, dual_u = 0.0, drel_u = 0.0, e_u = 0.0, hn_u = 0.0;
  if (in && t < T)
    for (int r = 0; r < c.n; ++r) {
      if (c.kind[r] != AGX_RES_CONTROL) continue;
      const int k = c.off[r] + l;
      const double rho = admm_rho(c.lb[k], c.ub[k], rs);
      const double z0 = z[k], y0 = y[k];
      const double zrel = kAlphaRelax * du + (1.0 - kAlphaRelax) * z0;
      double zn = zrel + y0 / rho;
      zn = fmin(fmax(zn, c.lb[k] - g[k]), c.ub[k] - g[k]);
      const double yn = y0 + rho * (zrel - zn);
      primal = fmax(primal, fabs(du - zn));
      primal_rel = fmax(primal_rel, fmax(fabs(du), fabs(zn)));
      dual_u += rho * (zn - z0);
      drel_u += yn;
      e_u += rho * du + (y0 - rho * z0) - yn;
      hn_u += yn - rho * zn;
      if (act) { z[k] = zn; y[k] = yn; }
    }
  // rows on q (collision distance, frame residuals): one Jacobian row per component, C d = g . dq (the same on the 32 lanes of the node)
  double dual_q = 0.0, drel_q = 0.0, e_q = 0.0, hn_q = 0.0;
  double dual_v = 0.0, drel_v = 0.0, e_v = 0.0, hn_v = 0.0;
  if (in)
    for (int r = 0; r < c.n; ++r) {  // state bounds: components q_l and v_l on lane l
      if (c.kind[r] != AGX_RES_STATE) continue;
#pragma unroll
      for (int hv = 0; hv < 2; ++hv) {
        const int k = c.off[r] + hv * NV + l;
        const double Cd = hv ? dv : dq;
        const double rho = admm_rho(c.lb[k], c.ub[k], rs);
        const double z0 = z[k], y0 = y[k];
        const double zrel = kAlphaRelax * Cd + (1.0 - kAlphaRelax) * z0;
        double zn = zrel + y0 / rho;
        zn = fmin(fmax(zn, c.lb[k] - g[k]), c.ub[k] - g[k]);
        const double yn = y0 + rho * (zrel - zn);
        primal = fmax(primal, fabs(Cd - zn));
        primal_rel = fmax(primal_rel, fmax(fabs(Cd), fabs(zn)));
        const double dz = rho * (zn - z0), de = rho * Cd + (y0 - rho * z0) - yn, hn = yn - rho * zn;
        if (hv) { dual_v += dz; drel_v += yn; e_v += de; hn_v += hn; }
        else { dual_q += dz; drel_q += yn; e_q += de; hn_q += hn; }
        if (act) { z[k] = zn; y[k] = yn; }
      }
    }
  for (int r = 0; r < c.n; ++r)
  for (int e = 0; e < ((c.kind[r] != AGX_RES_CONTROL && c.kind[r] != AGX_RES_STATE) ? c.nr[r] : 0); ++e) {
    const int k = c.off[r] + e;
    const double gq = in ? cjac[(node * AGX_MAX_DENSE + c.coll_slot[r] + e) * 32 + l] : 0.0;
    const double Cd = sum32(gq * dq);
    const double rho = admm_rho(c.lb[k], c.ub[k], rs);
    const double z0 = z[k], y0 = y[k];
    const double zrel = kAlphaRelax * Cd + (1.0 - kAlphaRelax) * z0;
    double zn = zrel + y0 / rho;
    zn = fmin(fmax(zn, c.lb[k] - g[k]), c.ub[k] - g[k]);
    const double yn = y0 + rho * (zrel - zn);
    primal = fmax(primal, fabs(Cd - zn));
    primal_rel = fmax(primal_rel, fmax(fabs(Cd), fabs(zn)));
    dual_q += gq * rho * (zn - z0);
    drel_q += gq * yn;
    e_q += gq * (rho * Cd + (y0 - rho * z0) - yn);
    hn_q += gq * (yn - rho * zn);
    __builtin_amdgcn_wave_barrier();  // every lane has read z0, y0
    if (act && l == 0) { z[k] = zn; y[k] = yn; }
  }
  if (in) {
    if (t < T) kkt = fmax(kkt, fabs((ax[A::Luu + l] + preg) * du + sig * (du - duc) + e_u));
    kkt = fmax(kkt, sm * fmax(fabs(hq + dreg * dq + sig * (dq - cq) + e_q), fabs((ax[A::Lvv + l] + dreg) * dv + sig * (dv - cv) + e_v)));
  }
  kkt = max32(kkt);
  gap = sum32(gap);
  primal = max32(primal);
  primal_rel = max32(primal_rel);
  const double dual = max32(fmax(fabs(dual_u), fmax(fabs(dual_q), fabs(dual_v)))), drel = max32(fmax(fabs(drel_u), fmax(fabs(drel_q), fabs(drel_v))));
  // gradient of the next iteration's augmented tile:  g = g0 + [taux M]' (h_u - sigma du) - sigma dx
  const double e_own = (t < T && in) ? hn_u - sig * du : 0.0;
  double gwn = 0.0, gqn = hn_q - sig * dq, gvn = hn_v - sig * dv;
  for (int k = 0; k < NV; ++k) {
    const double e = __shfl(e_own, half + k, 64);
    gwn += ax[A::M + k * A::LD + lc] * e * tm;
    gqn += ax[A::tq + k * A::LD + lc] * e * tm;
    gvn += ax[A::tv + k * A::LD + lc] * e * tm;
  }
  if (act) {
    if (in) {
      double *q2 = qt2s + node * Q::SIZE;
      if (t < T) q2[Q::gw + l] = qt[Q::gw + l] + gwn;
      q2[Q::gx + l] = qt[Q::gx + l] + gqn;
      q2[Q::gx + NV + l] = qt[Q::gx + NV + l] + gvn;
      if (t < T) dus[((long long)b * T + t) * NV + l] = du;
      cx[l] = dq; cx[NV + l] = dv;  // prox centre of the next iteration
    }
    if (l == 0) {
      double *ns = nodestat + node * 4;
      ns[0] = kkt; ns[1] = qt[Q::cost]; ns[2] = gap;
      double *as = admmstat + node * 4;
      as[0] = primal; as[1] = dual; as[2] = primal_rel; as[3] = drel;
    }
  }
}

// K3 for problems with general cost rows (agx_general.hpp): the optimality identities with every block,
//   Lu + Fu' lam' = -(Lqu' dq + (Luu + preg) du)
//   Lx + Fx' lam' - lam = -(Lxx dx + Lxu du + dreg dx),  Lxx = [[Lqq, Lqv], [Lqv', diag(Lvv) + Lvvd]]
// one lane per node; auxg = Lqv | Lvvd | Lqu of the node.
template <int NV>
__global__ void __launch_bounds__(64) k_node_kkt_gen(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                     const double *__restrict__ auxs, const double *__restrict__ auxg,
                                                     const double *__restrict__ dxs, const double *__restrict__ wss,
                                                     double *__restrict__ dus, double *__restrict__ nodestat,
                                                     const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV;
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long node = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= (long long)o.B * (T + 1)) return;
  const int b = (int)(node / (T + 1)), t = (int)(node % (T + 1));
  const DevState &S = st[b];
  if (S.done) return;
  const double preg = S.preg, dreg = S.dreg;
  const double *qt = qts + node * Q::SIZE;
  const double *ax = auxs + node * A::SIZE;
  const double *ag = auxg + node * (3 * A::B2);
  const double *Lqv = ag, *Lvvd = ag + A::B2, *Lqu = ag + 2 * A::B2;
  const double *dx = dxs + node * NX;
  double kkt = 0.0, gap = 0.0, du[NV];
  for (int i = 0; i < NV; ++i) du[i] = 0.0;
  if (t < T) {
    const double *w = wss + ((long long)b * T + t) * NV;
    for (int i = 0; i < NX; ++i) { kkt = fmax(kkt, fabs(qt[Q::f + i])); gap += fabs(qt[Q::f + i]); }
    for (int i = 0; i < NV; ++i) {
      double s = 0.0;
      for (int l = 0; l < NV; ++l)
        s += ax[A::M + i * A::LD + l] * w[l] + ax[A::tq + i * A::LD + l] * dx[l] + ax[A::tv + i * A::LD + l] * dx[NV + l];
      du[i] = s;
      dus[((long long)b * T + t) * NV + i] = s;
    }
    for (int i = 0; i < NV; ++i) {
      double s = (ax[A::Luu + i] + preg) * du[i];
      for (int l = 0; l < NV; ++l) s += Lqu[l * A::LD + i] * dx[l];
      kkt = fmax(kkt, fabs(s));
    }
  }
  if (t > 0)
    for (int i = 0; i < NV; ++i) {
      double hq = dreg * dx[i], hv = (ax[A::Lvv + i] + dreg) * dx[NV + i];
      for (int j = 0; j < NV; ++j) {
        hq += ax[A::Lqq + i * A::LD + j] * dx[j] + Lqv[i * A::LD + j] * dx[NV + j] + Lqu[i * A::LD + j] * du[j];
        hv += Lqv[j * A::LD + i] * dx[j] + Lvvd[i * A::LD + j] * dx[NV + j];
      }
      kkt = fmax(kkt, fmax(fabs(hq), fabs(hv)));
    }
  double *ns = nodestat + node * 4;
  ns[0] = kkt; ns[1] = qt[Q::cost]; ns[2] = gap;
  if (!o.has_con) ns[3] = 0.0;  // as k_node_kkt
}

// Exit path for large nv: the Hessian blocks of every node with CSQP's proximal terms,
//   H + sigma ([taux M]' [taux M] + I_x),  into a second tile; one 256-thread workgroup per node,
// M | tq | tv staged in LDS, thread (i, j) forms element [i][j] of the six blocks.
template <int NV>
__global__ void __launch_bounds__(256) k_sigma_tile_big(const DevOcp *__restrict__ op, const double *__restrict__ qts,
                                                        double *__restrict__ qt2s, const double *__restrict__ auxs) {
  typedef QT<NV> Q;
  typedef AUX<NV> A;
  __shared__ double sM[NV * NV], sq[NV * NV], sv[NV * NV];
  const DevOcp &o = *op;
  const int T = o.T, tid = threadIdx.x, nt = blockDim.x;
  const long long node = blockIdx.x;
  const int t = (int)(node % (T + 1));
  const double *qt = qts + node * Q::SIZE;
  double *q2 = qt2s + node * Q::SIZE;
  const double *ax = auxs + node * A::SIZE;
  const double sig = kSigma;
  if (t < T) {
    for (int e = tid; e < NV * NV; e += nt) {
      const int i = e / NV, j = e % NV;
      sM[e] = ax[A::M + i * A::LD + j]; sq[e] = ax[A::tq + i * A::LD + j]; sv[e] = ax[A::tv + i * A::LD + j];
    }
  }
  __syncthreads();
  for (int e = tid; e < NV * NV; e += nt) {
    const int i = e / NV, j = e % NV;
    double hww = 0.0, hqw = 0.0, hvw = 0.0, hqq = 0.0, hqv = 0.0, hvv = 0.0;
    if (t < T)
      for (int l = 0; l < NV; ++l) {
        const double Mli = sM[l * NV + i], tqli = sq[l * NV + i], tvli = sv[l * NV + i];
        const double Mlj = sM[l * NV + j], tqlj = sq[l * NV + j], tvlj = sv[l * NV + j];
        hww += Mli * Mlj; hqw += tqli * Mlj; hvw += tvli * Mlj;
        hqq += tqli * tqlj; hqv += tqli * tvlj; hvv += tvli * tvlj;
      }
    const double d = (i == j) ? sig : 0.0;
    const int o2 = i * Q::LD + j;
    q2[Q::Hqq + o2] = qt[Q::Hqq + o2] + sig * hqq + d;
    q2[Q::Hqv + o2] = qt[Q::Hqv + o2] + sig * hqv;
    q2[Q::Hvv + o2] = qt[Q::Hvv + o2] + sig * hvv + d;
    if (t < T) {
      q2[Q::Hww + o2] = qt[Q::Hww + o2] + sig * hww;
      q2[Q::Hqw + o2] = qt[Q::Hqw + o2] + sig * hqw;
      q2[Q::Hvw + o2] = qt[Q::Hvw + o2] + sig * hvw;
    }
  }
}

// K = M Kw - taux for large nv: 64 lanes per node, lane j < 2 nv = column of K
template <int NV>
__global__ void __launch_bounds__(256) k_gains_to_u_big(const DevOcp *__restrict__ op, const double *__restrict__ auxs,
                                                        const double *__restrict__ Kws, double *__restrict__ Kout,
                                                        const DevState *__restrict__ st) {
  constexpr int NX = 2 * NV;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T;
  const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long node = unit >> 6;
  const int j = (int)(unit & 63);
  if (node >= (long long)o.B * T || j >= NX) return;
  const int b = (int)(node / T), t = (int)(node % T);
  if (st ? !st[b].ls_acc : false) return;  // the sigma sweep in front of this launch skipped the instance: its Kws hold direction gains (st == null: every instance, constrained path)
  const double *ax = auxs + ((long long)b * (T + 1) + t) * A::SIZE;
  const double *Kw = Kws + node * NV * NX;
  double *K = Kout + node * NV * NX;
  const double *tx = (j < NV) ? ax + A::tq + j : ax + A::tv + (j - NV);
  for (int i = 0; i < NV; ++i) {
    double acc = -tx[i * A::LD];
    for (int l = 0; l < NV; ++l) acc += ax[A::M + i * A::LD + l] * Kw[l * NX + j];
    K[i * NX + j] = acc;
  }
}

// The dense feedback-gain GEMM of the exit path on the matrix cores:  K = M Kw - taux  per node,
// (nv x nv) (nv x 2nv) with nv = 30 padded to 32 x 64, as 2 x 4 output tiles of
// v_mfma_f64_16x16x4_f64 (8 k-steps each); one wave per node.  Operand maps (cdna_hip_programming.md,
// "f64 MFMA"): A[l & 15][k = l >> 4], B[k = l >> 4][l & 15], C/D col = l & 15, row = (l >> 4) + 4 reg.
typedef double agx_d4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ void __launch_bounds__(64) k_gains_to_u_mfma(const DevOcp *__restrict__ op, const double *__restrict__ auxs,
                                                        const double *__restrict__ Kws, double *__restrict__ Kout,
                                                        const DevState *__restrict__ st) {
  static_assert(NV > 16 && NV <= 32, "tiling below assumes 16 < nv <= 32");
  constexpr int NX = 2 * NV;
  typedef AUX<NV> A;
  const DevOcp &o = *op;
  const int T = o.T, lane = threadIdx.x;
  const long long node = blockIdx.x;  // b * T + t
  const int b = (int)(node / T), t = (int)(node % T);
  if (!st[b].ls_acc) return;  // not swept by the sigma pass in front of this launch
  const double *ax = auxs + ((long long)b * (T + 1) + t) * A::SIZE;
  const double *Kw = Kws + node * NV * NX;
  double *K = Kout + node * NV * NX;
  const int l15 = lane & 15, l4 = lane >> 4;
  // Every operand is loaded unconditionally (clamped index, 0 / 1 factor) and ahead of the MFMA chain that consumes it:
  // a load inside a condition is a branch, and a load between two MFMAs of one accumulator chain exposes its latency
  // 64 times per node (this kernel was 0.42 ms for 0.9 GB that way).
  // this lane's A operands for the 8 k-steps of both row tiles: M[16 ti + l15][4 ks + l4]
  double a[2][8];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int arow = 16 * ti + l15, ar = arow < NV ? arow : NV - 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kk = 4 * ks + l4, kc = kk < NV ? kk : NV - 1;
      a[ti][ks] = ax[A::M + ar * A::LD + kc] * ((arow < NV && kk < NV) ? 1.0 : 0.0);
    }
  }
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) {
    const int col = 16 * tj + l15, cc = col < NX ? col : NX - 1;
    const bool cq = cc < NV;
    double bv[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kk = 4 * ks + l4, kc = kk < NV ? kk : NV - 1;
      bv[ks] = Kw[kc * NX + cc] * ((kk < NV && col < NX) ? 1.0 : 0.0);
    }
    agx_d4 acc[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // C <- -taux tile: taux = [tq | tv], element [row][col]
        const int row = 16 * ti + l4 + 4 * r, rc = row < NV ? row : NV - 1;
        const double tx = ax[(cq ? A::tq + cc : A::tv + (cc - NV)) + rc * A::LD];
        acc[ti][r] = -tx * ((row < NV && col < NX) ? 1.0 : 0.0);
      }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][ks], bv[ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][ks], bv[ks], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + l4 + 4 * r;
        if (row < NV && col < NX) K[row * NX + col] = acc[ti][r];
      }
  }
}

}  // namespace agx
