// agimus_hip.hip -- C ABI of libagimus_hip.so (see include/agimus_hip.h).
// Host-side orchestration of the gfx950 kernels in agx_kernels.hpp: device-resident
// horizon buffers, the SQP iteration loop, warm-start shift, resident reference
// trajectories.  No CPU fallback: every compute entry point needs a HIP device.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "agx_kernels.hpp"

namespace {

thread_local std::string g_err;
int fail(const std::string &msg) {
  g_err = msg;
  return -1;
}
#define HIPCHK(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                \
  } while (0)

}  // namespace

struct agx_model {
  DevModel h;   // padded to the compiled capacity (see pad_capacity)
  int nvu = 0;  // joints of the caller's model
};

struct agx_ocp {
  DevModel hm;
  DevOcp ho;
  int nv = 0, nx = 0, nu = 0, T = 0, B = 0, tile = 0, stride = 0, device = 0;
  // Model sizes at run time: the kernels exist for a few CAPACITIES (7 joints: eight lanes per node; 30 / 32: a workgroup per
  // node).  A model with fewer joints is padded with massless, unit-armature joints that couple to nothing (pad_model): every
  // cross term with a real joint is an exact zero, the pad block of each QP is an independent problem with zero data, so the
  // real entries of xs, us, K, the costs and KKT residuals are what the exact-size recursion gives.  nv / nx / nu / stride above
  // are the internal (capacity) sizes; nvu / stride_u the caller's.  Arrays crossing the C ABI are repacked on the host.
  int nvu = 0, stride_u = 0;
  bool padded = false;
  std::vector<int> refmap[2];        // internal reference-tile element -> element of the caller's tile (-1: pad), running / terminal layout
  std::vector<double> reffill[2];    // value of the pad elements
  std::vector<double> stage;         // host staging of repacked arrays
  std::vector<double> h_first_u;     // first-node results in the caller's layout
  bool chain = false;
  std::vector<double> dt;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // device buffers
  DevModel *d_model = nullptr;
  DevOcp *d_ocp = nullptr;
  double *d_dt = nullptr, *d_xs = nullptr, *d_us = nullptr, *d_x0 = nullptr, *d_tiles = nullptr;
  double *d_Kws = nullptr, *d_kws = nullptr, *d_Kout = nullptr, *d_dx = nullptr, *d_du = nullptr;
  double *d_Kws_lqr = nullptr, *d_kws_lqr = nullptr;  // gains of the plain LQR pass when it runs next to the ADMM factorisation
  bool admm_prefactor = true;  // AGX_ADMM_PREFACTOR=0: LQR pass, then the factorisation inside the first ADMM iteration
  double *d_qt = nullptr, *d_aux = nullptr, *d_w = nullptr, *d_nodestat = nullptr;  // QP tiles, aux tiles, acceleration steps
  // constrained problems (agx_admm.hpp): augmented tiles, constraint values / collision Jacobians,
  // multipliers y (persistent across solves), slack z, prox centre, per-node residual norms
  bool has_con = false;
  std::vector<int> shift_nodes;  // large models: nodes with dt_i != dt_0 (integrated by the warm-start shift)
  int *d_shift_nodes = nullptr;
  bool general = false;       // ControlGrav / FrameVelocity cost rows: one-lane GEN kernels (agx_general.hpp)
  double *d_auxg = nullptr;   // [B][T+1][3 nv 8]: Lqv | Lvvd | Lqu of every node (general problems)
  double *d_qt2 = nullptr, *d_cg = nullptr, *d_cjac = nullptr, *d_y = nullptr, *d_z = nullptr, *d_cx = nullptr, *d_admmstat = nullptr,
         *d_fac = nullptr,  // Riccati factors of every node for the gradient-only ADMM sweeps [B][T][192]
         *d_segP = nullptr;  // closed-loop transition products of the horizon segments [B][kSeg][4][64] (agx_admm.hpp)
  bool admm_segments = true;  // AGX_ADMM_SEGMENTS=0: gradient-only sweeps on one wave per instance
  int qt_size = 0, aux_size = 0;
  bool k1_lanes = true;  // AGX_K1_LANES=0 selects the one-lane-per-node derivative kernel
  bool lanes_ok = true;  // problem fits the LDS staging of the 8-lanes-per-node kernel
  bool lanes_coll = false;  // ... in its variant with one collision cost row
  bool speculate = true;  // AGX_SPECULATE_GAINS=0: gains sweep only on exit
  bool gains_mfma = true; // AGX_GAINS_MFMA=0: scalar K = M Kw - taux for large models
  bool fuse_kkt = false;     // AGX_FUSED_KKT=1: K3 inside the forward pass of k_riccati_mx instead of its own launch (measured: no gain)
  bool riccati_mx = true;    // AGX_RICCATI_MX=0: nv <= 7 sweeps on the 8 x 8 lane grid (k_riccati) instead of the MFMA operand layout (k_riccati_mx)
  bool riccati_mfma = true;  // AGX_RICCATI_MFMA=0: large models sweep with the LDS Gauss-Jordan kernel (k_riccati_big)
  bool riccati_blk = true;   // AGX_RICCATI_BLK=0: the matrix-core sweep with the per-wave v_readlane elimination (k_riccati_mfma) instead of the blocked inverse (k_riccati_blk)
  // Exact two-level sweep (agx_riccati_mx2.hpp): the horizon in mx2_S segments swept in parallel; 0 = the one-wave sweep.
  // Chosen from the batch at creation (small batches leave most of the chip idle), AGX_MX2_SEGMENTS=n overrides (0: off).
  int mx2_S = 0;
  double *d_mx2_elem = nullptr, *d_mx2_bnd = nullptr, *d_mx2_cl = nullptr;  // [2][B][S][3][256] segment elements, [2][B][S][256] boundary value functions, [B][S][256] transitions under the gains
  bool k1_fused = true;     // AGX_K1_FUSED=0: running and terminal nodes of the derivative pass as two launches (profiling)
  // Batch policy (agx_ocp_set_quorum): the SQP loop of a batch step ends once this fraction of the instances has
  // finished, the ADMM loop of an SQP iteration once this fraction of the QPs has converged; the others keep their
  // iterate (solved = 0 / qp_iters = max_qp_iters) and continue from it at the next MPC step, as a lone controller
  // that ran into max_solve_time would.  1.0 = wait for everyone (the default).
  double quorum_sqp = 1.0, quorum_qp = 1.0;
  bool con_lanes = true;          // constraint rows are control limits / state bounds / collision distances: k_con_eval_lj (AGX_CON_LANES=0: one lane per node)
  bool admm_loop_always = false;  // AGX_ADMM_LOOP=2 (tests): k_admm_loop whatever the number of unfinished instances
  bool admm_loop = true;      // AGX_ADMM_LOOP=0: every ADMM iteration as three launches (sweep, update, reduce) instead of k_admm_loop
  int n_unfinished = 1 << 30; // instances the SQP loop still works on (host's last count): k_admm_loop serves the tail of a step, when
                              // few instances are left (with the whole batch active the node-parallel update across the chip is faster)
  bool fold_publish = false;  // small batches (B <= 204, polled hand-off): the head / accept kernels hand the counters to the host themselves (last workgroup to arrive) and the host asks after the head from the second iteration on; AGX_FOLD_PUBLISH=0/1 overrides
  int n_cu = 256;         // compute units of the device (grid of the persistent kernels)
  bool no_empty = false;  // AGX_NO_EMPTY_LAUNCHES=1 (profiling): the host asks after the head of the step whether anybody searches and skips the trial launches otherwise, so that per-kernel averages are those of working launches
  double *d_ref = nullptr;  // owned tile [B][T+1][stride]
  // asynchronous reference upload (agx_ocp_set_refs_async): second tile / frame table filled by the copy stream while the
  // solver works on the first; the next solve waits for the copy's event and swaps the two
  double *d_ref_back = nullptr;
  int *d_frames_back = nullptr;
  bool refs_pending = false, refs_pending_frames = false;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_refs = nullptr, ev_snap = nullptr, ev_dl = nullptr;
  // asynchronous result download (agx_ocp_download_async): device-side snapshot of xs | us | K taken in the solver's stream
  double *d_snap = nullptr;
  bool dl_pending = false;
  int *d_frames = nullptr;  // owned [B][T+1][AGX_MAX_ROWS]
  bool frames_set = false;
  DevState *d_state = nullptr;
  int *d_ndone = nullptr;
  // pinned, fine-grained (coherent) mapped host words, 64 bit each: [0] finished-instance count, [1] its sequence
  // stamp, [2] stamp of the packed results, [3] scratch value, [4] ADMM converged count, [5] its stamp.
  // Stamps are 64-bit and only ever grow (no wrap in practice); slots start at ~0, which no stamp takes.
  unsigned long long *h_ndone = nullptr;
  unsigned long long *h_ndone_dev = nullptr;  // the same words as the device sees them (mapped host memory)
  unsigned long long seq = 0;
  unsigned ls_handed = 0, ls_stale = 0;  // last values seen of the device counters of handed-on line-search trials / iterations ended with stale tiles (d_ndone[3], [4]; never reset)
  bool poll = true;  // AGX_HOST_POLL=0: stream-ordered copies + synchronize instead of polled mapped words
  double *d_first = nullptr, *h_first = nullptr;  // packed first-node results [B][first_stride], device / pinned host
  int first_stride = 0;
  double *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  RefView rv{};
  // resident trajectory
  double *d_traj = nullptr, *d_pts = nullptr;
  double *d_sine = nullptr;  // q0, amp, puls, scale, t0
  int n_points = 0;
  // non-uniform horizon indexes (TrajectoryBuffer): empty = node t looks at sample k0 + t
  std::vector<int> hidx;
  int *d_hidx = nullptr;
  int win_k0 = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
  int last_max_iter = 0;
  // in-situ kernel timing (agx_ocp_profile): event pairs around the launches of the SQP loop
  bool prof = false;
  std::vector<hipEvent_t> prof_ev;  // [2 * k] start, [2 * k + 1] stop
  std::vector<int> prof_kind;
  double prof_ms[3] = {0, 0, 0};
  long long prof_n[3] = {0, 0, 0};
};

namespace {

int set_device(agx_ocp *o) {
  HIPCHK(hipSetDevice(o->device));
  return 0;
}

// capacity a model of nv joints runs at: the smallest compiled size that holds it (0: none)
int pad_capacity(int nv);

// ---- repacking between the caller's layout (nvu joints) and the internal one (nv = capacity) -----------------------------
// rows of `blocks` blocks of nvu doubles  <->  rows of `blocks` blocks of nv doubles
void pad_rows(double *dst, const double *src, size_t rows, int blocks, int nvu, int nv, double fill) {
  for (size_t r = 0; r < rows; ++r)
    for (int b = 0; b < blocks; ++b) {
      double *d = dst + (r * blocks + b) * nv;
      const double *q = src + (r * blocks + b) * nvu;
      for (int i = 0; i < nvu; ++i) d[i] = q[i];
      for (int i = nvu; i < nv; ++i) d[i] = fill;
    }
}
void unpad_rows(double *dst, const double *src, size_t rows, int blocks, int nvu, int nv) {
  for (size_t r = 0; r < rows; ++r)
    for (int b = 0; b < blocks; ++b) {
      double *d = dst + (r * blocks + b) * nvu;
      const double *q = src + (r * blocks + b) * nv;
      for (int i = 0; i < nvu; ++i) d[i] = q[i];
    }
}
// host -> device of [rows][blocks][nvu] into [rows][blocks][nv]; synchronous for padded handles (staging is reused)
int up(agx_ocp *o, double *d_dst, const double *h_src, size_t rows, int blocks, double fill = 0.0) {
  if (!o->padded) {
    HIPCHK(hipMemcpyAsync(d_dst, h_src, sizeof(double) * rows * blocks * o->nv, hipMemcpyHostToDevice, o->stream));
    return 0;
  }
  o->stage.resize(rows * blocks * o->nv);
  pad_rows(o->stage.data(), h_src, rows, blocks, o->nvu, o->nv, fill);
  HIPCHK(hipMemcpyAsync(d_dst, o->stage.data(), sizeof(double) * o->stage.size(), hipMemcpyHostToDevice, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}
// device -> host of [rows][blocks][nv] into [rows][blocks][nvu]; padded handles synchronize inside
int down(agx_ocp *o, double *h_dst, const double *d_src, size_t rows, int blocks) {
  if (!o->padded) {
    HIPCHK(hipMemcpyAsync(h_dst, d_src, sizeof(double) * rows * blocks * o->nv, hipMemcpyDeviceToHost, o->stream));
    return 0;
  }
  o->stage.resize(rows * blocks * o->nv);
  HIPCHK(hipMemcpyAsync(o->stage.data(), d_src, sizeof(double) * o->stage.size(), hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  unpad_rows(h_dst, o->stage.data(), rows, blocks, o->nvu, o->nv);
  return 0;
}
// gains: [n][nv][2 nv] on the device -> [n][nvu][2 nvu]
int down_gains(agx_ocp *o, double *h_dst, const double *d_src, size_t n) {
  if (!o->padded) {
    HIPCHK(hipMemcpyAsync(h_dst, d_src, sizeof(double) * n * o->nu * o->nx, hipMemcpyDeviceToHost, o->stream));
    return 0;
  }
  const int nv = o->nv, nvu = o->nvu;
  o->stage.resize(n * nv * 2 * nv);
  HIPCHK(hipMemcpyAsync(o->stage.data(), d_src, sizeof(double) * o->stage.size(), hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  for (size_t r = 0; r < n; ++r)
    for (int i = 0; i < nvu; ++i)
      unpad_rows(h_dst + (r * nvu + i) * 2 * nvu, o->stage.data() + (r * nv + i) * 2 * nv, 1, 2, nvu, nv);
  return 0;
}
// reference tile [nodes][stride_u] (running layout, `term_every` > 0: every term_every-th node in the terminal layout)
void pad_ref_tile(const agx_ocp *o, double *dst, const double *src, size_t B, int T) {
  for (size_t b = 0; b < B; ++b)
    for (int t = 0; t <= T; ++t) {
      const int lay = t == T ? 1 : 0;
      const double *q = src + (b * (T + 1) + t) * o->stride_u;
      double *d = dst + (b * (T + 1) + t) * o->stride;
      const int *map = o->refmap[lay].data();
      const double *fill = o->reffill[lay].data();
      for (int e = 0; e < o->stride; ++e) d[e] = map[e] >= 0 ? q[map[e]] : fill[e];
    }
}

int ensure_scratch(agx_ocp *o, size_t bytes) {
  if (bytes <= o->scratch_bytes) return 0;
  if (o->d_scratch) HIPCHK(hipFree(o->d_scratch));
  o->d_scratch = nullptr;
  o->scratch_bytes = 0;
  HIPCHK(hipMalloc(&o->d_scratch, bytes));
  o->scratch_bytes = bytes;
  return 0;
}

void fill_rows(const agx_cost_row *rows, int n, int nv, int nvu, DevRows &d) {
  std::memset(&d, 0, sizeof(d));
  d.n = n;
  d.nvu = nvu;
  int off = 0;
  for (int r = 0; r < n; ++r) {
    d.kind[r] = rows[r].kind;
    d.act[r] = rows[r].activation;
    d.active[r] = rows[r].active;
    d.frame[r] = rows[r].frame;
    d.frame_b[r] = rows[r].frame_b;
    d.alpha[r] = rows[r].alpha;
    d.weight[r] = rows[r].weight;
    d.nref[r] = agx_row_nref(rows[r].kind, nv);
    d.nr[r] = agx_row_nr(rows[r].kind, nv);
    d.off[r] = off;
    off += 1 + d.nref[r] + d.nr[r];
    if (rows[r].active && (rows[r].kind == AGX_RES_CONTROL_GRAV || rows[r].kind == AGX_RES_FRAME_VELOCITY)) d.general = 1;
  }
}

// component k of a residual of `kind` over nv (capacity) joints -> the caller's component over nvu joints, -1 for a pad joint
int user_component(int kind, int k, int nv, int nvu, bool is_ref) {
  const bool joint_blocks = kind == AGX_RES_STATE || kind == AGX_RES_CONTROL || (kind == AGX_RES_CONTROL_GRAV && !is_ref);
  if (!joint_blocks || nv == nvu) return k;
  const int blk = k / nv, i = k % nv;
  return i < nvu ? blk * nvu + i : -1;
}

int fill_cons(const agx_constraint_row *rows, int n, int nv, int nvu, const DevModel &m, DevCons &d, bool terminal) {
  std::memset(&d, 0, sizeof(d));
  int off = 0;
  for (int r = 0; r < n; ++r) {
    const agx_constraint_row &c = rows[r];
    if (!c.active) continue;
    // the terminal node has no control: residuals on u vanish there (crocoddyl evaluates them with nu = 0)
    if (terminal && (c.kind == AGX_RES_CONTROL || c.kind == AGX_RES_CONTROL_GRAV)) continue;
    if (d.n >= AGX_MAX_CONS) return fail("agx_ocp_create: at most 4 active constraint rows per node type");
    if (c.kind != AGX_RES_STATE && c.kind != AGX_RES_CONTROL && !agx::cons_has_dense_rows(c.kind))
      return fail("agx_ocp_create: unknown constraint residual kind");
    const int nr = agx_row_nr(c.kind, nv), nref = agx_row_nref(c.kind, nv);
    if (off + nr > AGX_MAX_NC) return fail("agx_ocp_create: more than " + std::to_string(AGX_MAX_NC) + " constraint components per node");
    if (!c.lower || !c.upper) return fail("agx_ocp_create: constraint bounds missing");
    const int i = d.n++;
    d.kind[i] = c.kind; d.frame[i] = c.frame; d.frame_b[i] = c.frame_b; d.off[i] = off; d.nr[i] = nr;
    for (int k = 0; k < nref; ++k) {
      const int ku = user_component(c.kind, k, nv, nvu, true);
      d.ref[i][k] = (c.ref && ku >= 0) ? c.ref[ku] : 0.0;
    }
    for (int k = 0; k < nr; ++k) {
      const int ku = user_component(c.kind, k, nv, nvu, false);  // pad joints: unbounded components
      d.lb[off + k] = ku >= 0 ? c.lower[ku] : -INFINITY; d.ub[off + k] = ku >= 0 ? c.upper[ku] : INFINITY;
      if (!(d.lb[off + k] <= d.ub[off + k])) return fail("agx_ocp_create: constraint with lower > upper");
    }
    if (c.kind == AGX_RES_COLLISION) {
      if (c.frame < 0 || c.frame >= m.nframes || c.frame_b < 0 || c.frame_b >= m.nframes || !agx::frame_has_geometry(m, c.frame) ||
          !agx::frame_has_geometry(m, c.frame_b))
        return fail("agx_ocp_create: collision constraint refers to a frame without geometry");
      if (agx::frame_is_box(m, c.frame) && agx::frame_is_box(m, c.frame_b))
        return fail("agx_ocp_create: box / box collision pairs are not supported");
      d.coll_slot[i] = d.ncoll++;
    }
    if (c.kind != AGX_RES_COLLISION && agx::cons_has_dense_rows(c.kind)) {  // nr scalar rows with dense gradients in q: nr Jacobian slots
      if (c.frame < 0 || c.frame >= m.nframes) return fail("agx_ocp_create: constraint frame id out of range");
      d.coll_slot[i] = d.ncoll;
      d.ncoll += nr;
    }
    if (c.kind == AGX_RES_FRAME_VELOCITY && (c.frame_b < 0 || c.frame_b > 2)) return fail("agx_ocp_create: FrameVelocity constraint: reference frame must be 0 (WORLD), 1 (LOCAL) or 2 (LOCAL_WORLD_ALIGNED)");
    if (d.ncoll > AGX_MAX_DENSE) return fail("agx_ocp_create: at most 8 constraint components with a dense Jacobian (collision pairs, frame residuals, ControlGrav) per node type");
    off += nr;
  }
  d.nc = off;
  return 0;
}

// ---- dispatch over the compiled (NV, CHAIN) instantiations --------------------
// The library is built as one translation unit per group of capacities (AGX_GROUP = 0: 7 joints, 1: 16, 30 and 32, compiled in
// parallel by backend.build(), namespace and entry points suffixed per group, csrc/agx_front.py
// generates the forwarding entry points) or, without AGX_GROUP, as a single translation unit.
#if defined(AGX_ONLY_NV7) || (defined(AGX_GROUP) && AGX_GROUP == 0)  // AGX_ONLY_NV7: development builds, short compile
#define AGX_FOR_NV(MACRO) MACRO(7)
#elif defined(AGX_ONLY_NV30) || (defined(AGX_GROUP) && AGX_GROUP == 1)
#define AGX_FOR_NV(MACRO) MACRO(16) MACRO(30) MACRO(32)
#else
#define AGX_FOR_NV(MACRO) MACRO(7) MACRO(16) MACRO(30) MACRO(32)
#endif

int pad_capacity(int nv) {
  int best = 0;
#define AGX_CAP(N) if (N >= nv && (best == 0 || N < best)) best = N;
  AGX_FOR_NV(AGX_CAP)
#undef AGX_CAP
  return best;
}

template <typename F>
int dispatch(int nv, bool chain, F &&f) {
  switch (nv) {
#define AGX_CASE(N)                                                  \
  case N:                                                            \
    if constexpr (N <= 8) { if (chain) return f(std::integral_constant<int, N>(), std::true_type()); } \
    return f(std::integral_constant<int, N>(), std::false_type());  /* large models: the tree code path serves chains too */
    AGX_FOR_NV(AGX_CASE)
#undef AGX_CASE
  }
  return fail("no kernel instantiation for nv = " + std::to_string(nv) + " (compiled capacities: 7, 30, 32)");
}

int ensure_canonical_tiles(agx_ocp *o) {
  if (o->d_tiles) return 0;
  HIPCHK(hipMalloc((void **)&o->d_tiles, sizeof(double) * (size_t)o->B * (o->T + 1) * o->tile));
  return 0;
}

int launch_calc_diff(agx_ocp *o, bool masked, bool running_only = false) {
  if (ensure_canonical_tiles(o)) return -1;
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    const long long units = (long long)o->B * o->T;
    const int grid = (int)((units + 63) / 64);
    if constexpr (NV <= 7) if (o->general) {
      hipLaunchKernelGGL((agx::k_calc_diff<NV, CH, true>), dim3(grid), dim3(64), 0, o->stream, o->d_model, o->d_ocp, o->d_dt, o->d_xs,
                         o->d_us, o->rv, o->d_tiles, masked ? o->d_state : nullptr);
      if (running_only) { HIPCHK(hipGetLastError()); return 0; }
      hipLaunchKernelGGL((agx::k_calc_diff_term<NV, CH, true>), dim3((o->B + 63) / 64), dim3(64), 0, o->stream, o->d_model, o->d_ocp,
                         o->d_xs, o->rv, o->d_tiles, masked ? o->d_state : nullptr);
      HIPCHK(hipGetLastError());
      return 0;
    }
    hipLaunchKernelGGL((agx::k_calc_diff<NV, CH>), dim3(grid), dim3(64), 0, o->stream, o->d_model, o->d_ocp, o->d_dt, o->d_xs,
                       o->d_us, o->rv, o->d_tiles, masked ? o->d_state : nullptr);
    if (running_only) { HIPCHK(hipGetLastError()); return 0; }
    hipLaunchKernelGGL((agx::k_calc_diff_term<NV, CH>), dim3((o->B + 63) / 64), dim3(64), 0, o->stream, o->d_model, o->d_ocp,
                       o->d_xs, o->rv, o->d_tiles, masked ? o->d_state : nullptr);
    HIPCHK(hipGetLastError());
    return 0;
  });
}

// K1 production: QP tiles in acceleration-input form.  Serial chains use the 8-lanes-per-node
// kernel (agx_k1_lanes.hpp); trees fall back to one lane per node.
// phase 1: the pass runs at the trial iterates (staging halves of xs / us) of the instances in the line search and leaves
// their tiles in place of the current ones (k_sqp_head / k_sqp_accept, agx_kernels.hpp); nv <= 7 only.
int launch_calc_qp(agx_ocp *o, bool running_only = false, bool term_only = false, int phase = 0) {
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    const long long units = (long long)o->B * o->T;
    const double *xs_in = phase ? o->d_xs + (size_t)o->B * (o->T + 1) * o->nx : o->d_xs;
    const double *us_in = phase ? o->d_us + (size_t)o->B * o->T * o->nu : o->d_us;
    if constexpr (NV <= 7) if (o->general) {
      if (!term_only)
        hipLaunchKernelGGL((agx::k_calc_qp<NV, CH, true>), dim3((int)((units + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp,
                           o->d_dt, xs_in, us_in, o->rv, o->d_qt, o->d_aux, o->d_state, o->d_auxg, phase);
      if (!running_only)
        hipLaunchKernelGGL((agx::k_calc_qp_term<NV, CH, true>), dim3((o->B + 63) / 64), dim3(64), 0, o->stream, o->d_model, o->d_ocp,
                           xs_in, o->rv, o->d_qt, o->d_aux, o->d_state, o->d_auxg, phase);
      HIPCHK(hipGetLastError());
      return 0;
    }
    bool lanes = false;
    if constexpr (NV <= 7) lanes = CH && o->k1_lanes && o->lanes_ok;
    if constexpr (NV <= 7) if (lanes) {
      const int n_run = (int)((units * 8 + 63) / 64), n_term = (int)(((long long)o->B * 8 + 63) / 64);
#define AGX_LAUNCH_LJ(COLL)                                                                                                              \
  do {                                                                                                                                   \
    if (o->k1_fused && !term_only && !running_only) { /* both node types in one launch */                                                \
      hipLaunchKernelGGL((agx::k_calc_qp_lj_all<NV, COLL>), dim3(n_run + n_term), dim3(64), 0, o->stream, o->d_model, o->d_ocp, o->d_dt, \
                         xs_in, us_in, o->rv, o->d_qt, o->d_aux, o->d_state, n_run, phase);                                             \
    } else {                                                                                                                             \
      if (!term_only)                                                                                                                    \
        hipLaunchKernelGGL((agx::k_calc_qp_lj<NV, false, COLL>), dim3(n_run), dim3(64), 0, o->stream, o->d_model, o->d_ocp, o->d_dt,     \
                           xs_in, us_in, o->rv, o->d_qt, o->d_aux, o->d_state, phase);                                                  \
      if (!running_only)                                                                                                                 \
        hipLaunchKernelGGL((agx::k_calc_qp_lj<NV, true, COLL>), dim3(n_term), dim3(64), 0, o->stream, o->d_model, o->d_ocp, o->d_dt,     \
                           xs_in, us_in, o->rv, o->d_qt, o->d_aux, o->d_state, phase);                                                  \
    }                                                                                                                                    \
  } while (0)
      if (o->lanes_coll) AGX_LAUNCH_LJ(true); else AGX_LAUNCH_LJ(false);
#undef AGX_LAUNCH_LJ
      HIPCHK(hipGetLastError());
      return 0;
    }
    if constexpr (NV > 8) {
      // large models: one workgroup per node, running and terminal nodes in one launch (agx_big_k1.hpp)
      (void)lanes; (void)running_only;
      if (term_only) return 0;  // the launch of the running nodes (profiled path) already covered the terminal ones
      hipLaunchKernelGGL((agx::k_calc_qp_wg<NV>), dim3((int)(units + o->B)), dim3(256), 0, o->stream, o->d_model, o->d_ocp, o->d_dt, xs_in,
                         us_in, o->rv, o->d_qt, o->d_aux, o->d_state, phase);
    } else if (!lanes) {
      if (!term_only)
        hipLaunchKernelGGL((agx::k_calc_qp<NV, CH>), dim3((int)((units + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp,
                           o->d_dt, xs_in, us_in, o->rv, o->d_qt, o->d_aux, o->d_state, (double *)nullptr, phase);
      if (!running_only)
        hipLaunchKernelGGL((agx::k_calc_qp_term<NV, CH>), dim3((o->B + 63) / 64), dim3(64), 0, o->stream, o->d_model, o->d_ocp,
                           xs_in, o->rv, o->d_qt, o->d_aux, o->d_state, (double *)nullptr, phase);
    }
    HIPCHK(hipGetLastError());
    return 0;
  });
}

// K3 (k_node_kkt) folded into the forward pass of K2 (AGX_FUSED_KKT=1; off by default: the 12-value prefetch sets of
// that pass end up behind register copies on the loop latch and the pass gains 65 us for the 51 us K3 takes alone):
// the MFMA-layout sweep of unconstrained problems without general cost rows (those have their own node kernels).
static inline bool fused_kkt(const agx_ocp *o) { return o->fuse_kkt && o->riccati_mx && o->nv <= 7 && !o->has_con && !o->general; }

// K2: direction sweep; with `pair` the speculative gains sweep of SQP iteration `iter` rides in the same launch
int launch_riccati(agx_ocp *o, int forward, bool pair = false, int iter = 0, const double *tiles = nullptr) {
  if (o->nv <= 7 && o->T + 1 > 512) return fail("the sweeps stage the step lengths of up to 511 nodes in LDS: horizon too long");
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    const double *qt = tiles ? tiles : o->d_qt;
    if constexpr (NV <= 7) if (o->mx2_S >= 2 && o->riccati_mx && !tiles) {
      // small batches: segments in parallel, exact boundary value functions (agx_riccati_mx2.hpp)
      const int S = o->mx2_S, P2 = pair ? 2 : 1;
      if (!o->d_mx2_elem) {
        HIPCHK(hipMalloc((void **)&o->d_mx2_elem, sizeof(double) * 2 * (size_t)o->B * S * 768));
        HIPCHK(hipMalloc((void **)&o->d_mx2_bnd, sizeof(double) * 2 * (size_t)o->B * S * 256));
        HIPCHK(hipMalloc((void **)&o->d_mx2_cl, sizeof(double) * (size_t)o->B * S * 256));
      }
      hipLaunchKernelGGL((agx::k_riccati_mx2_elem<NV>), dim3(P2 * o->B * S), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_aux,
                         o->d_Kws, o->d_kws, o->d_Kout, o->d_state, o->d_mx2_elem, o->d_mx2_bnd, o->d_mx2_cl, S, pair ? 1 : 0, 0, 1, iter);
      hipLaunchKernelGGL((agx::k_riccati_mx2_sweep<NV>), dim3(P2 * o->B * (S - 1)), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_aux,
                         o->d_Kws, o->d_kws, o->d_Kout, o->d_state, o->d_mx2_elem, o->d_mx2_bnd, o->d_mx2_cl, S, pair ? 1 : 0, 0, 1);
      if (forward)
        hipLaunchKernelGGL((agx::k_riccati_mx2_fwd<NV>), dim3(o->B * S), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_Kws, o->d_kws,
                           o->d_dx, o->d_w, o->d_state, o->d_mx2_cl, S);
      HIPCHK(hipGetLastError());
      return 0;
    }
    if constexpr (NV <= 7) {
      const int fwd = forward ? (fused_kkt(o) ? 2 : 1) : 0;  // 2: K3 rides along in the forward pass
      if (pair && o->riccati_mx)
        hipLaunchKernelGGL((agx::k_riccati_mx_pair<NV>), dim3(16 * ((o->B + 7) / 8)), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_aux,
                           o->d_Kws, o->d_kws, o->d_dx, o->d_w, o->d_Kout, o->d_state, iter, fused_kkt(o) ? 2 : 1, o->d_du, o->d_nodestat);
      else if (pair)
        hipLaunchKernelGGL((agx::k_riccati_pair<NV>), dim3(2 * o->B), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_aux,
                           o->d_Kws, o->d_kws, o->d_dx, o->d_w, o->d_du, o->d_Kout, o->d_state, iter);
      else if (o->riccati_mx)
        hipLaunchKernelGGL((agx::k_riccati_mx<NV, false>), dim3(o->B), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, qt, o->d_aux,
                           o->d_Kws, o->d_kws, o->d_dx, o->d_w, o->d_Kout, o->d_state, fwd, 0, o->d_du, o->d_nodestat);
      else
        hipLaunchKernelGGL((agx::k_riccati<NV, false>), dim3(o->B), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, qt, o->d_aux,
                           o->d_Kws, o->d_kws, o->d_dx, o->d_w, o->d_du, o->d_Kout, o->d_state, forward, 0);
    } else {
      (void)pair; (void)iter;
      if constexpr (NV >= 16) if (o->riccati_mfma) {
        bool blk = o->riccati_blk;
        if constexpr (NV == 16) blk = true;  // k_riccati_mfma tiles 16 < nv <= 32 only
        if (blk)
          hipLaunchKernelGGL((agx::k_riccati_blk<NV>), dim3(o->B), dim3(256), 0, o->stream, o->d_ocp, o->d_dt, qt, o->d_Kws, o->d_kws, o->d_dx,
                             o->d_w, o->d_state, forward, 0);
        if constexpr (NV > 16) if (!blk)
          hipLaunchKernelGGL((agx::k_riccati_mfma<NV>), dim3(o->B), dim3(256), 0, o->stream, o->d_ocp, o->d_dt, qt, o->d_Kws, o->d_kws, o->d_dx,
                             o->d_w, o->d_state, forward, 0);
        HIPCHK(hipGetLastError());
        return 0;
      }
      hipLaunchKernelGGL((agx::k_riccati_big<NV>), dim3(o->B), dim3(256), 0, o->stream, o->d_ocp, o->d_dt, qt, o->d_Kws, o->d_kws, o->d_dx,
                         o->d_w, o->d_state, forward, 0);
    }
    HIPCHK(hipGetLastError());
    return 0;
  });
}

// K3 (node shares of the KKT residual, cost, gaps; du) and the head of the step: instance totals, convergence test and the
// first trial iterate of the line search (the caller runs the trial rounds: line_search_rounds).
agx::HostWords host_words(agx_ocp *o, bool line_search_counters, unsigned long long seq) {
  agx::HostWords hw{nullptr, nullptr, nullptr, nullptr, 0};
  if (!o->poll || !o->fold_publish || seq == 0) return hw;
  hw.done = o->h_ndone_dev + 0;
  if (line_search_counters) { hw.handed = o->h_ndone_dev + 6; hw.stale = o->h_ndone_dev + 7; }
  hw.seq = o->h_ndone_dev + 1;
  hw.stamp = seq;
  return hw;
}

// `seq` != 0 (small batches): the head's last workgroup hands the finished-instance count to the host under that stamp
int launch_step(agx_ocp *o, int iter, int max_iter, int mode, bool with_node_kkt = true, bool with_step = true, unsigned long long seq = 0) {
  if (o->T + 1 > 512) return fail("step kernel supports horizons up to 511 nodes");
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    (void)CH;
    const long long nodes = (long long)o->B * (o->T + 1);
    double *xs_t = o->d_xs + (size_t)o->B * (o->T + 1) * o->nx, *us_t = o->d_us + (size_t)o->B * o->T * o->nu;
    if constexpr (NV <= 7) if (o->general) {
      if (with_node_kkt)
        hipLaunchKernelGGL((agx::k_node_kkt_gen<NV>), dim3((int)((nodes + 63) / 64)), dim3(64), 0, o->stream, o->d_ocp, o->d_qt, o->d_aux,
                           o->d_auxg, o->d_dx, o->d_w, o->d_du, o->d_nodestat, o->d_state);
      if (with_step)
        hipLaunchKernelGGL((agx::k_sqp_head<NV>), dim3(o->B), dim3(128), 0, o->stream, o->d_ocp, o->d_xs, o->d_us, o->d_dx, o->d_du, xs_t, us_t,
                           o->d_nodestat, o->d_state, iter, max_iter, mode, o->d_ndone, host_words(o, false, seq));
      HIPCHK(hipGetLastError());
      return 0;
    }
    if (with_node_kkt && !fused_kkt(o)) {
      if constexpr (NV <= 7)
        hipLaunchKernelGGL((agx::k_node_kkt<NV>), dim3((int)((nodes * 8 + 255) / 256)), dim3(256), 0, o->stream, o->d_ocp, o->d_qt, o->d_aux,
                           o->d_dx, o->d_w, o->d_du, o->d_nodestat, o->d_state);
      else
        hipLaunchKernelGGL((agx::k_node_kkt_big<NV>), dim3((int)((nodes + 1) / 2)), dim3(64), 0, o->stream, o->d_ocp, o->d_qt, o->d_aux,
                           o->d_dx, o->d_w, o->d_du, o->d_nodestat, o->d_state);
    }
    if (!with_step) { HIPCHK(hipGetLastError()); return 0; }
    hipLaunchKernelGGL((agx::k_sqp_head<NV>), dim3(o->B), dim3(128), 0, o->stream, o->d_ocp, o->d_xs, o->d_us, o->d_dx, o->d_du, xs_t, us_t,
                       o->d_nodestat, o->d_state, iter, max_iter, mode, o->d_ndone, host_words(o, false, seq));
    HIPCHK(hipGetLastError());
    return 0;
  });
}

// exit path: the sigma (proximal) Riccati sweep that yields the gains the solver reports, one kernel.
// gmode 0: every instance; 2: only instances whose last direction has no (speculative) sweep yet.
int launch_gains(agx_ocp *o, int gmode = 0) {
  if (o->nv <= 7 && o->T + 1 > 512) return fail("the sweeps stage the step lengths of up to 511 nodes in LDS: horizon too long");
  if (o->nv > 7 && !o->d_qt2) HIPCHK(hipMalloc((void **)&o->d_qt2, sizeof(double) * (size_t)o->B * (o->T + 1) * o->qt_size));
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    if constexpr (NV <= 7) {
      if (o->riccati_mx)
        hipLaunchKernelGGL((agx::k_riccati_mx<NV, true>), dim3(o->B), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_aux, o->d_Kws,
                           o->d_kws, o->d_dx, o->d_w, o->d_Kout, o->d_state, 0, gmode, (double *)nullptr, (double *)nullptr);
      else
      hipLaunchKernelGGL((agx::k_riccati<NV, true>), dim3(o->B), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_aux, o->d_Kws,
                         o->d_kws, o->d_dx, o->d_w, o->d_du, o->d_Kout, o->d_state, 0, gmode);
    } else {
      // large models: sigma-augmented tiles, sweep (instances selected by gmode, as riccati_body), gains to u-space for
      // the instances the sweep took
      const long long nodes = (long long)o->B * (o->T + 1);
      const int gsel = gmode == 0 ? 1 : gmode;
      hipLaunchKernelGGL((agx::k_sigma_tile_big<NV>), dim3((int)nodes), dim3(256), 0, o->stream, o->d_ocp, o->d_qt, o->d_qt2, o->d_aux);
      bool swept = false;
      if constexpr (NV >= 16) if (o->riccati_mfma) {
        bool blk = o->riccati_blk;
        if constexpr (NV == 16) blk = true;
        if (blk)
          hipLaunchKernelGGL((agx::k_riccati_blk<NV>), dim3(o->B), dim3(256), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt2, o->d_Kws, o->d_kws,
                             o->d_dx, o->d_w, o->d_state, 0, gsel);
        if constexpr (NV > 16) if (!blk)
          hipLaunchKernelGGL((agx::k_riccati_mfma<NV>), dim3(o->B), dim3(256), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt2, o->d_Kws, o->d_kws,
                             o->d_dx, o->d_w, o->d_state, 0, gsel);
        swept = true;
      }
      if (!swept)
      hipLaunchKernelGGL((agx::k_riccati_big<NV>), dim3(o->B), dim3(256), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt2, o->d_Kws, o->d_kws,
                         o->d_dx, o->d_w, o->d_state, 0, gsel);
      if constexpr (NV > 16) {
        if (o->gains_mfma) {  // the dense feedback-gain GEMM on the matrix cores
          hipLaunchKernelGGL((agx::k_gains_to_u_mfma<NV>), dim3(o->B * o->T), dim3(64), 0, o->stream, o->d_ocp, o->d_aux, o->d_Kws, o->d_Kout,
                             o->d_state);
          HIPCHK(hipGetLastError());
          return 0;
        }
      }
      const long long units = (long long)o->B * o->T * 64;
      hipLaunchKernelGGL((agx::k_gains_to_u_big<NV>), dim3((int)((units + 255) / 256)), dim3(256), 0, o->stream, o->d_ocp, o->d_aux,
                         o->d_Kws, o->d_Kout, o->d_state);
    }
    HIPCHK(hipGetLastError());
    return 0;
  });
}

int reset_state(agx_ocp *o) {
  hipLaunchKernelGGL(agx::k_reset_state, dim3((o->B + 255) / 256), dim3(256), 0, o->stream, o->d_state, o->B, o->d_ndone);
  HIPCHK(hipGetLastError());
  return 0;
}

int prof_mark(agx_ocp *o, int kind, bool start) {
  if (!o->prof) return 0;
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  HIPCHK(hipEventRecord(e, o->stream));
  o->prof_ev.push_back(e);
  if (start) o->prof_kind.push_back(kind);
  return 0;
}
int prof_collect(agx_ocp *o) {
  if (!o->prof || o->prof_ev.empty()) return 0;
  HIPCHK(hipStreamSynchronize(o->stream));
  for (size_t k = 0; k < o->prof_kind.size(); ++k) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, o->prof_ev[2 * k], o->prof_ev[2 * k + 1]));
    o->prof_ms[o->prof_kind[k]] += ms;
    o->prof_n[o->prof_kind[k]] += 1;
  }
  for (hipEvent_t e : o->prof_ev) (void)hipEventDestroy(e);
  o->prof_ev.clear();
  o->prof_kind.clear();
  return 0;
}

// Hand-off of one word from the stream to the host without a copy or a synchronize: a one-thread
// kernel stores value and sequence stamp into mapped pinned memory, the host spins on the stamp.
int publish(agx_ocp *o, int slot_value, int slot_seq, const int *d_value, unsigned long long seq) {
  hipLaunchKernelGGL(agx::k_publish, dim3(1), dim3(1), 0, o->stream, d_value, o->h_ndone_dev + slot_value, o->h_ndone_dev + slot_seq, seq);
  HIPCHK(hipGetLastError());
  return 0;
}
int wait_stamp(agx_ocp *o, int slot_seq, unsigned long long seq) {
  volatile unsigned long long *w = o->h_ndone + slot_seq;
  const auto t_begin = std::chrono::steady_clock::now();
  for (unsigned long spins = 1;; ++spins) {
    if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == seq) return 0;
    if ((spins & 0x3FFF) == 0) {  // every so often make sure the stream is still alive and not stuck
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 120.0)
        return fail("timed out after 120 s waiting for the device (stamp never published)");
      const hipError_t e = hipStreamQuery(o->stream);
      if (e == hipSuccess) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == seq) return 0;
        return fail("stream drained without publishing its stamp");
      }
      if (e != hipErrorNotReady) return fail(std::string("stream error while waiting: ") + hipGetErrorString(e));
    }
    __builtin_ia32_pause();
  }
}

// stream-ordered read of one device int (polled stamp or copy + synchronize)
int read_int(agx_ocp *o, const int *d_value, int slot_value, int slot_seq, int *out) {
  const unsigned long long seq = ++o->seq;
  if (o->poll) {
    if (publish(o, slot_value, slot_seq, d_value, seq)) return -1;
    if (wait_stamp(o, slot_seq, seq)) return -1;
  } else {
    o->h_ndone[slot_value] = 0;  // the 4-byte copy below fills the low half (little endian)
    HIPCHK(hipMemcpyAsync(o->h_ndone + slot_value, d_value, sizeof(int), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
  }
  *out = (int)__atomic_load_n(o->h_ndone + slot_value, __ATOMIC_ACQUIRE);
  return 0;
}

int quorum_count(int B, double q) {
  if (!(q < 1.0)) return B;
  const int n = (int)std::ceil(q * B - 1e-9);
  return n < 1 ? 1 : (n > B ? B : n);
}

// Constraint values / Jacobian rows / violation of every node at (xs, us): 8 lanes per node for serial chains whose rows are
// control limits, state bounds and collision distances (k_con_eval_lj), else one lane per node (k_con_eval).
template <int NV, bool CH>
void launch_con_eval(agx_ocp *o, const double *xs, const double *us, int phase) {
  if constexpr (NV <= 7) {
    const long long nodes = (long long)o->B * (o->T + 1);
    if (CH && o->con_lanes)
      hipLaunchKernelGGL((agx::k_con_eval_lj<NV>), dim3((int)((nodes * 8 + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp, xs, us, o->d_cg,
                         o->d_cjac, o->d_nodestat, o->d_state, phase);
    else
      hipLaunchKernelGGL((agx::k_con_eval<NV, CH>), dim3((int)((nodes + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp, xs, us, o->d_cg,
                         o->d_cjac, o->d_nodestat, o->d_state, phase);
  }
}

// Constrained direction of one SQP iteration (SolverCSQP::computeDirection): the plain LQR pass has
// run (equality-QP initial guess: dx, w); now du, the constraint data and the ADMM loop.
// prefactor: the plain LQR pass has NOT run yet -- it is launched here, in one kernel with the factorisation of the
// augmented Hessians of the first ADMM iteration (k_riccati_lqr_prefactor).
int admm_direction(agx_ocp *o, bool prefactor = false) {
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    if constexpr (NV > 7) {
      // Large models: control-limit rows only (agx_ocp_create), every ADMM iteration factorises (k_riccati_blk on the
      // augmented tile); the node kernels are k_admm_tile_big / k_admm_update_big (agx_big.hpp), norms and rho schedule
      // as for the 7-joint path (k_admm_reduce).
      (void)CH; (void)prefactor;
      if constexpr (NV < 16) return fail("constraints for large models need the blocked sweep (capacity >= 16)");
      else {
      const long long nodes = (long long)o->B * (o->T + 1);
      const int g1 = (int)((nodes + 255) / 256);
      if (launch_step(o, 0, 0, 0, true, false)) return -1;  // du of the initial guess (k_node_kkt_big)
      HIPCHK(hipMemsetAsync(o->d_ndone + 1, 0, sizeof(int), o->stream));
      hipLaunchKernelGGL((agx::k_admm_init<NV>), dim3(g1), dim3(256), 0, o->stream, o->d_ocp, o->d_dx, o->d_cx, o->d_z, o->d_state, o->d_ndone + 1);
      hipLaunchKernelGGL((agx::k_con_eval_wg<NV>), dim3((int)nodes), dim3(256), 0, o->stream, o->d_model, o->d_ocp, o->d_xs, o->d_us, o->d_cg, o->d_cjac,
                         o->d_nodestat, o->d_state, 0);
      HIPCHK(hipGetLastError());
      const int max_qp = o->ho.max_qp;
      for (int iter = 1; iter <= max_qp; ++iter) {
        if (iter == 1 || (iter > 2 && (iter - 1) % agx::kRhoInterval == 0))  // Hessian part: first iteration and after a rho update
          hipLaunchKernelGGL((agx::k_admm_tile_big<NV>), dim3((int)nodes), dim3(256), 0, o->stream, o->d_ocp, o->d_qt, o->d_qt2, o->d_aux, o->d_cx,
                             o->d_du, o->d_cjac, o->d_y, o->d_z, o->d_state);
        hipLaunchKernelGGL((agx::k_riccati_blk<NV>), dim3(o->B), dim3(256), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt2, o->d_Kws, o->d_kws, o->d_dx,
                           o->d_w, o->d_state, 1, 0);
        hipLaunchKernelGGL((agx::k_admm_update_big<NV>), dim3((int)((nodes + 1) / 2)), dim3(64), 0, o->stream, o->d_ocp, o->d_qt, o->d_aux, o->d_dx,
                           o->d_w, o->d_du, o->d_cx, o->d_cg, o->d_cjac, o->d_y, o->d_z, o->d_nodestat, o->d_admmstat, o->d_qt2, o->d_state);
        hipLaunchKernelGGL(agx::k_admm_reduce, dim3(o->B), dim3(128), 0, o->stream, o->d_ocp, o->d_admmstat, o->d_state, iter, o->d_ndone + 1);
        HIPCHK(hipGetLastError());
        if (iter % 4 == 0 || iter == max_qp) {
          int n_conv = 0;
          if (read_int(o, o->d_ndone + 1, 4, 5, &n_conv)) return -1;
          if (n_conv >= quorum_count(o->B, o->quorum_qp)) {
            if (n_conv < o->B && iter < max_qp) {
              hipLaunchKernelGGL(agx::k_admm_cap, dim3((o->B + 255) / 256), dim3(256), 0, o->stream, o->d_state, o->B, iter);
              HIPCHK(hipGetLastError());
            }
            break;
          }
        }
      }
      // the gains the solver reports: those of the last ADMM backward pass, in u-space
      const long long units = (long long)o->B * o->T * 64;
      hipLaunchKernelGGL((agx::k_gains_to_u_big<NV>), dim3((int)((units + 255) / 256)), dim3(256), 0, o->stream, o->d_ocp, o->d_aux, o->d_Kws,
                         o->d_Kout, (const DevState *)nullptr);
      HIPCHK(hipGetLastError());
      return 0;
      }
    } else {
    const long long nodes = (long long)o->B * (o->T + 1);
    const int g8 = (int)((nodes * 8 + 255) / 256), g8b = (int)((nodes * 8 + 127) / 128), g1 = (int)((nodes + 255) / 256);
    if (prefactor) {
      if (!o->d_Kws_lqr) {
        HIPCHK(hipMalloc((void **)&o->d_Kws_lqr, sizeof(double) * (size_t)o->B * o->T * o->nu * o->nx));
        HIPCHK(hipMalloc((void **)&o->d_kws_lqr, sizeof(double) * (size_t)o->B * o->T * o->nu));
      }
      // constraint data and the augmented Hessians need only (xs, us) and rho: before the LQR pass
      launch_con_eval<NV, CH>(o, o->d_xs, o->d_us, 0);
      hipLaunchKernelGGL(agx::k_admm_pre, dim3((o->B + 255) / 256), dim3(256), 0, o->stream, o->d_ocp, o->d_state);
      hipLaunchKernelGGL((agx::k_admm_tile<NV>), dim3(g8b), dim3(128), 0, o->stream, o->d_ocp, o->d_qt, o->d_qt2, o->d_aux, o->d_cx,
                         o->d_du, o->d_cjac, o->d_y, o->d_z, o->d_state, 0);  // its gradient part is rewritten below
      hipLaunchKernelGGL((agx::k_riccati_lqr_prefactor<NV>), dim3(2 * o->B), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_qt2,
                         o->d_aux, o->d_Kws, o->d_kws, o->d_Kws_lqr, o->d_kws_lqr, o->d_dx, o->d_w, o->d_du, o->d_Kout, o->d_state, o->d_fac);
      if (o->admm_segments)
        hipLaunchKernelGGL((agx::k_seg_products<NV>), dim3(o->B * agx::kSeg), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_Kws, o->d_segP,
                           o->d_state);
      HIPCHK(hipGetLastError());
    }
    if (launch_step(o, 0, 0, 0, true, false)) return -1;  // du of the initial guess (k_node_kkt)
    HIPCHK(hipMemsetAsync(o->d_ndone + 1, 0, sizeof(int), o->stream));
    hipLaunchKernelGGL((agx::k_admm_init<NV>), dim3(g1), dim3(256), 0, o->stream, o->d_ocp, o->d_dx, o->d_cx, o->d_z, o->d_state,
                       o->d_ndone + 1);
    if (!prefactor)
      launch_con_eval<NV, CH>(o, o->d_xs, o->d_us, 0);
    HIPCHK(hipGetLastError());
    const int max_qp = o->ho.max_qp;
    // Polls the count of converged QPs; true when the loop ends here (quorum reached: the others are capped at `iter`)
    auto quorum_reached = [&](int iter, bool *stop) -> int {
      int n_conv = 0;
      *stop = false;
      if (read_int(o, o->d_ndone + 1, 4, 5, &n_conv)) return -1;
      if (n_conv >= quorum_count(o->B, o->quorum_qp)) {
        if (n_conv < o->B && iter < max_qp) {  // quorum reached: the others stop here with the iterations they ran
          hipLaunchKernelGGL(agx::k_admm_cap, dim3((o->B + 255) / 256), dim3(256), 0, o->stream, o->d_state, o->B, iter);
          HIPCHK(hipGetLastError());
        }
        *stop = true;
      }
      return 0;
    };
    for (int iter = 1; iter <= max_qp;) {
      // augmented Hessians change at the first iteration and after a rho update (k_admm_reduce decides at
      // multiples of kRhoInterval); in between k_admm_update leaves the next gradient behind
      const bool boundary = iter == 1 || (iter > 2 && (iter - 1) % agx::kRhoInterval == 0);
      if (!boundary && o->admm_loop && o->admm_segments && (o->admm_loop_always || 4 * o->n_unfinished <= o->B)) {
        // gradient-only iterations up to the next rho check in one launch per instance (k_admm_loop); with a quorum < 1 in the
        // chunks of the host's polling schedule, so that which instances are cut does not depend on the workgroups' progress
        int last = ((iter - 1) / agx::kRhoInterval + 1) * agx::kRhoInterval;
        if (o->quorum_qp < 1.0) last = std::min(last, ((iter + 3) / 4) * 4);
        last = std::min(last, max_qp);
        hipLaunchKernelGGL((agx::k_admm_loop<NV>), dim3(o->B), dim3(64 * agx::kSeg), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt, o->d_qt2, o->d_aux,
                           o->d_Kws, o->d_kws, o->d_dx, o->d_w, o->d_du, o->d_cx, o->d_cg, o->d_cjac, o->d_y, o->d_z, o->d_nodestat,
                           o->d_admmstat, o->d_state, o->d_fac, o->d_segP, iter, last, o->d_ndone + 1,
                           o->general ? (const double *)o->d_auxg : (const double *)nullptr);
        HIPCHK(hipGetLastError());
        iter = last + 1;
        bool stop;
        if (quorum_reached(last, &stop)) return -1;
        if (stop) break;
        continue;
      }
      const bool pre = prefactor && iter == 1;  // Hessian and factors of this iteration exist: gradient only
      if (boundary)
        hipLaunchKernelGGL((agx::k_admm_tile<NV>), dim3(g8b), dim3(128), 0, o->stream, o->d_ocp, o->d_qt, o->d_qt2, o->d_aux, o->d_cx,
                           o->d_du, o->d_cjac, o->d_y, o->d_z, o->d_state, pre ? 1 : 0);
      const bool may_refactor = !pre && boundary;
      hipLaunchKernelGGL((agx::k_riccati_admm<NV>), dim3(o->B), dim3(64 * agx::kSeg), 0, o->stream, o->d_ocp, o->d_dt, o->d_qt2, o->d_aux,
                         o->d_Kws, o->d_kws, o->d_dx, o->d_w, o->d_du, o->d_Kout, o->d_state, o->d_fac,
                         o->admm_segments ? (const double *)o->d_segP : (const double *)nullptr, pre ? 1 : 0);
      if (may_refactor && o->admm_segments)  // new gains: the segments' closed-loop products for the gradient-only sweeps that follow
        hipLaunchKernelGGL((agx::k_seg_products<NV>), dim3(o->B * agx::kSeg), dim3(64), 0, o->stream, o->d_ocp, o->d_dt, o->d_Kws, o->d_segP,
                           o->d_state);
      hipLaunchKernelGGL((agx::k_admm_update<NV>), dim3(g8), dim3(256), 0, o->stream, o->d_ocp, o->d_qt, o->d_aux, o->d_dx, o->d_w,
                         o->d_du, o->d_cx, o->d_cg, o->d_cjac, o->d_y, o->d_z, o->d_nodestat, o->d_admmstat, o->d_qt2, o->d_state,
                         o->general ? (const double *)o->d_auxg : (const double *)nullptr);
      hipLaunchKernelGGL(agx::k_admm_reduce, dim3(o->B), dim3(128), 0, o->stream, o->d_ocp, o->d_admmstat, o->d_state, iter,
                         o->d_ndone + 1);
      HIPCHK(hipGetLastError());
      if (iter % 4 == 0 || iter == max_qp) {
        bool stop;
        if (quorum_reached(iter, &stop)) return -1;
        if (stop) break;
      }
      ++iter;
    }
    // the gains the solver reports: those of the last ADMM backward pass, in u-space
    const long long units = (long long)o->B * o->T * 16;
    hipLaunchKernelGGL((agx::k_gains_to_u<NV>), dim3((int)((units + 255) / 256)), dim3(256), 0, o->stream, o->d_ocp, o->d_aux, o->d_Kws,
                       o->d_Kout, o->d_state);
    HIPCHK(hipGetLastError());
    return 0;
    }
  });
}

// agx_ocp_refs_activate: the tile staged by agx_ocp_set_refs_async becomes the solver's tile -- the solver's stream waits
// for the copy (no host wait) and the two device tiles swap roles.
int adopt_pending_refs(agx_ocp *o) {
  if (!o->refs_pending) return 0;
  HIPCHK(hipStreamWaitEvent(o->stream, o->ev_refs, 0));
  std::swap(o->d_ref, o->d_ref_back);
  if (o->refs_pending_frames) std::swap(o->d_frames, o->d_frames_back);
  o->rv.base = o->d_ref;
  o->rv.bstride = (long long)(o->T + 1) * o->stride;
  o->rv.tstride = o->stride;
  o->rv.term_off = 0;
  o->rv.frames = o->refs_pending_frames ? o->d_frames : nullptr;
  o->refs_pending = false;
  return 0;
}
int ensure_copy_stream(agx_ocp *o) {
  if (o->copy_stream) return 0;
  HIPCHK(hipStreamCreateWithFlags(&o->copy_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&o->ev_refs, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&o->ev_snap, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&o->ev_dl, hipEventDisableTiming));
  return 0;
}

// Trial rounds of the line search of SQP iteration `it` (nv <= 7; k_sqp_head has written the first trial iterate):
// derivative pass (+ constraint evaluation) at the trial points, k_sqp_accept, and -- only while the counter of handed-on
// trials grows, i.e. when somebody rejected a step length -- the same again.  The finished-instance count comes back
// under the same stamp, and so does the counter of iterations that ended with every trial rejected: only then does the
// next iteration need a derivative pass of its own (`need_k1`); after an accepted trial the tiles are already there.
int line_search_rounds(agx_ocp *o, int it, int max_iter, bool *need_k1, int *n_done_out) {
  double *xs_t = o->d_xs + (size_t)o->B * (o->T + 1) * o->nx, *us_t = o->d_us + (size_t)o->B * o->T * o->nu;
  for (int round = 0; round < 10; ++round) {
    if (o->prof && round == 0 && it == 0) {  // the kernel the roofline is quoted on, timed alone: running nodes of the trial pass of the first
                                             // iteration (every unfinished instance searches; later iterations serve a few stragglers: not the launch the bytes are counted for)
      if (prof_mark(o, 0, true)) return -1;
      if (launch_calc_qp(o, true, false, 1)) return -1;
      if (prof_mark(o, 0, false)) return -1;
      if (launch_calc_qp(o, false, true, 1)) return -1;
    } else if (launch_calc_qp(o, false, false, 1)) return -1;
    const unsigned long long seq = ++o->seq;
    int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
      constexpr int NV = decltype(NVc)::value;
      constexpr bool CH = decltype(CHc)::value;
      (void)CH;
      if constexpr (NV <= 7) if (o->has_con) {
        launch_con_eval<NV, CH>(o, xs_t, us_t, 1);
      }
      if constexpr (NV > 7) if (o->has_con) {
        const long long nodes = (long long)o->B * (o->T + 1);
        hipLaunchKernelGGL((agx::k_con_eval_wg<NV>), dim3((int)nodes), dim3(256), 0, o->stream, o->d_model, o->d_ocp, xs_t, us_t, o->d_cg, o->d_cjac,
                           o->d_nodestat, o->d_state, 1);
      }
      if (prof_mark(o, 2, true)) return -1;
      hipLaunchKernelGGL((agx::k_sqp_accept<NV>), dim3(o->B), dim3(128), 0, o->stream, o->d_ocp, o->d_xs, o->d_us, o->d_dx, o->d_du, xs_t, us_t,
                         o->d_qt, o->d_nodestat, o->d_state, it, max_iter, o->d_ndone, host_words(o, true, seq));
      HIPCHK(hipGetLastError());
      return prof_mark(o, 2, false);
    });
    if (rc) return rc;
    if (o->poll && o->fold_publish) {
      if (wait_stamp(o, 1, seq)) return -1;  // stored by the last workgroup of k_sqp_accept
    } else if (o->poll) {
      hipLaunchKernelGGL(agx::k_publish3, dim3(1), dim3(1), 0, o->stream, o->d_ndone, o->h_ndone_dev + 0, o->h_ndone_dev + 6, o->h_ndone_dev + 7,
                         o->h_ndone_dev + 1, seq);
      HIPCHK(hipGetLastError());
      if (wait_stamp(o, 1, seq)) return -1;
    } else {
      o->h_ndone[0] = 0; o->h_ndone[6] = 0; o->h_ndone[7] = 0;
      HIPCHK(hipMemcpyAsync(o->h_ndone, o->d_ndone, sizeof(int), hipMemcpyDeviceToHost, o->stream));
      HIPCHK(hipMemcpyAsync(o->h_ndone + 6, o->d_ndone + 3, sizeof(int), hipMemcpyDeviceToHost, o->stream));
      HIPCHK(hipMemcpyAsync(o->h_ndone + 7, o->d_ndone + 4, sizeof(int), hipMemcpyDeviceToHost, o->stream));
      HIPCHK(hipStreamSynchronize(o->stream));
    }
    *n_done_out = (int)__atomic_load_n(o->h_ndone, __ATOMIC_ACQUIRE);
    const unsigned handed = (unsigned)__atomic_load_n(o->h_ndone + 6, __ATOMIC_ACQUIRE);
    const unsigned stale = (unsigned)__atomic_load_n(o->h_ndone + 7, __ATOMIC_ACQUIRE);
    const bool more = handed != o->ls_handed;
    if (stale != o->ls_stale) *need_k1 = true;
    o->ls_handed = handed;
    o->ls_stale = stale;
    if (!more) break;
  }
  return 0;
}

// The SQP loop of SolverCSQP::solve on the resident buffers.
int solve_resident(agx_ocp *o, int max_iter, double max_time, bool prologue_done = false) {
  if (max_iter <= 0) max_iter = 1000;
  o->last_max_iter = max_iter;
  auto t0 = std::chrono::steady_clock::now();
  if (!prologue_done) {  // k_mpc_prologue has reset the state and pinned x0 already
    if (reset_state(o)) return -1;
    hipLaunchKernelGGL(agx::k_pin_x0, dim3((o->B * o->nx + 255) / 256), dim3(256), 0, o->stream, o->d_xs, o->d_x0, o->B, o->T, o->nx);
  }
  // Every iteration after the first finds its tiles already there (the accepted trial of the line search left them); a
  // derivative pass of its own is needed at the start and after an iteration that ended with every step length rejected.
  bool need_k1 = true;
  // the exit fix-up of the gains is launched only if an instance can have finished (or the loop can
  // have ended) in an iteration whose direction sweep was not paired with the gains sweep
  bool need_fixup = false;
  int prev_done = 0;
  for (int it = 0; it < max_iter; ++it) {
    const bool pair = o->speculate && it >= 1 && !o->has_con && o->nv <= 7;  // large models: gains only on exit
    o->n_unfinished = o->B - prev_done;
    // derivative pass: running and terminal nodes in one launch; under agx_ocp_profile the running
    // nodes get their own launch so that the kernel the roofline is quoted on is timed alone
    if (need_k1) {
      if (o->prof && it == 0) {
        if (prof_mark(o, 0, true)) return -1;
        if (launch_calc_qp(o, true, false)) return -1;
        if (prof_mark(o, 0, false)) return -1;
        if (launch_calc_qp(o, false, true)) return -1;
      } else if (launch_calc_qp(o, false, false)) return -1;
    }
    need_k1 = false;
    if (prof_mark(o, 1, true)) return -1;
    // from the second iteration on (where warm-started MPC steps converge) the gains sweep rides along
    if (o->has_con && o->admm_prefactor && o->riccati_mx && o->nv <= 7 && !o->general) {
      if (admm_direction(o, true)) return -1;  // LQR pass inside, next to the ADMM factorisation
    } else {
      if (launch_riccati(o, 1, pair, it)) return -1;
      if (o->has_con && admm_direction(o)) return -1;
    }
    if (prof_mark(o, 1, false)) return -1;
    const bool last = it + 1 == max_iter;
    if (prof_mark(o, 2, true)) return -1;
    const unsigned long long seq_head = ++o->seq;
    if (launch_step(o, it, max_iter, 1, !o->has_con, true, seq_head)) return -1;
    if (prof_mark(o, 2, false)) return -1;
    int n_done = 0;
    // The trial passes overwrite the tiles of this iterate.  Where the loop may end right after this iteration without a
    // paired gains sweep (iteration cap, time limit, quorum), the sweep that yields the reported gains runs first --
    // for the instances the head has not finished (those keep their tiles for the fix-up on exit).
    if (!pair && !o->has_con && (last || max_time > 0.0 || o->quorum_sqp < 1.0)) {
      if (launch_gains(o, 4)) return -1;
    }
    // Small batches: from the second iteration on a warm-started step usually ends at the head, whose last workgroup has
    // handed the finished count over: the host asks before it launches trial passes that would find nobody searching (in the
    // first iteration it does not wait: somebody nearly always searches, and the trial pass starts right behind the head).
    bool searching = true;
    if (o->poll && o->fold_publish && it >= 1) {
      if (wait_stamp(o, 1, seq_head)) return -1;
      n_done = (int)__atomic_load_n(o->h_ndone, __ATOMIC_ACQUIRE);
      searching = n_done < o->B;
    } else if (o->prof || o->no_empty) {  // timing / profiling runs: no launches that find nothing to do
      if (read_int(o, o->d_ndone, 0, 1, &n_done)) return -1;
      searching = n_done < o->B;
    }
    if (searching && line_search_rounds(o, it, max_iter, &need_k1, &n_done)) return -1;
    if (last) { need_fixup = need_fixup || !pair; break; }
    if (!pair && n_done > prev_done) need_fixup = true;
    prev_done = n_done;
    if (n_done >= quorum_count(o->B, o->quorum_sqp)) {
      if (n_done < o->B) o->last_max_iter = it + 1;  // the stragglers stop here: iterations that really ran
      break;
    }
    if (max_time > 0.0) {
      // SolverCSQP max_solve_time (ocp_base_croco.py:70-71): checked once per SQP iteration; unfinished instances
      // report the iterations that really ran
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > max_time) { need_fixup = need_fixup || !pair; o->last_max_iter = it + 1; break; }
    }
  }
  if (!o->has_con && need_fixup && launch_gains(o, 2)) return -1;  // instances whose last direction has no gains sweep yet
  return prof_collect(o);
}

}  // namespace

extern "C" {

const char *agx_last_error(void) { return g_err.c_str(); }

int agx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int agx_row_nref(int kind, int nv) {
  switch (kind) {
    case AGX_RES_STATE: return 2 * nv;
    case AGX_RES_CONTROL: return nv;
    case AGX_RES_CONTROL_GRAV: return 0;
    case AGX_RES_FRAME_PLACEMENT: return 12;
    case AGX_RES_FRAME_TRANSLATION: return 3;
    case AGX_RES_FRAME_ROTATION: return 9;
    case AGX_RES_FRAME_VELOCITY: return 6;
    case AGX_RES_COLLISION: return 0;
  }
  return -1;
}
int agx_row_nr(int kind, int nv) {
  switch (kind) {
    case AGX_RES_STATE: return 2 * nv;
    case AGX_RES_CONTROL: return nv;
    case AGX_RES_CONTROL_GRAV: return nv;
    case AGX_RES_FRAME_PLACEMENT: return 6;
    case AGX_RES_FRAME_TRANSLATION: return 3;
    case AGX_RES_FRAME_ROTATION: return 3;
    case AGX_RES_FRAME_VELOCITY: return 6;
    case AGX_RES_COLLISION: return 1;
  }
  return -1;
}
int agx_ref_stride(const agx_ocp_desc *d, int nv) {
  int s0 = 0, s1 = 0;
  for (int r = 0; r < d->n_running_rows; ++r) s0 += 1 + agx_row_nref(d->running_rows[r].kind, nv) + agx_row_nr(d->running_rows[r].kind, nv);
  for (int r = 0; r < d->n_terminal_rows; ++r) s1 += 1 + agx_row_nref(d->terminal_rows[r].kind, nv) + agx_row_nr(d->terminal_rows[r].kind, nv);
  return std::max(std::max(s0, s1), 1);
}

int agx_model_create(const agx_model_desc *d, agx_model **out) {
  if (!d || !out) return fail("agx_model_create: null argument");
  if (d->nv < 1 || d->nv > AGX_MAX_NV) return fail("agx_model_create: nv out of range");
  if (d->nframes < 0 || d->nframes > AGX_MAX_FRAMES) return fail("agx_model_create: too many frames");
  const int cap = pad_capacity(d->nv);
  if (cap == 0) return fail("agx_model_create: no compiled capacity holds nv = " + std::to_string(d->nv));
  agx_model *m = new agx_model();
  m->nvu = d->nv;
  DevModel &h = m->h;
  std::memset(&h, 0, sizeof(h));
  h.nv = cap;
  h.nframes = d->nframes;
  h.is_chain = 1;
  for (int i = 0; i < d->nv; ++i) {
    h.parent[i] = d->parent[i];
    if (d->parent[i] >= i || d->parent[i] < -1) { delete m; return fail("agx_model_create: parents must precede children"); }
    if (d->parent[i] != i - 1) h.is_chain = 0;
    h.anc[i] = (1u << i) | (d->parent[i] >= 0 ? h.anc[d->parent[i]] : 0u);
    for (int j = 0; j <= i; ++j)
      if ((h.anc[i] >> j) & 1u) h.desc[j] |= (1u << i);
    std::memcpy(h.placement[i], d->placement + 12 * i, sizeof(double) * 12);
    std::memcpy(h.axis[i], d->axis + 3 * i, sizeof(double) * 3);
    h.mass[i] = d->mass[i];
    std::memcpy(h.com[i], d->com + 3 * i, sizeof(double) * 3);
    std::memcpy(h.inertia[i], d->inertia + 9 * i, sizeof(double) * 9);
    h.armature[i] = d->armature ? d->armature[i] : 0.0;
  }
  // pad joints (robot_model.py:231-257 takes any URDF and any set of locked joints; the kernels come in a few capacities): massless,
  // unit armature, identity placement -- appended to a serial chain (the eight-lane kernel needs one), children of the universe in a tree
  for (int i = d->nv; i < cap; ++i) {
    h.parent[i] = h.is_chain ? i - 1 : -1;
    h.anc[i] = (1u << i) | (h.parent[i] >= 0 ? h.anc[h.parent[i]] : 0u);
    for (int j = 0; j <= i; ++j)
      if ((h.anc[i] >> j) & 1u) h.desc[j] |= (1u << i);
    h.placement[i][0] = h.placement[i][4] = h.placement[i][8] = 1.0;
    h.axis[i][2] = 1.0;
    h.armature[i] = 1.0;
  }
  h.gravity[0] = d->gravity ? d->gravity[0] : 0.0;
  h.gravity[1] = d->gravity ? d->gravity[1] : 0.0;
  h.gravity[2] = d->gravity ? d->gravity[2] : -9.81;
  for (int f = 0; f < d->nframes; ++f) {
    h.frame_parent[f] = d->frame_parent[f];
    if (d->frame_parent[f] >= d->nv) { delete m; return fail("agx_model_create: bad frame parent"); }
    std::memcpy(h.frame_placement[f], d->frame_placement + 12 * f, sizeof(double) * 12);
    h.frame_radius[f] = d->frame_radius ? d->frame_radius[f] : 0.0;
    h.frame_halflen[f] = d->frame_halflen ? d->frame_halflen[f] : 0.0;
    for (int e = 0; e < 3; ++e) h.frame_box[f][e] = d->frame_box ? d->frame_box[3 * f + e] : 0.0;
    if (h.frame_box[f][0] > 0.0 && !(h.frame_box[f][1] > 0.0 && h.frame_box[f][2] > 0.0)) {
      delete m;
      return fail("agx_model_create: box half extents must all be positive");
    }
  }
  *out = m;
  return 0;
}
void agx_model_destroy(agx_model *m) { delete m; }

int agx_ocp_create(const agx_model *m, const agx_ocp_desc *d, int batch, int device, agx_ocp **out) {
  if (!m || !d || !out) return fail("agx_ocp_create: null argument");
  if (batch < 1) return fail("agx_ocp_create: batch must be positive");
  if (d->horizon < 1) return fail("agx_ocp_create: horizon must be positive");
  if (d->n_running_rows > AGX_MAX_ROWS || d->n_terminal_rows > AGX_MAX_ROWS) return fail("agx_ocp_create: too many cost rows");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("agx_ocp_create: no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail("agx_ocp_create: bad device index");
  for (int r = 0; r < d->n_running_rows + d->n_terminal_rows; ++r) {
    const agx_cost_row &row = r < d->n_running_rows ? d->running_rows[r] : d->terminal_rows[r - d->n_running_rows];
    switch (row.kind) {
      case AGX_RES_STATE: case AGX_RES_CONTROL: case AGX_RES_FRAME_PLACEMENT: case AGX_RES_FRAME_TRANSLATION: case AGX_RES_FRAME_ROTATION:
      case AGX_RES_COLLISION: break;
      case AGX_RES_CONTROL_GRAV: case AGX_RES_FRAME_VELOCITY:
        if (m->h.nv > 7) return fail("agx_ocp_create: ControlGrav / FrameVelocity rows are implemented for nv <= 7");
        if (row.kind == AGX_RES_FRAME_VELOCITY && (row.frame < 0 || row.frame >= m->h.nframes || row.frame_b < 0 || row.frame_b > 2))
          return fail("agx_ocp_create: FrameVelocity row needs a valid frame and a reference frame 0 (WORLD), 1 (LOCAL) or 2 (LOCAL_WORLD_ALIGNED)");
        break;
      default: return fail("agx_ocp_create: residual kind " + std::to_string(row.kind) + " is not implemented on the HIP path yet");
    }
    if (row.activation != AGX_ACT_WEIGHTED_QUAD && (row.kind == AGX_RES_CONTROL_GRAV || row.kind == AGX_RES_FRAME_VELOCITY))
      return fail("agx_ocp_create: ActivationModelExp / QuadExp are not implemented for ControlGrav / FrameVelocity residuals");
    if (row.activation != AGX_ACT_WEIGHTED_QUAD && !(row.alpha > 0.0)) return fail("agx_ocp_create: activation alpha must be positive");
    if ((row.kind == AGX_RES_FRAME_PLACEMENT || row.kind == AGX_RES_FRAME_TRANSLATION || row.kind == AGX_RES_FRAME_ROTATION) &&
        (row.frame < 0 || row.frame >= m->h.nframes))
      return fail("agx_ocp_create: frame id out of range");
    if (row.kind == AGX_RES_COLLISION) {
      if (row.frame < 0 || row.frame >= m->h.nframes || row.frame_b < 0 || row.frame_b >= m->h.nframes)
        return fail("agx_ocp_create: collision pair refers to a geometry frame out of range");
      if (!agx::frame_has_geometry(m->h, row.frame) || !agx::frame_has_geometry(m->h, row.frame_b))
        return fail("agx_ocp_create: collision pair refers to a frame without geometry (radius 0, no box)");
      if (agx::frame_is_box(m->h, row.frame) && agx::frame_is_box(m->h, row.frame_b))
        return fail("agx_ocp_create: box / box collision pairs are not supported");
    }
  }
  agx_ocp *o = new agx_ocp();
  o->hm = m->h;
  o->nv = m->h.nv; o->nx = 2 * o->nv; o->nu = o->nv;
  o->chain = m->h.is_chain != 0;
  if (const char *e = getenv("AGX_K1_LANES")) o->k1_lanes = (e[0] != '0');
  if (const char *e = getenv("AGX_SPECULATE_GAINS")) o->speculate = (e[0] != '0');
  if (const char *e = getenv("AGX_GAINS_MFMA")) o->gains_mfma = (e[0] != '0');
  if (const char *e = getenv("AGX_RICCATI_MFMA")) o->riccati_mfma = (e[0] != '0');
  if (const char *e = getenv("AGX_RICCATI_BLK")) o->riccati_blk = (e[0] != '0');
  if (const char *e = getenv("AGX_RICCATI_MX")) o->riccati_mx = (e[0] != '0');
  if (const char *e = getenv("AGX_FUSED_KKT")) o->fuse_kkt = (e[0] != '0');
  if (const char *e = getenv("AGX_NO_EMPTY_LAUNCHES")) o->no_empty = (e[0] != '0');
  o->fold_publish = batch <= 204;
  if (const char *e = getenv("AGX_FOLD_PUBLISH")) o->fold_publish = (e[0] != '0');
  if (const char *e = getenv("AGX_ADMM_LOOP")) { o->admm_loop = (e[0] != '0'); o->admm_loop_always = (e[0] == '2'); }
  if (const char *e = getenv("AGX_K1_FUSED")) o->k1_fused = (e[0] != '0');
  o->T = d->horizon; o->B = batch; o->device = device;
  {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) o->n_cu = n;
  }
  o->tile = AGX_TILE_DOUBLES(o->nv);
  const int ld = o->nv <= 8 ? 8 : 32;
  o->qt_size = 6 * o->nv * ld + 5 * ld + 8;  // QT<NV>::SIZE
  o->aux_size = 4 * o->nv * ld + 3 * ld;     // AUX<NV>::SIZE
  o->stride = agx_ref_stride(d, o->nv);
  o->dt.assign(d->dt, d->dt + d->horizon);
  std::memset(&o->ho, 0, sizeof(o->ho));
  o->ho.T = o->T; o->ho.B = o->B; o->ho.stride = o->stride;
  fill_rows(d->running_rows, d->n_running_rows, o->nv, m->nvu, o->ho.rows[0]);
  fill_rows(d->terminal_rows, d->n_terminal_rows, o->nv, m->nvu, o->ho.rows[1]);
  o->ho.tol = d->termination_tolerance;
  o->ho.mu_dyn = d->mu_dynamic;
  o->ho.mu_con = d->mu_constraint;
  o->nvu = m->nvu;
  o->padded = o->nvu != o->nv;
  o->stride_u = agx_ref_stride(d, o->nvu);
  if (o->padded) {
    // reference-tile maps: row r of a layout sits at off_u[r] in the caller's tile, at off[r] in the internal one; the joint
    // blocks of State / Control references and activation weights grow from nvu to nv entries (pad weights 1: the pad
    // residuals are identically zero, the weight only keeps the pad block of the Hessians well conditioned)
    for (int lay = 0; lay < 2; ++lay) {
      const DevRows &R = o->ho.rows[lay];
      o->refmap[lay].assign(o->stride, -1);
      o->reffill[lay].assign(o->stride, 0.0);
      int off_u = 0;
      for (int r = 0; r < R.n; ++r) {
        const int kind = R.kind[r], nref_u = agx_row_nref(kind, o->nvu), nr_u = agx_row_nr(kind, o->nvu);
        o->refmap[lay][R.off[r]] = off_u;
        for (int k = 0; k < R.nref[r]; ++k) {
          const int ku = user_component(kind, k, o->nv, o->nvu, true);
          o->refmap[lay][R.off[r] + 1 + k] = ku >= 0 ? off_u + 1 + ku : -1;
        }
        for (int k = 0; k < R.nr[r]; ++k) {
          const int ku = user_component(kind, k, o->nv, o->nvu, false);
          o->refmap[lay][R.off[r] + 1 + R.nref[r] + k] = ku >= 0 ? off_u + 1 + nref_u + ku : -1;
          if (ku < 0) o->reffill[lay][R.off[r] + 1 + R.nref[r] + k] = 1.0;
        }
        off_u += 1 + nref_u + nr_u;
      }
    }
  }
  if (fill_cons(d->running_constraints, d->n_running_constraints, o->nv, o->nvu, m->h, o->ho.cons[0], false) ||
      fill_cons(d->terminal_constraints, d->n_terminal_constraints, o->nv, o->nvu, m->h, o->ho.cons[1], true)) { delete o; return -1; }
  for (int lay = 0; lay < 2; ++lay)
    for (int r = 0; r < o->ho.cons[lay].n; ++r) {
      const int k = o->ho.cons[lay].kind[r];
      if (k != AGX_RES_CONTROL && k != AGX_RES_STATE && k != AGX_RES_COLLISION) o->con_lanes = false;
    }
  if (const char *e = getenv("AGX_CON_LANES")) o->con_lanes = o->con_lanes && (e[0] != '0');
  o->has_con = o->ho.cons[0].nc + o->ho.cons[1].nc > 0;
  o->general = o->ho.rows[0].general || o->ho.rows[1].general;
  o->ho.has_con = o->has_con ? 1 : 0;
  o->ho.max_qp = d->max_qp_iters > 0 ? d->max_qp_iters : 1000;
  o->ho.eps_abs = d->eps_abs;
  o->ho.eps_rel = d->eps_rel;
  o->ho.use_filter = d->use_filter_line_search ? 1 : 0;
  // (the filter test lives in k_sqp_accept, which serves every model size and every kind of cost row)
  if (o->has_con && o->nv > 7) {  // large models: ConstraintModelControlLimit only (agx_big.hpp)
    for (int lay = 0; lay < 2; ++lay)
      for (int r = 0; r < o->ho.cons[lay].n; ++r)
        if (o->ho.cons[lay].kind[r] == AGX_RES_FRAME_VELOCITY || o->ho.cons[lay].kind[r] == AGX_RES_CONTROL_GRAV) {
          delete o;
          return fail("agx_ocp_create: FrameVelocity / ControlGrav constraints are implemented for models of at most 7 joints (after padding)");
        }
  }
  {
    auto n_frame_rows = [](const DevRows &r) {
      int n = 0;
      for (int i = 0; i < r.n; ++i)
        n += (r.kind[i] == AGX_RES_FRAME_PLACEMENT || r.kind[i] == AGX_RES_FRAME_TRANSLATION || r.kind[i] == AGX_RES_FRAME_ROTATION);
      return n;
    };
    auto n_collision_rows = [](const DevRows &r) {
      int n = 0;
      for (int i = 0; i < r.n; ++i) n += (r.kind[i] == AGX_RES_COLLISION);
      return n;
    };
    // Exp / QuadExp activations on vector residuals (ocp_croco_generic.py:118-131) are evaluated by the one-lane-per-node kernel
    auto n_vector_exp_rows = [&]() {
      int n = 0;
      for (int lay = 0; lay < 2; ++lay)
        for (int i = 0; i < o->ho.rows[lay].n; ++i)
          if (o->ho.rows[lay].active[i] && o->ho.rows[lay].act[i] != AGX_ACT_WEIGHTED_QUAD && o->ho.rows[lay].kind[i] != AGX_RES_COLLISION) ++n;
      return n;
    };
    // at most one collision cost row per node type, capsule / sphere / box pair, serial chain: the COLL variant
    o->lanes_coll = n_collision_rows(o->ho.rows[0]) + n_collision_rows(o->ho.rows[1]) > 0;
    o->lanes_ok = o->stride <= agx::kLjRef && n_frame_rows(o->ho.rows[0]) <= 2 && n_frame_rows(o->ho.rows[1]) <= 2 &&
                  n_collision_rows(o->ho.rows[0]) <= 1 && n_collision_rows(o->ho.rows[1]) <= 1 && !o->general && !n_vector_exp_rows();
  }
  if (o->nv > 8) {
    // the workgroup-per-node kernels stage the reference tile and the dense residual rows of a node in LDS
    auto n_dense = [](const DevRows &r) {
      int n = 0;
      for (int i = 0; i < r.n; ++i)
        if (r.active[i])
          n += r.kind[i] == AGX_RES_FRAME_PLACEMENT ? 6 : ((r.kind[i] == AGX_RES_FRAME_TRANSLATION || r.kind[i] == AGX_RES_FRAME_ROTATION) ? 3 : (r.kind[i] == AGX_RES_COLLISION ? 1 : 0));
      return n;
    };
    if (o->stride > agx::kWgRef) { delete o; return fail("agx_ocp_create: reference tile of " + std::to_string(o->stride) + " doubles per node exceeds the large-model limit of " + std::to_string(agx::kWgRef)); }
    if (n_dense(o->ho.rows[0]) > agx::kWgJ || n_dense(o->ho.rows[1]) > agx::kWgJ) { delete o; return fail("agx_ocp_create: more than 16 frame / collision residual components per node (large models)"); }
  }
  // two-level sweep for small batches of unconstrained problems on the MFMA-layout kernel.  Only with Gauss-Newton Hessians of
  // quadratic activations: then Hww = M (Luu + preg) M is positive definite at every node and no sweep -- neither the ordinary one
  // nor a segment's zero-terminal one, whose pivots are the smaller ones -- can meet a non-positive pivot.  An Exp / QuadExp cost
  // row has negative curvature inside its bell: such problems keep the one-wave sweep and its breakdown handling.
  bool convex_rows = true;
  for (int lay = 0; lay < 2; ++lay)
    for (int r = 0; r < o->ho.rows[lay].n; ++r)
      if (o->ho.rows[lay].active[r] && o->ho.rows[lay].act[r] != AGX_ACT_WEIGHTED_QUAD) convex_rows = false;
  if (o->nv <= 7 && o->riccati_mx && !o->has_con && !o->fuse_kkt && convex_rows) {
    // ten segments while the two sweeps of a paired launch (2 B S waves of 254 VGPRs) fit the 2 048 wave slots of the chip
    // at two per SIMD; fewer above that; below five segments the 2.4 x arithmetic is not paid back (measured, B = 256: none)
    int S = std::min(10, 1024 / o->B);
    if (S < 5) S = 0;
    if (const char *e = getenv("AGX_MX2_SEGMENTS")) S = atoi(e);
    if (S > agx::kMx2MaxSeg) S = agx::kMx2MaxSeg;
    while (S >= 2 && o->T / S < 4) --S;  // segments of at least four nodes
    o->mx2_S = S >= 2 ? S : 0;
  }
  // probe that a kernel instantiation exists
  if (dispatch(o->nv, o->chain, [](auto, auto) -> int { return 0; })) { delete o; return -1; }
  if (set_device(o)) { delete o; return -1; }
  const size_t B = o->B, T = o->T, nx = o->nx, nu = o->nu;
#define ALLOC(ptr, count)                                                                                   \
  do {                                                                                                      \
    hipError_t e_ = hipMalloc((void **)&(ptr), sizeof(*(ptr)) * (count));                                  \
    if (e_ != hipSuccess) { agx_ocp_destroy(o); return fail(std::string("hipMalloc " #ptr ": ") + hipGetErrorString(e_)); } \
    (void)hipMemset((ptr), 0, sizeof(*(ptr)) * (count));                                                   \
  } while (0)
  ALLOC(o->d_model, 1);
  ALLOC(o->d_ocp, 1);
  ALLOC(o->d_dt, T);
  ALLOC(o->d_xs, 2 * B * (T + 1) * nx);  // second half: shift staging
  ALLOC(o->d_us, 2 * B * T * nu);
  ALLOC(o->d_x0, B * nx);
  ALLOC(o->d_Kws, B * T * nu * nx);
  ALLOC(o->d_kws, B * T * nu);
  ALLOC(o->d_Kout, B * T * nu * nx);
  ALLOC(o->d_dx, B * (T + 1) * nx);
  ALLOC(o->d_du, B * T * nu);
  ALLOC(o->d_w, B * T * nu);
  ALLOC(o->d_nodestat, B * (T + 1) * 4);
  ALLOC(o->d_qt, B * (T + 1) * (size_t)o->qt_size);
  ALLOC(o->d_aux, B * (T + 1) * (size_t)o->aux_size);
  ALLOC(o->d_ref, B * (T + 1) * (size_t)o->stride);
  ALLOC(o->d_frames, B * (T + 1) * AGX_MAX_ROWS);
  ALLOC(o->d_state, B);
  ALLOC(o->d_ndone, 8);  // [0] finished instances, [1] instances whose ADMM loop has ended, [2] instances in the split line search
  if (o->nv > 8) {
    for (int t = 0; t < o->T; ++t)
      if (o->dt[t] != o->dt[0]) o->shift_nodes.push_back(t);
    if (!o->shift_nodes.empty()) {
      ALLOC(o->d_shift_nodes, o->shift_nodes.size());
      (void)hipMemcpy(o->d_shift_nodes, o->shift_nodes.data(), sizeof(int) * o->shift_nodes.size(), hipMemcpyHostToDevice);
    }
  }
  if (o->general) ALLOC(o->d_auxg, B * (T + 1) * (size_t)(3 * o->nv * 8));
  if (o->has_con) {
    if (!o->d_qt2) ALLOC(o->d_qt2, B * (T + 1) * (size_t)o->qt_size);
    ALLOC(o->d_cg, B * (T + 1) * AGX_MAX_NC);
    ALLOC(o->d_cjac, B * (T + 1) * AGX_MAX_DENSE * 32);  // 24 per row up to 7 joints (q | v | u, 8 each), 32 above (q only)
    ALLOC(o->d_y, B * (T + 1) * AGX_MAX_NC);
    ALLOC(o->d_z, B * (T + 1) * AGX_MAX_NC);
    ALLOC(o->d_cx, B * (T + 1) * nx);
    ALLOC(o->d_admmstat, B * (T + 1) * 4);
    ALLOC(o->d_fac, B * T * 192);
    ALLOC(o->d_segP, B * agx::kSeg * 256);
    if (const char *e = getenv("AGX_ADMM_SEGMENTS")) o->admm_segments = (e[0] != '0');
    if (const char *e = getenv("AGX_ADMM_PREFACTOR")) o->admm_prefactor = (e[0] != '0');
  }
#undef ALLOC
  // the polled hand-off needs FINE-GRAINED host memory (the stamp must not overtake the data it guards): ask for it
  // explicitly; when the runtime cannot give it, fall back to stream-ordered copies + synchronize
  if (hipHostMalloc((void **)&o->h_ndone, 8 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    o->poll = false;
    if (hipHostMalloc((void **)&o->h_ndone, 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { agx_ocp_destroy(o); return fail("hipHostMalloc failed"); }
  }
  std::memset(o->h_ndone, 0xff, 8 * sizeof(unsigned long long));
  if (o->poll && hipHostGetDevicePointer((void **)&o->h_ndone_dev, o->h_ndone, 0) != hipSuccess) o->poll = false;
  if (const char *e = getenv("AGX_HOST_POLL")) o->poll = o->poll && (e[0] != '0');
  if (hipMemcpy(o->d_model, &o->hm, sizeof(DevModel), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(o->d_ocp, &o->ho, sizeof(DevOcp), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(o->d_dt, o->dt.data(), sizeof(double) * T, hipMemcpyHostToDevice) != hipSuccess) {
    agx_ocp_destroy(o);
    return fail("agx_ocp_create: upload of the problem description failed");
  }
  if (hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking) != hipSuccess) { agx_ocp_destroy(o); return fail("hipStreamCreate failed"); }
  o->own_stream = true;
  (void)hipEventCreate(&o->ev0);
  (void)hipEventCreate(&o->ev1);
  (void)hipEventCreateWithFlags(&o->ev_done, hipEventDisableTiming);
  o->rv.base = o->d_ref;
  o->rv.bstride = (long long)(T + 1) * o->stride;
  o->rv.tstride = o->stride;
  o->rv.term_off = 0;
  o->rv.frames = nullptr;
  if (reset_state(o) || hipStreamSynchronize(o->stream) != hipSuccess) { agx_ocp_destroy(o); return fail("agx_ocp_create: state reset failed"); }
  *out = o;
  return 0;
}

void agx_ocp_destroy(agx_ocp *o) {
  if (!o) return;
  (void)hipSetDevice(o->device);
  if (o->stream) (void)hipStreamSynchronize(o->stream);
  if (o->copy_stream) (void)hipStreamSynchronize(o->copy_stream);
  void *ptrs[] = {o->d_mx2_elem, o->d_mx2_bnd, o->d_mx2_cl, o->d_ref_back, o->d_frames_back, o->d_snap, o->d_model, o->d_ocp, o->d_dt, o->d_xs, o->d_us, o->d_x0, o->d_tiles, o->d_Kws, o->d_kws, o->d_Kout, o->d_dx,
                  o->d_du, o->d_ref, o->d_frames, o->d_state, o->d_ndone, o->d_scratch, o->d_traj, o->d_pts, o->d_sine, o->d_qt, o->d_aux, o->d_w, o->d_nodestat,
                  o->d_qt2, o->d_cg, o->d_cjac, o->d_y, o->d_z, o->d_cx, o->d_admmstat, o->d_fac, o->d_hidx, o->d_auxg, o->d_shift_nodes, o->d_segP, o->d_Kws_lqr, o->d_kws_lqr};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (o->h_ndone) (void)hipHostFree(o->h_ndone);
  if (o->ev0) (void)hipEventDestroy(o->ev0);
  if (o->ev1) (void)hipEventDestroy(o->ev1);
  if (o->ev_done) (void)hipEventDestroy(o->ev_done);
  if (o->ev_refs) (void)hipEventDestroy(o->ev_refs);
  if (o->ev_snap) (void)hipEventDestroy(o->ev_snap);
  if (o->ev_dl) (void)hipEventDestroy(o->ev_dl);
  if (o->copy_stream) (void)hipStreamDestroy(o->copy_stream);
  if (o->d_first && !o->poll) (void)hipFree(o->d_first);
  if (o->h_first) (void)hipHostFree(o->h_first);
  if (o->own_stream && o->stream) (void)hipStreamDestroy(o->stream);
  delete o;
}

int agx_ocp_set_stream(agx_ocp *o, void *hip_stream) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  if (o->stream) HIPCHK(hipStreamSynchronize(o->stream));
  if (hip_stream) {
    if (o->own_stream && o->stream) (void)hipStreamDestroy(o->stream);
    o->stream = (hipStream_t)hip_stream;
    o->own_stream = false;
  } else if (!o->own_stream) {
    HIPCHK(hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking));
    o->own_stream = true;
  }
  return 0;
}

int agx_ocp_sync(agx_ocp *o) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_set_quorum(agx_ocp *o, double sqp_fraction, double qp_fraction) {
  if (!o) return fail("null handle");
  if (!(sqp_fraction > 0.0 && sqp_fraction <= 1.0) || !(qp_fraction > 0.0 && qp_fraction <= 1.0))
    return fail("agx_ocp_set_quorum: fractions must be in (0, 1]");
  o->quorum_sqp = sqp_fraction;
  o->quorum_qp = qp_fraction;
  return 0;
}

int agx_ocp_reset_duals(agx_ocp *o) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  if (o->has_con) HIPCHK(hipMemsetAsync(o->d_y, 0, sizeof(double) * (size_t)o->B * (o->T + 1) * AGX_MAX_NC, o->stream));
  hipLaunchKernelGGL(agx::k_reset_rho, dim3((o->B + 255) / 256), dim3(256), 0, o->stream, o->d_state, o->B);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_set_geom_placement(agx_ocp *o, int frame, const double *se3) {
  if (!o || !se3) return fail("agx_ocp_set_geom_placement: null argument");
  if (frame < 0 || frame >= o->hm.nframes) return fail("agx_ocp_set_geom_placement: frame out of range");
  if (set_device(o)) return -1;
  std::memcpy(o->hm.frame_placement[frame], se3, sizeof(double) * 12);
  // in-stream update of the one placement inside the resident model
  HIPCHK(hipMemcpyAsync((char *)o->d_model + offsetof(DevModel, frame_placement) + sizeof(double) * 12 * frame, o->hm.frame_placement[frame],
                        sizeof(double) * 12, hipMemcpyHostToDevice, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_set_refs(agx_ocp *o, const double *ref_tile, const int32_t *frame_ids) {
  if (!o || !ref_tile) return fail("agx_ocp_set_refs: null argument");
  if (set_device(o)) return -1;
  if (o->refs_pending) { HIPCHK(hipEventSynchronize(o->ev_refs)); o->refs_pending = false; }  // superseded
  const size_t n = (size_t)o->B * (o->T + 1);
  if (o->padded) {  // the caller's tile is laid out for nvu joints
    o->stage.resize(n * o->stride);
    pad_ref_tile(o, o->stage.data(), ref_tile, o->B, o->T);
    ref_tile = o->stage.data();
  }
  HIPCHK(hipMemcpyAsync(o->d_ref, ref_tile, sizeof(double) * n * o->stride, hipMemcpyHostToDevice, o->stream));
  if (frame_ids) HIPCHK(hipMemcpyAsync(o->d_frames, frame_ids, sizeof(int) * n * AGX_MAX_ROWS, hipMemcpyHostToDevice, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));  // the caller may reuse its buffers
  o->rv.base = o->d_ref;
  o->rv.bstride = (long long)(o->T + 1) * o->stride;
  o->rv.tstride = o->stride;
  o->rv.term_off = 0;
  o->rv.frames = frame_ids ? o->d_frames : nullptr;
  return 0;
}

int agx_ocp_set_refs_device(agx_ocp *o, const double *d_ref_tile, const int32_t *d_frame_ids, int adopt) {
  if (!o || !d_ref_tile) return fail("agx_ocp_set_refs_device: null argument");
  if (set_device(o)) return -1;
  const size_t n = (size_t)o->B * (o->T + 1);
  if (o->padded) {  // a device tile in the caller's layout: through the host (padded models are not the fast path)
    std::vector<double> tmp(n * o->stride_u);
    HIPCHK(hipMemcpyAsync(tmp.data(), d_ref_tile, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    o->stage.resize(n * o->stride);
    pad_ref_tile(o, o->stage.data(), tmp.data(), o->B, o->T);
    HIPCHK(hipMemcpyAsync(o->d_ref, o->stage.data(), sizeof(double) * n * o->stride, hipMemcpyHostToDevice, o->stream));
    if (d_frame_ids) HIPCHK(hipMemcpyAsync(o->d_frames, d_frame_ids, sizeof(int) * n * AGX_MAX_ROWS, hipMemcpyDeviceToDevice, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    o->rv.base = o->d_ref;
    o->rv.frames = d_frame_ids ? o->d_frames : nullptr;
  } else if (adopt) {
    o->rv.base = d_ref_tile;
    o->rv.frames = d_frame_ids;
  } else {
    HIPCHK(hipMemcpyAsync(o->d_ref, d_ref_tile, sizeof(double) * n * o->stride, hipMemcpyDeviceToDevice, o->stream));
    if (d_frame_ids) HIPCHK(hipMemcpyAsync(o->d_frames, d_frame_ids, sizeof(int) * n * AGX_MAX_ROWS, hipMemcpyDeviceToDevice, o->stream));
    o->rv.base = o->d_ref;
    o->rv.frames = d_frame_ids ? o->d_frames : nullptr;
  }
  o->rv.bstride = (long long)(o->T + 1) * o->stride;
  o->rv.tstride = o->stride;
  o->rv.term_off = 0;
  return 0;
}

int agx_host_alloc(size_t bytes, void **out) {
  if (!out || bytes == 0) return fail("agx_host_alloc: bad argument");
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return 0;
}
int agx_host_free(void *p) {
  if (p) HIPCHK(hipHostFree(p));
  return 0;
}

int agx_ocp_set_refs_async(agx_ocp *o, const double *ref_tile, const int32_t *frame_ids) {
  if (!o || !ref_tile) return fail("agx_ocp_set_refs_async: null argument");
  if (set_device(o)) return -1;
  if (ensure_copy_stream(o)) return -1;
  const size_t n = (size_t)o->B * (o->T + 1);
  if (!o->d_ref_back) HIPCHK(hipMalloc((void **)&o->d_ref_back, sizeof(double) * n * o->stride));
  if (frame_ids && !o->d_frames_back) HIPCHK(hipMalloc((void **)&o->d_frames_back, sizeof(int) * n * AGX_MAX_ROWS));
  if (o->refs_pending) HIPCHK(hipEventSynchronize(o->ev_refs));  // a second upload before any solve: replaces the first
  // the back tile is free: the solve that read it has returned (solves are synchronous to the host)
  if (o->padded) {  // repacked through the handle's staging vector: the copy has to finish before this call returns
    o->stage.resize(n * o->stride);
    pad_ref_tile(o, o->stage.data(), ref_tile, o->B, o->T);
    ref_tile = o->stage.data();
  }
  HIPCHK(hipMemcpyAsync(o->d_ref_back, ref_tile, sizeof(double) * n * o->stride, hipMemcpyHostToDevice, o->copy_stream));
  if (frame_ids) HIPCHK(hipMemcpyAsync(o->d_frames_back, frame_ids, sizeof(int) * n * AGX_MAX_ROWS, hipMemcpyHostToDevice, o->copy_stream));
  HIPCHK(hipEventRecord(o->ev_refs, o->copy_stream));
  if (o->padded) HIPCHK(hipStreamSynchronize(o->copy_stream));
  o->refs_pending = true;
  o->refs_pending_frames = frame_ids != nullptr;
  return 0;
}
int agx_ocp_refs_wait(agx_ocp *o) {
  if (!o) return fail("null handle");
  if (o->refs_pending) HIPCHK(hipEventSynchronize(o->ev_refs));
  return 0;
}
int agx_ocp_refs_activate(agx_ocp *o) {
  if (!o) return fail("null handle");
  if (!o->refs_pending) return fail("agx_ocp_refs_activate: no staged reference tile (agx_ocp_set_refs_async)");
  if (set_device(o)) return -1;
  return adopt_pending_refs(o);
}

int agx_ocp_download_async(agx_ocp *o, double *xs, double *us, double *K) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  if (ensure_copy_stream(o)) return -1;
  const size_t B = o->B, T = o->T, nx = o->nx, nu = o->nu;
  if (o->padded) {  // repacked on the host: the synchronous route
    if (xs && down(o, xs, o->d_xs, B * (T + 1), 2)) return -1;
    if (us && down(o, us, o->d_us, B * T, 1)) return -1;
    if (K && down_gains(o, K, o->d_Kout, B * T)) return -1;
    HIPCHK(hipStreamSynchronize(o->stream));
    return 0;
  }
  const size_t n_xs = B * (T + 1) * nx, n_us = B * T * nu, n_K = B * T * nu * nx;
  if (!o->d_snap) HIPCHK(hipMalloc((void **)&o->d_snap, sizeof(double) * (n_xs + n_us + n_K)));
  if (o->dl_pending) HIPCHK(hipEventSynchronize(o->ev_dl));  // the previous download still reads the snapshot
  // snapshot in the solver's stream (device to device, ~0.1 ms for 100 MB), then the copy engine drains it while the
  // solver goes on with the next step
  if (xs) HIPCHK(hipMemcpyAsync(o->d_snap, o->d_xs, sizeof(double) * n_xs, hipMemcpyDeviceToDevice, o->stream));
  if (us) HIPCHK(hipMemcpyAsync(o->d_snap + n_xs, o->d_us, sizeof(double) * n_us, hipMemcpyDeviceToDevice, o->stream));
  if (K) HIPCHK(hipMemcpyAsync(o->d_snap + n_xs + n_us, o->d_Kout, sizeof(double) * n_K, hipMemcpyDeviceToDevice, o->stream));
  HIPCHK(hipEventRecord(o->ev_snap, o->stream));
  HIPCHK(hipStreamWaitEvent(o->copy_stream, o->ev_snap, 0));
  if (xs && us && K && us == xs + n_xs && K == us + n_us) {  // one contiguous destination (xs | us | K): one transfer
    HIPCHK(hipMemcpyAsync(xs, o->d_snap, sizeof(double) * (n_xs + n_us + n_K), hipMemcpyDeviceToHost, o->copy_stream));
  } else {
    if (xs) HIPCHK(hipMemcpyAsync(xs, o->d_snap, sizeof(double) * n_xs, hipMemcpyDeviceToHost, o->copy_stream));
    if (us) HIPCHK(hipMemcpyAsync(us, o->d_snap + n_xs, sizeof(double) * n_us, hipMemcpyDeviceToHost, o->copy_stream));
    if (K) HIPCHK(hipMemcpyAsync(K, o->d_snap + n_xs + n_us, sizeof(double) * n_K, hipMemcpyDeviceToHost, o->copy_stream));
  }
  HIPCHK(hipEventRecord(o->ev_dl, o->copy_stream));
  o->dl_pending = true;
  return 0;
}
int agx_ocp_download_wait(agx_ocp *o) {
  if (!o) return fail("null handle");
  if (o->dl_pending) HIPCHK(hipEventSynchronize(o->ev_dl));
  o->dl_pending = false;
  return 0;
}

int agx_ocp_upload_x0(agx_ocp *o, const double *x0) {
  if (!o || !x0) return fail("agx_ocp_upload_x0: null argument");
  if (set_device(o)) return -1;
  if (up(o, o->d_x0, x0, o->B, 2)) return -1;
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_upload_warmstart(agx_ocp *o, const double *xs_ws, const double *us_ws) {
  if (!o || !xs_ws || !us_ws) return fail("agx_ocp_upload_warmstart: null argument");
  if (set_device(o)) return -1;
  if (up(o, o->d_xs, xs_ws, (size_t)o->B * (o->T + 1), 2)) return -1;
  if (up(o, o->d_us, us_ws, (size_t)o->B * o->T, 1)) return -1;
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_solve_resident(agx_ocp *o, int max_iter, double max_time) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  return solve_resident(o, max_iter, max_time);
}

int agx_ocp_download(agx_ocp *o, double *xs, double *us, double *K, agx_status *st) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  const size_t B = o->B, T = o->T, nx = o->nx, nu = o->nu;
  (void)nx; (void)nu;
  if (xs && down(o, xs, o->d_xs, B * (T + 1), 2)) return -1;
  if (us && down(o, us, o->d_us, B * T, 1)) return -1;
  if (K && down_gains(o, K, o->d_Kout, B * T)) return -1;
  std::vector<DevState> hs;
  if (st) {
    hs.resize(B);
    HIPCHK(hipMemcpyAsync(hs.data(), o->d_state, sizeof(DevState) * B, hipMemcpyDeviceToHost, o->stream));
  }
  HIPCHK(hipStreamSynchronize(o->stream));
  if (st)
    for (size_t b = 0; b < B; ++b) {
      st[b].kkt = hs[b].kkt; st[b].cost = hs[b].cost; st[b].merit = hs[b].merit; st[b].gap_norm = hs[b].gap;
      st[b].iter = hs[b].done ? hs[b].iter : o->last_max_iter;
      st[b].qp_iters = hs[b].qp_iters; st[b].solved = hs[b].solved; st[b].flags = hs[b].flags;
    }
  return 0;
}

// What a controller consumes per cycle, packed on the device and copied with ONE transfer into a
// pinned host buffer owned by the handle: per instance
//   [ us0 (nu) | K0 (nu x ndx, row major) | x1 (nx) | kkt cost merit gap iter qp_iters solved flags ].
int agx_ocp_first_packed(agx_ocp *o, const double **host, int *stride) {
  if (!o || !host) return fail("agx_ocp_first_packed: null argument");
  if (set_device(o)) return -1;
  const int FS = o->nu + o->nu * o->nx + o->nx + 8;
  if (!o->h_first) {
    if (o->poll && hipHostMalloc((void **)&o->h_first, sizeof(double) * (size_t)o->B * FS, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      o->poll = false;  // no fine-grained mapped memory: copy + synchronize from here on
    }
    if (!o->poll) HIPCHK(hipHostMalloc((void **)&o->h_first, sizeof(double) * (size_t)o->B * FS, hipHostMallocDefault));
    if (o->poll) {
      HIPCHK(hipHostGetDevicePointer((void **)&o->d_first, o->h_first, 0));  // the pack kernel writes host memory directly
    } else {
      HIPCHK(hipMalloc((void **)&o->d_first, sizeof(double) * (size_t)o->B * FS));
    }
    o->first_stride = FS;
  }
  const long long n = (long long)o->B * FS;
  hipLaunchKernelGGL(agx::k_pack_first, dim3((int)((n + 255) / 256)), dim3(256), 0, o->stream, o->d_us, o->d_Kout, o->d_xs, o->d_state,
                     o->d_first, o->B, o->T, o->nx, o->nu, o->last_max_iter);
  HIPCHK(hipGetLastError());
  if (o->poll) {
    const unsigned long long seq = ++o->seq;
    if (publish(o, 3, 2, o->d_ndone, seq)) return -1;
    if (wait_stamp(o, 2, seq)) return -1;
  } else {
    HIPCHK(hipMemcpyAsync(o->h_first, o->d_first, sizeof(double) * n, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
  }
  if (o->padded) {  // [us0 | K0 | x1 | status] of the caller's nvu joints
    const int nv = o->nv, nvu = o->nvu, FSu = nvu + nvu * 2 * nvu + 2 * nvu + 8;
    o->h_first_u.resize((size_t)o->B * FSu);
    for (int b = 0; b < o->B; ++b) {
      const double *r = o->h_first + (size_t)b * FS;
      double *w = o->h_first_u.data() + (size_t)b * FSu;
      unpad_rows(w, r, 1, 1, nvu, nv);
      for (int i = 0; i < nvu; ++i) unpad_rows(w + nvu + (size_t)i * 2 * nvu, r + nv + (size_t)i * 2 * nv, 1, 2, nvu, nv);
      unpad_rows(w + nvu + nvu * 2 * nvu, r + nv + nv * 2 * nv, 1, 2, nvu, nv);
      std::memcpy(w + nvu + nvu * 2 * nvu + 2 * nvu, r + nv + nv * 2 * nv + 2 * nv, sizeof(double) * 8);
    }
    *host = o->h_first_u.data();
    if (stride) *stride = FSu;
    return 0;
  }
  *host = o->h_first;
  if (stride) *stride = FS;
  return 0;
}

int agx_ocp_download_first(agx_ocp *o, double *us0, double *K0, double *x1, agx_status *st) {
  const double *h = nullptr;
  int FS = 0;
  if (agx_ocp_first_packed(o, &h, &FS)) return -1;
  const int nu = o->nvu, nx = 2 * o->nvu, nk = nu * nx;
  for (int b = 0; b < o->B; ++b) {
    const double *r = h + (size_t)b * FS;
    if (us0) std::memcpy(us0 + (size_t)b * nu, r, sizeof(double) * nu);
    if (K0) std::memcpy(K0 + (size_t)b * nk, r + nu, sizeof(double) * nk);
    if (x1) std::memcpy(x1 + (size_t)b * nx, r + nu + nk, sizeof(double) * nx);
    if (st) {
      const double *q = r + nu + nk + nx;
      st[b].kkt = q[0]; st[b].cost = q[1]; st[b].merit = q[2]; st[b].gap_norm = q[3];
      st[b].iter = (int)q[4]; st[b].qp_iters = (int)q[5]; st[b].solved = (int)q[6]; st[b].flags = (int)q[7];
    }
  }
  return 0;
}

int agx_ocp_solve(agx_ocp *o, const double *x0, const double *xs_ws, const double *us_ws, int max_iter, double max_time,
                  double *xs, double *us, double *K, agx_status *st) {
  if (!o || !x0 || !xs_ws || !us_ws) return fail("agx_ocp_solve: null argument");
  if (agx_ocp_upload_x0(o, x0)) return -1;
  if (agx_ocp_upload_warmstart(o, xs_ws, us_ws)) return -1;
  if (solve_resident(o, max_iter, max_time)) return -1;
  return agx_ocp_download(o, xs, us, K, st);
}

int agx_ocp_shift_warmstart(agx_ocp *o) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    const long long units = (long long)o->B * o->T;
    const long long n = (long long)o->B * (o->T + 1) * o->nx;
    if constexpr (NV > 8) {
      // large models: plain copies for the nodes with dt_i == dt_0, one workgroup per node that has to be integrated
      (void)units;
      hipLaunchKernelGGL(agx::k_shift_copy, dim3((int)((n + 255) / 256)), dim3(256), 0, o->stream, o->d_dt, o->d_xs, o->d_us, o->B, o->T, o->nx, o->nu);
      if (!o->shift_nodes.empty())
        hipLaunchKernelGGL((agx::k_integrate_wg<NV>), dim3(o->B * (int)o->shift_nodes.size()), dim3(256), 0, o->stream, o->d_model, o->dt[0], o->d_xs,
                           o->d_us, o->d_xs + n, o->d_shift_nodes, (int)o->shift_nodes.size(), o->T);
    } else
    hipLaunchKernelGGL((agx::k_shift<NV, CH>), dim3((int)((units + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp, o->d_dt, o->d_xs, o->d_us);
    hipLaunchKernelGGL(agx::k_shift_commit, dim3((int)((n + 255) / 256)), dim3(256), 0, o->stream, o->d_xs, o->d_us, o->B, o->T, o->nx, o->nu);
    HIPCHK(hipGetLastError());
    return 0;
  });
}

int agx_ocp_x0_from_prediction(agx_ocp *o) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  hipLaunchKernelGGL(agx::k_x0_from_pred, dim3((o->B * o->nx + 255) / 256), dim3(256), 0, o->stream, o->d_x0, o->d_xs, o->B, o->T, o->nx);
  HIPCHK(hipGetLastError());
  return 0;
}

int agx_ocp_integrate(agx_ocp *o, int n, const double *x, const double *u, double *xnext) {
  if (!o || !x || !u || !xnext || n < 1) return fail("agx_ocp_integrate: bad argument");
  if (set_device(o)) return -1;
  const size_t bytes = sizeof(double) * n * (2 * o->nx + o->nu);
  if (ensure_scratch(o, bytes)) return -1;
  double *dx = o->d_scratch, *du = dx + (size_t)n * o->nx, *dn = du + (size_t)n * o->nu;
  if (up(o, dx, x, n, 2) || up(o, du, u, n, 1)) return -1;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    if constexpr (NV > 8)
      hipLaunchKernelGGL((agx::k_integrate_wg<NV>), dim3(n), dim3(256), 0, o->stream, o->d_model, o->dt[0], dx, du, dn, (const int *)nullptr, 0, 0);
    else
      hipLaunchKernelGGL((agx::k_integrate<NV, CH>), dim3((n + 63) / 64), dim3(64), 0, o->stream, o->d_model, o->dt[0], n, dx, du, dn);
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  if (down(o, xnext, dn, n, 2)) return -1;
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_model_rnea(agx_ocp *o, int n, const double *q, const double *v, const double *a, double *tau) {
  if (!o || !q || !v || !a || !tau || n < 1) return fail("agx_model_rnea: bad argument");
  if (set_device(o)) return -1;
  const size_t cnt = (size_t)n * o->nv;
  if (ensure_scratch(o, sizeof(double) * 4 * cnt)) return -1;
  double *dq = o->d_scratch, *dv = dq + cnt, *da = dv + cnt, *dt = da + cnt;
  if (up(o, dq, q, n, 1) || up(o, dv, v, n, 1) || up(o, da, a, n, 1)) return -1;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    hipLaunchKernelGGL((agx::k_rnea<NV, CH>), dim3((n + 63) / 64), dim3(64), 0, o->stream, o->d_model, n, dq, dv, da, dt);
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  if (down(o, tau, dt, n, 1)) return -1;
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_model_frame_placement(agx_ocp *o, int n, int frame, const double *q, double *out) {
  if (!o || !q || !out || n < 1) return fail("agx_model_frame_placement: bad argument");
  if (frame < 0 || frame >= o->hm.nframes) return fail("agx_model_frame_placement: frame id out of range");
  if (set_device(o)) return -1;
  const size_t cnt = (size_t)n * o->nv;
  if (ensure_scratch(o, sizeof(double) * (cnt + (size_t)n * 12))) return -1;
  double *dq = o->d_scratch, *dout = dq + cnt;
  if (up(o, dq, q, n, 1)) return -1;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    hipLaunchKernelGGL((agx::k_frame<NV, CH>), dim3((n + 63) / 64), dim3(64), 0, o->stream, o->d_model, n, frame, dq, dout);
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, dout, sizeof(double) * n * 12, hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_model_frame_jacobian(agx_ocp *o, int n, int frame, int local, const double *q, double *J) {
  if (!o || !q || !J || n < 1) return fail("agx_model_frame_jacobian: bad argument");
  if (frame < 0 || frame >= o->hm.nframes) return fail("agx_model_frame_jacobian: frame id out of range");
  if (set_device(o)) return -1;
  const size_t cnt = (size_t)n * o->nv;
  if (ensure_scratch(o, sizeof(double) * (cnt + 6 * cnt))) return -1;
  double *dq = o->d_scratch, *dout = dq + cnt;
  if (up(o, dq, q, n, 1)) return -1;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    hipLaunchKernelGGL((agx::k_frame_jacobian<NV, CH>), dim3((n + 63) / 64), dim3(64), 0, o->stream, o->d_model, n, frame, local, dq, dout);
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  if (down(o, J, dout, (size_t)n * 6, 1)) return -1;
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_get_residuals(agx_ocp *o, int row, double *out) {
  if (!o || !out) return fail("agx_ocp_get_residuals: null argument");
  if (row < 0 || row >= o->ho.rows[0].n) return fail("agx_ocp_get_residuals: row out of range");
  if (set_device(o)) return -1;
  const int nr = o->ho.rows[0].nr[row];
  const size_t cnt = (size_t)o->B * o->T * nr;
  if (ensure_scratch(o, sizeof(double) * cnt)) return -1;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    const long long units = (long long)o->B * o->T;
    hipLaunchKernelGGL((agx::k_residuals<NV, CH>), dim3((int)((units + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp, o->d_xs, o->d_us, o->rv, row, o->d_scratch);
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  {
    const int kind = o->ho.rows[0].kind[row];
    const bool joint_blocks = kind == AGX_RES_STATE || kind == AGX_RES_CONTROL || kind == AGX_RES_CONTROL_GRAV;
    if (o->padded && joint_blocks) { if (down(o, out, o->d_scratch, (size_t)o->B * o->T, nr / o->nv)) return -1; }
    else HIPCHK(hipMemcpyAsync(out, o->d_scratch, sizeof(double) * cnt, hipMemcpyDeviceToHost, o->stream));
  }
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_calc_diff(agx_ocp *o, double *tiles) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  if (launch_calc_diff(o, false)) return -1;
  if (tiles && o->padded) {
    // canonical tiles Fx | Fu | f | Lx | Lu | Lxx | Lxu | Luu | cost of the caller's nvu joints out of the capacity-sized ones
    const size_t nodes = (size_t)o->B * (o->T + 1);
    const int nv = o->nv, nvu = o->nvu, NX = 2 * nv, NU = nv, nxu = 2 * nvu, nuu = nvu, TU = AGX_TILE_DOUBLES(nvu);
    o->stage.resize(nodes * o->tile);
    HIPCHK(hipMemcpyAsync(o->stage.data(), o->d_tiles, sizeof(double) * o->stage.size(), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    auto xm = [&](int i) { return i < nvu ? i : nv + (i - nvu); };
    for (size_t n = 0; n < nodes; ++n) {
      const double *c = o->stage.data() + n * o->tile;
      double *u = tiles + n * TU;
      const double *cFx = c, *cFu = cFx + NX * NX, *cf = cFu + NX * NU, *cLx = cf + NX, *cLu = cLx + NX, *cLxx = cLu + NU,
                   *cLxu = cLxx + NX * NX, *cLuu = cLxu + NX * NU, *ccost = cLuu + NU * NU;
      double *uFx = u, *uFu = uFx + nxu * nxu, *uf = uFu + nxu * nuu, *uLx = uf + nxu, *uLu = uLx + nxu, *uLxx = uLu + nuu,
             *uLxu = uLxx + nxu * nxu, *uLuu = uLxu + nxu * nuu, *ucost = uLuu + nuu * nuu;
      for (int i = 0; i < nxu; ++i) {
        for (int j = 0; j < nxu; ++j) { uFx[i * nxu + j] = cFx[xm(i) * NX + xm(j)]; uLxx[i * nxu + j] = cLxx[xm(i) * NX + xm(j)]; }
        for (int j = 0; j < nuu; ++j) { uFu[i * nuu + j] = cFu[xm(i) * NU + j]; uLxu[i * nuu + j] = cLxu[xm(i) * NU + j]; }
        uf[i] = cf[xm(i)]; uLx[i] = cLx[xm(i)];
      }
      for (int i = 0; i < nuu; ++i) {
        uLu[i] = cLu[i];
        for (int j = 0; j < nuu; ++j) uLuu[i * nuu + j] = cLuu[i * NU + j];
      }
      *ucost = *ccost;
    }
    return 0;
  }
  if (tiles) HIPCHK(hipMemcpyAsync(tiles, o->d_tiles, sizeof(double) * o->B * (o->T + 1) * (size_t)o->tile, hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

// One QP direction at the resident (xs, us) through the production kernels: K1 (QP tiles),
// K2 (Riccati + forward), the prologue of K4 (du, KKT) and the exit path (reported gains).
int agx_ocp_direction(agx_ocp *o, double *K, double *k, double *dx, double *du, double *kkt) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  if (reset_state(o)) return -1;
  if (launch_calc_qp(o)) return -1;
  if (launch_riccati(o, 1, 0)) return -1;
  if (launch_step(o, 0, 1, 0)) return -1;
  if (launch_gains(o)) return -1;
  const size_t B = o->B, T = o->T, nx = o->nx, nu = o->nu;
  (void)nx; (void)nu;
  if (K && down_gains(o, K, o->d_Kout, B * T)) return -1;
  if (k && down(o, k, o->d_kws, B * T, 1)) return -1;
  if (dx && down(o, dx, o->d_dx, B * (T + 1), 2)) return -1;
  if (du && down(o, du, o->d_du, B * T, 1)) return -1;
  std::vector<DevState> hs(B);
  HIPCHK(hipMemcpyAsync(hs.data(), o->d_state, sizeof(DevState) * B, hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  if (kkt) for (size_t b = 0; b < B; ++b) kkt[b] = hs[b].kkt;
  return 0;
}

int agx_ocp_qp_tiles(agx_ocp *o, double *qt, double *aux, int *qt_size, int *aux_size) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  if (qt_size) *qt_size = o->qt_size;
  if (aux_size) *aux_size = o->aux_size;
  if (!qt && !aux) return 0;
  if (reset_state(o)) return -1;
  if (launch_calc_qp(o)) return -1;
  const size_t n = (size_t)o->B * (o->T + 1);
  if (qt) HIPCHK(hipMemcpyAsync(qt, o->d_qt, sizeof(double) * n * o->qt_size, hipMemcpyDeviceToHost, o->stream));
  if (aux) HIPCHK(hipMemcpyAsync(aux, o->d_aux, sizeof(double) * n * o->aux_size, hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

#ifdef AGX_WG_PROFILE
// development only: cycle stamps of one workgroup of the large-model derivative pass (not part of the ABI)
int agx_dev_wg_stamps(long long *out, int *n) {
  int zero = 0;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(agx::g_wg_ts), sizeof(long long) * 64) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(agx::g_wg_n), sizeof(int)) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(agx::g_wg_n), &zero, sizeof(int));
  return 0;
}
#endif

int agx_ocp_time_kernel(agx_ocp *o, int which, int reps, double *avg_ms) {
  if (!o || !avg_ms || reps < 1) return fail("agx_ocp_time_kernel: bad argument");
  if (set_device(o)) return -1;
  // state for the timed kernel: fresh solver state, QP tiles and a direction at the resident point
  if (reset_state(o)) return -1;
  if (which == 1 || which == 2 || which == 5 || which == 6 || which == 7) { if (launch_calc_qp(o)) return -1; }
  if (which == 2) { if (launch_riccati(o, 1, 0)) return -1; }
  // warm-up launch, then `reps` timed launches bracketed by events on the problem's stream
  for (int pass = 0; pass < 2; ++pass) {
    const int n = pass == 0 ? 1 : reps;
    if (pass == 1) HIPCHK(hipEventRecord(o->ev0, o->stream));
    for (int r = 0; r < n; ++r) {
      int rc = 0;
      if (which == 0) rc = launch_calc_qp(o, false);
      else if (which == 1) rc = launch_riccati(o, 1, 0);
      else if (which == 2) rc = launch_step(o, 0, 1 << 30, 1 | 4);
      else if (which == 3) rc = launch_calc_qp(o, true);
      else if (which == 4) rc = launch_calc_diff(o, false, true);
      else if (which == 5) rc = launch_riccati(o, 0, 0);
      else if (which == 6) rc = launch_gains(o);
      else if (which == 7) rc = launch_riccati(o, 1, true, 1);
      else return fail("agx_ocp_time_kernel: unknown kernel");
      if (rc) return rc;
    }
    if (pass == 1) HIPCHK(hipEventRecord(o->ev1, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
  }
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, o->ev0, o->ev1));
  *avg_ms = (double)ms / reps;
  return 0;
}

int agx_ocp_profile(agx_ocp *o, int enable, double *ms_sum, long long *count) {
  if (!o) return fail("null handle");
  if (ms_sum && count)
    for (int k = 0; k < 3; ++k) { ms_sum[k] = o->prof_ms[k]; count[k] = o->prof_n[k]; }
  if (enable != (o->prof ? 1 : 0)) {
    o->prof = enable != 0;
    for (int k = 0; k < 3; ++k) { o->prof_ms[k] = 0.0; o->prof_n[k] = 0; }
  }
  return 0;
}

// The resident generators track ONE end-effector frame: frame-based cost rows look at it on every node
// (what the host path does per node with `obj.id = ...`, ocp_croco_generic.py:208).
static int traj_frames(agx_ocp *o, int frame) {
  std::vector<int> fr((size_t)o->B * (o->T + 1) * AGX_MAX_ROWS, -1);
  for (int b = 0; b < o->B; ++b)
    for (int t = 0; t <= o->T; ++t) {
      const DevRows &rows = o->ho.rows[t == o->T ? 1 : 0];
      for (int r = 0; r < rows.n; ++r)
        if (rows.kind[r] == AGX_RES_FRAME_PLACEMENT || rows.kind[r] == AGX_RES_FRAME_TRANSLATION || rows.kind[r] == AGX_RES_FRAME_ROTATION)
          fr[((size_t)b * (o->T + 1) + t) * AGX_MAX_ROWS + r] = frame;
    }
  HIPCHK(hipMemcpyAsync(o->d_frames, fr.data(), sizeof(int) * fr.size(), hipMemcpyHostToDevice, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  o->frames_set = true;
  return 0;
}

int agx_traj_sine_create(agx_ocp *o, int n_points, double dt, const double *q0, const double *amp, const double *pulsation,
                         const double *scale_duration, const double *t0, const double *w_q, const double *w_qdot,
                         const double *w_effort, const double *w_pose, int frame) {
  if (!o || !q0 || !amp || !pulsation || !scale_duration || !t0 || !w_q || !w_qdot || !w_effort || !w_pose)
    return fail("agx_traj_sine_create: null argument");
  if (n_points < o->T + 1) return fail("agx_traj_sine_create: need at least T+1 samples");
  if (frame < 0 || frame >= o->hm.nframes) return fail("agx_traj_sine_create: frame id out of range");
  if (set_device(o)) return -1;
  const size_t B = o->B, nv = o->nv;
  if (o->d_traj) { (void)hipFree(o->d_traj); o->d_traj = nullptr; }
  if (o->d_pts) { (void)hipFree(o->d_pts); o->d_pts = nullptr; }
  if (o->d_sine) { (void)hipFree(o->d_sine); o->d_sine = nullptr; }
  HIPCHK(hipMalloc((void **)&o->d_traj, sizeof(double) * B * n_points * 2 * o->stride));
  HIPCHK(hipMalloc((void **)&o->d_pts, sizeof(double) * B * n_points * (4 * nv + 12)));
  HIPCHK(hipMalloc((void **)&o->d_sine, sizeof(double) * (4 * B * nv + B)));
  double *d = o->d_sine;
  // pad joints: amplitude 0 (they stay at rest), scale duration 1 (it divides)
  if (up(o, d, q0, B, 1) || up(o, d + B * nv, amp, B, 1) || up(o, d + 2 * B * nv, pulsation, B, 1) || up(o, d + 3 * B * nv, scale_duration, B, 1, 1.0)) return -1;
  HIPCHK(hipMemcpyAsync(d + 4 * B * nv, t0, sizeof(double) * B, hipMemcpyHostToDevice, o->stream));
  agx::SineParams sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.q0 = d; sp.amp = d + B * nv; sp.puls = d + 2 * B * nv; sp.scale = d + 3 * B * nv; sp.t0 = d + 4 * B * nv;
  for (size_t i = 0; i < nv; ++i) {  // pad joints: weight 1 on an identically zero residual
    const bool real = i < (size_t)o->nvu;
    sp.w_q[i] = real ? w_q[i] : 1.0; sp.w_qdot[i] = real ? w_qdot[i] : 1.0; sp.w_effort[i] = real ? w_effort[i] : 1.0;
  }
  for (int i = 0; i < 6; ++i) sp.w_pose[i] = w_pose[i];
  sp.dt = dt; sp.n_points = n_points; sp.frame = frame;
  o->n_points = n_points;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    const long long units = (long long)B * n_points;
    hipLaunchKernelGGL((agx::k_sine_fill<NV, CH>), dim3((int)((units + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp, sp, o->d_traj, o->d_pts);
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(o->stream));
  if (traj_frames(o, frame)) return -1;
  return agx_traj_set_window(o, 0);
}

int agx_traj_generic_create(agx_ocp *o, int n_points, const double *q, const double *dq, const double *ddq, const double *w_q,
                            const double *w_qdot, const double *w_effort, const double *w_pose, int frame) {
  if (!o || !q || !dq || !ddq || !w_q || !w_qdot || !w_effort || !w_pose) return fail("agx_traj_generic_create: null argument");
  if (n_points < o->T + 1) return fail("agx_traj_generic_create: trajectory shorter than the horizon");
  if (frame < 0 || frame >= o->hm.nframes) return fail("agx_traj_generic_create: frame id out of range");
  if (set_device(o)) return -1;
  const size_t B = o->B, nv = o->nv, n = B * (size_t)n_points * nv;
  if (o->d_traj) { (void)hipFree(o->d_traj); o->d_traj = nullptr; }
  if (o->d_pts) { (void)hipFree(o->d_pts); o->d_pts = nullptr; }
  if (o->d_sine) { (void)hipFree(o->d_sine); o->d_sine = nullptr; }
  HIPCHK(hipMalloc((void **)&o->d_traj, sizeof(double) * B * n_points * 2 * o->stride));
  HIPCHK(hipMalloc((void **)&o->d_pts, sizeof(double) * B * n_points * (4 * nv + 12)));
  HIPCHK(hipMalloc((void **)&o->d_sine, sizeof(double) * 3 * n));
  double *d = o->d_sine;
  if (up(o, d, q, B * (size_t)n_points, 1) || up(o, d + n, dq, B * (size_t)n_points, 1) || up(o, d + 2 * n, ddq, B * (size_t)n_points, 1)) return -1;
  agx::SineParams sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.gq = d; sp.gdq = d + n; sp.gddq = d + 2 * n;
  for (size_t i = 0; i < nv; ++i) {  // pad joints: weight 1 on an identically zero residual
    const bool real = i < (size_t)o->nvu;
    sp.w_q[i] = real ? w_q[i] : 1.0; sp.w_qdot[i] = real ? w_qdot[i] : 1.0; sp.w_effort[i] = real ? w_effort[i] : 1.0;
  }
  for (int i = 0; i < 6; ++i) sp.w_pose[i] = w_pose[i];
  sp.dt = 0.0; sp.n_points = n_points; sp.frame = frame;
  o->n_points = n_points;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    const long long units = (long long)B * n_points;
    hipLaunchKernelGGL((agx::k_sine_fill<NV, CH>), dim3((int)((units + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp, sp, o->d_traj, o->d_pts);
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(o->stream));
  if (traj_frames(o, frame)) return -1;
  return agx_traj_set_window(o, 0);
}

int agx_traj_cartesian_sine_create(agx_ocp *o, int n_points, double dt, const double *q0, const double *amp, const double *pulsation,
                                   double scale_duration, double precision, int it_max, const double *w_q, const double *w_qdot,
                                   const double *w_effort, const double *w_pose, int frame) {
  if (!o || !q0 || !amp || !pulsation || !w_q || !w_qdot || !w_effort || !w_pose) return fail("agx_traj_cartesian_sine_create: null argument");
  if (n_points < o->T + 1) return fail("agx_traj_cartesian_sine_create: trajectory shorter than the horizon");
  if (frame < 0 || frame >= o->hm.nframes) return fail("agx_traj_cartesian_sine_create: frame id out of range");
  if (!(scale_duration > 0.0) || !(precision > 0.0) || it_max < 1) return fail("agx_traj_cartesian_sine_create: scale_duration, precision, it_max must be positive");
  if (o->nv > 7) return fail("agx_traj_cartesian_sine_create: nv <= 7");
  if (set_device(o)) return -1;
  const size_t B = o->B, nv = o->nv, n = B * (size_t)n_points * nv;
  if (o->d_traj) { (void)hipFree(o->d_traj); o->d_traj = nullptr; }
  if (o->d_pts) { (void)hipFree(o->d_pts); o->d_pts = nullptr; }
  if (o->d_sine) { (void)hipFree(o->d_sine); o->d_sine = nullptr; }
  HIPCHK(hipMalloc((void **)&o->d_traj, sizeof(double) * B * n_points * 2 * o->stride));
  HIPCHK(hipMalloc((void **)&o->d_pts, sizeof(double) * B * n_points * (4 * nv + 12)));
  // q | dq | ddq samples, then the per-instance parameters q0 | amp | pulsation and the failure flags
  const size_t npar = B * (nv + 6), npose = B * (size_t)n_points * 12;
  o->n_points = 0;  // until the trajectory is complete: a failed build leaves nothing a later window / MPC step could consume
  HIPCHK(hipMalloc((void **)&o->d_sine, sizeof(double) * (3 * n + npar + npose) + sizeof(int) * B));
  double *d = o->d_sine, *dpar = d + 3 * n, *dpose = dpar + npar;
  int *dfail = reinterpret_cast<int *>(dpose + npose);
  HIPCHK(hipMemsetAsync(d + 2 * n, 0, sizeof(double) * n, o->stream));  // ddq = 0
  if (up(o, dpar, q0, B, 1)) return -1;
  HIPCHK(hipMemcpyAsync(dpar + B * nv, amp, sizeof(double) * B * 3, hipMemcpyHostToDevice, o->stream));
  HIPCHK(hipMemcpyAsync(dpar + B * nv + B * 3, pulsation, sizeof(double) * B * 3, hipMemcpyHostToDevice, o->stream));
  agx::CartSineParams cp;
  cp.q0 = dpar; cp.amp = dpar + B * nv; cp.puls = dpar + B * nv + B * 3;
  cp.dt = dt; cp.scale = scale_duration; cp.precision = precision;
  cp.n_points = n_points; cp.frame = frame; cp.it_max = it_max;
  cp.q = d; cp.dq = d + n; cp.pose = dpose; cp.fail = dfail;
  agx::SineParams sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.gq = d; sp.gdq = d + n; sp.gddq = d + 2 * n;
  for (size_t i = 0; i < nv; ++i) {  // pad joints: weight 1 on an identically zero residual
    const bool real = i < (size_t)o->nvu;
    sp.w_q[i] = real ? w_q[i] : 1.0; sp.w_qdot[i] = real ? w_qdot[i] : 1.0; sp.w_effort[i] = real ? w_effort[i] : 1.0;
  }
  for (int i = 0; i < 6; ++i) sp.w_pose[i] = w_pose[i];
  sp.dt = 0.0; sp.n_points = n_points; sp.frame = frame;
  sp.gpose = dpose;
  int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    if constexpr (NV <= 7) {
      hipLaunchKernelGGL((agx::k_cartesian_sine_ik<NV, CH>), dim3((int)((B + 63) / 64)), dim3(64), 0, o->stream, o->d_model, (int)B, cp);
      const long long units = (long long)B * n_points;
      hipLaunchKernelGGL((agx::k_sine_fill<NV, CH>), dim3((int)((units + 63) / 64)), dim3(64), 0, o->stream, o->d_model, o->d_ocp, sp, o->d_traj, o->d_pts);
    }
    HIPCHK(hipGetLastError());
    return 0;
  });
  if (rc) return rc;
  std::vector<int> failed(B);
  HIPCHK(hipMemcpyAsync(failed.data(), dfail, sizeof(int) * B, hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  for (size_t b = 0; b < B; ++b)
    if (failed[b]) {
      char msg[160];
      std::snprintf(msg, sizeof(msg), "inverse kinematics failed to converge: instance %zu at point %d (it_max %d)", b, failed[b] - 1, it_max);
      // nothing half-built stays behind: set_window / mpc_step refuse a handle without a trajectory
      (void)hipFree(o->d_traj); o->d_traj = nullptr;
      (void)hipFree(o->d_pts); o->d_pts = nullptr;
      (void)hipFree(o->d_sine); o->d_sine = nullptr;
      return fail(msg);
    }
  o->n_points = n_points;
  if (traj_frames(o, frame)) return -1;
  return agx_traj_set_window(o, 0);
}

int agx_traj_set_horizon_indexes(agx_ocp *o, const int32_t *idx) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  o->hidx.clear();
  if (!idx) return 0;  // back to uniform
  bool uniform = true;
  for (int t = 0; t <= o->T; ++t) {
    if (idx[t] < 0 || (t > 0 && idx[t] <= idx[t - 1])) return fail("agx_traj_set_horizon_indexes: indexes must be non-negative and increasing");
    uniform = uniform && idx[t] == t;
  }
  if (uniform) return 0;
  o->hidx.assign(idx, idx + o->T + 1);
  if (!o->d_hidx) HIPCHK(hipMalloc((void **)&o->d_hidx, sizeof(int) * (o->T + 1)));
  HIPCHK(hipMemcpyAsync(o->d_hidx, o->hidx.data(), sizeof(int) * (o->T + 1), hipMemcpyHostToDevice, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_traj_set_window(agx_ocp *o, int k0) {
  if (!o) return fail("null handle");
  if (!o->d_traj) return fail("agx_traj_set_window: no resident trajectory");
  const int last = o->hidx.empty() ? o->T : o->hidx[o->T];
  if (k0 < 0 || k0 + last + 1 > o->n_points) return fail("agx_traj_set_window: window leaves the trajectory");
  o->win_k0 = k0;
  o->rv.frames = o->d_frames;
  if (o->hidx.empty()) {
    o->rv.base = o->d_traj + (long long)k0 * 2 * o->stride;
    o->rv.bstride = (long long)o->n_points * 2 * o->stride;
    o->rv.tstride = 2 * o->stride;
    o->rv.term_off = o->stride;
    return 0;
  }
  if (set_device(o)) return -1;
  const long long n = (long long)o->B * (o->T + 1) * o->stride;
  hipLaunchKernelGGL(agx::k_gather_window, dim3((int)((n + 255) / 256)), dim3(256), 0, o->stream, o->d_traj, o->d_ref, o->d_hidx, o->B, o->T,
                     o->stride, o->n_points, k0);
  HIPCHK(hipGetLastError());
  o->rv.base = o->d_ref;
  o->rv.bstride = (long long)(o->T + 1) * o->stride;
  o->rv.tstride = o->stride;
  o->rv.term_off = 0;
  return 0;
}

int agx_traj_get_point(agx_ocp *o, int k, double *q, double *v, double *a, double *u, double *pose) {
  if (!o || !o->d_pts) return fail("agx_traj_get_point: no resident trajectory");
  if (k < 0 || k >= o->n_points) return fail("agx_traj_get_point: sample out of range");
  if (set_device(o)) return -1;
  const size_t B = o->B, nv = o->nv, w = 4 * nv + 12;
  std::vector<double> h(B * w);
  HIPCHK(hipMemcpy2DAsync(h.data(), sizeof(double) * w, o->d_pts + (size_t)k * w, sizeof(double) * o->n_points * w, sizeof(double) * w, B, hipMemcpyDeviceToHost, o->stream));
  HIPCHK(hipStreamSynchronize(o->stream));
  for (size_t b = 0; b < B; ++b) {
    const double *p = &h[b * w];
    const size_t nvu = o->nvu;  // the sample's joint blocks hold nv (capacity) entries, the caller's nvu
    if (q) std::memcpy(q + b * nvu, p, sizeof(double) * nvu);
    if (v) std::memcpy(v + b * nvu, p + nv, sizeof(double) * nvu);
    if (a) std::memcpy(a + b * nvu, p + 2 * nv, sizeof(double) * nvu);
    if (u) std::memcpy(u + b * nvu, p + 3 * nv, sizeof(double) * nvu);
    if (pose) std::memcpy(pose + b * 12, p + 4 * nv, sizeof(double) * 12);
  }
  return 0;
}

static int ws_from_ref(agx_ocp *o, int k0, int set_x0) {
  const long long units = (long long)o->B * (o->T + 1);
  hipLaunchKernelGGL(agx::k_ws_from_ref, dim3((int)((units + 255) / 256)), dim3(256), 0, o->stream, o->d_xs, o->d_us, o->d_x0, o->d_pts,
                     o->B, o->T, o->nv, o->n_points, k0, set_x0, o->hidx.empty() ? nullptr : o->d_hidx);
  HIPCHK(hipGetLastError());
  return 0;
}

int agx_traj_warmstart_from_reference(agx_ocp *o) {
  if (!o || !o->d_pts) return fail("agx_traj_warmstart_from_reference: no resident trajectory");
  if (set_device(o)) return -1;
  return ws_from_ref(o, o->win_k0, 1);
}

int agx_ocp_download_x0(agx_ocp *o, double *x0) {
  if (!o || !x0) return fail("agx_ocp_download_x0: null argument");
  if (set_device(o)) return -1;
  if (down(o, x0, o->d_x0, o->B, 2)) return -1;
  HIPCHK(hipStreamSynchronize(o->stream));
  return 0;
}

int agx_ocp_feedback_rollout(agx_ocp *o, int n_substeps, double dt_sub, const double *disturbance) {
  if (!o) return fail("null handle");
  if (n_substeps < 1 || !(dt_sub > 0.0)) return fail("agx_ocp_feedback_rollout: bad sub-stepping");
  if (set_device(o)) return -1;
  double *d_dist = nullptr;
  if (disturbance) {
    if (ensure_scratch(o, sizeof(double) * o->B * o->nu)) return -1;
    d_dist = o->d_scratch;
    if (up(o, d_dist, disturbance, o->B, 1)) return -1;
  }
  return dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
    constexpr int NV = decltype(NVc)::value;
    constexpr bool CH = decltype(CHc)::value;
    hipLaunchKernelGGL((agx::k_feedback_rollout<NV, CH>), dim3((o->B + 63) / 64), dim3(64), 0, o->stream, o->d_model, o->d_us, o->d_Kout, o->d_x0,
                       d_dist, o->B, o->T, n_substeps, dt_sub);
    HIPCHK(hipGetLastError());
    if (disturbance) HIPCHK(hipStreamSynchronize(o->stream));  // the caller may reuse its buffer
    return 0;
  });
}

int agx_ocp_mpc_step(agx_ocp *o, int k0, int max_iter, int first) {
  if (!o) return fail("null handle");
  if (set_device(o)) return -1;
  if (agx_traj_set_window(o, k0)) return -1;
  bool prologue_done = false;
  if (first == 1) {
    if (ws_from_ref(o, k0, 1)) return -1;
  } else {
    // x0 <- xs[1] (first == 0; first == 2: x0 was set by the caller / the feedback rollout), warm-start
    // shift, x0 pin and state reset: one launch when the shifted nodes of an instance fit in LDS
    const size_t lds = sizeof(double) * (size_t)o->T * (o->nx + o->nu);
    if (o->nv <= 8 && lds <= 60 * 1024) {
      int rc = dispatch(o->nv, o->chain, [&](auto NVc, auto CHc) -> int {
        constexpr int NV = decltype(NVc)::value;
        constexpr bool CH = decltype(CHc)::value;
        if constexpr (NV <= 8) {
          hipLaunchKernelGGL((agx::k_mpc_prologue<NV, CH>), dim3(o->B), dim3(128), lds, o->stream, o->d_model, o->d_ocp, o->d_dt, o->d_xs,
                             o->d_us, o->d_x0, o->d_state, o->d_ndone, first == 0 ? 1 : 0);
          HIPCHK(hipGetLastError());
        }
        return 0;
      });
      if (rc) return rc;
      prologue_done = true;
    } else {
      if (first == 0 && agx_ocp_x0_from_prediction(o)) return -1;
      if (agx_ocp_shift_warmstart(o)) return -1;
    }
  }
  return solve_resident(o, max_iter, 0.0, prologue_done);
}

}  // extern "C"
