// oracle/agx_analytic.cpp
//
// TEST / BENCH INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH (built into oracle/liboracle.so).
//
// Second CPU leg of bench.py's cpu_baseline ("port-analytic"): the per-node calc / calcDiff with the ANALYTICAL
// derivatives the HIP kernels use -- the node arithmetic of agimus_controller_amd/csrc/agx_device.hpp compiled for the
// host (AGX_HOST_BUILD), one node per call, no cross-lane code -- behind the checker's own SQP / Riccati loop
// (oracle/agx_oracle.cpp, solve_one with `analytic`).  What it stands for upstream: Pinocchio's computeAllTerms +
// computeRNEADerivatives inside crocoddyl's DifferentialActionModelFreeFwdDynamics
// (agimus_controller/ocp/ocp_croco_generic.py:688-711), i.e. what the reference's CPU path really runs, where the
// parity checker differentiates automatically (dual numbers) and is several times slower per node.
//
// It is NOT the parity checker: it shares its formulas with the kernels.  tests/test_oracle_golden.py compares it
// with the automatic-differentiation checker (a third derivation of the same tiles, on the CPU).
#define AGX_HOST_BUILD
#include "../agimus_controller_amd/csrc/agx_device.hpp"

#include <cstring>

namespace {

struct AnaProblem {
  DevModel m;
  DevRows rows[2];
  int stride;
};

void fill_rows_host(const agx_cost_row *rows, int n, int nv, DevRows &d) {
  std::memset(&d, 0, sizeof(d));
  d.n = n;
  int off = 0;
  for (int r = 0; r < n; ++r) {
    d.kind[r] = rows[r].kind; d.act[r] = rows[r].activation; d.active[r] = rows[r].active;
    d.frame[r] = rows[r].frame; d.frame_b[r] = rows[r].frame_b; d.alpha[r] = rows[r].alpha; d.weight[r] = rows[r].weight;
    d.nref[r] = agx_row_nref(rows[r].kind, nv); d.nr[r] = agx_row_nr(rows[r].kind, nv);
    d.off[r] = off;
    off += 1 + d.nref[r] + d.nr[r];
  }
}

// canonical tile Fx | Fu | f | Lx | Lu | Lxx | Lxu | Luu | cost of one node (the arithmetic of k_calc_diff / k_calc_diff_term)
template <int NV, bool CHAIN>
void node_diff(const AnaProblem &P, bool term, double dt, const double *x, const double *u, const double *xnext_ws, const double *ref,
               const int *frames, double *tile) {
  using namespace agx;
  constexpr int NX = 2 * NV, NU = NV;
  typedef TileOff<NV> TO;
  const DevModel &m = P.m;
  std::memset(tile, 0, sizeof(double) * TO::SIZE);
  Kin<NV> k;
  kinematics<NV, CHAIN>(m, x, k);
  CostAcc<NV> c;
  if (term) {
    node_costs<NV, CHAIN, true, true>(m, P.rows[1], k, x, nullptr, ref, frames, c);
    for (int i = 0; i < NX; ++i) tile[TO::Fx + i * NX + i] = 1.0;
    tile[TO::cost] = c.cost;
    for (int i = 0; i < NV; ++i) {
      tile[TO::Lx + i] = c.Lq[i]; tile[TO::Lx + NV + i] = c.Lv[i];
      tile[TO::Lxx + (NV + i) * NX + NV + i] = c.Lvv[i];
      for (int j = 0; j < NV; ++j) tile[TO::Lxx + i * NX + j] = c.Lqq[i][j];
    }
    return;
  }
  Dyn<NV> d;
  static thread_local double nle[NV], M[NV][NV], Minv[NV][NV], qdd[NV], dq[NV][NV], dv[NV][NV];
  bias_and_inertia<NV, CHAIN>(m, k, x + NV, d, nle, M);
  spd_inverse<NV>(M, Minv);
  for (int i = 0; i < NV; ++i) {
    double a = 0.0;
    for (int j = 0; j < NV; ++j) a += Minv[i][j] * (u[j] - nle[j]);
    qdd[i] = a;
  }
  for (int i = 0; i < NV; ++i) {
    tile[TO::f + i] = x[i] + dt * x[NV + i] + dt * dt * qdd[i] - xnext_ws[i];
    tile[TO::f + NV + i] = x[NV + i] + dt * qdd[i] - xnext_ws[NV + i];
  }
  rnea_derivatives<NV, CHAIN>(m, k, d, x + NV, qdd, dq, dv);
  const double dt2 = dt * dt;
  for (int i = 0; i < NV; ++i)
    for (int j = 0; j < NV; ++j) {
      double aq = 0.0, av = 0.0;
      for (int l = 0; l < NV; ++l) { aq -= Minv[i][l] * dq[l][j]; av -= Minv[i][l] * dv[l][j]; }
      tile[TO::Fx + i * NX + j] = (i == j ? 1.0 : 0.0) + dt2 * aq;
      tile[TO::Fx + i * NX + NV + j] = (i == j ? dt : 0.0) + dt2 * av;
      tile[TO::Fx + (NV + i) * NX + j] = dt * aq;
      tile[TO::Fx + (NV + i) * NX + NV + j] = (i == j ? 1.0 : 0.0) + dt * av;
      tile[TO::Fu + i * NU + j] = dt2 * Minv[i][j];
      tile[TO::Fu + (NV + i) * NU + j] = dt * Minv[i][j];
    }
  node_costs<NV, CHAIN, false, true>(m, P.rows[0], k, x, u, ref, frames, c);
  tile[TO::cost] = dt * c.cost;
  for (int i = 0; i < NV; ++i) {
    tile[TO::Lx + i] = dt * c.Lq[i]; tile[TO::Lx + NV + i] = dt * c.Lv[i]; tile[TO::Lu + i] = dt * c.Lu[i];
    tile[TO::Lxx + (NV + i) * NX + NV + i] = dt * c.Lvv[i];
    tile[TO::Luu + i * NU + i] = dt * c.Luu[i];
    for (int j = 0; j < NV; ++j) tile[TO::Lxx + i * NX + j] = dt * c.Lqq[i][j];
  }
}

template <int NV, bool CHAIN>
void node_value(const AnaProblem &P, bool term, double dt, const double *x, const double *u, const double *ref, const int *frames,
                double *xnext, double *cost) {
  using namespace agx;
  if (term) {
    node_calc_terminal<NV, CHAIN>(P.m, P.rows[1], x, ref, frames, cost);
    for (int i = 0; i < 2 * NV; ++i) xnext[i] = x[i];
  } else {
    node_calc_running<NV, CHAIN>(P.m, P.rows[0], dt, x, u, ref, frames, xnext, cost);
  }
}

template <typename F>
bool by_size(const AnaProblem &P, F &&f) {
  const bool ch = P.m.is_chain != 0;
  switch (P.m.nv) {
    case 7: if (ch) f(std::integral_constant<int, 7>(), std::true_type()); else f(std::integral_constant<int, 7>(), std::false_type()); return true;
    case 30: f(std::integral_constant<int, 30>(), std::false_type()); return true;
  }
  return false;
}

}  // namespace

extern "C" {

// NULL when the model size / row kinds are outside what this leg covers (sizes 7 and 30; State, Control, frame and
// collision cost rows; no constraints): the caller then keeps the automatic-differentiation path.
void *ana_create(const agx_model_desc *d, const agx_ocp_desc *od) {
  if (!d || !od || (d->nv != 7 && d->nv != 30) || d->nframes > AGX_MAX_FRAMES) return nullptr;
  for (int r = 0; r < od->n_running_rows + od->n_terminal_rows; ++r) {
    const agx_cost_row &row = r < od->n_running_rows ? od->running_rows[r] : od->terminal_rows[r - od->n_running_rows];
    if (row.kind == AGX_RES_CONTROL_GRAV || row.kind == AGX_RES_FRAME_VELOCITY) return nullptr;
  }
  AnaProblem *P = new AnaProblem();
  DevModel &h = P->m;
  std::memset(&h, 0, sizeof(h));
  h.nv = d->nv; h.nframes = d->nframes; h.is_chain = 1;
  for (int i = 0; i < d->nv; ++i) {
    h.parent[i] = d->parent[i];
    if (d->parent[i] != i - 1) h.is_chain = 0;
    h.anc[i] = (1u << i) | (d->parent[i] >= 0 ? h.anc[d->parent[i]] : 0u);
    std::memcpy(h.placement[i], d->placement + 12 * i, sizeof(double) * 12);
    std::memcpy(h.axis[i], d->axis + 3 * i, sizeof(double) * 3);
    h.mass[i] = d->mass[i];
    std::memcpy(h.com[i], d->com + 3 * i, sizeof(double) * 3);
    std::memcpy(h.inertia[i], d->inertia + 9 * i, sizeof(double) * 9);
    h.armature[i] = d->armature ? d->armature[i] : 0.0;
  }
  h.gravity[0] = d->gravity ? d->gravity[0] : 0.0; h.gravity[1] = d->gravity ? d->gravity[1] : 0.0; h.gravity[2] = d->gravity ? d->gravity[2] : -9.81;
  for (int f = 0; f < d->nframes; ++f) {
    h.frame_parent[f] = d->frame_parent[f];
    std::memcpy(h.frame_placement[f], d->frame_placement + 12 * f, sizeof(double) * 12);
    h.frame_radius[f] = d->frame_radius ? d->frame_radius[f] : 0.0;
    h.frame_halflen[f] = d->frame_halflen ? d->frame_halflen[f] : 0.0;
    for (int e = 0; e < 3; ++e) h.frame_box[f][e] = d->frame_box ? d->frame_box[3 * f + e] : 0.0;
  }
  fill_rows_host(od->running_rows, od->n_running_rows, d->nv, P->rows[0]);
  fill_rows_host(od->terminal_rows, od->n_terminal_rows, d->nv, P->rows[1]);
  return P;
}
void ana_destroy(void *p) { delete static_cast<AnaProblem *>(p); }

void ana_node_diff(void *p, int term, double dt, const double *x, const double *u, const double *xnext_ws, const double *ref,
                   const int32_t *frames, double *tile) {
  const AnaProblem &P = *static_cast<AnaProblem *>(p);
  by_size(P, [&](auto NVc, auto CHc) { node_diff<decltype(NVc)::value, decltype(CHc)::value>(P, term != 0, dt, x, u, xnext_ws, ref, frames, tile); });
}
void ana_node_calc(void *p, int term, double dt, const double *x, const double *u, const double *ref, const int32_t *frames,
                   double *xnext, double *cost) {
  const AnaProblem &P = *static_cast<AnaProblem *>(p);
  by_size(P, [&](auto NVc, auto CHc) { node_value<decltype(NVc)::value, decltype(CHc)::value>(P, term != 0, dt, x, u, ref, frames, xnext, cost); });
}

}  // extern "C"
