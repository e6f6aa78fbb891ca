// oracle/agx_oracle.cpp
//
// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the library built from this file (oracle/liboracle.so).
//
// CPU fp64 restatement of the arithmetic behind agimus_controller's
// OCPCrocoGeneric.solve() (agimus_controller/agimus_controller/ocp_base_croco.py:142-182),
// which upstream delegates to Crocoddyl / Pinocchio / mim_solvers.  Those
// libraries are NOT vendored in the reference and are not installed here, so
// this file restates their published algorithms.
//
// PARITY: PINNED to the reference's own golden file
// tests/resources/simple_ocp_croco_results.pkl (extracted without unpickling into
// tests/golden/simple_ocp_croco_results.npz): from a cold start this restatement
// reproduces its xs / us / K to 1.4e-13 / 1.6e-11 / 1.0e-11 after 33 SQP iterations
// (tests/test_oracle_golden.py::test_golden_cold_start_reproduced) -- the Panda table
// of factory/robot_tables.py satisfies the golden trajectory's dynamics to 1e-11, i.e.
// it is that model.  Further pins: the reference's model-independent known answers
// (tests/test_ocp_croco_generic.py:48-72,93-113, tests/test_mpc_unicycle.py:253-257,
// tests/test_warm_start_shift_previous_reference.py:107-117), physical identities and
// finite differences.  UNPINNED (no fixture of the reference covers them; recalled
// forms): colmpc Exp / QuadExp activations and distance Jacobians, the ADMM loop with
// active constraints, the filter line search, the Quu-breakdown path.
//
// Deliberately written differently from the HIP path so that agreement means
// something: link-local Featherstone recursions and *forward-mode automatic
// differentiation* (dual numbers) for every Jacobian, where the HIP kernels use
// hand-derived world-frame analytical derivatives.
//
// Build: make -C oracle   (g++ -O2 -fopenmp -shared -fPIC)

#include "../include/agimus_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// analytic-derivative node evaluation (oracle/agx_analytic.cpp): bench.py's "port-analytic" CPU leg only, never the checker
extern "C" {
void *ana_create(const agx_model_desc *d, const agx_ocp_desc *od);
void ana_destroy(void *p);
void ana_node_diff(void *p, int term, double dt, const double *x, const double *u, const double *xnext_ws, const double *ref,
                   const int32_t *frames, double *tile);
void ana_node_calc(void *p, int term, double dt, const double *x, const double *u, const double *ref, const int32_t *frames,
                   double *xnext, double *cost);
}

namespace {

// ---------------------------------------------------------------------------
// forward-mode dual numbers
// ---------------------------------------------------------------------------
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  Dual(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
};
template <int N> inline Dual<N> operator+(const Dual<N> &a, const Dual<N> &b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N> &a, const Dual<N> &b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N> &a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N> &a, const Dual<N> &b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N> &a, const Dual<N> &b) { Dual<N> r; double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> inline Dual<N> operator+(const Dual<N> &a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double b, const Dual<N> &a) { return a + b; }
template <int N> inline Dual<N> operator-(const Dual<N> &a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double b, const Dual<N> &a) { Dual<N> r = -a; r.v += b; return r; }
template <int N> inline Dual<N> operator*(const Dual<N> &a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double b, const Dual<N> &a) { return a * b; }
template <int N> inline Dual<N> operator/(const Dual<N> &a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N> &b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N> &operator+=(Dual<N> &a, const Dual<N> &b) { a = a + b; return a; }
template <int N> inline Dual<N> &operator-=(Dual<N> &a, const Dual<N> &b) { a = a - b; return a; }
template <int N> inline Dual<N> chain(const Dual<N> &a, double f, double df) { Dual<N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = df * a.d[i]; return r; }
template <int N> inline Dual<N> sqrt(const Dual<N> &a) { double s = std::sqrt(a.v); return chain(a, s, 0.5 / s); }
template <int N> inline Dual<N> sin(const Dual<N> &a) { return chain(a, std::sin(a.v), std::cos(a.v)); }
template <int N> inline Dual<N> cos(const Dual<N> &a) { return chain(a, std::cos(a.v), -std::sin(a.v)); }
template <int N> inline Dual<N> exp(const Dual<N> &a) { double e = std::exp(a.v); return chain(a, e, e); }
template <int N> inline Dual<N> atan2(const Dual<N> &y, const Dual<N> &x) {
  double den = x.v * x.v + y.v * y.v;
  Dual<N> r; r.v = std::atan2(y.v, x.v);
  for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) / den;
  return r;
}
inline double val(double a) { return a; }
template <int N> inline double val(const Dual<N> &a) { return a.v; }
using std::atan2; using std::cos; using std::exp; using std::sin; using std::sqrt;

// ---------------------------------------------------------------------------
// model (host copy of agx_model_desc)
// ---------------------------------------------------------------------------
struct Model {
  int nv = 0, nframes = 0;
  std::vector<int> parent, frame_parent;
  std::vector<double> placement, axis, mass, com, inertia, armature, effort_limit, frame_placement;
  std::vector<double> frame_radius, frame_halflen;  // collision geometry carried by frames (0 = none / sphere)
  std::vector<double> frame_box;                    // [nframes][3] box half extents (0 = not a box)
  double gravity[3] = {0, 0, -9.81};
};

void copy_model(const agx_model_desc *d, Model &m) {
  m.nv = d->nv; m.nframes = d->nframes;
  m.parent.assign(d->parent, d->parent + d->nv);
  m.placement.assign(d->placement, d->placement + 12 * d->nv);
  m.axis.assign(d->axis, d->axis + 3 * d->nv);
  m.mass.assign(d->mass, d->mass + d->nv);
  m.com.assign(d->com, d->com + 3 * d->nv);
  m.inertia.assign(d->inertia, d->inertia + 9 * d->nv);
  m.armature.assign(d->armature, d->armature + d->nv);
  if (d->effort_limit) m.effort_limit.assign(d->effort_limit, d->effort_limit + d->nv);
  if (d->gravity) for (int i = 0; i < 3; ++i) m.gravity[i] = d->gravity[i];
  if (d->nframes > 0) {
    m.frame_parent.assign(d->frame_parent, d->frame_parent + d->nframes);
    m.frame_placement.assign(d->frame_placement, d->frame_placement + 12 * d->nframes);
    m.frame_radius.assign(d->nframes, 0.0);
    m.frame_halflen.assign(d->nframes, 0.0);
    if (d->frame_radius) m.frame_radius.assign(d->frame_radius, d->frame_radius + d->nframes);
    if (d->frame_halflen) m.frame_halflen.assign(d->frame_halflen, d->frame_halflen + d->nframes);
    m.frame_box.assign(3 * d->nframes, 0.0);
    if (d->frame_box) m.frame_box.assign(d->frame_box, d->frame_box + 3 * d->nframes);
  }
}

// ---------------------------------------------------------------------------
// small 3-vector helpers, generic in the scalar
// ---------------------------------------------------------------------------
template <class S> inline void cross(const S *a, const S *b, S *c) {
  S c0 = a[1] * b[2] - a[2] * b[1];
  S c1 = a[2] * b[0] - a[0] * b[2];
  S c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
template <class S> inline void matvec3(const S *R, const S *x, S *y) {  // y = R x
  S y0 = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  S y1 = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  S y2 = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
template <class S> inline void matTvec3(const S *R, const S *x, S *y) {  // y = R^T x
  S y0 = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  S y1 = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  S y2 = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
template <class S> inline void matmul3(const S *A, const S *B, S *C) {  // C = A B
  S t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = t[i];
}

// Rotation of joint i: R = Rfix * Rodrigues(axis, q).  (liMi of Pinocchio.)
template <class S> void joint_rotation(const Model &m, int i, const S &q, S *R) {
  const double *ax = &m.axis[3 * i];
  S c = cos(q), s = sin(q);
  S omc = 1.0 - c;
  S Rq[9];
  Rq[0] = c + omc * (ax[0] * ax[0]);
  Rq[1] = omc * (ax[0] * ax[1]) - s * ax[2];
  Rq[2] = omc * (ax[0] * ax[2]) + s * ax[1];
  Rq[3] = omc * (ax[1] * ax[0]) + s * ax[2];
  Rq[4] = c + omc * (ax[1] * ax[1]);
  Rq[5] = omc * (ax[1] * ax[2]) - s * ax[0];
  Rq[6] = omc * (ax[2] * ax[0]) - s * ax[1];
  Rq[7] = omc * (ax[2] * ax[1]) + s * ax[0];
  Rq[8] = c + omc * (ax[2] * ax[2]);
  S Rf[9];
  for (int k = 0; k < 9; ++k) Rf[k] = S(m.placement[12 * i + k]);
  matmul3(Rf, Rq, R);
}

// ---------------------------------------------------------------------------
// RNEA in link-local coordinates (Featherstone, RBDA table 5.1), spatial
// vectors ordered [linear; angular].  Restates pinocchio::rnea, the call at
// warm_start_reference.py:78 and inside crocoddyl's free-forward dynamics.
// ---------------------------------------------------------------------------
template <class S>
void rnea(const Model &m, const S *q, const S *v, const S *a, S *tau, bool with_gravity) {
  const int n = m.nv;
  static thread_local std::vector<S> R, vl, vw, al, aw, fl, fw;  // reused: no heap traffic per call
  R.resize(9 * n); vl.resize(3 * n); vw.resize(3 * n); al.resize(3 * n); aw.resize(3 * n); fl.resize(3 * n); fw.resize(3 * n);
  for (int i = 0; i < n; ++i) {
    S *Ri = &R[9 * i];
    joint_rotation(m, i, q[i], Ri);
    const double *p = &m.placement[12 * i + 9];
    const double *ax = &m.axis[3 * i];
    S pv[3] = {S(p[0]), S(p[1]), S(p[2])};
    S pl[3], pw[3], pal[3], paw[3];  // parent velocity / acceleration
    int par = m.parent[i];
    if (par >= 0) {
      for (int k = 0; k < 3; ++k) { pl[k] = vl[3 * par + k]; pw[k] = vw[3 * par + k]; pal[k] = al[3 * par + k]; paw[k] = aw[3 * par + k]; }
    } else {
      for (int k = 0; k < 3; ++k) { pl[k] = S(0.0); pw[k] = S(0.0); pal[k] = S(with_gravity ? -m.gravity[k] : 0.0); paw[k] = S(0.0); }
    }
    // actInv: lin' = R^T (lin - p x ang), ang' = R^T ang
    S t[3], u[3];
    cross(pv, pw, t);
    for (int k = 0; k < 3; ++k) u[k] = pl[k] - t[k];
    matTvec3(Ri, u, &vl[3 * i]);
    matTvec3(Ri, pw, &vw[3 * i]);
    cross(pv, paw, t);
    for (int k = 0; k < 3; ++k) u[k] = pal[k] - t[k];
    matTvec3(Ri, u, &al[3 * i]);
    matTvec3(Ri, paw, &aw[3 * i]);
    // joint motion S qd = (0; axis qd)
    S jw[3] = {ax[0] * v[i], ax[1] * v[i], ax[2] * v[i]};
    for (int k = 0; k < 3; ++k) vw[3 * i + k] += jw[k];
    // a += S qdd + v x (S qd): motion cross (vl,vw) x (0,jw) = (vl x jw ; vw x jw)
    S c1[3], c2[3];
    cross(&vl[3 * i], jw, c1);
    cross(&vw[3 * i], jw, c2);
    for (int k = 0; k < 3; ++k) { al[3 * i + k] += c1[k]; aw[3 * i + k] += ax[k] * a[i] + c2[k]; }
    // f = I a + v x* (I v)
    const double mass = m.mass[i];
    const double *c = &m.com[3 * i];
    const double *I = &m.inertia[9 * i];
    S cv[3] = {S(c[0]), S(c[1]), S(c[2])};
    S Iv[9];
    for (int k = 0; k < 9; ++k) Iv[k] = S(I[k]);
    S hl[3], hw[3], gl[3], gw[3], tmp[3], tmp2[3];
    // h = I v
    cross(cv, &vw[3 * i], tmp);
    for (int k = 0; k < 3; ++k) hl[k] = mass * (vl[3 * i + k] - tmp[k]);
    matvec3(Iv, &vw[3 * i], tmp);
    cross(cv, hl, tmp2);
    for (int k = 0; k < 3; ++k) hw[k] = tmp[k] + tmp2[k];
    // g = I a
    cross(cv, &aw[3 * i], tmp);
    for (int k = 0; k < 3; ++k) gl[k] = mass * (al[3 * i + k] - tmp[k]);
    matvec3(Iv, &aw[3 * i], tmp);
    cross(cv, gl, tmp2);
    for (int k = 0; k < 3; ++k) gw[k] = tmp[k] + tmp2[k];
    // v x* h = (w x hl ; w x hw + vl x hl)
    S x1[3], x2[3], x3[3];
    cross(&vw[3 * i], hl, x1);
    cross(&vw[3 * i], hw, x2);
    cross(&vl[3 * i], hl, x3);
    for (int k = 0; k < 3; ++k) { fl[3 * i + k] = gl[k] + x1[k]; fw[3 * i + k] = gw[k] + x2[k] + x3[k]; }
  }
  for (int i = n - 1; i >= 0; --i) {
    const double *ax = &m.axis[3 * i];
    tau[i] = ax[0] * fw[3 * i] + ax[1] * fw[3 * i + 1] + ax[2] * fw[3 * i + 2];
    int par = m.parent[i];
    if (par >= 0) {
      const double *p = &m.placement[12 * i + 9];
      S pv[3] = {S(p[0]), S(p[1]), S(p[2])};
      S Rf[3], Rn[3], t[3];
      matvec3(&R[9 * i], &fl[3 * i], Rf);
      matvec3(&R[9 * i], &fw[3 * i], Rn);
      cross(pv, Rf, t);
      for (int k = 0; k < 3; ++k) { fl[3 * par + k] += Rf[k]; fw[3 * par + k] += Rn[k] + t[k]; }
    }
  }
}

// Joint-space inertia by unit accelerations (column j = rnea(q, 0, e_j) without
// gravity), plus armature on the diagonal: what crocoddyl's armature branch of
// DifferentialActionModelFreeFwdDynamics factorises.
template <class S> void mass_matrix(const Model &m, const S *q, S *M) {
  const int n = m.nv;
  static thread_local std::vector<S> zero, e, col;
  zero.assign(n, S(0.0)); e.assign(n, S(0.0)); col.resize(n);
  for (int j = 0; j < n; ++j) {
    e[j] = S(1.0);
    rnea(m, q, zero.data(), e.data(), col.data(), false);
    e[j] = S(0.0);
    for (int i = 0; i < n; ++i) M[i * n + j] = col[i];
  }
  for (int i = 0; i < n; ++i) M[i * n + i] = M[i * n + i] + m.armature[i];
}

// In-place Cholesky (lower) and solve; generic scalar.
template <class S> void cholesky(int n, S *A) {
  for (int j = 0; j < n; ++j) {
    S d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      S s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
}
template <class S> void chol_solve(int n, const S *L, S *b) {
  for (int i = 0; i < n; ++i) {
    S s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    S s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

// a = (M + diag(armature))^-1 (u - nle(q, v))     
template <class S> void forward_dynamics(const Model &m, const S *q, const S *v, const S *u, S *a) {
  const int n = m.nv;
  static thread_local std::vector<S> M, nle, zero;
  M.resize(n * n); nle.resize(n); zero.assign(n, S(0.0));
  mass_matrix(m, q, M.data());
  rnea(m, q, v, zero.data(), nle.data(), true);
  for (int i = 0; i < n; ++i) a[i] = u[i] - nle[i];
  cholesky(n, M.data());
  chol_solve(n, M.data(), a);
}

// World placement of every joint, then of one frame.
template <class S> void joint_placements(const Model &m, const S *q, S *Rw, S *pw) {
  for (int i = 0; i < m.nv; ++i) {
    S Ri[9];
    joint_rotation(m, i, q[i], Ri);
    const double *p = &m.placement[12 * i + 9];
    S pv[3] = {S(p[0]), S(p[1]), S(p[2])};
    int par = m.parent[i];
    if (par >= 0) {
      matmul3(&Rw[9 * par], Ri, &Rw[9 * i]);
      S t[3];
      matvec3(&Rw[9 * par], pv, t);
      for (int k = 0; k < 3; ++k) pw[3 * i + k] = pw[3 * par + k] + t[k];
    } else {
      for (int k = 0; k < 9; ++k) Rw[9 * i + k] = Ri[k];
      for (int k = 0; k < 3; ++k) pw[3 * i + k] = pv[k];
    }
  }
}
template <class S> void frame_placement(const Model &m, int frame, const S *q, S *R, S *p) {
  static thread_local std::vector<S> Rw, pw;
  Rw.resize(9 * m.nv); pw.resize(3 * m.nv);
  joint_placements(m, q, Rw.data(), pw.data());
  const double *fp = &m.frame_placement[12 * frame];
  int par = m.frame_parent[frame];
  S Rf[9], pf[3];
  for (int k = 0; k < 9; ++k) Rf[k] = S(fp[k]);
  for (int k = 0; k < 3; ++k) pf[k] = S(fp[9 + k]);
  if (par >= 0) {
    matmul3(&Rw[9 * par], Rf, R);
    S t[3];
    matvec3(&Rw[9 * par], pf, t);
    for (int k = 0; k < 3; ++k) p[k] = pw[3 * par + k] + t[k];
  } else {
    for (int k = 0; k < 9; ++k) R[k] = Rf[k];
    for (int k = 0; k < 3; ++k) p[k] = pf[k];
  }
}

// log maps with pinocchio's conventions (pinocchio/spatial/log.hxx), used by
// crocoddyl::ResidualModelFramePlacement: r = log6(Mref^-1 oMf), [lin; ang],
// Rq = Jlog6(rMf) * fJf.  The branch for theta near pi (explicit formula with the
// sign taken from the antisymmetric part) matters: at q = 0 the Panda tool frame
// is exactly pi away from the identity, and the branch decides which way the
// reference's golden solution (tests/test_ocp_croco_base.py) turns the wrist.
inline void log3d(const double *R, double *w) {
  double tr = R[0] + R[4] + R[8];
  if (tr > 3.0) tr = 3.0;
  if (tr < -1.0) tr = -1.0;
  const double ct = 0.5 * (tr - 1.0);
  const double theta = std::acos(ct);
  if (theta >= M_PI - 1e-2) {
    const double cphi = -ct;
    const double beta = theta * theta / (1.0 + cphi);
    const double v0 = (R[0] + cphi) * beta, v1 = (R[4] + cphi) * beta, v2 = (R[8] + cphi) * beta;
    w[0] = (R[7] > R[5] ? 1.0 : -1.0) * (v0 > 0.0 ? std::sqrt(v0) : 0.0);
    w[1] = (R[2] > R[6] ? 1.0 : -1.0) * (v1 > 0.0 ? std::sqrt(v1) : 0.0);
    w[2] = (R[3] > R[1] ? 1.0 : -1.0) * (v2 > 0.0 ? std::sqrt(v2) : 0.0);
  } else {
    const double t = 0.5 * (theta > 1e-8 ? theta / std::sin(theta) : 1.0);
    w[0] = t * (R[7] - R[5]);
    w[1] = t * (R[2] - R[6]);
    w[2] = t * (R[3] - R[1]);
  }
}
inline void log6d(const double *R, const double *p, double *r) {
  double w[3];
  log3d(R, w);
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double alpha, beta;
  if (t2 < 1e-12) {
    alpha = 1.0 - t2 / 12.0;
    beta = 1.0 / 12.0 + t2 / 720.0;
  } else {
    const double t = std::sqrt(t2), st = std::sin(t), ct = std::cos(t);
    alpha = t * st / (2.0 * (1.0 - ct));
    beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
  }
  double wxp[3];
  cross(w, p, wxp);
  const double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  for (int k = 0; k < 3; ++k) {
    r[k] = alpha * p[k] - 0.5 * wxp[k] + beta * wp * w[k];
    r[3 + k] = w[k];
  }
}
// Jlog3 / Jlog6 for a right (local) perturbation, pinocchio::Jlog3 / Jlog6.
inline void jlog3d(const double *w, double *J) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  double alpha, diag;
  if (t < 1e-4) {
    alpha = 1.0 / 12.0 + t2 / 720.0;
    diag = 0.5 * (2.0 - t2 / 6.0);
  } else {
    const double st = std::sin(t), ct = std::cos(t), s1 = st / (1.0 - ct);
    alpha = 1.0 / t2 - s1 / (2.0 * t);
    diag = 0.5 * t * s1;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) J[3 * i + j] = alpha * w[i] * w[j] + (i == j ? diag : 0.0);
  J[1] -= 0.5 * w[2]; J[2] += 0.5 * w[1];
  J[3] += 0.5 * w[2]; J[5] -= 0.5 * w[0];
  J[6] -= 0.5 * w[1]; J[7] += 0.5 * w[0];
}
inline void jlog6d(const double *R, const double *p, double *J /*6x6*/) {
  double w[3], TL[9];
  log3d(R, w);
  jlog3d(w, TL);
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  double beta, bdot;
  if (t < 1e-4) {
    beta = 1.0 / 12.0 + t2 / 720.0;
    bdot = 1.0 / 360.0;
  } else {
    const double tinv = 1.0 / t, t2inv = tinv * tinv, st = std::sin(t), ct = std::cos(t), i22 = 1.0 / (2.0 * (1.0 - ct));
    beta = t2inv - st * tinv * i22;
    bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * i22;
  }
  const double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  double v3[3], Cm[9], TR[9];
  for (int k = 0; k < 3; ++k) v3[k] = (bdot * wp) * w[k] - (t2 * bdot + 2.0 * beta) * p[k];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Cm[3 * i + j] = v3[i] * w[j] + beta * w[i] * p[j] + (i == j ? wp * beta : 0.0);
  Cm[1] -= 0.5 * p[2]; Cm[2] += 0.5 * p[1];
  Cm[3] += 0.5 * p[2]; Cm[5] -= 0.5 * p[0];
  Cm[6] -= 0.5 * p[1]; Cm[7] += 0.5 * p[0];
  matmul3(Cm, TL, TR);
  for (int i = 0; i < 36; ++i) J[i] = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { J[6 * i + j] = TL[3 * i + j]; J[6 * (i + 3) + 3 + j] = TL[3 * i + j]; J[6 * i + 3 + j] = TR[3 * i + j]; }
}

// residual value through the log map, tangents through Jlog * (local twist of the
// AD-propagated placement): exactly crocoddyl's Rq = Jlog6(rMf) * fJf, with the
// frame Jacobian obtained by automatic differentiation of the forward kinematics.
inline void log6_residual(const double *R, const double *p, double *r) { log6d(R, p, r); }
inline void log3_residual(const double *R, double *r) { log3d(R, r); }
template <int N> void log6_residual(const Dual<N> *R, const Dual<N> *p, Dual<N> *r) {
  double Rv[9], pv[3], rv[6], J[36];
  for (int k = 0; k < 9; ++k) Rv[k] = R[k].v;
  for (int k = 0; k < 3; ++k) pv[k] = p[k].v;
  log6d(Rv, pv, rv);
  jlog6d(Rv, pv, J);
  for (int k = 0; k < 6; ++k) r[k] = Dual<N>(rv[k]);
  for (int d = 0; d < N; ++d) {
    double dR[9], dp[3], W[9], xi[6];
    for (int k = 0; k < 9; ++k) dR[k] = R[k].d[d];
    for (int k = 0; k < 3; ++k) dp[k] = p[k].d[d];
    matTvec3(Rv, dp, xi);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) W[3 * i + j] = Rv[i] * dR[j] + Rv[3 + i] * dR[3 + j] + Rv[6 + i] * dR[6 + j];
    xi[3] = 0.5 * (W[7] - W[5]); xi[4] = 0.5 * (W[2] - W[6]); xi[5] = 0.5 * (W[3] - W[1]);
    for (int i = 0; i < 6; ++i) { double s = 0.0; for (int j = 0; j < 6; ++j) s += J[6 * i + j] * xi[j]; r[i].d[d] = s; }
  }
}
template <int N> void log3_residual(const Dual<N> *R, Dual<N> *r) {
  double Rv[9], rv[3], J[9];
  for (int k = 0; k < 9; ++k) Rv[k] = R[k].v;
  log3d(Rv, rv);
  jlog3d(rv, J);
  for (int k = 0; k < 3; ++k) r[k] = Dual<N>(rv[k]);
  for (int d = 0; d < N; ++d) {
    double dR[9], W[9], xi[3];
    for (int k = 0; k < 9; ++k) dR[k] = R[k].d[d];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) W[3 * i + j] = Rv[i] * dR[j] + Rv[3 + i] * dR[3 + j] + Rv[6 + i] * dR[6 + j];
    xi[0] = 0.5 * (W[7] - W[5]); xi[1] = 0.5 * (W[2] - W[6]); xi[2] = 0.5 * (W[3] - W[1]);
    for (int i = 0; i < 3; ++i) r[i].d[d] = J[3 * i] * xi[0] + J[3 * i + 1] * xi[1] + J[3 * i + 2] * xi[2];
  }
}

// Velocity of a frame (pinocchio::getFrameVelocity): [linear; angular] in
//   type 0 WORLD                the spatial velocity seen at the world origin, world axes
//   type 1 LOCAL                velocity of the frame origin / angular velocity in the frame's own axes
//   type 2 LOCAL_WORLD_ALIGNED  velocity of the frame origin / angular velocity in world axes
// by the forward recursion  w_i = w_p + z_i qd_i,  v_i = v_p + w_p x (p_i - p_p)  in world coordinates.
template <class S> void frame_velocity(const Model &m, int frame, int type, const S *q, const S *qd, S *out6) {
  static thread_local std::vector<S> Rw, pw, wv, vv;
  const int nv = m.nv;
  Rw.resize(9 * nv); pw.resize(3 * nv); wv.resize(3 * nv); vv.resize(3 * nv);
  joint_placements(m, q, Rw.data(), pw.data());
  for (int i = 0; i < nv; ++i) {
    const int par = m.parent[i];
    S wp[3] = {S(0.0), S(0.0), S(0.0)}, vp[3] = {S(0.0), S(0.0), S(0.0)}, d[3];
    for (int k = 0; k < 3; ++k) d[k] = pw[3 * i + k];
    if (par >= 0)
      for (int k = 0; k < 3; ++k) { wp[k] = wv[3 * par + k]; vp[k] = vv[3 * par + k]; d[k] = pw[3 * i + k] - pw[3 * par + k]; }
    S c[3];
    cross(wp, d, c);
    const double *ax = &m.axis[3 * i];
    S axs[3] = {S(ax[0]), S(ax[1]), S(ax[2])}, z[3];
    matvec3(&Rw[9 * i], axs, z);
    for (int k = 0; k < 3; ++k) { vv[3 * i + k] = vp[k] + c[k]; wv[3 * i + k] = wp[k] + z[k] * qd[i]; }
  }
  S R[9], p[3];
  frame_placement(m, frame, q, R, p);  // (recomputes the joint placements; the checker favours clarity)
  const int par = m.frame_parent[frame];
  S w[3] = {S(0.0), S(0.0), S(0.0)}, v[3] = {S(0.0), S(0.0), S(0.0)};
  if (par >= 0) {
    S d[3], c[3];
    for (int k = 0; k < 3; ++k) { w[k] = wv[3 * par + k]; d[k] = p[k] - pw[3 * par + k]; }
    cross(w, d, c);
    for (int k = 0; k < 3; ++k) v[k] = vv[3 * par + k] + c[k];
  }
  if (type == 0) {
    S c[3];
    cross(w, p, c);  // v_O = v_f - w x p_f
    for (int k = 0; k < 3; ++k) { out6[k] = v[k] - c[k]; out6[3 + k] = w[k]; }
  } else if (type == 1) {
    matTvec3(R, v, out6);
    matTvec3(R, w, out6 + 3);
  } else {
    for (int k = 0; k < 3; ++k) { out6[k] = v[k]; out6[3 + k] = w[k]; }
  }
}

// ---------------------------------------------------------------------------
// Closest points of two segments a0 + s (a1 - a0), b0 + t (b1 - b0), s, t in [0, 1]
// (Ericson, Real-Time Collision Detection 5.1.9 -- the capsule/capsule narrow phase of coal that
// colmpc.ResidualDistanceCollision goes through after factory/robot_model.py:261-302 turned the
// arm links into capsules).  Degenerate segments (spheres) are handled.
// ---------------------------------------------------------------------------
inline double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
void closest_seg_seg(const double *a0, const double *a1, const double *b0, const double *b1, double &s, double &t) {
  const double eps = 1e-14;
  double d1[3], d2[3], r[3];
  for (int k = 0; k < 3; ++k) { d1[k] = a1[k] - a0[k]; d2[k] = b1[k] - b0[k]; r[k] = a0[k] - b0[k]; }
  const double a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
  const double e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
  const double f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2];
  if (a <= eps && e <= eps) { s = 0.0; t = 0.0; return; }
  if (a <= eps) { s = 0.0; t = clamp01(f / e); return; }
  const double c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2];
  if (e <= eps) { t = 0.0; s = clamp01(-c / a); return; }
  const double b = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
  const double denom = a * e - b * b;
  s = (denom > eps * a * e) ? clamp01((b * f - c * e) / denom) : 0.0;
  t = (b * s + f) / e;
  if (t < 0.0) { t = 0.0; s = clamp01(-c / a); }
  else if (t > 1.0) { t = 1.0; s = clamp01((b - c) / a); }
}

// parameter s in [-h, h] of the point of the segment c + s d closest to the box |x_i| <= b_i (box
// frame): dist^2 is convex along the segment, bisection on its monotone derivative
double seg_box_param(const double *c, const double *d, double h, const double *b) {
  auto fp = [&](double s) {
    double g = 0.0;
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

// signed distance of two geometry frames: sgn |pa - pb| - ra - rb with pa, pb the witness points
// (closest points of the capsule segments; for a box against a capsule / sphere the closest point of
// the segment and its clamp on the box, or -- segment point inside the box, sgn = -1 -- its projection
// on the nearest face).  The witness points are held fixed in their bodies while differentiating,
// which gives exactly  d'(q) = n' (Ja(pa) - Jb(pb))  (App. A.6).
template <class S> S collision_distance(const Model &m, int fa, int fb, const S *q) {
  S Ra[9], pa[3], Rb[9], pb[3];
  frame_placement(m, fa, q, Ra, pa);
  frame_placement(m, fb, q, Rb, pb);
  const double ha = m.frame_halflen[fa], hb = m.frame_halflen[fb];
  const bool box_a = m.frame_box[3 * fa] > 0.0, box_b = m.frame_box[3 * fb] > 0.0;
  S d2 = S(0.0);
  double sgn = 1.0;
  if (box_a || box_b) {
    const S *Rx = box_b ? Rb : Ra, *px = box_b ? pb : pa, *Rc = box_b ? Ra : Rb, *pc = box_b ? pa : pb;
    const double *half = &m.frame_box[3 * (box_b ? fb : fa)];
    const double hc = box_b ? ha : hb;
    double cl[3], dl[3];
    for (int i = 0; i < 3; ++i) {
      cl[i] = 0.0; dl[i] = 0.0;
      for (int k = 0; k < 3; ++k) {
        cl[i] += val(Rx[3 * k + i]) * (val(pc[k]) - val(px[k]));
        dl[i] += val(Rx[3 * k + i]) * val(Rc[3 * k + 2]);
      }
    }
    const double s = seg_box_param(cl, dl, hc, half);
    double x[3], y[3];
    bool inside = true;
    for (int i = 0; i < 3; ++i) {
      x[i] = cl[i] + s * dl[i];
      y[i] = x[i] < -half[i] ? -half[i] : (x[i] > half[i] ? half[i] : x[i]);
      inside = inside && (y[i] == x[i]);
    }
    if (inside) {
      int best = 0;
      double depth = half[0] - fabs(x[0]);
      for (int i = 1; i < 3; ++i)
        if (half[i] - fabs(x[i]) < depth) { depth = half[i] - fabs(x[i]); best = i; }
      y[best] = x[best] < 0.0 ? -half[best] : half[best];
      sgn = -1.0;
    }
    for (int k = 0; k < 3; ++k) {
      S cc = pc[k] + s * Rc[3 * k + 2];
      S cx = px[k] + (Rx[3 * k + 0] * y[0] + Rx[3 * k + 1] * y[1] + Rx[3 * k + 2] * y[2]);
      S e = cc - cx;
      d2 = d2 + e * e;
    }
  } else {
    double a0[3], a1[3], b0[3], b1[3];
    for (int k = 0; k < 3; ++k) {
      a0[k] = val(pa[k]) - ha * val(Ra[3 * k + 2]); a1[k] = val(pa[k]) + ha * val(Ra[3 * k + 2]);
      b0[k] = val(pb[k]) - hb * val(Rb[3 * k + 2]); b1[k] = val(pb[k]) + hb * val(Rb[3 * k + 2]);
    }
    double sa, sb;
    closest_seg_seg(a0, a1, b0, b1, sa, sb);
    for (int k = 0; k < 3; ++k) {
      S ca = pa[k] + ((2.0 * sa - 1.0) * ha) * Ra[3 * k + 2];
      S cb = pb[k] + ((2.0 * sb - 1.0) * hb) * Rb[3 * k + 2];
      S e = ca - cb;
      d2 = d2 + e * e;
    }
  }
  return sgn * sqrt(d2) - (m.frame_radius[fa] + m.frame_radius[fb]);
}

// ---------------------------------------------------------------------------
// problem description
// ---------------------------------------------------------------------------
struct Ocp {
  int T = 0;
  std::vector<double> dt;
  std::vector<agx_cost_row> rows[2];  // 0 running, 1 terminal
  std::vector<int> row_off[2];
  int stride = 0;
  double tol = 1e-3, mu_dyn = 10.0, mu_con = 10.0;
  int max_qp = 200;
  double eps_abs = 1e-6, eps_rel = 0.0;
  bool use_filter = false;  // SolverCSQP.use_filter_line_search
  // ConstraintListItem rows (lower <= r(x,u) <= upper), 0 running / 1 terminal
  struct ConRow { int kind, frame, frame_b; std::vector<double> ref, lower, upper; };
  std::vector<ConRow> cons[2];
  int nc[2] = {0, 0};
};

}  // namespace

extern "C" {
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
  return std::max(s0, s1);
}
}

namespace {

void copy_ocp(const agx_ocp_desc *d, int nv, Ocp &o) {
  o.T = d->horizon;
  o.dt.assign(d->dt, d->dt + d->horizon);
  o.rows[0].assign(d->running_rows, d->running_rows + d->n_running_rows);
  o.rows[1].assign(d->terminal_rows, d->terminal_rows + d->n_terminal_rows);
  for (int s = 0; s < 2; ++s) {
    int off = 0;
    o.row_off[s].clear();
    for (auto &r : o.rows[s]) { o.row_off[s].push_back(off); off += 1 + agx_row_nref(r.kind, nv) + agx_row_nr(r.kind, nv); }
  }
  o.stride = agx_ref_stride(d, nv);
  o.tol = d->termination_tolerance;
  o.mu_dyn = d->mu_dynamic;
  o.mu_con = d->mu_constraint;
  o.max_qp = d->max_qp_iters;
  o.eps_abs = d->eps_abs;
  o.eps_rel = d->eps_rel;
  o.use_filter = d->use_filter_line_search != 0;
  for (int s = 0; s < 2; ++s) {
    const int n = s == 0 ? d->n_running_constraints : d->n_terminal_constraints;
    const agx_constraint_row *rows = s == 0 ? d->running_constraints : d->terminal_constraints;
    o.cons[s].clear();
    o.nc[s] = 0;
    for (int i = 0; i < n; ++i) {
      if (!rows[i].active) continue;
      Ocp::ConRow c;
      c.kind = rows[i].kind; c.frame = rows[i].frame; c.frame_b = rows[i].frame_b;
      const int nref = agx_row_nref(c.kind, nv), nr = agx_row_nr(c.kind, nv);
      c.ref.assign(nref, 0.0);
      if (rows[i].ref) c.ref.assign(rows[i].ref, rows[i].ref + nref);
      if (c.kind == AGX_RES_FRAME_PLACEMENT || c.kind == AGX_RES_FRAME_ROTATION)
        if (!rows[i].ref) { c.ref[0] = c.ref[4] = c.ref[8] = 1.0; }
      c.lower.assign(rows[i].lower, rows[i].lower + nr);
      c.upper.assign(rows[i].upper, rows[i].upper + nr);
      o.nc[s] += nr;
      o.cons[s].push_back(c);
    }
  }
}

// ---------------------------------------------------------------------------
// One shooting node: IntegratedActionModelEuler(DifferentialActionModelFreeFwdDynamics)
// calc (S = double) or calc + calcDiff (S = Dual) -- SURVEY App. A.1-A.3.
// ---------------------------------------------------------------------------
struct NodeOut {  // plain doubles
  std::vector<double> xnext, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu;
  double cost = 0.0;
  std::vector<double> residuals;  // concatenated per-row residual values
};

// One residual of the YAML schema (ocp_croco_generic.py:147-550) at (x, u): shared by cost rows and
// constraint rows.  r has agx_row_nr(kind) entries.
template <class S>
void eval_residual(const Model &m, int kind, int frame, int frame_b, const double *rref, bool terminal, const S *x,
                   const S *u, S *rout) {
  const int nv = m.nv;
  const S *q = x;
  const int nr = agx_row_nr(kind, nv);
  S *r = rout;
    switch (kind) {
      case AGX_RES_STATE:
        for (int i = 0; i < 2 * nv; ++i) r[i] = x[i] - rref[i];
        break;
      case AGX_RES_CONTROL:
        for (int i = 0; i < nv; ++i) r[i] = terminal ? S(0.0) : (u[i] - rref[i]);
        break;
      case AGX_RES_CONTROL_GRAV: {
        std::vector<S> zero(nv, S(0.0)), g(nv);
        rnea(m, q, zero.data(), zero.data(), g.data(), true);
        for (int i = 0; i < nv; ++i) r[i] = terminal ? S(0.0) : (u[i] - g[i]);
      } break;
      case AGX_RES_FRAME_PLACEMENT: {
        S R[9], p[3];
        frame_placement(m, frame, q, R, p);
        S Rr[9], pr[3];
        for (int k = 0; k < 9; ++k) Rr[k] = S(rref[k]);
        for (int k = 0; k < 3; ++k) pr[k] = S(rref[9 + k]);
        // rMf = Mref^-1 * oMf
        S Rt[9] = {Rr[0], Rr[3], Rr[6], Rr[1], Rr[4], Rr[7], Rr[2], Rr[5], Rr[8]};
        S Rrel[9], d[3], prel[3];
        matmul3(Rt, R, Rrel);
        for (int k = 0; k < 3; ++k) d[k] = p[k] - pr[k];
        matvec3(Rt, d, prel);
        log6_residual(Rrel, prel, r);
      } break;
      case AGX_RES_FRAME_TRANSLATION: {
        S R[9], p[3];
        frame_placement(m, frame, q, R, p);
        for (int k = 0; k < 3; ++k) r[k] = p[k] - rref[k];
      } break;
      case AGX_RES_FRAME_ROTATION: {
        S R[9], p[3];
        frame_placement(m, frame, q, R, p);
        S Rt[9] = {S(rref[0]), S(rref[3]), S(rref[6]), S(rref[1]), S(rref[4]), S(rref[7]), S(rref[2]), S(rref[5]), S(rref[8])};
        S Rrel[9];
        matmul3(Rt, R, Rrel);
        log3_residual(Rrel, r);
      } break;
      case AGX_RES_FRAME_VELOCITY: {
        // crocoddyl ResidualModelFrameVelocity (ocp_croco_generic.py:360-432): frame velocity in the
        // chosen reference frame (frame_b: 0 WORLD, 1 LOCAL, 2 LOCAL_WORLD_ALIGNED) minus the reference twist
        S vf[6];
        frame_velocity(m, frame, frame_b, q, x + nv, vf);
        for (int k = 0; k < 6; ++k) r[k] = vf[k] - rref[k];
      } break;
      case AGX_RES_COLLISION:
        // colmpc.ResidualDistanceCollision (ocp_croco_generic.py:524-533): signed distance of the pair
        r[0] = collision_distance(m, frame, frame_b, q);
        break;
      default:
        for (int i = 0; i < nr; ++i) r[i] = S(0.0);
        break;
    }
}

template <class S>
void node_eval(const Model &m, const Ocp &o, bool terminal, double dt, const S *x, const S *u,
               const double *ref, const int32_t *frames, S *xnext, std::vector<S> &res,
               std::vector<double> &act_r, std::vector<double> &act_rr, std::vector<double> &row_w,
               std::vector<int> &row_nr, double &cost) {
  const int nv = m.nv;
  const S *q = x, *v = x + nv;
  if (!terminal) {
    static thread_local std::vector<S> a;
    a.resize(nv);
    forward_dynamics(m, q, v, u, a.data());
    // semi-implicit Euler, crocoddyl IntegratedActionModelEuler::calc (App. A.2)
    for (int i = 0; i < nv; ++i) {
      S vn = v[i] + dt * a[i];
      xnext[nv + i] = vn;
      xnext[i] = q[i] + dt * v[i] + (dt * dt) * a[i];
    }
  } else {
    for (int i = 0; i < 2 * nv; ++i) xnext[i] = x[i];
  }
  const auto &rows = o.rows[terminal ? 1 : 0];
  const auto &offs = o.row_off[terminal ? 1 : 0];
  res.clear(); act_r.clear(); act_rr.clear(); row_w.clear(); row_nr.clear();
  cost = 0.0;
  for (size_t ri = 0; ri < rows.size(); ++ri) {
    const agx_cost_row &row = rows[ri];
    const double *tile = ref + offs[ri];
    const double w_item = tile[0];
    const double *rref = tile + 1;
    const int nref = agx_row_nref(row.kind, nv);
    const int nr = agx_row_nr(row.kind, nv);
    const double *aw = rref + nref;
    int frame = frames ? frames[ri] : row.frame;
    if (frame < 0) frame = row.frame;
    static thread_local std::vector<S> r;
    r.resize(nr);
    eval_residual(m, row.kind, frame, row.frame_b, rref, terminal, x, u, r.data());
    double a_val = 0.0;
    if (row.activation == AGX_ACT_WEIGHTED_QUAD) {
      for (int j = 0; j < nr; ++j) {
        double rv = val(r[j]);
        // ActivationModelWeightedQuad: a = 1/2 sum w r^2 (tests/test_ocp_croco_generic.py:70-72)
        a_val += 0.5 * aw[j] * rv * rv;
        act_r.push_back(aw[j] * rv);
        act_rr.push_back(aw[j]);
      }
    } else {
      // colmpc activations (ocp_croco_generic.py:118-143), restated from recall (parity
      // unpinned): QuadExp a = exp(-|r|^2 / alpha), Exp a = exp(-|r| / alpha); diagonal second derivative.
      double n2 = 0.0;
      for (int j = 0; j < nr; ++j) n2 += val(r[j]) * val(r[j]);
      const double al = row.alpha;
      if (row.activation == AGX_ACT_QUAD_EXP) {
        a_val = std::exp(-n2 / al);
        for (int j = 0; j < nr; ++j) {
          const double rv = val(r[j]);
          act_r.push_back(-2.0 * rv * a_val / al);
          act_rr.push_back((-2.0 / al + 4.0 * rv * rv / (al * al)) * a_val);
        }
      } else {
        const double n = std::sqrt(n2);
        a_val = std::exp(-n / al);
        for (int j = 0; j < nr; ++j) {
          const double rv = val(r[j]);
          act_r.push_back(n > 0.0 ? -a_val / al * rv / n : 0.0);
          act_rr.push_back(a_val / (al * al));
        }
      }
    }
    const bool active = row.active != 0;
    row_w.push_back(active ? w_item : 0.0);
    row_nr.push_back(nr);
    if (active) cost += w_item * a_val;
    for (int j = 0; j < nr; ++j) res.push_back(r[j]);
  }
  if (!terminal) cost *= dt;
}

template <int N>
void node_calc_diff_n(const Model &m, const Ocp &o, bool terminal, double dt, const double *x,
                      const double *u, const double *ref, const int32_t *frames, NodeOut &out) {
  typedef Dual<N> D;
  const int nv = m.nv, nx = 2 * nv, nu = nv;
  static thread_local std::vector<D> xd, ud, xn, res;
  static thread_local std::vector<double> ar, arr, rw;
  static thread_local std::vector<int> rnr;
  xd.resize(nx); ud.resize(nu); xn.resize(nx); res.reserve(64);
  for (int i = 0; i < nx; ++i) { xd[i] = D(x[i]); xd[i].d[i] = 1.0; }
  for (int i = 0; i < nu; ++i) { ud[i] = D(terminal ? 0.0 : u[i]); ud[i].d[nx + i] = 1.0; }
  double cost;
  node_eval<D>(m, o, terminal, dt, xd.data(), ud.data(), ref, frames, xn.data(), res, ar, arr, rw, rnr, cost);
  out.cost = cost;
  out.xnext.resize(nx); out.Fx.assign(nx * nx, 0.0); out.Fu.assign(nx * nu, 0.0);
  out.Lx.assign(nx, 0.0); out.Lu.assign(nu, 0.0); out.Lxx.assign(nx * nx, 0.0); out.Lxu.assign(nx * nu, 0.0); out.Luu.assign(nu * nu, 0.0);
  for (int i = 0; i < nx; ++i) {
    out.xnext[i] = xn[i].v;
    for (int j = 0; j < nx; ++j) out.Fx[i * nx + j] = xn[i].d[j];
    for (int j = 0; j < nu; ++j) out.Fu[i * nu + j] = terminal ? 0.0 : xn[i].d[nx + j];
  }
  const double scale = terminal ? 1.0 : dt;
  out.residuals.resize(res.size());
  int k = 0;
  for (size_t ri = 0; ri < rnr.size(); ++ri) {
    const double w = rw[ri] * scale;
    for (int j = 0; j < rnr[ri]; ++j, ++k) {
      out.residuals[k] = res[k].v;
      if (w == 0.0) continue;
      const double *g = res[k].d;  // [d/dx (nx) | d/du (nu)]
      // Gauss-Newton: Lx = Rx^T a_r, Lxx = Rx^T a_rr Rx
      for (int a = 0; a < nx; ++a) {
        out.Lx[a] += w * ar[k] * g[a];
        for (int b = 0; b < nx; ++b) out.Lxx[a * nx + b] += w * arr[k] * g[a] * g[b];
        for (int b = 0; b < nu; ++b) out.Lxu[a * nu + b] += w * arr[k] * g[a] * g[nx + b];
      }
      for (int a = 0; a < nu; ++a) {
        out.Lu[a] += w * ar[k] * g[nx + a];
        for (int b = 0; b < nu; ++b) out.Luu[a * nu + b] += w * arr[k] * g[nx + a] * g[nx + b];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Constraints of one node: g(x,u) stacked over the active ConstraintListItem rows, bounds, and
// (DIFF) the Jacobians Gx [nc][nx], Gu [nc][nu]  (crocoddyl ConstraintModelManager; ocp_croco_generic.py:554-647).
// ---------------------------------------------------------------------------
struct ConNode {
  int nc = 0;
  std::vector<double> g, Gx, Gu, lb, ub;
};
template <class S>
void eval_constraints(const Model &m, const Ocp &o, bool terminal, const S *x, const S *u, std::vector<S> &g) {
  const auto &rows = o.cons[terminal ? 1 : 0];
  g.clear();
  static thread_local std::vector<S> r;
  for (const auto &c : rows) {
    const int nr = agx_row_nr(c.kind, m.nv);
    r.resize(nr);
    eval_residual(m, c.kind, c.frame, c.frame_b, c.ref.data(), terminal, x, u, r.data());
    for (int j = 0; j < nr; ++j) g.push_back(r[j]);
  }
}
void con_bounds(const Ocp &o, bool terminal, ConNode &n) {
  n.lb.clear(); n.ub.clear();
  for (const auto &c : o.cons[terminal ? 1 : 0]) {
    n.lb.insert(n.lb.end(), c.lower.begin(), c.lower.end());
    n.ub.insert(n.ub.end(), c.upper.begin(), c.upper.end());
  }
  n.nc = (int)n.lb.size();
}
template <int N>
void node_constraints_diff_n(const Model &m, const Ocp &o, bool terminal, const double *x, const double *u, ConNode &n) {
  typedef Dual<N> D;
  const int nv = m.nv, nx = 2 * nv, nu = nv;
  static thread_local std::vector<D> xd, ud, g;
  xd.resize(nx); ud.resize(nu);
  for (int i = 0; i < nx; ++i) { xd[i] = D(x[i]); xd[i].d[i] = 1.0; }
  for (int i = 0; i < nu; ++i) { ud[i] = D(terminal ? 0.0 : u[i]); ud[i].d[nx + i] = 1.0; }
  eval_constraints<D>(m, o, terminal, xd.data(), ud.data(), g);
  con_bounds(o, terminal, n);
  n.g.resize(n.nc); n.Gx.assign((size_t)n.nc * nx, 0.0); n.Gu.assign((size_t)n.nc * nu, 0.0);
  for (int k = 0; k < n.nc; ++k) {
    n.g[k] = g[k].v;
    for (int j = 0; j < nx; ++j) n.Gx[(size_t)k * nx + j] = g[k].d[j];
    for (int j = 0; j < nu; ++j) n.Gu[(size_t)k * nu + j] = terminal ? 0.0 : g[k].d[nx + j];
  }
}
void node_constraints_diff(const Model &m, const Ocp &o, bool terminal, const double *x, const double *u, ConNode &n) {
  const int nd = 3 * m.nv;
  if (nd <= 8) node_constraints_diff_n<8>(m, o, terminal, x, u, n);
  else if (nd <= 24) node_constraints_diff_n<24>(m, o, terminal, x, u, n);
  else node_constraints_diff_n<96>(m, o, terminal, x, u, n);
}
// l1 norm of the violation of lb <= g <= ub at one node (mim_solvers SolverCSQP::calc / tryStep)
double node_constraint_violation(const Model &m, const Ocp &o, bool terminal, const double *x, const double *u) {
  if (o.nc[terminal ? 1 : 0] == 0) return 0.0;
  static thread_local std::vector<double> g, uz;
  static thread_local ConNode n;
  uz.assign(m.nv, 0.0);
  eval_constraints<double>(m, o, terminal, x, terminal ? uz.data() : u, g);
  con_bounds(o, terminal, n);
  double v = 0.0;
  for (int k = 0; k < n.nc; ++k) v += std::max(n.lb[k] - g[k], 0.0) + std::max(g[k] - n.ub[k], 0.0);
  return v;
}

void node_calc_diff(const Model &m, const Ocp &o, bool terminal, double dt, const double *x,
                    const double *u, const double *ref, const int32_t *frames, NodeOut &out) {
  const int nd = 3 * m.nv;
  if (nd <= 8) node_calc_diff_n<8>(m, o, terminal, dt, x, u, ref, frames, out);
  else if (nd <= 24) node_calc_diff_n<24>(m, o, terminal, dt, x, u, ref, frames, out);
  else node_calc_diff_n<96>(m, o, terminal, dt, x, u, ref, frames, out);
}

void node_calc(const Model &m, const Ocp &o, bool terminal, double dt, const double *x, const double *u,
               const double *ref, const int32_t *frames, double *xnext, double &cost,
               std::vector<double> *residuals = nullptr) {
  static thread_local std::vector<double> res, ar, arr, rw, uz;
  static thread_local std::vector<int> rnr;
  uz.assign(m.nv, 0.0);
  node_eval<double>(m, o, terminal, dt, x, terminal ? uz.data() : u, ref, frames, xnext, res, ar, arr, rw, rnr, cost);
  if (residuals) *residuals = res;
}

void pack_tile(int nv, const NodeOut &n, const double *xs_next, double *tile) {
  const int nx = 2 * nv, nu = nv;
  double *p = tile;
  std::memcpy(p, n.Fx.data(), sizeof(double) * nx * nx); p += nx * nx;
  std::memcpy(p, n.Fu.data(), sizeof(double) * nx * nu); p += nx * nu;
  for (int i = 0; i < nx; ++i) p[i] = xs_next ? n.xnext[i] - xs_next[i] : 0.0;  // gap f_{t+1}
  p += nx;
  std::memcpy(p, n.Lx.data(), sizeof(double) * nx); p += nx;
  std::memcpy(p, n.Lu.data(), sizeof(double) * nu); p += nu;
  std::memcpy(p, n.Lxx.data(), sizeof(double) * nx * nx); p += nx * nx;
  std::memcpy(p, n.Lxu.data(), sizeof(double) * nx * nu); p += nx * nu;
  std::memcpy(p, n.Luu.data(), sizeof(double) * nu * nu); p += nu * nu;
  p[0] = n.cost;
}

// ---------------------------------------------------------------------------
// QP direction: Riccati backward + linear forward + KKT,
// the unconstrained branch of mim_solvers::SolverCSQP::computeDirection.
// tiles [T+1][TILE] in the layout of AGX_TILE_DOUBLES.
// ---------------------------------------------------------------------------
struct Direction {
  std::vector<double> K, k, dx, du, Vxx, Vx, lag;
  double kkt = 0.0;
};

// sigma > 0 selects the proximal (ADMM) form of mim_solvers::SolverCSQP::backwardPass:
// sigma on Vxx_T, Qxx, Quu and -sigma*dxtilde/-sigma*dutilde on the gradients (prox
// centre = the previous iterate cx, cu).  preg/dreg are crocoddyl's primal/dual
// regularisations (Quu += preg, Vxx += dreg), floor reg_min = 1e-9.
bool direction(int nv, int T, const double *tiles, Direction &d, double sigma = 0.0, double preg = 0.0,
               double dreg = 0.0, const double *cx = nullptr, const double *cu = nullptr) {
  const int nx = 2 * nv, nu = nv, TILE = AGX_TILE_DOUBLES(nv);
  const int oFx = 0, oFu = oFx + nx * nx, of = oFu + nx * nu, oLx = of + nx, oLu = oLx + nx, oLxx = oLu + nu, oLxu = oLxx + nx * nx, oLuu = oLxu + nx * nu;
  d.K.assign((size_t)T * nu * nx, 0.0); d.k.assign((size_t)T * nu, 0.0);
  d.dx.assign((size_t)(T + 1) * nx, 0.0); d.du.assign((size_t)T * nu, 0.0);
  d.Vxx.assign((size_t)(T + 1) * nx * nx, 0.0); d.Vx.assign((size_t)(T + 1) * nx, 0.0);
  d.lag.assign((size_t)(T + 1) * nx, 0.0);
  const double *tt = tiles + (size_t)T * TILE;
  std::memcpy(&d.Vxx[(size_t)T * nx * nx], tt + oLxx, sizeof(double) * nx * nx);
  std::memcpy(&d.Vx[(size_t)T * nx], tt + oLx, sizeof(double) * nx);
  for (int i = 0; i < nx; ++i) {
    d.Vxx[(size_t)T * nx * nx + i * nx + i] += sigma + dreg;
    if (cx) d.Vx[(size_t)T * nx + i] -= sigma * cx[(size_t)T * nx + i];
  }
  std::vector<double> Vp(nx), FxTV(nx * nx), FuTV(nu * nx), Qxx(nx * nx), Qxu(nx * nu), Quu(nu * nu), Qx(nx), Qu(nu), L(nu * nu), col(nu);
  bool ok = true;
  for (int t = T - 1; t >= 0; --t) {
    const double *n = tiles + (size_t)t * TILE;
    const double *Fx = n + oFx, *Fu = n + oFu, *f = n + of, *Lx = n + oLx, *Lu = n + oLu, *Lxx = n + oLxx, *Lxu = n + oLxu, *Luu = n + oLuu;
    const double *Vn = &d.Vxx[(size_t)(t + 1) * nx * nx];
    const double *vn = &d.Vx[(size_t)(t + 1) * nx];
    for (int i = 0; i < nx; ++i) { double s = vn[i]; for (int j = 0; j < nx; ++j) s += Vn[i * nx + j] * f[j]; Vp[i] = s; }
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) { double s = 0; for (int k = 0; k < nx; ++k) s += Fx[k * nx + i] * Vn[k * nx + j]; FxTV[i * nx + j] = s; }
    for (int i = 0; i < nu; ++i) for (int j = 0; j < nx; ++j) { double s = 0; for (int k = 0; k < nx; ++k) s += Fu[k * nu + i] * Vn[k * nx + j]; FuTV[i * nx + j] = s; }
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) { double s = Lxx[i * nx + j]; for (int k = 0; k < nx; ++k) s += FxTV[i * nx + k] * Fx[k * nx + j]; Qxx[i * nx + j] = s; }
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nu; ++j) { double s = Lxu[i * nu + j]; for (int k = 0; k < nx; ++k) s += FxTV[i * nx + k] * Fu[k * nu + j]; Qxu[i * nu + j] = s; }
    for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) { double s = Luu[i * nu + j]; for (int k = 0; k < nx; ++k) s += FuTV[i * nx + k] * Fu[k * nu + j]; Quu[i * nu + j] = s; }
    for (int i = 0; i < nx; ++i) { double s = Lx[i]; for (int k = 0; k < nx; ++k) s += Fx[k * nx + i] * Vp[k]; Qx[i] = s; }
    for (int i = 0; i < nu; ++i) { double s = Lu[i]; for (int k = 0; k < nx; ++k) s += Fu[k * nu + i] * Vp[k]; Qu[i] = s; }
    for (int i = 0; i < nx; ++i) { Qxx[i * nx + i] += sigma; if (cx) Qx[i] -= sigma * cx[(size_t)t * nx + i]; }
    for (int i = 0; i < nu; ++i) { Quu[i * nu + i] += sigma + preg; if (cu) Qu[i] -= sigma * cu[(size_t)t * nu + i]; }
    L = Quu;
    cholesky(nu, L.data());
    // LLT failure (SolverCSQP backwardPass): a pivot of the factorisation that is not positive (sqrt of it: not a positive number).
    // (Until round 3 this tested  a_jj - sum_{k<j} a_jk^2  of the UNFACTORED matrix, which is not the pivot: it flagged positive
    // definite matrices with large off-diagonal entries.)
    for (int j = 0; j < nu; ++j) if (!(L[j * nu + j] > 0.0)) ok = false;
    double *K = &d.K[(size_t)t * nu * nx], *kk = &d.k[(size_t)t * nu];
    for (int j = 0; j < nx; ++j) {
      for (int i = 0; i < nu; ++i) col[i] = Qxu[j * nu + i];
      chol_solve(nu, L.data(), col.data());
      for (int i = 0; i < nu; ++i) K[i * nx + j] = col[i];
    }
    for (int i = 0; i < nu; ++i) kk[i] = Qu[i];
    chol_solve(nu, L.data(), kk);
    double *V = &d.Vxx[(size_t)t * nx * nx], *vx = &d.Vx[(size_t)t * nx];
    for (int i = 0; i < nx; ++i) { double s = Qx[i]; for (int k = 0; k < nu; ++k) s -= K[k * nx + i] * Qu[k]; vx[i] = s; }
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) { double s = Qxx[i * nx + j]; for (int k = 0; k < nu; ++k) s -= Qxu[i * nu + k] * K[k * nx + j]; V[i * nx + j] = s; }
    for (int i = 0; i < nx; ++i) for (int j = i + 1; j < nx; ++j) { double s = 0.5 * (V[i * nx + j] + V[j * nx + i]); V[i * nx + j] = s; V[j * nx + i] = s; }
    for (int i = 0; i < nx; ++i) V[i * nx + i] += dreg;
  }
  // forward (dx_0 = x0 - xs_0 = 0 because the solver pins xs_0 = x0)
  for (int t = 0; t < T; ++t) {
    const double *n = tiles + (size_t)t * TILE;
    const double *Fx = n + oFx, *Fu = n + oFu, *f = n + of;
    const double *K = &d.K[(size_t)t * nu * nx], *kk = &d.k[(size_t)t * nu];
    const double *dx = &d.dx[(size_t)t * nx];
    double *du = &d.du[(size_t)t * nu], *dxn = &d.dx[(size_t)(t + 1) * nx];
    for (int i = 0; i < nu; ++i) { double s = -kk[i]; for (int j = 0; j < nx; ++j) s -= K[i * nx + j] * dx[j]; du[i] = s; }
    for (int i = 0; i < nx; ++i) { double s = f[i]; for (int j = 0; j < nx; ++j) s += Fx[i * nx + j] * dx[j]; for (int j = 0; j < nu; ++j) s += Fu[i * nu + j] * du[j]; dxn[i] = s; }
  }
  for (int t = 0; t <= T; ++t) {
    const double *V = &d.Vxx[(size_t)t * nx * nx], *vx = &d.Vx[(size_t)t * nx], *dx = &d.dx[(size_t)t * nx];
    double *lm = &d.lag[(size_t)t * nx];
    for (int i = 0; i < nx; ++i) { double s = vx[i]; for (int j = 0; j < nx; ++j) s += V[i * nx + j] * dx[j]; lm[i] = s; }
  }
  // KKT by definition with the multipliers (mim_solvers checkKKTConditions)
  double kkt = 0.0;
  for (int t = 0; t < T; ++t) {
    const double *n = tiles + (size_t)t * TILE;
    const double *Fx = n + oFx, *Fu = n + oFu, *f = n + of, *Lx = n + oLx, *Lu = n + oLu;
    const double *ln = &d.lag[(size_t)(t + 1) * nx], *lt = &d.lag[(size_t)t * nx];
    if (t > 0)
      for (int i = 0; i < nx; ++i) { double s = Lx[i] - lt[i]; for (int k = 0; k < nx; ++k) s += Fx[k * nx + i] * ln[k]; kkt = std::max(kkt, std::fabs(s)); }
    for (int i = 0; i < nu; ++i) { double s = Lu[i]; for (int k = 0; k < nx; ++k) s += Fu[k * nu + i] * ln[k]; kkt = std::max(kkt, std::fabs(s)); }
    for (int i = 0; i < nx; ++i) kkt = std::max(kkt, std::fabs(f[i]));
  }
  {
    const double *Lx = tiles + (size_t)T * TILE + oLx;
    const double *lt = &d.lag[(size_t)T * nx];
    for (int i = 0; i < nx; ++i) kkt = std::max(kkt, std::fabs(Lx[i] - lt[i]));
  }
  d.kkt = kkt;
  return ok;
}

// ---------------------------------------------------------------------------
// Constrained QP direction: mim_solvers::SolverCSQP::computeDirection restated from recall
// (no fixture of the reference covers it: parity UNPINNED).
//   equality-QP initial guess (plain LQR) -> prox centres (dx, du);
//   ADMM iterations: backward pass on  H + sigma I + G' rho G,  g + G'(y - rho z) - sigma (centre),
//   forward pass, then per node
//     z_rel = alpha C d + (1 - alpha) z,  z = clip(z_rel + y / rho, lb - g, ub - g),  y += rho (z_rel - z)
//   rho adapted every 25 iterations from the primal / dual residual ratio; stop when both residuals
//   are below eps_abs + eps_rel * scale.  Duals y and rho persist across SQP iterations and solves
//   (reset_y = reset_rho = false, the solver defaults).  The backward pass is redone in full at every
//   iteration here; the reference re-factorises only when rho changed, which is the same arithmetic.
// ---------------------------------------------------------------------------
struct Admm {
  std::vector<std::vector<double>> y, z, rho;  // per node
  double rho_sparse = 1e-1;
  bool init = false;
  int qp_iters = 0;
};
const double kRhoMin = 1e-6, kRhoMax = 1e3, kAdaptiveRhoTol = 5.0, kAlphaRelax = 1.6, kSigma = 1e-6;
const int kRhoInterval = 25;

void admm_apply_rho(const std::vector<ConNode> &cn, Admm &a) {
  for (size_t t = 0; t < cn.size(); ++t) {
    a.rho[t].resize(cn[t].nc);
    for (int k = 0; k < cn[t].nc; ++k) {
      const double lb = cn[t].lb[k], ub = cn[t].ub[k];
      if (lb == -INFINITY && ub == INFINITY) a.rho[t][k] = kRhoMin;
      else if (std::fabs(lb - ub) <= 1e-6) a.rho[t][k] = 1e3 * a.rho_sparse;
      else a.rho[t][k] = a.rho_sparse;
    }
  }
}

bool direction_admm(int nv, int T, const double *tiles, const std::vector<ConNode> &cn, const Ocp &o, Admm &a,
                    double preg, double dreg, Direction &d, std::vector<double> &aug) {
  const int nx = 2 * nv, nu = nv, TILE = AGX_TILE_DOUBLES(nv);
  const int oLx = nx * nx + nx * nu + nx, oLu = oLx + nx, oLxx = oLu + nu, oLxu = oLxx + nx * nx, oLuu = oLxu + nx * nu;
  if (!a.init || (int)a.y.size() != T + 1) {
    a.y.assign(T + 1, {}); a.z.assign(T + 1, {}); a.rho.assign(T + 1, {});
    a.rho_sparse = 1e-1;
    a.init = true;
  }
  for (int t = 0; t <= T; ++t) {
    if ((int)a.y[t].size() != cn[t].nc) a.y[t].assign(cn[t].nc, 0.0);
    a.z[t].assign(cn[t].nc, 0.0);  // reset_params(): z = 0, y kept
  }
  admm_apply_rho(cn, a);
  // equality-constrained QP initial guess
  Direction d0;
  bool ok = direction(nv, T, tiles, d0, 0.0, preg, dreg);
  std::vector<double> cx = d0.dx, cu = d0.du;
  aug.assign(tiles, tiles + (size_t)(T + 1) * TILE);
  std::vector<double> Cd, zprev, zrel, h;
  a.qp_iters = o.max_qp;
  for (int iter = 1; iter <= o.max_qp; ++iter) {
    // augmented tiles
    for (int t = 0; t <= T; ++t) {
      const ConNode &n = cn[t];
      if (n.nc == 0) continue;
      const double *src = tiles + (size_t)t * TILE;
      double *dst = aug.data() + (size_t)t * TILE;
      std::memcpy(dst + oLx, src + oLx, sizeof(double) * (TILE - oLx));
      h.resize(n.nc);
      for (int k = 0; k < n.nc; ++k) h[k] = a.y[t][k] - a.rho[t][k] * a.z[t][k];
      for (int k = 0; k < n.nc; ++k) {
        const double rk = a.rho[t][k];
        const double *gx = &n.Gx[(size_t)k * nx], *gu = &n.Gu[(size_t)k * nu];
        for (int i = 0; i < nx; ++i) {
          dst[oLx + i] += gx[i] * h[k];
          for (int j = 0; j < nx; ++j) dst[oLxx + i * nx + j] += rk * gx[i] * gx[j];
          if (t < T) for (int j = 0; j < nu; ++j) dst[oLxu + i * nu + j] += rk * gx[i] * gu[j];
        }
        if (t < T)
          for (int i = 0; i < nu; ++i) {
            dst[oLu + i] += gu[i] * h[k];
            for (int j = 0; j < nu; ++j) dst[oLuu + i * nu + j] += rk * gu[i] * gu[j];
          }
      }
    }
    const bool ok_iter = direction(nv, T, aug.data(), d, kSigma, preg, dreg, cx.data(), cu.data());
    ok = ok_iter && ok;
    if (!ok_iter) {  // Quu of the augmented problem not positive definite: the direction is discarded, multipliers stay
      a.qp_iters = iter;
      break;
    }
    // update_lagrangian_parameters
    double norm_primal = 0.0, norm_dual = 0.0, norm_primal_rel = 0.0, norm_dual_rel = 0.0;
    for (int t = 0; t <= T; ++t) {
      const ConNode &n = cn[t];
      if (n.nc == 0) continue;
      const double *dx = &d.dx[(size_t)t * nx], *du = t < T ? &d.du[(size_t)t * nu] : nullptr;
      Cd.assign(n.nc, 0.0); zprev = a.z[t]; zrel.resize(n.nc);
      for (int k = 0; k < n.nc; ++k) {
        double c = 0.0;
        for (int j = 0; j < nx; ++j) c += n.Gx[(size_t)k * nx + j] * dx[j];
        if (du) for (int j = 0; j < nu; ++j) c += n.Gu[(size_t)k * nu + j] * du[j];
        Cd[k] = c;
        zrel[k] = kAlphaRelax * c + (1.0 - kAlphaRelax) * a.z[t][k];
        double zn = zrel[k] + a.y[t][k] / a.rho[t][k];
        zn = std::min(std::max(zn, n.lb[k] - n.g[k]), n.ub[k] - n.g[k]);
        a.z[t][k] = zn;
        a.y[t][k] += a.rho[t][k] * (zrel[k] - zn);
        norm_primal = std::max(norm_primal, std::fabs(c - zn));
        norm_primal_rel = std::max(norm_primal_rel, std::max(std::fabs(c), std::fabs(zn)));
      }
      for (int j = 0; j < nx; ++j) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < n.nc; ++k) { s1 += n.Gx[(size_t)k * nx + j] * a.rho[t][k] * (a.z[t][k] - zprev[k]); s2 += n.Gx[(size_t)k * nx + j] * a.y[t][k]; }
        norm_dual = std::max(norm_dual, std::fabs(s1)); norm_dual_rel = std::max(norm_dual_rel, std::fabs(s2));
      }
      if (du)
        for (int j = 0; j < nu; ++j) {
          double s1 = 0.0, s2 = 0.0;
          for (int k = 0; k < n.nc; ++k) { s1 += n.Gu[(size_t)k * nu + j] * a.rho[t][k] * (a.z[t][k] - zprev[k]); s2 += n.Gu[(size_t)k * nu + j] * a.y[t][k]; }
          norm_dual = std::max(norm_dual, std::fabs(s1)); norm_dual_rel = std::max(norm_dual_rel, std::fabs(s2));
        }
    }
    cx = d.dx; cu = d.du;
    // update_rho_vec
    {
      double scale = std::sqrt((norm_primal * norm_dual_rel) / (norm_dual * norm_primal_rel));
      double est = scale * a.rho_sparse;
      est = std::min(std::max(est, kRhoMin), kRhoMax);
      if (iter % kRhoInterval == 0 && iter > 1)
        if (est > a.rho_sparse * kAdaptiveRhoTol || est < a.rho_sparse / kAdaptiveRhoTol) {
          a.rho_sparse = est;
          admm_apply_rho(cn, a);
        }
    }
    if (norm_primal <= o.eps_abs + o.eps_rel * norm_primal_rel && norm_dual <= o.eps_abs + o.eps_rel * norm_dual_rel) {
      a.qp_iters = iter;
      break;
    }
  }
  // KKT with the constraint multipliers (checkKKTConditions): stationarity with + G' y
  const int oFx = 0, oFu = nx * nx, of = oFu + nx * nu;
  double kkt = 0.0;
  for (int t = 0; t < T; ++t) {
    const double *n = tiles + (size_t)t * TILE;
    const double *Fx = n + oFx, *Fu = n + oFu, *f = n + of, *Lx = n + oLx, *Lu = n + oLu;
    const double *ln = &d.lag[(size_t)(t + 1) * nx], *lt = &d.lag[(size_t)t * nx];
    const ConNode &c = cn[t];
    if (t > 0)
      for (int i = 0; i < nx; ++i) {
        double s = Lx[i] - lt[i];
        for (int k = 0; k < nx; ++k) s += Fx[k * nx + i] * ln[k];
        for (int k = 0; k < c.nc; ++k) s += c.Gx[(size_t)k * nx + i] * a.y[t][k];
        kkt = std::max(kkt, std::fabs(s));
      }
    for (int i = 0; i < nu; ++i) {
      double s = Lu[i];
      for (int k = 0; k < nx; ++k) s += Fu[k * nu + i] * ln[k];
      for (int k = 0; k < c.nc; ++k) s += c.Gu[(size_t)k * nu + i] * a.y[t][k];
      kkt = std::max(kkt, std::fabs(s));
    }
    for (int i = 0; i < nx; ++i) kkt = std::max(kkt, std::fabs(f[i]));
  }
  {
    const double *Lx = tiles + (size_t)T * TILE + oLx;
    const double *lt = &d.lag[(size_t)T * nx];
    const ConNode &c = cn[T];
    for (int i = 0; i < nx; ++i) {
      double s = Lx[i] - lt[i];
      for (int k = 0; k < c.nc; ++k) s += c.Gx[(size_t)k * nx + i] * a.y[T][k];
      kkt = std::max(kkt, std::fabs(s));
    }
  }
  d.kkt = kkt;
  return ok;
}

// ---------------------------------------------------------------------------
// SQP outer loop for one instance.
// ---------------------------------------------------------------------------
// Scratch of one solve.  The batch entry points keep ONE PER THREAD alive across instances and calls (thread_local):
// a workspace built per instance maps and unmaps its half-megabyte vectors for every solve, and 64 threads doing that
// at once queue up on the process's address-space lock (bench.py's CPU leg went 11 x on 64 threads).
struct Workspace {
  std::vector<double> tiles, xs_try, us_try, xn, aug;
  Direction dir, dirK;
  NodeOut node;
  std::vector<ConNode> cn;
};

void eval_tiles(const Model &m, const Ocp &o, const double *xs, const double *us, const double *ref,
                const int32_t *frames, Workspace &w, double &cost, double &gap1, double *con1 = nullptr, void *ana = nullptr) {
  const int nv = m.nv, nx = 2 * nv, nu = nv, T = o.T, TILE = AGX_TILE_DOUBLES(nv);
  w.tiles.resize((size_t)(T + 1) * TILE);
  cost = 0.0; gap1 = 0.0;
  if (ana) {  // analytic derivatives (bench baseline): the canonical tile straight from oracle/agx_analytic.cpp
    const int of = nx * nx + nx * nu;
    if (con1) *con1 = 0.0;
    for (int t = 0; t <= T; ++t) {
      const bool term = (t == T);
      double *tile = &w.tiles[(size_t)t * TILE];
      ana_node_diff(ana, term, term ? 0.0 : o.dt[t], xs + (size_t)t * nx, term ? nullptr : us + (size_t)t * nu,
                    term ? nullptr : xs + (size_t)(t + 1) * nx, ref + (size_t)t * o.stride, frames ? frames + (size_t)t * AGX_MAX_ROWS : nullptr, tile);
      cost += tile[TILE - 1];
      if (!term) for (int i = 0; i < nx; ++i) gap1 += std::fabs(tile[of + i]);
    }
    return;
  }
  const bool has_con = o.nc[0] + o.nc[1] > 0;
  if (has_con) w.cn.resize(T + 1);
  if (con1) *con1 = 0.0;
  for (int t = 0; t <= T; ++t) {
    const bool term = (t == T);
    if (has_con) {
      ConNode &c = w.cn[t];
      node_constraints_diff(m, o, term, xs + (size_t)t * nx, term ? nullptr : us + (size_t)t * nu, c);
      if (con1) for (int k = 0; k < c.nc; ++k) *con1 += std::max(c.lb[k] - c.g[k], 0.0) + std::max(c.g[k] - c.ub[k], 0.0);
    }
    node_calc_diff(m, o, term, term ? 0.0 : o.dt[t], xs + (size_t)t * nx, term ? nullptr : us + (size_t)t * nu,
                   ref + (size_t)t * o.stride, frames ? frames + (size_t)t * AGX_MAX_ROWS : nullptr, w.node);
    pack_tile(nv, w.node, term ? nullptr : xs + (size_t)(t + 1) * nx, &w.tiles[(size_t)t * TILE]);
    cost += w.node.cost;
    if (!term) for (int i = 0; i < nx; ++i) gap1 += std::fabs(w.node.xnext[i] - xs[(size_t)(t + 1) * nx + i]);
  }
}

void solve_one(const Model &m, const Ocp &o, const double *ref, const int32_t *frames, const double *x0,
               const double *xs_ws, const double *us_ws, int max_iter, double max_time, double *xs, double *us,
               double *K, agx_status *st, Workspace &w, Admm *admm = nullptr, void *ana = nullptr) {
  const int nv = m.nv, nx = 2 * nv, nu = nv, T = o.T;
  const bool has_con = (o.nc[0] + o.nc[1] > 0) && admm != nullptr;
  if (has_con) ana = nullptr;  // the analytic leg covers unconstrained problems
  auto t_start = std::chrono::steady_clock::now();
  std::memcpy(xs, xs_ws, sizeof(double) * (T + 1) * nx);
  std::memcpy(us, us_ws, sizeof(double) * T * nu);
  std::memcpy(xs, x0, sizeof(double) * nx);  // xs_[0] = problem.x0
  if (max_iter <= 0) max_iter = 1000;
  std::memset(st, 0, sizeof(*st));
  std::memset(K, 0, sizeof(double) * T * nu * nx);
  w.xs_try.resize((size_t)(T + 1) * nx); w.us_try.resize((size_t)T * nu); w.xn.resize(nx);
  int it = 0;
  // crocoddyl regularisation state (SolverDDP defaults: reg_min 1e-9, reg_max 1e9, factor 10,
  // th_stepdec 0.5, th_stepinc 0.01); sigma = 1e-6 is SolverCSQP's proximal weight.
  const double reg_min = 1e-9, reg_max = 1e9, sigma = 1e-6;
  double preg = reg_min, dreg = reg_min;
  Direction &dirK = w.dirK;
  auto final_gains = [&]() {
    if (has_con) return;  // constrained: K already holds the gains of the last ADMM backward pass
    // K reported by the solver comes from the sigma-regularised ADMM backward pass
    direction(nv, T, w.tiles.data(), dirK, sigma, preg, dreg, w.dir.dx.data(), w.dir.du.data());
    std::memcpy(K, dirK.K.data(), sizeof(double) * T * nu * nx);
  };
  bool have_dir = false;
  for (; it < max_iter; ++it) {
    double cost, gap1, con1 = 0.0;
    eval_tiles(m, o, xs, us, ref, frames, w, cost, gap1, &con1, ana);
    const double merit = cost + o.mu_dyn * gap1 + o.mu_con * con1;
    bool ok;
    if (has_con) {
      ok = direction_admm(nv, T, w.tiles.data(), w.cn, o, *admm, preg, dreg, w.dir, w.aug);
      w.dir.kkt = std::max(w.dir.kkt, con1);  // checkKKTConditions: KKT = max(KKT, constraint_norm)
      std::memcpy(K, w.dir.K.data(), sizeof(double) * T * nu * nx);
    } else {
      ok = direction(nv, T, w.tiles.data(), w.dir, 0.0, preg, dreg);
    }
    have_dir = true;
    if (!ok) w.dir.kkt = std::nan("");  // discarded direction: its KKT residual is undefined (never "converged")
    st->kkt = w.dir.kkt; st->cost = cost; st->merit = merit; st->gap_norm = gap1; st->qp_iters = has_con ? admm->qp_iters : 1;
    if (!ok) st->flags |= 1;
    if (w.dir.kkt <= o.tol) { st->solved = 1; break; }
    // merit line search, alpha = 2^-n, n = 0..9, no rollout
    bool accepted = false;
    double alpha = 1.0, used = 1.0;
    for (int n = 0; n < 10; ++n, alpha *= 0.5) {
      used = alpha;
      if (n > 0) st->flags |= 4;  // a step length was rejected in this solve (agx_status.flags bit 2)
      for (size_t i = 0; i < (size_t)(T + 1) * nx; ++i) w.xs_try[i] = xs[i] + alpha * w.dir.dx[i];
      for (size_t i = 0; i < (size_t)T * nu; ++i) w.us_try[i] = us[i] + alpha * w.dir.du[i];
      double cost_try = 0.0, gap_try = 0.0, con_try = 0.0;
      for (int t = 0; t <= T; ++t) {
        const bool term = (t == T);
        double c;
        if (ana)
          ana_node_calc(ana, term, term ? 0.0 : o.dt[t], &w.xs_try[(size_t)t * nx], term ? nullptr : &w.us_try[(size_t)t * nu],
                        ref + (size_t)t * o.stride, frames ? frames + (size_t)t * AGX_MAX_ROWS : nullptr, w.xn.data(), &c);
        else
        node_calc(m, o, term, term ? 0.0 : o.dt[t], &w.xs_try[(size_t)t * nx], term ? nullptr : &w.us_try[(size_t)t * nu],
                  ref + (size_t)t * o.stride, frames ? frames + (size_t)t * AGX_MAX_ROWS : nullptr, w.xn.data(), c);
        cost_try += c;
        if (!term) for (int i = 0; i < nx; ++i) gap_try += std::fabs(w.xn[i] - w.xs_try[(size_t)(t + 1) * nx + i]);
        if (has_con) con_try += node_constraint_violation(m, o, term, &w.xs_try[(size_t)t * nx], term ? nullptr : &w.us_try[(size_t)t * nu]);
      }
      const double merit_try = cost_try + o.mu_dyn * gap_try + o.mu_con * con_try;
      if (o.use_filter) {
        // filter of size 1 (the solver's default): rejected only if no better in cost AND gaps AND constraints
        const bool worse = (cost <= cost_try) && (gap1 <= gap_try) && (con1 <= con_try);
        if (!worse) { accepted = true; break; }
      } else if (merit > merit_try) { accepted = true; break; }
    }
    const bool last = (it + 1 == max_iter);
    if (last || !accepted) final_gains();  // gains belong to the point the direction was computed at
    if (accepted) {
      std::memcpy(xs, w.xs_try.data(), sizeof(double) * (T + 1) * nx);
      std::memcpy(us, w.us_try.data(), sizeof(double) * T * nu);
    } else {
      st->flags |= 2;
    }
    if (used > 0.5) { preg = std::max(preg / 10.0, reg_min); dreg = std::max(dreg / 10.0, reg_min); }
    if (used <= 0.01) {
      preg = std::min(preg * 10.0, reg_max); dreg = std::min(dreg * 10.0, reg_max);
      if (preg == reg_max) { ++it; break; }
    }
    if (max_time > 0.0) {
      double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
      if (el > max_time) { if (!last && accepted) final_gains(); ++it; break; }
    }
  }
  if (st->solved && have_dir) final_gains();
  st->iter = it;
  for (size_t i = 0; i < (size_t)(T + 1) * nx; ++i) if (!std::isfinite(xs[i])) st->flags |= 1;
}

struct OrcOcp {
  Model m;
  Ocp o;
  int B;
  void *ana = nullptr;   // analytic-derivative leg (oracle/agx_analytic.cpp), NULL when not covered
  bool use_ana = false;  // orc_set_analytic: bench.py's "port-analytic" baseline
  ~OrcOcp() { if (ana) ana_destroy(ana); }
  std::vector<Admm> admm;  // per instance; duals and rho persist across solves like the solver object's
};

thread_local std::string g_err;

}  // namespace

extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }

int orc_ocp_create(const agx_model_desc *md, const agx_ocp_desc *od, int batch, void **out) {
  if (!md || !od || !out || md->nv < 1 || md->nv > AGX_MAX_NV) { g_err = "orc_ocp_create: bad arguments"; return -1; }
  OrcOcp *p = new OrcOcp();
  copy_model(md, p->m);
  copy_ocp(od, md->nv, p->o);
  p->B = batch;
  p->admm.resize(batch);
  p->ana = ana_create(md, od);
  *out = p;
  return 0;
}
void orc_ocp_destroy(void *h) { delete static_cast<OrcOcp *>(h); }

int orc_rnea(void *h, int n, const double *q, const double *v, const double *a, double *tau) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  const int nv = p->m.nv;
  for (int i = 0; i < n; ++i) rnea<double>(p->m, q + (size_t)i * nv, v + (size_t)i * nv, a + (size_t)i * nv, tau + (size_t)i * nv, true);
  return 0;
}
int orc_mass_matrix(void *h, const double *q, double *M) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  mass_matrix<double>(p->m, q, M);
  return 0;
}
int orc_forward_dynamics(void *h, int n, const double *q, const double *v, const double *u, double *a) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  const int nv = p->m.nv;
  for (int i = 0; i < n; ++i) forward_dynamics<double>(p->m, q + (size_t)i * nv, v + (size_t)i * nv, u + (size_t)i * nv, a + (size_t)i * nv);
  return 0;
}
int orc_frame_placement(void *h, int n, int frame, const double *q, double *out) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  if (frame < 0 || frame >= p->m.nframes) { g_err = "orc_frame_placement: bad frame"; return -1; }
  for (int i = 0; i < n; ++i) frame_placement<double>(p->m, frame, q + (size_t)i * p->m.nv, out + (size_t)i * 12, out + (size_t)i * 12 + 9);
  return 0;
}
int orc_log6(const double *M12, double *r6) { log6d(M12, M12 + 9, r6); return 0; }
int orc_jlog6(const double *M12, double *J36) { jlog6d(M12, M12 + 9, J36); return 0; }

int orc_integrate(void *h, int n, const double *x, const double *u, double *xnext) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  const int nv = p->m.nv, nx = 2 * nv;
  Ocp bare = p->o;
  bare.rows[0].clear(); bare.row_off[0].clear();
  for (int i = 0; i < n; ++i) {
    double c;
    node_calc(p->m, bare, false, p->o.dt[0], x + (size_t)i * nx, u + (size_t)i * nv, nullptr, nullptr, xnext + (size_t)i * nx, c);
  }
  return 0;
}

// One node: calc + calcDiff into a tile (gap is xnext - xs_next when given).
int orc_node_calc_diff(void *h, int terminal, double dt, const double *x, const double *u, const double *ref,
                       const int32_t *frames, const double *xs_next, double *tile, double *xnext, double *residuals) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  NodeOut n;
  node_calc_diff(p->m, p->o, terminal != 0, dt, x, u, ref, frames, n);
  pack_tile(p->m.nv, n, xs_next, tile);
  if (xnext) std::memcpy(xnext, n.xnext.data(), sizeof(double) * 2 * p->m.nv);
  if (residuals) std::memcpy(residuals, n.residuals.data(), sizeof(double) * n.residuals.size());
  return 0;
}
int orc_node_calc(void *h, int terminal, double dt, const double *x, const double *u, const double *ref,
                  const int32_t *frames, double *xnext, double *cost, double *residuals) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  std::vector<double> res;
  node_calc(p->m, p->o, terminal != 0, dt, x, u, ref, frames, xnext, *cost, &res);
  if (residuals) std::memcpy(residuals, res.data(), sizeof(double) * res.size());
  return 0;
}

// tiles for B instances at (xs, us): [B][T+1][TILE]
int orc_calc_diff(void *h, const double *ref, const int32_t *frames, const double *xs, const double *us, double *tiles) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  const int nv = p->m.nv, nx = 2 * nv, T = p->o.T, TILE = AGX_TILE_DOUBLES(nv);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < p->B; ++b) {
    static thread_local Workspace w;
    double c, g;
    eval_tiles(p->m, p->o, xs + (size_t)b * (T + 1) * nx, us + (size_t)b * T * nv, ref + (size_t)b * (T + 1) * p->o.stride,
               frames ? frames + (size_t)b * (T + 1) * AGX_MAX_ROWS : nullptr, w, c, g);
    std::memcpy(tiles + (size_t)b * (T + 1) * TILE, w.tiles.data(), sizeof(double) * (T + 1) * TILE);
  }
  return 0;
}

// One QP direction as the solver computes it: plain pass with (preg, dreg) gives k, dx, du
// and the KKT residual; the sigma (ADMM) pass around that direction gives the reported gains K.
int orc_direction(void *h, const double *tiles, double preg, double dreg, double *K, double *k, double *dx, double *du,
                  double *kkt) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  const int nv = p->m.nv, nx = 2 * nv, nu = nv, T = p->o.T, TILE = AGX_TILE_DOUBLES(nv);
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < p->B; ++b) {
    Direction d, dk;
    direction(nv, T, tiles + (size_t)b * (T + 1) * TILE, d, 0.0, preg, dreg);
    direction(nv, T, tiles + (size_t)b * (T + 1) * TILE, dk, 1e-6, preg, dreg, d.dx.data(), d.du.data());
    if (K) std::memcpy(K + (size_t)b * T * nu * nx, dk.K.data(), sizeof(double) * T * nu * nx);
    if (k) std::memcpy(k + (size_t)b * T * nu, d.k.data(), sizeof(double) * T * nu);
    if (dx) std::memcpy(dx + (size_t)b * (T + 1) * nx, d.dx.data(), sizeof(double) * (T + 1) * nx);
    if (du) std::memcpy(du + (size_t)b * T * nu, d.du.data(), sizeof(double) * T * nu);
    if (kkt) kkt[b] = d.kkt;
  }
  return 0;
}

int orc_solve(void *h, const double *ref, const int32_t *frames, const double *x0, const double *xs_ws,
              const double *us_ws, int max_iter, double max_time, double *xs, double *us, double *K,
              agx_status *st, int nthreads) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  const int nv = p->m.nv, nx = 2 * nv, nu = nv, T = p->o.T;
  (void)nthreads;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads > 0 ? nthreads : 1)
  for (int b = 0; b < p->B; ++b) {
    static thread_local Workspace w;
    solve_one(p->m, p->o, ref + (size_t)b * (T + 1) * p->o.stride, frames ? frames + (size_t)b * (T + 1) * AGX_MAX_ROWS : nullptr,
              x0 + (size_t)b * nx, xs_ws + (size_t)b * (T + 1) * nx, us_ws + (size_t)b * T * nu, max_iter, max_time,
              xs + (size_t)b * (T + 1) * nx, us + (size_t)b * T * nu, K + (size_t)b * T * nu * nx, st + b, w, &p->admm[b],
              p->use_ana ? p->ana : nullptr);
  }
  return 0;
}

// 1: the per-node derivatives come from the analytic formulas of oracle/agx_analytic.cpp (bench baseline "port-analytic");
// returns 0 when the problem is outside what that leg covers (then nothing changes), 1 otherwise
int orc_set_analytic(void *h, int on) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  if (on && !p->ana) return 0;
  p->use_ana = on != 0;
  return 1;
}

// forget the constraint multipliers and rho of every instance (a freshly constructed solver)
int orc_reset_duals(void *h) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  for (auto &a : p->admm) a = Admm();
  return 0;
}

// constraint values of one node: g [nc], and the node's bounds (test hook)
int orc_node_constraints(void *h, int terminal, const double *x, const double *u, double *g, double *Gx, double *Gu, int *nc) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  ConNode n;
  node_constraints_diff(p->m, p->o, terminal != 0, x, u, n);
  if (nc) *nc = n.nc;
  if (g) std::memcpy(g, n.g.data(), sizeof(double) * n.nc);
  if (Gx) std::memcpy(Gx, n.Gx.data(), sizeof(double) * n.Gx.size());
  if (Gu) std::memcpy(Gu, n.Gu.data(), sizeof(double) * n.Gu.size());
  return 0;
}

// WarmStartShiftPreviousSolution.shift (warm_start_shift_previous_solution.py:85-109)
int orc_shift_warmstart(void *h, double *xs, double *us) {
  OrcOcp *p = static_cast<OrcOcp *>(h);
  const int nv = p->m.nv, nx = 2 * nv, nu = nv, T = p->o.T;
  Ocp bare = p->o;
  bare.rows[0].clear(); bare.row_off[0].clear();
  const double dt0 = p->o.dt[0];
  for (int b = 0; b < p->B; ++b) {
    double *X = xs + (size_t)b * (T + 1) * nx, *U = us + (size_t)b * T * nu;
    std::vector<double> xn(nx);
    for (int i = 0; i < T; ++i) {
      if (p->o.dt[i] == dt0) {
        std::memcpy(X + (size_t)i * nx, X + (size_t)(i + 1) * nx, sizeof(double) * nx);
        if (i < T - 1) std::memcpy(U + (size_t)i * nu, U + (size_t)(i + 1) * nu, sizeof(double) * nu);
      } else {
        double c;
        node_calc(p->m, bare, false, dt0, X + (size_t)i * nx, U + (size_t)i * nu, nullptr, nullptr, xn.data(), c);
        std::memcpy(X + (size_t)i * nx, xn.data(), sizeof(double) * nx);
      }
    }
  }
  return 0;
}

}  // extern "C"
