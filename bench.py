#!/usr/bin/env python3
"""bench.py -- MPC steps/s of the MI355X-native OCP solve path.

One "step" = one receding-horizon MPC step (MPC.run, agimus_controller/agimus_controller/mpc.py:32-66)
of every instance of the batch, fully device resident: horizon window of the resident reference trajectory
(SURVEY 8(d)), x0 <- previous xs[1], warm-start shift, SQP solve (CSQP semantics, max_iter / tol of the ROS
defaults), then the download of what the controller publishes (us[0], K[0], x1 and the solver status).
Inputs sit in HBM before the timed region starts.

  python bench.py [--gpus N --steps K --warmup W --batch B --horizon T --max-iter I --workload W --scaling weak|strong]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL).  Instances are
independent, so ranks share nothing while they solve: the collectives are the barriers, the MAX over ranks of the
timed region and, after it, one all_gather of the per-instance status words (plus, in the host_refs leg, the
scatter of the reference tiles from rank 0 and the gather of the first-node results).
--scaling weak (default): --batch instances PER GPU.  --scaling strong: --batch instances in total, B / N per rank.
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from agimus_controller_amd import _abi, backend, batched, workloads  # noqa: E402
from agimus_controller_amd.factory import robot_tables as rt  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
# Algorithmic doubles per node and launch (nv = 7): K1 read x,u + reference tile and
# write the 673-double derivative tile; K2+K3 one Riccati backward + one linear forward; K4 one trial.
ALGO_DOUBLES = {"calc_qp": 775, "riccati": 778 + 448, "step": 138}
PARITY = {
    "sine": "pinned: golden file of the reference reproduced to 1e-9 (tests/test_oracle_golden.py, tests/test_hip_parity.py)",
    "generic": "pinned: golden file of the reference reproduced to 1e-9 (same kernels as the sine workload)",
    "humanoid": "pinned for the arithmetic (same algorithm as the golden case); the 30-DoF tree is synthetic, HIP == checker",
    "collision": "UNPINNED: colmpc distance / QuadExp and the ADMM loop are recalled forms, HIP == this repository's checker only",
    "cartesian": "UNPINNED: colmpc distance / QuadExp and the ADMM loop are recalled forms, HIP == this repository's checker only",
}
K1_NAME = {
    "sine": "k_calc_qp_lj", "generic": "k_calc_qp_lj", "humanoid": "k_calc_qp_wg", "collision": "k_calc_qp_lj_coll", "cartesian": "k_calc_qp_lj_coll",
}
K1_TEXT = {
    "k_calc_qp_lj": "k_calc_qp_lj (node-parallel derivative pass, running nodes, 8 lanes per node)",
    "k_calc_qp_wg": "k_calc_qp_wg<30> (one workgroup per node: LDS-resident dynamics, tree sums and contractions on the fp64 matrix cores)",
    "k_calc_qp_lj_coll": ("k_calc_qp_lj<7, COLL> (8 lanes per node, the collision cost row evaluated by the node's lanes); launches of later SQP "
                          "iterations skip finished instances, so the in-situ average is over partly empty launches: an upper bound of the rate"),
}


def ncores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def cpu_quota():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited.  The affinity mask of a
    GPU box shows every hardware thread of the host, the quota is what the process really gets."""
    try:
        txt = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if txt[0] != "max":
            return float(txt[0]) / float(txt[1])
    except Exception:  # noqa: BLE001
        pass
    try:
        q = float(pathlib.Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        per = float(pathlib.Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        if q > 0:
            return q / per
    except Exception:  # noqa: BLE001
        pass
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="MPC instances per GPU (weak scaling) or in total (strong scaling); default: the "
                                                            "BASELINE.json shape of the workload: 1024; collision / cartesian 256; humanoid 512")
    ap.add_argument("--horizon", type=int, default=None, help="default: 100; collision / cartesian 200; humanoid 50")
    ap.add_argument("--max-iter", type=int, default=10, help="SQP iteration cap (ROS default 10)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--quorum", type=float, default=None,
                    help="batch policy (agx_ocp_set_quorum): a batch step ends once this fraction of the instances has finished, the rest "
                         "carry their iterate into the next step unsolved (as a lone controller hitting max_solve_time).  Default 1.0 "
                         "(everyone, as the CPU baseline and upstream); the collision / cartesian workloads report the 0.985 policy in "
                         "an extra leg (`quorum_0985`), never as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch1", action="store_true", help="skip the extra legs (batch 1 latency, max_iter 3, full download, host refs): profiling runs")
    ap.add_argument("--cpu-instances", type=int, default=0, help="instances of the CPU sample (0 = 2 per core, at most 256)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work the baseline sample is sized for")
    ap.add_argument("--loop", choices=("prediction", "feedback"), default="prediction",
                    help="how the next measured state is produced: previous xs[1] (the reference's dummy_mpc_test) or the Riccati "
                         "feedback law rolled out on the model at 1 kHz (SURVEY 8(f-3))")
    ap.add_argument("--disturb-sigma", type=float, default=0.2,
                    help="disturbed leg: the measured state of every step is the predicted one + N(0, sigma^2) rad on the joint positions and "
                         "N(0, (5 sigma)^2) rad/s on the velocities (seeded per instance and step), so that steps need several SQP iterations "
                         "and the line search backtracks")
    ap.add_argument("--workload", choices=("sine", "generic", "humanoid", "collision", "cartesian"), default="sine",
                    help="sine: BASELINE configs[1] at batch 1024 (the headline); generic: configs[3] generic_trajectory + pick-and-place costs; "
                         "humanoid: configs[4] synthetic 30-DoF tree (T 50, B 512); collision / cartesian: configs[2], collision-avoidance cost "
                         "+ distance constraint (T 200, B 256) with configuration-space / cartesian sine references")
    args = ap.parse_args()
    shape = {"collision": (256, 200), "cartesian": (256, 200), "humanoid": (512, 50)}.get(args.workload, (1024, 100))
    if args.batch is None:
        args.batch = shape[0]
    if args.horizon is None:
        args.horizon = shape[1]
    if args.quorum is None:
        args.quorum = 1.0  # `value` is like-for-like with the CPU leg: every instance runs to convergence or max_iter
    return args


def algo_doubles(nv):
    """SURVEY 8(d) per-node doubles: its nv = 7 figures as stated there, its formulas for other nv
    (ndx = 2 nv, nu = nv)."""
    if nv == 7:
        return dict(ALGO_DOUBLES)
    ndx, nu = 2 * nv, nv
    R = 2 * ndx + 2 * nu + 12 + 6 + 1
    D = 13 * nv * nv + 5 * nv + 1
    return {"calc_qp": ndx + nu + R + D, "riccati": (D + nu * ndx + nu) + (nu * ndx + nu + ndx * ndx + ndx * nu + ndx + 2 * ndx + nu),
            "step": 2 * (ndx + nu) + R + ndx + 1}


def make_problem(T, workload="sine"):
    if workload == "humanoid":
        table = rt.humanoid30_table()
        tcp = len(table.frame_names) - 1
        running, terminal = workloads.goal_reaching_rows(tcp)
        po = _abi.PackedOcp(30, [0.01] * T, running, terminal, termination_tolerance=1e-3, max_qp_iters=100)
        return table, tcp, po
    if workload in ("collision", "cartesian"):
        # ocp_traj_tracking_collision_avoidance.yaml: QuadExp(alpha 1e-4) cost + distance >= 1 cm on one pair; the sphere
        # obstacle sits where the link-7 capsule of part of the batch comes close (the reference's test obstacle at
        # x = 1.535 m is out of the arm's reach and would never be active); for the cartesian sine the sphere sits 0.24 m
        # in front of the link-7 capsule: the references of ~ 1/4 of the batch pass within the 1 cm bound
        xyz = (0.27, 0.22, 0.70) if workload == "collision" else (0.49, 0.222, 0.487)
        table = rt.panda_collision_table(0.1, obstacle_xyz=xyz, obstacle_radius=0.06, obstacle_length=0.0)
        tcp = table.frame_id("panda_hand_tcp")
        running, terminal = workloads.collision_avoidance_rows(table, tcp, alpha=1e-4)
        fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
        con = [_abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.01, upper=np.inf, frame=fa, frame_b=fb, name="collision")]
        po = _abi.PackedOcp(7, [0.01] * T, running, terminal, termination_tolerance=1e-3, max_qp_iters=100, running_constraints=con)
        return table, tcp, po
    table = rt.panda_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    running, terminal = workloads.goal_reaching_rows(tcp) if workload == "sine" else workloads.regulation_rows(terminal_weight=0.0)
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, termination_tolerance=1e-3, max_qp_iters=100)
    return table, tcp, po


class HostRefs:
    """The reference tiles of the resident trajectory rebuilt on the host, sample by sample, from what
    agx_traj_get_point returns (q, dq, ddq, feed-forward effort, end-effector pose): what a caller without the
    device-side generators hands to agx_ocp_set_refs (OCPCrocoGeneric.set_reference_weighted_trajectory,
    ocp_croco_generic.py:855-892).  Mirrors k_sine_fill row by row."""

    def __init__(self, hip, po, w, n_samples, n_inst):
        self.po, self.T, self.nv = po, po.horizon, po.nv
        nv = self.nv
        self.run = np.zeros((n_inst, n_samples, po.stride))
        self.term = np.zeros((n_inst, n_samples, po.stride))
        self.x = np.zeros((n_inst, n_samples, 2 * nv))
        self.u = np.zeros((n_inst, n_samples, nv))
        bc = lambda v, n: np.broadcast_to(np.asarray(v, dtype=float), (n,))  # noqa: E731
        for k in range(n_samples):
            q, v, a, u, pose = (z[:n_inst] for z in hip.traj_point(k))
            self.x[:, k], self.u[:, k] = np.concatenate([q, v], 1), u
            for rows, offs, dst in ((po.running, po.running_offsets, self.run), (po.terminal, po.terminal_offsets, self.term)):
                for r, off in zip(rows, offs):
                    seg = dst[:, k, off:]
                    seg[:, 0] = r.weight
                    if r.kind == _abi.RES_STATE:
                        seg[:, 1:1 + 2 * nv] = self.x[:, k]
                        seg[:, 1 + 2 * nv:1 + 3 * nv], seg[:, 1 + 3 * nv:1 + 4 * nv] = bc(w["w_q"], nv), bc(w["w_qdot"], nv)
                    elif r.kind == _abi.RES_CONTROL:
                        seg[:, 1:1 + nv], seg[:, 1 + nv:1 + 2 * nv] = u, bc(w["w_effort"], nv)
                    elif r.kind == _abi.RES_FRAME_PLACEMENT:
                        seg[:, 1:13], seg[:, 13:19] = pose, bc(w["w_pose"], 6)

    def first(self, n):
        sub = HostRefs.__new__(HostRefs)
        sub.po, sub.T, sub.nv = self.po, self.T, self.nv
        sub.run, sub.term, sub.x, sub.u = self.run[:n], self.term[:n], self.x[:n], self.u[:n]
        return sub

    def window(self, k0, out=None, lo=0, hi=None):
        """Horizon window starting at sample k0 (instances lo:hi) in the tile layout of agx_ocp_set_refs."""
        T = self.T
        out = np.empty((self.run.shape[0], T + 1, self.po.stride)) if out is None else out
        out[lo:hi, :T] = self.run[lo:hi, k0:k0 + T]
        out[lo:hi, T] = self.term[lo:hi, k0 + T]
        return out


def disturbance_noise(B, n_steps, nv, sigma, seed0):
    """[B][n_steps][2 nv] state noise of the disturbed leg: instance b draws from default_rng(977 + seed0 + b)."""
    out = np.empty((B, n_steps, 2 * nv))
    for b in range(B):
        rng = np.random.default_rng(977 + seed0 + b)
        out[b, :, :nv] = rng.normal(0.0, sigma, (n_steps, nv))
        out[b, :, nv:] = rng.normal(0.0, 5.0 * sigma, (n_steps, nv))
    return out


def _cpu_leg(args, table, po, refs: HostRefs, seconds, analytic, noise=None):
    """One CPU leg: MPC loop of the first instances on the host cores (OpenMP over instances) + the same loop for one
    instance on one thread (the reference's default n_threads = 1, ocp_param_base.py:65).  None if the leg does not apply."""
    from oracle.oracle import Oracle  # test infrastructure: only the cpu_baseline leg of bench.py uses it

    avail = ncores()
    B, T = refs.run.shape[0], args.horizon
    o = Oracle(table, po, B)
    if analytic and not o.set_analytic(True):
        return None
    xs0, us0 = refs.x[:, : T + 1].copy(), refs.u[:, :T].copy()
    x0 = xs0[:, 0].copy()
    ref0 = refs.window(0)
    # thread count: the best of a few candidates on one untimed SQP iteration (big hosts oversubscribe easily)
    best, scan = None, {}
    for nt in sorted({min(avail, c) for c in (8, 16, 32, 64, avail)}):
        o.solve(ref0, None, x0, xs0, us0, 1, nthreads=nt)  # thread pool / page warm-up (per-thread workspaces are first touched here)
        t0 = time.perf_counter()
        o.solve(ref0, None, x0, xs0, us0, 1, nthreads=nt)
        el = time.perf_counter() - t0
        scan[str(nt)] = round(el * 1e3, 3)
        if best is None or el < best[0]:
            best = (el, nt)
    cores = best[1]
    # one thread: the same SQP iteration on the first 4 instances, scaled to the sample (a whole-sample pass on one thread
    # would take minutes for the large model)
    n_one = min(B, 4)
    o_one = Oracle(table, po, n_one)
    if analytic:
        o_one.set_analytic(True)
    o_one.solve(ref0[:n_one], None, x0[:n_one], xs0[:n_one], us0[:n_one], 1, nthreads=1)
    t0 = time.perf_counter()
    o_one.solve(ref0[:n_one], None, x0[:n_one], xs0[:n_one], us0[:n_one], 1, nthreads=1)
    scan["1 (scaled from %d instances)" % n_one] = round((time.perf_counter() - t0) * 1e3 * B / n_one, 3)
    o.reset_duals()
    n_max = refs.run.shape[1] - T - 1
    xs, us = xs0, us0
    iters, n_steps = [], 0
    t_start = time.perf_counter()
    while n_steps < n_max and (n_steps < 2 or time.perf_counter() - t_start < seconds):
        if n_steps > 0:
            x0 = xs[:, 1].copy()
            if noise is not None:
                x0 += noise[:B, min(n_steps, noise.shape[1] - 1)]
            xs, us = o.shift_warmstart(xs, us)
        xs, us, K, st = o.solve(refs.window(n_steps), None, x0, xs, us, args.max_iter, nthreads=cores)
        iters.append(float(st["iter"].mean()))
        n_steps += 1
    el = time.perf_counter() - t_start
    o1 = Oracle(table, po, 1)
    if analytic:
        o1.set_analytic(True)
    xs1, us1 = xs0[:1].copy(), us0[:1].copy()
    x01 = xs1[:, 0].copy()
    o1.solve(ref0[:1], None, x01, xs1, us1, 1, nthreads=1)
    o1.reset_duals()
    t1 = time.perf_counter()
    n1 = 0
    while n1 < n_max and (n1 < 2 or time.perf_counter() - t1 < seconds / 3.0):
        if n1 > 0:
            x01 = xs1[:, 1].copy()
            xs1, us1 = o1.shift_warmstart(xs1, us1)
        xs1, us1, _, _ = o1.solve(refs.window(n1)[:1], None, x01, xs1, us1, args.max_iter, nthreads=1)
        n1 += 1
    el1 = time.perf_counter() - t1
    what = ("analytic RNEA / CRBA derivatives (the kernels' node arithmetic compiled for the host, oracle/agx_analytic.cpp) behind the "
            "checker's SQP / Riccati loop" if analytic else "CPU restatement with automatic differentiation (oracle/agx_oracle.cpp: the parity checker)")
    return {
        "value": B * n_steps / el,
        "unit": "MPC steps/s",
        "cores": cores,
        "kind": "port-analytic" if analytic else "port-ad",
        "cpu_quota_cores": cpu_quota(),  # cgroup limit of the container (None = none): what `cores` threads can really draw
        "hardware_threads_visible": avail,
        "thread_scan_ms": scan,  # one SQP iteration of the whole sample per thread count: how the host scales
        "sample": f"{B} instances x {n_steps} steps of the same workload (T={T}), OpenMP over instances; {what}; "
                  f"not the Crocoddyl/mim_solvers binaries; mean SQP iters {np.mean(iters):.2f}",
        "seconds": el,
        "single_thread": {"value": n1 / el1, "unit": "MPC steps/s", "cores": 1, "sample": f"1 instance x {n1} MPC steps, 1 thread"},
    }


def cpu_baseline(args, table, po, refs: HostRefs, seconds):
    """The same MPC steps on the host cores: a bounded sample of the workload (first instances, first steps) sized for
    ~`seconds` of CPU work per leg.  Two legs: "port-analytic" (analytic derivatives, what a Pinocchio-based CPU path does;
    unconstrained workloads) is the baseline when it applies; "port-ad" (the parity checker, dual-number automatic
    differentiation, an order of magnitude slower per node) is always reported next to it."""
    ad = _cpu_leg(args, table, po, refs, seconds / 2.0, False)
    ana = _cpu_leg(args, table, po, refs, seconds, True)
    if ana is None:
        ad["kind"] = "port"
        ad["note"] = "constrained workload: only the automatic-differentiation checker covers the ADMM loop"
        return ad
    ana["port_ad"] = ad
    if args.disturb_sigma > 0.0 and args.workload in ("sine", "generic"):
        # the disturbed loop (same noise law) on the CPU: what the GPU leg `disturbed` is to be read against
        nz = disturbance_noise(refs.run.shape[0], 64, po.nv, args.disturb_sigma, 1234)
        dl = _cpu_leg(args, table, po, refs, seconds / 2.0, True, noise=nz)
        if dl is not None:
            ana["disturbed"] = {k: dl[k] for k in ("value", "unit", "cores", "sample", "seconds")}
    return ana


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = torch = None
    rehearsal = False
    if world > 1:
        import torch
        import torch.distributed as dist

        # AGX_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- lets the N > 1 code path (seeding by
        # global instance index, barriers, MAX of the timed region, status gather, rank-0 report) run on a one-GPU box
        rehearsal = os.environ.get("AGX_BENCH_REHEARSAL", "0") == "1"
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            import datetime

            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=5))
    coll_dev = "cpu" if (rehearsal or world == 1) else "cuda"  # tensors of the collectives: RCCL moves device memory

    T, dt = args.horizon, 0.01
    if args.scaling == "strong":
        lo, hi = batched.shard_bounds(args.batch, rank, world)
        B, first_instance, global_batch = hi - lo, lo, args.batch
    else:
        B, first_instance, global_batch = args.batch, rank * args.batch, world * args.batch
    table, tcp, po = make_problem(T, args.workload)
    hip = backend.HipOcp(table, po, B, device=local_rank)
    if args.quorum < 1.0:
        hip.set_quorum(args.quorum, args.quorum)
    n_extra = 0 if args.no_batch1 else 100
    n_points = args.warmup + args.steps + T + 2 + min(10, T // 2) + n_extra  # + in-situ profile steps + the extra legs
    # per-instance seeds follow the GLOBAL instance index so every rank works on different instances
    nv = table.nv
    seed0 = 1234 + first_instance
    q0, amp, puls, scale, t0 = workloads.sine_batch_params(B, nv=nv, seed0=seed0, q0=(None if nv == 7 else np.zeros(nv)),
                                                           lower=table.lower_position_limit, upper=table.upper_position_limit)
    w = dict(workloads.SINE_WEIGHTS)
    if args.workload == "humanoid":
        hip.sine_trajectory(n_points, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = "synthetic 30-DoF humanoid tree (seed 7) sine_wave_configuration_space, goal-reaching costs"
    elif args.workload == "collision":
        hip.sine_trajectory(n_points, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = ("Panda 7-DoF sine_wave_configuration_space, collision-avoidance costs + distance >= 1 cm constraint "
                         "(ocp_traj_tracking_collision_avoidance.yaml; ADMM, max_qp_iters 100)")
    elif args.workload == "cartesian":
        # BASELINE configs[2] as written: sine_wave_cartesian_space references (inverse kinematics of every instance and
        # point by k_cartesian_sine_ik, outside the timed region) resident in HBM, collision-avoidance costs + constraint
        cq0, camp, cpuls = workloads.cartesian_sine_batch_params(B, seed0=seed0, lower=table.lower_position_limit,
                                                                 upper=table.upper_position_limit)
        hip.cartesian_sine_trajectory(n_points, dt, cq0, camp, cpuls, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = ("Panda 7-DoF sine_wave_cartesian_space (amplitude (0.1, 0.1, 0) m x U(0.5, 1.2), IK on the device at setup), "
                         "collision-avoidance costs + distance >= 1 cm constraint (ocp_traj_tracking_collision_avoidance.yaml; ADMM, "
                         "max_qp_iters 100)")
    elif args.workload == "sine":
        hip.sine_trajectory(n_points, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = "Panda 7-DoF sine_wave_configuration_space, ocp_goal_reaching.yaml costs"
    else:
        # BASELINE configs[3]: q/dq/ddq arrays from seeded smooth random accelerations (tests/test_generic_trajectory.py:147-160
        # upstream), pick-and-place cost set and weights (trajectory_weigths_params.yaml:4-9)
        w = dict(w_q=3.0, w_qdot=0.12, w_effort=8e-4, w_pose=0.0)
        gq, gdq, gddq = workloads.generic_batch_arrays(B, n_points, dt, seed0=seed0, q0=q0)
        hip.generic_trajectory(gq, gdq, gddq, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = "Panda 7-DoF generic_trajectory (seeded smooth random accelerations), pick-and-place costs (control_reg + state_reg)"

    def step(k):
        if args.loop == "feedback" and k > 0:
            hip.feedback_rollout(int(round(dt / 1e-3)), 1e-3)
            hip.mpc_step(k, args.max_iter, first=2)
        else:
            hip.mpc_step(k, args.max_iter, first=(k == 0))
        return hip.download_first(copy=False)

    def sync_all():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        hip.sync()

    iters, iters_max, solved_hist = [], [], []
    for k in range(args.warmup):
        st = step(k)[3]
    sync_all()
    t_start = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        st = step(k)[3]
        iters.append(float(st["iter"].mean()))
        iters_max.append(int(st["iter"].max()))
        if k % 8 == 0:
            solved_hist.append(float(np.asarray(st["solved"]).mean()))
    sync_all()
    elapsed = time.perf_counter() - t_start
    solved_hist.append(float(np.asarray(st["solved"]).mean()))
    solved_local = np.asarray(st["solved"], dtype=np.float64).copy()
    iter_local = np.asarray(st["iter"], dtype=np.float64).copy()
    solved_frac, status_ranks = float(solved_local.mean()), 1
    if dist is not None:
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the step closes with ONE all_gather of the per-instance status words; ranks may hold B or B +- 1
        # instances under strong scaling: padded to the largest shard
        bmax = -(-global_batch // world) if args.scaling == "strong" else B
        mine = torch.full((bmax, 2), -1.0, dtype=torch.float64, device=coll_dev)
        mine[:B, 0] = torch.from_numpy(solved_local).to(coll_dev)
        mine[:B, 1] = torch.from_numpy(iter_local).to(coll_dev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        allst = torch.cat(every).cpu().numpy()
        valid = allst[:, 0] >= 0
        solved_frac, status_ranks = float(allst[valid, 0].mean()), world
        assert int(valid.sum()) == global_batch, (int(valid.sum()), global_batch)

    # per-kernel device time, measured live and in situ: a few more MPC steps of the same loop with
    # HIP events around every launch on the solver's stream (outside the timed region above)
    kernels = {}
    k_next = args.warmup + args.steps
    n_prof = min(10, T // 2)
    if rank == 0:
        # the large-model derivative kernel takes running and terminal nodes in one launch
        nodes = {"calc_qp": B * (T + 1) if nv > 8 else B * T, "riccati": B * (T + 1), "step": B * (T + 1)}
        ALGO = algo_doubles(nv)
        hip.profile(True)
        for k in range(k_next, k_next + n_prof):
            step(k)
        ms_sum, cnt = hip.profile(False)
        for i, name in enumerate(("calc_qp", "riccati", "step")):
            ms = ms_sum[i] / max(cnt[i], 1)
            algo = ALGO[name] * 8 * nodes[name]
            kernels[name] = {"ms": ms, "launches": cnt[i], "algorithmic_bytes": algo, "GBps": algo / (ms * 1e-3) / 1e9,
                             "frac_hbm": algo / (ms * 1e-3) / HBM_PEAK}
    else:
        for k in range(k_next, k_next + n_prof):
            step(k)
    k_next += n_prof

    result = None
    extra = world == 1 and not args.no_batch1
    if rank == 0:
        k1 = kernels["calc_qp"]
        # HBM bytes from separate rocprofv3 --pmc passes (scripts/collect_profiles.sh), committed under profiles/:
        # never measured by this run
        traffic, step_traffic, tsrc = None, None, None
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                doc = json.loads(tfile.read_text())
                traffic = doc.get(f"{K1_NAME[args.workload].replace('_coll', '')}:{args.workload},B={B},T={T}")
                step_traffic = doc.get(f"step:{args.workload},B={B},T={T}")
                tsrc = doc.get("_source")
            except Exception:
                traffic = None
        ms_step = elapsed / args.steps * 1e3
        result = {
            "metric": f"MPC steps/sec (horizon={T}, Panda 7-DoF)",
            "value": global_batch * args.steps / elapsed,
            "unit": "MPC steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{workload_name}, horizon={T}, "
                            f"dt=0.01, batch={B} independent MPC instances per GPU (seed 1234+b), "
                            + ("closed loop on own prediction" if args.loop == "prediction" else "closed loop through the Riccati feedback rollout at 1 kHz"),
                "horizon": T,
                "batch_per_gpu": B,
                "global_batch": global_batch,
                "max_iter": args.max_iter,
                "termination_tolerance": 1e-3,
                "parallelism": f"batch-sharded x{world} ({args.scaling} scaling), no collective while solving; "
                               f"status words of {status_ranks} rank(s) all_gathered after the timed region",
                "batch_quorum": args.quorum,
                "parity": PARITY[args.workload],
                "mean_sqp_iters_per_step": float(np.mean(iters)),
                "mean_sqp_iters_of_slowest_instance": float(np.mean(iters_max)),  # what a batch step costs
                "solved_fraction_last_step": solved_frac,
                "solved_fraction_mean": float(np.mean(solved_hist)),  # every 8th step of the timed region
                "step_includes": "window select, x0<-xs[1], warm-start shift, SQP solve, D2H of us[0],K[0],x1,status",
            },
            "roofline": {
                "kernel": K1_TEXT[K1_NAME[args.workload]],
                "bound": "hbm",
                "achieved": k1["GBps"],
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": k1["frac_hbm"],
                "traffic": traffic,
                "traffic_source": ((tsrc or "profiles/pmc_traffic.json") + " (separate rocprofv3 --pmc passes, not measured by this run)") if traffic else None,
                "avg_launch_ms": k1["ms"],
                "algorithmic_bytes_per_launch": k1["algorithmic_bytes"],
            },
            "kernels": kernels,
        }
        if step_traffic:
            # all kernels of one MPC step: counter bytes (profiles/) over the step time of THIS run
            result["roofline_step"] = {"traffic_per_step": step_traffic, "GBps": step_traffic / (ms_step * 1e-3) / 1e9,
                                       "frac": step_traffic / (ms_step * 1e-3) / HBM_PEAK,
                                       "note": "HBM bytes of every kernel of one MPC step (profiles/, WRITE_SIZE + 2 FETCH_SIZE) / ms_per_step of this run"}
        if args.workload in ("collision", "cartesian"):
            result["metric"] = f"MPC steps/sec (horizon={T}, Panda 7-DoF, collision avoidance)"
        if args.workload == "humanoid":
            result["metric"] = f"MPC steps/sec (horizon={T}, 30-DoF humanoid)"
        if extra and args.workload == "sine":
            # BASELINE.json configs[1]: the same workload at batch = 1 (latency of one controller)
            h1 = backend.HipOcp(table, po, 1, device=local_rank)
            p1 = workloads.sine_batch_params(1, lower=table.lower_position_limit, upper=table.upper_position_limit)
            n1 = 200
            np1 = args.warmup + n1 + T + 2
            h1.sine_trajectory(np1, dt, *p1, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
            for k in range(args.warmup):
                h1.mpc_step(k, args.max_iter, first=(k == 0))
                h1.download_first(copy=False)
            h1.sync()
            lat = []
            for k in range(args.warmup, args.warmup + n1):
                t1 = time.perf_counter()
                h1.mpc_step(k, args.max_iter, first=(k == 0))
                h1.download_first(copy=False)
                lat.append((time.perf_counter() - t1) * 1e3)
            lat = np.sort(np.array(lat))
            ms1 = float(lat.mean())
            result["batch1"] = {"ms_per_step": ms1, "median_ms": float(np.median(lat)), "p99_ms": float(lat[int(0.99 * (len(lat) - 1))]),
                                "steps": int(len(lat)), "value": 1e3 / ms1, "unit": "MPC steps/s",
                                "workload": "same, batch = 1 (BASELINE.json configs[1]); per-step host wall time incl. D2H; Riccati sweeps by the exact "
                                            "two-level scheme of small batches (csrc/agx_riccati_mx2.hpp; AGX_MX2_SEGMENTS=0 restores the one-wave sweep)"}
            h1.close()
            # ... and at batch 128: what one GPU holds when configs[3]'s 1024 instances are sharded over the 8 GPUs of a node
            h8 = backend.HipOcp(table, po, 128, device=local_rank)
            p8 = workloads.sine_batch_params(128, lower=table.lower_position_limit, upper=table.upper_position_limit)
            n8 = 100
            h8.sine_trajectory(args.warmup + n8 + T + 2, dt, *p8, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
            for k in range(args.warmup):
                h8.mpc_step(k, args.max_iter, first=(k == 0))
                h8.download_first(copy=False)
            h8.sync()
            lat8 = []
            for k in range(args.warmup, args.warmup + n8):
                t1 = time.perf_counter()
                h8.mpc_step(k, args.max_iter, first=False)
                h8.download_first(copy=False)
                lat8.append((time.perf_counter() - t1) * 1e3)
            lat8 = np.sort(np.array(lat8))
            result["batch128"] = {"median_ms": float(np.median(lat8)), "p99_ms": float(lat8[int(0.99 * (len(lat8) - 1))]), "steps": int(len(lat8)),
                                  "value": 128 * 1e3 / float(lat8.mean()), "unit": "MPC steps/s",
                                  "workload": "same, batch = 128 on one GPU (the per-GPU share of a global batch of 1024 on 8 GPUs: the strong-scaling point; "
                                              "two-level Riccati sweeps)"}
            h8.close()
        if extra and args.workload in ("sine", "generic") and args.disturb_sigma > 0.0:
            # Disturbed leg: closed loop on the prediction never leaves the easy path (one SQP iteration, alpha = 1).  Here the
            # measured state is the predicted one plus seeded noise: several SQP iterations per step, rejected step lengths,
            # the regularisation schedule at work -- the same loop, same noise, is timed on the CPU below.
            nd = 24
            noise = disturbance_noise(B, nd + 2, nv, args.disturb_sigma, seed0)
            hist = np.zeros(args.max_iter + 1, dtype=np.int64)
            flags_any = backtracked = 0
            x1 = np.array(hip.download_first(copy=True)[2])
            for i in range(2):  # settle into the disturbed regime
                hip.upload_x0(x1 + noise[:, i])
                hip.mpc_step(k_next + i, args.max_iter, first=2)
                x1 = np.array(hip.download_first(copy=False)[2])
            hip.sync()
            t1 = time.perf_counter()
            for i in range(2, nd + 2):
                hip.upload_x0(x1 + noise[:, i])
                hip.mpc_step(k_next + i, args.max_iter, first=2)
                _, _, x1v, std = hip.download_first(copy=False)
                x1 = np.array(x1v)
                hist += np.bincount(np.minimum(np.asarray(std["iter"]).astype(np.int64), args.max_iter), minlength=args.max_iter + 1)
                fl = np.asarray(std["flags"]).astype(np.int64)
                flags_any += int(np.count_nonzero(fl & 2))
                backtracked += int(np.count_nonzero(fl & 4))
            hip.sync()
            msd = (time.perf_counter() - t1) / nd * 1e3
            k_next += nd + 2
            result["disturbed"] = {"ms_per_step": msd, "value": B / (msd * 1e-3), "unit": "MPC steps/s", "sigma_q_rad": args.disturb_sigma,
                                   "sigma_v_rad_s": 5 * args.disturb_sigma, "steps": nd,
                                   "sqp_iter_histogram": hist.tolist(), "mean_sqp_iters": float((hist * np.arange(hist.size)).sum() / max(hist.sum(), 1)),
                                   "instance_steps_that_backtracked": backtracked,  # some step length < 1 was tried (status flag bit 2)
                                   "instance_steps_with_a_failed_line_search": flags_any,  # all ten step lengths rejected (bit 1)
                                   "note": "x0 of every step = predicted state + seeded Gaussian noise (uploaded from the host: 8 B nx bytes per "
                                           "step inside the timed loop); statuses read every step"}
        if extra and args.max_iter != 3:
            # also the pick-and-place iteration cap (max_iter 3) on the same workload
            na = 8
            hip.sync()
            t1 = time.perf_counter()
            for k in range(k_next, k_next + na):
                hip.mpc_step(k, 3, first=False)
                hip.download_first(copy=False)
            hip.sync()
            msa = (time.perf_counter() - t1) / na * 1e3
            k_next += na
            result["max_iter_3"] = {"ms_per_step": msa, "value": B / (msa * 1e-3), "unit": "MPC steps/s",
                                    "note": "same workload with the pick-and-place cap max_iter = 3"}
        if extra and args.workload in ("collision", "cartesian") and args.quorum >= 1.0:
            # batch policy leg: the same loop with a quorum of 0.985 (agx_ocp_set_quorum) -- the slowest 1.5 % of the instances
            # are cut every step and carry their iterate over, as a lone controller hitting max_solve_time; upstream has no
            # such cut and neither has the CPU leg, so this number is reported NEXT TO `value`, never as it
            nq = 20
            hip.set_quorum(0.985, 0.985)
            for k in range(k_next, k_next + 2):
                hip.mpc_step(k, args.max_iter, first=False)
                hip.download_first(copy=False)
            hip.sync()
            t1 = time.perf_counter()
            sq = []
            for k in range(k_next + 2, k_next + 2 + nq):
                hip.mpc_step(k, args.max_iter, first=False)
                sq.append(float(np.asarray(hip.download_first(copy=False)[3]["solved"]).mean()))
            hip.sync()
            msq = (time.perf_counter() - t1) / nq * 1e3
            k_next += nq + 2
            hip.set_quorum(1.0, 1.0)
            result["quorum_0985"] = {"ms_per_step": msq, "value": B / (msq * 1e-3), "unit": "MPC steps/s", "solved_fraction_mean": float(np.mean(sq)),
                                     "note": "batch quorum 0.985 for the SQP and the ADMM loop; counts every instance of the batch as a step, "
                                             "solved or cut"}
        if extra:
            # the same step with the FULL result download (xs, us, K of every node): device-side snapshot in the
            # solver's stream, drained into page-locked arrays by the copy stream while the next step is being solved
            # (agx_ocp_download_async); every step's results are complete on the host before the step after next starts
            nfull = 6
            shapes = ((B, T + 1, 2 * nv), (B, T, nv), (B, T, nv, 2 * nv))
            sizes = [int(np.prod(sh)) for sh in shapes]

            def result_block():  # xs | us | K in ONE page-locked block: one transfer per step
                blk = backend.pinned_array((sum(sizes),))
                offs = np.cumsum([0] + sizes)
                return tuple(blk[offs[i]:offs[i + 1]].reshape(shapes[i]) for i in range(3))

            res = [result_block() for _ in range(2)]
            hip.sync()
            t1 = time.perf_counter()
            for i, k in enumerate(range(k_next, k_next + nfull)):
                hip.mpc_step(k, args.max_iter, first=False)
                hip.download_wait()                 # results of step k - 1 (they travelled during this solve)
                hip.download_async(*res[i % 2])
            hip.download_wait()
            msf = (time.perf_counter() - t1) / nfull * 1e3
            k_next += nfull
            nbytes = int(sum(a.nbytes for a in res[0]))
            t2 = time.perf_counter()
            hip.download_async(*res[0])
            hip.download_wait()
            d2h = nbytes / (time.perf_counter() - t2) / 1e9
            result["full_download"] = {"ms_per_step": msf, "value": B / (msf * 1e-3), "unit": "MPC steps/s",
                                       "bytes_per_step": nbytes, "d2h_GBps_pinned": d2h,
                                       "note": "PCIe-inclusive: xs, us, K of all nodes copied to page-locked host arrays every step "
                                               "(agx_ocp_download_async: snapshot + copy stream, overlapped with the next solve)"}
    # ---- host_refs leg (the step as the reference's caller sees it): the reference tiles of every step come
    # from the host (agx_ocp_set_refs: H2D of [B][T+1][stride]), then shift + solve + download of the first-node results.
    # With N > 1 ranks the tiles of the whole job originate on rank 0 and travel through batched.scatter_rows (RCCL
    # point-to-point), the first-node results go back through gather_rows: the data-path payload of SURVEY 8(e).
    host_leg = not args.no_batch1
    n_cpu = args.cpu_instances or min(2 * ncores(), 256, B)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    refs = None
    if host_leg:
        refs = HostRefs(hip, po, w, min(n_points, max(k_next + T + 16, T + 1 + 64)), B)
    elif want_cpu:
        refs = HostRefs(hip, po, w, min(n_points, T + 1 + 64), n_cpu)
    host_err = None
    if host_leg:
        # Three page-locked host tiles in rotation: while step k is solved, the tile of step k+1 travels to the handle's second
        # device tile on the copy stream (agx_ocp_set_refs_async) and the tile of step k+2 is assembled by worker threads
        # (numpy copies and the solve call release the GIL).  An extra leg after the timed region: whatever happens in it, the
        # measured line above is still printed.  A rank whose LOCAL work fails keeps taking part in the collectives of every step,
        # so nobody is left waiting inside gather_rows / scatter_rows; the error flags are MAX-reduced after the leg.
        import concurrent.futures as cf

        nh, n_glob = 8, (global_batch if args.scaling == "strong" else world * B)
        n_workers = max(1, min(8, ncores() - 1))
        pool = cf.ThreadPoolExecutor(n_workers)
        tiles = [backend.pinned_array((B, T + 1, po.stride)) for _ in range(3)]
        cuts = np.linspace(0, B, n_workers + 1).astype(int)

        def build(k, out):
            return [pool.submit(refs.window, k, out, int(lo), int(hi)) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]

        def through_ranks(tile):  # N > 1: the tiles of the whole job originate on rank 0 (stands in for one generator process)
            if world == 1:
                return tile
            full = batched.gather_rows(tile, n_glob, device=coll_dev)
            tile[...] = batched.scatter_rows(full, n_glob, device=coll_dev)
            return tile

        try:
            cf.wait(build(k_next, tiles[0]))
            hip.set_refs_async(through_ranks(tiles[0]))
            pending = build(k_next + 1, tiles[1])
        except Exception as e:  # noqa: BLE001
            host_err, pending = repr(e), []
        sync_all()
        t1 = time.perf_counter()
        for i in range(nh):
            first = np.zeros((B, nv + nv * 2 * nv + 2 * nv))
            nxt = tiles[(i + 1) % 3]
            try:
                if host_err is None:
                    hip.refs_activate()           # tile i, staged during step i - 1
                    cf.wait(pending)
            except Exception as e:  # noqa: BLE001
                host_err = repr(e)
            nxt = through_ranks(nxt)
            try:
                if host_err is None:
                    hip.set_refs_async(nxt)   # travels while step i is solved
                    pending = build(k_next + i + 2, tiles[(i + 2) % 3])
                    hip.x0_from_prediction()
                    hip.shift_warmstart()
                    hip.solve_resident(args.max_iter)
                    us0, K0, x1, _ = hip.download_first(copy=True)
                    first = np.concatenate([us0, K0.reshape(B, -1), x1], 1)
            except Exception as e:  # noqa: BLE001
                host_err = repr(e)
            if world > 1:
                batched.gather_rows(first, n_glob, device=coll_dev)
        if dist is not None:
            flag = torch.tensor([0.0 if host_err is None else 1.0], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag.item()) > 0.0 and host_err is None:
                host_err = "the leg failed on another rank"
        sync_all()
        msh = (time.perf_counter() - t1) / nh * 1e3
        try:
            cf.wait(pending)
            hip.refs_wait()
        except Exception:  # noqa: BLE001
            pass
        pool.shutdown()
        # the upload alone (same tile, page-locked, nothing else running): the rate of the link
        h2d = None
        try:
            if host_err is None:
                hip.sync()
                t2 = time.perf_counter()
                for _ in range(3):
                    hip.set_refs_async(tiles[0])
                    hip.refs_wait()
                h2d = tiles[0].nbytes * 3 / (time.perf_counter() - t2) / 1e9
                hip.refs_activate()
        except Exception:  # noqa: BLE001
            h2d = None
        k_next += nh + 2
        if rank == 0 and host_err is not None:
            result["host_refs"] = {"ms_per_step": None, "value": None, "unit": "MPC steps/s", "note": f"leg failed: {host_err}"}
        elif rank == 0:
            result["host_refs"] = {"ms_per_step": msh, "value": global_batch / (msh * 1e-3), "unit": "MPC steps/s",
                                   "h2d_bytes_per_step_per_gpu": int(tiles[0].nbytes), "h2d_GBps_pinned": h2d, "host_threads": n_workers,
                                   "note": "PCIe-inclusive: reference tiles [B][T+1][stride] assembled on the host every step in page-locked "
                                           "memory and staged by agx_ocp_set_refs_async / agx_ocp_refs_activate (the tile of step k+1 "
                                           "travels while step k is solved), shift, solve, first-node download"
                                           + ("; tiles scattered from / results gathered to rank 0 over RCCL" if world > 1 else "")}
    if rank == 0:
        if want_cpu:
            try:
                result["cpu_baseline"] = cpu_baseline(args, table, po, refs.first(n_cpu), args.cpu_seconds)
            except Exception as e:  # the GPU numbers stand on their own
                result["cpu_baseline"] = {"value": None, "unit": "MPC steps/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": f"failed: {e!r}"}
        print(json.dumps(result), flush=True)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    hip.close()


if __name__ == "__main__":
    main()
