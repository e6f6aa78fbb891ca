#!/usr/bin/env python3
"""bench.py -- MPC steps/s of the MI355X-native OCP solve path.

One "step" = one receding-horizon MPC step (MPC.run, agimus_controller/agimus_controller/mpc.py:32-66)
of every instance of the batch, fully device resident: horizon window of the resident sine-wave
reference, x0 <- previous xs[1], warm-start shift, SQP solve (CSQP semantics,
max_iter / tol of the ROS defaults), then the download of what the controller publishes
(us[0], K[0], x1 and the solver status).  Inputs sit in HBM before the timed region starts.

  python bench.py [--gpus N --steps K --warmup W --batch B --horizon T --max-iter I]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU).  Instances are
independent, so ranks share nothing on the data path: the only collectives are the barriers and
the MAX over ranks of the timed region.  Per-GPU work is fixed: scaling = "weak".
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

from agimus_controller_amd import _abi, backend, workloads  # noqa: E402
from agimus_controller_amd.factory import robot_tables as rt  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
# Algorithmic doubles per node and launch (nv = 7): K1 read x,u + reference tile and
# write the 673-double derivative tile; K2+K3 one Riccati backward + one linear forward; K4 one trial.
ALGO_DOUBLES = {"calc_qp": 775, "riccati": 778 + 448, "step": 138}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="MPC instances per GPU (default: the BASELINE.json shape of the workload: "
                                                            "1024; collision / cartesian 256; humanoid 512)")
    ap.add_argument("--horizon", type=int, default=None, help="default: 100; collision / cartesian 200; humanoid 50")
    ap.add_argument("--max-iter", type=int, default=10, help="SQP iteration cap (ROS default 10)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch1", action="store_true", help="skip the batch = 1 latency leg (profiling runs)")
    ap.add_argument("--cpu-instances", type=int, default=0, help="instances of the CPU sample (0 = 2 per core)")
    ap.add_argument("--loop", choices=("prediction", "feedback"), default="prediction",
                    help="how the next measured state is produced: previous xs[1] (the reference's dummy_mpc_test) or the Riccati "
                         "feedback law rolled out on the model at 1 kHz (SURVEY 8(f-3))")
    ap.add_argument("--workload", choices=("sine", "generic", "humanoid", "collision", "cartesian"), default="sine",
                    help="sine: BASELINE configs[1] (the headline); generic: configs[3] generic_trajectory + pick-and-place costs; "
                         "humanoid: configs[4] synthetic 30-DoF tree (use --horizon 50 --batch 512); "
                         "collision: configs[2] shape, collision-avoidance cost + distance constraint (use --horizon 200 --batch 256)")
    args = ap.parse_args()
    shape = {"collision": (256, 200), "cartesian": (256, 200), "humanoid": (512, 50)}.get(args.workload, (1024, 100))
    if args.batch is None:
        args.batch = shape[0]
    if args.horizon is None:
        args.horizon = shape[1]
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


def cpu_baseline(args, table, tcp, po, n_steps=24):
    """The same MPC steps on the host cores with the CPU restatement under oracle/ ("port", OpenMP
    over the instances): a bounded sample of the workload (first instances, first steps)."""
    from oracle.oracle import Oracle  # test infrastructure: only this leg of bench.py uses it

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = avail
    B = args.cpu_instances or min(2 * avail, 256)
    T, dt = args.horizon, 0.01
    q0, amp, puls, scale, t0 = workloads.sine_batch_params(B, lower=table.lower_position_limit, upper=table.upper_position_limit)
    o = Oracle(table, po, B)
    w = workloads.SINE_WEIGHTS
    running, terminal = po.running, po.terminal

    def sample(k):
        t = t0 + k * dt
        s = np.clip(t[:, None] / scale, 0.0, 1.0)
        ramp = 10 * s**3 - 15 * s**4 + 6 * s**5
        dramp = np.where((s > 0) & (s < 1), (30 * s**2 - 60 * s**3 + 30 * s**4) / scale, 0.0)
        ddramp = np.where((s > 0) & (s < 1), (60 * s - 180 * s**2 + 120 * s**3) / scale**2, 0.0)
        sw, cw = np.sin(puls * t[:, None]), np.cos(puls * t[:, None])
        q = q0 + amp * ramp * sw
        dq = amp * (dramp * sw + ramp * puls * cw)
        ddq = amp * (ddramp * sw + 2 * dramp * puls * cw - ramp * puls**2 * sw)
        return q, dq, o.rnea(q, dq, ddq).reshape(B, 7), o.frame_placement(tcp, q)

    pts = [sample(k) for k in range(T + 1 + n_steps)]

    def window(k0):
        ref = po.new_ref_tile(B)
        for t in range(T + 1):
            q, dq, u, pose = pts[k0 + t]
            rows, offs = (terminal, po.terminal_offsets) if t == T else (running, po.running_offsets)
            for r, off in zip(rows, offs):
                seg = ref[:, t, off:]
                seg[:, 0] = 1.0
                if r.kind == _abi.RES_STATE:
                    seg[:, 1:15] = np.concatenate([q, dq], 1)
                    seg[:, 15:22], seg[:, 22:29] = w["w_q"], w["w_qdot"]
                elif r.kind == _abi.RES_CONTROL:
                    seg[:, 1:8], seg[:, 8:15] = u, w["w_effort"]
                else:
                    seg[:, 1:13], seg[:, 13:19] = pose, w["w_pose"]
        return ref

    refs = [window(k) for k in range(n_steps)]
    xs = np.stack([np.concatenate([p[0], p[1]], 1) for p in pts[: T + 1]], 1)
    us = np.stack([p[2] for p in pts[:T]], 1)
    x0 = xs[:, 0].copy()
    # thread count: the best of a few candidates on one untimed iteration (big hosts oversubscribe easily)
    best = None
    for nt in sorted({min(avail, c) for c in (16, 32, 64, avail)}):
        o.solve(refs[0], None, x0, xs, us, 1, nthreads=nt)  # thread pool / page warm-up
        t0 = time.perf_counter()
        o.solve(refs[0], None, x0, xs, us, 1, nthreads=nt)
        el = time.perf_counter() - t0
        if best is None or el < best[0]:
            best = (el, nt)
    cores = best[1]
    t_start = time.perf_counter()
    iters = []
    for k in range(n_steps):
        if k > 0:
            x0 = xs[:, 1].copy()
            xs, us = o.shift_warmstart(xs, us)
        xs, us, K, st = o.solve(refs[k], None, x0, xs, us, args.max_iter, nthreads=cores)
        iters.append(float(st["iter"].mean()))
    el = time.perf_counter() - t_start
    # the reference's default n_threads = 1 (ocp_param_base.py:65): one instance on one thread, same MPC loop
    o1 = Oracle(table, po, 1)
    xs1 = np.stack([np.concatenate([p[0][:1], p[1][:1]], 1) for p in pts[: T + 1]], 1)
    us1 = np.stack([p[2][:1] for p in pts[:T]], 1)
    x01 = xs1[:, 0].copy()
    o1.solve(refs[0][:1], None, x01, xs1, us1, 1, nthreads=1)
    t1 = time.perf_counter()
    n1 = 0
    for k in range(min(n_steps, 12)):
        if k > 0:
            x01 = xs1[:, 1].copy()
            xs1, us1 = o1.shift_warmstart(xs1, us1)
        xs1, us1, _, _ = o1.solve(refs[k][:1], None, x01, xs1, us1, args.max_iter, nthreads=1)
        n1 += 1
    el1 = time.perf_counter() - t1
    return {
        "single_thread": {"value": n1 / el1, "unit": "MPC steps/s", "cores": 1, "sample": f"1 instance x {n1} MPC steps, 1 thread"},
        "value": B * n_steps / el,
        "unit": "MPC steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{B} instances x {n_steps} steps of the same workload (T={T}), OpenMP over instances; "
                  f"CPU restatement (oracle/), not the Crocoddyl/mim_solvers binaries; mean SQP iters {np.mean(iters):.2f}",
        "seconds": el,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        # AGX_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- lets the N > 1 code path (seeding by
        # global instance index, barriers, MAX of the timed region, rank-0 report) run on a one-GPU box
        rehearsal = os.environ.get("AGX_BENCH_REHEARSAL", "0") == "1"
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B, T, dt = args.batch, args.horizon, 0.01
    table, tcp, po = make_problem(T, args.workload)
    hip = backend.HipOcp(table, po, B, device=local_rank)
    n_points = args.warmup + max(args.steps, 200) + T + 2 + 32  # + in-situ profile steps + full-download steps
    # per-instance seeds follow the GLOBAL instance index so every rank works on different instances
    nv = table.nv
    q0, amp, puls, scale, t0 = workloads.sine_batch_params(B, nv=nv, seed0=1234 + rank * B, q0=(None if nv == 7 else np.zeros(nv)),
                                                           lower=table.lower_position_limit, upper=table.upper_position_limit)
    w = workloads.SINE_WEIGHTS
    if args.workload == "humanoid":
        hip.sine_trajectory(n_points, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = "synthetic 30-DoF humanoid tree (seed 7) sine_wave_configuration_space, goal-reaching costs"
    elif args.workload == "collision":
        hip.sine_trajectory(n_points, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = ("Panda 7-DoF sine_wave_configuration_space, collision-avoidance costs + distance >= 1 cm constraint "
                         "(ocp_traj_tracking_collision_avoidance.yaml; ADMM, max_qp_iters 100)")
    elif args.workload == "cartesian":
        # BASELINE configs[2] as written: sine_wave_cartesian_space references (lockstep inverse kinematics of the batch,
        # outside the timed region) resident in HBM as q/dq/ddq arrays, collision-avoidance costs + constraint
        cq0, camp, cpuls = workloads.cartesian_sine_batch_params(B, seed0=1234 + rank * B, lower=table.lower_position_limit,
                                                                 upper=table.upper_position_limit)
        gq, gdq, gddq = workloads.cartesian_sine_batch_arrays(hip, tcp, n_points, dt, cq0, camp, cpuls)
        hip.generic_trajectory(gq, gdq, gddq, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = ("Panda 7-DoF sine_wave_cartesian_space (amplitude (0.1, 0.1, 0) m x U(0.5, 1.2), IK on the host at setup), "
                         "collision-avoidance costs + distance >= 1 cm constraint (ocp_traj_tracking_collision_avoidance.yaml; ADMM, "
                         "max_qp_iters 100)")
    elif args.workload == "sine":
        hip.sine_trajectory(n_points, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
        workload_name = "Panda 7-DoF sine_wave_configuration_space, ocp_goal_reaching.yaml costs"
    else:
        # BASELINE configs[3]: q/dq/ddq arrays from seeded smooth random accelerations (tests/test_generic_trajectory.py:147-160
        # upstream), pick-and-place cost set and weights (trajectory_weigths_params.yaml:4-9)
        w = dict(w_q=3.0, w_qdot=0.12, w_effort=8e-4, w_pose=0.0)
        gq, gdq, gddq = workloads.generic_batch_arrays(B, n_points, dt, seed0=1234 + rank * B, q0=q0)
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
            import torch

            dist.barrier()
            torch.cuda.synchronize()
        hip.sync()

    iters, iters_max = [], []
    for k in range(args.warmup):
        st = step(k)[3]
    sync_all()
    t_start = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        st = step(k)[3]
        iters.append(float(st["iter"].mean()))
        iters_max.append(int(st["iter"].max()))
    sync_all()
    elapsed = time.perf_counter() - t_start
    solved_frac = float(st["solved"].mean())
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-kernel device time, measured live and in situ: a few more MPC steps of the same loop with
    # HIP events around every launch on the solver's stream (outside the timed region above)
    kernels = {}
    if rank == 0:
        nodes = {"calc_qp": B * T, "riccati": B * (T + 1), "step": B * (T + 1)}
        ALGO = algo_doubles(nv)
        hip.profile(True)
        k0 = args.warmup + args.steps
        for k in range(k0, k0 + min(10, T // 2)):
            if k + T + 1 <= n_points:
                step(k)
        ms_sum, cnt = hip.profile(False)
        for i, name in enumerate(("calc_qp", "riccati", "step")):
            ms = ms_sum[i] / max(cnt[i], 1)
            algo = ALGO[name] * 8 * nodes[name]
            kernels[name] = {"ms": ms, "launches": cnt[i], "algorithmic_bytes": algo, "GBps": algo / (ms * 1e-3) / 1e9,
                             "frac_hbm": algo / (ms * 1e-3) / HBM_PEAK}

    result = None
    if rank == 0:
        k1 = kernels["calc_qp"]
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic.json"  # HBM bytes per launch from separate rocprofv3 --pmc passes
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get(f"k_calc_qp_lj,B={B},T={T}")
            except Exception:
                traffic = None
        result = {
            "metric": "MPC steps/sec (horizon=100, Panda 7-DoF)",
            "value": world * B * args.steps / elapsed,
            "unit": "MPC steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{workload_name}, horizon={T}, "
                            f"dt=0.01, batch={B} independent MPC instances per GPU (seed 1234+b), "
                            + ("closed loop on own prediction" if args.loop == "prediction" else "closed loop through the Riccati feedback rollout at 1 kHz"),
                "horizon": T,
                "batch_per_gpu": B,
                "global_batch": world * B,
                "max_iter": args.max_iter,
                "termination_tolerance": 1e-3,
                "parallelism": f"batch-sharded x{world}, no data-path collective",
                "mean_sqp_iters_per_step": float(np.mean(iters)),
                "mean_sqp_iters_of_slowest_instance": float(np.mean(iters_max)),  # what a batch step costs
                "solved_fraction_last_step": solved_frac,
                "step_includes": "window select, x0<-xs[1], warm-start shift, SQP solve, D2H of us[0],K[0],x1,status",
            },
            "roofline": {
                "kernel": "k_calc_qp_lj (node-parallel derivative pass, running nodes, 8 lanes per node)",
                "bound": "hbm",
                "achieved": k1["GBps"],
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": k1["frac_hbm"],
                "traffic": traffic,
                "avg_launch_ms": k1["ms"],
                "algorithmic_bytes_per_launch": k1["algorithmic_bytes"],
            },
            "kernels": kernels,
        }
        if args.workload in ("collision", "cartesian"):
            result["metric"] = f"MPC steps/sec (horizon={T}, Panda 7-DoF, collision avoidance)"
            result["roofline"]["kernel"] = "k_calc_qp<7> (one lane per node: problems with a collision cost row do not use the 8-lane kernel yet)"
        if args.workload == "humanoid":
            result["metric"] = f"MPC steps/sec (horizon={T}, 30-DoF humanoid)"
            result["roofline"]["kernel"] = "k_calc_qp_wg<30> (one workgroup per node: LDS-resident dynamics, fp64 MFMA contractions)"
        if world == 1 and not args.no_batch1 and args.workload == "sine":
            # BASELINE.json configs[1]: the same workload at batch = 1 (latency of one controller)
            h1 = backend.HipOcp(table, po, 1, device=local_rank)
            p1 = workloads.sine_batch_params(1, lower=table.lower_position_limit, upper=table.upper_position_limit)
            h1.sine_trajectory(n_points, dt, *p1, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
            n1 = min(200, n_points - T - 2 - args.warmup)
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
                                "workload": "same, batch = 1 (BASELINE.json configs[1]); per-step host wall time incl. D2H"}
            h1.close()
        if world == 1 and not args.no_batch1 and args.max_iter != 3:
            # also the pick-and-place iteration cap (max_iter 3) on the same workload
            ka = args.warmup + args.steps + min(10, T // 2)
            na = 8
            hip.sync()
            t1 = time.perf_counter()
            for k in range(ka, ka + na):
                hip.mpc_step(k, 3, first=False)
                hip.download_first(copy=False)
            hip.sync()
            msa = (time.perf_counter() - t1) / na * 1e3
            result["max_iter_3"] = {"ms_per_step": msa, "value": B / (msa * 1e-3), "unit": "MPC steps/s",
                                    "note": "same workload with the pick-and-place cap max_iter = 3"}
        if world == 1 and not args.no_batch1:
            # the same step with the FULL result download (xs, us, K of every node) into pageable memory
            kf = args.warmup + args.steps + min(10, T // 2) + 8
            t1 = time.perf_counter()
            nfull = 3
            for k in range(kf, kf + nfull):
                hip.mpc_step(k, args.max_iter, first=False)
                hip.download()
            msf = (time.perf_counter() - t1) / nfull * 1e3
            result["full_download"] = {"ms_per_step": msf, "value": B / (msf * 1e-3), "unit": "MPC steps/s",
                                       "bytes_per_step": int(8 * B * ((T + 1) * 2 * nv + T * nv + T * 2 * nv * nv)),
                                       "note": "PCIe-inclusive: xs, us, K of all nodes copied to host every step"}
        if not args.no_cpu_baseline and world == 1 and args.workload == "sine":
            try:
                result["cpu_baseline"] = cpu_baseline(args, table, tcp, po)
            except Exception as e:  # the GPU numbers stand on their own
                result["cpu_baseline"] = {"value": None, "unit": "MPC steps/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": f"failed: {e!r}"}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    hip.close()


if __name__ == "__main__":
    main()
