"""Turns the raw output of scripts/collect_profiles.sh (gpurun_out/prof_<workload>/) into the committed summaries under
profiles/:  r03_kernel_stats_<workload>.txt, r03_pmc_<workload>.txt and the HBM byte counts bench.py cites
(profiles/pmc_traffic.json: bytes per launch of the derivative kernel, bytes per MPC step of all kernels).
HBM bytes = WRITE_SIZE + 2 x FETCH_SIZE, both in KiB per dispatch (MI355X_MICROARCH.md, HBM section: WRITE_SIZE exact for
streaming stores, FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950: doubled = upper bound)."""
import json, pathlib, re, sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
K1 = {"sine": ("k_calc_qp_lj<7, false, false>", "k_calc_qp_lj"), "generic": ("k_calc_qp_lj<7, false, false>", "k_calc_qp_lj"),
      "humanoid": ("k_calc_qp_wg<30>", "k_calc_qp_wg"), "cartesian": ("k_calc_qp_lj<7, false, true>", "k_calc_qp_lj"),
      "collision": ("k_calc_qp_lj<7, false, true>", "k_calc_qp_lj")}
SHAPE = {"sine": (1024, 100), "generic": (1024, 100), "humanoid": (512, 50), "cartesian": (256, 200), "collision": (256, 200)}
N_STEPS = 3 + 20 + 10  # warm-up + timed + in-situ timing steps of the profiled bench command


def parse_pmc(path):
    out, name = {}, None
    for line in path.read_text().splitlines():
        m = re.match(r"(\S.*?) dispatches~ (\d+)", line)
        if m:
            name = m.group(1)
            out.setdefault(name, {"n": int(m.group(2))})
            continue
        m = re.match(r"\s+(\S+)\s+total\s+(\d+)\s+per-dispatch\s+([\d.]+)", line)
        if m and name:
            out[name][m.group(1)] = (float(m.group(2)), float(m.group(3)))
    return out


ROUND = "r03"


def main():
    traffic_file = ROOT / "profiles" / "pmc_traffic.json"
    traffic = json.loads(traffic_file.read_text()) if traffic_file.exists() else {}
    for w in sys.argv[1:]:
        src = ROOT / "gpurun_out" / f"prof_{w}"
        B, T = SHAPE[w]
        cmd = f"AGX_NO_EMPTY_LAUNCHES=1 AGX_K1_FUSED=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {w} --no-cpu-baseline --no-batch1 --steps 20 --warmup 3"
        head = (f"# {cmd}   (MI355X, ROCm 7.2, round 3; scripts/collect_profiles.sh {w})\n"
                f"# B = {B} instances, T = {T}; {N_STEPS} MPC steps in the run; summary of the rocpd database (scripts/rocpd_stats.py).\n"
                "# AGX_NO_EMPTY_LAUNCHES=1 skips the trial launches of the line search when nobody searches, AGX_K1_FUSED=0 launches running and terminal nodes separately.\n")
        if w in ("cartesian", "collision"):
            head += ("# The derivative kernel's AVERAGE covers every launch of the 10 SQP iterations of a step, most of which serve the few instances still\n"
                     "# iterating (DESIGN.md section 5); the launch the roofline of bench.py is quoted on is the full one of the first iteration: the MAX column.\n")
        (ROOT / "profiles" / f"{ROUND}_kernel_stats_{w}.txt").write_text(head + (src / "kernel_stats.txt").read_text())
        fetch, write = parse_pmc(src / "pmc_fetch.txt"), parse_pmc(src / "pmc_write.txt")
        lines = [f"# HBM counters of the same command, separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (KiB per dispatch).",
                 "# bytes = WRITE_SIZE + 2 x FETCH_SIZE (gfx950 read-side correction of MI355X_MICROARCH.md: upper bound for narrow reads).",
                 f"{'kernel':72s} {'calls':>6s} {'fetch KiB/launch':>18s} {'write KiB/launch':>18s} {'MB/launch':>12s} {'MB/step':>10s}"]
        step_total = 0.0
        for name in sorted(set(fetch) | set(write), key=lambda n: -(write.get(n, {}).get("WRITE_SIZE", (0, 0))[0] + 2 * fetch.get(n, {}).get("FETCH_SIZE", (0, 0))[0])):
            f, wr = fetch.get(name, {}).get("FETCH_SIZE", (0.0, 0.0)), write.get(name, {}).get("WRITE_SIZE", (0.0, 0.0))
            n = max(fetch.get(name, {}).get("n", 0), write.get(name, {}).get("n", 0))
            tot = (wr[0] + 2 * f[0]) * 1024
            if "sine_fill" in name or "ws_from_ref" in name or "frame" in name.split("(")[0]:
                continue  # set-up kernels, not part of an MPC step
            step_total += tot
            lines.append(f"{name[:72]:72s} {n:6d} {f[1]:18.1f} {wr[1]:18.1f} {(wr[1] + 2 * f[1]) * 1024 / 1e6:12.2f} {tot / N_STEPS / 1e6:10.2f}")
            if K1[w][0] in name:
                traffic[f"{K1[w][1]}:{w},B={B},T={T}"] = (wr[1] + 2 * f[1]) * 1024
        lines.append(f"# all kernels of an MPC step: {step_total / N_STEPS / 1e6:.1f} MB per step")
        traffic[f"step:{w},B={B},T={T}"] = step_total / N_STEPS
        mf = src / "pmc_mfma.txt"
        if mf.exists():
            lines.append("# fp64 matrix-core counters (separate pass: --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES), per dispatch:")
            for name, d in parse_pmc(mf).items():
                if d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", (0, 0))[0] > 0:
                    lines.append(f"#   {name[:60]:60s} MFMA_MOPS_F64 {d['SQ_INSTS_VALU_MFMA_MOPS_F64'][1]:14.0f}  MFMA_BUSY_CYCLES {d['SQ_VALU_MFMA_BUSY_CYCLES'][1]:14.0f}  SQ_BUSY_CYCLES {d['SQ_BUSY_CYCLES'][1]:14.0f}")
        (ROOT / "profiles" / f"{ROUND}_pmc_{w}.txt").write_text("\n".join(lines) + "\n")
    traffic["_source"] = f"profiles/{ROUND}_pmc_<workload>.txt"
    traffic["_comment"] = ("HBM bytes from separate rocprofv3 --pmc passes: WRITE_SIZE + 2 x FETCH_SIZE (gfx950 read-side correction, upper bound), "
                           "KiB = 1024 B; '<kernel>:<workload>,B,T' = per launch of the derivative kernel, 'step:<workload>,B,T' = all kernels of one MPC step")
    traffic_file.write_text(json.dumps(traffic, indent=1) + "\n")


if __name__ == "__main__":
    main()
