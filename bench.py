#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched iLQR hot path on MI355X.

Metric (BASELINE.json): iLQR iterations/sec (7-DoF, T=200, batch 4096) + final-cost rel-err vs the reference.
Workload at N=1: config C3 of SURVEY.md 8(d) = configs[2] of BASELINE.json, the configuration the metric is quoted on:
PosOrn 1st-order System on the 7-DoF Panda chain, T=200, dt=0.05, B=4096 seeded instances, AL-iLQR with the
tutorial's inequality row (q_6 <= 2.0, penalty .25, scaling 1.1, multiplier update every 5 iterations), 20 iterations
per solve, line search on, early stop off (fixed work).  A "step" = one such solve of the whole batch from U0.
One problem-iteration = one backward Riccati sweep + one accepted forward rollout of one instance.

N>1 (driver: torch.distributed.run, one rank per GPU): every rank solves its own 4096-instance shard (weak
scaling, no data-path collective); the only exchange is the all-gather of the converged costs over RCCL, once per
step, inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_traffic(category):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary (profiles/*_kernels.txt, two
    separate passes: FETCH_SIZE, WRITE_SIZE; on gfx950 FETCH_SIZE counts half of a coalesced stream, so it is doubled --
    MI355X_MICROARCH.md).  bench.py cannot run the profiler around itself; the summary is produced by
    scripts/collect_profiles.sh with this same command.  Returns (bytes or None, source)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_kernels.txt")))
    if not files:
        return None, None
    for f in reversed(files):  # newest summary that holds the counter passes for this kernel (summaries of other solvers have none)
        fetch = write = None
        for line in open(f):
            if ("k_" + category) not in line:
                continue
            if "FETCH_SIZE" in line:
                fetch = float(line.split("avg=")[1].split()[0]) * 1024 * 2
            elif "WRITE_SIZE" in line:
                write = float(line.split("avg=")[1].split()[0]) * 1024
        if fetch is not None and write is not None:
            return int(fetch + write), "profiles/" + os.path.basename(f)
    return None, None


def algorithmic_bytes(nx, nu, m, T, B):
    """fp64 bytes one launch of each kernel must move if A_k, B_k are never stored (SURVEY.md 8d model):
    backward: read x,u ; write K,d (+ read lambda, I_k for AL)
    forward (per line-search trial): read K,d,x,u ; write x,u (+ write I_k, read lambda for AL)."""
    steps = (T - 1) * B
    bwd = 8 * ((nx + nu) + (nu * nx + nu) + 2 * m) * steps
    fwd = 8 * ((nu * nx + nu) + (nx + nu) + (nx + nu) + 2 * m) * steps
    return bwd, fwd


def cpu_baseline(cfg, inp, nb_iter, budget_s=12.0):
    """The CPU oracle (oracle/ilqr_oracle.c, plain C restatement of the reference algorithm, -O3, one thread) timed on
    a bounded sample of the SAME workload on this host.  Reported beside the GPU number, never inside it."""
    from tests.helpers import oracle_solve_instance, panda_segs

    segs = panda_segs()
    oracle_solve_instance(cfg, inp, 0, 1, False, segs)  # warm up (loads the .so)
    n, t0 = 0, time.perf_counter()
    B = inp["q0"].shape[0]
    costs = []
    while n < B:
        r = oracle_solve_instance(cfg, inp, n, nb_iter, False, segs)
        costs.append(r["cost"])
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 8:
            break
    dt = time.perf_counter() - t0
    out = dict(value=n * nb_iter / dt, unit="problem-iterations/s", cores=1, kind="port",
               sample=f"first {n} instances of the same seeded batch x {nb_iter} iterations, {dt:.1f} s, single thread")
    # the same restatement on all host cores (instances are independent; ctypes releases the GIL around the C call): a second,
    # shorter sample, reported beside the single-thread figure
    try:
        from concurrent.futures import ThreadPoolExecutor

        nthr = min(16, max(1, len(os.sched_getaffinity(0))))  # a one-GPU box is given 16 cores' worth of CPU
        if nthr > 1:
            from tests.helpers import oracle_system_of_instance, orc

            pool = min(B, 4 * nthr)
            systems = [oracle_system_of_instance(cfg, inp, i, segs) for i in range(pool)]  # built outside the timed region (Python, holds the GIL)

            def solve(k):
                i, s = k % pool, systems[k % pool]
                u0 = inp["U0"][i].reshape(-1)
                if cfg["solver"] == "al":
                    al = cfg["al"]
                    orc.solve_al(s, inp["A"], inp["b"], inp["lambda0"][i], u0, nb_iter, al["lag"], al["penalty"], al["scaling"], True, False)
                else:
                    orc.solve_recursive(s, u0, nb_iter, True, False)

            deadline = time.perf_counter() + 5.0
            done = [0] * nthr

            def worker(w):
                k = w
                while time.perf_counter() < deadline:
                    solve(k)
                    k += nthr
                    done[w] += 1

            t1 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(worker, range(nthr)))
            dt2 = time.perf_counter() - t1
            out["all_cores"] = dict(value=sum(done) * nb_iter / dt2, cores=nthr, sample=f"{sum(done)} instances, {dt2:.1f} s, {nthr} threads")
    except Exception as e:  # the baseline is informative only
        out["all_cores"] = dict(error=str(e))
    return out, costs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", help="workload (ilqr_planner_amd.workloads.config); C3 = the metric's configuration")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    from ilqr_planner_amd import capi, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}` (WORLD_SIZE={world})", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = workloads.config(args.config)
    B = int(args.batch or cfg["B"])
    nb_iter = int(args.iters or cfg["nb_iter"])
    ctx = capi.Context(local_rank)
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by the library's launches and torch's collectives
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, seed=cfg["seed"] + 1000 * rank)
    p = workloads.load_batch(ctx, desc, inp, B)  # inputs resident in HBM from here on
    psi = None
    if cfg["solver"] == "batch_cp":
        from tests.helpers import psi_of

        psi = psi_of(cfg["psi"], cfg["T"], p.dims.n_u)
    cost_dev = torch.empty(B, dtype=torch.float64, device="cuda")
    gathered = torch.empty(world * B, dtype=torch.float64, device="cuda") if world > 1 else None

    def step():
        if cfg["solver"] == "al":
            p.reset_multipliers()
        workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False, psi=psi)
        p.get_cost_dev(cost_dev.data_ptr())
        if world > 1:
            dist.all_gather_into_tensor(gathered, cost_dev)  # the one collective: converged costs over RCCL/xGMI

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    if not os.environ.get("ILQR_BENCH_NOPROF"):
        ctx.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    prof = {n: ctx.profile_get(w) for n, w in (("rollout", capi.PROF_ROLLOUT), ("backward", capi.PROF_BACKWARD), ("forward", capi.PROF_FORWARD),
                                               ("apply", capi.PROF_APPLY))}
    cost = cost_dev.cpu().numpy()
    status = p.status()
    at = p.trace(nb_iter)[1] if cfg["solver"] != "batch_cp" else None

    if rank == 0:
        nx, nu, m = p.dims.n_x, p.dims.n_u, p.m
        bwd_bytes, fwd_bytes = algorithmic_bytes(nx, nu, m, cfg["T"], B)
        # mean number of step sizes the reference's do/while would have tried, and the share of instance-iterations whose
        # winner was not alpha = 1 (those are re-rolled by the second forward pass)
        trials = float(np.mean(1 + np.round(-np.log2(at)))) if at is not None else 1.0
        frac_apply = float(np.mean(at < 1.0)) if at is not None else 0.0
        v1 = os.environ.get("ILQR_HIP_PATH") == "v1"
        kern = {}
        # algorithmic bytes per launch.  backward: read x,u (+lambda,I) ; write K,d.  forward: v1 = one full read+write per
        # sequential trial; v2 = ONE pass over K,d,x,u (+ write of x(1),u(1)) whatever the number of trials.  "apply" is the
        # elementwise blend of the winner (read xbar,ubar,x(1),u(1); write x,u) for the instance-iterations with alpha != 1.
        blend_bytes = 8 * 3 * (nx + nu) * (cfg["T"] - 1) * B * frac_apply
        for name, byts in (("backward", bwd_bytes), ("forward", fwd_bytes * (trials if v1 else 1.0)), ("apply", blend_bytes)):
            ms, n = prof[name]
            if n:
                avg = ms / n
                kern[name] = dict(avg_ms=avg, launches=n, total_ms=ms, alg_bytes=byts, gbs=byts / (avg * 1e-3) / 1e9)
        dom = max(kern, key=lambda k: kern[k]["total_ms"]) if kern else None
        roof = None
        if dom:
            k = kern[dom]
            traffic, traffic_src = pmc_traffic(dom) if args.config == "C3" and not v1 else (None, None)
            roof = dict(bound="hbm", kernel=dom, achieved=round(k["gbs"], 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(k["gbs"] / HBM_PEAK_GBS, 5),
                        traffic=traffic, traffic_source=traffic_src, avg_launch_ms=round(k["avg_ms"], 4), alg_bytes_per_launch=int(k["alg_bytes"]),
                        other={n: dict(avg_launch_ms=round(v["avg_ms"], 4), achieved=round(v["gbs"], 2)) for n, v in kern.items() if n != dom},
                        mean_line_search_trials=round(trials, 3))
        out = {
            "metric": "iLQR iterations/sec (7-DoF, T=200, batch 4096) + final-cost rel-err vs Eigen ref",
            "value": world * B * nb_iter * args.steps / elapsed,
            "unit": "problem-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {('PosOrn', 'PosOrnTime', 'JointSpace', 'JointSpaceTime')[cfg['kind']]} nb_deriv={cfg['nb_deriv']} "
                                   f"{'7-DoF Panda chain' if cfg['kind'] < 2 else str(cfg.get('dof', 7)) + ' joints'}, "
                                   f"T={cfg['T']}, batch {B}/GPU, solver={cfg['solver']}, {nb_iter} iterations/solve, line search on, early stop off",
                       "global_batch": world * B, "horizon": cfg["T"], "iterations_per_step": nb_iter, "parallelism": f"instances sharded x{world}"},
            "batch_sweeps_per_s": nb_iter * args.steps / elapsed,
            "nonfinite_frac": float(np.mean((status & 1) != 0)),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1 and cfg["solver"] != "batch_cp":  # the CPU baseline belongs to the N = 1 line only
            cb, ccost = cpu_baseline(cfg, inp, nb_iter)
            out["cpu_baseline"] = cb
            # the reference's own published timing, for orientation only: not measured here, other hardware, other horizon
            cb["reference_published_not_measured_here"] = dict(
                value=745, unit="iterations/s", what="ILQRRecursive, PosOrn 1st order, T=100, one instance, one thread, unknown CPU",
                source="pylqr_planner/Tutorials/POS_ORN_SYS.ipynb:342-348 (time= fields of the stored output; BASELINE.md)")
            ref = np.array(ccost)
            rel = np.abs(cost[: len(ref)] - ref) / np.maximum(np.abs(ref), 1e-12)
            out["final_cost_rel_err_vs_oracle"] = {"median": float(np.median(rel)), "p90": float(np.quantile(rel, 0.9)), "max": float(rel.max()),
                                                   "frac_within_1e-4": float(np.mean(rel <= 1e-4)), "n": int(len(ref)),
                                                   "note": "AL-iLQR is a discontinuous map: instances outside 1e-4 are those where the oracle itself moves by >1e-7 under a 1e-15 perturbation of q0 (DESIGN.md, Parity)"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    p.close()
    ctx.close()


if __name__ == "__main__":
    main()
