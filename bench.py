#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched iLQR hot path on MI355X.

Metric (BASELINE.json): iLQR iterations/sec (7-DoF, T=200, batch 4096) + final-cost rel-err vs the reference.
Workload at N=1: config C3 of SURVEY.md 8(d) = configs[2] of BASELINE.json, the configuration the metric is quoted on:
PosOrn 1st-order System on the 7-DoF Panda chain, T=200, dt=0.05, B=4096 seeded instances, AL-iLQR with the
tutorial's inequality row (q_6 <= 2.0, penalty .25, scaling 1.1, multiplier update every 5 iterations), 20 iterations
per solve, line search on, early stop off (fixed work).  A "step" = one such solve of the whole batch from U0.
One problem-iteration = one backward Riccati sweep + one accepted forward rollout of one instance.
`--config C4` (configs[3]: PosOrnTime 2nd order, 4096 instances per GPU -- 32768 over 8 GPUs) and `--config C5` (configs[4]:
Batch-CP, T=400, B=8192) run the other BASELINE configurations through the same contract.

N>1 (driver: torch.distributed.run, one rank per GPU): every rank solves its own shard (weak scaling, no data-path
collective); the only exchange is the all-gather of the converged costs over RCCL (sharding.gather_costs), once per step,
inside the timed region.  `--gather-trajectories` adds the gather of X / U (sharding.gather_trajectories).

What is timed: `steps` solves between two fences (device synchronize + barrier + synchronize), per-launch profiling OFF, inputs
resident in HBM.  Kernel durations for the roofline come from a SECOND pass with HIP events on the library's stream, after the
timed region.  The oracle is used only for the `cpu_baseline` sample and the parity report (N = 1), never inside the timed region.
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
METRIC = "iLQR iterations/sec (7-DoF, T=200, batch 4096) + final-cost rel-err vs Eigen ref"


# ----------------------------------------------------------------------------- multi-rank plumbing (covered by tests/test_sharding_gloo.py)

def fence(device_sync, dist=None):
    """Both sides of the timed region: drain the device, meet the other ranks, drain again."""
    device_sync()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    device_sync()


def timed_region(step, steps, device_sync, dist=None):
    """EXACTLY `steps` calls of step() between two fences; returns this rank's wall time in seconds."""
    fence(device_sync, dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence(device_sync, dist)
    return time.perf_counter() - t0


def max_over_ranks(elapsed, dist=None, device="cpu"):
    """MAX of the per-rank wall times (the job is as slow as its slowest shard)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed)
    import torch

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_step(local_cost, total, local_X=None, local_U=None):
    """The one exchange of a step: converged costs of all shards on every rank (and, on request, the trajectories on rank 0)."""
    from ilqr_planner_amd import sharding

    out = dict(cost=sharding.gather_costs(local_cost, total))
    if local_X is not None:
        out["X"] = sharding.gather_trajectories(local_X, total, dst=0)
        out["U"] = sharding.gather_trajectories(local_U, total, dst=0)
    return out


# ----------------------------------------------------------------------------- roofline bookkeeping

def pmc_traffic(config, category):
    """HBM bytes per launch of a kernel category from the committed rocprofv3 --pmc summary of this round (profiles/r03_<config>_kernels.txt;
    two separate passes FETCH_SIZE / WRITE_SIZE; on gfx950 FETCH_SIZE counts half of a coalesced stream and is doubled --
    MI355X_MICROARCH.md).  bench.py cannot run the profiler around itself: the summary is produced by scripts/collect_profiles.sh with
    this same command.  Returns (bytes or None, source)."""
    f = os.path.join(ROOT, "profiles", f"r03_{config.lower()}_kernels.txt")
    if not os.path.exists(f):
        return None, None
    fetch = write = None
    for line in open(f):
        if not line.startswith(f"pmc[{category}]"):
            continue
        if "FETCH_SIZE" in line:
            fetch = float(line.split("avg=")[1].split()[0]) * 1024 * 2
        elif "WRITE_SIZE" in line:
            write = float(line.split("avg=")[1].split()[0]) * 1024
    if fetch is None or write is None:
        return None, None
    return int(fetch + write), "profiles/" + os.path.basename(f)


def algorithmic_bytes(cfg, nx, nu, m, B, fused):
    """fp64 bytes one launch of each kernel category must move (SURVEY.md 8d model: A_k, B_k never stored).
    Riccati solvers, per instance-step:  backward reads xbar, ubar (+ lambda, I_k) and writes K, d -- the fused sweep also reads x(1), u(1)
    and writes the accepted x, u (the APPLY pass it replaces), and neither reads nor writes I_k;  forward = ONE pass over K, d, xbar, ubar
    + write of x(1), u(1) whatever the number of step sizes.
    Batch-CP in coefficient space, per instance and SOLVE: the horizon is walked at the start (reads U0: keypoint states and the quadratic
    forms of the control cost) and at the end (reads U0, writes U = U0 + PSI w and X); an iteration touches keypoint-sized data only."""
    T = cfg["T"]
    if cfg["solver"] == "batch_cp":
        return dict(rollout=8 * (T - 1) * nu * B, apply=8 * (2 * (T - 1) * nu + T * nx) * B, backward=None, forward=None)
    steps = (T - 1) * B
    bwd = 8 * ((nx + nu) + (nu * nx + nu) + (m if fused else 2 * m)) * steps
    if fused:
        bwd += 8 * 2 * (nx + nu) * steps
    fwd = 8 * ((nu * nx + nu) + (nx + nu) + (nx + nu)) * steps
    return dict(backward=bwd, forward=fwd, apply=None, rollout=None)


# ----------------------------------------------------------------------------- CPU baseline (the oracle, timed beside the GPU number)

def cpu_baseline(cfg, inp, nb_iter, psi=None, budget_s=12.0):
    """The CPU oracle (oracle/ilqr_oracle.c, plain C restatement of the reference algorithm, -O3, one thread) timed on a bounded sample
    of the SAME workload on this host; then the same on all host cores.  Reported beside the GPU number, never inside it."""
    from tests.helpers import oracle_solve_instance, oracle_system_of_instance, orc, panda_segs

    segs = panda_segs()

    def solve_one(i, s=None):
        if cfg["solver"] == "batch_cp":
            s = s or oracle_system_of_instance(cfg, inp, i, segs)
            r = orc.solve_batch_cp(s, psi, inp["U0"][i].reshape(-1), nb_iter, False)
            return dict(cost=float(r["trace_cost"][-1]), iters=r["iters"])
        return oracle_solve_instance(cfg, inp, i, nb_iter, False, segs)

    solve_one(0)  # warm up (loads the .so)
    n, t0 = 0, time.perf_counter()
    B = inp["q0"].shape[0]
    res = []
    while n < B:
        res.append(solve_one(n))
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 8:
            break
    dt = time.perf_counter() - t0
    out = dict(value=n * nb_iter / dt, unit="problem-iterations/s", cores=1, kind="port",
               sample=f"first {n} instances of the same seeded batch x {nb_iter} iterations, {dt:.1f} s, single thread")
    try:  # the same restatement on all host cores (instances are independent; ctypes releases the GIL around the C call)
        from concurrent.futures import ThreadPoolExecutor

        nthr = min(16, max(1, len(os.sched_getaffinity(0))))  # a one-GPU box is given 16 cores' worth of CPU
        if nthr > 1:
            pool = min(B, 4 * nthr)
            systems = [oracle_system_of_instance(cfg, inp, i, segs) for i in range(pool)]  # built outside the timed region (Python, holds the GIL)
            deadline = time.perf_counter() + 5.0
            done = [0] * nthr

            def worker(w):
                k = w
                while time.perf_counter() < deadline:
                    i = k % pool
                    if cfg["solver"] == "batch_cp":
                        orc.solve_batch_cp(systems[i], psi, inp["U0"][i].reshape(-1), nb_iter, False)
                    elif cfg["solver"] == "al":
                        al = cfg["al"]
                        orc.solve_al(systems[i], inp["A"], inp["b"], inp["lambda0"][i], inp["U0"][i].reshape(-1), nb_iter, al["lag"], al["penalty"],
                                     al["scaling"], True, False)
                    else:
                        orc.solve_recursive(systems[i], inp["U0"][i].reshape(-1), nb_iter, True, False)
                    k += nthr
                    done[w] += 1

            t1 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(worker, range(nthr)))
            dt2 = time.perf_counter() - t1
            out["all_cores"] = dict(value=sum(done) * nb_iter / dt2, cores=nthr, sample=f"{sum(done)} instances, {dt2:.1f} s, {nthr} threads")
    except Exception as e:  # the baseline is informative only
        out["all_cores"] = dict(error=str(e))
    out["reference_published_not_measured_here"] = dict(  # the reference's own published timing, for orientation only
        value=745, unit="iterations/s", what="ILQRRecursive, PosOrn 1st order, T=100, one instance, one thread, unknown CPU",
        source="pylqr_planner/Tutorials/POS_ORN_SYS.ipynb:342-348 (time= fields of the stored output; BASELINE.md)")
    return out, res


def _par_map(fn, items, nthr=None):
    """fn over items on the host cores (the oracle's C calls release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    nthr = nthr or min(16, max(1, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(nthr) as ex:
        return list(ex.map(fn, items))


def _rel_stats(rel):
    import numpy as np

    r = np.asarray(rel, float)
    f = r[np.isfinite(r)]
    if not len(f):
        return {"n": int(len(r)), "n_one_sided_nan": int(len(r))}
    return {"n": int(len(r)), "frac_within_1e-4": float(np.mean(r <= 1e-4)), "median": float(np.median(f)), "p90": float(np.quantile(f, 0.9)),
            "max": float(f.max()), "n_one_sided_nan": int(np.sum(~np.isfinite(r)))}


def oracle_self_consistency(cfg, inp, nb_iter, oracle_res, n_max=400):
    """How reproducible is the reference's own map on this workload?  The oracle end to end against its own algebraically neutral
    variants (orc_set_variant: Qxu := Qux^T / pivot-free inverse / fused multiply-adds -- all equal in exact arithmetic) on the same
    sample: the share of instances whose final costs agree within 1e-4.  This is the ceiling any second implementation can be held to
    end to end; the per-iteration proof is what separates a kernel bug from this sensitivity."""
    import numpy as np

    from tests.helpers import oracle_solve_instance, orc, panda_segs

    segs = panda_segs()
    n = min(len(oracle_res), n_max)
    base = np.array([oracle_res[i]["cost"] for i in range(n)])
    out = {"n": n}
    for var, name in ((1, "Qxu=Qux^T"), (2, "pivot-free inverse"), (4, "fused multiply-adds")):
        orc.set_variant(var)
        try:
            c = np.array([r["cost"] for r in _par_map(lambda i: oracle_solve_instance(cfg, inp, i, nb_iter, False, segs), range(n))])
        finally:
            orc.set_variant(0)
        both_nan = ~np.isfinite(c) & ~np.isfinite(base)
        rel = np.where(both_nan, 0.0, np.where(np.isfinite(c) & np.isfinite(base), np.abs(c - base) / np.maximum(np.abs(base), 1e-12), np.inf))
        out[name] = _rel_stats(rel)
    out["note"] = "the oracle against its own neutral variants, same instances, same iteration count: the reference map's end-to-end reproducibility"
    return out


def converged_comparison(ctx, cfg, desc, inp, B, n, nb_cap=100):
    """GPU against oracle where the iteration is allowed to finish: early stop on, at most nb_cap iterations, the first n instances."""
    import numpy as np

    from ilqr_planner_amd import workloads
    from tests.helpers import oracle_solve_instance, panda_segs

    segs = panda_segs()
    p = workloads.load_batch(ctx, desc, inp, B)
    workloads.run_solver(p, cfg, nb_iter=nb_cap, early_stop=True)
    cg, ig, Ug = p.cost()[:n], p.iters()[:n], p.U()[:n]
    p.close()
    res = _par_map(lambda i: oracle_solve_instance(cfg, inp, i, nb_cap, True, segs), range(n))
    co, io = np.array([r["cost"] for r in res]), np.array([r["iters"] for r in res])
    both_nan = ~np.isfinite(cg) & ~np.isfinite(co)
    rel = np.where(both_nan, 0.0, np.where(np.isfinite(cg) & np.isfinite(co), np.abs(cg - co) / np.maximum(np.abs(co), 1e-12), np.inf))
    out = _rel_stats(rel)
    out.update(nb_iter_cap=nb_cap, early_stop=True, frac_stopped_gpu=float(np.mean(ig < nb_cap)), frac_stopped_oracle=float(np.mean(io < nb_cap)),
               frac_same_iteration_count=float(np.mean(ig == io)), mean_iterations_gpu=float(ig.mean()), mean_iterations_oracle=float(io.mean()))
    bad = np.where(rel > 1e-4)[0]
    if len(bad):  # where do the outliers sit?  both sides stopped by the reference's own test at different points = different local solutions
        du = np.array([np.abs(Ug[i] - res[i]["U"]).max() / max(1.0, np.abs(res[i]["U"]).max()) for i in bad])
        out["outliers"] = {"n": int(len(bad)), "both_stopped": int(np.sum((ig[bad] < nb_cap) & (io[bad] < nb_cap))),
                           "gpu_cost_lower": int(np.sum(cg[bad] < co[bad])), "oracle_cost_lower": int(np.sum(co[bad] < cg[bad])),
                           "median_control_distance_rel": float(np.median(du)),
                           "reading": "both_stopped = the reference's own stopping test fired on both sides at different solutions (the same expanding map, "
                                      "run further); lower cost on either side about equally often = no bias"}
    return out


def parity_report(ctx, cfg, desc, inp, B, nb_iter, oracle_res):
    """Per-instance parity proof on the instances the CPU-baseline leg solved with the oracle, for both kernel sets (tests/parity_proof.py):
    every sampled instance is within 1e-4 of the oracle's end-to-end run, or each of its GPU iterations is reproduced by one oracle
    iteration from the GPU's own state (the GPU's decisions follow from the oracle's trial costs, cost at the GPU's step size to 1e-9;
    decisions inside the rounding of the oracle's own comparison and ill-conditioned steps are counted)."""
    import numpy as np

    from ilqr_planner_amd import workloads
    from tests import parity_proof as pp

    out = {}
    n = len(oracle_res)
    prev = os.environ.get("ILQR_HIP_PATH")
    try:
        for path in ("v2", "v1"):
            os.environ["ILQR_HIP_PATH"] = path
            p = workloads.load_batch(ctx, desc, inp, B)
            workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False)
            summ, rel, failures = pp.check_batch(p, cfg, inp, nb_iter, False, workloads.run_solver, lambda i: oracle_res[i], always=(0, 1),
                                                 indices=range(n))
            p.close()
            r = rel[:n][np.isfinite(rel[:n])]
            out[path] = {"n": n, "frac_within_1e-4": summ["frac_within_1e4"], "frac_proven_tie": summ["frac_proven_tie"],
                         "frac_proven_stepwise": summ["frac_proven_stepwise"], "frac_unexplained": summ["frac_unexplained"],
                         "final_cost_rel_err": {"median": float(np.median(r)), "p90": float(np.quantile(r, 0.9)), "max": float(r.max())},
                         "proof": {k: summ[k] for k in ("n_proofs", "n_steps_checked", "n_tie_decisions", "n_steps_ill_conditioned", "worst_ill_ratio")},
                         "unexplained_instances": [f["i"] for f in failures]}
    finally:
        if prev is None:
            os.environ.pop("ILQR_HIP_PATH", None)
        else:
            os.environ["ILQR_HIP_PATH"] = prev
    out["oracle_self_consistency"] = oracle_self_consistency(cfg, inp, nb_iter, oracle_res)
    out["converged"] = converged_comparison(ctx, cfg, desc, inp, B, min(n, 256))
    out["note"] = ("within = final cost within 1e-4 relative of the oracle's own end-to-end solve; proven = outside it, but every GPU iteration is "
                   "one oracle iteration from the GPU's own state: the GPU's trajectory handed over, the GPU's accept / reject decisions following "
                   "from the ORACLE's cost of every step size, the cost at the GPU's step size equal to 1e-9 (stepwise), up to comparisons decided "
                   "inside 1e-9 of the oracle's own cost0 (tie; n_tie_decisions) and steps whose deviation the oracle's own neutral variants "
                   "reproduce or exceed (n_steps_ill_conditioned, worst_ill_ratio = deviation / largest sensitivity over the oracle's neutral variants and one-ulp input perturbations, gate <= 10).  "
                   "oracle_self_consistency = the oracle against its own neutral variants end to end: the share within 1e-4 there is what the "
                   "reference's expanding, discontinuous map allows ANY second implementation on this non-converged workload; "
                   "converged = the same instances run to the reference's own stopping test")
    return out


def parity_report_batch(p, cfg, inp, psi, nb_iter, n):
    """The batch solvers' per-instance proof (tests/parity_proof.py check_batch_solver) on the first n instances of the benchmarked batch."""
    import numpy as np

    from tests import parity_proof as pp

    p.solve_batch_cp(psi, nb_iter, False)
    summ, rel, failures, runs = pp.check_batch_solver(p, cfg, inp, psi, nb_iter, False, lambda q, k, es: q.solve_batch_cp(psi, k, es), always=(0, 1),
                                                      indices=range(n))
    r = rel[:n]
    f = r[np.isfinite(r)]
    return {"n": n, "frac_within_1e-4": summ["frac_within_1e4"], "frac_proven_tie": summ["frac_proven_tie"], "frac_proven_stepwise": summ["frac_proven_stepwise"],
            "frac_unexplained": summ["frac_unexplained"],
            "cost_trace_rel_err": {"median": float(np.median(f)) if len(f) else None, "max": float(f.max()) if len(f) else None,
                                   "n_other_step_size_sequence": int(np.sum(~np.isfinite(r)))},
            "proof": {k: summ[k] for k in ("n_proofs", "n_steps_checked", "n_tie_decisions", "n_steps_ill_conditioned", "worst_ill_ratio")},
            "unexplained_instances": [f_["i"] for f_ in failures],
            "note": "within = the oracle's step-size sequence and every printed cost within 1e-4 of the oracle's end-to-end run; proven = every GPU "
                    "iteration is one oracle iteration from the GPU's own controls (printed cost to 1e-9, decisions from the oracle's trial costs)"}


# ----------------------------------------------------------------------------- main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", help="workload (ilqr_planner_amd.workloads.config); C3 = the metric's configuration")
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: the configuration's per-GPU batch)")
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU-baseline sample and the parity report (profiler runs)")
    ap.add_argument("--gather-trajectories", action="store_true", help="N>1: also gather X / U to rank 0 every step")
    ap.add_argument("--no-split", action="store_true", help="one kernel at a time on one stream (ilqr_ctx_set_split(0)): for rocprofv3 runs whose "
                    "per-kernel durations are to be compared with the roofline block")
    args = ap.parse_args()

    import numpy as np
    import torch

    from ilqr_planner_amd import capi, sharding, workloads

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
    B = int(args.batch or cfg["B"])        # per-GPU shard (weak scaling): C4 = 4096 per GPU, 32768 over 8
    total = world * B
    lo, hi = sharding.shard_range(total, rank, world)
    assert hi - lo == B
    nb_iter = int(args.iters or cfg["nb_iter"])
    ctx = capi.Context(local_rank)
    if args.no_split:
        ctx.set_split(False)
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by the library's launches and torch's collectives
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    desc, inp = workloads.make_batch(ctx, cfg, B=B, seed=cfg["seed"] + 1000 * rank)
    p = workloads.load_batch(ctx, desc, inp, B)  # inputs resident in HBM from here on
    nx, nu = p.dims.n_x, p.dims.n_u
    psi = workloads.psi_of(cfg["psi"], cfg["T"], nu) if cfg["solver"] == "batch_cp" else None
    cost_dev = torch.empty(B, dtype=torch.float64, device="cuda")
    X_dev = torch.empty((B, cfg["T"], nx), dtype=torch.float64, device="cuda") if args.gather_trajectories and world > 1 else None
    U_dev = torch.empty((B, cfg["T"] - 1, nu), dtype=torch.float64, device="cuda") if X_dev is not None else None

    def step():
        if cfg["solver"] == "al":
            p.reset_multipliers()
        workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False, psi=psi)
        p.get_cost_dev(cost_dev.data_ptr())
        if X_dev is not None:
            p.get_X_dev(X_dev.data_ptr())
            p.get_U_dev(U_dev.data_ptr())
        if world > 1:
            gather_step(cost_dev, total, X_dev, U_dev)  # the one collective: converged costs (+ trajectories) over RCCL/xGMI

    for _ in range(args.warmup):
        step()
    ctx.profile(False)
    elapsed = max_over_ranks(timed_region(step, args.steps, torch.cuda.synchronize, dist), dist, "cuda")

    # second pass, per-launch HIP events on: kernel durations for the roofline (not part of the headline)
    ctx.profile_reset()
    ctx.profile(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ctx.profile(False)
    prof = {n: ctx.profile_get(w) for n, w in (("rollout", capi.PROF_ROLLOUT), ("backward", capi.PROF_BACKWARD), ("forward", capi.PROF_FORWARD),
                                               ("apply", capi.PROF_APPLY), ("other", capi.PROF_OTHER))}
    cost = cost_dev.cpu().numpy()
    status = p.status()
    riccati = cfg["solver"] != "batch_cp"
    at = p.trace(nb_iter)[1] if riccati else None

    if rank == 0:
        m = p.m
        v1 = os.environ.get("ILQR_HIP_PATH") == "v1"
        fused = (not v1) and cfg["kind"] in (0, 2) and cfg["nb_deriv"] == 1 and riccati
        alg = algorithmic_bytes(cfg, nx, nu, m, B, fused)
        trials = float(np.mean(1 + np.round(-np.log2(at)))) if at is not None else None
        kern = {}
        for name in ("rollout", "backward", "forward", "apply", "other"):
            ms, n = prof[name]
            if n:
                avg = ms / n
                byts = alg.get(name)
                if byts is not None and name == "forward" and v1:
                    byts = byts * trials  # the generic kernels re-roll the horizon once per trial
                kern[name] = dict(avg_ms=avg, launches=n, total_ms=ms, alg_bytes=byts, gbs=(byts / (avg * 1e-3) / 1e9) if byts else None)
        rated = {k: v for k, v in kern.items() if v["alg_bytes"]}
        dom = max(rated, key=lambda k: rated[k]["total_ms"]) if rated else None
        roof = None
        if dom:
            k = kern[dom]
            traffic, traffic_src = pmc_traffic(args.config, dom) if not v1 else (None, None)
            roof = dict(bound="hbm", kernel=dom, achieved=round(k["gbs"], 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(k["gbs"] / HBM_PEAK_GBS, 5),
                        traffic=traffic, traffic_source=traffic_src, avg_launch_ms=round(k["avg_ms"], 4), alg_bytes_per_launch=int(k["alg_bytes"]),
                        launches_per_step={n_: v["launches"] // 2 for n_, v in kern.items()},
                        other={n_: dict(avg_launch_ms=round(v["avg_ms"], 4), achieved=round(v["gbs"], 2) if v["gbs"] else None) for n_, v in kern.items() if n_ != dom},
                        timing="HIP events on the library's stream, separate pass after the timed region")
            if trials is not None:
                roof["mean_line_search_trials"] = round(trials, 3)
            if fused:  # uniform control weights: K = N / dt is symmetric and the sweep writes its upper triangle (36 of the 56 doubles of a gain record)
                roof["traffic_note"] = ("algorithmic bytes count the full gain record K | d of SURVEY 8(d) (56 doubles per instance-step); the kernels move its packed "
                                        "symmetric form (36 doubles), so the measured traffic lies BELOW the algorithmic bytes")
            # what share of the solve the rated kernels are: a roofline fraction speaks for its own kernel only
            tot_ms = sum(v["total_ms"] for v in kern.values())
            roof["rated_share_of_solve"] = round(sum(v["total_ms"] for v in rated.values()) / tot_ms, 4) if tot_ms else None
            roof["dominant_share_of_solve"] = round(k["total_ms"] / tot_ms, 4) if tot_ms else None
            if not riccati:  # Batch-CP: the horizon is walked twice per solve (rated); the iterations work on keypoint-sized data
                unr = {n_: round(v["total_ms"] / tot_ms, 4) for n_, v in kern.items() if n_ not in rated}
                roof["unrated"] = {"share_of_solve": unr, "bound": "latency",
                                   "note": "backward = linearize + solve of the 14 x 14 normal equations, forward = the two line-search passes: per-iteration launches on "
                                           "keypoint-sized data (7 x 14 W, 14 x 14 H), each the chain of ONE FK / cost evaluation or ONE solve at one to two waves per SIMD; "
                                           "SQ_WAIT_ANY / SQ_WAVE_CYCLES = 52-77 % (profiles/r02_sq_counters.txt): no memory or matrix-core roofline applies, "
                                           "the figure of the rated kernel is NOT the configuration's"}
        kinds = ("PosOrn", "PosOrnTime", "JointSpace", "JointSpaceTime")
        out = {
            "metric": METRIC,
            "value": total * nb_iter * args.steps / elapsed,
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
            "config": {"workload": f"{args.config}: {kinds[cfg['kind']]} nb_deriv={cfg['nb_deriv']} "
                                   f"{'7-DoF Panda chain' if cfg['kind'] < 2 else str(cfg.get('dof', 7)) + ' joints'}, "
                                   f"T={cfg['T']}, batch {B}/GPU x {world} GPU = {total}, solver={cfg['solver']}, {nb_iter} iterations/solve, "
                                   f"{'line search on, ' if riccati else ''}early stop off",
                       "global_batch": total, "horizon": cfg["T"], "iterations_per_step": nb_iter, "parallelism": f"instances sharded x{world}",
                       "exchange": "all-gather of costs" + (" + gather of X, U to rank 0" if X_dev is not None else "") if world > 1 else "none"},
            "batch_sweeps_per_s": nb_iter * args.steps / elapsed,
            "nonfinite_frac": float(np.mean((status & 1) != 0)),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline and the parity report belong to the N = 1 line only
            cb, ores = cpu_baseline(cfg, inp, nb_iter, psi)
            out["cpu_baseline"] = cb
            if riccati:
                out["parity"] = parity_report(ctx, cfg, desc, inp, B, nb_iter, ores)
            else:
                out["parity"] = parity_report_batch(p, cfg, inp, psi, nb_iter, len(ores))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    p.close()
    ctx.close()


if __name__ == "__main__":
    main()
