"""Child process of tests/test_gpu_rccl_world1.py: the N > 1 leg of bench.py at world size 1 on the real device -- RCCL process group
on the GPU, the library's launches and the collectives on ONE shared non-null torch stream (exactly bench.py's set-up), no host
synchronisation between a solve and the gather that reads its results.  Prints one JSON line."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))  # before anything else touches the GPU

    import bench
    from ilqr_planner_amd import capi, workloads

    ctx = capi.Context(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    cfg = workloads.config("C4")
    B, T = 96, cfg["T"]
    desc, inp = workloads.make_batch(ctx, cfg, B=B)
    p = workloads.load_batch(ctx, desc, inp, B)
    nx, nu = p.dims.n_x, p.dims.n_u
    cost_dev = torch.empty(B, dtype=torch.float64, device="cuda")
    X_dev = torch.empty((B, T, nx), dtype=torch.float64, device="cuda")
    U_dev = torch.empty((B, T - 1, nu), dtype=torch.float64, device="cuda")
    got = []
    for nb_iter in (2, 5, 3):  # back to back: a gather that ran ahead of (or behind) its solve would return another solve's numbers
        workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False)
        p.get_cost_dev(cost_dev.data_ptr())
        p.get_X_dev(X_dev.data_ptr())
        p.get_U_dev(U_dev.data_ptr())
        out = bench.gather_step(cost_dev, B, X_dev, U_dev)
        got.append({k: v.clone() for k, v in out.items()})  # clones are ordered on the same stream
    t = bench.max_over_ranks(0.25, dist, "cuda")
    bench.fence(torch.cuda.synchronize, dist)
    ok, worst = True, 0.0
    for nb_iter, g in zip((2, 5, 3), got):
        workloads.run_solver(p, cfg, nb_iter=nb_iter, early_stop=False)
        c, X, U = p.cost(), p.X(), p.U()  # host getters synchronise
        for name, a, b in (("cost", g["cost"].cpu().numpy(), c), ("X", g["X"].cpu().numpy(), X), ("U", g["U"].cpu().numpy(), U)):
            same = np.array_equal(a, b, equal_nan=True)
            ok = ok and same
            if not same:
                worst = max(worst, float(np.nanmax(np.abs(a - b))))
    dist.barrier()
    dist.destroy_process_group()
    p.close()
    ctx.close()
    print(json.dumps(dict(ok=bool(ok), worst=worst, max_over_ranks=t, backend="nccl", world=1, finite_frac=float(np.mean(np.isfinite(c))))))


if __name__ == "__main__":
    main()
