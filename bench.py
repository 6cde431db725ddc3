#!/usr/bin/env python3
"""bench.py -- warm-start LAP pipeline throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one resident batch:
    row features (HIP) -> OneGNN forward (PyTorch-ROCm + HIP aggregation) -> v = min(C - u) (HIP)
    -> batched lapjv_seeded (HIP)  [-> RCCL gather of the assignments when N > 1]
Workload = BASELINE.json configs[2] (K3): batch=32 per GPU, n=2048, mixed families
(8 each uniform / sparse / metric / clustered), OneGNN hidden=192 layers=4 random init,
fp64 costs already in HBM when the timed region starts.  Weak scaling: every rank owns its
own batch of 32; value = instances all ranks solved / max-over-ranks wall time.

The steps are software-pipelined over two HIP streams (default; --no-overlap turns it off): the
dense sweeps + OneGNN of step k+1 run beside the per-instance solver of step k, which holds one
CU per instance (32 of 256).  Every timed step still enqueues one full set of dense stages and
one solve; the work inside the timed region is exactly K of each.

Usage:  python bench.py [--gpus N --steps K --warmup W]
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG = ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# torch is imported inside run_rank(): the --gpus N launcher below must start its rank processes
# before this process has made any GPU call (and it never makes one itself)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_CUS = 256


def serial_elems(stats_row, n):
    """E of SURVEY.md 8(d) from the kernel's own counters (identical to the oracle's)."""
    return int(stats_row[8] + stats_row[7] + stats_row[9] + n * stats_row[10] + n * stats_row[11])


def solver_source_sha():
    """Identity of the solver build a PMC record belongs to: sha256 of the kernel sources."""
    import hashlib
    h = hashlib.sha256()
    for f in ("jv_solver.hip", "coop_ssp.hip", "device_utils.hpp"):
        h.update((PKG / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def recorded_traffic(kernel_prefix="jv_instance_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc
    cannot run inside the timed process; see profiles/*_pmc_traffic.txt).  Only a record taken from
    THIS build of the solver counts (tools/summarize_pmc.py stores the source hash): a record of an
    older kernel is refused rather than reported beside a newer kernel's time.  None if absent."""
    best = None
    sha = solver_source_sha()
    for f in sorted((ROOT / "profiles").glob("r*_pmc_traffic.json")):
        try:
            meta = json.loads(f.read_text())
        except Exception:
            continue
        if meta.get("solver_source_sha") != sha:
            continue
        for k, d in meta.get("kernels", {}).items():
            if k.startswith(kernel_prefix):
                best = d.get("hbm_bytes_per_launch_corrected")
    return best


def cpu_pipeline_once(C_b, sd, u_given=None):
    """Oracle pipeline for one instance (NumPy features + torch-CPU OneGNN + C restatement of
    lapjv_seeded) -- the checker, used here as the CPU baseline ("port")."""
    from oracle import features_np, jv, one_gnn_ref
    if u_given is not None:  # K2: features + min-trick + seeded solve with the given u
        features_np.compute_row_features(C_b)
        u, v = u_given, features_np.min_trick(C_b, u_given)
    else:
        u, v = one_gnn_ref.predict(sd, C_b)
    return jv.seeded_raw(C_b, u, v)


def cpu_baseline(C_host, sd, sample_idx, u_given=None):
    """One thread (the reference's methodology pins 1 thread, scripts/gnn_benchmark.py:26-31)."""
    import torch
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    outs = [cpu_pipeline_once(C_host[b], sd, None if u_given is None else u_given[b]) for b in sample_idx]
    dt = time.perf_counter() - t0
    return len(sample_idx) / dt, dt, outs


def _node_worker(args):
    """One process of the whole-node CPU leg: regenerates the batch, times its share."""
    B, n, fams, seed, hidden, layers, idx = args
    for p in (str(ROOT), str(PKG)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import torch as _torch
    _torch.set_num_threads(1)
    from gnn import OneGNN
    from solvers.generators import mixed_instances
    mats, _ = mixed_instances(B, n, idx, families=fams, seed=seed)  # only this worker's instances
    _torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in OneGNN(21, hidden=hidden, layers=layers).state_dict().items()}
    t0 = time.perf_counter()
    for Cb in mats:
        cpu_pipeline_once(Cb, sd)
    return time.perf_counter() - t0, len(idx)


def cpu_baseline_node(B, n, fams, seed, hidden, layers, per_proc=2):
    """Whole-node CPU leg: one single-threaded process per host core, each
    running the oracle pipeline on `per_proc` instances of the same batch; aggregate
    instances/s = all instances / slowest process."""
    import multiprocessing as mp
    # the cores this process may actually use (a one-GPU share of the host, not every core of it)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))
    jobs = [(B, n, fams, seed, hidden, layers, [(c * per_proc + q) % B for q in range(per_proc)])
            for c in range(cores)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_node_worker, jobs)
    wall = time.perf_counter() - t0
    slowest = max(r[0] for r in res)
    total = sum(r[1] for r in res)
    return total / slowest, cores, total, slowest, wall


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start N rank processes (one per GPU)
    through torch.distributed.run and forward rank 0's JSON line.  This parent process never touches
    the GPU (no torch.cuda call, liblapwarm_hip.so not loaded) and never exec()s: the ranks are
    children, their exit code is ours, and a line whose n_gpus differs from N is an error."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for raw in proc.stdout:
        sys.stdout.write(raw)
        sys.stdout.flush()
        t = raw.strip()
        if t.startswith("{") and '"n_gpus"' in t:
            try:
                line = json.loads(t)
            except ValueError:
                pass
    rc = proc.wait()
    if rc != 0:
        print("bench.py: a rank process failed (exit code %d)" % rc, file=sys.stderr)
        return rc
    if line is None or line.get("n_gpus") != n_ranks:
        print("bench.py: expected one JSON line with n_gpus == %d, got %r" % (
            n_ranks, None if line is None else line.get("n_gpus")), file=sys.stderr)
        return 3
    return 0


def standin_pipeline(C_host):
    """--standin: a CPU stand-in for the device pipeline (row-wise argmin, NOT a solve) so that the
    launcher / rendezvous / gather / JSON flow can be rehearsed where there is no GPU
    (tests/test_bench_launcher.py).  Its JSON line says so and can never be read as a measurement."""
    import torch
    x = torch.from_numpy(np.argmin(C_host, axis=2).astype(np.int64))

    def solve_local():
        return {"x": x.clone()}
    return solve_local


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=32, help="instances per GPU")
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--hidden", type=int, default=192)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--threads-hint", type=int, default=0)
    ap.add_argument("--families", type=str, default="uniform,sparse,metric,clustered")
    ap.add_argument("--cpu-sample", type=int, default=32,
                    help="instances of the same batch timed through the CPU oracle pipeline on rank 0 at N=1 "
                         "(0 = skip); 32 x n=2048 is ~10 s of single-core work")
    ap.add_argument("--no-node-baseline", action="store_true",
                    help="skip the one-process-per-core CPU leg (rank 0, N=1, K3 only)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the stages of a step back to back on one stream (no two-stream pipelining)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="additionally report (separate key, never `value`) the throughput with this many "
                         "independent K3 batches in flight on separate streams")
    ap.add_argument("--backend", type=str, default="nccl",
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to rehearse the "
                         "multi-rank flow on a single GPU)")
    ap.add_argument("--config", type=str, default="K3", choices=["K2", "K3", "K4", "K5"],
                    help="K3 (default) is the configuration the metric is quoted on; K2 / K4 / K5 are the other "
                         "single-GPU-sized BASELINE configs (K4 = one GPU's 32-instance slice of batch 256; "
                         "K5 = n=16384 large-n stress, batch 1)")
    ap.add_argument("--standin", action="store_true",
                    help="launcher rehearsal without a GPU: CPU stand-in instead of the device pipeline, gloo "
                         "backend, tiny sizes; the JSON line is marked as a rehearsal, not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # no torchrun around us: become the launcher (before anything below can touch a GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if world_env is not None and int(world_env) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: the flag and the launcher disagree"
                         % (args.gpus, world_env))
    run_rank(args)


def run_rank(args):
    import torch
    if args.standin:
        return run_rank_standin(args)
    if args.config == "K2":    # batch=64 n=512 uniform, optimal-dual seeds instead of the GNN
        args.batch, args.n, args.families = 64, 512, "uniform"
    elif args.config == "K4":  # batch=256 n=4096 over 8 GPUs -> 32 per GPU
        args.batch, args.n, args.families = 32, 4096, "uniform"
        args.cpu_sample = min(args.cpu_sample, 8)  # ~0.7 s per n=4096 instance on one core
    elif args.config == "K5":  # n=16384, 2 GiB of fp64 costs per instance (--n / --batch: the other large-n sizes)
        args.batch = args.batch if args.batch != 32 else 1
        args.n = args.n if args.n != 2048 else 16384
        args.families = "uniform"
        args.cpu_sample = min(args.cpu_sample, 1)
    if args.steps is None:
        args.steps = 1 if args.config == "K5" else 10
    if args.warmup is None:
        args.warmup = 1 if args.config == "K5" else 3

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback for the HIP path)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit("one GPU per rank is required with the nccl (RCCL) backend")
    local_dev = local_rank % max(1, ndev)  # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from gnn import OneGNN, WarmStartPipeline
    from gnn.bench_core import run_sharded
    from lap import _hip
    from solvers.generators import mixed_batch

    B, n = args.batch, args.n
    fams = tuple(args.families.split(","))
    if args.config == "K3":
        C_host, names = mixed_batch(B, n, families=fams, seed=1234 + rank)
    else:  # RandomState(42+i).uniform, as scripts/gnn_large_scale_benchmark.py:243-251
        C_host = np.stack([np.random.RandomState(42 + rank * B + i).uniform(0, 1, (n, n)) for i in range(B)])
        names = ["uniform"] * B
    torch.manual_seed(0)
    model = OneGNN(21, hidden=args.hidden, layers=args.layers).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    pipe = WarmStartPipeline(model, dev, threads_hint=args.threads_hint)
    C = torch.from_numpy(C_host).to(dev)
    lib = _hip.load()
    lib.lapwarm_profile_enable(1)

    u_k2 = None
    if args.config == "K2":
        _, u_k2, _, _ = pipe.optimal_duals_batch(C)  # "oracle u": not part of the timed step
    overlap = not args.no_overlap
    solver_ms = []
    k2_predict = None
    if u_k2 is not None:
        from gnn.features import min_trick_device, row_features_device

        def k2_predict(Cb):                           # K2's dense stages: features + col-min, given duals
            row_features_device(Cb)
            return u_k2, min_trick_device(Cb, u_k2)

    def solve_local():
        if u_k2 is not None and not overlap:
            u, v = k2_predict(C)                      # K2 times features + col-min + seeded JV
            x, y, ret, stats = pipe.seeded_batch(C, u, v)
            return {"x": x, "y": y, "ret": ret, "stats": stats, "u": u_k2, "v": v}
        if overlap:
            # solve what was submitted during the previous step; submit this step's dense stages
            out = pipe.pipeline_step(C_next=C)
            out["done"].synchronize()  # the gather below (and the caller) read x
            return out
        return pipe.solve_batch(C)

    def after_step(out):
        # reading the bracket waits for this step's solver kernel only
        solver_ms.append(lib.lapwarm_profile_last_solver_ms())

    def device_sync():
        torch.cuda.synchronize(dev)

    if overlap:
        pipe.pipeline_submit(C, k2_predict)  # primes the pipeline (untimed, like the warm-up steps)
    out, elapsed = run_sharded(solve_local, args.steps, args.warmup, distributed=distributed,
                               gather_on_host=(args.backend != "nccl"), device_sync=device_sync,
                               after_step=after_step)
    if overlap:
        pipe.pipeline_drain()
    solver_ms = solver_ms[-args.steps:]

    stats = out["stats"].cpu().numpy()
    ret = out["ret"].cpu().numpy()
    if rank == 0:
        total_instances = B * world * args.steps
        value = total_instances / elapsed
        E = sum(serial_elems(stats[b], n) for b in range(B))
        solver_avg_ms = float(np.mean(solver_ms))
        alg_bytes = 8.0 * E
        achieved = alg_bytes / (solver_avg_ms * 1e-3) / 1e9
        branches = {int(k): int(c) for k, c in zip(*np.unique(stats[:, 0], return_counts=True))}
        from lap import _hip as _h
        members = int(_h.load().lapwarm_coop_members(n))  # > 0: cooperative shortest-path kernel (n >= 4428)
        # one workgroup = one instance = one CU, or `members` single-wave workgroups per instance
        cus_busy = min(B * max(1, members), N_CUS)
        helpers = bool(_h.load().lapwarm_solver_uses_helpers(n))
        cus_kernel = min(2 * B, N_CUS) if helpers else cus_busy  # + one (mostly idle) helper CU per instance
        # (the grid pads the batch to a multiple of 8 so that helper and solver share an XCD; padding exits at once)
        line = {
            "metric": "LAP instances/sec (whole node), n=%d warm-start pipeline" % n,
            "value": round(value, 3),
            "unit": "instances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (args.config + ": batch=%d/GPU n=%d families (%s), " % (B, n, "/".join(fams))) + (
                    "optimal-dual u (from cold JV, untimed), row features + min-trick + lapjv_seeded, "
                    "costs resident in HBM" if args.config == "K2" else
                    "OneGNN H=%d L=%d random init, features+OneGNN+min-trick+lapjv_seeded end-to-end, "
                    "costs resident in HBM" % (args.hidden, args.layers)),
                "global_batch": B * world,
                "n": n,
                "parallelism": "batch-sharded x%d, one %s gather of assignments" % (
                    world, "RCCL" if args.backend == "nccl" else args.backend),
                "stage_overlap": (("two HIP streams: row features + min-trick of step k+1 beside the solver of step k"
                                   if u_k2 is not None else
                                   "two HIP streams: dense sweeps + OneGNN of step k+1 beside the solver of step k")
                                  if overlap else "none (stages back to back on one stream)"),
                "solver_threads_hint": args.threads_hint,
                "solver_helper_workgroups": helpers,
                "solver_coop_members_per_instance": members,
                "branches": branches,
                "ret_nonzero": int((ret != 0).sum()),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("jv_instance_kernel (greedy, micro-ARR) + coop_ssp_kernel (shortest paths, %d single-wave "
                           "members per instance) + jv_instance_kernel (outputs): the three launches of one solve" % members
                           if members else
                           "jv_instance_kernel (per-instance seeded JV: greedy, micro-ARR, SSP / cold-JV fallback)"),
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6),
                "traffic": recorded_traffic() if args.config == "K3" and (B, n) == (32, 2048) else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(solver_avg_ms, 3),
                "cus_occupied": cus_kernel,
                "cus_solving": cus_busy,
                "cus_total": N_CUS,
                "per_cu_GBps": round(achieved / cus_busy, 3),
                "note": "latency-bound chain of dependent row scans on ONE CU per instance (cus_solving); with "
                        "solver_helper_workgroups a second, mostly idle workgroup per instance pulls announced "
                        "rows into the shared L2 (cus_occupied of cus_total CUs); a CU pulls ~25 GB/s from HBM at "
                        "best, tools/micro/mlp_bench.hip; achieved = 8*E/launch time with E counted by the kernel; "
                        "traffic = HBM bytes per launch from separate rocprofv3 --pmc passes "
                        "(profiles/*_pmc_traffic.txt, 2*FETCH_SIZE+WRITE_SIZE)",
            },
        }
        if world == 1 and args.cpu_sample > 0:
            k = min(args.cpu_sample, B)
            idx = [int(round(i * (B - 1) / max(1, k - 1))) for i in range(k)] if k > 1 else [0]
            idx = sorted(set(idx))
            cpu_val, cpu_dt, cpu_out = cpu_baseline(C_host, sd, idx,
                                                    u_k2.cpu().numpy() if u_k2 is not None else None)
            # parity spot-check outside the timed region: same (u, v) -> same assignment
            from oracle import jv
            u = out["u"].cpu().numpy().astype(np.float64)
            v = out["v"].cpu().numpy()
            x = out["x"].cpu().numpy()
            exact = 0
            for b in idx:
                r, xo, _, _ = jv.seeded_raw(C_host[b], u[b], v[b])
                exact += int(r == ret[b] and (r != 0 or np.array_equal(xo, x[b])))
            line["cpu_baseline"] = {
                "value": round(cpu_val, 4),
                "unit": "instances/s",
                "cores": 1,
                "host_cores": os.cpu_count(),
                "usable_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                "kind": "port",
                "what": "this repository's CPU restatement of the reference pipeline (oracle/: NumPy features, "
                        "torch-CPU OneGNN forward, C restatement of lapjv_seeded verified bit-exact against the "
                        "reference build) -- the reference itself cannot travel to the GPU box",
                "sample": "%d of the %d bench instances (%s), oracle pipeline, 1 thread, %.1f s"
                          % (len(idx), B, ",".join(names[b] for b in idx), cpu_dt),
            }
            line["parity_spot_check"] = {"instances": len(idx), "bit_exact": exact}
            # the port against the reference's own C++ (oracle/_ref: built where /root/reference exists,
            # travels as a prebuilt file): solver leg only, same (u, v), a few instances
            try:
                from oracle import ref as _ref
                if _ref.LIB_PATH.exists():
                    tp = tr = 0.0
                    same = 0
                    sub = idx[:4]
                    for b in sub:
                        t0 = time.perf_counter()
                        rp, xp, _, _ = jv.seeded_raw(C_host[b], u[b], v[b])
                        t1 = time.perf_counter()
                        rr, xr, _ = _ref.seeded_raw(C_host[b], u[b], v[b])
                        t2 = time.perf_counter()
                        tp += t1 - t0
                        tr += t2 - t1
                        same += int(rp == rr and np.array_equal(xp, xr))
                    line["cpu_baseline"]["port_over_reference"] = round(tp / max(tr, 1e-9), 3)
                    line["cpu_baseline"]["port_over_reference_note"] = (
                        "solver leg (lapjv_seeded) only: oracle/jv_oracle.c (with its element counters) against the "
                        "reference's lapjv.cpp + lapjv_seeded.cpp compiled unmodified (oracle/_ref), %d instances, "
                        "%d identical assignments" % (len(sub), same))
            except Exception as e:  # the checker must never break the benchmark line
                line["cpu_baseline"]["port_over_reference_note"] = "not measured: %s" % e
            if args.config == "K3" and not args.no_node_baseline:
                nv, cores, total, slowest, wall = cpu_baseline_node(B, n, fams, 1234 + rank, args.hidden, args.layers)
                line["cpu_baseline_node"] = {
                    "value": round(nv, 3),
                    "unit": "instances/s",
                    "cores": cores,
                    "kind": "port",
                    "sample": "%d single-threaded processes (one per host core) x %d instances of the same batch, "
                              "oracle pipeline; slowest process %.1f s, wall incl. start-up %.1f s"
                              % (cores, total // max(1, cores), slowest, wall),
                }
        if world == 1 and args.inflight > 1 and u_k2 is None:
            # several independent batches in flight (a service that keeps the other 224 CUs busy);
            # clearly NOT the configured one-batch-per-step metric
            pipes = [WarmStartPipeline(model, dev, threads_hint=args.threads_hint) for _ in range(args.inflight)]
            streams = [torch.cuda.Stream(dev) for _ in range(args.inflight)]
            Cs = [C] + [C.clone() for _ in range(args.inflight - 1)]
            lib.lapwarm_profile_enable(0)
            for rep in range(2):  # warm-up pass, timed pass
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    for pp, ss, cc in zip(pipes, streams, Cs):
                        with torch.cuda.stream(ss):
                            pp.solve_batch(cc, want_stats=False)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
            line["throughput_batches_in_flight"] = {
                "batches_in_flight": args.inflight,
                "value": round(args.inflight * B * args.steps / dt, 3),
                "unit": "instances/s",
                "note": "NOT the metric: %d independent K3 batches on separate HIP streams per step" % args.inflight,
            }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


def run_rank_standin(args):
    """The multi-rank flow of run_rank() with the CPU stand-in: same run_sharded(), same gather, same
    JSON keys; backend gloo, no GPU, no library."""
    import torch
    import torch.distributed as dist
    from gnn.bench_core import run_sharded
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if os.environ.get("LAPWARM_STANDIN_FAIL_RANK") == str(rank):  # test hook: a rank that dies
        raise SystemExit(7)
    B, n = min(args.batch, 4), min(args.n, 32)
    C_host = np.stack([np.random.RandomState(42 + rank * B + i).uniform(0, 1, (n, n)) for i in range(B)])
    steps = args.steps if args.steps is not None else 2
    warmup = args.warmup if args.warmup is not None else 1
    out, elapsed = run_sharded(standin_pipeline(C_host), steps, warmup, distributed=distributed,
                               gather_on_host=True)
    if rank == 0:
        gathered = out["x_all"] if distributed else out["x"]
        assert tuple(gathered.shape) == (B * world, n)
        print(json.dumps({
            "metric": "REHEARSAL (bench.py --standin): launcher + gloo gather only, not a measurement",
            "value": None, "unit": "instances/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "standin",
            "config": {"workload": "stand-in (row argmin on CPU), batch=%d/rank n=%d" % (B, n),
                       "global_batch": B * world, "parallelism": "batch-sharded x%d, one gloo gather" % world},
        }), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
