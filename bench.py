#!/usr/bin/env python3
"""Headline benchmark: cokriging grid-points/s at n_obs = 20k bivariate (BASELINE.json configs[2]).

One step = one pass of the hot path on the GPU(s): assemble Sigma (K1), blocked FP64-MFMA
Cholesky (K3), assemble c0 (K2), forward substitution + fused prediction / variance reductions
(K4) for the 8 833-point 0.5-degree CONUS grid.  Coordinates and values are resident in HBM
before the timed region; only the 141 KB of prediction coordinates and the 141 KB of results
cross PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n-obs 20000] [--params A|B]

N > 1 is launched by the driver under torch.distributed.run (one rank per GPU, RCCL): Sigma is
partitioned 1-D block-column-cyclic, panels are broadcast over xGMI at each Cholesky step, the
prediction points are sharded by rank ("scaling": "strong" -- total work fixed).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X FP64 matrix peak (AMD datasheet; the local guide lists no FP64 figure)
PEAK_HBM_GBS = 8000.0


def trailing_update_flops(N, NB=512):
    """Algorithmic flops of all Cholesky trailing updates (lower triangle only): for panel K the
    trailing matrix has T = N - (K+1) NB rows; T (T + 1) / 2 entries x 2 NB flops."""
    tot = 0.0
    K = 0
    while True:
        T = N - (K + 1) * NB
        if T <= 0:
            break
        tot += T * (T + 1) / 2 * 2 * NB
        K += 1
    return tot


def aux_update_flops(N, m, NB=512):
    tot = 0.0
    K = 0
    while True:
        T = N - (K + 1) * NB
        if T <= 0:
            break
        tot += 2.0 * (m + 1) * T * NB
        K += 1
    return tot


def cpu_baseline(N_full, m_full):
    """The reference path on this host's cores, stage by stage, on a bounded sample of the same workload
    (SURVEY.md section 8d / BASELINE.md section 3): same lattice, same model, same 0.5-degree grid resolution,
    n = 3 000 per process (N = 6 000) and the first 4 000 grid points.  The oracle's functions are the
    reference's lines restated (numpy, scipy.special.kv, sklearn haversine, LAPACK through scipy), timed in the
    order Predictor.__call__ runs them (src/joint_prediction.py:49-78).  `value` counts the stages that produce
    the output (Sigma, c0, cho_factor, cho_solve, the reductions as the reference does them: a full m x N x m
    product for a diagonal); the two diagnostic stages (m x m _pred_cov, _verify_model) are listed separately.
    `extrapolation` scales each stage by its operation count to the full size -- labelled as such."""
    import numpy as np
    from scipy.linalg import cho_factor, cho_solve
    from oracle import cokrige_oracle as orc
    from sif_xco2_cokriging_amd import synth
    n, m = 3000, 4000
    N = 2 * n
    pb = synth.conus_problem(n, seed=20003)
    p = orc.Params.from_flat(pb["params"])
    pc, metric = pb["pcoords"][:m], pb["metric"]
    st = {}

    def timed(key, fn):
        t0 = time.perf_counter()
        out = fn()
        st[key] = time.perf_counter() - t0
        return out

    S = timed("joint_cov_s", lambda: orc.joint_cov(p, pb["coords"], metric))                      # :124-153
    c0 = timed("pred_cross_cov_s", lambda: orc.pred_cross_cov(p, pb["coords"], pc, 0, metric))    # :104-122
    z = np.hstack(pb["values"])
    cf = timed("cho_factor_s", lambda: cho_factor(S.copy(), lower=True, overwrite_a=True, check_finite=False))   # :69
    W = timed("cho_solve_s", lambda: cho_solve(cf, c0.copy(), overwrite_b=True, check_finite=False).T)          # :68-73
    C_pp = timed("pred_cov_diagnostic_s", lambda: orc.pred_cov(p, pc, 0, metric))                 # :94-102 (m x m)

    def reduce_():
        var = np.diagonal(C_pp - np.matmul(W, c0))                                                # :74
        with np.errstate(invalid="ignore"):
            return np.matmul(W, z), np.nan_to_num(np.sqrt(var))                                   # :77-78
    timed("reductions_s", reduce_)

    def verify_():
        try:
            cho_factor(np.vstack([np.hstack([C_pp, c0.T]), np.hstack([c0, S])]), overwrite_a=True)   # :260-274
        except Exception:
            pass
    timed("verify_model_diagnostic_s", verify_)
    core = ("joint_cov_s", "pred_cross_cov_s", "cho_factor_s", "cho_solve_s", "reductions_s")
    dt = sum(st[k] for k in core)
    rN, rm = N_full / N, m_full / m
    scale = {"joint_cov_s": rN ** 2, "pred_cross_cov_s": rN * rm, "cho_factor_s": rN ** 3, "cho_solve_s": rN ** 2 * rm,
             "reductions_s": rN * rm ** 2, "pred_cov_diagnostic_s": rm ** 2, "verify_model_diagnostic_s": (
                 (N_full + m_full) / (N + m)) ** 3}
    ext = {k: st[k] * scale[k] for k in st}
    ext_core = sum(ext[k] for k in core)
    # `value` is the rate on the SAMPLE (300 x less work than the configuration the GPU ran); `value_at_config` is the
    # like-for-like figure at the bench's own size -- an extrapolation by operation count (labelled below)
    return {"value": m / dt, "unit": "grid-points/s", "value_at_config": m_full / ext_core,
            "value_at_config_is": f"EXTRAPOLATED from the sample by operation count to N={N_full}, m={m_full} (see extrapolation)",
            "cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle stages of joint Predictor.__call__, n_obs={n}/process (N={N}), m={m} grid points, "
                      f"{dt:.1f} s for the output-producing stages (+ {st['pred_cov_diagnostic_s'] + st['verify_model_diagnostic_s']:.1f} s "
                      f"diagnostics); numpy/scipy, BLAS threads = all {os.cpu_count()} host cores, covariance assembly single-threaded as in the reference",
            "stages": {k: round(v, 4) for k, v in st.items()},
            "extrapolation": {"label": "EXTRAPOLATED from the sample by operation count, not measured",
                              "to": f"N={N_full}, m={m_full}", "stages_s": {k: round(v, 2) for k, v in ext.items()},
                              "output_stages_s": round(ext_core, 1), "grid_points_per_s": m_full / ext_core,
                              "with_diagnostics_grid_points_per_s": m_full / sum(ext.values())}}


def measured_traffic():
    """(bytes per launch, source) -- HBM-side bytes per launch of the dominant kernel from the newest committed PMC passes
    (profiles/rNN_traffic.json; scripts/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
    runs of this very command, KB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane
    streams).  A committed measurement of the same workload, NOT a live counter of this run -- `traffic_source` in the
    bench line says which file it came from; (None, None) if absent."""
    for name in ("r04_traffic.json",):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            try:
                return json.load(open(p)).get("k_tall_group_bytes_per_launch"), f"profiles/{name} (committed rocprofv3 --pmc pass of this command, not a live counter of this run)"
            except Exception:
                return None, None
    return None, None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a CHILD process and
    pass its exit code on.  This process never touches the GPU (device_count() does not initialise it)."""
    import socket
    import subprocess
    import torch
    backend = os.environ.get("CK_DIST_BACKEND", "nccl")
    nd = torch.cuda.device_count()
    if backend == "nccl" and nd < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {nd} GPU(s) visible (RCCL needs one per rank; "
                         f"CK_DIST_BACKEND=gloo rehearses the multi-rank path on fewer)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-obs", type=int, default=20000)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2],
                    help="index into BASELINE.json configs: 2 (default, the metric's config: CONUS lattice, haversine) "
                         "or 1 (unit square, Euclidean, n_obs = 5000, 100 x 100 grid -- a side measurement)")
    ap.add_argument("--params", default="A")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the secondary block: one pass of BASELINE configs[3] (n_obs = 50 000 per process, N = 100 000) on the same ranks")
    ap.add_argument("--sweeps", default="overlapped", choices=["overlapped", "sequential"],
                    help="single GPU: the timed step's factorisation and substitution as two overlapped sweeps (ck_factor_predict, "
                         "the product path, default) or one after the other (ck_factor, ck_predict: what the profiling scripts "
                         "run, so that a kernel's launches do not share the chip with the other sweep's)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    from sif_xco2_cokriging_amd import native, synth
    # one rank per GPU; CK_DIST_BACKEND=gloo + fewer GPUs than ranks is the single-GPU rehearsal of
    # the multi-rank path (RCCL refuses two ranks on one device)
    backend = os.environ.get("CK_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPU(s) visible: RCCL needs one GPU per rank")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    params = synth.SET_A if args.params == "A" else synth.SET_B
    if args.config == 1:
        if args.n_obs == 20000:
            args.n_obs = 5000
        pb = synth.unit_square_problem(args.n_obs, grid_side=100)
    else:
        pb = synth.conus_problem(args.n_obs, seed=20003, params=params)
    n = args.n_obs
    N = 2 * n
    m = len(pb["pcoords"])

    h = native.Handle(local_rank)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    pv = pb["params"]
    h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
    h.set_metric(pb["metric"])
    for k in range(2):
        h.set_data(k, pb["coords"][k], pb["values"][k])

    for kv in filter(None, os.environ.get("CK_BENCH_OPTIONS", "").split(",")):   # experiments: "name=value,..." -> ck_set_option
        name, value = kv.split("=")
        h.set_option(name.strip(), int(value))
    if world == 1:
        h.set_option("time_gemm", 1)

        def step():
            # the product path (joint_prediction.Predictor.__call__ on a new model): assemble Sigma, assemble c0, then the
            # factorisation and the forward substitution as two OVERLAPPED sweeps (ck_factor_predict), reduce
            h.assemble_joint()
            info, pred, err = h.factor_predict(0, pb["pcoords"])
            if info != 0:
                raise RuntimeError(f"Sigma not positive definite at minor {info}")
            return pred, err

        def step_sequential():
            # the same work with the sweeps one after the other (ck_factor, ck_predict): what the per-kernel figures of the
            # roofline block are measured on -- overlapped, a launch's duration includes the other sweep's share of the chip
            h.assemble_joint()
            info = h.factor()
            if info != 0:
                raise RuntimeError(f"Sigma not positive definite at minor {info}")
            return h.predict(0, pb["pcoords"])
        if args.sweeps == "sequential":
            step = step_sequential
            h.set_option("solve_la", 0)   # "sequential" means it: every launch of either sweep with the chip to itself
    else:
        from sif_xco2_cokriging_amd import distributed
        # CK_PANEL_EXCHANGE = broadcast | sag | p2p | auto (default: one mid-size panel through each at warm-up, the
        # fastest is kept); CK_PANEL_GROUP = 1 | 2 | 3 ... | auto (default: one pass of the per-panel schedule against one
        # of the grouped one at warm-up) -- nothing about the multi-GPU form is chosen untimed
        pg = os.environ.get("CK_PANEL_GROUP", "auto")
        runner = distributed.DistributedJoint(h, rank, world, dist_module=dist, device=torch.device("cuda", local_rank),
                                              exchange=os.environ.get("CK_PANEL_EXCHANGE", "auto"),
                                              panel_group=pg if pg == "auto" else int(pg),
                                              # CK_CHAIN_STREAM=1 (with CK_PANEL_GROUP=1): the next panel's column update, step and
                                              # exchange on a second stream under the bulk update -- opt-in until a multi-GPU box has run it
                                              chain_stream=os.environ.get("CK_CHAIN_STREAM", "0") == "1")
        runner.prepare(m_total=m)
        h.set_option("time_gemm", 2)   # HIP events around this rank's Sigma trailing-update launches
        cand = os.environ.get("CK_PANEL_EXCHANGE_CANDIDATES")
        tuning_error = None
        try:
            runner.calibrate(candidates=cand.split(",") if cand else None)
            runner.autotune(0, pb["pcoords"])   # untimed calibration passes, in front of the W warm-up steps
        except Exception as e:   # noqa: BLE001 -- a tuning step must never cost the run its number: plain defaults instead
            tuning_error = f"{type(e).__name__}: {e}"
            runner.exchange, runner.G = "broadcast", 1

        def step():
            return runner.predict(0, pb["pcoords"])
        ex_final, G_final, comm_info, tune_info = runner.exchange, runner.G, runner.comm_info, runner.tune_info

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    tim = []
    rank_tim = []
    for _ in range(args.steps):
        pred, err = step()
        tim.append(h.timings())
        if world > 1:
            rank_tim.append(dict(runner.timings))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    tim_fused = tim
    seq_ms = None
    result_check = None
    resident = None
    if world == 1 and args.sweeps == "sequential":
        seq_ms = ms_per_step
    elif world == 1:
        # stage measurements: sequential passes of the same workload, outside the timed region -- and the check that the
        # product schedule of the timed steps (one sweep over the tall matrix, two streams) gives the sequence's BITS
        n_seq = max(1, min(args.steps, 3))
        h.set_option("solve_la", 0)     # the reference passes: ck_predict's sweep one launch after the other (chip to itself)
        step_sequential()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tim = []
        for _ in range(n_seq):
            pred_s, err_s = step_sequential()
            tim.append(h.timings())
        torch.cuda.synchronize()
        seq_ms = (time.perf_counter() - t0) / n_seq * 1e3
        # one more prediction on the resident factor with the product's defaults (Predictor.__call__ for the second field, a new
        # grid): ck_predict with the chain of the next panel group under the bulk of the current one (option solve_la)
        h.set_option("solve_la", -1)
        h.predict(0, pb["pcoords"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred_r, err_r = h.predict(0, pb["pcoords"])
        resident = {"ms": (time.perf_counter() - t0) * 1e3, "sweep_ms": h.timings()["solve_ms"],
                    "equals_sequential_pass": bool(np.array_equal(pred_r, pred_s) and np.array_equal(err_r, err_s)),
                    "what": "ck_predict on the resident factor, product defaults (host arrays -> host results): K2, the forward sweep with "
                            "look-ahead, reduce"}
        result_check = {"timed_steps_equal_sequential_passes": bool(np.array_equal(pred, pred_s) and np.array_equal(err, err_s)),
                        "max_abs_diff": float(max(np.max(np.abs(pred - pred_s)), np.max(np.abs(err - err_s)))),
                        "what": "(pred, pred_err) of the last timed step (ck_factor_predict, the product schedule) against the last "
                                "sequential pass (ck_factor, ck_predict) on all grid points; the run fails if they differ"}
    per_rank = None
    if dist is not None:
        # every rank's own breakdown of a step (HIP events on its stream, mean over the timed steps)
        keys = ("panel_ms", "update_ms", "bcast_wait_ms", "assemble_ms", "finish_ms")
        mine = torch.tensor([float(np.mean([t.get(k, 0.0) for t in rank_tim])) for k in keys], dtype=torch.float64, device="cuda")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(zip(keys, [round(float(x), 3) for x in t.cpu().tolist()])) for t in allr]

    # PCIe-inclusive figure (never `value`): host arrays to host results in one timed region -- new handle, model, upload of the
    # sites (with the Hilbert sort on the host), tables, assembly, factorisation, sweep, results back.  In front of the
    # configs[3] block: a handle created right after 47 GB were freed pays a hipMalloc stall of seconds that is not this path's.
    pcie = None
    if world == 1 and args.config == 2 and n == 20000:
        t0 = time.perf_counter()
        h2 = native.Handle(local_rank)
        h2.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h2.set_metric(pb["metric"])
        for k in range(2):
            h2.set_data(k, pb["coords"][k], pb["values"][k])
        h2.assemble_joint()
        if args.sweeps == "sequential":
            h2.factor()
            h2.predict(0, pb["pcoords"])
        else:
            h2.factor_predict(0, pb["pcoords"])
        dtc = time.perf_counter() - t0
        h2.close()
        pcie = {"ms": dtc * 1e3, "grid_points_per_s": m / dtc,
                "what": "one cold pass, host arrays -> host results, new handle (not the headline value)"}

    # ---- secondary block: BASELINE configs[3] (n_obs = 50 000 per process, N = 100 000) on the same ranks, one timed pass ----
    config3 = None
    want_c3 = (not args.no_config3 and args.config == 2 and n == 20000
               and os.environ.get("CK_BENCH_CONFIG3", "1" if (world == 1 or backend == "nccl") else "0") == "1")
    if want_c3:
        try:
            pb3 = synth.conus_problem(50000, seed=20004, params=params)
            m3 = len(pb3["pcoords"])
            if world == 1:
                h.close()          # its 10 GB are not needed any more; N = 100 000 takes 40 + 7 GB
            h3 = native.Handle(local_rank)
            h3.set_stream(torch.cuda.current_stream().cuda_stream)
            h3.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
            h3.set_metric(pb3["metric"])
            for k in range(2):
                h3.set_data(k, pb3["coords"][k], pb3["values"][k])
            if world == 1:
                def step3():
                    h3.assemble_joint()
                    info3, p3, e3 = h3.factor_predict(0, pb3["pcoords"])
                    if info3 != 0:
                        raise RuntimeError(f"configs[3]: Sigma not positive definite at minor {info3}")
                    return p3, e3
            else:
                runner.h = None
                del runner
                h.close()
                torch.cuda.empty_cache()
                r3 = distributed.DistributedJoint(h3, rank, world, dist_module=dist, device=torch.device("cuda", local_rank),
                                                  exchange=ex_final, panel_group=G_final)
                r3.prepare(m_total=m3)

                def step3():
                    return r3.predict(0, pb3["pcoords"])
            step3()                # untimed: first use (allocations, tables)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            p3, e3 = step3()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            dt3 = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt3], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt3 = float(t.item())
            N3 = 100000
            t3 = h3.timings()
            config3 = {"workload": f"configs[3]: n_obs=50000/process (N={N3}), 0.05-degree CONUS lattice, haversine, {m3}-point grid, "
                                   f"Matern set {args.params}, block-column-cyclic x{world}",
                       "ms_per_step": dt3 * 1e3, "grid_points_per_s": m3 / dt3, "steps": 1, "warmup": 1,
                       "frac_of_mfma_peak_whole_step": (N3 ** 3 / 3 + N3 ** 2 * m3) / dt3 / 1e12 / PEAK_F64_MFMA_TFLOPS,
                       "finite": bool(np.all(np.isfinite(p3)) and np.all(np.isfinite(e3))),
                       "assemble_sigma_ms": t3["assemble_sigma_ms"], "assemble_c0_ms": t3["assemble_aux_ms"]}
            if world > 1:
                config3["rank0"] = dict(r3.timings)
                config3["exchange"], config3["panel_group"] = r3.exchange, r3.G
            h3.close()
        except Exception as e:   # noqa: BLE001 -- a secondary block never costs the run its headline line
            config3 = {"error": f"{type(e).__name__}: {e}"}

    if dist is not None:
        # nothing below needs the other ranks: they leave; rank 0 times the CPU baseline and prints the line
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank == 0:
        out = {
            "metric": "cokriging grid-points/s at n_obs=20k bivariate",
            "value": m / (ms_per_step / 1e3),
            "unit": "grid-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"configs[1]: n_obs={n}/process (N={N}) uniform sites on the unit square, Euclidean, "
                                    f"full 2n x 2n solve, {m}-point grid, closed-form Matern set B (unit square)")
                       if args.config == 1 else
                                   (f"configs[2]: n_obs={n}/process (N={N}) SIF+XCO2-like residuals on the 0.05-degree "
                                    f"CONUS lattice, haversine, full 2n x 2n solve, {m}-point 0.5-degree grid, "
                                    f"Matern set {args.params}"),
                       "n_obs": n, "N": N, "m": m, "params": pv, "partition": f"block-column-cyclic x{world}"},
        }
        nK = -(-N // 512)
        Npad = nK * 512
        cov_bytes = 8.0 * (N * (N + 1) / 2)
        k2_bytes = 8.0 * N * m

        def hbm(b, ms):
            return {"GB": b / 1e9, "ms": ms, "GBs": b / (ms / 1e3) / 1e9, "frac_of_hbm_peak": b / (ms / 1e3) / 1e9 / PEAK_HBM_GBS}
        if world > 1:
            out["per_rank"] = per_rank
            names = {"broadcast": "panel broadcast", "sag": "panel scatter + in-place all-gather (two collectives)",
                     "p2p": "panel scatter + point-to-point all-gather (batch_isend_irecv)"}
            out["comm"] = {"collective": names.get(ex_final, ex_final) + f" over {backend}, one per 512-column panel, look-ahead kept",
                           "exchange": ex_final, "exchange_calibration": comm_info,
                           "panel_group": G_final, "panel_group_tuning": tune_info, "tuning_error": tuning_error,
                           "panels": nK, "bytes_received_per_rank_per_step": int(sum((nK * 512 - K * 512) * 512 * 8 + 8 * 64 * 64 * 8
                                                                                     for K in range(nK) if K % world != 0)),
                           "note": "bcast_wait_ms = time the rank's stream waited for a panel after its own updates were done "
                                   "(exposed communication); panel_ms / update_ms = panel steps / trailing + right-hand-side updates"}
            # rank 0's share of the trailing updates: block column J (owned if J % world == 0) receives J panels
            flops = sum(J * ((Npad - J * 512) * 512 - 512 * 511 / 2) * 2 * 512 for J in range(0, nK, world))
            tl = tim[-1]
            syrk_s = np.mean([t["syrk_ms"] for t in tim]) / 1e3
            out["roofline"] = {
                "kernel": "k_syrk_group_d on rank 0 (the multi-GPU form: K = 512 G, the owned block columns)",
                "bound": "mfma", "achieved": flops / syrk_s / 1e12 if syrk_s > 0 else None, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / syrk_s / 1e12 / PEAK_F64_MFMA_TFLOPS if syrk_s > 0 else None, "traffic": None,
                "traffic_source": "not measured: the PMC passes (profiles/r04_traffic.json) are of the single-GPU command; a launch of the "
                                  "multi-GPU form covers this rank's block columns only",
                "launches_per_step": tl["syrk_launches"], "avg_launch_ms": tl["syrk_ms"] / max(tl["syrk_launches"], 1),
                "algorithmic_flops_per_step": flops,
                "measured_on": "the timed steps: HIP events around every Sigma trailing-update launch of rank 0 on its stream",
            }
            # rank 0's covariance assembly: its owned block columns of Sigma + its shard of the c0^T rows
            own_bytes = 8.0 * sum(max((N - K * 512) * 512 - 512 * 511 / 2, 0.0) for K in range(0, nK, world))
            chunk = -(-m // world)
            k1_ms = float(np.mean([t["assemble_sigma_ms"] for t in tim]))
            k2_ms = float(np.mean([t["assemble_aux_ms"] for t in tim]))
            out["stages"] = {"rank0": {k: float(np.mean([t.get(k, 0.0) for t in rank_tim])) for k in ("assemble_ms", "panel_ms", "update_ms", "bcast_wait_ms", "finish_ms")},
                             "cov_assembly": {"K1_sigma_rank0": hbm(own_bytes, k1_ms), "K2_c0_rank0": hbm(8.0 * N * chunk, k2_ms),
                                              "combined_rank0": hbm(own_bytes + 8.0 * N * chunk, k1_ms + k2_ms),
                                              "definition": "rank 0's share of SURVEY 8(d)'s bytes (its block columns of the lower triangle, "
                                                            "its shard of the c0 rows) over its own K1 + K2 device time"},
                             "whole_step_frac_of_mfma_peak_x_gpus": (N ** 3 / 3 + N ** 2 * m) / (ms_per_step / 1e3) / 1e12 / (PEAK_F64_MFMA_TFLOPS * world)}
        else:
            tl = tim[-1]
            tf = tim_fused[-1]
            tall = args.sweeps != "sequential" and tf["fused_sweeps_ms"] > 0 and tf["syrk_launches"] > 0
            # the PMC passes are of the headline workload only
            traffic, traffic_source = measured_traffic() if (args.config == 2 and n == 20000) else (None, None)
            fl_all = trailing_update_flops(N) + aux_update_flops(N, m)
            if tall:
                # the dominant kernel of the timed steps is k_tall_group_d: every update of the tall matrix [Sigma; c0^T; z^T]
                span_s = float(np.mean([t["fused_sweeps_ms"] for t in tim_fused])) / 1e3
                sum_ms = float(np.mean([t["syrk_ms"] for t in tim_fused]))
                union_s = float(np.mean([t["tall_union_ms"] for t in tim_fused])) / 1e3
                nl = tf["syrk_launches"]
                out["roofline"] = {
                    "kernel": "k_tall_group_d (every trailing update of the tall matrix [Sigma; c0^T; z^T]: Cholesky AND forward substitution, "
                              "v_mfma_f64_16x16x4_f64, 128x128 tiles, 8 waves, LDS-DMA staging, K = 512 x group)",
                    "bound": "mfma", "achieved": fl_all / union_s / 1e12, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": fl_all / union_s / 1e12 / PEAK_F64_MFMA_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_source,
                    "launches_per_step": nl, "avg_launch_ms": sum_ms / max(nl, 1), "sum_launch_ms": sum_ms,
                    "union_launch_ms": union_s * 1e3, "span_ms": span_s * 1e3,
                    "frac_over_sweep_span": fl_all / span_s / 1e12 / PEAK_F64_MFMA_TFLOPS,
                    # for the record: the same flops over the SUM of the launch durations (= launches x avg_launch_ms, what a
                    # per-kernel average from rocprofv3 multiplies out to).  The launches of the two streams overlap each other,
                    # so a launch's duration includes the time it shares the chip with another launch of the same kernel: this
                    # figure counts that time twice and is a lower bound, not the kernel's rate
                    "frac_over_sum_of_launch_durations": fl_all / (sum_ms / 1e3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                    "algorithmic_flops_per_step": fl_all,
                    # everything a step computes (N^3/3 + N^2 m: the kernel's launches + the panel steps) over the whole step's
                    # wall time (assembly, sweep, reductions, host): the figure VERDICT r03 quotes as "whole step"
                    "whole_step": {"flops": N ** 3 / 3 + N ** 2 * m, "ms": ms_per_step,
                                   "frac_of_mfma_peak": (N ** 3 / 3 + N ** 2 * m) / (ms_per_step / 1e3) / 1e12 / PEAK_F64_MFMA_TFLOPS},
                    "measured_on": "the timed steps themselves, HIP events around every launch of the kernel on the stream it is launched "
                                   "on: achieved = the algorithmic flops of its launches of a step (2 K per lower-triangle entry and per entry of the m + 1 "
                                   "right-hand-side rows; padding and the upper halves of diagonal tiles are executed but not counted) "
                                   "over union_launch_ms, the time during which the kernel is running at all -- its launches run on two "
                                   "streams and overlap each other (look-ahead), so sum_launch_ms exceeds it (avg_launch_ms = sum / "
                                   "launches is what rocprofv3's kernel stats average: profiles/r04_bench_n20k_kernel_stats.csv).  "
                                   "frac_over_sweep_span divides by the sweep's whole span instead (first launch to last: also the "
                                   "first panel chain and the tail, where no update launch runs); whole_step is everything a step "
                                   "computes over ms_per_step",
                }
            else:
                flops = trailing_update_flops(N)
                syrk_s = np.mean([t["syrk_ms"] for t in tim]) / 1e3
                out["roofline"] = {
                    "kernel": "k_syrk_group_d (Cholesky trailing update over panel groups, v_mfma_f64_16x16x4_f64, 128x128 tiles, 8 waves, LDS-DMA staging)",
                    "bound": "mfma", "achieved": flops / syrk_s / 1e12, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / syrk_s / 1e12 / PEAK_F64_MFMA_TFLOPS,
                    "traffic": None, "traffic_source": None,
                    "launches_per_step": tl["syrk_launches"], "avg_launch_ms": tl["syrk_ms"] / max(tl["syrk_launches"], 1),
                    "algorithmic_flops_per_step": flops,
                    "measured_on": "the timed steps (--sweeps sequential)" if args.sweeps == "sequential" else
                                   f"{len(tim)} sequential passes behind the timed region (ms_per_step of those passes: {seq_ms:.1f})",
                }
            syrk_seq_s = np.mean([t["syrk_ms"] for t in tim]) / 1e3
            aux_s = np.mean([t["aux_gemm_ms"] for t in tim]) / 1e3
            # covariance assembly = K1 (Sigma, lower triangle) + K2 (c0^T rows), algorithmic bytes
            # 8 [N (N + 1) / 2 + N m]; stage timers = HIP events around the device work of each (mean over the TIMED steps)
            k1_ms = float(np.mean([t["assemble_sigma_ms"] for t in tim_fused]))
            k2_ms = float(np.mean([t["assemble_aux_ms"] for t in tim_fused]))
            cov_assembly = {"K1_sigma": hbm(cov_bytes, k1_ms), "K2_c0": hbm(k2_bytes, k2_ms),
                            "combined": hbm(cov_bytes + k2_bytes, k1_ms + k2_ms),
                            "definition": "SURVEY 8(d): 8 [N (N + 1) / 2 + N m] algorithmic bytes over K1 + K2, HIP events around the device "
                                          "work of each, mean over the timed steps"}
            out["result_check"] = result_check
            out["stages"] = {
                "timed_steps": None if args.sweeps == "sequential" else
                               {"assemble_sigma_ms": tf["assemble_sigma_ms"], "assemble_c0_ms": tf["assemble_aux_ms"],
                                "sweep_ms": tf["fused_sweeps_ms"] if tf["fused_sweeps_ms"] > 0 else None,
                                "sweeps": ("ONE sweep over the tall matrix [Sigma; c0^T; z^T] (ck_factor_predict, option tall_sweep)" if tall else
                                           "two overlapped sweeps (ck_factor_predict)") if tf["fused_sweeps_ms"] > 0 else
                                          "sequential inside ck_factor_predict (more than 128 panels: the library's automatic rule)",
                                "chain_end_ms": tf["factor_ms"],
                                "reduce_ms": tf["reduce_ms"]},
                "sequential_passes_ms_per_step": seq_ms,
                "note": "factor_ms .. solve_gemm_tflops below are from the sequential passes (ck_factor, then ck_predict: k_syrk_group_d / "
                        "k_aux_group_d, each with the chip to itself)",
                "factor_ms": tl["factor_ms"], "solve_ms": tl["solve_ms"], "reduce_ms": tl["reduce_ms"],
                "cholesky_tflops": (N ** 3 / 3) / (tl["factor_ms"] / 1e3) / 1e12,
                "cholesky_frac_of_mfma_peak": (N ** 3 / 3) / (tl["factor_ms"] / 1e3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                "syrk_frac_of_mfma_peak_sequential": trailing_update_flops(N) / syrk_seq_s / 1e12 / PEAK_F64_MFMA_TFLOPS if syrk_seq_s > 0 else None,
                "solve_gemm_tflops": aux_update_flops(N, m) / aux_s / 1e12 if aux_s > 0 else None,
                "cov_assembly": cov_assembly,
                "cov_assembly_GBs": cov_assembly["combined"]["GBs"],
                "cov_assembly_frac_of_hbm_peak": cov_assembly["combined"]["frac_of_hbm_peak"],
                # 14 FP64 operations per tabulated entry (3 sub, mul, 2 fma: squared chord;
                # sub: offset from the interval centre; 7 fma: Horner) against the 78.6 TF vector peak
                "cov_assembly_frac_of_fp64_valu_peak": 14.0 * (N * (N + 1) / 2) / (k1_ms / 1e3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                # factor reused: grid-points/s of one more ck_predict on the resident L (K2 + K4 + reduce)
                "amortised_grid_points_per_s": m / ((tl["assemble_aux_ms"] + tl["solve_ms"] + tl["reduce_ms"]) / 1e3),
                "resident_factor_predict": resident,
            }
            if pcie is not None:
                out["pcie_inclusive"] = pcie
        if config3 is not None:
            out["config3"] = config3
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, m)
        print(json.dumps(out), flush=True)
        if (result_check is not None and not result_check["timed_steps_equal_sequential_passes"]) or \
                (resident is not None and not resident["equals_sequential_pass"]):
            raise SystemExit("bench.py: the timed steps' results differ from the sequential passes' (result_check)")


if __name__ == "__main__":
    main()
