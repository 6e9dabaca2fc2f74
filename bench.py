#!/usr/bin/env python3
"""bench.py -- W2 patient-pairs/sec for the full N x N distance matrix (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N              (no launcher: ONE process drives the N devices, pilot_ot_multi_*)

One "step" = one complete pass of the hot path over the workload: every one of the N^2 ordered patient pairs solved
with POT's sinkhorn_stabilized semantics (prep kernel + pair-grid kernel + tau-tracking kernel), and for N_gpus > 1
the RCCL all-gather + row interleave that assembles the full matrix on every rank.  Inputs (N x K proportions, K x K
cost) are resident in HBM before the timed region.  Everything goes through the C ABI of libpilot_ot.so (ctypes); this
script imports neither torch nor any other GPU framework -- under a launcher it only reads RANK / LOCAL_RANK /
WORLD_SIZE, and the RCCL unique id travels through a temp file (pilot_amd/multi.py).

Workload: BASELINE configs[2] at reg = 0.1 -- 600 patients x 50 cell types x 30 PCA dims, the configuration the metric
is quoted on (it fits one GPU).  Total work is fixed as GPUs are added (the 600^2 pair grid is row-sharded
round-robin), hence "scaling": "strong".  The line also carries, as extra keys: `value_host_to_host` (numpy in ->
numpy out, SURVEY.md 8(d)'s definition of t), `reg_sweep` (reg 0.01 / 0.1 / 1.0 with parity against the oracle on a row
sample), `exact_emd` (the reference's DEFAULT mode on the same cohort) and `c4` (2000 x 100, the shape that scales).

`roofline` is for the dominant kernel (the MFMA pair-grid kernel): achieved = algorithmic flop of one launch / its mean
duration (HIP events on the launch stream, recorded inside the timed region).  `cpu_baseline` is the CPU oracle (C fp64
restatement of POT's loop) timed on this box's host cores on a bounded sample of the same workload (rank 0, 1 GPU);
its all-cores leg covers the whole grid when that takes seconds (c3 does), and `parity` is then max|step output - oracle|
over EVERY pair the step produced.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA == f32 vector peak
PEAK_F64_MFMA_TFLOPS = 78.6
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
PEAK_F16_MFMA_TFLOPS = 2500.0   # same pipe, same dense rate for fp16 inputs
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03")  # newest first: the offline rocprofv3 measurements bench.py quotes next to its live ones


def profile_file(name):
    """(path, round) of the newest committed profiles/rNN/<name>, or (None, None)."""
    for r in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", r, name)
        if os.path.exists(path):
            return path, r
    return None, None


def host_cores():
    """CPU cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a 256-thread box
    whose container is limited to a few cores must not be reported as 256 -- VERDICT r03 weak #6)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                            # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                        # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def algorithmic_flops(iters, K, period=20):
    """per pair, iters*(4K^2+2K) + ceil(iters/period)*(2K^2+3K) + 3K^2."""
    it = iters.astype(np.float64)
    return float(np.sum(it * (4 * K * K + 2 * K) + np.ceil(it / period) * (2 * K * K + 3 * K) + 3 * K * K))


class DevBuf:
    def __init__(self, L, nbytes):
        from pilot_amd import _lib
        self.L, self.p = L, ctypes.c_void_p()
        _lib.check(L.pilot_ot_dev_alloc(ctypes.byref(self.p), int(nbytes)))

    def free(self):
        if self.p:
            self.L.pilot_ot_dev_free(self.p)
            self.p = ctypes.c_void_p()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", help="synthetic config (c2 | c3 | c4)")
    ap.add_argument("--reg", type=float, default=0.1)
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "fp64", "bf16x3", "f16x2"])
    ap.add_argument("--mode", default="sinkhorn", choices=["sinkhorn", "emd", "cellw2"],
                    help="emd: time the exact-OT pair grid (the reference's default mode) instead; cellw2: the cell-level W2 "
                         "extension at BASELINE config 5 (200 patients x 5000 cells x 30 dims; takes about two minutes)")
    ap.add_argument("--cell-patients", type=int, default=200)
    ap.add_argument("--cell-cells", type=int, default=5000)
    ap.add_argument("--ramp-steps", type=int, default=-1,
                    help="untimed calls before the warm-up steps that bring the GPU clocks up from idle (0 to disable; "
                         "default: 150 for grids up to the size of c3, 5 beyond -- about 0.2 s either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_host_to_host / reg_sweep / exact_emd / c4 / c5_cellw2")
    ap.add_argument("--no-c5", action="store_true", help="skip the c5_cellw2 key of the default line (one 35 s pass over BASELINE config 5)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU work budget of the baseline sample")
    ap.add_argument("--force-comm", action="store_true",
                    help="testing on a 1-GPU box: run the one-process-per-GPU path (RCCL communicator, all-gather, max over "
                         "ranks) with a world of one")
    ap.add_argument("--logical-shards", action="store_true",
                    help="testing on a 1-GPU box: --gpus N becomes N logical shards on device 0 (peer-copy gather)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    from pilot_amd import _lib, engine, multi, sharding
    from pilot_amd.synthetic import CONFIGS, make_problem

    L = _lib.load()
    n_dev = _lib.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X; pilot_amd has no CPU path")
    single_process_multi = world == 1 and args.gpus > 1
    if world > 1:
        _lib.check(L.pilot_ot_set_device(local_rank))
    cfg = CONFIGS[args.config]
    P, M = make_problem(**cfg)
    N, K = P.shape
    prec = args.precision
    run_prec = prec                       # what the library is asked for
    if prec == "auto":
        # what AUTO resolves to for this shape: the label of the line.  Beyond max(M)/reg = 60 AUTO means the two-band
        # bf16-split tracking kernel with an f64 pass for the pairs that need it ("mixed"), which only "auto" selects
        code = L.pilot_ot_auto_precision_for(float(M.max()) / args.reg, K, int(np.array_equal(M, M.T)))
        prec = {1: "fp32", 2: "mixed", 3: "bf16x3", 6: "f16x2"}[code]
        run_prec = "auto" if prec == "mixed" else prec

    if args.mode == "emd":
        out = bench_emd(args, L, P, M, cfg)
        print(json.dumps(out), flush=True)
        return
    if args.mode == "cellw2":
        print(json.dumps(bench_cellw2(args)), flush=True)
        return

    per_rank = None
    if single_process_multi:
        devices = [0] * args.gpus if args.logical_shards else list(range(args.gpus))
        with multi.stdout_to_stderr():       # (RCCL's version banner is a printf: this program's stdout carries one JSON line)
            mp = multi.MultiPlan(P, M, devices=devices)

        def step():
            mp.sinkhorn(args.reg, precision=run_prec)

        def fence():
            mp.sync()
        comm = None
    else:
        rb, re_, rs = sharding.shard_rows(N, rank, world)
        n_local = sharding.n_local_rows(N, rank, world)
        n_pad = sharding.n_padded_rows(N, world)
        plan = engine.DevicePlan(P, M, n_rows_max=max(n_pad, 1))       # P, M -> HBM (resident from here on)
        # HIP events around the pair-grid kernel on every 4th step of the timed region (on every step the four event records
        # cost a 0.69 ms step 2 %; `roofline.kernel_ms` is the mean of the steps that carry them)
        TIMING_STRIDE = 4 if args.steps >= 8 else 1
        plan.enable_timing(TIMING_STRIDE)
        with multi.stdout_to_stderr():       # (RCCL's version banner is a printf: this program's stdout carries one JSON line)
            comm = multi.Comm(rank, world) if (world > 1 or args.force_comm) else None
        if comm:
            d_stage, d_full = DevBuf(L, 8 * world * n_pad * N), DevBuf(L, 8 * N * N)
            zeros = np.zeros(n_pad * N)                                  # (kept alive across the copy)
            _lib.check(L.pilot_ot_memcpy_h2d(plan.dE, zeros.ctypes.data, 8 * n_pad * N))   # padding rows = 0

        def step():
            plan.run(args.reg, row_begin=rb, row_end=re_, row_step=rs, precision=run_prec)
            if comm:
                comm.all_gather_rows(plan.dE, n_pad, N, d_stage.p, d_full.p)

        def fence():
            if comm:
                comm.barrier()
            plan.sync()

    # the GPU sits idle while the inputs are generated and the plan is built: bring the clocks up before the W warm-up steps
    # (untimed, disclosed as `clock_ramp_s`; same number of calls on every rank)
    if args.ramp_steps < 0:
        args.ramp_steps = 150 if N * N * K <= 30_000_000 else 5      # (a function of the workload only: every rank agrees)
    t_ramp = time.perf_counter()
    for _ in range(args.ramp_steps):
        step()
    fence()
    ramp_s = time.perf_counter() - t_ramp
    for _ in range(args.warmup):
        step()
    fence()
    if not single_process_multi:
        plan.enable_timing(TIMING_STRIDE)          # (counters restart: the timed steps 0, 4, 8, .. carry the events)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if comm:
        elapsed = comm.all_reduce_max(elapsed)
    ms_per_step = 1e3 * elapsed / args.steps
    value = N * N / (elapsed / args.steps)

    # ---- the assembled matrix + per-pair update counts ------------------------------------------------------------
    if single_process_multi:
        E, info = mp.fetch(info=True)
        iters = info["iters"]
        grid_ms, gather_ms = mp.times_ms()
        kern_ms, track_ms = float(np.max(grid_ms)), None
        rn, rr = mp.rccl_info()
        per_rank = {"grid_ms_per_shard": [round(float(x), 4) for x in grid_ms], "gather_ms": round(gather_ms, 4),
                    "rccl_ranks": rn, "rccl_user_rank": rr,
                    "rccl_note": "ncclCommCount / ncclCommUserRank of every shard's communicator (0 / -1: peer-copy gather, no RCCL)"}
    else:
        main_ms, track = plan.kernel_times_ms(max_n=min((args.steps + TIMING_STRIDE - 1) // TIMING_STRIDE, 64))
        kern_ms = float(np.mean(main_ms)) if len(main_ms) else float("nan")
        track_ms = float(np.mean(track)) if len(track) else None
        if track_ms is not None and track_ms > kern_ms:        # small reg: every pair runs the tau-tracking launch
            kern_ms, track_ms = track_ms, kern_ms
        _, info = plan.fetch(n_rows=n_local)
        iters = info["iters"]
        if comm:
            E = np.empty((N, N))
            _lib.check(L.pilot_ot_memcpy_d2h(E.ctypes.data, d_full.p, 8 * N * N))
            cnt, urank = comm.info()
            # the collective alone (all-gather + row interleave of this rank's block), host-timed over a few calls
            fence()
            tg = time.perf_counter()
            for _ in range(10):
                comm.all_gather_rows(plan.dE, n_pad, N, d_stage.p, d_full.p)
            fence()
            gather_ms = 1e2 * (time.perf_counter() - tg)
            per_rank = {"rank0_kernel_ms": round(kern_ms, 4), "rank0_step_minus_kernel_ms": round(ms_per_step - kern_ms, 4),
                        "max_kernel_ms_over_ranks": round(comm.all_reduce_max(kern_ms), 4),
                        "gather_ms": round(comm.all_reduce_max(gather_ms), 4),
                        "rccl_ranks": cnt, "rccl_user_rank": urank, "max_rank_seen": int(comm.all_reduce_max(float(rank))),
                        "rccl_note": "ncclCommCount / ncclCommUserRank of rank 0's communicator; max_rank_seen = all-reduce(max) of "
                                     "every rank's index"}
        else:
            E = plan.fetch(n_rows=N)[0]
    assert E.shape == (N, N) and np.isfinite(E).all(), "bench produced a non-finite matrix"
    if args.reg >= 0.05:   # converged entropic costs are symmetric; a bad row interleave would break this
        assert float(np.abs(E - E.T).max()) < 1e-5, "assembled matrix is not symmetric: bad row interleave?"

    # ---- roofline of the dominant kernel (this rank's launches inside the timed region) -------------------------------
    share = 1.0 if not single_process_multi else 1.0 / args.gpus
    roofline, roofline_hbm = make_roofline(prec, K, iters, kern_ms, track_ms, share)
    if world == 1 and not single_process_multi:
        attach_offline_profile(roofline, args.config, args.reg, prec)

    n_gpus = args.gpus
    out = {
        "metric": "W2 patient-pairs/sec (full NxN EMD matrix)", "value": round(value, 1), "unit": "pairs/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": DTYPE_NOTE[prec],
        "data": "synthetic",
        "config": {"workload": "%s: %d patients x %d cell types x %d PCA dims, Sinkhorn reg=%g "
                               "(POT sinkhorn_stabilized semantics), all N^2 ordered pairs"
                               % (args.config, N, K, cfg["n_dims"], args.reg),
                   "n_patients": N, "n_cell_types": K, "n_pca": cfg["n_dims"], "reg": args.reg,
                   "pairs_per_step": N * N,
                   "parallelism": "pair-grid rows dealt round-robin over %d GPU(s)%s; %s"
                                  % (n_gpus, " + 1 RCCL all-gather" if n_gpus > 1 else "",
                                     "one process per GPU (pilot_ot_comm_*)" if world > 1 else
                                     ("one process, %d devices (pilot_ot_multi_*)" % n_gpus if single_process_multi
                                      else "single device"))},
        "roofline": roofline, "roofline_hbm": roofline_hbm,
        "value_device_resident": round(value, 1),
        "timed_region": "`value` (= value_device_resident): P, M resident in HBM -> N x N matrix resident in HBM on every rank, "
                        "as the bench contract prescribes; SURVEY 8(d)'s t (host arrays in -> host matrix out, PCIe inclusive) "
                        "is `value_host_to_host`, measured in the same run",
        "clock_ramp_s": round(ramp_s, 3), "clock_ramp_steps": args.ramp_steps,
    }
    if per_rank:
        out["multi_gpu"] = per_rank
        fl = shard_floor(args.config, n_gpus)
        if fl:
            per_rank["predicted_floor"] = fl
        per_rank["scaling_note"] = ("strong scaling: total work fixed.  c3 (this line) is bounded by the serial update chain of its slowest "
                                    "pairs, not by the collective (see predicted_floor); the `c4` key is the shape whose shards stay "
                                    "long enough to scale linearly")

    extras = not args.no_extras
    if extras:
        # the shape that scales: every rank takes part, so it is measured at every N
        out["c4"] = bench_c4(L, rank, world, comm, args, single_process_multi)
    if rank == 0 and world == 1 and not single_process_multi:
        if extras:
            out["value_host_to_host"] = host_to_host(P, M, args.reg, run_prec)
            out["precision_ladder"] = precision_ladder(P, M, args.reg, args.config)
            out["reg_sweep"] = reg_sweep(P, M, K)
            out["exact_emd"] = exact_emd_record(L, P, M, args.config, with_cpu=not args.no_cpu_baseline)
            if args.config == "c3" and not args.no_c5:
                c5 = bench_cellw2(args)       # BASELINE configs[4] (an extension): one pass over its 40 000 pairs, ~35 s
                out["c5_cellw2"] = {k: c5[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "checks", "cpu_baseline", "parity")}
            if args.config in ("c2", "c3"):
                from pilot_amd.synthetic import make_cells
                cohort = make_cells(cfg["n_patients"], cfg["n_types"], cfg["n_dims"], cfg["seed"], cfg["cells_per_patient"])
                out["e2e_tl_s"] = e2e_tl(cohort)
                out["prepass"] = prepass_record(L, cohort, with_cpu=not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            cb, cb_eq, cb_all, upd = cpu_baseline(P, M, args.reg, args.cpu_seconds, E, iters)
            out["cpu_baseline"], out["cpu_baseline_equal_updates"], out["cpu_baseline_all_cores"] = cb, cb_eq, cb_all
            out["equal_work_note"] = upd
            out["parity"] = {"pairs": cb_all["pairs"], "whole_grid": cb_all["whole_grid"], "max_abs_diff_vs_fp64_oracle": cb_all["max_abs_diff_vs_gpu"],
                             "tolerance": 1e-5, "what": "the step's output against the fp64 oracle (POT's stopping rule) on the pairs of "
                                                       "cpu_baseline_all_cores; the oracle is the checker here, never the thing timed as `value`"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if single_process_multi:
        mp.close()
    else:
        plan.close()
    if comm:
        comm.barrier()
        d_stage.free(); d_full.free()
        comm.close()



KERNEL_CFG = {"fp32": "CfgF32x16", "fp64": "CfgF64x16", "bf16x3": "CfgS32x16", "f16x2": "CfgH32x16", "mixed": "CfgS32x16 (two exponent bands, tracking)"}
DTYPE_NOTE = {"fp32": "f32", "fp64": "f64",
              "mixed": "f32 (exact 3-way bf16 splits in two exponent bands of the Gibbs kernel; pairs that leave the f32 range redone in f64)",
              "bf16x3": "f32 (products as exact 3-way bf16 splits on the bf16 MFMA, f32 accumulate)",
              "f16x2": "f32 (products as 2-way fp16 splits, 22 significant bits, on the f16 MFMA, f32 accumulate)"}


def make_roofline(prec, K, iters, kern_ms, track_ms, share=1.0):
    """`roofline` of the pair-grid kernel against the pipe it EXECUTES on.

    fp32 / fp64: v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 -- achieved = the algorithmic flop of SURVEY 8(d) over the
    kernel time, peak = the dense f32-input / f64 MFMA rate.
    bf16x3 / f16x2: the products run as piece products on the 16-bit matrix pipe (2.5 PF dense): achieved = the piece-MFMA
    flop ISSUED per launch (per 16-pair update 2 products x RT row-tiles x ceil(RT/2) k-blocks x (6 | 3) MFMAs of
    16x16x32 = 16 384 flop each; padding of K to 16 / 32 and the 3 or 6 piece products per term are part of it) over the
    kernel time, peak 2 500 TF.  `useful_frac` prices only the ALGORITHMIC flop against that same peak, and
    `f32_equivalent` is last round's figure (algorithmic flop against the f32-input MFMA peak): it can exceed 1 because
    this formulation does not execute on that pipe -- it is a comparison, not a roofline fraction."""
    flops_launch = algorithmic_flops(iters, K) * share
    s_bytes = 8 if prec == "fp64" else 4
    pairs_launch = int(iters.size * share)
    bytes_launch = float(pairs_launch) * (2 * K * s_bytes + s_bytes)     # 2*K*s + s per pair
    t = kern_ms * 1e-3
    alg_tf = flops_launch / t / 1e12
    roofline = {"bound": "mfma", "kernel": "pilot::sinkhorn_stream_kernel<%s, ...>" % KERNEL_CFG[prec]}
    if prec in ("bf16x3", "f16x2", "mixed"):
        rt = (K + 15) // 16
        terms = {"bf16x3": 6, "f16x2": 3, "mixed": 9}[prec]      # (mixed: 6 band-0 + 3 band-1 piece products per term block)
        mfma_launch = float(iters.sum()) / 16.0 * 2 * rt * ((rt + 1) // 2) * terms * share
        ex_tf = mfma_launch * 16384.0 / t / 1e12
        peak = PEAK_F16_MFMA_TFLOPS if prec == "f16x2" else PEAK_BF16_MFMA_TFLOPS
        roofline.update({
            "achieved": round(ex_tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ex_tf / peak, 4),
            "pipe": "v_mfma_f32_16x16x32_%s (dense 16-bit matrix pipe)" % ("f16" if prec == "f16x2" else "bf16"),
            "achieved_is": "piece-MFMA flop issued per launch / kernel time: %d piece products per term block, K padded %d -> %d "
                           "(contraction) x %d (rows)" % (terms, K, 32 * ((rt + 1) // 2), 16 * rt),
            "mfma_per_launch": mfma_launch,
            "useful_frac": round(alg_tf / peak, 4),
            "f32_equivalent": {"achieved": round(alg_tf, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "ratio": round(alg_tf / PEAK_F32_MFMA_TFLOPS, 4),
                               "note": "algorithmic f32 flop of SURVEY 8(d) against the f32-input MFMA peak: may exceed 1 (the "
                                       "products do not run on that pipe); a comparison with an f32-MFMA kernel, not a fraction "
                                       "of a roof"}})
    else:
        peak = PEAK_F64_MFMA_TFLOPS if prec == "fp64" else PEAK_F32_MFMA_TFLOPS
        roofline.update({"achieved": round(alg_tf, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(alg_tf / peak, 4),
                         "pipe": "v_mfma_f64_16x16x4_f64" if prec == "fp64" else "v_mfma_f32_16x16x4_f32",
                         "achieved_is": "algorithmic flop of SURVEY 8(d) per launch / kernel time"})
    roofline.update({
        "traffic": None,
        "kernel_ms": round(kern_ms, 4), "kernel_ms_source": "HIP events on the launch stream inside the timed region (every 4th step when steps >= 8), mean",
        "track_kernel_ms": round(track_ms, 4) if track_ms is not None else None,
        "algorithmic_flop_per_launch": flops_launch, "pairs_per_launch": pairs_launch,
        "mean_updates_per_pair": round(float(iters.mean()), 2)})
    roofline_hbm = {
        "bound": "hbm", "achieved": round(bytes_launch / t / 1e9, 2), "peak": PEAK_HBM_GBS,
        "unit": "GB/s", "frac": round(bytes_launch / t / 1e9 / PEAK_HBM_GBS, 5),
        "algorithmic_bytes_per_pair": 2 * K * s_bytes + s_bytes,
    }
    return roofline, roofline_hbm


def attach_offline_profile(roofline, config, reg, prec):
    """rocprofv3 measurements of this exact workload committed under profiles/ (bench.py cannot profile itself): HBM traffic
    per launch from the FETCH_SIZE / WRITE_SIZE passes, the kernel-trace average duration, and the matrix / vector pipe
    occupancy triple (SQ_VALU_MFMA_BUSY_CYCLES, 4 x SQ_ACTIVE_INST_VALU, SQ_VALU_MFMA_COEXEC_CYCLES over the SIMD cycles
    of the launch = GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); written by tools/gpu_round_report.sh, keyed by workload."""
    path, rnd = profile_file("traffic.json")
    if not path:
        return
    try:
        with open(path) as fh:
            tr = json.load(fh).get("%s|%g|%s" % (config, reg, prec))
    except (OSError, ValueError):
        return
    if not tr:
        return
    roofline["traffic"] = tr["traffic_bytes"]
    roofline["traffic_source"] = ("profiles/%s/traffic.json (rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE per launch, measured at git %s)"
                                  % (rnd, tr.get("git", "?")))
    if tr.get("kernel_ms_rocprofv3"):
        roofline["kernel_ms_rocprofv3"] = tr["kernel_ms_rocprofv3"]
        roofline["frac_rocprofv3"] = round(roofline["frac"] * roofline["kernel_ms"] / tr["kernel_ms_rocprofv3"], 4)
    for k in ("mfma_busy", "valu_busy", "coexec", "sq_insts_mfma", "sq_insts_valu"):
        if tr.get(k) is not None:
            roofline[k] = tr[k]
    if tr.get("mfma_busy") is not None:
        roofline["pmc_note"] = ("mfma_busy / valu_busy / coexec: fractions of the launch's SIMD cycles the matrix pipe was busy, a vector "
                                "instruction was executing, and both at once (profiles/%s/rocprofv3_pmc_summary_bench_c3.txt)" % rnd)


# ---- extras ---------------------------------------------------------------------------------------------------------
def e2e_tl(ad, reps=3):
    """What a user of the reference calls: tl.wasserstein_distance(adata) on the cell-level cohort of this config (cells x PCA
    dims + three obs columns, object dtype like the reference's tutorials, then categorical like AnnData stores them) ->
    adata.uns, wall time, both modes (best of 3 after one warm-up call).  Host label work + H2D of the embedding + every
    device kernel + frames."""
    from pilot_amd import tl
    os.environ.setdefault("PILOT_AMD_NO_RESULTS_DIR", "1")
    out = {"cells": int(ad.X.shape[0]), "what": "tl.wasserstein_distance(adata, ...) end to end, best of %d; obs label columns of object dtype "
                                               "(the `_categorical` keys: the same columns as pandas Categoricals, what AnnData stores)" % reps}
    obs_object = ad.obs
    for suffix, obs in (("", obs_object), ("_categorical", obs_object.astype("category"))):
        ad.obs = obs
        for mode, key in (("reg", "sinkhorn_reg0.1"), ("unreg", "exact_emd")):
            best = float("inf")
            for r in range(reps + 1):
                ad.uns = {}
                t = time.perf_counter()
                tl.wasserstein_distance(ad, emb_matrix="X_pca", regularized=mode, reg=0.1)
                if r:
                    best = min(best, time.perf_counter() - t)
            out[key + suffix] = round(best, 4)
    ad.obs = obs_object
    ad.uns = {}
    return out


def prepass_record(L, ad, with_cpu=True, reps=10):
    """The device pre-pass of tl.wasserstein_distance on the cell-level cohort (Cluster_Representations,
    Trajectory.py:400-430, and the per-type medians of cost_matrix, :462-466): (sample, type) histogram + proportions + first rows +
    K x D medians from one upload of the two code columns, the embedding resident in HBM.  `ms` = wall clock of the C-ABI call
    (code columns H2D and results D2H inside), `device_ms` = HIP events around its kernels; the roofline prices the kernels
    against HBM on the bytes the algorithm moves: the codes, one read + one write of the embedding (rows grouped by type, as
    order-preserving keys), then one read per 8-bit radix digit."""
    import ctypes
    import pandas as pd
    from pilot_amd import _lib, engine, tl
    X = ad.obsm["X_pca"]
    C, D = X.shape
    ccodes, cells = tl._first_appearance_codes(ad.obs["cell_types"])
    scodes, samples = tl._first_appearance_codes(ad.obs["sampleID"])
    K, N = len(cells), len(samples)
    up = engine.EmbeddingUpload(X)
    wall, dev = [], []
    try:
        for r in range(reps + 2):
            t = time.perf_counter()
            P, first, cen = up.prepass(ccodes, scodes, N, K, regulizer=0.2, n_total=C)
            dt = time.perf_counter() - t
            ms = ctypes.c_float(0.0)
            _lib.check(L.pilot_ot_prepass_device_ms(ctypes.byref(ms)))
            if r >= 2:
                wall.append(dt * 1e3); dev.append(float(ms.value))
    finally:
        up.close()
    s = X.itemsize
    sweeps = 2 + s                                            # group: read + write; one read per 8-bit digit of an s-byte key
    model = C * D * s * sweeps + C * 4 * 3                    # + the code columns: both read by the count pass, one by the grouping
    dev_ms = float(np.median(dev))
    traffic, traffic_src = None, None
    tf, rnd = profile_file("prepass_traffic.json")
    if tf:
        t = json.load(open(tf))
        if t.get("cells") == C and t.get("dims") == D:
            traffic, traffic_src = t.get("hbm_bytes_per_call"), "profiles/%s/prepass_traffic.json (%s)" % (rnd, t.get("how", "rocprofv3 --pmc"))
    out = {"what": "proportions + first rows + per-type medians on the device, %d cells x %d dims (%s), %d types, %d samples" % (C, D, X.dtype, K, N),
           "ms": round(float(np.median(wall)), 4), "device_ms": round(dev_ms, 4),
           "roofline": {"bound": "hbm", "achieved": round(model / (dev_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(model / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": int(model),
                        "model": "codes (12 B/cell) + embedding x (1 read + 1 write + %d digit passes)" % s,
                        "frac_one_read": round(C * D * s / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        "traffic": traffic, "traffic_source": traffic_src}}
    if with_cpu:
        # the reference's own lines on the same array: cost_matrix's median loop (Trajectory.py:462-466) -- pandas, one thread
        data = pd.DataFrame(X, columns=["PCA_%d" % (i + 1) for i in range(D)])
        annot = ad.obs[["cell_types", "sampleID", "status"]]
        annot.columns = ["cell_type", "sampleID", "status"]
        t = time.perf_counter()
        ref = [list(data[annot[annot.columns[0]] == i].median(axis=0)) for i in annot[annot.columns[0]].unique()]
        cpu_s = time.perf_counter() - t
        out["cpu_baseline"] = {"value": round(cpu_s * 1e3, 1), "unit": "ms", "cores": 1, "kind": "reference",
                               "sample": "the reference's own median loop (Trajectory.py:462-466, pandas boolean mask + DataFrame.median per "
                                         "cell type) on the same %d x %d array, whole; medians only (its proportions loop is not timed)" % (C, D)}
        out["parity"] = {"medians_bit_exact_vs_pandas": bool(np.array_equal(np.asarray(ref, dtype=np.float64), cen, equal_nan=True))}
    return out


def shard_floor(config, n_gpus):
    """What ONE GPU takes for a 1/G row shard of this workload (tools/shard_floor.py, measured on one MI355X): the time a
    perfectly overlapped G-GPU run cannot beat.  None when the table has no entry."""
    path, rnd = profile_file("shard_floor.json")
    try:
        with open(path) as fh:
            e = json.load(fh)["floor"].get(config, {}).get(str(n_gpus))
    except (OSError, ValueError, KeyError, TypeError):
        return None
    if not e:
        return None
    return {"one_gpu_shard_kernel_ms": e["kernel_ms"], "one_gpu_shard_call_ms": e["call_ms"],
            "source": "profiles/%s/shard_floor.json (tools/shard_floor.py: rows 0::%d on one GPU)" % (rnd, n_gpus)}


def host_to_host(P, M, reg, prec, reps=10):
    """numpy in -> numpy out through the host-buffer entry point (H2D of P and M, all kernels, D2H of the matrix):
    SURVEY.md 8(d)'s definition of t.  PCIe-inclusive; never `value`."""
    from pilot_amd import engine
    N = P.shape[0]
    for _ in range(3):
        engine.sinkhorn_grid(P, M, reg, precision=prec)
    t = time.perf_counter()
    for _ in range(reps):
        engine.sinkhorn_grid(P, M, reg, precision=prec)
    dt = (time.perf_counter() - t) / reps
    return {"value": round(N * N / dt, 1), "unit": "pairs/s", "ms_per_call": round(1e3 * dt, 4),
            "what": "engine.sinkhorn_grid(P, M): host numpy arrays in, host numpy matrix out, mean of %d calls" % reps}



def precision_ladder(P, M, reg, config, steps=10, row_step=60):
    """The same step at every precision the ABI offers: per rung the step and kernel time, pairs/s, the
    fraction of the pipe it executes on (make_roofline), mean updates per pair and max|gpu - oracle| on rows 0, 60, ..
    (6 000 of the 360 000 pairs at c3) against the fp64 CPU oracle.  `fp32` is IEEE f32 FMA chains on the f32-input MFMA,
    `bf16x3` exact 3-way splits, `f16x2` the default (22-bit products), `fp64` follows POT update for update."""
    from oracle import oracle as O
    from pilot_amd import _lib, engine
    N, K = P.shape
    Eo, io = O.sinkhorn_grid(P, M, reg, row_step=row_step, n_threads=host_cores(), return_info=True)
    last_o = (io["flags"] & O.FLAG_ABSORB_ON_LAST) > 0
    rungs = []
    plan = engine.DevicePlan(P, M)
    plan.enable_timing(True)
    for prec in ("fp32", "bf16x3", "f16x2", "fp64"):
        try:
            t = time.perf_counter()
            while time.perf_counter() - t < 0.25:         # clocks back up after the oracle leg
                plan.run(reg, precision=prec)
                plan.sync()
            t = time.perf_counter()
            for _ in range(steps):
                plan.run(reg, precision=prec)
            plan.sync()
            dt = (time.perf_counter() - t) / steps
        except Exception as e:      # a precision the shape does not support (e.g. f16x2 beyond its range) is reported, not fatal
            rungs.append({"precision": prec, "error": str(e)[:200]})
            continue
        main_ms, track = plan.kernel_times_ms(max_n=steps)
        kern_ms, track_ms = float(np.mean(main_ms)), float(np.mean(track))
        if track_ms > kern_ms:
            kern_ms, track_ms = track_ms, kern_ms
        E, info = plan.fetch()
        rf, _ = make_roofline(prec, K, info["iters"], kern_ms, track_ms)
        attach_offline_profile(rf, config, reg, prec)
        skip = last_o | ((info["flags"][::row_step] & _lib.FLAG_ABSORB_LAST) > 0)
        d = np.abs(E[::row_step] - Eo)
        rungs.append({
            "precision": prec, "dtype": DTYPE_NOTE[prec], "ms_per_step": round(1e3 * dt, 4), "pairs_per_s": round(N * N / dt, 1),
            "kernel_ms": round(kern_ms, 4), "pipe": rf["pipe"], "achieved_tflops": rf["achieved"], "peak_tflops": rf["peak"],
            "frac": rf["frac"], "useful_frac": rf.get("useful_frac", rf["frac"]),
            "f32_equivalent_ratio": rf.get("f32_equivalent", {}).get("ratio"),
            "mfma_busy": rf.get("mfma_busy"), "valu_busy": rf.get("valu_busy"), "coexec": rf.get("coexec"),
            "mean_updates_per_pair": round(float(info["iters"].mean()), 2),
            "sample_max_abs_diff_vs_oracle": float(d[~skip].max()), "sample_pairs": int((~skip).sum()),
            "sample_same_update_count": int((info["iters"][::row_step] == io["iters"]).sum())})
    plan.close()
    return {"workload": "%s, reg %g" % (config, reg), "oracle": "oracle/pilot_oracle.c (fp64), rows 0,%d,.. x all columns" % row_step,
            "oracle_mean_updates_per_pair": round(float(io["iters"].mean()), 2), "rungs": rungs}


def sweep_roofline(key, ms):
    """Matrix-pipe roofline of a sweep row: the dominant kernel of the workload from the committed rocprofv3 PMC passes
    (profiles/rNN/sweep_rooflines.json, tools/sweep_pmc.sh) -- MFMA instructions per launch, the share of the launch's SIMD cycles
    the matrix pipe / the vector unit were busy, and `frac` = matrix-pipe busy share x the kernel's share of the step."""
    path, rnd = profile_file("sweep_rooflines.json")
    if not path:
        return None
    try:
        with open(path) as fh:
            e = json.load(fh).get(key)
    except (OSError, ValueError):
        return None
    if not e:
        return None
    return {"bound": "mfma", "kernel": e["kernel"], "pipe": "16-bit matrix pipe (v_mfma_f32_16x16x32 / 16x16x16, f16 or bf16 pieces)",
            "mfma_per_launch": e["sq_insts_mfma"], "valu_per_launch": e["sq_insts_valu"], "mfma_busy": e["mfma_busy"], "valu_busy": e["valu_busy"],
            "coexec": e["coexec"], "kernel_ms_rocprofv3": round(e["kernel_us_profiled"] / 1e3, 4), "frac": e["mfma_busy"],
            "frac_is": "share of the launch's SIMD cycles with the matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES): issue utilisation of the pipe that "
                       "bounds the kernel, padding and piece products included",
            "source": "profiles/%s/sweep_rooflines.json (rocprofv3 --pmc, git %s)" % (rnd, e.get("git", "?"))}


def reg_sweep(P, M, K, regs=(0.01, 0.1, 1.0), row_step=60):
    """BASELINE config 3 names the sweep reg 0.01 / 0.1 / 1.0: per reg the step time with precision='auto', what AUTO
    picked, and parity against the oracle on rows 0, 60, 120, ... (10 rows x 600 columns)."""
    from oracle import oracle as O
    from pilot_amd import _lib, engine
    N = P.shape[0]
    rows = []
    plan = engine.DevicePlan(P, M)
    for reg in regs:
        t = time.perf_counter()                 # the oracle leg below leaves the GPU idle: warm the clocks back up
        while time.perf_counter() - t < 0.3:
            plan.run(reg, precision="auto")
            plan.sync()
        reps = 3 if reg < 0.05 else 20
        t = time.perf_counter()
        for _ in range(reps):
            plan.run(reg, precision="auto")
        plan.sync()
        dt = (time.perf_counter() - t) / reps
        E, info = plan.fetch()
        tc = time.perf_counter()
        Eo, io = O.sinkhorn_grid(P, M, reg, row_step=row_step, n_threads=host_cores(), return_info=True)
        dt_cpu = time.perf_counter() - tc
        tc = time.perf_counter()
        O.sinkhorn_grid(P, M, reg, row_begin=0, row_end=1, n_threads=1)       # one row on one thread, like the reference's loop
        dt_cpu1 = time.perf_counter() - tc
        Es, its, fl = E[::row_step], info["iters"][::row_step], info["flags"][::row_step]
        last = ((io["flags"] & O.FLAG_ABSORB_ON_LAST) > 0) | ((fl & _lib.FLAG_ABSORB_LAST) > 0)
        same = its == io["iters"]
        d = np.abs(Es - Eo)
        rows.append({
            "reg": reg, "ms_per_matrix": round(1e3 * dt, 4), "pairs_per_s": round(N * N / dt, 1),
            "pairs_f64": int(((info["flags"] & _lib.FLAG_F64) > 0).sum()), "pairs_total": int(E.size),
            "mean_updates_gpu": round(float(info["iters"].mean()), 2),
            "sample_pairs": int(Eo.size),
            "sample_max_abs_diff": float(d[~last].max()),
            "sample_max_abs_diff_same_update_count": float(d[same & ~last].max()) if (same & ~last).any() else None,
            "sample_same_update_count": int(same.sum()),
            "sample_capped_oracle": int((io["iters"] >= 1000).sum()), "sample_capped_gpu": int((its >= 1000).sum()),
            "sample_absorb_on_last_oracle": int(((io["flags"] & O.FLAG_ABSORB_ON_LAST) > 0).sum()),
            "sample_absorb_on_last_gpu": int(((fl & _lib.FLAG_ABSORB_LAST) > 0).sum()),
            "sample_mean_updates_oracle": round(float(io["iters"].mean()), 2),
            "cpu_pairs_per_s": round(Eo.size / dt_cpu, 1), "cpu_cores": host_cores(),
            "cpu_pairs_per_s_one_thread": round(N / dt_cpu1, 1),
            "cpu_what": "the fp64 oracle (POT's rule, kind \"port\") on the same sample, OpenMP over pairs on cpu_cores; one thread: row 0 (%d pairs)" % N,
            "roofline": sweep_roofline("c3|%g" % reg, 1e3 * dt),
        })
    plan.close()
    return {"precision": "auto", "sample": "rows 0,%d,.. x all columns" % row_step, "rows": rows}


def emd_instruction_accounting(config):
    """Per-pair instruction counts and HBM traffic of the exact-OT kernel from the committed rocprofv3 PMC passes of this
    workload (profiles/rNN/emd_instr.json, written by tools/make_emd_instr_json.py from tools/profile_pmc*.sh summaries)."""
    path, rnd = profile_file("emd_instr.json")
    if not path:
        return None
    try:
        with open(path) as fh:
            e = json.load(fh).get(config)
    except (OSError, ValueError):
        return None
    if e:
        e = dict(e, source="profiles/%s/emd_instr.json" % rnd)
    return e


def time_emd_grid(L, P, M, reps=5, ramp_s=0.2):
    """(seconds per matrix, E, n_aug) of pilot_ot_emd_grid_dev on a resident cohort: symmetric cost -> j >= i solved, mirrored."""
    from pilot_amd import _lib, engine
    N = P.shape[0]
    plan = engine.DevicePlan(P, M)
    sym = 2 if plan.sym else 0             # PILOT_OT_EMD_MIRROR for a symmetric cost
    def run():
        _lib.check(L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, sym, 0, N, 1, plan.dE, plan.dIt, None))
    t = time.perf_counter()
    while time.perf_counter() - t < ramp_s:
        run(); plan.sync()
    t = time.perf_counter()
    for _ in range(reps):
        run()
    plan.sync()
    dt = (time.perf_counter() - t) / reps
    E = np.empty((N, N)); n_aug = np.empty((N, N), dtype=np.int32)
    _lib.check(L.pilot_ot_memcpy_d2h(E.ctypes.data, plan.dE, 8 * N * N))
    _lib.check(L.pilot_ot_memcpy_d2h(n_aug.ctypes.data, plan.dIt, 4 * N * N))
    plan.close()
    return dt, E, n_aug


def emd_cpu_legs(P, M, E, budget_s):
    """Network simplex (oracle/pilot_oracle.c::pilot_oracle_emd2_ns, kind "port": the algorithm family of POT's LEMON solver)
    on one thread over a bounded row sample, and on every core over the whole grid when that fits ~10 s (then also the parity
    check of EVERY pair the launch produced)."""
    from oracle import oracle as O
    N = P.shape[0]
    t = time.perf_counter()
    O.emd_grid(P, M, row_begin=0, row_end=1, n_threads=1, fast="ns")
    per_row = time.perf_counter() - t
    n_rows = int(max(1, min(N, budget_s / max(per_row, 1e-9))))
    step = max(1, N // n_rows)
    t = time.perf_counter()
    Eo = O.emd_grid(P, M, row_step=step, n_threads=1, fast="ns")
    dt1 = time.perf_counter() - t
    one = {"value": round(Eo.size / dt1, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": "rows 0,%d,.. x all columns (%d ordered pairs), one thread: a NETWORK SIMPLEX on the bipartite transportation graph "
                     "(spanning-tree basis, block-search pricing), the algorithm family of POT's own solver (LEMON; not available on "
                     "this box); max|gpu-oracle| = %.2e" % (step, Eo.size, float(np.abs(E[::step] - Eo).max()))}
    ncpu = host_cores()
    step2 = 1 if dt1 / Eo.size * N * N / ncpu <= 10.0 else max(1, step // 8)
    t = time.perf_counter()
    Eo2 = O.emd_grid(P, M, row_step=step2, n_threads=ncpu, fast="ns")
    dt2 = time.perf_counter() - t
    err2 = float(np.abs(E[::step2] - Eo2).max())
    allc = {"value": round(Eo2.size / dt2, 1), "unit": "pairs/s", "cores": ncpu, "kind": "port",
            "sample": "%d ordered pairs%s, network simplex, OpenMP over pairs; max|gpu-oracle| over them = %.2e"
                      % (Eo2.size, " = EVERY pair of the launch" if step2 == 1 else "", err2)}
    parity = {"pairs": int(Eo2.size), "whole_grid": step2 == 1, "max_abs_diff_vs_fp64_oracle": err2, "tolerance": 1e-12,
              "what": "the launch's output against the oracle's network simplex (a different algorithm from the kernel's; the LP value "
                      "is unique) on the pairs of cpu_baseline_all_cores"}
    return one, allc, parity


def real_cohort_record(L, with_cpu):
    """The reference test's own cohort (test/test_pilot.py:9-23: Kidney_IgAN_G, 634 patients x 14 clusters) from the committed
    fixture: proportions and cost as the reference's code produced them; both modes of the pair grid, device-resident."""
    from pilot_amd import engine
    path = os.path.join(ROOT, "tests", "golden", "kidney_igan_g_634x14x14.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path, allow_pickle=True)
    P = np.ascontiguousarray(g["proportions"], dtype=np.float64)
    M = np.ascontiguousarray(g["cost"] / g["cost"].max(), dtype=np.float64)
    N, K = P.shape
    dt, E, n_aug = time_emd_grid(L, P, M, reps=10)
    rec = {"cohort": "Kidney_IgAN_G (tests/golden/kidney_igan_g_634x14x14.npz): %d patients x %d clusters" % (N, K),
           "exact_emd_ms_per_matrix": round(1e3 * dt, 3), "exact_emd_pairs_per_s": round(N * N / dt, 1),
           "mean_augmentations_per_solved_pair": round(float(n_aug[np.triu_indices(N)].mean()), 2)}
    plan = engine.DevicePlan(P, M)
    for _ in range(20):
        plan.run(0.1, precision="auto")
    plan.sync()
    t = time.perf_counter()
    for _ in range(20):
        plan.run(0.1, precision="auto")
    plan.sync()
    rec["sinkhorn_reg0.1_ms_per_matrix"] = round(1e3 * (time.perf_counter() - t) / 20, 3)
    plan.close()
    if with_cpu:
        one, allc, parity = emd_cpu_legs(P, M, E, 2.0)
        rec["exact_emd_cpu_baseline"], rec["exact_emd_cpu_baseline_all_cores"], rec["exact_emd_parity"] = one, allc, parity
    return rec


def exact_emd_record(L, P, M, config, with_cpu=True, budget_s=4.0):
    """The reference's DEFAULT mode (regularized='unreg', Trajectory.py:507-511) on the same cohort, device-resident: time,
    CPU baselines (network simplex, one thread and all cores), whole-grid parity, the committed instruction accounting of
    the kernel, and the reference test's own cohort."""
    N, K = P.shape
    dt, E, n_aug = time_emd_grid(L, P, M)
    iu = np.triu_indices(N)
    out = {"ms_per_matrix": round(1e3 * dt, 3), "pairs_per_s": round(N * N / dt, 1), "dtype": "f64",
           "what": "pilot_ot_emd_grid_dev, all N^2 ordered pairs (symmetric cost: j >= i solved, mirrored); host-timed over 5 back-to-back launches",
           "mean_augmentations_per_solved_pair": round(float(n_aug[iu].mean()), 2),
           "roofline": {"bound": "hbm", "achieved": round(N * N * (2 * K * 8 + 8) / dt / 1e9, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(N * N * (2 * K * 8 + 8) / dt / 1e9 / PEAK_HBM_GBS, 6),
                        "note": "instruction-issue-bound augmenting-path search (no MFMA, branchy fp64): the streaming-model bytes of "
                                "SURVEY 8(d) are what the north star names; what bounds it is in `instructions`"}}
    acc = emd_instruction_accounting(config)
    if acc:
        out["instructions"] = acc
    if with_cpu:
        out["cpu_baseline"], out["cpu_baseline_all_cores"], out["parity"] = emd_cpu_legs(P, M, E, budget_s)
    rc = real_cohort_record(L, with_cpu)
    if rc:
        out["real_cohort"] = rc
    return out


def bench_c4(L, rank, world, comm, args, single_process_multi, steps=3):
    """BASELINE configs[3]: 2000 patients x 100 cell types, reg 0.1 -- 55 ms of pair grid on one GPU, the shape whose row
    shards stay long enough to scale (c3's 1.5 ms step is bounded by its slowest pairs' serial chains)."""
    from pilot_amd import _lib, engine, multi, sharding
    from pilot_amd.synthetic import CONFIGS, make_problem
    P, M = make_problem(**CONFIGS["c4"])
    N = P.shape[0]
    if single_process_multi:
        devices = [0] * args.gpus if args.logical_shards else list(range(args.gpus))
        with multi.stdout_to_stderr():
            mp = multi.MultiPlan(P, M, devices=devices)
        mp.sinkhorn(0.1, precision="auto"); mp.sync()
        t = time.perf_counter()
        for _ in range(steps):
            mp.sinkhorn(0.1, precision="auto")
        mp.sync()
        dt = (time.perf_counter() - t) / steps
        g, ga = mp.times_ms()
        mp.close()
        return {"workload": "c4: 2000 x 100, reg 0.1, precision auto", "ms_per_step": round(1e3 * dt, 3), "pairs_per_s": round(N * N / dt, 1),
                "grid_ms_per_shard": [round(float(x), 3) for x in g], "gather_ms": round(ga, 3),
                "predicted_floor": shard_floor("c4", args.gpus)}
    rb, re_, rs = sharding.shard_rows(N, rank, world)
    n_pad = sharding.n_padded_rows(N, world)
    plan = engine.DevicePlan(P, M, n_rows_max=n_pad)
    plan.enable_timing(True)
    if comm:
        d_stage, d_full = DevBuf(L, 8 * world * n_pad * N), DevBuf(L, 8 * N * N)
        zeros = np.zeros(n_pad * N)
        _lib.check(L.pilot_ot_memcpy_h2d(plan.dE, zeros.ctypes.data, 8 * n_pad * N))

    def step():
        plan.run(0.1, row_begin=rb, row_end=re_, row_step=rs, precision="auto")
        if comm:
            comm.all_gather_rows(plan.dE, n_pad, N, d_stage.p, d_full.p)

    def fence():
        if comm:
            comm.barrier()
        plan.sync()
    step(); fence()
    t = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = (time.perf_counter() - t) / steps
    if comm:
        dt = comm.all_reduce_max(dt)
    main_ms, _ = plan.kernel_times_ms(max_n=steps)
    res = {"workload": "c4: 2000 x 100, reg 0.1, precision auto", "ms_per_step": round(1e3 * dt, 3), "pairs_per_s": round(N * N / dt, 1),
           "rank0_kernel_ms": round(float(np.mean(main_ms)), 3)}
    if world == 1:
        _, info = plan.fetch(n_rows=N)
        fl = algorithmic_flops(info["iters"], P.shape[1])
        res["roofline_frac_f32_mfma"] = round(fl / (float(np.mean(main_ms)) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
        res["roofline"] = sweep_roofline("c4|0.1", 1e3 * dt)
    if comm:
        res["max_kernel_ms_over_ranks"] = round(comm.all_reduce_max(float(np.mean(main_ms))), 3)
        fence()
        tg = time.perf_counter()
        for _ in range(5):
            comm.all_gather_rows(plan.dE, n_pad, N, d_stage.p, d_full.p)
        fence()
        res["gather_ms"] = round(comm.all_reduce_max(2e2 * (time.perf_counter() - tg)), 3)
        res["predicted_floor"] = shard_floor("c4", world)
    plan.close()
    if comm:
        d_stage.free(); d_full.free()
    return res


def cpu_baseline(P, M, reg, budget_s, E_gpu, iters_gpu):
    """Time the CPU oracle on a bounded sample of the SAME workload (rows 0, s, 2s, ... x all columns);
    also the checker: the sampled rows must agree with what the GPU produced.  Three legs: POT's rule on one thread (the
    reference is a single-threaded loop), the same rows with the f32 kernels' stop floor (equal update counts: what the GPU
    actually executes), and OpenMP over pairs on the cores this process may use."""
    from oracle import oracle as O
    N = P.shape[0]
    t = time.perf_counter()
    O.sinkhorn_grid(P, M, reg, row_begin=0, row_end=1)              # one row: per-pair cost estimate
    per_pair = (time.perf_counter() - t) / N
    n_rows = int(max(1, min(N, budget_s / max(per_pair * N, 1e-9))))
    step = max(1, N // n_rows)
    t = time.perf_counter()
    Eo, io = O.sinkhorn_grid(P, M, reg, row_step=step, n_threads=1, return_info=True)
    dt1 = time.perf_counter() - t
    err = float(np.abs(E_gpu[::step] - Eo).max())
    base = {"value": round(Eo.size / dt1, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "rows 0,%d,2*%d,.. (%d rows x %d columns = %d ordered pairs) of the same workload, "
                      "single thread like the reference's Python loop; max|gpu-oracle| on the sample = %.2e"
                      % (step, step, Eo.shape[0], N, Eo.size, err)}
    # equal work: the oracle with the stop floor of the f32 kernels (8 f32 ulps of ||b||_2 instead of POT's 1e-9, which f32
    # cannot resolve), so that both sides run (nearly) the same number of updates per pair
    t = time.perf_counter()
    Ee, ie = O.sinkhorn_grid(P, M, reg, row_step=step, n_threads=1, return_info=True, stop_floor_ulps=8.0)
    dte = time.perf_counter() - t
    eq = {"value": round(Ee.size / dte, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
          "mean_updates_per_pair": round(float(ie["iters"].mean()), 2),
          "gpu_mean_updates_per_pair_same_sample": round(float(iters_gpu[::step].mean()), 2),
          "pairs_with_the_gpu_update_count": int((ie["iters"] == iters_gpu[::step]).sum()), "pairs": int(Ee.size),
          "sample": "the same %d pairs, one thread, stopThr floored at 8 f32 ulps of ||b||_2 like the f32 kernels (NOT POT's rule: "
                    "an equal-work comparison only); max|gpu - this| = %.2e" % (Ee.size, float(np.abs(E_gpu[::step] - Ee).max()))}
    ncpu = host_cores()
    # all cores: the WHOLE grid when that is ~10 s of host time (c3: 3 s on 16 cores) -- then this leg is also the parity
    # check of every pair the step produced -- else four times the single-thread sample
    step2 = 1 if dt1 / Eo.size * N * N / ncpu <= 10.0 else max(1, step // 4)
    t = time.perf_counter()
    Eo2 = O.sinkhorn_grid(P, M, reg, row_step=step2, n_threads=ncpu)
    dt2 = time.perf_counter() - t
    err2 = float(np.abs(E_gpu[::step2] - Eo2).max())
    allc = {"value": round(Eo2.size / dt2, 1), "unit": "pairs/s", "cores": ncpu, "kind": "port",
            "cores_source": "len(os.sched_getaffinity(0)) capped by the cgroup CPU quota; os.cpu_count() = %d" % (os.cpu_count() or 1),
            "speedup_over_one_thread": round(Eo2.size / dt2 / (Eo.size / dt1), 2),
            "max_abs_diff_vs_gpu": err2, "pairs": int(Eo2.size), "whole_grid": step2 == 1,
            "sample": "%d ordered pairs%s, OpenMP over pairs (one scratch block per thread); max|gpu-oracle| over them = %.2e"
                      % (Eo2.size, " = EVERY pair of the step" if step2 == 1 else "", err2)}
    mo, mg = float(io["iters"].mean()), float(iters_gpu[::step].mean())
    upd = {"oracle_mean_updates_per_pair": round(mo, 2), "gpu_mean_updates_per_pair_same_sample": round(mg, 2),
           "gpu_over_oracle_updates": round(mg / mo, 4),
           "note": "the f32 kernel floors POT's stopThr 1e-9 at 8 ulp * ||b||_2 (f32 cannot resolve 1e-9), so a pair stops at "
                   "the same or an earlier error check than the fp64 oracle: `value` (f32) and `cpu_baseline` (fp64, full "
                   "update count) are the same pairs but not the same number of updates; `cpu_baseline_equal_updates` is the "
                   "oracle with the same floor; roofline.achieved counts only the updates the GPU executed"}
    return base, eq, allc, upd


def bench_cellw2(args, reg=0.1, D=30):
    """--mode cellw2: BASELINE config 5 (an extension, not in the reference): every ordered pair of `--cell-patients`
    patients with `--cell-cells` cells each, log-domain Sinkhorn on raw cell clouds, cells resident in HBM."""
    from pilot_amd import engine
    from pilot_amd.synthetic import make_cell_clouds
    Np, nc = args.cell_patients, args.cell_cells
    X, offs, scale = make_cell_clouds(Np, nc, D, seed=6)
    co = engine.CellCohort(X, offs)
    co.w2_grid(scale, reg, row_begin=0, row_end=1, num_iter_max=3)         # warm-up (code objects, clocks)
    t = time.perf_counter()
    W, info = co.w2_grid(scale, reg, return_info=True)
    dt = time.perf_counter() - t
    kern_s = co.last_kernel_ms * 1e-3
    pieces = co.last_pieces
    terms = 3 if pieces == 2 else 6
    co.close()
    upd = info["iters"].astype(np.float64)
    flop = float((upd * 2 * 2.0 * nc * nc * D).sum() + 2.0 * nc * nc * D * upd.size)     # two dot-product passes per update + the value pass
    conv = info["iters"] < 1000
    Dp = 32 * ((D + 31) // 32)
    sym = float(np.abs(W - W.T)[conv & conv.T].max())
    # CPU baseline: the C / OpenMP fp64 oracle (kind "port": the extension has no reference implementation) on a few pairs of the
    # SAME cohort, all cores inside a pair; also the parity check of those pairs
    cpu, errs = None, []
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        ncpu = host_cores()
        pairs = [(0, 1), (2, 3), (5, 4)][:max(1, min(3, Np // 2))]
        t = time.perf_counter()
        vals = [O.cell_w2_c(X[offs[i]:offs[i + 1]], X[offs[j]:offs[j + 1]], scale, reg, n_threads=ncpu, return_info=True) for i, j in pairs]
        dtc = time.perf_counter() - t
        errs = [abs(v - W[i, j]) for (v, inf), (i, j) in zip(vals, pairs) if inf["iters"] < 1000]
        cpu = {"value": round(len(pairs) / dtc, 4), "unit": "pairs/s", "cores": ncpu, "kind": "port",
               "sample": "%d ordered pairs %s of the same cohort by oracle/pilot_oracle.c::pilot_oracle_cell_w2 (fp64, POT sinkhorn_log control "
                         "flow, OpenMP inside a pair on %d threads), %s updates; max|gpu-oracle| on its converged pairs = %s"
                         % (len(pairs), pairs, ncpu, [inf["iters"] for _, inf in vals], ("%.2e" % max(errs)) if errs else "n/a")}
    # parity at this config's own settings beyond the few pairs timed above: tools/cellw2_parity_c5.py (32 ordered pairs of the same
    # cohort against the fp64 oracle), committed under profiles/
    parity = {"live": None if cpu is None else {"pairs": len(errs), "max_abs_diff": max(errs) if errs else None}, "tolerance": 1e-5}
    ppath, prnd = profile_file("cellw2_parity_c5.json")
    if ppath and (Np, nc, D, reg) == (200, 5000, 30, 0.1):
        with open(ppath) as fh:
            pj = json.load(fh)
        parity["offline"] = {"pairs": pj["pairs"], "pairs_converged_in_oracle": pj["pairs_converged_in_oracle"], "max_abs_diff": pj["max_abs_diff"],
                             "median_abs_diff": pj["median_abs_diff"], "gpu_stops_at_oracle_check_or_earlier": pj["gpu_stops_at_oracle_check_or_earlier"],
                             "source": "profiles/%s/cellw2_parity_c5.json (tools/cellw2_parity_c5.py)" % prnd}
    traffic, traffic_src = None, None
    tpath, trnd = profile_file("cellw2_traffic.json")
    if tpath:
        try:
            with open(tpath) as fh:
                e = json.load(fh).get("%dx%dx%d" % (Np, nc, D))
            if e:
                traffic, traffic_src = e["traffic_bytes"], "profiles/%s/cellw2_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this launch, git %s)" % (trnd, e.get("git", "?"))
        except (OSError, ValueError, KeyError):
            pass
    return {
        "metric": "cell-level W2 patient-pairs/sec (full NxN matrix; extension, BASELINE config 5)", "value": round(Np * Np / dt, 2),
        "unit": "pairs/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": round(1e3 * dt, 1), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": ("f32 potentials; dot products from 2 fp16 pieces per coordinate (22 significant bits) on the f16 MFMA" if pieces == 2 else
                  "f32 potentials; dot products as exact 3-way bf16 splits of the coordinates on the bf16 MFMA"), "data": "synthetic",
        "config": {"workload": "c5: %d patients x %d cells x %d dims, entropic W2 reg=%g (POT sinkhorn_log control flow), all N^2 "
                               "ordered pairs" % (Np, nc, D, reg), "n_patients": Np, "cells_per_patient": nc, "n_dims": D, "reg": reg},
        "roofline": {"bound": "mfma", "kernel": "pilot::cell_w2_kernel<1, true, %s>" % ("true" if pieces == 2 else "false"),
                     "achieved": round(flop * terms * Dp / D / kern_s / 1e12, 1),
                     "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(flop * terms * Dp / D / kern_s / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                     "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": round(co.last_kernel_ms, 1),
                     "note": "achieved = 16-bit MFMA flop executed: %d piece products per term (%s), "
                             "D padded to %d; the algorithmic (f32-equivalent) dot-product rate is 2 n_p n_q D per pass = %.1f TFLOP/s "
                             "(the f32-input MFMA peak is %.1f)" % (terms, "2 fp16 pieces of both operands" if pieces == 2 else
                                                                    "exact 3-way bf16 splits of both operands", Dp, flop / kern_s / 1e12,
                                                                    PEAK_F32_MFMA_TFLOPS),
                     "dot_tflops_f32_equivalent": round(flop / kern_s / 1e12, 2),
                     "mean_updates_per_pair": round(float(upd.mean()), 2)},
        "checks": {"pairs_converged": int(conv.sum()), "pairs": int(conv.size), "max_asymmetry_of_converged_pairs": sym},
        "cpu_baseline": cpu, "parity": parity,
    }


def bench_emd(args, L, P, M, cfg):
    """--mode emd: the exact-OT pair grid (reference default), its own line with its own cpu_baseline."""
    from oracle import oracle as O
    from pilot_amd import _lib, engine
    N, K = P.shape
    plan = engine.DevicePlan(P, M)
    mode = 2 if plan.sym else 0

    def run():
        _lib.check(L.pilot_ot_emd_grid_dev(plan.plan, plan.dP, plan.dM, mode, 0, N, 1, plan.dE, plan.dIt, None))
    for _ in range(args.warmup):
        run()
    plan.sync()
    t = time.perf_counter()
    for _ in range(args.steps):
        run()
    plan.sync()
    dt = (time.perf_counter() - t) / args.steps
    E = np.empty((N, N)); n_aug = np.empty((N, N), dtype=np.int32)
    _lib.check(L.pilot_ot_memcpy_d2h(E.ctypes.data, plan.dE, 8 * N * N))
    _lib.check(L.pilot_ot_memcpy_d2h(n_aug.ctypes.data, plan.dIt, 4 * N * N))
    plan.close()
    out = {
        "metric": "exact-EMD patient-pairs/sec (full NxN matrix, reference default mode)", "value": round(N * N / dt, 1),
        "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d patients x %d cell types, exact OT (ot.emd2 semantics), all N^2 ordered pairs"
                               % (args.config, N, K), "n_patients": N, "n_cell_types": K},
        "roofline": {"bound": "hbm", "achieved": round(N * N * (2 * K * 8 + 8) / dt / 1e9, 3), "peak": PEAK_HBM_GBS,
                     "unit": "GB/s", "frac": round(N * N * (2 * K * 8 + 8) / dt / 1e9 / PEAK_HBM_GBS, 6), "traffic": None,
                     "note": "latency / instruction-issue-bound augmenting-path search, one wavefront per pair; "
                             "the streaming-model bytes are what the north star names",
                     "mean_augmentations_per_solved_pair": round(float(n_aug[np.triu_indices(N)].mean()), 2)},
    }
    acc = emd_instruction_accounting(args.config)
    if acc:
        out["instructions"] = acc
    if not args.no_cpu_baseline:
        ncpu = host_cores()
        step = max(1, N // 4)
        t = time.perf_counter()
        Eo = O.emd_grid(P, M, row_step=step, n_threads=1, fast="ns")
        dt1 = time.perf_counter() - t
        out["cpu_baseline"] = {"value": round(Eo.size / dt1, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": "rows 0,%d,.. x all columns (%d pairs), one thread: a NETWORK SIMPLEX on the bipartite transportation "
                                         "graph (oracle/pilot_oracle.c::pilot_oracle_emd2_ns: spanning-tree basis, block-search pricing), the "
                                         "algorithm family of POT's own solver (LEMON network simplex; not available on this box); "
                                         "max|gpu-oracle| = %.2e" % (step, Eo.size, float(np.abs(E[::step] - Eo).max()))}
        t = time.perf_counter()
        Es = O.emd_grid(P, M, row_step=step, n_threads=1, fast=True)
        dts = time.perf_counter() - t
        out["cpu_baseline_ssp"] = {"value": round(Es.size / dts, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": "the same pairs by the HIP kernel's own algorithm (successive shortest paths, diagonal warm start) on "
                                             "one CPU thread; max|gpu-oracle| = %.2e" % float(np.abs(E[::step] - Es).max())}
        # all cores: the whole grid when that is ~10 s of host time (then also the parity check of EVERY pair), else 8 x the sample
        step2 = 1 if dt1 / Eo.size * N * N / ncpu <= 10.0 else max(1, step // 8)
        t = time.perf_counter()
        Eo2 = O.emd_grid(P, M, row_step=step2, n_threads=ncpu, fast="ns")
        dt2 = time.perf_counter() - t
        err2 = float(np.abs(E[::step2] - Eo2).max())
        out["cpu_baseline_all_cores"] = {"value": round(Eo2.size / dt2, 1), "unit": "pairs/s", "cores": ncpu, "kind": "port",
                                         "sample": "%d pairs%s, network simplex, OpenMP over pairs; max|gpu-oracle| over them = %.2e"
                                                   % (Eo2.size, " = EVERY pair of the step" if step2 == 1 else "", err2)}
        out["parity"] = {"pairs": int(Eo2.size), "whole_grid": step2 == 1, "max_abs_diff_vs_fp64_oracle": err2, "tolerance": 1e-12,
                         "what": "the step's output against the oracle's network simplex (a different algorithm from the kernel's; the LP "
                                 "value is unique) on the pairs of cpu_baseline_all_cores"}
    return out


if __name__ == "__main__":
    main()
