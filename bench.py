#!/usr/bin/env python
"""bench.py — E-step loglik-evals/sec of the MI355X-native SMC++ engine (BASELINE.json metric).

One "step" = one eval of SURVEY.md §8(d): parameters marked dirty -> E-step (host eigensystem prep + upload +
forward/backward chains + sufficient statistics on the GPU) -> loglik, over this rank's contig(s), with the
observation arrays already resident in HBM (they are uploaded when the inference manager is constructed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|c2|c3|c4|c5] [--no-cpu] [--chunk ROWS]

N > 1: launched by torch.distributed.run, one rank per GPU; every rank owns one synthetic 100 Mbp contig (weak
scaling, contigs are independent HMMs) and the ranks exchange ONE all-reduce(sum, fp64) of the packed
[loglik | gamma0 | xisum | gamma_sums] statistics per step over RCCL.

Prints one JSON line with the contract fields plus "roofline" (dominant kernel, timed live with HIP events on the
engine's own stream) and "cpu_baseline" (the compiled reference `oracle/_ref` — or the C restatement `oracle/` if
that is absent — timed on this box's host cores on a bounded prefix of the same contig).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (M, n, params fixture, description)
    "headline": (64, 20, "params_M64_n20.npz", "1 synthetic 100 Mbp contig per GPU, M=64, n=20 (BASELINE.json metric shape)"),
    "c2": (32, 10, "params_M32_n10.npz", "1 synthetic 100 Mbp contig per GPU, M=32, n=10 (configs[1])"),
    "c5": (256, 50, "params_M256_n50.npz", "1 synthetic 100 Mbp contig per GPU, M=256, n=50 (configs[4])"),
    # whole genome: the 22 autosome-like contigs (2 872 Mbp) sharded longest-first over the ranks; STRONG scaling
    "c3": (64, 20, "params_M64_n20.npz", "22 synthetic contigs, 2872 Mbp in total, M=64, n=20, LPT-sharded over the GPUs (configs[2])"),
    # two populations, both distinguished lineages in population 1, split 0.5; the parameters come
    # from the engine's own JointCSFS preparation, computed once outside the timed region like the fixtures above
    "c4": (48, 10, None, "1 synthetic two-population 100 Mbp contig per GPU, M=48, n1=n2=10, a=(2,0), split=0.5 (configs[3])"),
}
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector = matrix peak (AMD spec)


def algorithmic_work(obs, M):
    """Algorithmic flops / bytes of ONE pass of the dominant (forward chain) kernel, DESIGN.md §Kernels:
    span-1 row: 2 M^2 (one mat-vec); span>1 row: 4 M^2 (two mat-vecs).  Bytes: alpha write 4M + row descriptor 8 +
    normaliser 8 (emission / eigenvalue-power vectors come from LDS-resident tables)."""
    R1 = int((obs[:, 0] == 1).sum())
    Re = len(obs) - R1
    flops = 2.0 * M * M * R1 + 4.0 * M * M * Re
    nbytes = len(obs) * (4.0 * M + 16.0)          # alpha row (float) + normaliser + descriptor
    return flops, nbytes, R1, Re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--length-mbp", type=float, default=100.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--eps-alpha", type=float, default=0.0)
    ap.add_argument("--eps-beta", type=float, default=0.0)
    ap.add_argument("--warm", action="store_true",
                    help="additionally measure the opt-in warm start on a sequence of perturbed parameter sets "
                         "(reported as an extra object; the headline value is always the cold E-step)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # SMCPP_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 code path (key union, packed all-reduce, barrier,
    # max over ranks) be exercised on a box with fewer GPUs than ranks (ranks share devices, the reduction runs on the
    # host).  The measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("SMCPP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    from smcpp_amd import _smcpp, synth
    # torch.distributed.run exports OMP_NUM_THREADS=1 to every rank; the engine's host phase (one eigensystem per eigen
    # key, in parallel) wants a handful of threads, which is what the reference's --cores / set_num_threads is for
    _smcpp.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(1, world))))
    M, n, fixture, desc = WORKLOADS[args.workload]
    length_bp = int(args.length_mbp * 1e6)
    if args.workload == "c4":
        from smcpp_amd import _engine
        from smcpp_amd.model import PiecewiseModel, TwoPopulationModel
        obs = synth.synth_contig_twopop(rank, length_bp, n, n)
        hs = synth.hidden_states(M)
        a, s_ = synth.model_pieces()
        tm = TwoPopulationModel(PiecewiseModel(a, s_, 1e4, pid="pop1"),
                                PiecewiseModel(1.5 + 0.5 * np.cos(np.arange(8)), s_[:8], 1e4, pid="pop2"), 0.5)
        keys4 = np.unique(obs[:, 1:], axis=0).astype(np.int32)
        d, p1, p2 = tm.for_pop("pop1"), tm.for_pop("pop1"), tm.for_pop("pop2")
        pi4, T4, E4 = _engine.host_prep_twopop(n, n, 2, 0, hs, 0.5, (d.a, d.s), (p1.a, p1.s), (p2.a, p2.s), tm.split,
                                               synth.THETA, synth.RHO, synth.ALPHA, keys4)
        par = dict(pi=pi4, T=T4, keys=keys4, E=E4, hs=hs, pol=0.5, theta=synth.THETA, rho=synth.RHO, alpha=synth.ALPHA)
        im = _smcpp.PyTwoPopInferenceManager(n, n, 2, 0, [obs], hs, ("pop1", "pop2"), 0.5, device=local_rank)
    elif args.workload == "c3":
        from smcpp_amd import dist as sd
        par = np.load(os.path.join(ROOT, "tests", "golden", fixture))
        owner = sd.lpt_shard(synth.C3_LENGTHS_MBP, world)
        mine = [i for i in range(len(owner)) if owner[i] == rank]
        contigs = [synth.synth_contig(i, int(synth.C3_LENGTHS_MBP[i] * 1e6), n) for i in mine]
        obs = np.concatenate(contigs)                    # only for the algorithmic work / CPU baseline bookkeeping
        im = _smcpp.PyOnePopInferenceManager(n, contigs, par["hs"], ("pop1",), float(par["pol"]), device=local_rank)
    else:
        par = np.load(os.path.join(ROOT, "tests", "golden", fixture))
        obs = synth.synth_contig(rank, length_bp, n)         # contig index = rank: independent contigs, weak scaling
        im = _smcpp.PyOnePopInferenceManager(n, [obs], par["hs"], ("pop1",), float(par["pol"]), device=local_rank)
    im.theta = float(par["theta"]); im.rho = float(par["rho"]); im.alpha = float(par["alpha"])
    if args.chunk or args.eps_alpha or args.eps_beta:
        im.set_chunking(args.chunk, args.eps_alpha, args.eps_beta)
    if world > 1:
        # global key dictionary so the packed gamma_sums blocks line up across ranks
        from smcpp_amd import dist as sd
        im.set_global_keys(sd.union_keys(im.keys))
    pi, T, keys, E = par["pi"], par["T"], par["keys"], par["E"]

    stats_buf = None

    def one_eval():
        nonlocal stats_buf
        im.set_raw(pi, T, keys, E)          # parameters dirty: eigensystems, uploads, everything is redone
        im.E_step()
        if world > 1:
            if backend == "nccl":
                # the packed statistics are written by one kernel straight into the tensor RCCL reduces
                if stats_buf is None:
                    stats_buf = torch.empty(im.stats_len(), dtype=torch.float64, device=dev)
                im.pack_stats_device(stats_buf.data_ptr())
                dist.all_reduce(stats_buf, op=dist.ReduceOp.SUM)       # the single collective of the E-step
                ll_sum = float(stats_buf[0].item())                   # (synchronises the reduction)
                im.unpack_stats_device(stats_buf.data_ptr(), stats_buf.numel())
                return ll_sum
            h = im.pack_stats()
            if stats_buf is None:
                stats_buf = torch.empty(len(h), dtype=torch.float64, device=red_dev)
            stats_buf.copy_(torch.from_numpy(h))
            dist.all_reduce(stats_buf, op=dist.ReduceOp.SUM)
            im.unpack_stats(stats_buf.cpu().numpy())
            return float(stats_buf[0].item())
        return im.loglik()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_eval()
    barrier()
    timings = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ll = one_eval()
        timings.append(im.last_timing())
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    if args.workload == "c3":
        value = args.steps / elapsed                     # whole-genome evals per second (the ranks share ONE eval)
    else:
        value = world * args.steps / elapsed             # contig-E-step evals per second, whole job

    # ---- roofline of the dominant kernel (forward chain pass), timed live with HIP events ----
    # smcpp_last_timing brackets the forward / backward pass launches with hipEvents recorded on the engine's stream.
    fwd_ms = float(np.median([t["forward_ms"] for t in timings]))   # (overlaps the backward passes on a 2nd stream)
    bwd_ms = float(np.median([t["backward_ms"] for t in timings]))
    fpasses = float(np.median([t["fwd_passes"] for t in timings]))
    bpasses = float(np.median([t["bwd_passes"] for t in timings]))
    flops, nbytes, R1, Re = algorithmic_work(obs, M)
    launches = max(fpasses, 1.0)
    per_launch_s = 1e-3 * fwd_ms / launches
    ach_tflops = flops / per_launch_s / 1e12 if per_launch_s > 0 else 0.0
    ach_gbs = nbytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
    if ach_gbs / HBM_PEAK_GBS >= ach_tflops / FP64_PEAK_TFLOPS:
        roof = dict(bound="hbm", achieved=ach_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach_gbs / HBM_PEAK_GBS)
    else:
        roof = dict(bound="mfma", achieved=ach_tflops, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=ach_tflops / FP64_PEAK_TFLOPS)
    # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs of this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md).  Only valid for the workload
    # the profile was taken on.
    traffic = None
    try:
        if args.workload == "headline" and args.length_mbp == 100.0:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_g_hbm_traffic_pmc.json")))["kernels"]
            k = [v for name, v in prof.items() if "k_fwd_coop" in name][0]
            # a pass that re-runs every chunk (the max over launches; converged check passes write nothing)
            traffic = 1024.0 * (2.0 * k["FETCH_SIZE_KB_max"] + k["WRITE_SIZE_KB_max"])
    except Exception:  # noqa: BLE001
        traffic = None
    roof.update(traffic=traffic, kernel="k_fwd_coop (forward chain pass; runs concurrently with k_bwd_coop)", launches_per_step=launches,
                avg_launch_ms=1e3 * per_launch_s, algorithmic_flops_per_launch=flops,
                algorithmic_bytes_per_launch=nbytes)

    # ---- optional: warm start (smcpp_set_warm_start) on a parameter trajectory, the way an optimiser calls the path ----
    warm_obj = None
    if args.warm and world == 1:
        rng = np.random.default_rng(11)

        def perturbed(scale):
            # every emission vector and the transition matrix move by a relative `scale` (rows of T renormalised)
            Ep = np.clip(E * (1.0 + scale * rng.standard_normal(E.shape)), 1e-12, 1.0)
            Tp = T * (1.0 + scale * rng.standard_normal(T.shape))
            Tp *= (T.sum(axis=1) / Tp.sum(axis=1))[:, None]
            return Tp, Ep

        def sequence(warm):
            im.set_warm_start(warm)
            im.set_raw(pi, T, keys, E); im.E_step()            # iteration 0 (cold either way)
            ts, lls = [], []
            for it in range(args.steps):
                Tp, Ep = perturbed(1e-2 / (1 + it))               # steps shrink as an EM run converges
                t0 = time.perf_counter()
                im.set_raw(pi, Tp, keys, Ep); im.E_step(); lls.append(im.loglik())
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), lls, im.last_timing()

        rng = np.random.default_rng(11); t_cold, ll_cold, _ = sequence(False)
        rng = np.random.default_rng(11); t_warm, ll_warm, tw = sequence(True)
        im.set_warm_start(False)
        warm_obj = {"note": "same sequence of perturbed parameter sets (relative step 1e-2/(1+it)) evaluated cold and with "
                            "smcpp_set_warm_start; not part of `value`",
                    "cold_ms_per_eval": 1e3 * t_cold, "warm_ms_per_eval": 1e3 * t_warm,
                    "max_rel_loglik_diff": float(max(abs(a - b) / abs(a) for a, b in zip(ll_cold, ll_warm))),
                    "warm_fwd_passes": tw["fwd_passes"], "warm_bwd_passes": tw["bwd_passes"]}

    out = None
    if rank == 0:
        med = {k: float(np.median([t[k] for t in timings])) for k in timings[0]}
        out = {
            "metric": "E-step loglik-evals/sec (100 Mbp, M=64, n=20)" if args.workload == "headline"
            else ("whole-genome E-step loglik-evals/sec (22 contigs, 2872 Mbp, M=64, n=20)" if args.workload == "c3"
                  else f"E-step loglik-evals/sec ({args.length_mbp:g} Mbp, M={M}, n={n})"),
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.workload == "c3" else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "M": M, "n": n, "rows_per_contig": int(len(obs)),
                       "span1_rows": R1, "eigen_rows": Re,
                       "contigs_per_gpu": len(contigs) if args.workload == "c3" else 1,
                       "length_mbp": float(sum(synth.C3_LENGTHS_MBP)) if args.workload == "c3" else args.length_mbp,
                       "loglik": ll, "parallelism": f"contig-sharded x{world}, 1 all-reduce/E-step" if world > 1 else "single GPU"},
            "split_ms": med,
            "roofline": roof,
        }
        if warm_obj:
            out["warm_start"] = warm_obj
        if not args.no_cpu and world == 1:          # the CPU baseline is timed at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(par, obs, args.cpu_seconds, M)
            if out["cpu_baseline"] and out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_1core"] = (value / world) / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(par, obs, budget_s, M):
    """The reference's own C++ E-step (oracle/_ref, compiled from /root/reference/src in the build container) on
    one host core, on a prefix of the same contig sized to ~budget_s seconds; evals/s are scaled by row count
    (the reference's cost is linear in rows: one thread per contig, inference_manager.cpp:89-94)."""
    try:
        from oracle import ref
        kind = "reference" if ref.available() else "port"
        if kind == "port":
            from oracle import oracle as orc
        pi, T, keys, E = par["pi"], par["T"], par["keys"], par["E"]
        probe = min(len(obs), 2000)
        fn = ref.estep if kind == "reference" else orc.estep
        fn(pi, T, keys, E, obs[:probe])                      # first call: library load, page-in
        t = time.perf_counter()
        fn(pi, T, keys, E, obs[:probe])
        per_row = (time.perf_counter() - t) / probe
        rows = int(min(len(obs), max(probe, budget_s / per_row)))
        t = time.perf_counter()
        r = (ref.estep if kind == "reference" else orc.estep)(pi, T, keys, E, obs[:rows])
        dt = time.perf_counter() - t
        full = dt * len(obs) / rows
        return {"value": 1.0 / full, "unit": "evals/s", "cores": 1, "kind": kind,
                "sample": f"first {rows} of {len(obs)} rows of the same contig in {dt:.1f} s, scaled by row count "
                          f"(1 thread = 1 contig, as the reference parallelises); host has {os.cpu_count()} cores",
                "us_per_row": 1e6 * dt / rows, "loglik_prefix": r["loglik"]}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "error": repr(e)}


if __name__ == "__main__":
    main()
