#!/usr/bin/env python
"""bench.py — E-step loglik-evals/sec of the MI355X-native SMC++ engine (BASELINE.json metric).

One "step" = one eval exactly as SURVEY.md §8(d) defines it: `set_params(model)` (parameters dirty) -> `E_step()`
(cold preparation A6-A10: rate function, pi, transition, conditioned SFS, emission table; eigensystems; upload;
forward / backward chains and sufficient statistics on the GPU) -> `loglik()`, over this rank's contig(s), with the
observation arrays already resident in HBM (they are uploaded when the inference manager is constructed).  This is
what `InferenceManager::Estep` does after `setParams` (src/inference_manager.cpp:108-114,213-229,256-260).
`split_ms.hmm_only_ms` additionally reports the same eval with the prepared parameters handed over by `set_raw`
(no A6-A10), which is what round 1 timed.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|c2|c3|c4|c5|posterior|posterior64|posterior128|pbinned|qgrad|shaping] [--no-cpu]

`--gpus N` with N > 1 and no torchrun environment re-launches itself under `torch.distributed.run` (one rank per
GPU, RCCL); under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  Every rank owns one synthetic 100 Mbp contig
(weak scaling; `c3` shards the 22 contigs of a whole genome longest-first, strong scaling) and the ranks exchange ONE
all-reduce(sum, fp64) of the packed [loglik | gamma0 | xisum | gamma_sums] statistics per step
(`smcpp_amd.dist.ShardedInferenceManager`).  If the box has fewer GPUs than ranks the ranks share
devices and reduce over gloo (a functional test of the N > 1 path, flagged in the output; never a measurement).

Prints one JSON line with the contract fields plus "roofline" (dominant kernel, timed live with HIP events on the
stream it is launched on) and "cpu_baseline" (the compiled reference `oracle/_ref`: its own cold preparation +
`HMM::Estep` on a bounded prefix of the same contig, one host core).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
    # two populations, both distinguished lineages in population 1, split 0.5
    "c4": (48, 10, None, "1 synthetic two-population 100 Mbp contig per GPU, M=48, n1=n2=10, a=(2,0), split=0.5 (configs[3])"),
    # the M-step's objective: Q with its forward-mode gradient along 16 directions on the headline manager's
    # statistics; one step = set_params(a, da) -> Q(val, jac)  (what L-BFGS-B calls tens to hundreds of times per E-step)
    "qgrad": (64, 20, "params_M64_n20.npz", "Q with 16 derivative directions (HMM::Q on adouble, src/hmm.cpp:155-193) after one E-step on "
              "1 synthetic 100 Mbp contig, M=64, n=20"),
    # posterior decoding: un-binned rows, long spans, small rho, save_gamma
    "posterior": (32, 8, None, "posterior decode: 1 un-binned contig, 1e6 rows, spans to 1e5, rho=6e-5, M=32, n=8, save_gamma"),
    # ... the same at M = 64: the eigenvector tables of the hybrid rows (133 KB per eigen key) do not fit LDS, so the dense
    # cooperative chains run (the regime hole DESIGN.md section 9 names; measured, not hidden)
    "posterior64": (64, 8, None, "posterior decode: 1 un-binned contig, 1e6 rows, spans to 1e5, rho=6e-5, M=64, n=8, save_gamma"),
    # ... and at M = 128 (round 6: the dense streamed chains take a row in one eigen-power step; the per-row posteriors come from
    # eigen-power pieces of 64 positions walked by scan steps - k_piece_vectors / k_gamma_rows_scan<., true> - instead of the scalar
    # eigensystem kernel of rounds 1-5; SMCPP_GAMMA_PIECES=0 times that one)
    "posterior128": (128, 8, None, "posterior decode: 1 un-binned contig, 1e6 rows, spans to 1e5, rho=6e-5, M=128, n=8, save_gamma"),
    # posterior decode of BINNED data (round 6: save_gamma keeps the eigen-free path - per-row posteriors of the span > 1 rows from scan
    # steps, k_gamma_rows_scan): the headline contig with save_gamma; parity = the decoded index of every column against golden G19
    "pbinned": (64, 20, "params_M64_n20.npz", "posterior decode of 1 binned synthetic 100 Mbp contig, M=64, n=20, save_gamma"),
    # SURVEY.md 8 f-2 on the device: what data_filter.py does to a contig before an inference manager sees it (integer, HBM-bound)
    "shaping": (0, 8, None, "pre-HMM data shaping Thin(400) -> Bin(100) -> Compress of 1 un-binned contig, 1e6 rows (4.8e8 bp), on the device"),
}
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector = matrix peak (AMD spec)


def chain_pass_work(obs_list, M):
    """Algorithmic work of ONE sequential pass of a chain over the rows (what `HMM::Estep`'s forward or backward loop
    needs, src/hmm.cpp:61-96,102-149): span-1 row = one M x M mat-vec (2 M^2 flop), span>1 row = two (4 M^2, through
    the eigenbasis).  Bytes of the forward pass: alpha row written (4M, float) + normaliser (8) + descriptor read (8);
    of the backward pass: beta row written (8M) + descriptor (8)."""
    R1 = sum(int((o[:, 0] == 1).sum()) for o in obs_list)
    R = sum(len(o) for o in obs_list)
    Re = R - R1
    flops = 2.0 * M * M * R1 + 4.0 * M * M * Re
    return flops, R * (4.0 * M + 16.0), R * (8.0 * M + 8.0), R1, Re


def eval_work(obs_list, M):
    """F_alg = 2 M^3 Re + 12 M^2 R + 4 M^3 G + 25 M^3 Ke,  B_alg = R (24 M + 32) per eval."""
    F = B = 0.0
    for o in obs_list:
        R = len(o)
        e = o[o[:, 0] > 1]
        Re = len(e)
        G = len(np.unique(e, axis=0)) if Re else 0
        Ke = len(np.unique(e[:, 1:], axis=0)) if Re else 0
        F += 2.0 * M ** 3 * Re + 12.0 * M * M * R + 4.0 * M ** 3 * G + 25.0 * M ** 3 * Ke
        B += R * (24.0 * M + 32.0)
    return F, B


def synth_posterior_contig(rows, n, seed=7):
    from smcpp_amd import synth
    return synth.synth_posterior_contig(rows, n, seed)


def cgroup_cpu_stat():
    """nr_periods / nr_throttled / throttled_usec of this container's CPU controller (cgroup v2), or None: the engine's host phase
    keeps OpenMP workers spinning between parallel regions (two-population managers), and a pod whose quota is exceeded is
    throttled for whole scheduler periods - the bench line carries the delta over the timed region."""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return {k: int(d[k]) for k in ("nr_periods", "nr_throttled", "throttled_usec") if k in d}
    except Exception:  # noqa: BLE001
        return None


def gpu_power_state():
    """Performance level and current clocks of the GPUs this process can see, from sysfs (readable without privileges), or None:
    recorded beside the timing because an eval whose host phase leaves the device idle for a few hundred microseconds (two
    populations) depends on how the box's power management treats an idle queue."""
    import glob
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if not os.path.exists(dev + "/pp_dpm_sclk"):
                continue
            cur = lambda f: next((l.split(":")[1].replace("*", "").strip() for l in open(dev + "/" + f) if "*" in l), None)  # noqa: E731
            out.append({"card": dev.split("/")[4], "perf_level": open(dev + "/power_dpm_force_performance_level").read().strip(),
                        "sclk": cur("pp_dpm_sclk"), "mclk": cur("pp_dpm_mclk")})
        except Exception:  # noqa: BLE001
            pass
    return out or None


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks of this script."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--length-mbp", type=float, default=100.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--eps-alpha", type=float, default=0.0)
    ap.add_argument("--eps-beta", type=float, default=0.0)
    ap.add_argument("--raw", action="store_true", help="time set_raw -> E_step -> loglik only (no cold preparation)")
    ap.add_argument("--check", action="store_true", help="N > 1: after the timed region gather every rank's host threads, key "
                    "dictionary and Q (four terms) and assert that the dictionaries agree and Q is bitwise identical on all ranks")
    ap.add_argument("--no-ref-width", action="store_true", help="skip the extra evals of the reference-width configuration (`value_ref_width`): "
                    "for rocprofv3 passes, whose per-kernel totals must hold the default path only")
    ap.add_argument("--warm", action="store_true", help="additionally report the opt-in warm start (smcpp_set_warm_start) on a "
                    "trajectory of perturbed parameters; never part of `value`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using the launcher's world size", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # Fewer GPUs than ranks (or SMCPP_BENCH_BACKEND=gloo): the ranks share devices and reduce on the host.  That
    # exercises the whole N > 1 path (sharding, key union, packed all-reduce, barrier, max over ranks) and is flagged
    # in the output; the measured configuration is always nccl (= RCCL), one rank per GPU.
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SMCPP_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
    if backend != "nccl":
        local_rank = local_rank % ndev
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
    from smcpp_amd import dist as sd
    from smcpp_amd.model import PiecewiseModel, TwoPopulationModel
    # torch.distributed.run exports OMP_NUM_THREADS=1 to every rank; the engine's host phase (conditioned SFS per
    # hidden state, one eigensystem per eigen key) wants a handful of threads — the reference's --cores / set_num_threads
    # (M >= 128: the 256 x 256 eigenproblems run on teams of 8 threads per eigen key, nonsym_eig_team.hpp)
    # (GPU box: the cgroup allows 16 CPUs; 64 hidden states over 15 threads measured 0.153 ms of cold preparation against 0.185 ms
    # with 12 and 0.20 ms with 16, where the OpenMP workers and the launching thread exceed the quota)
    default_threads = "15"
    # never more than this rank's share of the CPUs the container may actually use (cgroup quota, not os.cpu_count():
    # the 1-GPU box shows 256 CPUs and allows 16; OpenMP workers spin between regions and eat the quota of their neighbours)
    avail = os.cpu_count() or 8
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = max(1, min(avail, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    share = max(1, avail // max(1, world) - (1 if avail // max(1, world) > 2 else 0))
    host_threads = max(1, min(int(os.environ.get("SMCPP_BENCH_THREADS", default_threads)), share))
    _smcpp.set_num_threads(host_threads)
    M, n, fixture, desc = WORKLOADS[args.workload]
    if args.workload == "shaping":
        return bench_shaping(args, n, desc, world, rank, torch)
    length_bp = int(args.length_mbp * 1e6)
    par = None
    sharded_kw = {}
    if args.workload == "c4":
        sharded_kw = dict(a=(2, 0))
        contigs = [synth.synth_contig_twopop(rank, length_bp, n, n)]
        hs = synth.hidden_states(M)
        a, s_ = synth.model_pieces()
        model = TwoPopulationModel(PiecewiseModel(a, s_, 1e4, pid="pop1"),
                                   PiecewiseModel(1.5 + 0.5 * np.cos(np.arange(8)), s_[:8], 1e4, pid="pop2"), 0.5)
        theta, rho, alpha, pol = synth.THETA, synth.RHO, synth.ALPHA, 0.5
        all_contigs = None

        def factory(obs, d):
            return _smcpp.PyTwoPopInferenceManager(n, n, 2, 0, obs, hs, ("pop1", "pop2"), pol, device=d)
    else:
        if args.workload in ("posterior", "posterior64", "posterior128"):
            hs = synth.hidden_states(M)
            a, s_ = synth.model_pieces()
            theta, rho, alpha, pol = 1e-4 * 2, 6e-5, 1.0, 0.5
            contigs = [synth_posterior_contig(1_000_000, n, seed=7 + rank)]
        else:
            par = np.load(os.path.join(ROOT, "tests", "golden", fixture))
            hs, a, s_ = par["hs"], par["a"], par["s"]
            theta, rho, alpha, pol = float(par["theta"]), float(par["rho"]), float(par["alpha"]), float(par["pol"])
            if args.workload == "c3":
                owner = sd.lpt_shard(synth.C3_LENGTHS_MBP, world)
                contigs = [synth.synth_contig(i, int(synth.C3_LENGTHS_MBP[i] * 1e6), n)
                           for i in range(len(owner)) if owner[i] == rank]
            else:
                contigs = [synth.synth_contig(rank, length_bp, n)]   # contig index = rank: independent contigs, weak scaling
        model = PiecewiseModel(a, s_, 1e4, pid="pop1")

        def factory(obs, d):
            return _smcpp.PyOnePopInferenceManager(n, obs, hs, ("pop1",), pol, device=d)

    if world > 1:
        # every rank generated only its own contigs; tell the sharded manager the global layout
        if args.workload == "c3":
            lengths = [int(x * 1e4) for x in synth.C3_LENGTHS_MBP]
            obs_all = [None] * len(lengths)
            owner = sd.lpt_shard(lengths, world)
            mine = [i for i in range(len(lengths)) if owner[i] == rank]
            for i, c in zip(mine, contigs):
                obs_all[i] = c
        else:
            lengths = [1] * world                      # one contig per rank; equal weights -> contig r on rank r
            obs_all = [None] * world
            obs_all[rank] = contigs[0]
        # (c4: the manager's own two-population factory, n = (n1, n2), a = (2, 0); one population: the factory above)
        if args.workload == "c4":
            sim = sd.ShardedInferenceManager((n, n), obs_all, hs, ("pop1", "pop2"), pol, device=local_rank, lengths=lengths, **sharded_kw)
        else:
            sim = sd.ShardedInferenceManager(n, obs_all, hs, ("pop1",), pol, device=local_rank, lengths=lengths, factory=factory)
        assert [i for i in sim.mine] == ([i for i in range(len(lengths)) if sd.lpt_shard(lengths, world)[i] == rank])
        im = sim.im
    else:
        sim = None
        im = factory(contigs, local_rank)
    top = sim if sim is not None else im
    top.theta = theta; top.rho = rho; top.alpha = alpha
    if args.workload in ("posterior", "posterior64", "posterior128", "pbinned"):
        im.save_gamma = True
    if args.chunk or args.eps_alpha or args.eps_beta:
        im.set_chunking(args.chunk, args.eps_alpha, args.eps_beta)

    if args.workload == "qgrad":
        return bench_qgrad(args, im, model, a, s_, M, n, desc, host_threads, world, rank, torch)

    # wall clock of the three calls of an eval as the CALLER sees them (three perf_counter reads, ~0.2 us): what an eval spends in the
    # Python binding in front of the C ABI (`top.model = ...`: the model object's own piece arithmetic, numpy -> pointers) shows up
    # here and in no engine-side interval
    phase_ns = [0, 0, 0, 0]
    pc = time.perf_counter_ns

    def one_eval():
        t_a = pc()
        top.model = model            # setParams: parameters dirty, A6-A10 + eigensystems + uploads are all redone
        t_b = pc()
        top.E_step()                 # (N > 1: includes the single all-reduce of the packed statistics)
        t_c = pc()
        ll_ = top.loglik()
        t_d = pc()
        phase_ns[0] += t_b - t_a; phase_ns[1] += t_c - t_b; phase_ns[2] += t_d - t_c; phase_ns[3] += 1
        return ll_

    raw = None

    def one_eval_raw():
        top.set_raw(*raw)            # prepared parameters handed over: no cold preparation in the E-step
        top.E_step()
        return top.loglik()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ll = one_eval()
    # the prepared parameters of the same model, for the hmm-only split (global key list when sharded)
    kk = sim.keys if sim is not None else im.keys
    if sim is None:
        ep = im.emission_probs
        raw = (im.pi, im.transition, kk, np.array([ep[tuple(int(x) for x in k)] for k in kk]))
    step = one_eval_raw if (args.raw and raw is not None) else one_eval
    for _ in range(args.warmup):
        step()
    barrier()
    # Python's cyclic collector is parked for the timed regions of this script: with torch imported a full collection walks ~10^6
    # objects (35 - 40 ms, measured: one Q call of 37 ms among forty of 0.15 ms) and where it falls is an accident of how many
    # containers the set-up allocated; the engine allocates nothing per eval that needs it
    import gc
    gc.collect(); gc.disable()
    timings, host_timings = [], []
    cg0 = cgroup_cpu_stat()
    t0 = time.perf_counter()
    # the engine's HIP-event intervals of an eval (chains, statistics, finalisation: `split_ms`, the roofline's kernel time) are read back
    # on every FOURTH eval of the timed region: reading them costs ~10 us of host time per eval, which is instrumentation, not the eval
    every = 4 if args.steps >= 20 else 1
    phase_ns[:] = [0, 0, 0, 0]
    for i in range(args.steps):
        ll = step()
        if i % every == 0:
            timings.append(im.last_timing())
            host_timings.append(im.last_host_timing())
    barrier()
    elapsed = time.perf_counter() - t0
    caller_ms = ({"set_model": 1e-6 * phase_ns[0] / phase_ns[3], "E_step": 1e-6 * phase_ns[1] / phase_ns[3],
                  "loglik": 1e-6 * phase_ns[2] / phase_ns[3],
                  "note": "mean wall clock of the three calls of an eval as the Python caller sees them; set_model holds the binding's "
                          "own work in front of smcpp_set_params (model object -> arrays)"} if phase_ns[3] else None)
    cg1 = cgroup_cpu_stat()
    cpu_throttle = ({k: cg1[k] - cg0[k] for k in cg0} if (cg0 and cg1) else None)
    per_rank = None
    if world > 1:
        # every rank's own clock over the timed region, its rows and positions (the load the LPT shard gave it): gathered AFTER
        # the timed region; `value` uses the MAX over ranks
        mine_t = torch.tensor([elapsed, float(sum(len(c) for c in contigs)), float(sum(int(c[:, 0].sum()) for c in contigs))],
                              dtype=torch.float64, device=red_dev)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        allt = np.array([t.cpu().numpy() for t in allt])
        elapsed = float(allt[:, 0].max())
        per_rank = {"ms_per_step": [1e3 * float(x) / args.steps for x in allt[:, 0]], "rows": [int(x) for x in allt[:, 1]],
                    "positions": [int(x) for x in allt[:, 2]],
                    "load_max_over_mean": float(allt[:, 1].max() / allt[:, 1].mean()),
                    "ranks_reported_by_backend": int(dist.get_world_size()), "backend": dist.get_backend()}
    ms_per_step = 1e3 * elapsed / args.steps
    if args.workload == "c3":
        value = args.steps / elapsed                     # whole-genome evals per second (the ranks share ONE eval)
    else:
        value = world * args.steps / elapsed             # contig-E-step evals per second, whole job
    med = {k: float(np.median([t[k] for t in timings])) for k in timings[0]}
    med.update({k: float(np.median([t[k] for t in host_timings])) for k in host_timings[0]})
    # what the timed E-steps actually ran (smcpp_describe: chain family, chunks, history passes, arithmetic of the stored passes)
    plan = im.describe()["plan"] if hasattr(im, "describe") else {}
    float_scans = bool(plan.get("float_scans_in_stored_passes"))

    # ---- the same eval at the REFERENCE'S WIDTH (outside `value`): every scan of the stored passes in fp64 ----
    # The default path of one-state-per-lane inputs forms the off-diagonal sums of a position in float (chains_ss.hpp:
    # ss_x_scan_fwd / _bwd) where the reference's beta recursion and span > 1 forward rows are double (src/hmm.cpp:72-81,97-149;
    # only alpha's storage and the span-1 forward row are float, include/hmm.h:35).  SMCPP_SS_MIXED=0 keeps them in fp64: that
    # configuration is timed here the same way (same warm-up, same number of steps, barrier + synchronize on both sides).
    ref_width = None
    if float_scans and not args.raw and not args.no_ref_width:
        from smcpp_amd import _engine as _E
        _E.set_option("SMCPP_SS_MIXED", "0")
        try:
            for _ in range(max(2, args.warmup)):
                one_eval()
            barrier()
            tr0 = time.perf_counter()
            for _ in range(args.steps):
                ll_rw = one_eval()
            barrier()
            el = time.perf_counter() - tr0
            if world > 1:
                tt = torch.tensor([el], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            trw = im.last_timing()
            assert not im.describe()["plan"]["float_scans_in_stored_passes"]
            ref_width = {"value": (1.0 if args.workload == "c3" else world) * args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                         "dtype": "f64 (alpha stored as float, as the reference: include/hmm.h:35)", "loglik": float(ll_rw),
                         "chains_wall_ms": trw["chains_wall_ms"], "passes": trw["fwd_passes"],
                         "how": "SMCPP_SS_MIXED=0, same process, same manager, same steps / warm-up, timed like `value`"}
        finally:
            _E.set_option("SMCPP_SS_MIXED", None)
            one_eval()                  # back on the default path (the hmm-only split and the checks below use it)

    # ---- hmm-only split: the same eval with set_raw (outside the timed region) ----
    if raw is not None and not args.raw:
        for _ in range(2):
            one_eval_raw()
        ts = []
        for _ in range(max(5, args.steps // 2)):
            torch.cuda.synchronize()
            t1 = time.perf_counter(); one_eval_raw(); ts.append(time.perf_counter() - t1)
        med["hmm_only_ms"] = 1e3 * float(np.median(ts))

    # ---- opt-in warm start on a parameter TRAJECTORY (outside the timed region; never part of `value`) ----
    # smcpp_set_warm_start: the chains of an E-step start from the boundary vectors the previous E-step converged to (one light
    # pass fewer).  An optimiser never evaluates the same point twice, so the models alternate between +-2 % perturbations of the
    # population sizes: every eval sees parameters ~4 % away from the previous eval's.
    warm = None
    if args.warm and args.workload in ("headline", "c2") and world == 1 and hasattr(im, "set_warm_start") and not args.raw:
        try:
            rng = np.random.default_rng(7)
            traj = [PiecewiseModel(np.asarray(a) * np.exp(0.02 * sgn * rng.standard_normal(len(a))), s_, 1e4, pid="pop1")
                    for sgn in (1.0, -1.0, 1.0, -1.0)]

            def run_traj(n):
                ts_ = []
                for i in range(n):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    top.model = traj[i % len(traj)]
                    top.E_step()
                    top.loglik()
                    ts_.append(time.perf_counter() - t1)
                return 1e3 * float(np.median(ts_[2:]))
            nw = max(8, args.steps // 2)
            cold_ms = run_traj(nw)
            im.set_warm_start(True)
            warm_ms = run_traj(nw)
            wt = im.last_timing()
            im.set_warm_start(False)
            warm = {"ms_per_eval": warm_ms, "evals_per_s": 1e3 / warm_ms, "cold_ms_per_eval_same_trajectory": cold_ms,
                    "passes": wt.get("fwd_passes"), "chains_wall_ms": wt.get("chains_wall_ms"),
                    "note": "opt-in smcpp_set_warm_start on a trajectory of +-2 % parameter steps; reported beside `value`, never in it"}
            top.model = model
            top.E_step()
        except Exception as ex:                                  # diagnostics only
            warm = {"error": str(ex)}

    # ---- roofline of the dominant kernel ----
    # The chain kernels (k_fwd_coop / k_bwd_coop, k_*_big for M > 64) dominate.  smcpp_last_timing brackets ALL pass
    # launches of each chain of one E-step with hipEvents on the stream they are launched on, so `kernel_ms_per_step`
    # is the kernel's time per step summed over its passes (= rocprofv3's total for the kernel / steps).  `achieved`
    # credits ONE pass of algorithmic work: the re-run passes of the chunk-parallel fixed point are overhead, not work.
    flops, bytes_f, bytes_b, R1, Re = chain_pass_work(contigs, M)
    fwd_ms, bwd_ms = med["forward_ms"], med["backward_ms"]
    mode = im.chain_mode() if hasattr(im, "chain_mode") else -1
    F_alg, B_alg = eval_work(contigs, M)
    if mode in (5, 6):
        # Scan chains (chains_ss.hpp): ONE kernel runs both directions (forward and backward wavefronts share a workgroup) and
        # executes NO matrix product: O(M) prefix scans per POSITION on the vector ALU, one wavefront per chunk.  Its bound is VALU
        # ISSUE, so that is what the roofline block prices: `achieved` = wave-level VALU instructions per second summed over
        # all launches of one E-step (SQ_INSTS_VALU of the committed rocprofv3 --pmc pass of this same command when there is one
        # for this workload, else the instruction count of the kernel's ISA per position x the positions every pass walks),
        # `peak` = one wave64 DPP / fp64 instruction per 4 cycles and SIMD (1024 SIMDs x the 2.4 GHz peak engine clock / 4).  The reference's dense
        # flops priced on this kernel's clock stay available as `frac_dense_equivalent` (rounds 1-3 quoted that as `frac`; it is
        # a speed against the reference's ALGORITHM, grows with M whatever the kernel does, and is not a utilisation).
        kname = "k_chain_ss"
        k_ms = med["chains_wall_ms"]
        k_flops, k_bytes = 2.0 * flops, bytes_f + bytes_b
        ach_tflops = k_flops / (1e-3 * k_ms) / 1e12 if k_ms > 0 else 0.0
        ach_gbs = k_bytes / (1e-3 * k_ms) / 1e9 if k_ms > 0 else 0.0
        if mode == 6:
            # hybrid rows (un-binned data): a row longer than the threshold is ONE eigen-power step - a dependent chain of ~200
            # instructions, i.e. ~4 scan positions' worth of ISSUE (the chunk balancing prices it at 8: that
            # is latency, two LDS round trips per row); shorter rows are expanded position by position
            positions = float(sum(int(np.where(c[:, 0] > 6, 4, c[:, 0]).sum()) for c in contigs))
        else:
            positions = float(sum(int(c[:, 0].sum()) for c in contigs))
        npl = (M + 63) // 64
        # VALU instructions per position in the inner loops of k_chain_ss<NPL> (llvm-objdump of the shipped code object; DESIGN.md
        # section 5): stored pass forward / backward, light float pass forward / backward.  One state per lane (round 5): every scan of
        # the stored passes in float - 25 / 31 VALU (+ 6 / 2 hazard slots) where the fp64 scans took 45 / 57 (+ 7); M <= 32: 21 / 25
        ipp = {1: (25, 31, 20, 28), 2: (56, 74, 28, 39), 3: (65, 88, 36, 53), 4: (74, 102, 41, 56)}.get(npl)
        if npl == 1 and M <= 32:
            ipp = (21, 25, 18, 25)
        if npl == 1 and args.workload in ("posterior", "posterior64"):
            ipp = (45, 57, 20, 28)                         # save_gamma keeps the fp64 scans
        sq = sq_counters(args.workload if (args.length_mbp == 100.0 and world == 1) else None, kname)
        passes = med["fwd_passes"]
        est_instr = None
        one_pass_instr = None
        if ipp is not None:
            # per E-step: `light` float passes + one full pass + the merge re-runs (which stop after the forgetting length: their
            # share of a full pass is what the timing split says, not a count) - a MODEL, used only without counters
            light = max(0.0, passes - 2.0)
            est_instr = positions * (light * (ipp[2] + ipp[3]) + 1.4 * (ipp[0] + ipp[1]))
            one_pass_instr = positions * (ipp[0] + ipp[1])     # what ONE sequential pass of both chains needs (the useful work)
        instr = sq["SQ_INSTS_VALU_per_step"] if sq else est_instr
        # peak: G wave64-instructions / s of the vector ALUs.  A SIMD needs 4 cycles per wave64 DPP or fp64 instruction - the two
        # kinds the scans are made of: 1024 SIMDs x 2.4 GHz / 4 = 614.4 (= the guide's 157.3 TFLOP/s FP32 vector peak / 256 flop per
        # v_pk_fma_f32 wave-instruction).  Measured, tools/dpp_lab.hip with 8 wavefronts per SIMD (profiles/r04_*_dpp_lab.log):
        # v_fmac_f32_dpp 587, v_mov_b32_dpp 587, v_fma_f64 583; with ONE wavefront per SIMD - this kernel's regime - 467 / 468 / 427.
        # (Plain 32-bit VALU instructions issue in 2 cycles: v_add_f32 1002 measured; they are 4 of the 25 - 59 per position.)
        # Until r04_k the block divided by 1024 x 2.4 = 2457.6 (one instruction per cycle and SIMD), which no instruction reaches.
        peak_4cycle = 1024 * 2.4 / 4.0
        # ... per instruction CLASS: plain 32-bit VALU instructions issue in 2 cycles, DPP / 64-bit / cross-lane ones in 4
        # (tools/dpp_lab.hip).  The class mix of the launched instantiation's loops comes from the shipped code object
        # (tools/isa_mix.py -> profiles/r0*_isa_mix.json): peak = 1024 SIMDs x 2.4 GHz / mean issue cycles of that mix.
        mix = isa_mix(npl, mode == 6, npl == 1 and M <= 32 and mode == 5)
        peak_ginstr = mix["peak_ginstr_per_s"] if mix else peak_4cycle
        one_wave_ginstr = 447.0                            # measured mean of the three, one wavefront per SIMD
        ach_ginstr = (instr / (1e-3 * k_ms) / 1e9) if (instr and k_ms > 0) else None
        useful_ginstr = (one_pass_instr / (1e-3 * k_ms) / 1e9) if (one_pass_instr and k_ms > 0) else None
        roof = dict(bound="valu-issue", achieved=ach_ginstr, peak=peak_ginstr, unit="G wave-instr/s",
                    frac=(ach_ginstr / peak_ginstr) if ach_ginstr else None,
                    # the same with only the instructions ONE sequential pass of both chains needs in the numerator: what the history
                    # passes and the merge re-run execute on top of that is redundant work of the chunk-parallel fixed point
                    frac_useful=(useful_ginstr / peak_ginstr) if useful_ginstr else None)
        other = {"bound_detail": "VALU issue: O(M) DPP scans per position over the semiseparable structure of T, one wavefront per "
                                 "SIMD on one contig (tools/dpp_lab.hip: 5.3 clocks between two issues of one wavefront, 4 with several); no matrix product executes",
                 "instr_source": (sq["source"] if sq else "ISA instruction count x positions walked (model; no counter profile for this workload)"),
                 "instr_per_position": (dict(zip(["full_fwd", "full_bwd", "light_fwd", "light_bwd"], ipp)) if ipp else None),
                 "peak_source": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 DPP / fp64 instruction (MI355X_MICROARCH.md: 157.3 TFLOP/s FP32 "
                                "vector = 614.4 G v_pk_fma_f32 / s); tools/dpp_lab.hip measures 583 - 587 with 8 wavefronts per SIMD, 427 - 468 with one",
                 "peak_all_4_cycle": peak_4cycle, "frac_all_4_cycle": (ach_ginstr / peak_4cycle) if ach_ginstr else None,
                 "class_mix": mix,
                 "frac_one_wavefront_issue_bound": (ach_ginstr / one_wave_ginstr) if ach_ginstr else None,
                 "one_pass_instr": one_pass_instr,
                 "executed_over_one_pass": (instr / one_pass_instr) if (instr and one_pass_instr) else None,
                 "sq_counters": sq,
                 "executed_flops_estimate": 30.0 * M * positions * 2.0,
                 "frac_dense_equivalent": ach_tflops / FP64_PEAK_TFLOPS, "dense_equivalent_tflops": ach_tflops,
                 "positions": positions, "positions_per_us": positions / (1e3 * k_ms) if k_ms > 0 else 0.0}
        note = ("both chains in one kernel; frac = executed wave-level VALU instructions / (kernel time x 1024 SIMDs x 2.4 GHz / mean issue "
                "cycles of the kernel's instruction-class mix); frac_useful = the same with the instructions of ONE sequential pass; "
                "frac_hbm = algorithmic alpha/beta/normaliser bytes of one pass / kernel time / 8 TB/s")
    else:
        # The chain kernels (k_fwd_coop / k_bwd_coop, k_*_big for M > 64) dominate.  smcpp_last_timing brackets ALL pass
        # launches of each chain of one E-step with hipEvents on the stream they are launched on, so `kernel_ms_per_step`
        # is the kernel's time per step summed over its passes (= rocprofv3's total for the kernel / steps).  `achieved`
        # credits ONE pass of algorithmic work: the re-run passes of the chunk-parallel fixed point are overhead, not work.
        dom_fwd = fwd_ms >= bwd_ms
        k_ms = fwd_ms if dom_fwd else bwd_ms
        k_flops, k_bytes = flops, (bytes_f if dom_fwd else bytes_b)
        fam = "_lock" if mode == 4 else "_big" if M > 64 else "_coop"
        kname = ("k_fwd" if dom_fwd else "k_bwd") + fam
        ach_tflops = flops / (1e-3 * k_ms) / 1e12 if k_ms > 0 else 0.0
        ach_gbs = k_bytes / (1e-3 * k_ms) / 1e9 if k_ms > 0 else 0.0
        roof = dict(bound="mfma", achieved=ach_tflops, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=ach_tflops / FP64_PEAK_TFLOPS)
        other = {"bound_detail": "latency of the dependent mat-vecs (VALU / L2 streaming; lock-step family: fp64 MFMA)",
                 "other_chain": {"kernel": ("k_bwd" if dom_fwd else "k_fwd") + fam,
                                 "kernel_ms_per_step": bwd_ms if dom_fwd else fwd_ms,
                                 "tflops": flops / (1e-3 * (bwd_ms if dom_fwd else fwd_ms)) / 1e12 if min(fwd_ms, bwd_ms) > 0 else 0.0}}
        note = "latency-bound sequential chains: frac = one pass of algorithmic flops / kernel time per step / fp64 peak"
    # HBM bytes per step from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs of this same command, corrected as MI355X_MICROARCH.md prescribes); only for the profiled workload
    traffic = traffic_src = None
    try:
        if args.workload == "headline" and args.length_mbp == 100.0 and world == 1:
            import glob
            pf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]*_hbm_traffic_pmc.json")))[-1]
            prof = json.load(open(pf))
            # every launch of the kernel (pass 0 / light / full / re-run): per-step bytes of all of them
            tr = [v["bytes_per_step"] for name, v in prof["kernels"].items() if kname in name]
            traffic = float(sum(tr)) if tr else None
            traffic_src = ("committed PMC passes of this command, not measured in this run: " + os.path.relpath(pf, ROOT)) if tr else None
    except Exception:  # noqa: BLE001
        traffic = traffic_src = None
    roof.update(
        traffic=traffic, traffic_source=traffic_src, frac_hbm=ach_gbs / HBM_PEAK_GBS,
        kernel=f"{kname} (all launches of one E-step)",
        kernel_ms_per_step=k_ms, passes=med["fwd_passes"],
        algorithmic_flops_one_pass=k_flops, algorithmic_bytes_one_pass=k_bytes,
        hbm_gbs=ach_gbs, hbm_frac=ach_gbs / HBM_PEAK_GBS, **other,
        # whole eval on SURVEY.md §8(d)'s F_alg / B_alg: a figure of merit against the REFERENCE's algorithm (its 2 M^3 Re term is
        # not executed by the restructured statistics), not a fraction of anything this engine executes
        eval={"F_alg": F_alg, "B_alg": B_alg, "reference_algorithm_tflops_equivalent": F_alg / (1e-3 * ms_per_step) / 1e12,
              "gbs_on_B_alg": B_alg / (1e-3 * ms_per_step) / 1e9},
        note=note)

    if args.workload == "posterior128":
        # per-row posteriors from eigen-power pieces + scan steps: every position of a span > 1 row is walked once forward and once backward
        # by the O(M) scan steps (two states per lane: 56 / 74 VALU instructions per step, as the stored passes of k_chain_ss<2>) on top of
        # two M x M products per 64-position piece on the matrix cores
        pos_e = float(sum(int(c[c[:, 0] > 1, 0].sum()) for c in contigs))
        g_ms = med["finalize_ms"]
        ginstr = pos_e * (56 + 74) / (1e-3 * max(g_ms, 1e-9)) / 1e9
        roof["posterior"] = {"per_row_gamma": plan.get("per_row_gamma") if isinstance(plan, dict) else None, "gamma_ms": g_ms,
                             "span_rows": Re, "positions_in_span_rows": pos_e, "pieces_of_64": pos_e / 64.0,
                             "piece_vector_flops": 2 * 2.0 * M * M * pos_e / 64.0}
        if g_ms > roof.get("kernel_ms_per_step", 0.0):
            chains_block = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_useful", "kernel", "kernel_ms_per_step") if k in roof}
            roof.update(bound="valu-issue", achieved=ginstr, peak=614.4, unit="G wave-instr/s", frac=ginstr / 614.4, frac_useful=ginstr / 614.4,
                        kernel="k_gamma_rows_scan<2, true> (+ k_piece_vectors, k_gamma_merge_pieces): per-row posteriors of the span > 1 rows from "
                               "eigen-power pieces walked by scan steps (hmm.cpp:113-121)", kernel_ms_per_step=g_ms, chains=chains_block,
                        note="dominant phase = the per-row gammas: achieved = positions of the span > 1 rows x 130 VALU instructions (one forward, "
                             "one backward scan step) / the finalisation interval; peak = one wave64 DPP / fp64 instruction per 4 cycles per SIMD")
    if args.workload in ("posterior", "posterior64"):
        # the product of this workload is the M x (L+1) posterior matrix: 8 M L bytes written by the statistics phase
        # (span-1 rows by k_s1_scalars, eigen rows by k_gamma_rows_mfma: 2 M^3 flops per eigen row on the matrix cores)
        gbytes = 8.0 * M * sum(len(c) + 1 for c in contigs)
        t_stat = 1e-3 * (med["stats_ms"] + med["finalize_ms"])
        roof["posterior"] = {"gamma_bytes": gbytes, "statistics_ms": 1e3 * t_stat, "gamma_write_gbs": gbytes / t_stat / 1e9,
                             "gamma_write_frac_of_hbm": gbytes / t_stat / 1e9 / HBM_PEAK_GBS,
                             "eigen_row_tflops": 2.0 * M ** 3 * Re / t_stat / 1e12,
                             # the per-row gamma kernel (k_gamma_rows_b: 2 M^3 flop per span row on the fp64 matrix cores) alone, on the
                             # finalisation interval it dominates - at M = 64 it, not the chain kernel above, is the longest kernel of the eval
                             "gamma_rows_tflops": 2.0 * M ** 3 * Re / (1e-3 * max(med["finalize_ms"], 1e-9)) / 1e12,
                             "gamma_rows_frac_of_fp64_mfma_peak": 2.0 * M ** 3 * Re / (1e-3 * max(med["finalize_ms"], 1e-9)) / 1e12 / FP64_PEAK_TFLOPS,
                             "fwd_passes": med["fwd_passes"], "bwd_passes": med["bwd_passes"]}
        g_ms = med["finalize_ms"]
        if g_ms > roof["kernel_ms_per_step"]:
            # the per-row gamma kernel is the LONGEST kernel of this workload (M = 64: 6.3 of 11.5 ms): it, not the chain kernel, is the
            # dominant kernel the roofline block must describe (VERDICT r05 "What's weak" 7); the chain kernel's block moves to `chains`
            g_tflops = 2.0 * M ** 3 * Re / (1e-3 * g_ms) / 1e12
            chains_block = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_useful", "kernel", "kernel_ms_per_step") if k in roof}
            roof.update(bound="mfma", achieved=g_tflops, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=g_tflops / FP64_PEAK_TFLOPS,
                        kernel="k_gamma_rows_b (per-row posterior of the span > 1 rows, hmm.cpp:113-121: 2 M^3 flop per row on v_mfma_f64_16x16x4)",
                        kernel_ms_per_step=g_ms, algorithmic_flops_one_pass=2.0 * M ** 3 * Re, frac_useful=g_tflops / FP64_PEAK_TFLOPS,
                        chains=chains_block,
                        note="dominant kernel = the per-row gamma kernel: achieved = 2 M^3 x (span > 1 rows) / its interval; `chains` holds the chain kernel's issue roofline")
    # ---- N > 1 consistency (--check): what every rank holds after the single all-reduce ----
    multi_check = None
    if args.check and world > 1:
        import hashlib
        qv = np.array(top.Q(separate=True), dtype=np.float64)
        mine = {"rank": rank, "host_threads": host_threads, "q": qv.tobytes().hex(), "contigs": [int(i) for i in sim.mine],
                "local_keys": int(len(im.keys)), "global_keys": hashlib.sha1(np.ascontiguousarray(sim.keys, dtype=np.int32).tobytes()).hexdigest(),
                "n_global_keys": int(len(sim.keys)), "loglik": float(ll)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        assert len({r["q"] for r in allr}) == 1, "Q differs between ranks"
        assert len({r["global_keys"] for r in allr}) == 1, "key dictionaries differ between ranks"
        assert len({r["loglik"] for r in allr}) == 1, "reduced log-likelihood differs between ranks"
        assert sorted(c for r in allr for c in r["contigs"]) == list(range(len(lengths))), "contigs are not partitioned"
        multi_check = {"ranks": world, "q_bitwise_identical": True, "q": [float(x) for x in qv], "host_threads_per_rank": [r["host_threads"] for r in allr],
                       "contigs_per_rank": [r["contigs"] for r in allr], "local_keys_per_rank": [r["local_keys"] for r in allr],
                       "global_keys": allr[0]["n_global_keys"], "cpu_quota": avail}
    # full-size parity against the compiled reference's recorded log-likelihoods (golden G16, tests/golden/make_golden_fullsize.py)
    parity_full = None
    try:
        gp = os.path.join(ROOT, "tests", "golden", f"G16_fullsize_{args.workload}.npz")
        if os.path.exists(gp) and args.length_mbp == 100.0 and not args.raw:
            z = np.load(gp)
            ref_ll = float(z["loglik"].sum()) if args.workload == "c3" else (float(z["loglik"][:world].sum()) if world <= len(z["loglik"]) else None)
            if args.workload == "c5":
                # G16's c5 entry is a 25 000-row prefix; the whole contig is golden G22 (tests/golden/make_golden_c5_full.py)
                g22 = os.path.join(ROOT, "tests", "golden", "G22_c5_full.npz")
                ref_ll = float(np.load(g22)["loglik"]) if (os.path.exists(g22) and world == 1) else None
                gp = g22
            if ref_ll is not None:
                parity_full = {"loglik_reference_full": ref_ll, "loglik_engine": float(ll), "rel_diff": abs(float(ll) - ref_ll) / abs(ref_ll),
                               "source": "tests/golden/" + os.path.basename(gp) + " (compiled reference, full size, all contigs of this run)"}
    except Exception:  # noqa: BLE001
        parity_full = None
    # posterior workloads: the decode's indices at full size against the compiled reference's (golden G20,
    # tests/golden/make_golden_argmax.py: argmax of every one of the 10^6 + 1 columns, the columns with a margin below 1e-3)
    try:
        gp = os.path.join(ROOT, "tests", "golden", "G19_headline.npz" if args.workload == "pbinned" else f"G20_{args.workload}.npz")
        if args.workload in ("posterior", "posterior64", "pbinned") and os.path.exists(gp) and world == 1 and not args.raw:
            z = np.load(gp)
            if synth.contig_crc(contigs[0]) == int(z["crc"]):
                arg = np.asarray(im.gamma_argmax(0)).astype(np.int64)
                ref_arg = z["gamma_argmax"].astype(np.int64)
                mism = np.nonzero(arg != ref_arg)[0]
                margin = np.full(len(ref_arg), float(z["low_margin_below"])); margin[z["low_margin_cols"]] = z["low_margin"]
                ref_ll = float(z["loglik"])
                parity_full = {"loglik_reference_full": ref_ll, "loglik_engine": float(ll), "rel_diff": abs(float(ll) - ref_ll) / abs(ref_ll),
                               "columns": int(len(ref_arg)), "argmax_mismatches": int(len(mism)),
                               "argmax_mismatches_with_reference_margin_above_1e-5": int((margin[mism] > 1e-5).sum()),
                               "reference_columns_with_margin_below_1e-5": int((margin < 1e-5).sum()),
                               "source": "tests/golden/" + os.path.basename(gp) + " (compiled reference HMM::Estep with save_gamma on the same "
                                         "rows; the engine ran set_params -> E_step, i.e. its own cold preparation)"}
    except Exception as ex:  # noqa: BLE001
        parity_full = {"error": repr(ex)}
    if ref_width is not None and parity_full is not None and parity_full.get("loglik_reference_full") is not None:
        rf = parity_full["loglik_reference_full"]
        ref_width["parity_full_size"] = {"loglik_reference_full": rf, "loglik_engine": ref_width["loglik"],
                                         "rel_diff": abs(ref_width["loglik"] - rf) / abs(rf)}
    out = None
    if rank == 0:
        out = {
            "metric": "E-step loglik-evals/sec (100 Mbp, M=64, n=20)" if args.workload == "headline"
            else ("whole-genome E-step loglik-evals/sec (22 contigs, 2872 Mbp, M=64, n=20)" if args.workload == "c3"
                  else f"E-step loglik-evals/sec ({desc})"),
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if args.workload == "c3" else "weak", "vs_baseline": None,
            # the arithmetic of the TIMED path: vector, diagonal term, normaliser bookkeeping and every statistic in fp64; alpha stored as
            # float (as the reference, include/hmm.h:35); the history-only passes in float; and - one state per lane, the default - the
            # off-diagonal scans of the stored passes in float as well.  `value_ref_width` below times the all-fp64-scan configuration.
            "dtype": ("f64 state + f32 scans (mixed)" if float_scans else "f64"), "data": "synthetic",
            "dtype_detail": {"float_scans_in_stored_passes": float_scans, "history_passes": "f32, store-free",
                             "alpha_storage": "f32 (reference: include/hmm.h:35)", "beta_storage_and_statistics": "f64",
                             "reference": "alpha f32, span-1 forward row f32, everything else f64 (src/hmm.cpp:59-152)"},
            "value_ref_width": (ref_width["value"] if ref_width else (value if not float_scans else None)),
            "ref_width": ref_width,
            "cpu_throttle_in_timed_region": cpu_throttle,
            "caller_ms": caller_ms,
            # what the caller's E_step call spends outside the engine's own host phase and outside the device intervals (launch /
            # wake-up / completion latencies around the kernels): ~0.01 - 0.03 ms; config C4 has been seen at 0.3 - 0.7 on some boxes
            "unaccounted_ms": (caller_ms["E_step"] - med.get("host_total_ms", med.get("host_prep_ms", 0.0)) - med.get("device_total_ms", 0.0)
                               if caller_ms and "device_total_ms" in med else None),
            "gpu_power_state": gpu_power_state(),
            "plan": plan,
            "config": {"workload": desc, "eval": "set_raw -> E_step -> loglik (no cold preparation)" if args.raw
                       else "set_params -> E_step -> loglik (SURVEY.md 8(d))",
                       "M": M, "n": n, "rows": int(sum(len(c) for c in contigs)),
                       "span1_rows": R1, "eigen_rows": Re, "contigs_per_gpu": len(contigs),
                       "length_mbp": float(sum(synth.C3_LENGTHS_MBP)) if args.workload == "c3" else args.length_mbp,
                       "loglik": ll, "host_threads": host_threads, "python_gc": "collected, then disabled for the timed region",
                       "parallelism": f"contig-sharded x{world}, 1 all-reduce/E-step" if world > 1 else "single GPU"},
            "split_ms": med,
            "split_ms_note": f"medians of the engine's HIP-event intervals, read back on every {every}th eval of the timed region ({len(timings)} evals)",
            "roofline": roof,
        }
        if parity_full is not None:
            out["parity_full_size"] = parity_full
        if multi_check is not None:
            out["multi_gpu_check"] = multi_check
        if warm is not None:
            out["warm_start"] = warm
        if world > 1:
            out["config"]["backend"] = "nccl (RCCL)" if backend == "nccl" else \
                f"{backend}: {world} ranks share {ndev} device(s) - functional test of the N>1 path, not a measurement"
            out["config"]["collective"] = ("one all_reduce(sum, f64) of %d doubles per E-step, issued on the engine's stream behind the pack "
                                           "kernel (no host wait before it)" % sim.im.stats_len()) if backend == "nccl" else \
                "one all_reduce(sum, f64) per E-step through the host (gloo)"
            out["per_rank"] = per_rank
        if not args.no_cpu and world == 1:                          # the CPU baseline is timed at N = 1 only
            # the reference's E-step gets the engine's prepared parameters of this model (pi, T, emission table), its
            # cold preparation is timed on the model itself (one population only: the two-population joint CSFS is in
            # a translation unit of the reference that needs GSL)
            model_args = None if args.workload == "c4" else (a, s_, hs, rho, theta, n)
            def engine_on_prefix(prefix):
                # the engine on exactly the rows the reference just ran (same prepared parameters): the in-run parity figure
                im2 = factory([np.ascontiguousarray(prefix)], local_rank)
                im2.theta = theta; im2.rho = rho; im2.alpha = alpha
                im2.set_raw(*raw)
                im2.E_step()
                return im2.loglik()
            out["cpu_baseline"] = cpu_baseline(raw, model_args, np.concatenate(contigs), args.cpu_seconds, args.raw,
                                               engine_on_prefix if len(contigs) == 1 else None, len(contigs))
            if out["cpu_baseline"] and out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"]
                # the reference parallelises over contigs (one OpenMP thread each): against ALL the threads it could use on this input
                rt = max(1, min(len(contigs), os.cpu_count() or 1))
                out["speedup_vs_cpu_reference_threads"] = {"threads": rt, "ratio_assuming_perfect_scaling": value / (out["cpu_baseline"]["value"] * rt)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_qgrad(args, im, model, a, s_, M, n, desc, host_threads, world, rank, torch):
    """`--workload qgrad`: evals/s of Q-with-gradient (16 directions, one per model piece) on the statistics of one E-step.
    value = the default route (conditioned SFS + emission table + Q reduction on the device, O(M) host part); the same call
    with the whole preparation on the host threads (rounds 1-3: smcpp_set_prep_mode(1)) is timed beside it."""
    from smcpp_amd import _engine as E
    if world != 1:
        raise SystemExit("--workload qgrad is a single-GPU measurement")
    im.model = model
    im.E_step()
    K = len(a)
    aa = np.ascontiguousarray(a, dtype=np.float64); ss = np.ascontiguousarray(s_, dtype=np.float64)
    da = np.ascontiguousarray(np.eye(K))
    val = np.zeros(4); jac = np.zeros((4, K))
    rng = np.random.default_rng(5)

    def one(i):
        # an optimiser never evaluates the same point twice: a fresh +-1 % perturbation per call
        ai = np.ascontiguousarray(aa * np.exp(0.01 * rng.standard_normal(K)))
        E.check(E.lib().smcpp_set_params(im._im, K, E.dptr(ai), E.dptr(da), K, E.dptr(ss)))
        E.check(E.lib().smcpp_q(im._im, E.dptr(val), E.dptr(jac)))
        return val.copy(), jac.copy()

    per_call = []

    def timed(steps, warmup):
        import gc
        for i in range(warmup):
            one(i)
        torch.cuda.synchronize()
        gc.collect(); gc.disable()           # (see main(): a full collection inside the timed region is 37 ms of a 6 ms region)
        per_call.clear()
        t0 = time.perf_counter()
        for i in range(steps):
            t1 = time.perf_counter()
            one(i)
            per_call.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    steps = max(args.steps, 40)
    dev_s = timed(steps, args.warmup)
    dev_calls = sorted(per_call)
    rng = np.random.default_rng(5)
    v_dev, j_dev = one(0)
    im.set_prep_mode(True)
    host_s = timed(max(10, steps // 4), 2)
    rng = np.random.default_rng(5)
    v_host, j_host = one(0)
    im.set_prep_mode(False)
    sc = np.abs(j_host).max(axis=1)
    out = {"metric": f"Q-with-gradient evals/sec ({desc})", "value": 1.0 / dev_s, "unit": "evals/s", "n_gpus": 1, "steps": steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * dev_s, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": desc, "eval": "set_params(a, da [16 x 16]) -> Q(val [4], jac [4 x 16])", "M": M, "n": n, "nder": K,
                      "host_threads": host_threads},
           "host_path": {"ms_per_step": 1e3 * host_s, "evals_per_s": 1.0 / host_s, "threads": host_threads,
                         "note": "the same call with the conditioned SFS, emission table, dense transition Jacobian and the Q sums on the "
                                 "host (smcpp_set_prep_mode(1): the route of rounds 1-3)"},
           "speedup_vs_host_path": host_s / dev_s,
           "per_call_ms": {"median": 1e3 * dev_calls[len(dev_calls) // 2], "min": 1e3 * dev_calls[0], "max": 1e3 * dev_calls[-1],
                           "note": "`value` is steps / total time of the timed region; the spread of the individual calls is reported beside it"},
           "parity": {"val_rel_diff_max": float(np.max(np.abs(v_dev - v_host) / np.abs(v_host))),
                      "jac_rel_diff_max_per_term": [float(x) for x in np.max(np.abs(j_dev - j_host), axis=1) / sc]}}
    print(json.dumps(out), flush=True)


def bench_shaping(args, n, desc, world, rank, torch):
    """`--workload shaping`: the pre-HMM data shaping of SURVEY.md 8 f-2 (thin_data -> bin_observations -> compress_repeated_obs,
    smcpp/data_filter.py:166-203) as device kernels (smcpp_amd/csrc/shaping.hpp) on one un-binned contig.  One step = the three
    steps on rows already resident in HBM (`smcpp_dev_shape` mode 3: HIP events on the stream the kernels run on); `roofline`:
    bytes every step must read and write / that time against 8 TB/s; `cpu_baseline`: this repository's numpy implementation of the
    same functions (`smcpp_amd.data`, pinned bit for bit against the reference's Cython by golden G23) on a bounded prefix."""
    from smcpp_amd import data as D, synth
    if world != 1:
        raise SystemExit("--workload shaping is a single-GPU measurement")
    raw = np.ascontiguousarray(synth.synth_posterior_contig(1_000_000, n, seed=7), dtype=np.int32)
    thinning, w, na = 400, 100, [2]
    P = int(raw[:, 0].astype(np.int64).sum())
    for _ in range(max(1, args.warmup)):
        out, _ = D.thin_bin_compress_device(raw, thinning, w, na, timing=True)
    ms, walls = [], []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        out, k_ms = D.thin_bin_compress_device(raw, thinning, w, na, timing=True)
        walls.append(time.perf_counter() - t0)
        ms.append(k_ms)
    k_ms = float(np.median(ms))
    t, ms_t = D.thin_data_device(raw, thinning, timing=True)
    b, ms_b = D.bin_observations_device(t, w, na, timing=True)
    c, ms_c = D.compress_repeated_obs_device(b, timing=True)
    assert np.array_equal(c, out)
    row_b = 4 * raw.shape[1]
    # what each step must move: its input rows once, its output rows once, and the 8-byte prefix sums it scans (written + read)
    alg = (len(raw) * (row_b + 16 + 16) + len(t) * row_b) + (len(t) * (row_b + 16) + len(b) * row_b) + (len(b) * (row_b + 16 + 16) + len(c) * row_b)
    gbs = alg / (1e-3 * k_ms) / 1e9
    # parity in the run: the host implementation on a prefix (the whole contig takes minutes of Python loops)
    pre = raw[:20_000]
    t0 = time.perf_counter()
    c_host = D.compress_repeated_obs(D.bin_observations(D.thin_data(pre, thinning), w, na))
    host_s = time.perf_counter() - t0
    c_dev = D.thin_bin_compress_device(pre, thinning, w, na)
    Ppre = int(pre[:, 0].astype(np.int64).sum())
    out = {"metric": f"Mbp/s ({desc})", "value": P / 1e6 / (1e-3 * k_ms), "unit": "Mbp/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": k_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 rows / int64 prefix sums", "data": "synthetic",
           "config": {"workload": desc, "rows_in": int(len(raw)), "positions": P, "rows_after_thin": int(len(t)), "rows_after_bin": int(len(b)),
                      "rows_out": int(len(c)), "thinning": thinning, "w": w, "timed": "device work with the input resident in HBM (HIP events, "
                      "smcpp_dev_shape mode 3)", "wall_ms_with_pcie_in_and_out": 1e3 * float(np.median(walls))},
           "split_ms": {"thin": ms_t, "bin": ms_b, "compress": ms_c, "pipeline": k_ms},
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "k_scan_* + k_thin_emit + k_bin_emit + k_compress_emit (all launches of one pipeline)", "kernel_ms_per_step": k_ms,
                        "algorithmic_bytes": alg,
                        "note": "integer, HBM-bound: bytes = rows in + rows out of every step + the 8-byte prefix sums it scans; the steps are "
                                "launch- and latency-bound at this size (a dozen small kernels, binary searches), far from the HBM roof"},
           "cpu_baseline": {"value": Ppre / 1e6 / host_s, "unit": "Mbp/s", "cores": 1, "kind": "port",
                            "sample": f"smcpp_amd.data (numpy / Python loops; bit-exact with the reference's Cython: golden G23) on the first {len(pre)} rows "
                                      f"({Ppre} bp) in {host_s:.1f} s; the reference's compiled Cython cannot be built on the GPU box",
                            "identical_to_device_on_the_sample": bool(np.array_equal(c_host, c_dev))}}
    out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)


def isa_mix(npl, hybrid, h32):
    """Class mix of the loops of the k_chain_ss instantiation a workload launches, from the newest committed
    profiles/r0*_isa_mix.json (tools/isa_mix.py on the shipped code object); None when there is none."""
    try:
        import glob
        pf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]*_isa_mix.json")))[-1]
        ks = json.load(open(pf))["kernels"]
        tag = "k_chain_ssILi%dELb%dE" % (npl, 1 if hybrid else 0)
        cands = [(k, v) for k, v in ks.items() if tag in k and k.split(tag)[1].startswith("Lb1E") and ("Lb1EEE" in k) == bool(h32)]
        cands = cands or [(k, v) for k, v in ks.items() if tag in k]
        k, v = cands[0]
        return {"source": os.path.relpath(pf, ROOT), "instantiation": k, "by_class": v["by_class"],
                "mean_issue_cycles_per_valu": v["mean_issue_cycles_per_valu"], "peak_ginstr_per_s": v["peak_ginstr_per_s"]}
    except Exception:  # noqa: BLE001
        return None


def sq_counters(workload, kname):
    """SQ counter totals of `kname` per E-step from the newest committed profiles/r0*_<workload>_sq_counters.json (rocprofv3
    --pmc passes of this same command, tools/pmc_sq_counters.sh + tools/summarize_sq.py); None when there is none."""
    if workload is None:
        return None
    try:
        import glob
        pf = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r0[0-9]*_{workload}_sq_counters.json")))[-1]
        prof = json.load(open(pf))
        tot = {}
        for name, v in prof["kernels"].items():
            if kname in name:
                for c, x in v["per_step"].items():
                    tot[c] = tot.get(c, 0.0) + float(x)
        if "SQ_INSTS_VALU" not in tot:
            return None
        out = {"source": "committed rocprofv3 --pmc passes of this command, not measured in this run: " + os.path.relpath(pf, ROOT),
               "SQ_INSTS_VALU_per_step": tot["SQ_INSTS_VALU"]}
        for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                  "SQ_INSTS_SALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
            if c in tot:
                out[c + "_per_step"] = tot[c]
        if tot.get("SQ_WAVE_CYCLES"):
            # SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md): ratios are unit-free
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                if c in tot:
                    out[c + "_over_WAVE_CYCLES"] = tot[c] / tot["SQ_WAVE_CYCLES"]
            out["wave_cycles_per_valu_instr"] = 4.0 * tot["SQ_WAVE_CYCLES"] / tot["SQ_INSTS_VALU"]
        return out
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(raw, model_args, obs, budget_s, raw_only, engine_on_prefix=None, n_contigs=1):
    """The reference's own C++ on one host core (oracle/_ref, compiled from /root/reference/src in the build
    container): its cold preparation (`ref_prep`: rate function, transition, conditioned SFS — what setParams +
    do_dirty_work recompute, src/inference_manager.cpp:213-229) timed in full, plus `HMM::Estep` on a prefix of the same
    contig sized to ~budget_s seconds and scaled by row count (the reference's E-step cost is linear in rows: one
    thread per contig, inference_manager.cpp:89-94)."""
    try:
        from oracle import ref
        kind = "reference" if ref.available() else "port"
        if kind == "port":
            from oracle import oracle as orc
        pi, T, keys, E = raw
        prep_s = 0.0
        if kind == "reference" and not raw_only and model_args is not None:
            args = model_args
            ref.prep(*args)                                   # first call builds the n-only tables (cached, as in the reference)
            ts = []
            for _ in range(3):
                t = time.perf_counter(); ref.prep(*args); ts.append(time.perf_counter() - t)
            prep_s = float(np.median(ts))
        probe = min(len(obs), 2000)
        fn = ref.estep if kind == "reference" else orc.estep
        fn(pi, T, keys, E, obs[:probe])                      # first call: library load, page-in
        t = time.perf_counter()
        fn(pi, T, keys, E, obs[:probe])
        per_row = (time.perf_counter() - t) / probe
        rows = int(min(len(obs), max(probe, budget_s / per_row)))
        t = time.perf_counter()
        r = fn(pi, T, keys, E, obs[:rows])
        dt = time.perf_counter() - t
        full = dt * len(obs) / rows
        parity = {}
        if engine_on_prefix is not None:
            try:
                ll_e = engine_on_prefix(obs[:rows])
                parity = {"loglik_prefix_engine": ll_e, "rel_diff_loglik": abs(ll_e - r["loglik"]) / abs(r["loglik"]),
                          "rel_diff_note": "engine (default chain family, set_raw with the same prepared parameters) vs the compiled "
                                           "reference on the SAME prefix rows, in this run; north_star bar 1e-6"}
            except Exception as ex:  # noqa: BLE001
                parity = {"rel_diff_error": repr(ex)}
        ref_threads = min(n_contigs, os.cpu_count() or 1)
        calib = {}
        if kind == "port":
            # the C restatement is SLOWER than the compiled reference (measured in the build container)
            calib = {"port_vs_reference_time_ratio": {"M32_n10": 1.24, "M64_n20": 1.11, "source": "BASELINE.md section 5"}}
        return {"value": 1.0 / (full + prep_s), "unit": "evals/s", "cores": 1, "kind": kind, **calib, **parity,
                "reference_threads_note": f"the reference runs one OpenMP thread per contig (src/inference_manager.cpp:89-94): on this "
                                          f"workload it would use {ref_threads} thread(s); the baseline is timed on 1 and scaled by rows, "
                                          f"i.e. with perfect scaling over its threads the reference would reach {ref_threads} x `value`",
                "sample": f"reference cold preparation timed in full ({1e3 * prep_s:.1f} ms) + HMM::Estep on the first "
                          f"{rows} of {len(obs)} rows of the same contig(s) in {dt:.1f} s, scaled by row count "
                          f"(1 thread = 1 contig, as the reference parallelises); host has {os.cpu_count()} cores",
                "prep_ms": 1e3 * prep_s, "estep_s_scaled": full, "us_per_row": 1e6 * dt / rows,
                "loglik_prefix": r["loglik"]}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "error": repr(e)}


if __name__ == "__main__":
    main()
