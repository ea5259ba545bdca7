"""The N > 1 path on the GPU engine: `smcpp_amd.dist.ShardedInferenceManager` with two ranks
sharing the one device of the test box (gloo reduction on the host; the measured configuration reduces the same
packed buffer over RCCL) against the single-rank manager holding every contig, plus the boundary fixes that belong
to it (reduced Q over keys a rank does not hold, buffer sizing of the gamma getter, getters before the first E-step)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _contigs(disjoint):
    """Five ragged contigs; with `disjoint`, contig 0 holds only rows with b == 0 and contig 1 only rows with b > 0 or
    a == -1 (plus the reduced keys), so the two ranks' key dictionaries differ in both directions."""
    from smcpp_amd import synth
    cs = [synth.synth_contig(50 + i, L, 10) for i, L in enumerate([700_000, 650_000, 90_000, 300_000, 60_000])]
    if disjoint:
        c0, c1 = cs[0], cs[1]
        cs[0] = np.ascontiguousarray(c0[(c0[:, 2] == 0) & (c0[:, 1] >= 0)])
        cs[1] = np.ascontiguousarray(c1[(c1[:, 2] > 0) | (c1[:, 3] == 0)])
        cs = cs[:2]
    return cs


def _setup(im, g, use_model):
    from smcpp_amd.model import PiecewiseModel
    im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
    if use_model:
        m = PiecewiseModel(g["a"], g["s"], 1e4, "pop1")
        m.differentiable = True
        im.model = m
    else:
        im.set_raw(g["pi"], g["T"], g["keys"], g["E"])


def _worker(rank, world, port, out_dir, disjoint, use_model):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smcpp_amd import dist as sd
    g = load_golden("G3_M32_n10_2Mbp")
    contigs = _contigs(disjoint)
    sim = sd.ShardedInferenceManager(10, contigs, g["hs"], ("pop1",), 0.5, device=0)
    _setup(sim, g, use_model)
    sim.E_step()
    res = dict(rank=rank, mine=[int(i) for i in sim.mine], loglik=sim.loglik(), logliks=list(map(float, sim.logliks())),
               q=list(map(float, sim.Q(separate=True))), local_keys=sim.im.keys.tolist(), keys=sim.keys.tolist())
    if use_model:
        q, jac = sim.Q_with_gradient()
        res["jac"] = jac.tolist()
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("disjoint,use_model", [(False, False), (False, True), (True, True), (True, False)])
def test_two_ranks_match_single_manager(tmp_path, disjoint, use_model):
    from smcpp_amd import _smcpp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), disjoint, use_model), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"r{i}.json")) for i in range(world)]
    g = load_golden("G3_M32_n10_2Mbp")
    contigs = _contigs(disjoint)
    im = _smcpp.PyOnePopInferenceManager(10, contigs, g["hs"], ("pop1",), 0.5)
    _setup(im, g, use_model)
    im.E_step()
    assert sorted(r[0]["mine"] + r[1]["mine"]) == list(range(len(contigs)))
    if disjoint:                                     # each rank lacks keys the other holds
        k0, k1 = set(map(tuple, r[0]["local_keys"])), set(map(tuple, r[1]["local_keys"]))
        assert k0 - k1 and k1 - k0
    assert r[0]["keys"] == r[1]["keys"] == im.keys.tolist()
    # a contig's chunk layout (hence where the chunk-parallel fixed point stops, within its tolerance: rows near a chunk start
    # carry up to eps = 2e-6 / 1e-6 of relative error in alpha / beta) depends on what else its manager holds: the sharded run and
    # the single manager agree to that tolerance, not bitwise (observed 2e-10 on the log-likelihood)
    for x in r:
        assert abs(x["loglik"] - im.loglik()) <= 1e-9 * abs(im.loglik())
        np.testing.assert_allclose(x["logliks"], im.logliks(), rtol=1e-8)   # (per contig: observed 1.1e-9; parity bar 1e-6)
    # every rank evaluates Q on the same reduced statistics: bitwise equal across ranks
    assert r[0]["q"] == r[1]["q"]
    np.testing.assert_allclose(r[0]["q"], im.Q(separate=True), rtol=1e-7)
    if use_model:
        assert r[0]["jac"] == r[1]["jac"]
        _, jac = im.Q_with_gradient()
        np.testing.assert_allclose(np.array(r[0]["jac"]), jac, rtol=1e-6, atol=1e-7 * np.abs(jac).max())


def _exact_worker(rank, world, port, out_dir, backend):
    """Two ranks; keeps the packed statistics before and after the all-reduce (gloo: ranks share device 0; nccl: one device each)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from smcpp_amd import dist as sd
    g = load_golden("G3_M32_n10_2Mbp")
    contigs = _contigs(True)
    sim = sd.ShardedInferenceManager(10, contigs, g["hs"], ("pop1",), 0.5, device=dev)
    sim.keep_stats = True
    _setup(sim, g, True)
    sim.E_step()
    q, jac = sim.Q_with_gradient()
    extra = {}
    if backend == "nccl":
        # the same exchange issued by the engine through RCCL's C API (own communicator): bitwise the same reduced buffer and Q
        sim2 = sd.ShardedInferenceManager(10, contigs, g["hs"], ("pop1",), 0.5, device=dev, direct_rccl=True)
        assert sim2._direct
        _setup(sim2, g, True)
        sim2.E_step()
        extra = dict(direct_reduced=sim2.im.rccl_fetch(), direct_q=np.array(sim2.Q(separate=True)), direct_ll=np.array(sim2.loglik()),
                     ll=np.array(sim.loglik()), q4=np.array(sim.Q(separate=True)))
    np.savez(os.path.join(out_dir, f"x{rank}.npz"), local=sim.last_local_stats, reduced=sim.last_reduced_stats, mine=np.array(sim.mine),
             q=q, jac=jac, nccl=np.array(sim._nccl), **extra)
    dist.barrier()
    dist.destroy_process_group()


def _check_exact(tmp_path, world):
    from smcpp_amd import _smcpp
    r = [np.load(tmp_path / f"x{i}.npz") for i in range(world)]
    g = load_golden("G3_M32_n10_2Mbp")
    contigs = _contigs(True)
    # (1) the collective: the reduced buffer is the sum of the ranks' local buffers - exactly (two summands: any order, one rounding)
    assert np.array_equal(r[0]["reduced"], r[1]["reduced"])
    assert np.array_equal(r[0]["reduced"], r[0]["local"] + r[1]["local"])
    # (2) a rank's local buffer is what a SINGLE manager over the same contigs packs - bitwise: the same contigs give the same chunk
    # layout, and every kernel reduces in a fixed order (the sharded-vs-one-manager comparison above can only hold to the fixed
    # point's tolerance, because one manager over ALL contigs cuts its chunks differently)
    for i in range(world):
        im = _smcpp.PyOnePopInferenceManager(10, [contigs[c] for c in r[i]["mine"]], g["hs"], ("pop1",), 0.5)
        _setup(im, g, True)
        im.E_step()
        im.set_global_keys(np.unique(np.vstack([c[:, 1:] for c in contigs]), axis=0).astype(np.int32))
        assert np.array_equal(im.pack_stats(), r[i]["local"]), i
    # (3) Q and its gradient on the reduced statistics: bitwise equal on every rank
    assert np.array_equal(r[0]["q"], r[1]["q"]) and np.array_equal(r[0]["jac"], r[1]["jac"])


def test_reduction_is_exact_two_ranks_one_device(tmp_path):
    """a tight check of the reduction itself, independent of the chunk layout (gloo, ranks share the device)."""
    mp.spawn(_exact_worker, args=(2, _free_port(), str(tmp_path), "gloo"), nprocs=2, join=True)
    _check_exact(tmp_path, 2)


def test_reduction_is_exact_two_ranks_rccl(tmp_path):
    """The same over RCCL (backend nccl), one rank per GPU: runs wherever two devices are visible (the 1-GPU test box skips)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    mp.spawn(_exact_worker, args=(2, _free_port(), str(tmp_path), "nccl"), nprocs=2, join=True)
    _check_exact(tmp_path, 2)
    assert bool(np.load(tmp_path / "x0.npz")["nccl"])
    for i in range(2):
        r = np.load(tmp_path / f"x{i}.npz")
        assert np.array_equal(r["direct_reduced"], r["reduced"]) and np.array_equal(r["direct_q"], r["q4"]) and r["direct_ll"] == r["ll"]


def _nccl_worker(rank, port, out_dir):
    """ONE rank, backend nccl (= RCCL): the device-buffer branch of ShardedInferenceManager.E_step (k_pack_stats writes into
    the reduced tensor, all_reduce on the device, k_unpack) that a gloo group never takes."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from smcpp_amd.dist import ShardedInferenceManager
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1)
    g = load_golden("G3_M32_n10_2Mbp")
    contigs = _contigs(True)
    sim = ShardedInferenceManager(10, contigs, g["hs"], ("pop1",), 0.5, always_reduce=True)
    assert sim._nccl and sim._reduce
    _setup(sim, g, True)
    sim.E_step()
    q, jac = sim.Q_with_gradient()
    res = dict(loglik=sim.loglik(), logliks=list(map(float, sim.logliks())), q=list(map(float, sim.Q(separate=True))),
               jac=jac.tolist(), buf_device=str(sim._buf.device), keys=sim.keys.tolist(), stream_ordered=sim._ext is not None)
    # the same exchange with a host wait between the pack kernel and the collective (round 4's form): bitwise the same buffer
    buf_a = sim._buf.cpu().numpy().copy()
    sim.stream_ordered = False; sim._buf = None
    sim.E_step()
    res["host_wait_same_buffer"] = bool(np.array_equal(buf_a, sim._buf.cpu().numpy())) and sim._ext is None
    res["host_wait_loglik"] = sim.loglik()
    # what the exchange adds to an eval (pack kernel + one-rank RCCL all-reduce + scalar read-back), both orderings
    import time

    def per_eval(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t) / reps
    local_us = per_eval(lambda: (sim.im.E_step(), sim.im.loglik()))
    wait_us = per_eval(lambda: (sim.E_step(), sim.loglik()))
    sim.stream_ordered = True; sim._buf = None
    ordered_us = per_eval(lambda: (sim.E_step(), sim.loglik()))
    # ... and issued by the engine itself through RCCL's C API (smcpp_rccl_*: own communicator, the engine's stream, polled scalar)
    sim2 = ShardedInferenceManager(10, contigs, g["hs"], ("pop1",), 0.5, always_reduce=True, direct_rccl=True)
    assert sim2._direct
    _setup(sim2, g, True)
    sim2.E_step()
    res["direct_loglik"] = sim2.loglik()
    res["direct_same_buffer"] = bool(np.array_equal(buf_a, sim2.im.rccl_fetch()))
    res["direct_q"] = list(map(float, sim2.Q(separate=True)))
    direct_us = per_eval(lambda: (sim2.E_step(), sim2.loglik()))
    res["exchange_us"] = dict(local_eval=local_us, host_wait=wait_us - local_us, stream_ordered=ordered_us - local_us,
                              direct_rccl=direct_us - local_us)
    print("nccl world of one: exchange cost per eval (us)", res["exchange_us"], flush=True)
    with open(os.path.join(out_dir, "nccl.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_nccl_device_buffer_branch_world_of_one(tmp_path):
    """The RCCL path of the sharded manager (pack on the device -> all_reduce -> unpack on the device) executed for real: a
    process group of ONE rank with backend nccl and `always_reduce`.  Against the plain manager on the same contigs: the
    reduced statistics are the manager's own, summed over its contigs by k_pack_stats (1e-12: summation order)."""
    from smcpp_amd import _smcpp
    mp.spawn(_nccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = json.load(open(tmp_path / "nccl.json"))
    g = load_golden("G3_M32_n10_2Mbp")
    contigs = _contigs(True)
    im = _smcpp.PyOnePopInferenceManager(10, contigs, g["hs"], ("pop1",), 0.5)
    _setup(im, g, True)
    im.E_step()
    assert r["buf_device"].startswith("cuda")
    assert r["stream_ordered"] and r["host_wait_same_buffer"] and r["host_wait_loglik"] == r["loglik"]
    print("exchange cost per eval (us):", r["exchange_us"])
    assert r["direct_same_buffer"] and r["direct_loglik"] == r["loglik"] and r["direct_q"] == r["q"]
    assert r["keys"] == im.keys.tolist()
    assert abs(r["loglik"] - im.loglik()) <= 1e-12 * abs(im.loglik())
    np.testing.assert_allclose(r["logliks"], im.logliks(), rtol=1e-13)
    np.testing.assert_allclose(r["q"], im.Q(separate=True), rtol=1e-11)
    _, jac = im.Q_with_gradient()
    np.testing.assert_allclose(np.array(r["jac"]), jac, rtol=1e-9, atol=1e-9 * np.abs(jac).max())


def test_reduced_q_needs_every_global_key():
    """set_raw on a rank that was not given the emission vector of a key other ranks hold: Q must fail loudly instead
    of silently dropping the key's statistics."""
    from smcpp_amd import _smcpp, synth
    g = load_golden("G3_M32_n10_2Mbp")
    c = synth.synth_contig(50, 300_000, 10)
    sub = np.ascontiguousarray(c[(c[:, 2] == 0) & (c[:, 1] >= 0)])
    im = _smcpp.PyOnePopInferenceManager(10, [sub], g["hs"], ("pop1",), 0.5)
    im.theta = float(g["theta"]); im.rho = float(g["rho"])
    loc = set(map(tuple, im.keys.tolist()))
    sel = [i for i, k in enumerate(g["keys"].tolist()) if tuple(k) in loc]
    im.set_raw(g["pi"], g["T"], g["keys"][sel], g["E"][sel])          # only this rank's keys
    im.E_step()
    im.set_global_keys(g["keys"])
    buf = im.pack_stats()
    M, Kg = 32, len(g["keys"])
    other = [i for i in range(Kg) if i not in sel][0]
    buf[1 + M + M * M + other * M: 1 + M + M * M + (other + 1) * M] = 1.0   # "another rank" saw that key
    im.unpack_stats(buf)
    with pytest.raises(RuntimeError, match="no emission vector"):
        im.Q(separate=True)
    im.set_raw(g["pi"], g["T"], g["keys"], g["E"])                     # every global key supplied: fine
    im.E_step()
    im.unpack_stats(buf)
    assert np.all(np.isfinite(im.Q(separate=True)))


def test_gamma_getter_is_sized_from_the_last_estep():
    """Toggling save_gamma after an E-step must not change what the getter returns (heap overflow)."""
    from smcpp_amd import _smcpp
    g = load_golden("G1_M16_n4")
    im = _smcpp.PyOnePopInferenceManager(4, [g["obs"]], g["hs"], ("p",), 0.5)
    # before the first E-step the getters return what a freshly constructed reference HMM holds (hmm.cpp:8-29): zeros,
    # and per key the positions it covers weighted by the constant-size default model's initial distribution
    assert np.all(im.xisums[0] == 0) and im.loglik() == 0.0 and np.all(im.gammas[0] == 0) and im.gammas[0].shape == (16, 1)
    hs = g["hs"]
    pi0 = np.exp(-hs[:-1]) - np.r_[np.exp(-hs[1:-1]), 0.0]
    pi0 /= pi0.sum()
    gs = im.gamma_sums[0]
    for k, v in gs.items():
        span = g["obs"][np.all(g["obs"][:, 1:] == np.array(k), axis=1), 0].sum()
        np.testing.assert_allclose(v, span * pi0, rtol=1e-13)
    im.theta = float(g["theta"]); im.rho = float(g["rho"])
    im.set_raw(g["pi"], g["T"], g["keys"], g["E"])
    q0 = np.array(im.Q(separate=True))
    assert q0[0] == 0 and q0[3] == 0 and q0[1] < 0 and q0[2] < 0
    im.save_gamma = True
    im.E_step()
    full = im.gammas[0]
    assert full.shape == (16, len(g["obs"]) + 1)
    im.save_gamma = False
    again = im.gammas[0]                              # still the stored M x (L+1) matrix
    assert again.shape == full.shape and np.array_equal(again, full)
    im.E_step()
    assert im.gammas[0].shape == (16, 1)
    im.save_gamma = True
    assert im.gammas[0].shape == (16, 1)              # flag toggled, nothing stored yet


def test_c5_shape_m256_n50_vs_oracle():
    """Config C5 at its real shape: M = 256, n = 50 (K ~ 100 keys), parameters from the engine's own preparation
    (checked against the compiled reference's table in tests/test_prep.py), 250 kbp against the C restatement."""
    from oracle import oracle
    from smcpp_amd import _smcpp, synth
    from smcpp_amd.model import PiecewiseModel
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "params_M256_n50.npz")))
    full = synth.synth_contig(0, 100_000_000, 50)               # the C5 contig (same generator, same seed)
    head = full[:600]                                           # its first ~250 kbp ...
    assert head[:, 0].sum() * 100 >= 200_000
    # ... followed by one row of every key the first 40 000 rows hold, so that the emission table is exercised at the
    # C5 key count without an oracle run over megabases (the C restatement costs ~0.1 s per eigen row at M = 256)
    seen = {tuple(r) for r in head[:, 1:].tolist()}
    extra = []
    for r in full[600:40_000].tolist():
        if tuple(r[1:]) not in seen:
            seen.add(tuple(r[1:])); extra.append(r)
    obs = np.ascontiguousarray(np.vstack([head, np.array(extra, dtype=np.int32).reshape(-1, 4)]), dtype=np.int32)
    im = _smcpp.PyOnePopInferenceManager(50, [obs], g["hs"], ("pop1",), float(g["pol"]))
    im.model = PiecewiseModel(g["a"], g["s"], 1e4, "pop1")
    im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
    im.set_chunking(200)                                       # 4 chunks: the chunk-parallel iteration is exercised
    im.E_step()
    keys = im.keys
    assert len(keys) >= 80
    ep = im.emission_probs
    Etab = np.array([ep[tuple(k)] for k in keys.tolist()])
    ref_E = {tuple(int(x) for x in k): e for k, e in zip(g["keys"], g["E"])}
    for k, e in zip(keys.tolist(), Etab):
        np.testing.assert_allclose(e, ref_E[tuple(k)], rtol=5e-14, atol=1e-18)
    np.testing.assert_allclose(im.transition, g["T"], rtol=1e-10, atol=1e-17)
    o = oracle.estep(im.pi, im.transition, keys, Etab, obs)
    assert abs(im.loglik() - o["loglik"]) <= 1e-6 * abs(o["loglik"])
    xs = im.xisums[0]
    assert np.max(np.abs(xs - o["xisum"]) / np.maximum(np.abs(o["xisum"]), 1e-300)) <= 5e-6
    for k, v in o["gamma_sums"].items():
        assert np.max(np.abs(im.gamma_sums[0][k] - v)) <= 5e-6 * max(np.abs(v).max(), 1e-300)
    q = np.array(im.Q(separate=True))
    assert np.all(np.abs(q - o["q"]) <= 5e-6 * np.maximum(np.abs(o["q"]), 1e-12))


def test_bench_gpus_2_self_launches():
    """`python bench.py --gpus 2` without a torchrun environment must start two ranks itself and report n_gpus = 2
    (on this one-GPU box the ranks share the device and reduce over gloo, which the output flags)."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--length-mbp", "10", "--no-cpu"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "set_params" in out["config"]["eval"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--length-mbp", "10", "--no-cpu"], capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    # rank 0 of the two-rank run holds the same contig as the single-rank run; the reduced log-likelihood adds rank 1's
    assert out["config"]["loglik"] < o1["config"]["loglik"] < 0


@pytest.mark.parametrize("workload,extra", [("c3", []), ("c4", ["--length-mbp", "20"])])
def test_bench_eight_ranks_dry_run(workload, extra):
    """`bench.py --gpus 8 --workload c3 | c4 --check` as the driver launches it on an 8-GPU node, here with the eight ranks
    sharing the one device over gloo (flagged in the output: a functional run, not a measurement): the contigs are partitioned,
    every rank holds the same global key dictionary and - after the single all-reduce - bitwise the same Q; each rank's host
    threads stay within its share of the CPU quota.  c3 = the 22 contigs of the whole genome LPT-sharded (strong scaling; also
    checked against the compiled reference's recorded per-contig log-likelihoods, golden G16); c4 = two-population managers built
    by ShardedInferenceManager's own factory."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", workload, "--steps", "2",
                        "--warmup", "1", "--no-cpu", "--check"] + extra, capture_output=True, text=True, env=env, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["value"] > 0
    mc = out["multi_gpu_check"]
    assert mc["ranks"] == 8 and mc["q_bitwise_identical"]
    assert all(1 <= t <= max(1, mc["cpu_quota"] // 8) for t in mc["host_threads_per_rank"])
    assert sorted(c for r in mc["contigs_per_rank"] for c in r) == list(range(22 if workload == "c3" else 8))
    if workload == "c3":
        assert out["scaling"] == "strong" and abs(out["parity_full_size"]["rel_diff"]) <= 1e-6
    else:
        assert out["scaling"] == "weak" and mc["global_keys"] >= 100


def test_c4_real_shape_two_population_model_path():
    """Config C4 at its real shape through `im.model = TwoPopulationModel(...)`: M = 48, n1 = n2 = 10, a = (2, 0),
    split 0.5, 6-int keys.  The emission table the engine prepares is compared with golden G13 (reference jcsfs.py +
    compiled reference C++), and the E-step with the C restatement of hmm.cpp fed with the FIXTURE's parameters (not the
    engine's own), on the first 12 000 rows of the C4 contig."""
    from oracle import oracle
    from smcpp_amd import _smcpp, synth
    from smcpp_amd.model import PiecewiseModel, TwoPopulationModel
    g = np.load(os.path.join(ROOT, "tests", "golden", "G13_c4_params.npz"))
    obs = np.ascontiguousarray(synth.synth_contig_twopop(0, 100_000_000, 10, 10)[:12_000])
    im = _smcpp.PyTwoPopInferenceManager(10, 10, 2, 0, [obs], g["hs"], ("pop1", "pop2"), float(g["pol"]))
    im.model = TwoPopulationModel(PiecewiseModel(g["a1"], g["s1"], 1e4, pid="pop1"),
                                  PiecewiseModel(g["a2"], g["s2"], 1e4, pid="pop2"), float(g["split"]))
    im.theta = float(g["theta"]); im.rho = float(g["rho"]); im.alpha = float(g["alpha"])
    im.E_step()
    ref_E = {tuple(int(x) for x in k): e for k, e in zip(g["keys"], g["E"])}
    ep = im.emission_probs
    keys = im.keys
    assert keys.shape[1] == 6 and len(keys) >= 100
    for k in keys.tolist():
        np.testing.assert_allclose(ep[tuple(k)], ref_E[tuple(k)], rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(im.pi, g["pi"], rtol=1e-13)
    np.testing.assert_allclose(im.transition, g["T"], rtol=1e-11, atol=1e-17)
    Etab = np.array([ref_E[tuple(k)] for k in keys.tolist()])
    o = oracle.estep(g["pi"], g["T"], keys, Etab, obs)
    assert abs(im.loglik() - o["loglik"]) <= 1e-6 * abs(o["loglik"])
    xs = im.xisums[0]
    assert np.max(np.abs(xs - o["xisum"]) / np.maximum(np.abs(o["xisum"]), 1e-300)) <= 5e-6
    for k, v in o["gamma_sums"].items():
        assert np.max(np.abs(im.gamma_sums[0][k] - v)) <= 5e-6 * max(np.abs(v).max(), 1e-300)
    q = np.array(im.Q(separate=True))
    assert np.all(np.abs(q - o["q"]) <= 5e-6 * np.maximum(np.abs(o["q"]), 1e-12))
    # round 5: the two batched conditioned-SFS problems of the joint CSFS run on the device (k_prep_csfs_raw) by default; the all-host
    # route (set_prep_mode(True)) forms the same table - the phases are the same code, the device library's exp / expm1 differ from
    # libm in the last ulp
    ll_dev = im.loglik()
    im.set_prep_mode(True)
    im.model = TwoPopulationModel(PiecewiseModel(g["a1"], g["s1"], 1e4, pid="pop1"),
                                  PiecewiseModel(g["a2"], g["s2"], 1e4, pid="pop2"), float(g["split"]))
    im.E_step()
    ep_host = im.emission_probs
    for k in keys.tolist():
        np.testing.assert_allclose(ep[tuple(k)], ep_host[tuple(k)], rtol=1e-11, atol=1e-16)
    assert abs(im.loglik() - ll_dev) <= 1e-10 * abs(ll_dev)
