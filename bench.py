#!/usr/bin/env python
"""bench.py -- PLDA LLR trials/sec (and fit-EM iters/sec) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.

Default workload (BASELINE.json configs[1], "C2", the configuration the metric is quoted on):
100 000 random i-vectors, featdim 200, 5 000 speakers; fit = statistics + 10 EM iterations +
GetOutput on the GPU (fp64); a "step" = one pass of the hot path over one batch = the
100k x 100k trials matrix (1e10 log-likelihood ratios, n = 1 per enrol model): fp64 bias terms,
fp64 -> fp32 operand packing, fp32-MFMA GEMM, 40 GB of fp32 scores written to HBM.  Inputs are
HBM-resident before the timed region; scores stay in HBM.

N > 1 (one process per GPU, torchrun): WEAK scaling by default -- the trials matrix grows with the job: N x 100k
enrol models against the same 100k tests, so every GPU scores as many trials per step as the single GPU does
(`--scaling strong` keeps the 100k x 100k matrix and splits it instead).  Every rank holds the replicated model,
enrol and test sets and calls the library's own sharded entry point `plda_score_matrix_sharded_local_dev`
(csrc/comm.hip; torch.distributed only carries the 128-byte unique id, the model broadcast and the timing
reduction): enrol rows are dealt out block-cyclically and every rank writes its blocks back to back into a
compact slab of M/N rows.
  value            = trials/s with the scores left row-sharded (no data-path collective: what
                     thresholding, counting, EER and z-norm consume);
  gather_inclusive = K steps of the 100k x 100k matrix with every block all-gathered in place over xGMI, on a side
                     stream, overlapped with the scoring of the following blocks (north_star's
                     "RCCL all-gather to assemble scores"): its volume -- (N-1)/N of 40 GB into every
                     rank -- not the GEMM bounds it, which is why it is reported beside `value`;
  multi_gpu        = who took part, as the TRANSPORT reports it (ncclCommCount / UserRank / CuDevice + PCI bus id of
                     every rank) and a cross-rank checksum of gathered row blocks against their owners' copies.
--config C3 / C4 run the other BASELINE shapes (C3: 10k models (n = 100) x 1M tests at D = 512;
C4: 40k models (n in 1..5) x 1.2M tests at D = 256, 192 GB of scores) through the same code.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 flop/clk
PEAK_FP64_MFMA_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64: half of that
CURRENT_ROUND = 6               # profiles/rNN_* files a bench line may cite

CONFIGS = {
    # name: fit rows, featdim, speakers, enrol models, enrol counts, test vectors
    "C2": dict(N=100000, D=200, K=5000, M=100000, counts=1, Nt=100000,
               what="C2: 100k i-vectors, featdim 200, 5k speakers; 100k x 100k trials, n = 1"),
    "C3": dict(N=1000000, D=512, K=10000, M=10000, counts=100, Nt=1000000,
               what="C3: 1M x-vectors, featdim 512, 10k speakers; 10k models (n = 100) x 1M tests"),
    "C4": dict(N=1200000, D=256, K=7200, M=40000, counts="1..5", Nt=1200000,
               what="C4: 1.2M utterances, 7.2k speakers, featdim 256; 40k models (n in 1..5) x 1.2M tests"),
}


def cpu_baseline(D, psi, seconds=12.0):
    """Faithful single-thread CPU restatement (oracle = "port") on a bounded sample of the
    same workload: per-trial Plda::LogLikelihoodRatio incl. the wrapper's per-call vector
    copies (pldamodule.cpp:258-277), n = 1, same D and psi as the GPU run."""
    from oracle import binding as ob
    ob.build()
    rng = np.random.default_rng(1234)
    m = 256
    U = rng.standard_normal((m, D)); V = rng.standard_normal((m, D))
    t0 = time.perf_counter(); ob.score_block(psi, U, 1, V); dt = time.perf_counter() - t0
    rate = m * m / dt
    side = int(max(256, min(6000, (rate * seconds) ** 0.5)))
    U = rng.standard_normal((side, D)); V = rng.standard_normal((side, D))
    t0 = time.perf_counter(); ob.score_block(psi, U, 1, V); dt = time.perf_counter() - t0
    return {"value": side * side / dt, "unit": "trials/s", "cores": 1, "kind": "port",
            "sample": "%dx%d trials, D=%d, n=1, oracle/plda_oracle.c per-trial LLR loop, %.1f s" % (side, side, D, dt)}


def cpu_best_effort(D, psi, side=12000):
    """the same LLR in batched GEMM form (NumPy restatement, fp64 BLAS on all
    host cores) -- context only; `cpu_baseline.value` stays the faithful per-trial path."""
    from oracle import plda_oracle_np as onp
    rng = np.random.default_rng(4321)
    U = rng.standard_normal((side, D)); V = rng.standard_normal((side, D))
    onp.llr_matrix(psi, U[:512], 1, V[:512])
    t0 = time.perf_counter(); onp.llr_matrix(psi, U, 1, V); dt = time.perf_counter() - t0
    return {"value": side * side / dt, "unit": "trials/s", "cores": os.cpu_count(),
            "sample": "%dx%d trials, fp64 GEMM form (oracle/plda_oracle_np.py), %.1f s" % (side, side, dt)}


def host_stats(X, y):
    """PldaStats::AddSamples with the wrapper's 1 / n_k weight (pldamodule.cpp:94-98) in NumPy fp64 (BLAS on the host
    cores): what `oracle.stats` computes with scalar loops -- used for the shapes where those loops would take minutes
    (C3: N D^2 = 2.6e11).  Returns the dict oracle.em_iter takes."""
    c = np.bincount(y).astype(np.int64)
    K = c.shape[0]
    order = np.argsort(y, kind="stable")
    m = np.add.reduceat(X[order], np.r_[0, np.cumsum(c)[:-1]], axis=0) / c[:, None]
    del order
    xs = X * np.sqrt(1.0 / c[y])[:, None]
    S = xs.T @ xs - m.T @ m
    del xs
    return dict(means=m, counts=c, scatter=S, sum=(m / c[:, None]).sum(0), class_weight=float((1.0 / c).sum()), example_weight=float(K))


def cpu_em_baseline(X, y, gpu_one_iter=None):
    """One Kaldi-style EM iteration (per-class loop, explicit inversions) of the oracle on
    the SAME statistics, single thread; its W, B are also the checker of the GPU's first
    iteration (`gpu_one_iter` = the engine's means / scatter / W / B after a 1-iteration fit)."""
    from oracle import binding as ob
    D = X.shape[1]
    big = float(X.shape[0]) * D * D > 2e10
    st = host_stats(X, y.astype(np.int64)) if big else ob.stats(X, y)
    t0 = time.perf_counter()
    W, B = ob.em_iter(st, np.eye(D), np.eye(D))
    dt = time.perf_counter() - t0
    res = {"em_iters_per_s": 1.0 / dt, "sample": "1 EM iteration at N=%d D=%d K=%d, %.1f s" % (X.shape[0], D, int(y.max()) + 1, dt),
           "statistics_by": "NumPy fp64 restatement of AddSamples (host BLAS)" if big else "oracle/plda_oracle.c"}
    if gpu_one_iter is not None:
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())    # noqa: E731
        res["gpu_vs_oracle_after_one_iteration"] = {
            "counts_equal": bool(np.array_equal(gpu_one_iter["counts"], st["counts"])),
            "means_rel_err": rel(gpu_one_iter["means"], st["means"]), "scatter_rel_err": rel(gpu_one_iter["scatter"], st["scatter"]),
            "W_rel_err": rel(gpu_one_iter["W"], W), "B_rel_err": rel(gpu_one_iter["B"], B)}
    return res


def skewed_labels(N, K, lo=5, hi=60, seed=2):
    """C2's rows labelled as real data is (BASELINE.md C2 "skewed-n_k variant"; the reason the reference sorts its classes by
    count, pldamodule.cpp:94-100): utterance counts drawn from [lo, hi], rescaled to N rows, the rounding's leftover one row each."""
    rng = np.random.default_rng(seed)
    nk = rng.integers(lo, hi + 1, K).astype(np.float64)
    nk = np.maximum(1, np.floor(nk * N / nk.sum())).astype(np.int64)
    while nk.sum() > N:                                  # (few rows per speaker: the floor of 1 overshoots; take from the largest)
        nk[np.argmax(nk)] -= 1
    nk[: N - nk.sum()] += 1
    return np.repeat(np.arange(K), nk).astype(np.uint64), nk


def fit_skewed(eng, dX, N, D, K, iters, X_host, with_cpu):
    """fit on C2's rows with unequal speaker counts (n_k in [5, 60]): the number of distinct counts G, which closed form of the
    grouped EM ran, statistics / EM / GetOutput ms, EM iterations/s, and W / B after ONE iteration against the oracle's."""
    import torch
    y, nk = skewed_labels(N, K)
    dy = torch.from_numpy(y.astype(np.int64)).to(dX.device)
    res = {"labels": "n_k drawn from [5, 60] and rescaled to the N rows (C2: 3 .. 38 per speaker), seed 2 (skewed_labels)", "distinct_counts_G": int(len(np.unique(nk))), "N": N, "D": D, "K": K}
    one = None
    if with_cpu:
        eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, 1)
        torch.cuda.synchronize(dX.device)
        one = eng.fit_internals()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, iters)
        torch.cuda.synchronize(dX.device)
        w = time.perf_counter() - t0
        ft = eng.fit_timings()
        if best is None or w < best[0]:
            best = (w, ft)
    w, ft = best
    plan = eng.fit_plan()
    res.update({"em_form": plan["form"], "groups": plan["groups"], "fit_wall_s": round(w, 5), "timing": "min of 3",
                "stats_ms": round(ft["stats_ms"], 3), "em_ms": round(ft["em_ms"], 3), "output_ms": round(ft["output_ms"], 3),
                "iters": ft["iters"], "em_ms_per_iter": round(ft["em_ms"] / max(ft["iters"], 1), 4),
                "em_iters_per_s": round(ft["iters"] / (ft["em_ms"] / 1e3), 2) if ft["em_ms"] > 0 else None})
    if with_cpu and X_host is not None:
        try:
            res["cpu_oracle"] = cpu_em_baseline(X_host, y, one)
        except Exception as e:   # noqa: BLE001 -- informative leg
            res["cpu_oracle"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def fit_large_dim(eng, dev, iters):
    """The reference's own large test shape (tests/pldatest.py:35-38: 10 000 x 1024 rows, 1 000 speakers of 10): fit wall clock,
    statistics / EM / GetOutput and the per-stage spans -- the tridiagonalisation's share of GetOutput at D = 1024."""
    import torch
    N, D, K = 10000, 1024, 1000
    g = torch.Generator(device=dev); g.manual_seed(5)
    dX = torch.rand((N, D), dtype=torch.float64, device=dev, generator=g)
    dy = torch.div(torch.arange(N, device=dev, dtype=torch.int64), 10, rounding_mode="floor")
    eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, iters)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, iters)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ft = eng.fit_timings()
    eng.trace_enable(True)
    eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, iters)
    torch.cuda.synchronize(dev)
    spans = eng.trace_read()
    eng.trace_enable(False)
    return {"shape": "tests/pldatest.py:35-38: 10000 x 1024, 1000 speakers of 10", "iters": iters, "fit_wall_ms": round(wall * 1e3, 3),
            "stats_ms": round(ft["stats_ms"], 3), "em_ms": round(ft["em_ms"], 3), "output_ms": round(ft["output_ms"], 3),
            "em_form": eng.fit_plan()["form"], "stages": [{"name": sp["name"], "ms": round(sp["ms"], 4)} for sp in spans]}


def end_to_end(eng, X, y, D, dout):
    """The reference user's view (SURVEY.md section 8d "also report end-to-end incl. H2D"): NumPy arrays in, NumPy
    arrays out through the drop-in API -- pageable host memory on both sides, PCIe inclusive.  Never `value`."""
    rng = np.random.default_rng(99)
    res = {}
    side = 20000
    E, T = rng.random((side, D)), rng.random((side, D))
    U, V = eng.transform_array(E, 1), eng.transform_array(T, 1)
    eng.score_matrix((1, U), (1, V), znorm=False)                   # warm: pinned ring, copy threads, code
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter(); S = eng.score_matrix((1, U), (1, V), znorm=False); best = min(best, time.perf_counter() - t0)
    res["score_matrix"] = {"shape": [side, side], "ms": round(best * 1e3, 2), "trials_per_s": side * side / best,
                           "bytes_to_host": side * side * 4, "d2h_GBps": round(side * side * 4 / best / 1e9, 1),
                           "how": "plda_score_matrix on NumPy arrays, fresh float32 output per call (page faults included): "
                                  "GEMM of slab i+1 | DMA of slab i into a pinned ring | host threads land slab i-1"}
    del S
    if X is not None:
        yy = y.astype(np.uint64)
        eng.fit(X, yy, 10)
        dt = 1e30
        for _ in range(3):        # min of 3, like score_matrix above (single samples scattered by 50 % between boxes)
            t0 = time.perf_counter(); eng.fit(X, yy, 10); dt = min(dt, time.perf_counter() - t0)
        res["fit"] = {"rows": int(X.shape[0]), "ms": round(dt * 1e3, 2), "h2d_bytes": int(X.nbytes), "timing": "min of 3",
                      "how": "liblda-style fit(X, y, 10) on NumPy arrays: label compaction on the host (counting, not sorting), upload through the pinned ring, statistics + EM + GetOutput"}
        eng.transform(X, yy)
        dt = 1e30
        for _ in range(3):
            t0 = time.perf_counter(); tr = eng.transform(X, yy); dt = min(dt, time.perf_counter() - t0)
        res["transform"] = {"rows": int(X.shape[0]), "labels": len(tr), "ms": round(dt * 1e3, 2), "timing": "min of 3",
                            "how": "transform(X, y) -> dict {label: (n, vec)} built without a per-key Python round trip"}
        ks = list(tr)[:64]
        t0 = time.perf_counter()
        for _ in range(40):
            for k in ks:
                eng.score(k, tr[k], tr[ks[0]])
        dt = (time.perf_counter() - t0) / (40 * len(ks))
        res["score_call"] = {"us_per_call": round(dt * 1e6, 2),
                             "how": "plda.score(id, (n, u), (n, v)) from Python: one LLR on the handle's host mirror of psi (plda_score_one)"}
    return res


def latest_traffic(M, Nt, dout):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary of THIS
    workload (profiles/rNN_traffic_trials_gemm.json, written by scripts/gpu_profile.sh); bench.py cannot
    collect counters itself."""
    import glob
    best = None
    # (files of the CURRENT round only: round 3's C3 line silently carried a round-2 counter file)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r%02d_*traffic_trials_gemm*.json" % CURRENT_ROUND))):
        try:
            j = json.load(open(f))
            if tuple(j.get("shape", (100000, 100000, 200))) == (M, Nt, dout):
                best = (float(j["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch; kernel %s)" % j.get("kernel", "?"))
        except Exception:
            pass
    return best if best else (None, None)


def describe_ranks(descs, scores_numel=0):
    """`multi_gpu` of the printed line from the ranks' own plda_comm_describe answers.  `comm_nranks` is what the transport in use
    reports; `rccl_nranks` / `rccl_version` appear ONLY when that transport is RCCL (then they come from ncclCommCount /
    ncclGetVersion inside the library) -- a peer / host / custom line must not carry a key that reads as "RCCL saw N ranks"."""
    d0 = descs[0]
    multi = {"transport": d0["transport"], "comm_nranks": d0["nranks"],
             "ranks": [{"rank": d_["rank"], "device": d_["device"], "pci_bus_id": d_["pci_bus_id"]} for d_ in descs],
             "distinct_devices": len({d_["pci_bus_id"] for d_ in descs}),
             "scores_per_rank_GB": round(scores_numel * 4 / 1e9, 2)}
    if d0["transport"] == "rccl":
        multi["rccl_nranks"] = d0["nranks"]
        multi["rccl_version"] = d0.get("rccl_version")
    return multi


def launch_command(n, argv, port=None):
    """The command line `python bench.py --gpus N ...` turns itself into when no launcher set WORLD_SIZE: the form the
    driver uses for N > 1 (one process per GPU, rendezvous on 127.0.0.1)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, transport="rccl"):
    import subprocess
    if transport == "rccl":
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception:       # noqa: BLE001
            have = 0
        if have < n:
            sys.stderr.write("bench.py: --gpus %d needs %d GPUs for the RCCL transport, this node shows %d "
                             "(--transport host|peer with --backend gloo lets ranks share a GPU: a logic check)\n" % (n, n, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.call(launch_command(n, argv), env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="BASELINE.json configuration (default: the metric's)")
    ap.add_argument("--rows", dest="n", type=int, default=0, help="override: fit rows = enrol rows = test rows (C2 shape)")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--speakers", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--targetdim", type=int, default=0, help="build extension: keep the top-psi dims (0 = all)")
    ap.add_argument("--block-rows", type=int, default=0,
                    help="N>1: rows per block of the block-cyclic row partition of the sharded leg (0 = one block per rank: "
                         "ceil(M / N) rounded up to 256); the gather leg always uses 4096-row blocks (overlap granularity)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the targetdim-150 extra measurement (profiling runs: every launch of the trials kernel "
                         "is then the timed workload)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = N x the enrol rows (per-GPU work fixed, the default); strong = the same matrix split N ways")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the gather-inclusive second timed region")
    ap.add_argument("--shard-fit", action="store_true",
                    help="N>1: shard the fit statistics by speaker through plda_fit_sharded_dev (all-reduce of the "
                         "scatter + all-gather of the centroids over RCCL, replica EM) instead of rank-0 fit + broadcast")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the control plane (nccl = RCCL)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host", "peer"],
                    help="collectives of the library's sharded entry points: rccl (xGMI; one GPU per rank) or host (pinned "
                         "staging + the torch.distributed group, e.g. --backend gloo: lets N ranks share ONE GPU -- a check of "
                         "this script's N > 1 path on a single-GPU box, not a measurement)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="logic check on ONE GPU: play R ranks in turn through plda_comm_emulate (no collective, no gather "
                         "timing); the printed rate is not a multi-GPU measurement")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher around it: become the launcher (one process per GPU under
    # torch.distributed.run, rendezvous on 127.0.0.1) -- never a single rank that prints n_gpus: 1 for --gpus 8
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_ranks:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], transport=args.transport))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    emu = args.emulate_ranks if world == 1 else 0
    if emu:
        args.scaling = "strong"        # the emulation is a logic check of the partition on one GPU's memory
    if args.transport in ("host", "peer"):
        local_rank = local_rank % torch.cuda.device_count()     # ranks may share a GPU under the host / peer transports
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from plda_amd import MPlda
    eng = MPlda(local_rank)
    # everything (torch ops, the engine's kernels) on ONE non-default stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    if world > 1:
        from plda_amd.sharding import init_comm
        init_comm(eng, device=dev, transport=args.transport)   # RCCL communicator inside libplda_hip.so (or the host transport)

    cfg = dict(CONFIGS[args.config])
    if args.n:
        cfg.update(N=args.n, M=args.n, Nt=args.n)
    if args.dim:
        cfg["D"] = args.dim
    if args.speakers:
        cfg["K"] = args.speakers
    N, D, K, M, Nt = cfg["N"], cfg["D"], cfg["K"], cfg["M"], cfg["Nt"]
    parts = (args.emulate_ranks if world == 1 else world) or 1
    M1 = M                                            # enrol rows of the single-GPU problem (the gather leg's matrix)
    if args.scaling == "weak" and parts > 1:
        M = M1 * parts                                # weak scaling: every rank scores M1 x Nt trials per step

    # ---- synthetic fit data: uniform [0,1) rows (the reference's usage, README.md:54), N / K utterances per speaker.
    #      C2: np.random.default_rng(2) on the host (as round 1); the larger shapes draw on the device ----
    X = y = None
    if args.config == "C2":
        rng = np.random.default_rng(2)
        X = rng.random((N, D))
        y = (np.arange(N) % K).astype(np.uint64)

    def fit_rows():
        if X is not None:
            return torch.from_numpy(X).to(dev), torch.from_numpy(y.astype(np.int64)).to(dev)
        g = torch.Generator(device=dev); g.manual_seed(2)
        return (torch.rand((N, D), dtype=torch.float64, device=dev, generator=g),
                (torch.arange(N, device=dev, dtype=torch.int64) % K))

    # ---- fit: rank 0 + broadcast of the model, or sharded by speaker over the library's RCCL communicator ----
    fit_info = None
    fit_skew = None
    gpu_one_iter = None
    if args.shard_fit and world > 1:
        dX, dy = fit_rows()
        mine = (dy % world) == rank                      # a partition BY SPEAKER: every centroid is rank-local
        dXl = dX[mine].contiguous()
        dyl = torch.div(dy[mine], world, rounding_mode="floor").contiguous()   # local dense labels 0..K_local-1
        k_local = int(dyl.max().item()) + 1
        del dX, dy
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        eng.fit_sharded_dev(dXl.data_ptr(), dXl.shape[0], D, dyl.data_ptr(), k_local, args.iters)
        torch.cuda.synchronize(dev)
        ft = eng.fit_timings()
        fit_info = {"sharded_by_speaker": True, "em_ms": round(ft["em_ms"], 3), "iters": ft["iters"],
                    "em_iters_per_s": round(ft["iters"] / (ft["em_ms"] / 1e3), 2) if ft["em_ms"] > 0 else None,
                    "fit_wall_s": round(time.perf_counter() - t0, 4), "N": N, "D": D, "K": K}
        del dXl, dyl
        model = eng.get_model()
        packed = np.concatenate([model["mean"], model["transform"].ravel(), model["psi"]])
    elif rank == 0:
        dX, dy = fit_rows()
        torch.cuda.synchronize(dev)
        if not args.no_cpu:
            eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, 1)         # one EM iteration: checked against the oracle's below
            torch.cuda.synchronize(dev)
            gpu_one_iter = eng.fit_internals()
            if X is None:                                                 # C3 / C4: the rows were drawn on the device
                X = dX.cpu().numpy()
                y = dy.cpu().numpy().astype(np.uint64)
        eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, args.iters)   # warm (allocations, code load)
        t0 = time.perf_counter()
        eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, args.iters)
        torch.cuda.synchronize(dev)
        fit_wall = time.perf_counter() - t0
        ft = eng.fit_timings()
        # per-stage spans of one more fit (HIP events on the stream around each stage: include/plda_hip.h plda_trace_*)
        eng.trace_enable(True)
        eng.fit_dev(dX.data_ptr(), N, D, dy.data_ptr(), K, args.iters)
        torch.cuda.synchronize(dev)
        spans = eng.trace_read()
        eng.trace_enable(False)
        stages = []
        for sp in spans:
            st = {"name": sp["name"], "ms": round(sp["ms"], 4)}
            if sp["unit"] == "bytes" and sp["ms"] > 0:
                st["GBps"] = round(sp["work"] / sp["ms"] / 1e6, 1)
                st["frac_hbm_8TBps"] = round(sp["work"] / sp["ms"] / 1e6 / 8000.0, 4)
            if sp["unit"] == "flop" and sp["ms"] > 0:
                st["TFLOPps"] = round(sp["work"] / sp["ms"] / 1e9, 2)
                st["frac_fp64_mfma_78.6"] = round(sp["work"] / sp["ms"] / 1e9 / 78.6, 4)
                if "(K2)" in sp["name"]:
                    # K2 computes the lower triangle only: `work` is the EXECUTED count (N + K) D (D + 1); the same time
                    # priced at the full-square 2 (N + K) D^2 of a plain GEMM is given beside it, labelled, never as the fraction
                    st["flop_counted"] = "executed: lower triangle, (N+K) D (D+1)"
                    st["frac_if_counted_as_full_square_2ND2"] = round(2.0 * (N + K) * D * D / sp["ms"] / 1e9 / 78.6, 4)
            stages.append(st)
        fit_info = {"stats_ms": round(ft["stats_ms"], 3), "em_ms": round(ft["em_ms"], 3),
                    "output_ms": round(ft["output_ms"], 3), "iters": ft["iters"],
                    "em_iters_per_s": round(ft["iters"] / (ft["em_ms"] / 1e3), 2) if ft["em_ms"] > 0 else None,
                    "fit_wall_s": round(fit_wall, 4), "N": N, "D": D, "K": K,
                    # roofline of the statistics pass: K1 reads N D 8 bytes once (HBM bound),
                    # K2 = 2 N D^2 algorithmic flop on the fp64 MFMA pipe (78.6 TFLOP/s); stage times are HIP-event
                    # spans and include the stage's small helper kernels
                    "stages": stages}
        fit_info["em_form"] = eng.fit_plan()["form"]
        model = eng.get_model()
        packed = np.concatenate([model["mean"], model["transform"].ravel(), model["psi"]])
        if args.config == "C2":
            # the same rows labelled as real data is (unequal speaker counts); the model scored below stays the uniform fit's
            try:
                fit_skew = fit_skewed(eng, dX, N, D, K, args.iters, X, not args.no_cpu)
            except Exception as e:   # noqa: BLE001 -- informative leg
                fit_skew = {"error": "%s: %s" % (type(e).__name__, e)}
            if not args.no_extra and world == 1:
                try:
                    fit_skew["fit_d1024"] = fit_large_dim(eng, dev, args.iters)
                except Exception as e:   # noqa: BLE001 -- informative leg
                    fit_skew["fit_d1024"] = {"error": "%s: %s" % (type(e).__name__, e)}
            eng.set_model(model["mean"], model["transform"], model["psi"])
        del dX, dy
    else:
        packed = np.zeros(D + D * D + D)
    if world > 1 and not args.shard_fit:
        t = torch.from_numpy(packed).to(dev)
        dist.broadcast(t, src=0)
        packed = t.cpu().numpy()
        if rank != 0:
            eng.set_model(packed[:D], packed[D:D + D * D].reshape(D, D), packed[D + D * D:])
    psi = packed[D + D * D:]
    if args.targetdim:
        eng.truncate(args.targetdim)
    dout = eng.dims()[0]

    # ---- enrol / test sets in the PLDA space (HBM-resident fp64), REPLICATED on every rank ----
    g = torch.Generator(device=dev); g.manual_seed(1000)
    if cfg["counts"] == "1..5":
        dn = torch.randint(1, 6, (M,), device=dev, dtype=torch.int32, generator=g)
        n_uniform = 0
    else:
        dn = None
        n_uniform = int(cfg["counts"])
    if args.config == "C2" and M == M1:
        dE = torch.from_numpy(np.random.default_rng(1000).random((M, D))).to(dev)
        dVr = torch.from_numpy(np.random.default_rng(7).random((Nt, D))).to(dev)
    else:
        dE = torch.rand((M, D), dtype=torch.float64, device=dev, generator=g)
        dVr = torch.rand((Nt, D), dtype=torch.float64, device=dev, generator=g)
    dU = torch.empty((M, dout), dtype=torch.float64, device=dev)
    dT = torch.empty((Nt, dout), dtype=torch.float64, device=dev)
    eng.transform_rows_dev(dE.data_ptr(), M, D, dn.data_ptr() if dn is not None else None, n_uniform, dU.data_ptr())
    eng.transform_rows_dev(dVr.data_ptr(), Nt, D, None, 1, dT.data_ptr())
    del dE, dVr
    from plda_amd.sharding import block_cyclic_rows
    gather_block = 4096
    if args.block_rows <= 0:
        args.block_rows = -(-(-(-M // parts)) // 256) * 256 if parts > 1 else 4096
    my_blocks = block_cyclic_rows(M, parts, rank if world > 1 else 0, args.block_rows)   # (emulation: rank 0's, for the shape)
    local_rows = sum(b - a for a, b in my_blocks) if world > 1 else M
    # N > 1: a compact slab of this rank's rows; the full matrix exists only in the gather leg (M1 rows)
    out = torch.empty((max(local_rows, 1) if world > 1 else M, Nt), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    dnp = dn.data_ptr() if dn is not None else None
    full = None

    def step(gather=False):
        if emu:
            for r in range(emu):
                eng.comm_emulate(emu, r)
                eng.score_matrix_sharded_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt,
                                             block_rows=args.block_rows, gather=False)
            eng.comm_emulate(1, 0)
        elif world == 1:
            eng.score_matrix_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt)
        elif gather:
            eng.score_matrix_sharded_dev(dU.data_ptr(), dnp, n_uniform, M1, dT.data_ptr(), Nt, full.data_ptr(), Nt,
                                         block_rows=gather_block, gather=True)
        else:
            eng.score_matrix_sharded_local_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt,
                                               block_rows=args.block_rows)

    def timed(gather):
        for _ in range(args.warmup):
            step(gather)
        torch.cuda.synchronize(dev)
        eng.profile_enable(True)
        eng.profile_read(reset=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(gather)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        prof = eng.profile_read(reset=True)
        eng.profile_enable(False)
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, prof

    elapsed, (gemm_ms, launches, gemm_flop) = timed(False)
    gemm_k = eng.score_last_shape()[2]      # algorithmic GEMM depth of the timed launches: D (uniform n), D + G - 1 (mixed n, G distinct counts), 2 D (depth-2D form)

    # ---- parity of the TIMED output: a 64 x 64 sample (rows of THIS rank's blocks, block boundaries included) against
    #      the fp64 oracle's per-trial LLR (oracle/plda_oracle.c) and against the engine's own fp64 trial-list kernel ----
    from plda_amd.sharding import local_row_index
    if world > 1:
        grow = local_row_index(M, world, rank, args.block_rows, device=dev)       # global row of every slab row
    elif emu:
        grow = torch.arange(M, device=dev)
    else:
        grow = torch.arange(M, device=dev)
    nloc = int(grow.numel())
    pick = torch.unique(torch.cat([torch.tensor([0, nloc - 1], device=dev),
                                   torch.linspace(0, nloc - 1, 62, device=dev).long()]))[:64]
    sel_t = torch.unique(torch.cat([torch.tensor([0, 5, Nt - 1], device=dev), torch.linspace(0, Nt - 1, 61, device=dev).long()]))[:64]
    sel_e = grow[pick]
    Uh, Th = dU[sel_e].cpu().numpy(), dT[sel_t].cpu().numpy()
    nh = dn[sel_e].cpu().numpy() if dn is not None else np.full(len(sel_e), n_uniform, np.int32)
    got = (out[pick] if world > 1 else out[sel_e])[:, sel_t].cpu().numpy().astype(np.float64)
    ne, nt_ = got.shape
    ref = eng.score_trials((nh, Uh), (1, Th), np.repeat(np.arange(ne), nt_), np.tile(np.arange(nt_), ne)).reshape(ne, nt_)
    spot = float(np.abs(got - ref).max())
    oracle_check = None
    if True:                 # (always: 4 096 trials of the per-trial oracle take milliseconds; --no-cpu only skips the timed CPU legs)
        try:
            from oracle import binding as ob
            ob.build()
            oref = ob.score_block(psi[:dout], Uh, nh, Th)
            tol = 1e-4 * np.maximum(np.abs(oref), np.abs(oref).mean())       # north_star: 1e-4 relative
            oracle_check = {"sample": "%d x %d trials of the timed output (this rank's rows)" % (ne, nt_),
                            "max_abs_err": float(np.abs(got - oref).max()), "max_err_over_tol": float((np.abs(got - oref) / tol).max()),
                            "within_1e-4": bool((np.abs(got - oref) <= tol).all()),
                            "trial_list_kernel_vs_oracle_max_abs": float(np.abs(ref - oref).max()),
                            "oracle": "oracle/plda_oracle.c per-trial LogLikelihoodRatio, fp64"}
        except Exception as e:   # noqa: BLE001
            oracle_check = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:                                   # every rank's sample must pass, not only rank 0's
        bad = torch.tensor([0.0 if (oracle_check is None or oracle_check.get("within_1e-4", False)) else 1.0, spot],
                           dtype=torch.float64, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        spot = float(bad[1].item())
        if oracle_check is not None and "within_1e-4" in oracle_check:
            oracle_check["within_1e-4_on_every_rank"] = bool(bad[0].item() == 0.0)

    # ---- who took part (as the transport reports it), and the gather-inclusive leg on the single-GPU matrix ----
    multi = None
    gather_info = None
    if world > 1:
        descs = [None] * world
        dist.all_gather_object(descs, eng.comm_describe())
        multi = describe_ranks(descs, out.numel())
    if world > 1 and not args.no_gather:
        try:   # the second leg must not cost the run its (already measured) main line
            err_g = None
            try:
                full = torch.empty((M1, Nt), dtype=torch.float32, device=dev)
                el_g, _ = timed(True)
            except Exception as e:   # noqa: BLE001
                err_g = "%s: %s" % (type(e).__name__, e)
            # a rank that failed must not leave the others waiting in the object gathers below: agree first
            okf = torch.tensor([0.0 if err_g else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if okf.item() == 0.0:
                raise RuntimeError(err_g or "the gather leg failed on another rank")
            # after a gathered step every rank holds every row: bit patterns of the first block of every OTHER rank, as
            # this rank received them, against the owner's own copy
            def cks(a, b):
                return int(full[a:b].view(torch.int32).to(torch.int64).sum().item())
            firsts = [block_cyclic_rows(M1, world, r, gather_block)[0] for r in range(world)]
            seen = [None] * world
            dist.all_gather_object(seen, [cks(a, b) for a, b in firsts])
            agree = all(seen[r][q] == seen[q][q] for r in range(world) for q in range(world))
            gather_info = {"value": float(M1) * Nt * args.steps / el_g, "unit": "trials/s", "ms_per_step": el_g / args.steps * 1e3,
                           "matrix": "%d x %d (the single-GPU problem, assembled on every rank)" % (M1, Nt),
                           "bytes_received_per_rank_per_step": int(M1 * Nt * 4 * (world - 1) / world),
                           "ingest_GBps_per_rank": round(M1 * Nt * 4 * (world - 1) / world / (el_g / args.steps) / 1e9, 1),
                           "gathered_blocks_bit_identical_to_their_owners": bool(agree),
                           "how": "plda_score_matrix_sharded_dev(gather=1): in-place all-gather of every %d x %d-row super-block on a "
                                  "side stream, overlapped with the scoring of the next one" % (world, gather_block)}
            if multi is not None:
                multi["cross_rank_checksum_ok"] = bool(agree)
            # ---- the same leg over the DIRECT-WRITE provider (plda_comm_init_peer: every rank pushes its blocks into the
            #      others' buffers through HIP IPC mappings, one copy stream per peer -- all xGMI links at once; a gloo
            #      group carries the IPC handles only).  The RCCL communicator is given back first.
            if args.transport == "rccl":
                try:
                    err_p = None
                    try:
                        eng.comm_destroy()
                        boot = dist.new_group(backend="gloo")
                        init_comm(eng, group=boot, device=dev, transport="peer")
                        el_p, _ = timed(True)
                        seen_p = [None] * world
                        dist.all_gather_object(seen_p, [cks(a, b) for a, b in firsts])
                        agree_p = all(seen_p[r][q] == seen[q][q] for r in range(world) for q in range(world))
                    except Exception as e:   # noqa: BLE001
                        err_p = "%s: %s" % (type(e).__name__, e)
                    okp = torch.tensor([0.0 if err_p else 1.0], dtype=torch.float64, device=dev)
                    dist.all_reduce(okp, op=dist.ReduceOp.MIN)
                    if okp.item() == 0.0:
                        raise RuntimeError(err_p or "the direct-write leg failed on another rank")
                    gather_info["peer_direct_write"] = {
                        "value": float(M1) * Nt * args.steps / el_p, "unit": "trials/s", "ms_per_step": el_p / args.steps * 1e3,
                        "ingest_GBps_per_rank": round(M1 * Nt * 4 * (world - 1) / world / (el_p / args.steps) / 1e9, 1),
                        "gathered_blocks_bit_identical_to_rccl": bool(agree_p), "transport": eng.comm_describe()["transport"],
                        "how": "plda_comm_init_peer: HIP-IPC mappings, one device-to-device push per peer and block on its own stream, "
                               "sequence-number flags in an uncached page instead of a rendezvous; nothing but 72-byte handles crosses the host"}
                except Exception as e:   # noqa: BLE001
                    gather_info["peer_direct_write"] = {"error": "%s: %s" % (type(e).__name__, e)}
            full = None
        except Exception as e:   # noqa: BLE001
            gather_info = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- norm(): z-norm statistics of the first 50k enrol models over a 200k-row cohort (the C5 shape at this D),
    #      outside the timed region: MPlda_norm, pldamodule.cpp:196-256 ----
    zn = None
    if rank == 0 and world == 1 and not emu and not args.no_extra and not args.targetdim:
        zM, zNb = min(M, 50000), 200000
        gz = torch.Generator(device=dev); gz.manual_seed(5)
        cohort = torch.rand((zNb, D), dtype=torch.float64, device=dev, generator=gz)
        zmean = torch.empty(zM, dtype=torch.float64, device=dev); zstd = torch.empty(zM, dtype=torch.float64, device=dev)
        eng.znorm_stats_dev(cohort.data_ptr(), zNb, zNb, D, dU.data_ptr(), zM, zmean.data_ptr(), zstd.data_ptr())
        torch.cuda.synchronize(dev)
        tz = time.perf_counter()
        eng.znorm_stats_dev(cohort.data_ptr(), zNb, zNb, D, dU.data_ptr(), zM, zmean.data_ptr(), zstd.data_ptr())
        torch.cuda.synchronize(dev)
        zms = (time.perf_counter() - tz) * 1e3
        # roofline (round-3 review, weak 8): the stages' HIP-event spans of one more call.  Work as EXECUTED: the cohort's
        # transform 2 Nb D Dout, the (D+1)-wide covariance as a lower triangle Nb (D+1)(D+2), the model GEMM 2 M D^2 -- all
        # on the fp64 matrix cores (78.6 TFLOP/s); HBM floor: the cohort read once + the centred rows written and read
        eng.trace_enable(True); eng.trace_read(reset=True)
        eng.znorm_stats_dev(cohort.data_ptr(), zNb, zNb, D, dU.data_ptr(), zM, zmean.data_ptr(), zstd.data_ptr())
        torch.cuda.synchronize(dev)
        zsp = eng.trace_read(reset=True); eng.trace_enable(False)
        d1 = dout + 1
        zflop = 2.0 * zNb * D * dout + float(zNb) * d1 * (d1 + 1) + 2.0 * zM * dout * dout
        zbytes = 8.0 * (zNb * D + 3.0 * zNb * d1 + 2.0 * zM * dout)
        zn = {"models": zM, "cohort": zNb, "ms": round(zms, 3),
              "pairs_the_reference_scores": zM * zNb, "finite": bool(torch.isfinite(zmean).all() and torch.isfinite(zstd).all()),
              "how": "cohort moments in fp64: (D+1)-wide SYRK over the cohort + one M x D x D GEMM (DESIGN.md, row a13)",
              "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_FP64_MFMA_TFLOPS, "flop_executed": zflop,
                           "achieved": round(zflop / zms / 1e9, 2), "frac": round(zflop / zms / 1e9 / PEAK_FP64_MFMA_TFLOPS, 4),
                           "hbm_bytes_algorithmic": zbytes, "hbm_floor_ms": round(zbytes / 8e12 * 1e3, 4),
                           "stages": [{"name": sp["name"], "ms": round(sp["ms"], 4)} for sp in zsp]}}
        del cohort, zmean, zstd

    # ---- transform (K4: rows -> PLDA space -> length norm, pldamodule.cpp:111-194) of the test side's rows, outside
    #      the timed region: 2 N D^2 flop on the fp64 MFMA pipe + a length-norm pass over the output ----
    tf = None
    if rank == 0 and world == 1 and not emu and not args.no_extra and not args.targetdim:
        gx = torch.Generator(device=dev); gx.manual_seed(6)
        Xt = torch.rand((Nt, D), dtype=torch.float64, device=dev, generator=gx)
        Yt = torch.empty((Nt, dout), dtype=torch.float64, device=dev)
        eng.transform_rows_dev(Xt.data_ptr(), Nt, D, None, 1, Yt.data_ptr())
        torch.cuda.synchronize(dev)

        def timed(reps):
            eng.trace_enable(True)
            eng.trace_read(reset=True)
            tt0 = time.perf_counter()
            for _ in range(reps):
                eng.transform_rows_dev(Xt.data_ptr(), Nt, D, None, 1, Yt.data_ptr())
            torch.cuda.synchronize(dev)
            w = (time.perf_counter() - tt0) / reps
            spans = [x for x in eng.trace_read(reset=True) if x["name"].startswith("transform.")]
            eng.trace_enable(False)
            return (spans[0]["ms"] / spans[0]["calls"] / 1e3 if spans else w), w      # HIP events around the kernel(s)

        # a call is a fifth of a millisecond: the first ones after a pause run 10-15 % slower than the steady state (the
        # same kernel, the same data: measured in sequences of ten-call batches), so -- like the W warm-up steps of the
        # headline metric -- the figure is taken behind 40 untimed calls; the first five calls are reported beside it
        first5, _ = timed(5)
        for _ in range(40):
            eng.transform_rows_dev(Xt.data_ptr(), Nt, D, None, 1, Yt.data_ptr())
        tsec, wall = timed(20)
        tf = {"rows": Nt, "D": D, "ms": round(tsec * 1e3, 3), "wall_ms": round(wall * 1e3, 3), "first_5_calls_ms": round(first5 * 1e3, 3),
              "rows_per_s": round(Nt / tsec, 1),
              "TFLOPps": round(2.0 * Nt * D * dout / tsec / 1e12, 2),
              "frac_fp64_mfma_78.6": round(2.0 * Nt * D * dout / tsec / (PEAK_FP64_MFMA_TFLOPS * 1e12), 4),
              "GBps_in_plus_out": round(8.0 * Nt * (D + dout) / tsec / 1e9, 1)}
        del Xt, Yt

    # ---- where the wall-clock fraction comes from (round 4): the product kernel with two stamp pairs per workgroup
    #      (PLDA_GEMM_VARIANT=47: s_memtime against the 100 MHz s_memrealtime) on the same operands -- the shader clock
    #      this box holds under the load, and workgroup 0's cycles per tile against the tile's MFMA cycles.  The
    #      fraction of the peak is (MFMA-busy in cycles) x (clock / 2.4 GHz) x (1 - tail); the clock differs between the
    #      boxes of a pool by 1-4 %, the cycles do not.  Outside the timed region, on a second handle. ----
    clock_info = None
    if rank == 0 and world == 1 and not emu and not args.no_extra and not args.targetdim and "bt4" in eng.score_last_kernel():
        try:
            import ctypes as C
            old_env = os.environ.get("PLDA_GEMM_VARIANT")
            os.environ["PLDA_GEMM_VARIANT"] = "47"
            e2 = MPlda(local_rank, diag=True)      # the diagnostic build: the product library does not contain this arm
            if old_env is None:
                del os.environ["PLDA_GEMM_VARIANT"]
            else:
                os.environ["PLDA_GEMM_VARIANT"] = old_env
            mdl = eng.get_model()
            e2.set_model(mdl["mean"], mdl["transform"], mdl["psi"])
            e2.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            dnp2 = dn.data_ptr() if dn is not None else None
            for _ in range(3):
                e2.score_matrix_dev(dU.data_ptr(), dnp2, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt)
            torch.cuda.synchronize(dev)
            raw = np.zeros(8 * 16 * 8 * 8, np.uint64)
            e2._ck(e2._lib.plda_profile_timeline(e2._h, C.c_void_p(raw.ctypes.data), raw.size))
            cyc, real, tiles = int(raw[0]), int(raw[1]), int(raw[2])
            kalg = eng.score_last_shape()[2]                 # algorithmic depth of the timed launches
            dp8 = (dout + 7) // 8 * 8
            kg = max(dp8, 16) if dn is None else (2 * dp8 if kalg == 2 * dout else dp8 + (kalg - dout + 7) // 8 * 8)
            ideal_tile = kg // 8 * 4096 + 1024          # MFMA cycles of a 256 x 256 tile per SIMD (k steps + the bias MFMAs)
            mhz = cyc / (real / 100.0)
            wg = raw[8:8 + 4 * 256].reshape(256, 4).astype(np.int64)       # per workgroup: cycles, start, end (100 MHz ticks), tiles
            end_spread_us = float((wg[:, 2].max() - wg[:, 2].min()) / 100.0)
            cpt = wg[:, 0] / np.maximum(wg[:, 3], 1)
            wmhz = wg[:, 0] / np.maximum((wg[:, 2] - wg[:, 1]) / 100.0, 1e-9)
            clock_info = {"shader_clock_MHz": round(float(np.median(wmhz)), 1), "clock_over_2400": round(float(np.median(wmhz)) / 2400.0, 4),
                          "clock_MHz_min_max_over_workgroups": [round(float(wmhz.min()), 1), round(float(wmhz.max()), 1)],
                          "tiles_per_workgroup_min_max": [int(wg[:, 3].min()), int(wg[:, 3].max())],
                          "cycles_per_tile": round(float(np.median(cpt)), 1), "mfma_cycles_per_tile": ideal_tile,
                          "mfma_busy_in_cycles": round(float(ideal_tile / np.median(cpt)), 4),
                          "workgroup0": {"cycles": cyc, "ticks_100MHz": real, "tiles": tiles, "MHz": round(mhz, 1)},
                          "end_spread_of_workgroups_us": round(end_spread_us, 1),
                          "how": "libplda_hip_diag.so, PLDA_GEMM_VARIANT=47 (the product kernel + s_memtime / s_memrealtime stamps per workgroup), last of 3 launches on the timed operands"}
            e2.set_stream(None)
            del e2
        except ImportError as ex:    # the diagnostic library was not built (python -m plda_amd.build --diag)
            if old_env is None:
                os.environ.pop("PLDA_GEMM_VARIANT", None)
            else:
                os.environ["PLDA_GEMM_VARIANT"] = old_env
            clock_info = {"skipped": str(ex)[:200]}
        except Exception as ex:      # a diagnostic: never fails the bench line
            clock_info = {"error": str(ex)[:200]}

    # ---- the consumer of the trials matrix (scoring/scorePLDA.py:302-318 -> scoring/eer.py:68-76), outside the timed region:
    #      the exact EER of the timed output (one full pass since round 5), and the same EER straight from the operands
    #      (plda_score_eer_dev: the scores exist one <= 4 GiB row slab at a time).  Synthetic labels: speaker = index mod 5000.
    eer_info = None
    if rank == 0 and world == 1 and not emu and not args.no_extra and not args.targetdim:
        try:
            from plda_amd import eer as geer
            nspk = max(2, min(5000, M // 2))
            de = (torch.arange(M, device=dev, dtype=torch.int64) % nspk).contiguous()
            dt_ = (torch.arange(Nt, device=dev, dtype=torch.int64) % nspk).contiguous()
            torch.cuda.synchronize(dev)

            def best(fn, reps=3):
                b_, r_ = 1e30, None
                for _ in range(reps):
                    t0_ = time.perf_counter(); r_ = fn(); b_ = min(b_, time.perf_counter() - t0_)
                return b_, r_
            geer.eer_from_matrix_dev(eng, out.data_ptr(), Nt, M, Nt, de.data_ptr(), dt_.data_ptr())
            t_m, r_m = best(lambda: geer.eer_from_matrix_dev(eng, out.data_ptr(), Nt, M, Nt, de.data_ptr(), dt_.data_ptr()))
            geer.eer_from_operands_dev(eng, dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, de.data_ptr(), dt_.data_ptr())
            t_o, r_o = best(lambda: geer.eer_from_operands_dev(eng, dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, de.data_ptr(), dt_.data_ptr()), 2)
            eer_info = {"of_the_timed_matrix_ms": round(t_m * 1e3, 3), "matrix_read_GBps": round(M * Nt * 4 / t_m / 1e9, 1),
                        "from_operands_ms": round(t_o * 1e3, 3), "from_operands_trials_per_s": M * Nt / t_o,
                        "step_plus_matrix_eer_ms": round(elapsed / args.steps * 1e3 + t_m * 1e3, 3),
                        "identical": bool(np.array_equal(r_m, r_o)), "eer": float(r_m[3]), "targets": int(r_m[4]),
                        "how": "plda_eer_matrix_dev on the timed output (pilot on every 32nd row, ONE full pass, exact refinement on the window's "
                               "scores); plda_score_eer_dev: the same from the operands, scores held one row slab at a time -- never %.0f GB" % (M * Nt * 4 / 1e9)}
            # (the slabs re-packed the test side and the timed buffer is untouched)
        except Exception as ex:      # noqa: BLE001 -- an extra leg
            eer_info = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}

    # ---- OPT-IN arm (never `value`, never `roofline`): the same trials with the contraction as three bf16 terms per operand
    #      (PLDA_SCORE_DTYPE=bf16x3, csrc/score_bf16x3.inc) on a second handle, same operands, same output buffer.  north_star
    #      prescribes fp32 MFMA for the trials GEMM, and that kernel sits at its ceiling; this arm trades the matrix pipe's
    #      fp32 rate for the bf16 one at fp32-grade accuracy (checked against the oracle below, same tolerance).  Its
    #      roofline is the HBM write of the scores. ----
    b3 = None
    if rank == 0 and world == 1 and not emu and not args.no_extra and not args.targetdim and os.environ.get("PLDA_SCORE_DTYPE", "") in ("", "f32"):
        try:
            os.environ["PLDA_SCORE_DTYPE"] = "bf16x3"
            e3 = MPlda(local_rank)
            del os.environ["PLDA_SCORE_DTYPE"]
            mdl = eng.get_model()
            e3.set_model(mdl["mean"], mdl["transform"], mdl["psi"])
            e3.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            for _ in range(2):
                e3.score_matrix_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            for _ in range(3):
                e3.score_matrix_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt)
            torch.cuda.synchronize(dev)
            dt3 = (time.perf_counter() - t3) / 3
            got3 = out[sel_e][:, sel_t].cpu().numpy().astype(np.float64)
            from oracle import binding as ob
            oref3 = ob.score_block(psi[:dout], Uh, nh, Th)
            tol3 = 1e-4 * np.maximum(np.abs(oref3), np.abs(oref3).mean())
            b3 = {"dtype": "bf16x3 (three bf16 terms per fp32 operand value, six v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulation)",
                  "ms_per_step": round(dt3 * 1e3, 3), "trials_per_s": M * Nt / dt3, "speedup_over_f32_step": round((elapsed / args.steps) / dt3, 3),
                  "kernel": e3.score_last_kernel(),
                  "oracle_check": {"max_abs_err": float(np.abs(got3 - oref3).max()), "max_err_over_tol": float((np.abs(got3 - oref3) / tol3).max()),
                                   "within_1e-4": bool((np.abs(got3 - oref3) <= tol3).all())},
                  "roofline": {"bound": "hbm", "unit": "GB/s", "achieved": round(M * Nt * 4 / dt3 / 1e9, 1), "peak": 8000.0,
                               "frac": round(M * Nt * 4 / dt3 / 8e12, 4), "algorithmic_bytes": M * Nt * 4,
                               "note": "whole step (prep + split + GEMM) against the fp32 scores' write; the part holds ~1.8 GHz under this kernel "
                                       "(scripts/probe/bf16x3_clock.py), where its MFMAs alone are two thirds of the step"},
                  "how": "opt-in: PLDA_SCORE_DTYPE=bf16x3 at plda_create; the default and every other number of this line is the fp32 MFMA path"}
            # the timed output buffer holds this arm's scores now: restore the fp32 ones for the legs below
            eng.score_matrix_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt)
            torch.cuda.synchronize(dev)
            e3.set_stream(None)
            del e3
        except Exception as ex:      # noqa: BLE001 -- an extra leg: never fails the bench line
            os.environ.pop("PLDA_SCORE_DTYPE", None)
            b3 = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}

    # ---- build extension named by BASELINE configs[1]: targetdim = 150 (top-psi dims), same trials ----
    td = None
    if rank == 0 and world == 1 and args.config == "C2" and not args.targetdim and dout > 150 and not args.no_extra:
        eng.truncate(150)
        dU150 = torch.empty((M, 150), dtype=torch.float64, device=dev)
        dT150 = torch.empty((Nt, 150), dtype=torch.float64, device=dev)
        dE2 = torch.from_numpy(np.random.default_rng(1000).random((M, D))).to(dev)
        eng.transform_rows_dev(dE2.data_ptr(), M, D, None, 1, dU150.data_ptr())
        dV2 = torch.from_numpy(np.random.default_rng(7).random((Nt, D))).to(dev)
        eng.transform_rows_dev(dV2.data_ptr(), Nt, D, None, 1, dT150.data_ptr())
        del dE2, dV2
        eng.score_matrix_dev(dU150.data_ptr(), None, 1, M, dT150.data_ptr(), Nt, out.data_ptr(), Nt)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(2):
            eng.score_matrix_dev(dU150.data_ptr(), None, 1, M, dT150.data_ptr(), Nt, out.data_ptr(), Nt)
        torch.cuda.synchronize(dev)
        dt150 = (time.perf_counter() - t1) / 2
        td = {"D_eff": 150, "ms_per_step": round(dt150 * 1e3, 3), "trials_per_s": M * Nt / dt150,
              "tflops": round(2 * 150 * M * Nt / dt150 / 1e12, 2),
              "note": "no reference parity exists for targetdim (SURVEY.md App. B Q3); GEMM depth padded to 152"}
        del dU150, dT150

    traffic, traffic_src = latest_traffic(M, Nt, dout) if world == 1 else (None, None)

    if rank == 0:
        trials = float(M) * Nt * args.steps                      # whole job (weak scaling: M = N x the single-GPU rows)
        dout8 = (dout + 7) // 8 * 8
        value = trials / elapsed
        avg_gemm_s = gemm_ms / 1e3 / max(launches, 1)
        achieved = (gemm_flop / max(launches, 1)) / avg_gemm_s / 1e12 if avg_gemm_s > 0 else 0.0
        res = {
            "metric": "PLDA LLR trials/sec", "value": value, "unit": "trials/s",
            "n_gpus": world, **({"emulated_ranks_on_one_gpu": emu} if emu else {}), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if (world > 1 or emu) else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s; fit %d EM iters; D_eff=%d" % (cfg["what"], args.iters, dout),
                       "trials_per_step": M * Nt, "enrol_models": M, "test_vectors": Nt,
                       "parallelism": "enrol rows block-cyclic over %d rank(s) (%s scaling: %d enrol models in all), scores left sharded in compact slabs"
                                      % (world, args.scaling, M) if world > 1 else "one GPU",
                       **({"value_excludes": "scores left row-sharded in compact [M/N, Nt] slabs: NO collective in the timed region; the "
                                             "matrix assembled on every rank by the all-gather over xGMI is gather_trials_per_s, beside value"}
                          if world > 1 else {}),
                       "score_dtype": "f32 (fp64 bias terms, fp32 MFMA contraction)", "fit_dtype": "f64"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": int(gemm_flop / max(launches, 1) / (2 * gemm_k) * 4),
                         "kernel": "%s (rank 0's launches)" % eng.score_last_kernel(),
                         "flop_per_trial": 2 * gemm_k,
                         **({"flop_per_trial_note": "mixed enrol counts bucketed by their %d distinct values: depth D + G - 1 = %d (executed, padded to 8-k steps: %d); "
                                                    "the depth-2D form of rounds 1-4 executed %d flop per trial"
                                                    % (gemm_k - dout + 1, gemm_k, dout8 + (gemm_k - dout + 7) // 8 * 8, 4 * dout8)}
                            if dn is not None and gemm_k != 2 * dout else {}),
                         "avg_kernel_ms": round(avg_gemm_s * 1e3, 4),
                         "launches": launches,
                         "hbm_write_GBps": round(gemm_flop / max(launches, 1) / (2 * gemm_k) * 4 / avg_gemm_s / 1e9, 1) if avg_gemm_s > 0 else None,
                         **({"clock": clock_info} if clock_info else {})},
            "fit": fit_info, "spot_check_max_abs_err": spot, "oracle_check": oracle_check,
        }
        if fit_info and fit_info.get("em_iters_per_s"):
            res["fit_em_iters_per_s"] = fit_info["em_iters_per_s"]           # equal speaker counts (G = 1)
        if fit_skew and "fit_d1024" in fit_skew:
            res["fit_d1024"] = fit_skew.pop("fit_d1024")
        if fit_skew:
            res["fit_skewed"] = fit_skew
            if fit_skew.get("em_iters_per_s"):
                res["fit_skewed_em_iters_per_s"] = fit_skew["em_iters_per_s"]   # n_k in [5, 60]: what real data looks like
        if multi:
            res["multi_gpu"] = multi
        if eer_info:
            res["eer"] = eer_info
        if b3:
            res["bf16x3_arm"] = b3
        if td:
            res["targetdim150"] = td
        if zn:
            res["znorm_stats"] = zn
        if tf:
            res["transform"] = tf
        if gather_info:
            res["gather_inclusive"] = gather_info
            # what crossed xGMI, where a reader of the parsed line sees it (round-4 review): the assembled-matrix rate over RCCL
            # and over the direct-write provider, and the bytes every rank took in per second
            if "value" in gather_info:
                res["gather_transport"] = (multi or {}).get("transport")
                res["gather_trials_per_s"] = gather_info["value"]
                res["gather_ingest_GBps_per_rank"] = gather_info.get("ingest_GBps_per_rank")
            pdw = gather_info.get("peer_direct_write") or {}
            if "value" in pdw:
                res["gather_peer_trials_per_s"] = pdw["value"]
                res["gather_peer_ingest_GBps_per_rank"] = pdw.get("ingest_GBps_per_rank")
        if multi:
            res["transport"] = multi["transport"]
            res["comm_nranks"] = multi["comm_nranks"]
            if "rccl_nranks" in multi:                       # only ever from ncclCommCount (describe_ranks)
                res["rccl_nranks"] = multi["rccl_nranks"]
                res["rccl_version"] = multi["rccl_version"]
            res["distinct_devices"] = multi.get("distinct_devices")
        if not args.no_cpu and world == 1:
            cb = cpu_baseline(dout, psi[:dout])
            try:
                cb["best_effort"] = cpu_best_effort(dout, psi[:dout])
            except Exception as e:
                cb["best_effort"] = {"error": str(e)}
            if X is not None:
                try:
                    cb["fit_em"] = cpu_em_baseline(X, y, gpu_one_iter)
                except Exception as e:  # the EM leg is informative only
                    cb["fit_em"] = {"error": str(e)}
            res["cpu_baseline"] = cb
        if world == 1 and not emu and not args.no_extra and not args.targetdim:
            try:
                if td:
                    eng.set_model(packed[:D], packed[D:D + D * D].reshape(D, D), psi)    # (the targetdim leg truncated it)
                del out
                res["end_to_end"] = end_to_end(eng, X if args.config == "C2" else None, y, D, dout)
            except Exception as e:   # noqa: BLE001 -- informative leg
                res["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(res), flush=True)
    if world > 1:
        eng.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
