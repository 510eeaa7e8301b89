#!/usr/bin/env python
"""bench.py -- PLDA LLR trials/sec (and fit-EM iters/sec) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.

Default workload (BASELINE.json configs[1], "C2", the configuration the metric is quoted on):
100 000 random i-vectors, featdim 200, 5 000 speakers; fit = statistics + 10 EM iterations +
GetOutput on the GPU (fp64); a "step" = one pass of the hot path over one batch = the
100k x 100k trials matrix (1e10 log-likelihood ratios, n = 1 per enrol model): fp64 bias terms,
fp64 -> fp32 operand packing, fp32-MFMA GEMM, 40 GB of fp32 scores written to HBM.  Inputs are
HBM-resident before the timed region; scores stay in HBM.

N > 1 (one process per GPU, torchrun): STRONG scaling of the same trials matrix.  Every rank holds
the replicated model, enrol and test sets and calls the library's own sharded entry point
`plda_score_matrix_sharded_dev` (RCCL inside libplda_hip.so; torch.distributed only carries the
128-byte unique id, the model broadcast and the timing reduction): enrol rows are dealt out
block-cyclically and every rank writes its blocks in place into the full matrix.
  value            = trials/s with the scores left row-sharded (no data-path collective: what
                     thresholding, counting, EER and z-norm consume);
  gather_inclusive = the same K steps with every block all-gathered in place over xGMI, on a side
                     stream, overlapped with the scoring of the following blocks (north_star's
                     "RCCL all-gather to assemble scores"): its volume -- (N-1)/N of 40 GB into every
                     rank -- not the GEMM bounds it, which is why it is reported beside `value`.
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


def cpu_em_baseline(X, y, seconds_cap=60.0):
    """One Kaldi-style EM iteration (per-class loop, explicit inversions) of the oracle on
    the SAME C2 statistics, single thread."""
    from oracle import binding as ob
    st = ob.stats(X, y)
    D = X.shape[1]
    t0 = time.perf_counter()
    ob.em_iter(st, np.eye(D), np.eye(D))
    dt = time.perf_counter() - t0
    return {"em_iters_per_s": 1.0 / dt, "sample": "1 EM iteration at N=%d D=%d K=%d, %.1f s" % (X.shape[0], D, int(y.max()) + 1, dt)}


def latest_traffic(M, Nt, dout):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary of THIS
    workload (profiles/rNN_traffic_trials_gemm.json, written by scripts/gpu_profile.sh); bench.py cannot
    collect counters itself."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_trials_gemm.json"))):
        try:
            j = json.load(open(f))
            if tuple(j.get("shape", (100000, 100000, 200))) == (M, Nt, dout):
                best = (float(j["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch; kernel %s)" % j.get("kernel", "?"))
        except Exception:
            pass
    return best if best else (None, None)


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
    ap.add_argument("--block-rows", type=int, default=4096, help="N>1: rows per block of the block-cyclic row partition")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the targetdim-150 extra measurement (profiling runs: every launch of the trials kernel "
                         "is then the timed workload)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the gather-inclusive second timed region")
    ap.add_argument("--shard-fit", action="store_true",
                    help="N>1: shard the fit statistics by speaker through plda_fit_sharded_dev (all-reduce of the "
                         "scatter + all-gather of the centroids over RCCL, replica EM) instead of rank-0 fit + broadcast")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the control plane (nccl = RCCL)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="logic check on ONE GPU: play R ranks in turn through plda_comm_emulate (no collective, no gather "
                         "timing); the printed rate is not a multi-GPU measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    emu = args.emulate_ranks if world == 1 else 0
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
        init_comm(eng, device=dev)          # RCCL communicator inside libplda_hip.so

    cfg = dict(CONFIGS[args.config])
    if args.n:
        cfg.update(N=args.n, M=args.n, Nt=args.n)
    if args.dim:
        cfg["D"] = args.dim
    if args.speakers:
        cfg["K"] = args.speakers
    N, D, K, M, Nt = cfg["N"], cfg["D"], cfg["K"], cfg["M"], cfg["Nt"]

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
            stages.append(st)
        fit_info = {"stats_ms": round(ft["stats_ms"], 3), "em_ms": round(ft["em_ms"], 3),
                    "output_ms": round(ft["output_ms"], 3), "iters": ft["iters"],
                    "em_iters_per_s": round(ft["iters"] / (ft["em_ms"] / 1e3), 2) if ft["em_ms"] > 0 else None,
                    "fit_wall_s": round(fit_wall, 4), "N": N, "D": D, "K": K,
                    # roofline of the statistics pass: K1 reads N D 8 bytes once (HBM bound),
                    # K2 = 2 N D^2 algorithmic flop on the fp64 MFMA pipe (78.6 TFLOP/s); stage times are HIP-event
                    # spans and include the stage's small helper kernels
                    "stages": stages}
        del dX, dy
        model = eng.get_model()
        packed = np.concatenate([model["mean"], model["transform"].ravel(), model["psi"]])
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
    if args.config == "C2":
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
    out = torch.empty((M, Nt), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    dnp = dn.data_ptr() if dn is not None else None

    def step(gather=False):
        if emu:
            for r in range(emu):
                eng.comm_emulate(emu, r)
                eng.score_matrix_sharded_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt,
                                             block_rows=args.block_rows, gather=False)
            eng.comm_emulate(1, 0)
        elif world == 1:
            eng.score_matrix_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt)
        else:
            eng.score_matrix_sharded_dev(dU.data_ptr(), dnp, n_uniform, M, dT.data_ptr(), Nt, out.data_ptr(), Nt,
                                         block_rows=args.block_rows, gather=gather)

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

    # ---- spot parity check of the timed output against the fp64 trial-list kernel (rows of THIS rank's blocks) ----
    from plda_amd.sharding import block_cyclic_rows
    mine = block_cyclic_rows(M, emu or world, (emu - 1) if emu else rank, args.block_rows) if (world > 1 or emu) else [(0, M)]
    rows = sorted({mine[0][0], mine[0][1] - 1, mine[-1][0], mine[-1][1] - 1})
    sel_e = torch.tensor(rows, device=dev)
    sel_t = torch.tensor([0, 5, Nt // 3, Nt - 1], device=dev)
    Uh, Th = dU[sel_e].cpu().numpy(), dT[sel_t].cpu().numpy()
    nh = dn[sel_e].cpu().numpy() if dn is not None else np.full(len(rows), n_uniform, np.int32)
    got = out[sel_e][:, sel_t].cpu().numpy()
    ref = eng.score_trials((nh, Uh), (1, Th), np.repeat(np.arange(len(rows)), 4), np.tile(np.arange(4), len(rows))).reshape(len(rows), 4)
    spot = float(np.abs(got - ref).max())

    gather_info = None
    if world > 1 and not args.no_gather:
        try:   # the second leg must not cost the run its (already measured) main line
            el_g, _ = timed(True)
            # after a gathered step every rank holds every row: check one row of another rank's block
            other = block_cyclic_rows(M, world, (rank + 1) % world, args.block_rows)[0][0]
            g2 = out[other, sel_t].cpu().numpy()
            r2 = eng.score_trials((nh[:1] if dn is None else dn[other:other + 1].cpu().numpy(), dU[other:other + 1].cpu().numpy()),
                                  (1, Th), np.zeros(4, np.int64), np.arange(4))
            gather_info = {"value": float(M) * Nt * args.steps / el_g, "unit": "trials/s", "ms_per_step": el_g / args.steps * 1e3,
                           "bytes_received_per_rank_per_step": int(M * Nt * 4 * (world - 1) / world),
                           "ingest_GBps_per_rank": round(M * Nt * 4 * (world - 1) / world / (el_g / args.steps) / 1e9, 1),
                           "peer_row_max_abs_err": float(np.abs(g2 - r2).max()),
                           "how": "plda_score_matrix_sharded_dev(gather=1): in-place ncclAllGather of every %d x %d-row super-block on a "
                                  "side stream, overlapped with the scoring of the next one" % (world, args.block_rows)}
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
        zn = {"models": zM, "cohort": zNb, "ms": round((time.perf_counter() - tz) * 1e3, 3),
              "pairs_the_reference_scores": zM * zNb, "finite": bool(torch.isfinite(zmean).all() and torch.isfinite(zstd).all()),
              "how": "cohort moments in fp64: (D+1)-wide SYRK over the cohort + one M x D x D GEMM (DESIGN.md, row a13)"}
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
        eng.trace_enable(True)
        eng.trace_read(reset=True)
        tt0 = time.perf_counter()
        for _ in range(5):
            eng.transform_rows_dev(Xt.data_ptr(), Nt, D, None, 1, Yt.data_ptr())
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - tt0) / 5
        sp = [x for x in eng.trace_read(reset=True) if x["name"].startswith("transform.")]
        eng.trace_enable(False)
        tsec = sp[0]["ms"] / sp[0]["calls"] / 1e3 if sp else wall      # HIP events around the kernel(s)
        tf = {"rows": Nt, "D": D, "ms": round(tsec * 1e3, 3), "wall_ms": round(wall * 1e3, 3), "rows_per_s": round(Nt / tsec, 1),
              "TFLOPps": round(2.0 * Nt * D * dout / tsec / 1e12, 2),
              "frac_fp64_mfma_78.6": round(2.0 * Nt * D * dout / tsec / (PEAK_FP64_MFMA_TFLOPS * 1e12), 4),
              "GBps_in_plus_out": round(8.0 * Nt * (D + dout) / tsec / 1e9, 1)}
        del Xt, Yt

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
        gemm_k = (2 if dn is not None else 1) * dout            # algorithmic GEMM depth: D (uniform n) or 2 D (mixed n)
        trials = float(M) * Nt * args.steps                      # whole job: the matrix is the same at every N
        value = trials / elapsed
        avg_gemm_s = gemm_ms / 1e3 / max(launches, 1)
        achieved = (gemm_flop / max(launches, 1)) / avg_gemm_s / 1e12 if avg_gemm_s > 0 else 0.0
        res = {
            "metric": "PLDA LLR trials/sec", "value": value, "unit": "trials/s",
            "n_gpus": world, **({"emulated_ranks_on_one_gpu": emu} if emu else {}), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s; fit %d EM iters; D_eff=%d" % (cfg["what"], args.iters, dout),
                       "trials_per_step": M * Nt, "parallelism": "enrol rows block-cyclic over %d rank(s), scores left sharded" % world,
                       "score_dtype": "f32 (fp64 bias terms, fp32 MFMA contraction)", "fit_dtype": "f64"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": M * Nt * 4,
                         "kernel": "trials_gemm_bt2_kernel (rank 0's launches)",
                         "flop_per_trial": 2 * gemm_k, "avg_kernel_ms": round(avg_gemm_s * 1e3, 4),
                         "launches": launches,
                         "hbm_write_GBps": round(gemm_flop / max(launches, 1) / (2 * gemm_k) * 4 / avg_gemm_s / 1e9, 1) if avg_gemm_s > 0 else None},
            "fit": fit_info, "spot_check_max_abs_err": spot,
        }
        if td:
            res["targetdim150"] = td
        if zn:
            res["znorm_stats"] = zn
        if tf:
            res["transform"] = tf
        if gather_info:
            res["gather_inclusive"] = gather_info
        if not args.no_cpu and world == 1:
            cb = cpu_baseline(dout, psi[:dout])
            try:
                cb["best_effort"] = cpu_best_effort(dout, psi[:dout])
            except Exception as e:
                cb["best_effort"] = {"error": str(e)}
            if X is not None:
                try:
                    cb["fit_em"] = cpu_em_baseline(X, y)
                except Exception as e:  # the EM leg is informative only
                    cb["fit_em"] = {"error": str(e)}
            res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if world > 1:
        eng.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
