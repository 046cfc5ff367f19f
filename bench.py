#!/usr/bin/env python3
"""bench.py - training-step throughput of the lamp hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU.  Started under a launcher (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) this process IS one rank.  Started bare, it spawns the N rank processes itself BEFORE anything touches the GPU, forwards
rank 0's line and exits non-zero unless all N ranks joined ONE RCCL communicator (`rccl_ranks` in the line = ncclCommCount).

A "step" is one pass of the hot path over one synthetic batch: Cnn.resnet(100) forward + backprop +
AdamW on a device-resident batch of B = 2048 CIFAR-shaped images per GPU (BASELINE.json config 3/4;
example-cifar100/src/main/scala/lamp/example/cifar/cnn.scala:89-137, run_cifar.sh:7), bf16 parameters
and activations with fp32 working copies in AdamW (AdamW.scala:48-85).  With N > 1 every rank computes
local gradients, one flat fp32 bucket is all-reduced over RCCL/xGMI, every rank applies the same step
(weak scaling: B per GPU fixed).  One JSON line is printed by rank 0.

Other workloads (parity-test configurations and secondary probes, not the headline line): --workload gemm | mlp | knn | attention | umap | lm.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3
PEAK_F64_TFLOPS = 78.6         # v_mfma_f64_16x16x4_f64: half the f32 rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; umap-e2e: 1 - a step is a complete 1M-point run)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 10; umap-e2e: 1)")
    ap.add_argument("--workload", default="resnet", choices=["resnet", "gemm", "mlp", "knn", "attention", "umap", "umap-e2e", "lm", "epoch"])
    ap.add_argument("--batch", type=int, default=2048, help="per-GPU batch (resnet)")
    ap.add_argument("--graph", action="store_true", help="lm on one GPU: capture forward + backprop into a HIP graph, replay it per step (optimiser eager); the default for resnet")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--umap-points", default="mixed", choices=["mixed", "weyl"], help="umap-e2e: closed form of the points (see the workload's comment)")
    ap.add_argument("--no-graph", action="store_true", help="resnet: issue the step eagerly instead of replaying forward + backprop from a HIP graph")
    ap.add_argument("--dry-launch", action="store_true", help="launch + rendezvous of the N ranks only (no GPU work): prints {\"dry_launch\": true, \"ranks\": N}")
    ap.add_argument("--min-window-s", type=float, default=0.5, help="repeat the K-step timed window until this much time is covered; the median window is reported")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 1 if a.workload == "umap-e2e" else (2 if a.workload == "epoch" else 20)
    if a.warmup is None:
        a.warmup = 1 if a.workload in ("umap-e2e", "epoch") else 10
    return a


def spawn_ranks(n):
    """bare `python bench.py --gpus N`: become the launcher.  N children (fresh interpreters, this process never initialises HIP), one
    per GPU, with the environment a torch.distributed.run launch would give them; rank 0's stdout is forwarded.  Exit code 0 only if
    every rank exits 0."""
    import socket
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rdzv = os.path.join(tempfile.gettempdir(), f"lamp_rdzv_{os.getpid()}_{port}.json")
    try:
        os.unlink(rdzv)                           # a leftover of a dead launch with the same pid and port
    except OSError:
        pass
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LAMP_RDZV_FILE=rdzv, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = rc or code or 1
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    for o in alive:
                        procs[o].terminate()
            time.sleep(0.05)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
        try:
            os.unlink(rdzv)
        except OSError:
            pass
    return rc


def closed_form_np(n, salt=0, scale=1.0):
    import numpy as np
    i = np.arange(n, dtype=np.int64) + salt
    return ((((i * 7919) % 1009).astype(np.float64) / 1009.0) - 0.5) * scale


def kernel_report(lib):
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    rows = []
    for line in buf.value.decode().splitlines():
        tag, n, ms, flops, byts = line.split()
        rows.append({"tag": tag, "launches": int(n), "total_ms": float(ms), "flops": float(flops), "bytes": float(byts)})
    return rows


def pmc_traffic(tag):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    separate runs of this same command, corrected as MI355X_MICROARCH.md prescribes; scripts/pmc_traffic.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        cls = json.load(open(files[-1]))["classes"].get(tag)
    except Exception:
        return None
    if not cls:
        return None
    return cls["traffic_bytes"], os.path.relpath(files[-1], ROOT)


def rocprof_class(tag):
    """average launch duration of a kernel class by the rocprofv3 kernel trace of this same command, from the newest committed
    profiles/r*_class_rocprof.json (scripts/class_rocprof.py builds it from the kernel-stats csv; the profile is of the EAGER step -
    rocprofv3 and hipGraphLaunch cannot be combined on this image - while the timed region replays the graph)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_class_rocprof.json")))
    if not files:
        return None
    try:
        j = json.load(open(files[-1]))
        cls = j["classes"].get(tag)
    except Exception:
        return None
    if not cls:
        return None
    return cls["avg_us"], os.path.relpath(files[-1], ROOT), j.get("note", "")


def roofline_of(rows):
    """roofline object for the kernel class that took the most time.  avg_us is the plain HIP-event bracket around every launch of
    the class (recorded on the launch's own stream in an UNTIMED pass after the timed region); nothing is subtracted: an (event,
    kernel, event) bracket reads 1 - 2 us longer than the rocprofv3 kernel trace of the same launch, so `frac` errs low."""
    if not rows:
        return None
    d = max(rows, key=lambda r: r["total_ms"])
    avg_s = d["total_ms"] / d["launches"] / 1e3
    extra = {"kernel": d["tag"], "launches": d["launches"], "avg_us": avg_s * 1e6, "timing": "hip events per launch, untimed pass, uncorrected"}
    traffic = pmc_traffic(d["tag"])
    if traffic is not None:
        extra["traffic_source"] = traffic[1]
    flops, byts = d["flops"], d["bytes"]            # per launch (lamp_kernel_timer_report averages the launchers' declarations)
    extra["algorithmic_bytes"] = byts
    extra["algorithmic_flops"] = flops
    traffic = traffic[0] if traffic is not None else None
    ai = flops / max(byts, 1.0)
    peak = PEAK_F32_TFLOPS if d["tag"].endswith("f32") else PEAK_F64_TFLOPS if d["tag"].endswith("f64") else PEAK_BF16_TFLOPS
    # compute bound on MI355X: arithmetic intensity above (about half of) the ridge of the pipe the class runs on - bf16 2.5 PF / 8 TB/s ~ 312
    # FLOP/B (threshold 150, conv / gemm tiles sit well above), f32 157 TF -> 20 (9.4), f64 78.6 TF -> 10 (4.7)
    if flops > 0 and ai > 150.0 * peak / PEAK_BF16_TFLOPS:
        if d["tag"].startswith("knn_split_f16x"):
            # the f32 search on the 16-bit matrix pipe (kernels/knn_split.hip): every f32 product is 3 f16 products (two planes per value).
            # The launcher declares the ALGORITHMIC work of the search; the roofline is priced on what the f16 pipe executes (same peak as bf16).
            terms = int(d["tag"][-1])
            extra["f16_products_per_f32_product"] = terms
            flops = flops * terms
            extra["executed_flops"] = flops
        ach = flops / avg_s / 1e12
        rp = rocprof_class(d["tag"])
        if rp is not None and not d["tag"].startswith("knn_split_f16x"):
            # labelled, never replacing the event-bracketed figure above: the committed kernel trace of this command
            extra["avg_us_rocprof"] = rp[0]
            extra["frac_rocprof"] = flops / (rp[0] * 1e-6) / 1e12 / peak
            extra["rocprof_source"] = rp[1] + " (committed rocprofv3 kernel trace of the eager step on another box; not measured in this run)"
        if d["tag"].startswith("knn_split_f16x"):
            extra["frac_algorithmic"] = d["flops"] / avg_s / 1e12 / peak      # SURVEY 8(d)'s definition: f32 products of the search, not the f16 products executed
        return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, **extra}
    ach = byts / avg_s / 1e9
    if d["tag"] == "umap_pairs2":
        # the layout kernel is bound by memory-side float atomics (64-byte requests, ~20 G/s chip-wide measured), not by bytes: its
        # launcher declares the UNMERGED request count (two per pair) in the flops slot; the kernel merges runs before issuing them
        extra["algorithmic_flops"] = 0.0
        extra["atomic_requests_unmerged"] = flops
        extra["unmerged_atomic_request_rate_Gps"] = flops / avg_s / 1e9
        extra["note"] = ("bound by memory-side f64 atomics (~20 G requests/s chip-wide on this part, measured): the byte fraction below "
                         "understates the kernel; the rate above counts two requests per pair, of which the segmented scan merges ~45 %")
    return {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": traffic, **extra}


def verify_data_parallel(lib, S, np, C, dist, comm, model, model_mod, opt, x, target, acc, B, rank, world, local_rank, modes):
    """Untimed self-check of the N-rank step (distributed/package.scala:690-759).
    1. One step on RAGGED per-rank batches (rank r drops its last 8 r samples, so the example weights differ): the local gradients
       (x n_r, f32) of every rank are all-gathered over RCCL, the product's exchange (exchange_and_step: bucket, all-reduce, division)
       runs on the same gradients, and rank 0 compares the result with sum(n_r g_r) / sum(n_r) computed in f64 on the host.
    2. The all-reduce of each exchange mode bracketed by HIP events (kernel timer class rccl_all_reduce) -> allreduce_us.
    3. Every rank hashes its module state + optimiser state; rank 0 gathers the digests over the control plane."""
    import hashlib
    out = {}
    nb = max(8, B - 8 * rank)
    xs, ts = x.slice(0, 0, nb), target.slice(0, 0, nb)
    n, grads = model.addTotalLossAndReturnGradientsAndNumExamples(xs, ts, acc)
    local = np.concatenate([g.to_numpy().astype(np.float32).ravel() for g in grads] + [np.array([1.0], np.float32)]) * np.float32(n)
    mine = S.STen.from_numpy(local, local_rank, S.F32)
    gathered = S.STen.zeros([world * local.size], S.F32, local_rank)
    lib.lamp_comm_all_gather(gathered, mine, comm)
    model.exchange_and_step(opt, grads, n, comm)               # grads now hold the averaged gradients, rounded to their dtype
    got = np.concatenate([g.to_numpy().astype(np.float64).ravel() for g in grads])
    allg = gathered.to_numpy().astype(np.float64).reshape(world, local.size)
    total_n = allg[:, -1].sum()
    want = allg[:, :-1].sum(0) / total_n
    worst, o = 0.0, 0
    for g in grads:
        k = int(np.prod(g.shape)) if g.shape else 1
        den = np.abs(want[o:o + k]).max()
        worst = max(worst, float(np.abs(got[o:o + k] - want[o:o + k]).max() / (den if den > 0 else 1.0)))
        o += k
    out["grad_avg_max_rel_err"] = worst
    out["grad_avg_examples_per_rank"] = [int(v) for v in allg[:, -1]]
    out["grad_avg_check"] = ("per tensor max|averaged - sum(n_r g_r)/sum(n_r)| / max|.|, the per-rank gradients all-gathered over RCCL, "
                             "reference sum in f64; bound 2^-7 (the averaged gradients are rounded to bf16)")
    # 2. all-reduce time per exchange mode
    lib.lamp_device_synchronize(); dist.barrier()
    ar = {}
    for name, fn in (modes or []):
        lib.lamp_kernel_timer_filter(b"rccl_all_reduce")
        lib.lamp_kernel_timer_enable(1)
        for _ in range(4):
            fn()
        lib.lamp_device_synchronize(); dist.barrier()
        lib.lamp_kernel_timer_enable(0)
        lib.lamp_kernel_timer_filter(None)
        rows = [r for r in kernel_report(lib) if r["tag"] == "rccl_all_reduce"]
        if rows:
            ar[name.split(":")[0]] = {"launches_per_step": rows[0]["launches"] / 4, "avg_us": rows[0]["total_ms"] / rows[0]["launches"] * 1e3,
                                      "bytes_per_launch": rows[0]["bytes"]}
    out["allreduce_us"] = ar
    # 3. replicas identical
    h = hashlib.sha256()
    for v in model_mod.state:
        h.update(np.ascontiguousarray(v.value.to_numpy()).tobytes())
    for t in opt.state:
        h.update(np.ascontiguousarray(t.to_numpy()).tobytes())
    digests = dist.all_gather(h.hexdigest())
    out["replicas_identical"] = len(set(digests)) == 1
    out["state_sha256"] = digests[0][:16]
    ok = out["replicas_identical"] and worst <= 2.0 ** -7
    if not ok:
        if rank == 0:
            print(f"bench.py: data-parallel self-check FAILED, refusing to report: {json.dumps(out)}; digests {digests}", file=sys.stderr)
        lib.lamp_device_synchronize()
        dist.barrier()
        raise SystemExit(3)
    return out


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))             # launcher: never touches the GPU itself
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when a communicator
    # is created): from here on file descriptor 1 is stderr, and the line goes out through a private duplicate of the original stdout.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if a.dry_launch:
        # the launch + rendezvous path without a GPU: every rank joins the control plane, receives rank 0's 128 id bytes, and the
        # collectives the timing protocol uses (barrier, max over ranks) run once
        from lamp_amd import distributed as D
        cp = D.init_control_plane(timeout=60.0)
        uid = D.exchange_unique_id(cp, lambda: bytes((7 * i + 1) % 256 for i in range(128)))
        assert uid == bytes((7 * i + 1) % 256 for i in range(128))
        ranks = cp.all_gather(rank)
        slowest = cp.all_reduce_max(float(rank))
        cp.barrier()
        if rank == 0:
            print(json.dumps({"dry_launch": True, "ranks": len(ranks), "rank_list": ranks, "max_rank": slowest, "n_gpus": a.gpus}), file=line_out, flush=True)
        cp.close()
        return

    # CPU baseline first, in a child process, on rank 0 at N = 1 only (before this process touches the GPU)
    cpu_baseline = None
    if rank == 0 and a.gpus == 1 and not a.no_cpu_baseline and a.workload in ("resnet", "gemm", "mlp", "knn", "umap", "umap-e2e"):
        try:
            wl = a.workload                   # kNN / UMAP: a bounded sample scaled to the unit of the line (oracle/cpu_baseline.py says how)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--workload", wl, "--budget-s", "15"],
                                 capture_output=True, text=True, timeout=600)
            cpu_baseline = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is informative; never fail the GPU measurement because of it
            cpu_baseline = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    from lamp_amd._capi import lib
    lib.load()
    ngpu = C.c_int(0)
    lib.lamp_get_num_gpus(C.byref(ngpu))
    if local_rank >= ngpu.value:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but this node shows {ngpu.value} GPU(s)")
    lib.lamp_set_device(local_rank)
    from lamp_amd import nn, sten as S
    import numpy as np

    dist = None
    comm = None
    rccl_ranks = 1
    # LAMP_BENCH_FORCE_COMM=1 (with RANK=0 WORLD_SIZE=1 in the environment) drives the complete multi-rank code path - control plane,
    # unique-id exchange, RCCL communicator, two-bucket overlapped exchange - on a single GPU
    if world > 1 or (os.environ.get("LAMP_BENCH_FORCE_COMM") == "1" and "RANK" in os.environ):
        from lamp_amd import distributed as D
        dist = D.init_control_plane()            # TCP control plane: unique id, barrier, max of times - no tensor library
        comm = D.rccl_communicator(dist)         # RCCL communicator: the data plane over xGMI
        rccl_ranks = D.comm_count(comm)
        if rccl_ranks != a.gpus:
            raise SystemExit(f"bench.py: RCCL communicator has {rccl_ranks} ranks, --gpus says {a.gpus}: refusing to report")

    def barrier():
        lib.lamp_device_synchronize()
        if dist is not None:
            dist.barrier()

    dtype = {"bf16": S.BF16, "f32": S.F32, "f64": S.F64}[a.dtype]
    result_extra = {}
    if a.workload == "resnet":
        B = a.batch
        lib.lamp_manual_seed(1234)                                   # same initial weights on every rank
        model_mod = nn.resnet(100, 0.0, dtype, local_rank)
        x = S.STen.from_numpy(closed_form_np(B * 3 * 32 * 32, 5 + rank * 7919).reshape(B, 3, 32, 32).astype(np.float32), local_rank, dtype)
        target = S.STen.from_numpy(((np.arange(B) * 7 + rank) % 100).astype(np.int64), local_rank)
        cw = S.STen.ones([100], dtype, local_rank)
        model = nn.SupervisedModel(model_mod, nn.SupervisedModel.NLL, cw)
        opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=(a.dtype == "bf16"))([p.value for p in model_mod.parameters])
        acc = S.STen.zeros([1], dtype, local_rank)
        step = lambda: model.train_step(opt, x, target, acc, comm)
        units_per_step = B
        metric, unit = "training-step samples/sec", "samples/s"
        config = {"workload": "example-cifar100 Cnn.resnet(100) training step (fwd+backprop+AdamW), synthetic CIFAR batch",
                  "per_gpu_batch": B, "global_batch": B * a.gpus, "parallelism": f"dp{a.gpus}" if a.gpus > 1 else "single",
                  "optimizer": "AdamW lr 1e-3 wd 0 beta2 0.95" + (" mixedPrecision" if a.dtype == "bf16" else "")}
    elif a.workload == "epoch":
        # the reference's own number is "instances/sec" of an EPOCH (IOLoops.scala:728-743) through
        # BatchStream.minibatchesFromFull (BatchStream.scala:528-592: shuffle -> gather -> pinned staging -> copy on another stream, one batch
        # ahead: IOLoops.scala:833-874).  A step here = one epoch of lamp_amd.loops.oneEpoch over 50 000 CIFAR-shaped records (u8 pixels cast
        # to float, as Cifar.loadImageFile does) with the eager training step; every variant of where the data set lives is timed and reported,
        # the headline value is the host-resident arrangement with the cast folded into the gather, at B = --batch.
        from lamp_amd import loops
        from lamp_amd.data import BatchStream
        NREC = 50000
        lib.lamp_manual_seed(1234)
        model_mod = nn.resnet(100, 0.0, dtype, local_rank)
        cw = S.STen.ones([100], dtype, local_rank)
        model = nn.SupervisedModel(model_mod, nn.SupervisedModel.NLL, cw)
        opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=(a.dtype == "bf16"))([p.value for p in model_mod.parameters])
        pix = ((np.arange(NREC * 3 * 32 * 32, dtype=np.int64) * 2654435761 + rank * 97) >> 9) % 256
        pix = pix.astype(np.uint8).reshape(NREC, 3, 32, 32)
        labels = S.STen.from_numpy(((np.arange(NREC) * 7 + rank) % 100).astype(np.int64), S.CPU)
        # pinned ONCE, as a training run does (cifar100.scala --pinned): every stream over them shares these buffers.  (Measured,
        # scripts/pinned_placement_probe.py: a pinned buffer that is freed and allocated again between variants lands on differently
        # fragmented pages now and then and gathers at 19 - 39 GB/s instead of 44 - 53.)
        host_u8 = S.STen.from_numpy(pix, S.CPU).pin()
        host_f32 = S.STen.from_numpy(pix.astype(np.float32), S.CPU).pin()
        dev_x = S.STen.from_numpy(pix.astype(np.float32), local_rank, dtype)
        order = np.random.default_rng(7).permutation(NREC)

        def stream_of(kind, B):
            if kind == "device_resident":
                return BatchStream.minibatchesFromFull(B, False, dev_x, labels, order=order, device=local_rank)
            src = host_f32 if kind == "host_f32_pinned" else host_u8
            return BatchStream.minibatchesFromFull(B, False, src, labels, order=order, device=local_rank, hostResident=True, outDtype=dtype)

        def epoch_rate(kind, B, epochs):
            st = stream_of(kind, B)
            loops.oneEpoch(0, model, opt, st)                      # warm-up epoch (allocator, packed weights, pinning)
            barrier()
            t0 = time.perf_counter()
            for e in range(epochs):
                loops.oneEpoch(e + 1, model, opt, st)
                if os.environ.get("LAMP_EPOCH_TRACE"):
                    barrier()
                    r_, u_, m_, d_ = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
                    lib.lamp_allocator_stats(local_rank, C.byref(r_), C.byref(u_), C.byref(m_)); lib.lamp_allocator_deferred_frees(local_rank, C.byref(d_))
                    sys.stderr.write(f"epoch trace {kind} B={B} epoch {e}: {time.perf_counter() - t0:.3f} s since start; reserved {r_.value >> 20} MB in use {u_.value >> 20} MB mallocs {m_.value} deferred frees {d_.value}\n")
            barrier()
            rate = NREC * epochs / (time.perf_counter() - t0)
            del st
            import gc; gc.collect()
            return rate

        epoch_variants = {}
        for kind in os.environ.get("LAMP_EPOCH_VARIANTS", "device_resident,host_f32_pinned,host_u8_pinned").split(","):
            for B in sorted({a.batch, 256}, reverse=True):
                epoch_variants[f"{kind}_B{B}"] = epoch_rate(kind, B, a.steps)
        # headline: the records stay on the host as the u8 pixels of the CIFAR file and the cast Cifar.loadImageFile applies (castToFloat) runs
        # inside the gather: 3 KB instead of 12 KB per record over PCIe.  (Measured: shader reads of host memory do not overlap the training
        # step - the float variant, 25 MB per batch, costs its full 0.4 ms per batch - while a DMA-engine copy of the same bytes is free;
        # scripts/epoch_interference_probe.py.)
        headline = stream_of("host_u8_pinned", a.batch)
        step = lambda: loops.oneEpoch(0, model, opt, headline)
        units_per_step = NREC
        metric, unit = "epoch instances/sec (IOLoops.oneEpoch over minibatchesFromFull)", "samples/s"
        config = {"workload": "example-cifar100 Cnn.resnet(100): one epoch of 50 000 CIFAR-shaped records through BatchStream.minibatchesFromFull "
                              "(host-resident u8 records in pinned memory, minibatch gathered over PCIe and cast by a GPU kernel queued one batch ahead), eager training step",
                  "per_gpu_batch": a.batch, "records": NREC, "parallelism": "single"}
        result_extra["epoch_variants_instances_per_s"] = epoch_variants
    elif a.workload == "lm":
        # example-autoregressivelm (model.scala:9-37, train.scala:40-66): byte-level GPT, 12 blocks x 768 x 12 heads, context 384,
        # bf16 parameters + mixed-precision AdamW (weight decay on the attention / MLP matrices only, clip 1), DP over the ranks
        from lamp_amd import transformer as TR
        B = a.batch if a.batch != 2048 else 64
        ctx, vocab, dim, heads, blocks = 384, 256, 768, 12, 12
        lib.lamp_manual_seed(1234)
        net = TR.LanguageModelLoss(ctx, vocab, blocks, dim, dim // heads, heads, dim * 4, 0.0, -1000, dtype, local_rank)
        toks = ((np.arange(B * (ctx + 1), dtype=np.int64) * 2654435761 + rank * 97) >> 7) % vocab
        toks = toks.reshape(B, ctx + 1)
        x = S.STen.from_numpy(np.ascontiguousarray(toks[:, :-1]), local_rank)
        target = S.STen.from_numpy(np.ascontiguousarray(toks[:, 1:]), local_rank)
        model = nn.SupervisedModel(net, nn.SupervisedModel.IDENTITY)
        params = net.parameters
        # state order per block: wQ wK wV wO w1 w2 b1 b2 scale1 scale2 after the two embeddings; decay on the six matrices
        wd = [0.0, 0.0] + [0.1 if i % 10 < 6 else 0.0 for i in range(10 * blocks)]
        opt = nn.AdamW_tagged([p.value for p in params], wd, 1e-4, 0.9, 0.95, clip=1.0, mixedPrecision=(a.dtype == "bf16"))
        acc = S.STen.zeros([1], dtype, local_rank)
        step = lambda: model.train_step(opt, x, target, acc, comm)
        units_per_step = B * ctx
        metric, unit = "language-model training tokens/sec", "tokens/s"
        config = {"workload": "example-autoregressivelm LanguageModelLoss training step (12 x 768 x 12 heads, context 384, vocabulary 256), synthetic tokens",
                  "attention": ("as written for CUDA: (batch, sequence, heads, d) views read as (batch, heads, sequence, d)" if os.environ.get("LAMP_ATTENTION_AS_WRITTEN_FOR_CUDA") == "1"
                                else "reference CPU semantics: causal per-head attention over the 384 positions (flash kernels on strided views)"),
                  "per_gpu_batch": B, "global_batch": B * a.gpus, "parallelism": f"dp{a.gpus}" if a.gpus > 1 else "single",
                  "optimizer": "AdamW lr 1e-4 wd 0.1 (matrices) beta2 0.95 clip 1" + (" mixedPrecision" if a.dtype == "bf16" else "")}
    elif a.workload == "gemm":
        n = 4096
        A_ = S.STen.from_numpy((closed_form_np(n * n, 1, 2.0)).reshape(n, n).astype(np.float32), local_rank, dtype)
        W_ = S.STen.from_numpy((closed_form_np(n * n, 77, 2.0)).reshape(n, n).astype(np.float32), local_rank, dtype)
        bias = S.STen.from_numpy(closed_form_np(n, 3, 1.0).reshape(1, n).astype(np.float32), local_rank, dtype)
        P_ = S.STen.ones([n, n], dtype, local_rank)
        dW, dX = S.STen.zeros([n, n], dtype, local_rank), S.STen.zeros([n, n], dtype, local_rank)
        out_h = C.c_void_p()

        def step():
            o = C.c_void_p()
            lib.lamp_linear_bias(C.byref(o), A_, W_, bias)                   # y = x.W + b
            y = S.STen(o)
            S.STen.addmm_out_transposed1(dW, dW, A_, P_, 1.0, 1.0)          # dW += x^T p
            S.STen.addmm_out_transposed2(dX, dX, P_, W_, 1.0, 1.0)          # dX += p W^T
            return y
        units_per_step = 3 * 2.0 * n ** 3 / 1e12
        metric, unit = "4096x4096 addmm fwd+bwd", "TFLOP/s"
        config = {"workload": "lamp Linear(4096,4096,bias) on x[4096,4096]: fwd addmm + dW (A^T.p) + dX (p.W^T)", "parallelism": "replicas"}
    elif a.workload == "knn":
        # BASELINE config 5, first half: brute-force kNN graph, 131072 query rows per step against 1M x 128 f32 points, k = 10
        n, nq, d, k = 1_000_000, 131_072, 128, 10
        rng = np.random.default_rng(0)
        pts = rng.random((n, d), dtype=np.float32) + (np.arange(n) % 16)[:, None].astype(np.float32)
        X = S.STen.from_numpy(pts, local_rank, S.F32)
        Qs = X.slice(0, rank * nq % (n - nq), rank * nq % (n - nq) + nq)

        def step():
            i = C.c_void_p()
            lib.lamp_knn_squared_euclidean(C.byref(i), None, X, Qs, k)
            return S.STen(i)
        units_per_step = nq
        metric, unit = "kNN queries/sec (1M x 128 f32 points, k = 10)", "queries/s"
        config = {"workload": "lamp.knn.knnSearch squared Euclidean, 131072 queries x 1M points x 128 features, k = 10 (split-f16 filter with the top-k fused in, exact f32 re-rank with a per-query proof, exact f32 MFMA kernel for the unproven queries)",
                  "parallelism": "query rows sharded" if a.gpus > 1 else "single"}
        a.dtype = "f32"
    elif a.workload == "attention":
        # ScaledDotProductAttention forward + backward, bf16, B 8 x 16 heads x S 4096 x d 128, non-causal
        Bz, H, Sq, D = 8, 16, 4096, 128
        rng = np.random.default_rng(0)
        q_, k_, v_, g_ = (S.STen.from_numpy(rng.standard_normal((Bz, H, Sq, D), dtype=np.float32), local_rank, S.BF16) for _ in range(4))

        def step():
            o, l = C.c_void_p(), C.c_void_p()
            lib.lamp_scaled_dot_product_attention(C.byref(o), C.byref(l), q_, k_, v_, 0, 0.0)
            O_, L_ = S.STen(o), S.STen(l)
            out3 = (C.c_void_p * 3)()
            lib.lamp_scaled_dot_product_attention_backward(out3, g_, q_, k_, v_, O_, L_, 0, 0.0)
            return [S.STen(h) for h in out3]
        units_per_step = 14.0 * Bz * H * Sq * Sq * D / 1e12      # 4 fwd + 10 bwd model flops per (query, key, feature)
        metric, unit = "attention fwd+bwd (model flops)", "TFLOP/s"
        config = {"workload": "ScaledDotProductAttention fwd + bwd, bf16, batch 8 x 16 heads x 4096 x 128, non-causal", "parallelism": "replicas"}
    elif a.workload == "umap-e2e":
        # BASELINE config 5 as ONE job: 1M x 128 f32 points resident in HBM -> kNN graph (k = 10, f32 search as `precision = SinglePrecision`) ->
        # exact f64 neighbour distances -> edge weights -> 500 layout iterations (f64, 5 negatives per edge, AdamW); a step = the whole run
        from lamp_amd import umap as U
        n, d_, kk, iters = 1_000_000, 128, 10, 500
        # closed-form pseudo-random points in [0, 1) + 16 clusters, "ties avoided by construction".  The survey's example,
        # frac((i d + j) 2654435761 / 2^32), is a rank-1 lattice (point i + 1 is point i shifted by one constant modulo 1 in every
        # coordinate): thousands of EXACTLY tied neighbour distances per query, i.e. no well defined neighbour sets.  The same index goes
        # through murmur3's 32-bit finaliser instead (--umap-points weyl: the lattice, as measured in rounds 1 - 2).
        def make_points(kind):
            idx_ = np.arange(n, dtype=np.uint64)[:, None] * np.uint64(d_) + np.arange(d_, dtype=np.uint64)[None, :]
            if kind == "weyl":
                h_ = (idx_ * np.uint64(2654435761)) % np.uint64(2 ** 32)
            else:
                m32 = np.uint64(0xffffffff)
                h_ = idx_ & m32
                h_ ^= h_ >> np.uint64(16); h_ = (h_ * np.uint64(0x85ebca6b)) & m32
                h_ ^= h_ >> np.uint64(13); h_ = (h_ * np.uint64(0xc2b2ae35)) & m32
                h_ ^= h_ >> np.uint64(16)
            pts = h_.astype(np.float64) / 2.0 ** 32 + (np.arange(n) % 16)[:, None]
            del idx_, h_
            return S.STen.from_numpy(pts.astype(np.float32), local_rank, S.F32), S.STen.from_numpy(pts, local_rank, S.F64)
        umap_pts = make_points(a.umap_points)
        phase = {}

        def run(timed_phases=None, X32=None, X64=None):
            X32 = X32 if X32 is not None else umap_pts[0]
            X64 = X64 if X64 is not None else umap_pts[1]
            def mark(name, t0):
                if timed_phases is not None:
                    lib.lamp_device_synchronize(); timed_phases[name] = time.perf_counter() - t0
                return time.perf_counter()
            t0 = time.perf_counter()
            knn = U.knn_search(X32, X32, kk, 1000)
            t0 = mark("knn_graph_s", t0)
            dd = C.c_void_p(); lib.lamp_knn_row_distances(C.byref(dd), X64, knn)
            dist = S.STen(dd)
            t0 = mark("neighbour_distances_s", t0)
            ew = U.edge_weights(dist, knn)
            t0 = mark("edge_weights_s", t0)
            layout, loss = U.optimize(ew, n, 0.1, iters, 0.0, 5, 42, True, 1.0, local_rank, 2)
            mark("layout_500_iterations_s", t0)
            return layout, loss, ew.shape[0]
        step = lambda: run()
        units_per_step = n
        metric, unit = "UMAP end-to-end points/sec (1M x 128: kNN graph + edge weights + 500-iteration layout)", "points/s"
        config = {"workload": "lamp.umap.Umap.umap on 1M x 128 synthetic points (closed form: murmur3-finalised index + 16 clusters) already resident in HBM: knnSearch (f32, k = 10) -> f64 neighbour distances -> "
                              "edgeWeights -> optimize (500 iterations, 5 negatives per edge, f64, AdamW clip 1)", "parallelism": "replicas"}
        a.dtype = "f32 kNN / f64 layout"
    elif a.workload == "umap":
        # BASELINE config 5, second half: one layout iteration of Umap.optimize at 1M points (9M edges, 5 negatives per edge, f64)
        from lamp_amd import umap as U
        n, kk = 1_000_000, 10
        rng = np.random.default_rng(0)
        knn_idx = (np.arange(n)[:, None] + 1 + rng.integers(0, n - 1, (n, kk))) % n
        knn_idx[:, 0] = np.arange(n)
        knn_dist = np.sort(rng.random((n, kk)), 1); knn_dist[:, 0] = 0.0
        ew = U.edge_weights(S.STen.from_numpy(knn_dist, local_rank, S.F64), S.STen.from_numpy(knn_idx.astype(np.int64), local_rank))
        state = {}

        def step():
            state["r"] = U.optimize(ew, n, 0.1, 1, 0.0, 5, 42 + len(state), True, 1.0, local_rank, 2)
        units_per_step = 1
        metric, unit = "UMAP layout iterations/sec (1M points)", "iterations/s"
        config = {"workload": "Umap.optimize, 1 iteration per step: 1M points, ~9M edges, 5 negatives per edge, f64, AdamW; includes the per-call "
                              "initial layout (rand) that a 500-iteration run pays once", "parallelism": "replicas"}
        a.dtype = "f64"
    else:
        B = 1024
        model_mod = nn.Sequential(nn.MLP(784, 10, [256], S.F32, local_rank), nn.Fun("logsoftmax", 1))
        x = S.STen.from_numpy(closed_form_np(B * 784).reshape(B, 784).astype(np.float32), local_rank)
        target = S.STen.from_numpy((np.arange(B) % 10).astype(np.int64), local_rank)
        model = nn.SupervisedModel(model_mod, 0, S.STen.ones([10], S.F32, local_rank))
        acc = S.STen.zeros([1], S.F32, local_rank)
        step = lambda: model.addTotalLossAndReturnGradientsAndNumExamples(x, target, acc)
        units_per_step = B
        metric, unit = "MLP fwd+bwd samples/sec", "samples/s"
        config = {"workload": "MLP 784-256-10 fwd+bwd fp32 batch 1024", "parallelism": "replicas"}
        a.dtype = "f32"

    graph = None
    step_eager = step
    modes = None                                  # multi-rank: [(name, step function)] - every exchange mode is timed, the faster one is `value`
    use_graph = not a.no_graph and ((a.workload in ("resnet", "mlp")) or (a.graph and a.workload == "lm"))
    under_profiler = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    if use_graph and not os.environ.get("LAMP_BENCH_GRAPH_UNDER_PROFILER") and under_profiler:
        # rocprofv3's kernel tracing dies with SIGSEGV inside hipGraphLaunch on this image (5 of 8 runs on one box, never without the
        # profiler; the faulting frames are the profiler's graph hooks under lamp_graph_launch): a profiled run measures the eager step,
        # whose kernels are the replayed ones, and says so
        use_graph = False
        config["hip_graph"] = "off: running under rocprofv3 (eager step, same kernels)"
    if use_graph:
        # The ResNet step is ~110 short launches: issued one by one the host needs ~0.5 ms of a 1.5 ms step and the device idles ~10 %
        # between kernels.  Forward + backprop are captured once into a HIP graph (after one eager step has set attributes and filled
        # the caches) and replayed per step - the same kernels on the same buffers, launched back to back; the optimiser (its step count
        # is a host value) stays eager.  --no-graph measures the eager step; the untimed roofline passes always run eagerly, because
        # HIP events cannot bracket the kernels of a replayed graph.
        lib.lamp_device_synchronize()            # weights / optimiser state were written on the null stream
        st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, local_rank, C.byref(st)); lib.lamp_stream_set_current(st)
        step()
        lib.lamp_device_synchronize()
        lib.lamp_graph_begin_capture()
        captured_n, captured_grads = model.addTotalLossAndReturnGradientsAndNumExamples(x, target, acc)
        graph = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(graph))
        if a.workload == "mlp":                   # config 1 is forward + backward only
            def step():
                lib.lamp_graph_launch(graph)
                return units_per_step
            config["hip_graph"] = "forward + backprop replayed from a HIP graph (--no-graph: eager step)"
        elif comm is None:
            def step():
                lib.lamp_graph_launch(graph)
                opt.step(captured_grads, 1.0)
                return units_per_step
            config["hip_graph"] = "forward + backprop replayed from a HIP graph, optimiser eager (--no-graph: eager step)"
        else:
            # N ranks: the same replay, then averageGradients over RCCL (one flat fp32 bucket * n, all-reduce, / sum n) and the optimiser.
            # The eager step (--no-graph) overlaps the exchange of the deep layers with the rest of backward instead; replaying the graph
            # saves more (~90 us of launch gaps per step) than the overlap hides (one ~1.5 MB all-reduce).
            def step():
                lib.lamp_graph_launch(graph)
                model.exchange_and_step(opt, captured_grads, captured_n, comm)
                return units_per_step
            config["hip_graph"] = ("forward + backprop replayed from a HIP graph, then gradient all-reduce (RCCL) + optimiser eager "
                                   "(--no-graph: eager step with the exchange overlapped with backward)")
            # which of the two wins depends on what the all-reduce costs between N real GPUs - a 1-rank communicator cannot tell
            # (VERDICT r2): both are timed in this run
            modes = [("graph_single_bucket: forward + backprop replayed from a HIP graph, one flat fp32 bucket all-reduced after backward", step),
                     ("eager_overlapped_two_buckets: eager step, deep bucket (90 % of the elements) all-reduced on a second stream during the rest of backward", step_eager)]
    if comm is not None and modes is None and a.workload in ("resnet", "lm"):
        modes = [("eager_overlapped_two_buckets: eager step, deep bucket (90 % of the elements) all-reduced on a second stream during the rest of backward", step_eager)]
    if comm is not None and a.workload in ("resnet", "lm"):
        # what a data-parallel run does before its first batch: rank 0's module + optimiser state on every rank (the replicas are already
        # identical here - same seed - so this changes no value; it puts the broadcast path on the wire before the measurement)
        model.sync_state(opt, comm, 0)
        config["dp_state_sync"] = "module + optimiser state broadcast from rank 0 before the first step"
    # ---- timed region: EXACTLY K steps between barrier + device synchronize on both sides, no instrumentation inside (kernel timers
    # off).  A K-step window shorter than --min-window-s is repeated and the MEDIAN window reported (every window is a complete
    # measurement by the contract; 20 ResNet steps are 30 ms, which one scheduling hiccup of the host distorts).
    def measure(fn):
        for _ in range(a.warmup):
            fn()
        lib.lamp_kernel_timer_enable(0)
        windows, enqueues = [], []
        total = 0.0
        while True:
            barrier()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn()
            enq = time.perf_counter() - t0            # host time to issue the steps (the device may still be running)
            barrier()
            w = time.perf_counter() - t0
            if dist is not None:
                w = dist.all_reduce_max(w)             # MAX over ranks; the same value on every rank, so all ranks stop together
            windows.append(w); enqueues.append(enq)
            total += w
            if total >= a.min_window_s or len(windows) >= 200:
                break
        order = sorted(range(len(windows)), key=lambda k: windows[k])
        mid = order[len(order) // 2]
        return windows[mid], enqueues[mid], windows

    alt_mode = None
    if modes:
        timed = [(name, fn) + measure(fn) for name, fn in modes]
        timed.sort(key=lambda t: t[2])             # the medians are already the MAX over ranks: every rank sorts alike
        name, step, elapsed, enqueue, windows = timed[0]
        config["exchange_mode"] = name
        alt_mode = [{"mode": t[0], "ms_per_step": t[2] / a.steps * 1e3, "value": units_per_step * a.gpus * a.steps / t[2],
                     "windows": len(t[4])} for t in timed[1:]]
    else:
        elapsed, enqueue, windows = measure(step)

    # ---- multi-rank self-verification (untimed, after the measurement): the line is only printed when the
    # replicas hold bit-identical state and the averaged gradients are the example-weighted mean of the per-rank gradients
    verify = None
    if comm is not None and a.workload == "resnet":
        verify = verify_data_parallel(lib, S, np, C, dist, comm, model, model_mod, opt, x, target, acc, B, rank, world, local_rank, modes)

    # ---- untimed passes for the roofline object (after the measurement, so they cannot disturb it)
    # 1. classification: every tagged launch bracketed by HIP events (on the stream it is launched on) -> per-class table
    PROFILE_STEPS = 2
    barrier()
    lib.lamp_kernel_timer_filter(None)
    lib.lamp_kernel_timer_enable(1)
    for _ in range(PROFILE_STEPS):
        step_eager()
    barrier()
    lib.lamp_kernel_timer_enable(0)
    class_rows = kernel_report(lib)
    dominant = max(class_rows, key=lambda r: r["total_ms"])["tag"] if class_rows else None
    # 2. the dominant class alone (events on ~10 launches per step instead of ~100: the kernels around it run as in the timed region)
    rows = []
    if dominant:
        lib.lamp_kernel_timer_filter(dominant.encode())
        lib.lamp_kernel_timer_enable(1)
        for _ in range(max(PROFILE_STEPS, min(a.steps, 10))):
            step_eager()
        barrier()
        lib.lamp_kernel_timer_enable(0)
        lib.lamp_kernel_timer_filter(None)
        rows = kernel_report(lib)

    if a.workload == "knn":
        # which search ran: the filter's own sample decides per data set whether the f16 pass pays (kernels/knn_split.hip)
        pl = C.c_int(0); lib.lamp_knn_split_last_planes(C.byref(pl))
        nf = C.c_int64(0); lib.lamp_knn_split_last_failed(C.byref(nf))
        result_extra["knn_path"] = (f"split-f16 filter ({pl.value} planes) + exact re-rank; {nf.value} of {units_per_step} queries went to the exact kernel"
                                    if pl.value else "exact f32 kernel for every query (the filter's sample predicted too few provable queries)")
    if a.workload == "umap-e2e":
        ph = {}
        _, last_loss, n_edges = run(ph)            # untimed, instrumented pass: seconds per phase (device synchronised between phases)
        result_extra["phases"] = ph
        result_extra["edges"] = int(n_edges)
        result_extra["final_loss"] = float(last_loss)
        if a.umap_points == "mixed" and a.gpus == 1:
            # the headline points differ from SURVEY 8d's example (a lattice of exactly tied distances, where the f16 kNN filter
            # cannot prove anything and the exact kernel runs): the same job on the survey's points, one warm-up and one timed run, so that the
            # line stays comparable with rounds 1 - 2
            sx32, sx64 = make_points("weyl")
            run(None, sx32, sx64)
            lib.lamp_device_synchronize()
            t_s = time.perf_counter()
            run(None, sx32, sx64)
            lib.lamp_device_synchronize()
            t_s = time.perf_counter() - t_s
            result_extra["survey_points"] = {"points": "SURVEY 8d's closed form frac((i d + j) 2654435761 / 2^32) + 16 clusters (a rank-1 lattice: tied distances)",
                                             "seconds": t_s, "value": n / t_s, "unit": "points/s"}
            del sx32, sx64
        lay_rows = [r for r in class_rows if r["tag"] == "umap_pairs2"]
        knn_rows = [r for r in class_rows if r["tag"].startswith("knn_fused") or r["tag"].startswith("knn_split")]
        if lay_rows:
            result_extra["roofline_layout"] = roofline_of(lay_rows)
        if knn_rows:
            result_extra["roofline_knn"] = roofline_of(knn_rows)
    # ---- the same K-step protocol, after the headline measurement and its roofline passes (never inside its timed region),
    # for the step in the reference example's own precisions (cifar100.scala:127-129) and at the small batch SURVEY 8d config 3 also asks for:
    # fresh model, one eager step, forward + backprop captured into a HIP graph, warm-up, K-step windows between device synchronisations, median.
    also = None
    if a.workload == "resnet" and a.gpus == 1 and rank == 0 and os.environ.get("LAMP_BENCH_ALSO", "1") != "0" and not under_profiler:
        def variant_ms(dt_name, Bv):
            dtv = {"bf16": S.BF16, "f32": S.F32, "f64": S.F64}[dt_name]
            lib.lamp_manual_seed(1234)
            mm = nn.resnet(100, 0.0, dtv, local_rank)
            xv = S.STen.from_numpy(closed_form_np(Bv * 3 * 32 * 32, 5).reshape(Bv, 3, 32, 32).astype(np.float32), local_rank, dtv)
            tv = S.STen.from_numpy(((np.arange(Bv) * 7) % 100).astype(np.int64), local_rank)
            mv = nn.SupervisedModel(mm, nn.SupervisedModel.NLL, S.STen.ones([100], dtv, local_rank))
            ov = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=(dt_name == "bf16"))([p_.value for p_ in mm.parameters])
            av = S.STen.zeros([1], dtv, local_rank)
            mv.train_step(ov, xv, tv, av, None)
            lib.lamp_device_synchronize()
            if use_graph:
                lib.lamp_graph_begin_capture()
                _, gv = mv.addTotalLossAndReturnGradientsAndNumExamples(xv, tv, av)
                gh = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(gh))
                def fn():
                    lib.lamp_graph_launch(gh)
                    ov.step(gv, 1.0)
            else:
                fn = lambda: mv.train_step(ov, xv, tv, av, None)
            for _ in range(a.warmup):
                fn()
            ws, total = [], 0.0
            while total < 0.3 and len(ws) < 50:
                lib.lamp_device_synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    fn()
                lib.lamp_device_synchronize()
                ws.append(time.perf_counter() - t0); total += ws[-1]
            ws.sort()
            return ws[len(ws) // 2] / a.steps * 1e3
        also = {"protocol": f"{a.warmup} warm-up + windows of {a.steps} steps between device synchronisations, median window; same step as the headline "
                            "(fwd + backprop + AdamW, HIP graph as the headline), run after it"}
        for key, dtn, Bv in (("f32_ms_per_step", "f32", a.batch), ("f64_ms_per_step", "f64", a.batch), ("b256_ms_per_step", a.dtype, 256),
                             ("b32_ms_per_step", a.dtype, 32)):
            if (dtn, Bv) == (a.dtype, a.batch):
                continue
            try:
                also[key] = variant_ms(dtn, Bv)
            except Exception as e:                       # informative keys: never lose the headline line because of them
                also[key] = f"failed: {e}"
    if rank == 0:
        value = units_per_step * a.gpus * a.steps / elapsed
        roof = roofline_of(rows)
        line = {"metric": metric, "value": value, "unit": unit, "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": a.dtype, "data": "synthetic", "config": config, "roofline": roof, "cpu_baseline": cpu_baseline,
                "rccl_ranks": rccl_ranks,
                "timed_windows": {"count": len(windows), "steps_each": a.steps, "reported": "median",
                                  "min_ms_per_step": min(windows) / a.steps * 1e3, "max_ms_per_step": max(windows) / a.steps * 1e3}}
        line.update(result_extra)
        if also is not None:
            line["also"] = also
        if alt_mode is not None:
            line["alt_mode"] = alt_mode
        if verify is not None:
            line.update(verify)
        if a.workload == "resnet":
            # whole-step figures against SURVEY.md 8(d): 153.3 MFLOP and ~1.4 MB algorithmic HBM bytes per sample per step
            per_gpu = value / a.gpus
            # bytes / flops per step as the launchers of this run declared them (sum over the kernel classes of the classification pass)
            decl_b = sum(r["bytes"] * r["launches"] for r in class_rows) / PROFILE_STEPS
            decl_f = sum(r["flops"] * r["launches"] for r in class_rows) / PROFILE_STEPS
            sps = per_gpu / units_per_step                                    # steps per second and GPU
            line["step_roofline"] = {"algorithmic_tflops": decl_f * sps / 1e12, "frac_bf16_mfma_peak": decl_f * sps / 1e12 / PEAK_BF16_TFLOPS,
                                     "algorithmic_GBps": decl_b * sps / 1e9, "frac_hbm_peak": decl_b * sps / 1e9 / PEAK_HBM_GBS,
                                     "frac_f32_mfma_peak": (decl_f * sps / 1e12 / PEAK_F32_TFLOPS) if a.dtype == "f32" else None,
                                     "frac_f64_mfma_peak": (decl_f * sps / 1e12 / PEAK_F64_TFLOPS) if a.dtype == "f64" else None,
                                     "declared_bytes_per_step": decl_b, "declared_flops_per_step": decl_f,
                                     "source": "sum of the launchers' declared algorithmic bytes / flops over every kernel class of one eager step",
                                     "survey_estimate": {"flops_per_sample": 153.3e6, "bytes_per_sample": 1.4e6,
                                                         "frac_bf16_mfma_peak": per_gpu * 153.3e6 / 1e12 / PEAK_BF16_TFLOPS,
                                                         "frac_hbm_peak": per_gpu * 1.4e6 / 1e9 / PEAK_HBM_GBS}}
        line["host_enqueue_ms_per_step"] = enqueue / a.steps * 1e3
        top = sorted(class_rows, key=lambda r: -r["total_ms"])[:int(os.environ.get("LAMP_BENCH_TOP", "10"))]
        line["kernel_classes"] = [{"tag": r["tag"], "launches_per_step": r["launches"] / PROFILE_STEPS, "ms_per_step": r["total_ms"] / PROFILE_STEPS}
                                  for r in top]
        print(json.dumps(line), file=line_out, flush=True)
    if comm is not None:
        lib.lamp_comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.close()


if __name__ == "__main__":
    main()
