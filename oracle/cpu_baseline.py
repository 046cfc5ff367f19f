"""CPU baseline for bench.py: the oracle's training step (the same ATen operator sequence lamp
dispatches, torch CPU kernels) timed on the host cores.  TEST/BENCH INFRASTRUCTURE ONLY.

Prints one JSON object: {"value": samples_per_sec, "unit": ..., "cores": T, "kind": "port", "sample": "..."}.
It is a reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="resnet")
    ap.add_argument("--batch", type=int, default=256)   # the same B or B / 8 of the GPU run's 2048
    ap.add_argument("--budget-s", type=float, default=15.0)
    ap.add_argument("--threads", type=int, default=32)
    a = ap.parse_args()
    import torch
    from oracle import lamp_oracle as O
    # intra-op threads: ATen's OpenMP loops over these small layers slow down badly past a few dozen threads
    # (measured: 45 s per batch-128 step with 256 threads on a 2 x 64-core host), so cap them
    T = min(os.cpu_count() or 1, a.threads)
    torch.set_num_threads(T)
    dt = torch.float32
    if a.workload == "resnet":
        m = O.resnet(100, dt)
        B = a.batch
        x = O.closed_form(B * 3 * 32 * 32, 5, 1.0, dt).reshape(B, 3, 32, 32)
        target = (torch.arange(B) * 7) % 100
        cw = torch.ones(100, dtype=dt)
        opt = O.AdamW([p.value for p in m.parameters()], 0.0, 1e-3, 0.9, 0.95)
        step = lambda: O.training_step(m, O.nll_loss(100, cw), x, target, opt)
        unit, per_step = "samples/s", B
        sample = (f"Cnn.resnet(100) fwd+bwd+AdamW, fp32, batch {B} (= B / {2048 // B} of the GPU run's 2048 per GPU, SURVEY 8d), ATen CPU kernels via torch.ops.aten, "
                  f"{T} intra-op threads on a host of {os.cpu_count()} logical cores")
    elif a.workload == "mlp":
        m = O.Sequential(O.mlp(784, 10, [256], dt), O.Fun(lambda v: v.logSoftMax(1)))
        B = 1024
        x = O.closed_form(B * 784, 0, 1.0, dt).reshape(B, 784)
        target = torch.arange(B) % 10
        cw = torch.ones(10, dtype=dt)
        step = lambda: O.training_step(m, O.nll_loss(10, cw), x, target, None)
        unit, per_step = "samples/s", B
        sample = "MLP 784-256-10 fwd+bwd, fp32, batch 1024 (BASELINE config 1, the full config)"
    elif a.workload in ("knn", "umap", "umap-e2e"):
        # BASELINE config 5 on the host, on a BOUNDED sample (256 TFLOP of distances take hours on a CPU): brute-force kNN of a
        # 50k x 128 f32 subsample (lamp.knn.knnSearch's op list: mm + norms + topk per 1000-query minibatch), UMAP layout iterations on a
        # 100k-point graph (10 neighbours, 5 negatives per edge, f64).  Both are scaled to the GPU run's unit in `sample`.
        import numpy as np
        if a.workload == "knn":
            n, d, k = 50_000, 128, 10
            rng = np.random.default_rng(0)
            pts = torch.from_numpy(rng.random((n, d), dtype=np.float32) + (np.arange(n) % 16)[:, None].astype(np.float32))
            nq = 5000
            step = lambda: O.knn_minibatched(pts, pts[:nq], k, 1000)
            unit, per_step = "queries/s", nq * (n / 1_000_000.0)            # a query against 1M points costs 20 x a query against 50k
            sample = (f"lamp.knn.knnSearch op list (mm + norms + topk per 1000-query minibatch), {nq} queries x {n} x {d} f32, k = {k}, ATen CPU; "
                      f"value is scaled to queries/s against 1M points (x {n / 1e6:.2f}: distance work is linear in the data set)")
        else:
            n, kk = 100_000, 10
            rng = np.random.default_rng(0)
            knn_idx = (np.arange(n)[:, None] + 1 + rng.integers(0, n - 1, (n, kk))) % n
            i1 = torch.from_numpy(np.repeat(np.arange(n), kk - 1)); i2 = torch.from_numpy(knn_idx[:, 1:].reshape(-1).copy())
            bw = torch.from_numpy(rng.random(n * (kk - 1)))
            rows = torch.stack([i1.double(), i2.double(), bw], 1)
            state = {"it": 0}

            def step():
                O.umap_optimize(rows, n, 0.1, 1, 0.0, 5, 42 + state["it"]); state["it"] += 1
            if a.workload == "umap":
                unit, per_step = "iterations/s", n / 1_000_000.0             # an iteration on 1M points costs 10 x one on 100k
                sample = (f"Umap.optimize op list, 1 iteration per step on {n} points ({n * (kk - 1)} edges, 5 negatives each, f64, AdamW), ATen CPU; "
                          f"value is scaled to iterations/s at 1M points (x {n / 1e6:.1f}: work is linear in the edges)")
            else:
                # end to end at 1M points = kNN (2 n^2 d flop) + 500 iterations: the layout sample above, plus the kNN sample's rate
                rng2 = np.random.default_rng(1)
                pts = torch.from_numpy(rng2.random((50_000, 128), dtype=np.float32))
                t0 = time.perf_counter(); O.knn_minibatched(pts, pts[:2000], 10, 1000); t_knn = time.perf_counter() - t0
                knn_1m_s = t_knn / 2000 * (1_000_000 / 50_000) * 1_000_000    # seconds for 1M queries against 1M points
                state["knn_1m_s"] = knn_1m_s
                unit, per_step = "points/s", 0.0                            # filled below from the two rates
                sample = (f"extrapolated from the two halves on the host: kNN of 2000 x 50k x 128 f32 queries scaled to 1M x 1M ({knn_1m_s:.0f} s) + 500 x the "
                          f"layout iteration measured on {n} points scaled x 10; ATen CPU")
    else:  # gemm
        n = 2048
        A_ = O.closed_form(n * n, 1, 2.0, torch.bfloat16).reshape(n, n)
        B_ = O.closed_form(n * n, 7, 2.0, torch.bfloat16).reshape(n, n)
        step = lambda: torch.mm(A_, B_)
        unit, per_step = "TFLOP/s", 2.0 * n ** 3 / 1e12
        sample = "bf16 mm 2048^3 (GPU run is 4096^3), ATen CPU"
    step()  # warm-up
    t0 = time.perf_counter()
    k = 0
    while True:
        step()
        k += 1
        if time.perf_counter() - t0 >= a.budget_s:
            break
    dtm = time.perf_counter() - t0
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    value = per_step * k / dtm
    if a.workload == "umap-e2e":
        it_1m_s = (dtm / k) * 10.0                                          # one layout iteration at 1M points
        total_s = state["knn_1m_s"] + 500 * it_1m_s
        value = 1_000_000 / total_s
        sample += f" -> {total_s:.0f} s end to end"
    print(json.dumps({"value": value, "unit": unit, "cores": T, "host_cores": os.cpu_count(), "kind": "port", "sample": sample + f"; {k} steps in {dtm:.1f} s on {cpu}"}))


if __name__ == "__main__":
    main()
