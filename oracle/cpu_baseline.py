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
    ap.add_argument("--batch", type=int, default=128)
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
        sample = f"Cnn.resnet(100) fwd+bwd+AdamW, fp32, batch {B} (GPU run uses batch 2048 per GPU), ATen CPU kernels via torch.ops.aten"
    elif a.workload == "mlp":
        m = O.Sequential(O.mlp(784, 10, [256], dt), O.Fun(lambda v: v.logSoftMax(1)))
        B = 1024
        x = O.closed_form(B * 784, 0, 1.0, dt).reshape(B, 784)
        target = torch.arange(B) % 10
        cw = torch.ones(10, dtype=dt)
        step = lambda: O.training_step(m, O.nll_loss(10, cw), x, target, None)
        unit, per_step = "samples/s", B
        sample = "MLP 784-256-10 fwd+bwd, fp32, batch 1024 (BASELINE config 1, the full config)"
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
    print(json.dumps({"value": per_step * k / dtm, "unit": unit, "cores": T, "kind": "port", "sample": sample + f"; {k} steps in {dtm:.1f} s on {cpu}"}))


if __name__ == "__main__":
    main()
