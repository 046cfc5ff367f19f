"""First GPU trip: runtime + elementwise + reduce + GEMM against numpy, plus a GEMM timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from lamp_amd.sten import *
from lamp_amd._capi import lib

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

buf = C.create_string_buffer(256); lib.lamp_device_name(buf, 256); print("device:", buf.value.decode())
rng = np.random.default_rng(0)
fails = 0
def check(name, got, want, tol):
    global fails
    e = rel(got, want)
    ok = e <= tol
    fails += (not ok)
    print(f"{'ok ' if ok else 'FAIL'} {name}: rel err {e:.3e} (tol {tol})")

for dt, npdt, tol in [(F32, np.float32, 1e-6), (F64, np.float64, 1e-14), (BF16, np.float32, 1e-2)]:
    a = rng.standard_normal((37, 129)).astype(npdt); b = rng.standard_normal((37, 129)).astype(npdt)
    if dt == BF16:
        a = bf16_bits_to_f32(f32_to_bf16_bits(a)).reshape(a.shape); b = bf16_bits_to_f32(f32_to_bf16_bits(b)).reshape(b.shape)
    A = STen.from_numpy(a, dtype=dt); B = STen.from_numpy(b, dtype=dt)
    check(f"add {dt}", (A + B).to_numpy(), a + b, tol)
    check(f"mul {dt}", (A * B).to_numpy(), a * b, tol)
    check(f"add bcast row {dt}", (A + STen.from_numpy(b[:1], dtype=dt)).to_numpy(), a + b[:1], tol)
    check(f"add bcast col {dt}", (A + STen.from_numpy(b[:, :1], dtype=dt)).to_numpy(), a + b[:, :1], tol)
    check(f"transpose add {dt}", (A.t + B.t).to_numpy(), (a + b).T, tol)
    check(f"relu {dt}", A.relu().to_numpy(), np.maximum(a, 0), tol)
    check(f"exp {dt}", A.exp().to_numpy(), np.exp(a), tol * 4)
    check(f"sum all {dt}", A.sum().to_numpy(), a.astype(np.float64).sum(), tol * 50)
    check(f"sum dim0 {dt}", A.sum(0, True).to_numpy(), a.astype(np.float64).sum(0, keepdims=True), tol * 10)
    check(f"sum dim1 {dt}", A.sum(1, False).to_numpy(), a.astype(np.float64).sum(1), tol * 10)
    check(f"mean dim1 {dt}", A.mean(1, True).to_numpy(), a.astype(np.float64).mean(1, keepdims=True), tol * 10)
    x4 = rng.standard_normal((6, 5, 7, 3)).astype(npdt)
    if dt == BF16: x4 = bf16_bits_to_f32(f32_to_bf16_bits(x4)).reshape(x4.shape)
    X4 = STen.from_numpy(x4, dtype=dt)
    check(f"sum dims(0,2,3) {dt}", X4.sum([0, 2, 3], False).to_numpy(), x4.astype(np.float64).sum((0, 2, 3)), tol * 10)
    check(f"sum dims(1,3) {dt}", X4.sum([1, 3], True).to_numpy(), x4.astype(np.float64).sum((1, 3), keepdims=True), tol * 10)
    check(f"sum dims(0,2) generic {dt}", X4.sum([0, 2], False).to_numpy(), x4.astype(np.float64).sum((0, 2)), tol * 10)
    big = rng.standard_normal((1 << 20,)).astype(npdt)
    if dt == BF16: big = bf16_bits_to_f32(f32_to_bf16_bits(big))
    check(f"sum big {dt}", STen.from_numpy(big, dtype=dt).sum().to_numpy(), big.astype(np.float64).sum(), 1e-2 if dt == BF16 else tol * 1000)
    # GEMM
    for (M, N, K) in [(64, 64, 64), (128, 128, 64), (100, 37, 53), (256, 384, 192), (1024, 256, 784), (1024, 10, 256)]:
        a = rng.standard_normal((M, K)).astype(npdt) ; b = rng.standard_normal((K, N)).astype(npdt)
        if dt == BF16:
            a = bf16_bits_to_f32(f32_to_bf16_bits(a)).reshape(a.shape); b = bf16_bits_to_f32(f32_to_bf16_bits(b)).reshape(b.shape)
        A = STen.from_numpy(a, dtype=dt); B = STen.from_numpy(b, dtype=dt)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        gt = {F32: 2e-6, F64: 1e-14, BF16: 8e-3}[dt]
        check(f"mm {dt} {M}x{N}x{K}", A.mm(B).to_numpy(), ref, gt)
        # transposed1: out = beta*out + a^T b ; a is [K', M'] ...
        out0 = rng.standard_normal((K, N)).astype(npdt)
        p = rng.standard_normal((M, N)).astype(npdt)
        if dt == BF16:
            out0 = bf16_bits_to_f32(f32_to_bf16_bits(out0)).reshape(out0.shape); p = bf16_bits_to_f32(f32_to_bf16_bits(p)).reshape(p.shape)
        O = STen.from_numpy(out0, dtype=dt); P = STen.from_numpy(p, dtype=dt)
        STen.addmm_out_transposed1(O, O, A, P, 1.0, 1.0)     # dB += A^T p
        check(f"addmm_t1 {dt} {M}x{N}x{K}", O.to_numpy(), out0.astype(np.float64) + a.astype(np.float64).T @ p.astype(np.float64), gt * 2)
        out1 = rng.standard_normal((M, K)).astype(npdt)
        if dt == BF16: out1 = bf16_bits_to_f32(f32_to_bf16_bits(out1)).reshape(out1.shape)
        O1 = STen.from_numpy(out1, dtype=dt)
        STen.addmm_out_transposed2(O1, O1, P, B, 1.0, 1.0)   # dA += p B^T
        check(f"addmm_t2 {dt} {M}x{N}x{K}", O1.to_numpy(), out1.astype(np.float64) + p.astype(np.float64) @ b.astype(np.float64).T, gt * 2)
    # bmm
    a = rng.standard_normal((3, 33, 65)).astype(npdt); b = rng.standard_normal((3, 65, 17)).astype(npdt)
    if dt == BF16:
        a = bf16_bits_to_f32(f32_to_bf16_bits(a)).reshape(a.shape); b = bf16_bits_to_f32(f32_to_bf16_bits(b)).reshape(b.shape)
    check(f"bmm {dt}", STen.from_numpy(a, dtype=dt).bmm(STen.from_numpy(b, dtype=dt)).to_numpy(), a.astype(np.float64) @ b.astype(np.float64), {F32: 2e-6, F64: 1e-14, BF16: 8e-3}[dt])

# timing: 4096^3 bf16
for n in (4096, 8192):
    a = (rng.random((n, n), dtype=np.float32) * 2 - 1); b = (rng.random((n, n), dtype=np.float32) * 2 - 1)
    A = STen.from_numpy(a, dtype=BF16); B = STen.from_numpy(b, dtype=BF16)
    out = STen.zeros([n, n], BF16)
    for _ in range(3): STen.mmOut(out, A, B)
    synchronize()
    t0 = time.perf_counter(); iters = 20
    for _ in range(iters): STen.mmOut(out, A, B)
    synchronize(); dt_ = (time.perf_counter() - t0) / iters
    print(f"bf16 mm {n}^3: {dt_*1e3:.3f} ms  {2*n**3/dt_/1e12:.1f} TFLOP/s")
    if n == 4096:
        Bt = STen.from_numpy(b.T.copy(), dtype=BF16)
        for nm, fn in [("NT", lambda: STen.addmm_out_transposed2(out, out, A, Bt, 0.0, 1.0)), ("TN", lambda: STen.addmm_out_transposed1(out, out, A, B, 0.0, 1.0))]:
            for _ in range(3): fn()
            synchronize(); t0 = time.perf_counter()
            for _ in range(iters): fn()
            synchronize(); dt_ = (time.perf_counter() - t0) / iters
            print(f"bf16 {nm} {n}^3: {dt_*1e3:.3f} ms  {2*n**3/dt_/1e12:.1f} TFLOP/s")
        ref = bf16_bits_to_f32(f32_to_bf16_bits(a)).reshape(n, n)[:64].astype(np.float64) @ bf16_bits_to_f32(f32_to_bf16_bits(b)).reshape(n, n).astype(np.float64)
        STen.mmOut(out, A, B)
        check("mm 4096 rows[:64]", out.to_numpy()[:64], ref, 8e-3)
a32 = rng.standard_normal((4096, 4096)).astype(np.float32)
A = STen.from_numpy(a32); out = STen.zeros([4096, 4096], F32)
for _ in range(2): STen.mmOut(out, A, A)
synchronize(); t0 = time.perf_counter()
for _ in range(5): STen.mmOut(out, A, A)
synchronize(); dt_ = (time.perf_counter() - t0) / 5
print(f"f32 mm 4096^3: {dt_*1e3:.3f} ms  {2*4096**3/dt_/1e12:.1f} TFLOP/s")
print("live tensors:", live_tensor_count())
print("FAILS:", fails)
sys.exit(1 if fails else 0)
