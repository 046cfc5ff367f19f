"""Attention forward timing: fused flash kernel vs the composed bmm + softmax + bmm path (LAMP_FLASH_ATTENTION=0).
usage: python scripts/attn_probe.py [B H S D causal]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import sten as S
from lamp_amd._capi import lib

B, H, Sq, D, causal = (int(a) for a in (sys.argv[1:6] + ["8", "16", "4096", "128", "0"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
q, k, v = (S.STen.from_numpy(rng.standard_normal((B, H, Sq, D), dtype=np.float32), 0, S.BF16) for _ in range(3))


def run(n):
    for _ in range(n):
        o, l = C.c_void_p(), C.c_void_p()
        lib.lamp_scaled_dot_product_attention(C.byref(o), C.byref(l), q, k, v, causal, 0.0)
        S.STen(o); S.STen(l)
    lib.lamp_device_synchronize()


run(3)
t = time.perf_counter(); run(20); dt = (time.perf_counter() - t) / 20
fl = 4.0 * B * H * Sq * Sq * D * (0.5 if causal else 1.0)
tag = f"B={B} H={H} S={Sq} D={D} causal={causal} flash={os.environ.get('LAMP_FLASH_ATTENTION', '1')}"
print(f"{tag} fwd: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TFLOP/s")

o, l = C.c_void_p(), C.c_void_p()
lib.lamp_scaled_dot_product_attention(C.byref(o), C.byref(l), q, k, v, causal, 0.0)
O, L = S.STen(o), S.STen(l)
go = S.STen.from_numpy(rng.standard_normal((B, H, Sq, D), dtype=np.float32), 0, S.BF16)


def runb(n):
    for _ in range(n):
        out3 = (C.c_void_p * 3)()
        lib.lamp_scaled_dot_product_attention_backward(out3, go, q, k, v, O, L, causal, 0.0)
        for h in out3:
            S.STen(h)
    lib.lamp_device_synchronize()


runb(2)
t = time.perf_counter(); runb(10); dt = (time.perf_counter() - t) / 10
print(f"{tag} bwd: {dt * 1e3:.3f} ms  {2.5 * fl / dt / 1e12:.1f} TFLOP/s (10 S^2 d model flops)")

lib.lamp_kernel_timer_enable(1)
run(3); runb(3)
buf = C.create_string_buffer(1 << 16)
lib.lamp_kernel_timer_report(buf, len(buf))
lib.lamp_kernel_timer_enable(0)
for line in buf.value.decode().splitlines():
    tag_, n, ms, flops, byts = line.split()
    if "sdpa" in tag_ or "gemm" in tag_:
        print(f"   {tag_:24s} n={n} avg {float(ms) / int(n) * 1e3:9.1f} us  {float(flops) * int(n) / max(float(ms), 1e-9) / 1e9:8.1f} TFLOP/s executed (flops column is per launch)")
