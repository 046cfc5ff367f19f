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
print(f"B={B} H={H} S={Sq} D={D} causal={causal} flash={os.environ.get('LAMP_FLASH_ATTENTION', '1')}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TFLOP/s")
