"""time bf16 [M, 768] x [768, 768] products around the one-workgroup-per-CU boundary of the 256 x 256 kernel (256 tiles = one round)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import sten as S
from lamp_amd._capi import lib
rng = np.random.default_rng(0)
for (M, N, K) in ((16384, 768, 768), (21504, 768, 768), (21760, 768, 768), (22016, 768, 768), (24576, 768, 768), (32768, 768, 768), (43520, 768, 768), (24576, 768, 3072), (21760, 768, 3072)):
    a = S.STen.from_numpy(rng.standard_normal((M, K), dtype=np.float32), 0, S.BF16)
    b = S.STen.from_numpy(rng.standard_normal((K, N), dtype=np.float32), 0, S.BF16)
    a.mm(b); lib.lamp_device_synchronize()
    t = time.perf_counter()
    for _ in range(50): a.mm(b)
    lib.lamp_device_synchronize()
    dt = (time.perf_counter() - t) / 50
    tiles = (M // 256) * (N // 256)
    print(f"{M:6d} x {N} x {K}: {tiles:4d} tiles ({tiles / 256:.2f} rounds)  {dt * 1e6:7.1f} us  {2 * M * N * K / dt / 1e12:7.1f} TFLOP/s")
