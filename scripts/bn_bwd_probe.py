"""Times the batch-norm backward of the ResNet step's shapes with the library's kernel timer (HIP events around each launch).
Run twice to compare: LAMP_BN_FUSED_BWD=0 (two kernels) and default (one pass).  Usage: python scripts/bn_bwd_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd import _capi, sten as S  # noqa: E402

lib = _capi.lib


def run(N, Cc, H, variant, reps=30):
    rng = np.random.default_rng(0)
    shape = (N, Cc, H, H)
    mk = lambda sh, sc=1.0: S.STen.from_numpy((rng.standard_normal(sh) * sc).astype(np.float32), 0).castToType(S.BF16)
    X, AD, GY = mk(shape, 2.0), mk(shape), mk(shape)
    Wt, Bt = mk((Cc,)), mk((Cc,))
    RM, RV = S.STen.zeros([Cc], S.BF16, 0), S.STen.ones([Cc], S.BF16, 0)
    fwd = (C.c_void_p * 3)()
    lib.lamp_native_batch_norm(fwd, X, Wt, Bt, RM, RV, 1, 0.1, 1e-5)
    y, sm, si = (S.STen(fwd[i]) for i in range(3))

    def backward():
        if variant == 2:
            out4 = (C.c_void_p * 4)()
            lib.lamp_native_batch_norm_add_relu_backward(out4, GY, X, AD, Wt, Bt, RM, RV, sm, si, 1, 1e-5, (C.c_uint8 * 4)(1, 1, 1, 1))
            return [S.STen(out4[i]) for i in range(4)]
        out = (C.c_void_p * 3)()
        mask = (C.c_uint8 * 3)(1, 1, 1)
        if variant == 1:
            lib.lamp_native_batch_norm_relu_backward(out, GY, X, Wt, Bt, RM, RV, sm, si, 1, 1e-5, mask)
        else:
            lib.lamp_native_batch_norm_backward(out, GY, X, Wt, RM, RV, sm, si, 1, 1e-5, mask)
        return [S.STen(out[i]) for i in range(3)]

    for _ in range(3):
        backward()
    lib.lamp_device_synchronize()
    lib.lamp_kernel_timer_enable(1)
    for _ in range(reps):
        backward()
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    tot = 0.0
    parts = []
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()[:3]
        us = float(ms) * 1e3 / int(cnt)
        parts.append(f"{name} {us:.1f}")
        tot += us
    print(f"N={N} C={Cc} H={H} variant={variant}: {tot:.1f} us  ({', '.join(parts)})", flush=True)


if __name__ == "__main__":
    for case in [(2048, 128, 8, 2), (2048, 128, 8, 1), (2048, 100, 8, 1), (2048, 64, 8, 1), (2048, 16, 16, 1), (2048, 6, 32, 1), (2048, 6, 32, 2)]:
        run(*case)
