"""Where are the missing points of BASELINE config 2?  (MI355X_MICROARCH.md "DVFS give-back" item 6.)

Needs the diagnostic build (stamps compiled in; the production library carries none):
    make -C lamp_amd/csrc -j8 EXTRA="-DGEMM_STAMP -DIG8D_STAMP" BUILD=build_stamp LIBDIR=../lib_stamp
    LAMP_LIB_PATH=lamp_amd/lib_stamp/liblamp_hip.so python scripts/gemm_clock_probe.py

For each of the three products of the 4096^2 Linear step (x.W + b, x^T.p, p.W^T; random operands), after >= 2 s of back-to-back launches,
thread 0 of every workgroup has stamped s_memtime (shader clock) and s_memrealtime (100 MHz) at kernel start / main loop start / main
loop end / kernel end:
    in-kernel clock          = d(s_memtime) / d(s_memrealtime) x 100 MHz over the main loop (median over workgroups)
    MFMA-issue utilisation   = matrix cycles the loop needs (2 waves per SIMD x 32 MFMAs x 16 cycles per k-step of 32) / loop cycles
    clock-adjusted fraction  = achieved FLOP/s / (peak x clock / 2.4 GHz)
Then the same stamps on ig_conv8d (128 -> 128 3x3, B = 2048) for the convolution's phases in microseconds AND cycles."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array
lib.load()
from lamp_amd import sten as S

PEAK = 2.5e15
n = 4096
rng = np.random.default_rng(0)
A_ = S.STen.from_numpy(rng.standard_normal((n, n), dtype=np.float32), 0, S.BF16)
W_ = S.STen.from_numpy(rng.standard_normal((n, n), dtype=np.float32), 0, S.BF16)
P_ = S.STen.from_numpy(rng.standard_normal((n, n), dtype=np.float32), 0, S.BF16)
bias = S.STen.from_numpy(rng.standard_normal((1, n), dtype=np.float32), 0, S.BF16)
dW, dX = S.STen.zeros([n, n], S.BF16, 0), S.STen.zeros([n, n], S.BF16, 0)


def fwd():
    o = C.c_void_p(); lib.lamp_linear_bias(C.byref(o), A_, W_, bias); S.STen(o).release()


products = [("x.W + b   (linear_bias, forward)", fwd),
            ("x^T.p     (addmm_out_transposed1, dW)", lambda: S.STen.addmm_out_transposed1(dW, dW, A_, P_, 1.0, 1.0)),
            ("p.W^T     (addmm_out_transposed2, dX)", lambda: S.STen.addmm_out_transposed2(dX, dX, P_, W_, 1.0, 1.0))]
stamp_fn = getattr(lib._dll if lib._dll else lib.load(), "lamp_debug_gemm_stamps", None)
if stamp_fn is None:
    raise SystemExit("this library has no stamps: build with EXTRA=-DGEMM_STAMP and point LAMP_LIB_PATH at it")
flop = 2.0 * n ** 3
print(f"# gemm_bf16_pp2_kernel, {n}^3 bf16, random operands; 256 x 256 x 64 tiles, 256 workgroups = one per CU")
for name, fn in products:
    for _ in range(20):
        fn()
    lib.lamp_device_synchronize()
    t0 = time.perf_counter(); cnt = 0
    while time.perf_counter() - t0 < 2.5:                      # >= 2 s of back-to-back launches: the clock has settled under load
        for _ in range(50):
            fn()
        cnt += 50
    lib.lamp_device_synchronize()
    wall = (time.perf_counter() - t0) / cnt
    buf = (C.c_uint64 * (512 * 8))()
    stamp_fn(buf)
    a = np.array(buf[:], dtype=np.uint64).reshape(512, 4, 2)[:256].astype(np.int64)     # [workgroup][stamp][shader clock, 100 MHz clock]
    cyc, rt = a[:, :, 0], a[:, :, 1]
    loop_cyc, loop_rt = cyc[:, 2] - cyc[:, 1], rt[:, 2] - rt[:, 1]
    clock = np.median(loop_cyc / np.maximum(loop_rt, 1)) * 100e6
    whole_rt = (rt[:, 3].max() - rt[:, 0].min()) / 100e6       # first workgroup start -> last workgroup end, seconds
    ksteps = n // 32
    need = ksteps * 2 * 32 * 16                                # per SIMD: two waves x 32 MFMAs (16 cycles each) per 32-deep k-step
    util = need / np.median(loop_cyc)
    tf = flop / wall
    print(f"{name}")
    print(f"  wall per launch (host clock, back to back) {wall * 1e6:8.1f} us = {tf / 1e12:7.1f} TFLOP/s = {tf / PEAK * 100:5.1f} % of 2.5 PF")
    print(f"  in-kernel span (first start -> last end)    {whole_rt * 1e6:8.1f} us   workgroup start spread {(rt[:, 0].max() - rt[:, 0].min()) / 100.0:6.2f} us")
    print(f"  phases, median over 256 workgroups (cycles | us by the 100 MHz clock): prologue {np.median(cyc[:, 1] - cyc[:, 0]):7.0f} | {np.median(rt[:, 1] - rt[:, 0]) / 100:6.2f}"
          f"   main loop {np.median(loop_cyc):8.0f} | {np.median(loop_rt) / 100:6.2f}   epilogue {np.median(cyc[:, 3] - cyc[:, 2]):7.0f} | {np.median(rt[:, 3] - rt[:, 2]) / 100:6.2f}")
    print(f"  in-kernel clock over the main loop          {clock / 1e9:8.3f} GHz (min {np.min(loop_cyc / np.maximum(loop_rt, 1)) / 10:5.3f}, max {np.max(loop_cyc / np.maximum(loop_rt, 1)) / 10:5.3f})")
    print(f"  MFMA-issue utilisation of the main loop     {util * 100:8.1f} %  ({need} matrix cycles per SIMD needed, {np.median(loop_cyc):.0f} spent)")
    print(f"  fraction of the peak AT THIS CLOCK          {tf / (PEAK * clock / 2.4e9) * 100:8.1f} %  (spec fraction {tf / PEAK * 100:5.1f} %)")

rt_fn = getattr(lib._dll, "lamp_debug_ig8d_rt_stamps", None)
if rt_fn is not None:
    N, Cin, Cout, k = 2048, 128, 128, 3
    x = S.STen.from_numpy(rng.standard_normal((N, Cin, 8, 8), dtype=np.float32), 0, S.BF16)
    w = S.STen.from_numpy(rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) * 0.05, 0, S.BF16)
    b = S.STen.from_numpy(np.zeros(Cout, dtype=np.float32), 0, S.BF16)

    def conv():
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), x, w, b, i64_array([1, 1]), i64_array([1, 1]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
        S.STen(o).release()
    for _ in range(20):
        conv()
    lib.lamp_device_synchronize()
    t0 = time.perf_counter(); cnt = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(50):
            conv()
        cnt += 50
    lib.lamp_device_synchronize()
    wall = (time.perf_counter() - t0) / cnt
    b1, b2 = (C.c_uint64 * (8 * 512))(), (C.c_uint64 * (8 * 512))()
    lib._dll.lamp_debug_ig8d_stamps(b1); rt_fn(b2)
    cyc = np.array(b1[:], dtype=np.uint64).reshape(512, 8)[:256].astype(np.int64)
    rt = np.array(b2[:], dtype=np.uint64).reshape(512, 8)[:256].astype(np.int64)
    names = ["weight DMA issued", "images in LDS (prologue)", "main loop", "epilogue arithmetic -> LDS", "stores issued", "stores retired"]
    print(f"# ig_conv8d_kernel<3, 8>, 128 -> 128 3x3, B = {N}: wall per launch {wall * 1e6:.1f} us (38.65 GFLOP -> {38.65e9 / wall / 1e12:.0f} TFLOP/s)")
    for i in range(1, 7):
        dc, dr = cyc[:, i] - cyc[:, i - 1], rt[:, i] - rt[:, i - 1]
        print(f"  {names[i - 1]:28s} median {np.median(dc):8.0f} cycles = {np.median(dr) / 100:6.2f} us")
    ml_c, ml_r = cyc[:, 3] - cyc[:, 2], rt[:, 3] - rt[:, 2]
    print(f"  in-kernel clock over the main loop {np.median(ml_c / np.maximum(ml_r, 1)) / 10:.3f} GHz; matrix cycles needed per SIMD 36864 -> utilisation {36864 / np.median(ml_c) * 100:.1f} %")
    print(f"  workgroup start spread {(rt[:, 0].max() - rt[:, 0].min()) / 100:.2f} us; first start -> last end {(rt[:, 6].max() - rt[:, 0].min()) / 100:.2f} us")
