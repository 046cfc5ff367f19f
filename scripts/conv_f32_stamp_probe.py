"""in-kernel phase timing of ig32_conv8_kernel (diagnostic build with -DIG32_STAMP only): prologue / main loop / epilogue cycles per
workgroup and the clock the chip holds (shader-clock ticks per 100 MHz tick)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
N, Cin, Cout, k = int(os.environ.get("N", "2048")), int(os.environ.get("CIN", "128")), int(os.environ.get("COUT", "128")), int(os.environ.get("KS", "3"))
rng = np.random.default_rng(0)
x = S.STen.from_numpy(rng.standard_normal((N, Cin, 8, 8), dtype=np.float32), 0, S.F32)
w = S.STen.from_numpy(rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) * 0.05, 0, S.F32)
b = S.STen.from_numpy(np.zeros(Cout, dtype=np.float32), 0, S.F32)
p = (k - 1) // 2
for it in range(30):
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), x, w, b, i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
    S.STen(o).release()
lib.lamp_device_synchronize()
buf = (C.c_uint64 * (8 * 1024))(); rt = (C.c_uint64 * (8 * 1024))()
lib._dll.lamp_debug_ig32_stamps(buf, rt)
nb = (N + 3) // 4
a = np.array(buf[:], dtype=np.uint64).reshape(1024, 8)[:nb].astype(np.int64)
r = np.array(rt[:], dtype=np.uint64).reshape(1024, 8)[:nb].astype(np.int64)
names = ["start", "prologue (images + stage 0 fragments)", "main loop", "epilogue issued", "stores landed"]
for i in range(1, 5):
    d = a[:, i] - a[:, i - 1]
    print(f"{names[i]:40s} median {np.median(d):9.0f}  min {d.min():9.0f}  max {d.max():9.0f} cycles")
tot = a[:, 4] - a[:, 0]
clk = (a[:, 2] - a[:, 1]) / np.maximum(r[:, 2] - r[:, 1], 1) * 100.0
print(f"workgroup total median {np.median(tot):.0f} cycles; clock in the main loop median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})")
t0 = r[:, 0].min()
print(f"first start -> last end {(r[:, 4].max() - t0) / 100.0:.1f} us; start of the last workgroup {(r[:, 0].max() - t0) / 100.0:.1f} us; "
      f"workgroups starting after 10 us: {(r[:, 0] - t0 > 1000).sum()} of {nb}")
seg = (C.c_uint64 * 16)()
if hasattr(lib._dll, "lamp_debug_ig32_segments") and lib._dll.lamp_debug_ig32_segments(seg) == 0:
    names = ["loads issued", "k-steps 0,1", "barrier 1", "wait + edge selects", "k-steps 2,3", "barrier 2"]
    T = ((Cin + 15) // 16) * k * k
    for w in range(2):
        v = [seg[w * 8 + i] / T for i in range(6)]
        print(f"wave {4 * w} (half {w}), cycles per stage: " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, v)) + f"; sum {sum(v):.0f}")
