"""Time lamp_convolution forward on one geometry (probe for kernel experiments)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array
lib.load()
from lamp_amd import sten as S
import numpy as np
N, Cin, Cout, k = 2048, 128, 128, 3
if len(sys.argv) > 1: Cin, Cout, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
x = S.STen.from_numpy(rng.standard_normal((N, Cin, 8, 8)).astype(np.float32), 0, S.BF16)
w = S.STen.from_numpy((rng.standard_normal((Cout, Cin, k, k)) * 0.05).astype(np.float32), 0, S.BF16)
b = S.STen.zeros([Cout], S.BF16, 0)
p = (k - 1) // 2
def run():
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), x, w, b, i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
    return S.STen(o)
for _ in range(5): run()
lib.lamp_device_synchronize()
R = 50
cal = C.c_double(0.0); lib.lamp_kernel_timer_calibrate(C.byref(cal))
# keep the GPU busy for ~10 ms first so that the host runs ahead and the event brackets see a full queue
big = S.STen.from_numpy(np.zeros((8192, 8192), np.float32), 0, S.BF16)
for _ in range(6):
    o = C.c_void_p(); lib.lamp_mm(C.byref(o), big, big); S.STen(o)
lib.lamp_kernel_timer_filter(b"conv_igemm_fprop_dgrad"); lib.lamp_kernel_timer_enable(1)
for _ in range(R): run()
lib.lamp_device_synchronize()
lib.lamp_kernel_timer_enable(0)
buf = C.create_string_buffer(1 << 16); lib.lamp_kernel_timer_report(buf, len(buf))
tag, n, ms, _, _ = buf.value.decode().split()
dt = (float(ms) / int(n)) * 1e-3 - cal.value * 1e-6
fl = 2.0 * N * 64 * Cin * Cout * k * k
print(f"variant={os.environ.get('LAMP_IG_VARIANT','default')} conv {Cin}->{Cout} k{k}: {dt*1e6:.1f} us/kernel (events - bracket), {fl/dt/1e12:.0f} TF/s")
