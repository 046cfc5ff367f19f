"""Time the narrow forward kernel on the stem / res1 geometries with pieces of its statistics epilogue switched off (diagnostic)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array
lib.load()
from lamp_amd import sten as S
import numpy as np
N = 2048
geoms = [(3, 6, 5, 32, 1), (6, 6, 3, 16, 1), (6, 16, 3, 16, 2)]
rng = np.random.default_rng(0)
big = S.STen.from_numpy(np.zeros((8192, 8192), np.float32), 0, S.BF16)
for (cin, cout, k, H, sd) in geoms:
    x = S.STen.from_numpy(rng.standard_normal((N, cin, H, H)).astype(np.float32), 0, S.BF16)
    w = S.STen.from_numpy((rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32), 0, S.BF16)
    b = S.STen.zeros([cout], S.BF16, 0)
    p = (k - 1) // 2
    def run():
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), x, w, b, i64_array([sd, sd]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
        return S.STen(o)
    for _ in range(5): run()
    lib.lamp_device_synchronize()
    cal = C.c_double(0.0); lib.lamp_kernel_timer_calibrate(C.byref(cal))
    for _ in range(6):
        o = C.c_void_p(); lib.lamp_mm(C.byref(o), big, big); S.STen(o)
    lib.lamp_kernel_timer_filter(b"conv_fwd_narrow"); lib.lamp_kernel_timer_enable(1)
    for _ in range(50): run()
    lib.lamp_device_synchronize()
    lib.lamp_kernel_timer_enable(0)
    buf = C.create_string_buffer(1 << 16); lib.lamp_kernel_timer_report(buf, len(buf))
    tag, n, ms, _, _ = buf.value.decode().split()
    dt = (float(ms) / int(n)) * 1e-3 - cal.value * 1e-6
    print(f"stats={os.environ.get('LAMP_NCV_BN_STATS','1')} dbg={os.environ.get('LAMP_NCV_STATS_DBG','0')} conv {cin}->{cout} k{k} H{H} s{sd}: {dt*1e6:.1f} us/kernel")
