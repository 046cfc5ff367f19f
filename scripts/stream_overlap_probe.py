"""Do two HIP streams hide the ~4.5 us per-launch floor of small kernels?  N small element-wise kernels on one stream against the same
N split over two streams (issued alternately by one host thread), eager and as a captured graph with two parallel branches."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib
lib.load()
from lamp_amd import sten as S
import numpy as np
n_el = int(sys.argv[1]) if len(sys.argv) > 1 else 2 * 1024 * 1024
a = S.STen.from_numpy(np.ones(n_el, dtype=np.float32), 0, S.BF16)
b = S.STen.from_numpy(np.ones(n_el, dtype=np.float32), 0, S.BF16)
o1, o2 = S.STen.zeros([n_el], S.BF16, 0), S.STen.zeros([n_el], S.BF16, 0)
s1, s2 = C.c_void_p(), C.c_void_p()
lib.lamp_stream_get_from_pool(0, 0, C.byref(s1)); lib.lamp_stream_get_from_pool(0, 0, C.byref(s2))
N = 400
def one():
    lib.lamp_stream_set_current(s1)
    for _ in range(N): lib.lamp_add_out(o1, a, b, 1.0)
def two():
    for i in range(N // 2):
        lib.lamp_stream_set_current(s1); lib.lamp_add_out(o1, a, b, 1.0)
        lib.lamp_stream_set_current(s2); lib.lamp_add_out(o2, a, b, 1.0)
for name, f in (("one stream", one), ("two streams", two)):
    f(); lib.lamp_device_synchronize()
    t = time.perf_counter(); f(); lib.lamp_device_synchronize(); dt = time.perf_counter() - t
    print(f"{name:12s} {n_el * 2 / 1e6:.1f} MB tensors: {dt / N * 1e6:.2f} us per kernel")

# the same inside a captured graph: one chain of N kernels against two parallel branches of N/2 (fork / join through stream waits)
def capture(two_branches):
    lib.lamp_stream_set_current(s1)
    lib.lamp_graph_begin_capture()
    if two_branches:
        lib.lamp_stream_wait_stream(s2, s1)              # fork: s2 joins the capture
        for i in range(N // 2):
            lib.lamp_stream_set_current(s1); lib.lamp_add_out(o1, a, b, 1.0)
            lib.lamp_stream_set_current(s2); lib.lamp_add_out(o2, a, b, 1.0)
        lib.lamp_stream_wait_stream(s1, s2)              # join
        lib.lamp_stream_set_current(s1)
    else:
        for _ in range(N): lib.lamp_add_out(o1, a, b, 1.0)
    g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
    return g
for name, tb in (("graph, one chain", False), ("graph, two branches", True)):
    g = capture(tb)
    lib.lamp_graph_launch(g); lib.lamp_device_synchronize()
    t = time.perf_counter()
    for _ in range(5): lib.lamp_graph_launch(g)
    lib.lamp_device_synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"{name:20s} {n_el * 2 / 1e6:.1f} MB tensors: {dt / N * 1e6:.2f} us per kernel")
