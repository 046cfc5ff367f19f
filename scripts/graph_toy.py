"""a THREE-kernel HIP graph (add, mul, relu on a 1 MiB vector) captured through the library's capture entry points and
replayed 2000 times - the reproducer for "rocprofv3 --kernel-trace dies inside hipGraphLaunch".  Run it bare and under
`rocprofv3 --kernel-trace -- python3 scripts/graph_toy.py`: if the toy dies under the profiler too, the fault is not in the captured nodes of
the training step (graph-private pool, event-joined reduction, HIP_FORCE_DEV_KERNARG)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
n = 1 << 18
a = S.STen.from_numpy(np.linspace(-1, 1, n, dtype=np.float32), 0)
b = S.STen.from_numpy(np.full(n, 0.5, dtype=np.float32), 0)
st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(st)); lib.lamp_stream_set_current(st)
lib.lamp_device_synchronize()
lib.lamp_graph_begin_capture()
c = (a + b)
d = c * b
e = d.relu()
g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
replays = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for _ in range(replays):
    lib.lamp_graph_launch(g)
lib.lamp_device_synchronize()
ref = np.maximum((np.linspace(-1, 1, n, dtype=np.float32) + 0.5) * 0.5, 0)
assert np.array_equal(e.to_numpy(), ref.astype(np.float32))
print(f"graph_toy ok: {replays} replays of a 3-kernel graph")
