"""Helper of tests/test_ops_gpu.py::test_eight_wave_weight_gradient_dma_and_register_forms_agree_bit_for_bit (not a test module): prints one
sha256 per geometry of the eight-wave weight-gradient kernel's results (dW, and the pair's second dW), for the form LAMP_WG8H_DMA selects in this
process.  The two forms differ only in how the dY tiles reach LDS (LDS-DMA behind hand-counted vmcnt waits, or registers + ds_write)."""
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array  # noqa: E402
lib.load()
import torch  # noqa: E402
from lamp_amd import sten as S  # noqa: E402
from tests.util import closed_form, to_sten, to_torch  # noqa: E402

# (N, Cin, Cout): image ranges of 1, 2, 3 ... images (LAMP_WGRAD_MIN_IPS from the caller), odd batches, Cout < 128, Cin off the 32-grid
CASES = [(1, 128, 128), (2, 128, 100), (3, 100, 100), (5, 128, 70), (7, 50, 70), (9, 34, 66), (64, 128, 128), (131, 100, 100), (259, 128, 100)]


def main():
    dt = torch.bfloat16
    one, p1, p0, z = i64_array([1, 1]), i64_array([1, 1]), i64_array([0, 0]), i64_array([0, 0])
    for N, ci, co in CASES:
        x = closed_form((N, ci, 8, 8), 3, 2.0, dt)
        ga, gb = closed_form((N, co, 8, 8), 23, 1.0, dt), closed_form((N, co, 8, 8), 31, 1.0, dt)
        wa, wb = closed_form((co, ci, 3, 3), 17, 0.5, dt), closed_form((co, ci, 1, 1), 19, 0.7, dt)
        X, GA, GB, WA, WB = to_sten(x), to_sten(ga), to_sten(gb), to_sten(wa), to_sten(wb)
        out3 = (C.c_void_p * 3)()
        mask = (C.c_uint8 * 3)(0, 1, 0)
        lib.lamp_convolution_backward(out3, GA, X, WA, one, p1, one, 2, 0, z, 1, mask)
        dw = to_torch(S.STen(out3[1]))
        o2 = (C.c_void_p * 2)()
        lib.lamp_convolution_backward_weight_pair(o2, X, GA, WA, one, p1, one, GB, WB, one, p0, one, 2, 1)
        pa, pb = to_torch(S.STen(o2[0])), to_torch(S.STen(o2[1]))
        h = hashlib.sha256()
        for t in (dw, pa, pb):
            h.update(t.contiguous().view(torch.int16).numpy().tobytes())
        print(f"{N} {ci} {co} {h.hexdigest()}", flush=True)
    lib.lamp_kernel_timer_enable(1)
    x = closed_form((64, 128, 8, 8), 3, 2.0, dt)
    g = closed_form((64, 128, 8, 8), 23, 1.0, dt)
    out3 = (C.c_void_p * 3)()
    lib.lamp_convolution_backward(out3, to_sten(g), to_sten(x), to_sten(closed_form((128, 128, 3, 3), 17, 0.5, dt)), one, p1, one, 2, 0, z, 1, (C.c_uint8 * 3)(0, 1, 0))
    to_torch(S.STen(out3[1]))
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    print("classes", " ".join(sorted({l.split()[0] for l in buf.value.decode().splitlines() if l.startswith("conv_")})))


if __name__ == "__main__":
    main()
