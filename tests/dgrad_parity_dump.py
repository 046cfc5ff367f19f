"""Helper of tests/test_ops_gpu.py::test_stride_two_input_gradient_pairs_by_row_parity (not a test module): the input gradients of the two stride-2 pairs of
Cnn.resnet (res1: 6 + 6 gradient channels, res2: 16 + 16) through lamp_convolution_backward_input_pair, in the form LAMP_NCV_DGRAD_PARITY selects, to an npz."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array  # noqa: E402
lib.load()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lamp_amd import sten as S  # noqa: E402
from tests.util import closed_form, to_sten, to_torch  # noqa: E402

out = {}
dt = torch.bfloat16
for name, N, Cin, H, Ca, Cb, with_add in (("res1", 70, 6, 32, 6, 6, False), ("res2", 67, 6, 16, 16, 16, True), ("res2big", 2050, 6, 16, 16, 16, False)):
    x = closed_form((N, Cin, H, H), 3, 2.0, dt)
    wa, wb = closed_form((Ca, Cin, 3, 3), 17, 0.5, dt), closed_form((Cb, Cin, 1, 1), 19, 0.7, dt)
    ho = (H - 1) // 2 + 1
    ga, gb = closed_form((N, Ca, ho, ho), 23, 1.0, dt), closed_form((N, Cb, ho, ho), 31, 1.0, dt)
    addend = closed_form((N, Cin, H, H), 29, 3.0, dt)
    sd, p1, p0, one = i64_array([2, 2]), i64_array([1, 1]), i64_array([0, 0]), i64_array([1, 1])
    o = C.c_void_p()
    lib.lamp_convolution_backward_input_pair(C.byref(o), to_sten(x), to_sten(ga), to_sten(wa), sd, p1, one, to_sten(gb), to_sten(wb), sd, p0, one, 2, 1,
                                             to_sten(addend) if with_add else None)
    out[name] = to_torch(S.STen(o)).float().numpy()
np.savez(sys.argv[1], **out)
