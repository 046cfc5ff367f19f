"""lamp-data over the host C ABI: tensor-list files, checkpoints, the CIFAR record reader and the minibatch stream.

Mirror of lamp.data.{Writer, Reader, BatchStream} and example-cifar100's Cifar.loadImageFile:
  Writer.writeTensorsIntoFile / writeCheckpoint        lamp-data/src/main/scala/lamp/data/Writer.scala:143-190
  Reader.readTensorsFromFile / loadFromFile            lamp-data/src/main/scala/lamp/data/Reader.scala:62-95
  BatchStream.minibatchesFromFull, everyNth            lamp-data/src/main/scala/lamp/data/BatchStream.scala:528-592, :378-400
  Cifar.loadImageFile                                  example-cifar100/src/main/scala/lamp/example/cifar/cifar100.scala:29-56
Every function is one call into liblamp_hip.so (lamp_amd/csrc/host/data.cpp).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from ._capi import lib, handle_array
from .sten import STen, F32


def writeTensorsIntoFile(tensors: Sequence[STen], file: str) -> None:
    lib.lamp_write_tensors_into_file(handle_array([t.h for t in tensors]), len(tensors), str(file).encode())


def readTensorsFromFile(file: str, device: int = -1, pin: bool = False) -> List[STen]:
    n = C.c_int64()
    lib.lamp_tensor_list_length(str(file).encode(), C.byref(n))
    out = (C.c_void_p * max(n.value, 1))()
    got = C.c_int64()
    lib.lamp_read_tensors_from_file(out, n.value, C.byref(got), str(file).encode(), device, int(pin))
    return [STen(out[i]) for i in range(got.value)]


def writeCheckpoint(file: str, model) -> None:
    lib.lamp_module_write_checkpoint(model.h, str(file).encode())


def loadFromFile(module, file: str) -> None:
    lib.lamp_module_load_from_file(module.h, str(file).encode())


def loadImageFile(file: str, numImages: int, dtype: int = F32, device: int = 0) -> Tuple[STen, STen]:
    """Cifar.loadImageFile: (fine labels i64 [n], images [n, 3, 32, 32])."""
    lab, img = C.c_void_p(), C.c_void_p()
    lib.lamp_cifar_load_image_file(C.byref(lab), C.byref(img), str(file).encode(), int(numImages), dtype, device)
    return STen(lab), STen(img)


class JavaRandom:
    """java.util.Random's documented 48-bit LCG (what scala.util.Random wraps), so that a Python driver can draw the shuffles a
    JVM driver would for the same seed.  Not pinned against a JVM here (none in the image)."""

    def __init__(self, seed: int):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def _next(self, bits: int) -> int:
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.seed >> (48 - bits)
        return v - (1 << bits) if v >= (1 << (bits - 1)) and bits == 32 else v

    def nextInt(self, bound: int) -> int:
        assert bound > 0
        if bound & (bound - 1) == 0:
            return (bound * self._next(31)) >> 31
        while True:
            bits = self._next(31)
            val = bits % bound
            if bits - val + (bound - 1) < (1 << 31):
                return val

    def shuffle(self, xs: Sequence[int]) -> List[int]:
        """scala.util.Random.shuffle (2.13): for n = length down to 2, swap(n - 1, nextInt(n))."""
        buf = list(xs)
        for n in range(len(buf), 1, -1):
            k = self.nextInt(n)
            buf[n - 1], buf[k] = buf[k], buf[n - 1]
        return buf


class BatchStream:
    """BatchStream over a device-resident data set; iterate to get (features, target) STen pairs until EndStream."""

    def __init__(self, handle):
        self.h = handle.value if isinstance(handle, C.c_void_p) else handle

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                lib.lamp_batch_stream_release(h)
            except Exception:
                pass

    @staticmethod
    def minibatchesFromFull(minibatchSize: int, dropLast: bool, features: STen, target: STen, rng=None, device: int = 0,
                            order: Optional[Sequence[int]] = None, hostResident: bool = False, outDtype: int = -1) -> "BatchStream":
        """`rng`: an object with shuffle(list) -> list (JavaRandom, or anything else); `order` overrides it.  hostResident: the features
        stay in (pinned) host memory as in the reference and every minibatch is gathered over PCIe one batch ahead, converted to outDtype."""
        n = features.shape[0]
        if order is None:
            order = rng.shuffle(list(range(n))) if rng is not None else list(range(n))
        arr = np.ascontiguousarray(order, dtype=np.int64)
        o = C.c_void_p()
        if hostResident:
            lib.lamp_batch_stream_from_full_host(C.byref(o), features, target, arr.ctypes.data_as(C.POINTER(C.c_int64)), len(arr), int(minibatchSize),
                                                 int(bool(dropLast)), device, int(outDtype))
        else:
            lib.lamp_batch_stream_from_full(C.byref(o), features, target, arr.ctypes.data_as(C.POINTER(C.c_int64)), len(arr), int(minibatchSize),
                                            int(bool(dropLast)), device)
        return BatchStream(o)

    def everyNth(self, n: int, offset: int) -> "BatchStream":
        lib.lamp_batch_stream_every_nth(self.h, int(n), int(offset))
        return self

    @property
    def numBatches(self) -> int:
        n = C.c_int64(); lib.lamp_batch_stream_num_batches(self.h, C.byref(n)); return n.value

    def nextBatch(self) -> Optional[Tuple[STen, STen]]:
        x, y = C.c_void_p(), C.c_void_p()
        lib.lamp_batch_stream_next(self.h, C.byref(x), C.byref(y))
        return None if not x.value else (STen(x), STen(y))

    def reset(self) -> None:
        lib.lamp_batch_stream_reset(self.h)

    def __iter__(self) -> Iterator[Tuple[STen, STen]]:
        while True:
            b = self.nextBatch()
            if b is None:
                return
            yield b
