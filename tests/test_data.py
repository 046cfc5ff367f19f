"""lamp-data rows: tensor-list files / checkpoints (Writer.scala, Reader.scala), Cifar.loadImageFile, BatchStream.

CPU tests follow lamp-data/src/test/scala/lamp/data/ReadWrite.test.scala ("io empty", "io empty 2") on host tensors, and
pin the descriptor text to the format the reference documents (Writer.scala:14-38, schemas.scala:30-56).  GPU tests follow
"checkpoint modules" and batchstream.test.scala.  The JVM is not available here: byte-level agreement with a file written by
the reference itself is unpinned; the descriptor grammar and blob layout are taken from the reference's specification."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from lamp_amd import data as D, sten as S
from lamp_amd._capi import LampError, lib


def _host(a, dtype=None):
    return S.STen.from_numpy(np.asarray(a), device=S.CPU, dtype=dtype)


def _reference_list():
    """the tensors of ReadWrite.test.scala 'io empty' (ones(23,23) f64 / f32 / i64, empties, ones(3,3), i8, bf16)"""
    return [_host(np.ones((23, 23))), _host(np.ones((23, 23), np.float32)), _host(np.ones((23, 23), np.int64)),
            _host(np.zeros((0,))), _host(np.zeros((0, 0), np.float32)), _host(np.zeros((0,), np.int64)), _host(np.ones((3, 3))),
            _host(np.ones((23,), np.int8)), _host(np.ones((23,), np.float32), dtype=S.BF16)]


@pytest.mark.parametrize("pin", [False, pytest.param(True, marks=pytest.mark.gpu)])   # pinned memory needs the HIP runtime's device
def test_io_round_trip_reference_list(tmp_path, pin):
    ts = _reference_list()
    f = str(tmp_path / "list")
    D.writeTensorsIntoFile(ts, f)
    back = D.readTensorsFromFile(f, S.CPU, pin)
    assert len(back) == len(ts)
    for a, b in zip(ts, back):
        assert a.shape == b.shape and a.dtype == b.dtype
        assert np.array_equal(a.to_numpy(), b.to_numpy())


def test_io_empty_2(tmp_path):
    ts = [_host(np.zeros((0,))), _host(np.zeros((0, 0), np.float32)), _host(np.zeros((0,), np.int64))]
    f = str(tmp_path / "e")
    D.writeTensorsIntoFile(ts, f)
    assert os.path.getsize(f + ".data") == 0
    back = D.readTensorsFromFile(f, S.CPU, False)
    assert [b.shape for b in back] == [[0], [0, 0], [0]] and [b.dtype for b in back] == [S.F64, S.F32, S.I64]
    D.writeTensorsIntoFile([], f)
    assert D.readTensorsFromFile(f, S.CPU, False) == []


def test_descriptor_and_blob_layout(tmp_path):
    """Known answer: the descriptor text (field order of the case classes, compact) and the blob bytes with 8-byte padding."""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)          # 24 B
    b = np.array([1, -2, 3], dtype=np.int8)                   # 3 B + 5 pad
    c = np.array([7], dtype=np.int64)                         # 8 B
    d = np.array([1.0, -2.5, 3.25], dtype=np.float32)         # bf16: 6 B + 2 pad
    f = str(tmp_path / "ck")
    D.writeTensorsIntoFile([_host(a), _host(b), _host(c), _host(d, dtype=S.BF16)], f)
    text = open(f).read()
    assert text == ('{"tensors":[{"dims":[2,3],"dataType":6,"byteOffset":0,"byteLength":24},'
                    '{"dims":[3],"dataType":1,"byteOffset":24,"byteLength":3},'
                    '{"dims":[1],"dataType":4,"byteOffset":32,"byteLength":8},'
                    '{"dims":[3],"dataType":15,"byteOffset":40,"byteLength":6}],'
                    '"location":"ck.data","byteOffset":0,"byteLength":48}')
    blob = open(f + ".data", "rb").read()
    bf = (np.array([1.0, -2.5, 3.25], np.float32).view(np.uint32) >> 16).astype("<u2").tobytes()
    assert blob == a.astype("<f4").tobytes() + b.tobytes() + b"\0" * 5 + c.astype("<i8").tobytes() + bf + b"\0" * 2


def test_reader_accepts_foreign_descriptor_and_checks_it(tmp_path):
    """A descriptor as another writer may produce it: whitespace, reordered and unknown fields, absolute location, padding gaps;
    and the reference's assertions (8-byte aligned tensor offsets, bounds, 4096-aligned list offset)."""
    blob = str(tmp_path / "w.bin")
    x = np.arange(5, dtype=np.float64)
    with open(blob, "wb") as fh:
        fh.write(b"\xff" * 16 + x.tobytes() + b"\0" * 8 + np.int64(9).tobytes())
    desc = {"byteLength": 72, "byteOffset": 0, "location": blob, "extra": {"k": [1, 2, {"z": None}]},
            "tensors": [{"byteLength": 40, "byteOffset": 16, "dataType": 7, "dims": [5], "note": "x"},
                        {"dims": [], "dataType": 4, "byteOffset": 64, "byteLength": 8}]}
    f = str(tmp_path / "w.json")
    json.dump(desc, open(f, "w"), indent=2)
    got = D.readTensorsFromFile(f)
    assert np.array_equal(got[0].to_numpy(), x) and got[1].shape == [] and int(got[1].to_numpy()) == 9
    for patch, msg in (({"byteOffset": 100}, "multiple of 4096"), ({"byteLength": 48}, "out of bound")):
        json.dump({**desc, **patch}, open(f, "w"))
        with pytest.raises(LampError, match=msg):
            D.readTensorsFromFile(f)
    bad = json.loads(json.dumps(desc)); bad["tensors"][0]["byteOffset"] = 12
    json.dump(bad, open(f, "w"))
    with pytest.raises(LampError, match="aligned to 8"):
        D.readTensorsFromFile(f)
    bad = json.loads(json.dumps(desc)); bad["tensors"][0]["dims"] = [4]
    json.dump(bad, open(f, "w"))
    with pytest.raises(LampError, match="do not match byteLength"):
        D.readTensorsFromFile(f)
    with pytest.raises(LampError, match="cannot open"):
        D.readTensorsFromFile(str(tmp_path / "missing"))


def test_java_random_known_answers():
    """java.util.Random's published behaviour: new Random(42).nextInt(10) sequence starts 0, 3, 8, 4, 0, 5, 5, 8, 9, 3 (widely
    quoted) and new Random(0).nextInt() & power-of-two bounds follow the documented LCG."""
    r = D.JavaRandom(42)
    assert [r.nextInt(10) for _ in range(10)] == [0, 3, 8, 4, 0, 5, 5, 8, 9, 3]
    r = D.JavaRandom(0)
    assert [r.nextInt(16) for _ in range(4)] == [(16 * v) >> 31 for v in _lcg31(0, 4)]
    p = D.JavaRandom(7).shuffle(list(range(100)))
    assert sorted(p) == list(range(100)) and p != list(range(100))


def _lcg31(seed, n):
    s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)
    out = []
    for _ in range(n):
        s = (s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        out.append(s >> 17)
    return out


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_checkpoint_modules_float_and_mixed(gpu, tmp_path):
    """ReadWrite.test.scala 'checkpoint modules - float' / '- mixed': write, load into a fresh module, states identical."""
    from lamp_amd import nn
    for dt2 in (S.F32, S.F64):
        net = nn.Sequential(nn.Linear(5, 5, S.F32), nn.Linear(5, 5, dt2))
        f = str(tmp_path / f"ck{dt2}")
        D.writeCheckpoint(f, net)
        net2 = nn.Sequential(nn.Linear(5, 5, S.F32), nn.Linear(5, 5, dt2))
        assert any(not np.array_equal(a.value.to_numpy(), b.value.to_numpy()) for a, b in zip(net.state, net2.state))
        D.loadFromFile(net2, f)
        for a, b in zip(net.state, net2.state):
            assert a.value.dtype == b.value.dtype and np.array_equal(a.value.to_numpy(), b.value.to_numpy())
    # the file is an ordinary tensor list: readable straight onto the device, bf16 ResNet state included
    net = nn.resnet(100, 0.0, S.BF16)
    f = str(tmp_path / "resnet")
    D.writeCheckpoint(f, net)
    back = D.readTensorsFromFile(f, 0, True)
    st = net.state
    assert len(back) == len(st)
    for a, b in zip(st, back):
        assert b.device == 0 and a.value.shape == b.shape and np.array_equal(a.value.to_numpy(), b.to_numpy())
    with pytest.raises(LampError, match="holds"):
        D.loadFromFile(nn.Sequential(net, nn.Linear(3, 3)), f)


@pytest.mark.gpu
def test_cifar_records(gpu, tmp_path):
    rng = np.random.default_rng(1)
    n = 257
    rec = rng.integers(0, 256, (n, 3074), dtype=np.uint8)
    rec[:, 1] = rng.integers(0, 100, n)
    f = str(tmp_path / "train.bin")
    with open(f, "wb") as fh:
        fh.write(rec.tobytes() + b"tail")                       # trailing bytes are ignored: length = numImages * 3074
    for dt, npdt in ((S.F32, np.float32), (S.F64, np.float64), (S.BF16, np.float32)):
        lab, img = D.loadImageFile(f, n, dt, 0)
        assert lab.dtype == S.I64 and np.array_equal(lab.to_numpy(), rec[:, 1].astype(np.int64))
        assert img.shape == [n, 3, 32, 32] and img.dtype == dt
        assert np.array_equal(img.to_numpy(), rec[:, 2:].reshape(n, 3, 32, 32).astype(npdt))   # 0..255 are exact in bf16
    with pytest.raises(LampError, match="fewer than"):
        D.loadImageFile(f, n + 1, S.F32, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("host_resident", [False, True])
def test_minibatches_from_full(gpu, host_resident):
    """batchstream.test.scala: every row is served exactly once per epoch in the given order, groups of minibatchSize, dropLast
    drops the LAST group (full or not), everyNth(n, k) keeps batches i % n == k."""
    n, mb = 103, 10
    x = np.arange(n * 6, dtype=np.float32).reshape(n, 2, 3)
    y = np.arange(n, dtype=np.int64) * 7
    dev = S.CPU if host_resident else 0
    fx, fy = S.STen.from_numpy(x, dev), S.STen.from_numpy(y, dev)
    order = D.JavaRandom(123).shuffle(list(range(n)))
    for drop in (False, True):
        st = D.BatchStream.minibatchesFromFull(mb, drop, fx, fy, order=order)
        groups = [order[i:i + mb] for i in range(0, n, mb)]
        if drop:
            groups = groups[:-1]
        assert st.numBatches == len(groups)
        for epoch in range(2):
            got = list(st)
            assert len(got) == len(groups)
            for (bx, by), g in zip(got, groups):
                assert bx.device == 0 and np.array_equal(bx.to_numpy(), x[g]) and np.array_equal(by.to_numpy(), y[g])
            assert st.nextBatch() is None
            st.reset()
    # a full last group is dropped too
    st = D.BatchStream.minibatchesFromFull(10, True, fx, fy, order=order[:100])
    assert st.numBatches == 9
    # everyNth: the shards of a 3-rank job partition the batches
    seen = []
    for r in range(3):
        st = D.BatchStream.minibatchesFromFull(mb, False, fx, fy, order=order).everyNth(3, r)
        idx = [i for i in range(11) if i % 3 == r]
        assert st.numBatches == len(idx)
        for (bx, by), i in zip(st, idx):
            assert np.array_equal(by.to_numpy(), y[order[i * mb:(i + 1) * mb]])
            seen.append(i)
    assert sorted(seen) == list(range(11))
    with pytest.raises(LampError, match="out of range"):
        D.BatchStream.minibatchesFromFull(mb, False, fx, fy, order=[0, n])


@pytest.mark.gpu
def test_minibatches_from_a_data_set_that_stays_in_host_memory(gpu):
    """BatchStream.minibatchesFromFull as the reference arranges it (BatchStream.scala:539-556: the records stay on the host, pinned; the
    minibatch travels on another stream one batch ahead, IOLoops.scala:833-874): the GPU gathers the rows over PCIe.  Same batches, bit for
    bit, as the device-resident stream - plain, with the cast folded into the gather (u8 records -> f32 / bf16, f32 -> bf16), under
    everyNth, across epochs and after a reset in the middle of an epoch; out-of-range rows raise."""
    n, mb = 1003, 64
    rng = np.random.default_rng(3)
    x32 = rng.standard_normal((n, 3, 5, 7)).astype(np.float32)
    x8 = rng.integers(0, 256, (n, 3, 5, 7)).astype(np.uint8)
    y = np.arange(n, dtype=np.int64) * 7
    order = D.JavaRandom(5).shuffle(list(range(n)))
    fy = S.STen.from_numpy(y, S.CPU)
    for host, out_dt, ref in ((x32, -1, x32), (x8, S.F32, x8.astype(np.float32)), (x32, S.BF16, None), (x8, S.BF16, None)):
        fx = S.STen.from_numpy(host, S.CPU)
        st = D.BatchStream.minibatchesFromFull(mb, False, fx, fy, order=order, hostResident=True, outDtype=out_dt)
        if ref is None:                                        # the device-resident stream of the same records cast on the GPU
            dref = D.BatchStream.minibatchesFromFull(mb, False, S.STen.from_numpy(host.astype(np.float32), 0, S.BF16), fy, order=order)
        assert st.numBatches == (n + mb - 1) // mb
        for epoch in range(2):
            got = list(st)
            assert len(got) == st.numBatches
            want = list(dref) if ref is None else None
            for i, (bx, by) in enumerate(got):
                g = order[i * mb:(i + 1) * mb]
                assert bx.device == 0 and by.device == 0 and np.array_equal(by.to_numpy(), y[g])
                assert np.array_equal(bx.to_numpy(), want[i][0].to_numpy() if ref is None else ref[g])
            st.reset()
            if ref is None:
                dref.reset()
        # a reset in the middle of an epoch discards the batch that was gathered ahead
        it = iter(st); next(it); next(it)
        st.reset()
        bx, by = st.nextBatch()
        assert np.array_equal(by.to_numpy(), y[order[:mb]])
    shards = []
    for r in range(3):
        st = D.BatchStream.minibatchesFromFull(mb, False, S.STen.from_numpy(x32, S.CPU), fy, order=order, hostResident=True).everyNth(3, r)
        for bx, by in st:
            shards.append(by.to_numpy())
    assert sorted(np.concatenate(shards).tolist()) == sorted(y.tolist())
    with pytest.raises(LampError, match="host memory"):
        D.BatchStream.minibatchesFromFull(mb, False, S.STen.from_numpy(x32, 0), fy, order=order, hostResident=True)
    # a conversion the gather does not have is refused when the stream is built, not at the first batch
    with pytest.raises(LampError, match="no gather converts"):
        D.BatchStream.minibatchesFromFull(mb, False, S.STen.from_numpy(y.reshape(n, 1), S.CPU), fy, order=order, hostResident=True, outDtype=S.BF16)
    # a consumer whose current stream changes between two calls still reads a complete batch (the batch handed out was queued one call
    # ago, on the stream that was current then): alternate the consumer between two streams, batch contents as above
    side = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(side))
    dflt = C.c_void_p(); lib.lamp_stream_get_default(0, C.byref(dflt))
    st = D.BatchStream.minibatchesFromFull(mb, False, S.STen.from_numpy(x32, S.CPU), fy, order=order, hostResident=True)
    try:
        for i in range(st.numBatches):
            lib.lamp_stream_set_current(side if i % 2 == 0 else dflt)
            bx, by = st.nextBatch()
            two = bx._u("lamp_mul_scalar", 2.0)             # consumed on the current stream at once
            g = order[i * mb:(i + 1) * mb]
            assert np.array_equal(bx.to_numpy(), x32[g]) and np.array_equal(by.to_numpy(), y[g])
            assert np.array_equal(two.to_numpy(), x32[g] * 2)
    finally:
        lib.lamp_stream_set_current(dflt)
        lib.lamp_device_synchronize()


@pytest.mark.gpu
def test_training_from_stream_and_resume(gpu, tmp_path):
    """End to end: CIFAR-style records -> device-resident stream -> a few training steps -> checkpoint -> a fresh model loaded
    from it continues bit-identically to the original."""
    import ctypes as C
    from lamp_amd import nn
    from lamp_amd._capi import lib
    rng = np.random.default_rng(3)
    n = 64
    rec = rng.integers(0, 256, (n, 3074), dtype=np.uint8)
    rec[:, 1] = np.arange(n) % 100
    f = str(tmp_path / "c.bin")
    open(f, "wb").write(rec.tobytes())
    lab, img = D.loadImageFile(f, n, S.F32, 0)

    def make():
        lib.lamp_manual_seed(5)
        net = nn.resnet(100, 0.0, S.F32)
        model = nn.SupervisedModel(net, nn.SupervisedModel.NLL, S.STen.ones([100], S.F32, 0))
        opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3)([p.value for p in net.parameters])
        return net, model, opt

    def run(model, opt, order):
        st = D.BatchStream.minibatchesFromFull(16, False, img, lab, order=order)
        for bx, by in st:
            model.train_step(opt, bx, by)

    net, model, opt = make()
    run(model, opt, list(range(n)))
    ck = str(tmp_path / "ck")
    D.writeCheckpoint(ck, net)
    ock = str(tmp_path / "ock")
    D.writeTensorsIntoFile(opt.state, ock)
    net2, model2, opt2 = make()
    D.loadFromFile(net2, ck)
    opt2.load(D.readTensorsFromFile(ock, 0))
    order2 = list(reversed(range(n)))
    run(model, opt, order2)
    run(model2, opt2, order2)
    for a, b in zip(net.state, net2.state):
        assert np.array_equal(a.value.to_numpy(), b.value.to_numpy())


@pytest.mark.gpu
def test_one_epoch_loops(gpu):
    """IOLoops.oneEpoch / validationOneEpoch: the epoch loop equals the hand-written per-batch loop; accumulating gradients over N
    batches steps once per N batches with the summed gradient; validation runs in eval mode and leaves the parameters alone."""
    from lamp_amd import nn, loops
    from lamp_amd._capi import lib
    rng = np.random.default_rng(0)
    n = 96
    x = S.STen.from_numpy(rng.standard_normal((n, 20)).astype(np.float32), 0)
    y = S.STen.from_numpy((np.arange(n) % 4).astype(np.int64), 0)

    def make():
        lib.lamp_manual_seed(11)
        net = nn.Sequential(nn.MLP(20, 4, [16], S.F32, 0), nn.Fun("logsoftmax", 1))
        model = nn.SupervisedModel(net, nn.SupervisedModel.NLL, S.STen.ones([4], S.F32, 0))
        opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-2)([p.value for p in net.parameters])
        return net, model, opt

    order = list(range(n))
    # (1) epoch loop == manual loop
    net1, model1, opt1 = make()
    st1 = D.BatchStream.minibatchesFromFull(16, False, x, y, order=order)
    logs = []
    l1 = loops.oneEpoch(0, model1, opt1, st1, logger=logs.append)
    assert logs and "Avg training loss in epoch 0 over 96 examples" in logs[0]
    net2, model2, opt2 = make()
    acc = S.STen.zeros([1], S.F32, 0)
    tot = 0
    for bx, by in D.BatchStream.minibatchesFromFull(16, False, x, y, order=order):
        tot += model2.train_step(opt2, bx, by, acc)
    assert tot == n and abs(l1 - float(acc.to_numpy()[0]) / n) < 1e-7
    for a, b in zip(net1.state, net2.state):
        assert np.array_equal(a.value.to_numpy(), b.value.to_numpy())
    # (2) accumulation over 2 batches of 16 == one step on the summed gradients of the two batches
    net3, model3, opt3 = make()
    st3 = D.BatchStream.minibatchesFromFull(16, False, x, y, order=order[:32])
    loops.oneEpoch(0, model3, opt3, st3, accumulateGradientOverNBatches=2)
    net4, model4, opt4 = make()
    net4.zeroGrad()
    grads = None
    for bx, by in D.BatchStream.minibatchesFromFull(16, False, x, y, order=order[:32]):
        _, grads = model4.addTotalLossAndReturnGradientsAndNumExamples(bx, by, None, False)
    opt4.step(grads, 1.0)
    for a, b in zip(net3.state, net4.state):
        assert np.allclose(a.value.to_numpy(), b.value.to_numpy(), rtol=0, atol=1e-7)
    # (3) validation: eval mode (batch norm uses its running statistics), parameters untouched, mode restored
    before = [v.value.to_numpy().copy() for v in net1.state]
    vl = loops.validationOneEpoch(model1, D.BatchStream.minibatchesFromFull(32, False, x, y, order=order))
    assert np.isfinite(vl) and all(np.array_equal(a, v.value.to_numpy()) for a, v in zip(before, net1.state))
    # a learning-rate factor of 0 leaves the parameters where they are (the schedule factor reaches the optimizer)
    loops.oneEpoch(1, model1, opt1, st1, learningRateScheduleFactor=0.0)
    changed = [not np.array_equal(a, v.value.to_numpy()) for a, v in zip(before, net1.state) if v.needsGrad]
    assert not any(changed)


def test_host_staging_cat_out_without_gpu():
    """the host half of Device.toBatched (device.scala:80-92): flat views concatenated into a slice of the staging buffer"""
    from lamp_amd._capi import lib, handle_array
    a, b = _host(np.arange(6, dtype=np.float32).reshape(2, 3)), _host(np.arange(4, dtype=np.float32) + 10)
    buf = _host(np.zeros(16, np.float32))
    va, vb, dst = a.view(-1), b.view(-1), buf.slice(0, 0, 10)      # keep the handles alive across the call
    lib.lamp_cat_out(dst, handle_array([va.h, vb.h]), 2, 0)
    assert np.array_equal(buf.to_numpy(), np.concatenate([np.arange(6), np.arange(4) + 10, np.zeros(6)]).astype(np.float32))


@pytest.mark.gpu
def test_to_batched(gpu):
    """Device.toBatched: several host tensors reach the device through one pinned buffer and one copy"""
    rng = np.random.default_rng(0)
    arrs = [rng.standard_normal(s).astype(np.float32) for s in ((3, 4), (5,), (2, 2, 2), (1,))]
    bufs = S.BufferPair.allocate(64, 0, S.F32)
    out = S.toBatched([_host(a) for a in arrs], bufs)
    for a, t in zip(arrs, out):
        assert t.device == 0 and t.shape == list(a.shape) and np.array_equal(t.to_numpy(), a)
    out2 = S.toBatched([_host(a + 1) for a in arrs], bufs)          # the buffers are reused; earlier results are clones
    for a, t, t2 in zip(arrs, out, out2):
        assert np.array_equal(t.to_numpy(), a) and np.array_equal(t2.to_numpy(), a + 1)


@pytest.mark.gpu
def test_recycled_pinned_blocks_wait_for_copies_that_still_read_them(gpu):
    """core/tensor.hip: a freed page-locked block is handed out again only after EVERY device this process has used is idle (what hipHostFree's
    implicit wait guaranteed) - also when the next owner is a thread that never selected a device (the reference's loader fiber).  A 64 MB
    pinned source is copied to the device without blocking and released at once; another thread pins a tensor of the same size (it receives
    the recycled block: same address) and overwrites it; the device copy must hold the OLD contents, every time."""
    import threading
    n = 16 << 20
    for trial in range(6):
        src = S.STen.from_numpy(np.full(n, float(trial + 1), dtype=np.float32), S.CPU).pin()
        addr = src.data_ptr
        dev = S.STen.zeros([n], S.F32, 0)
        dev.copyFrom(src, True)                          # non_blocking: the DMA may still be reading `src` when it is released
        src.release()
        seen = {}

        def loader():
            t = S.STen.from_numpy(np.full(n, -1.0, dtype=np.float32), S.CPU).pin()   # recycles the block and writes it
            seen["addr"] = t.data_ptr
            t.release()
        th = threading.Thread(target=loader); th.start(); th.join()
        got = dev.to_numpy()
        assert np.all(got == np.float32(trial + 1)), f"trial {trial}: the copy read a recycled block ({got[:4]})"
        assert seen["addr"] == addr, "the freed pinned block was not recycled (the test would prove nothing)"
        dev.release()
