"""N > 1 logic on CPU.

* two gloo processes run the exchange of the data-parallel step with the PRODUCT's bucket code: every rank packs its oracle
  gradients with lamp_flatten_into_ (host tensors, scale = local example count, example count appended), gloo all-reduces the
  bucket, lamp_unflatten_from_ divides by the summed count - compared with the oracle's averageGradients;
* the TCP control plane of lamp_amd.distributed (unique-id hand-over, barrier, max over ranks) between two processes;
* bench.py's own launcher: `--gpus 2 --dry-launch` starts two ranks that rendezvous, and a real `--gpus 2` run on a box
  without two GPUs exits non-zero instead of printing a line.

The device kernels of the same step are covered on the GPU in test_flat_bucket_and_single_rank_collectives, and the complete
overlapped two-bucket step (second stream, events, AdamW) in test_overlapped_data_parallel_step_single_rank (world size 1).
"""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from lamp_amd import distributed as D
from oracle import lamp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from lamp_amd import sten as S
    from lamp_amd._capi import lib, handle_array
    lib.load()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # local gradients of a small model on this rank's shard (uneven shards: the weighting matters)
    torch.manual_seed(0)
    m = O.Sequential(O.mlp(12, 3, [8], torch.float64), O.Fun(lambda v: v.logSoftMax(1)))
    n_r = [5, 3][rank]
    x = O.closed_form(8 * 12, 0, 1.0, torch.float64).reshape(8, 12)[sum([5, 3][:rank]):][:n_r]
    t = (torch.arange(8) % 3)[sum([5, 3][:rank]):][:n_r]
    _, grads = O.training_step(m, O.nll_loss(3, torch.ones(3, dtype=torch.float64)), x, t, None)
    # the product's bucket code on host tensors: averageGradients (distributed/package.scala:690-719) as
    # bucket = [n_r * g ... | n_r] -> all-reduce(sum) -> g = bucket / bucket[last]
    hg = [S.STen.from_numpy(g.numpy(), S.CPU) for g in grads]
    offs, total = D.bucket_layout([g.numel() for g in grads])
    bucket = S.STen.zeros([total], S.F32, S.CPU)
    lib.lamp_flatten_into_(bucket, handle_array([g.h for g in hg]), len(hg), float(n_r))
    packed = bucket.to_numpy().copy()
    for g, o in zip(grads, offs):                              # the layout bucket_layout describes is the one the library writes
        assert np.array_equal(packed[o:o + g.numel()], (g.reshape(-1).float() * float(n_r)).numpy())   # f32(scale) * f32(g), as the device kernel
    packed[-1] = n_r
    tb = torch.from_numpy(packed)
    dist.all_reduce(tb)                                        # lamp_comm_all_reduce on the GPU path
    summed = S.STen.from_numpy(tb.numpy(), S.CPU)
    outs = [S.STen.zeros(list(g.shape), S.F64, S.CPU) for g in grads]
    lib.lamp_unflatten_from_(handle_array([o.h for o in outs]), len(outs), summed, 1)
    avg = [torch.from_numpy(o.to_numpy()) for o in outs]
    torch.save({"grads": grads, "avg": avg, "n": n_r}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    ref = O.average_gradients([r0["grads"], r1["grads"]], [r0["n"], r1["n"]])      # distributed/package.scala:690-719
    for a0, a1, r in zip(r0["avg"], r1["avg"], ref):
        assert torch.equal(a0, a1), "every rank must hold the same averaged gradient"
        assert torch.allclose(a0.double(), r, rtol=1e-6, atol=1e-7)


_CP_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
from lamp_amd import distributed as D
cp = D.init_control_plane(timeout=30.0)
r = cp.get_rank()
uid = D.exchange_unique_id(cp, lambda: bytes(range(128)))
assert uid == bytes(range(128)), "every rank ends up with root's 128 bytes"
assert cp.all_reduce_max(1.0 + r) == float(cp.get_world_size())
assert cp.all_reduce_sum(0.5) == 0.5 * cp.get_world_size()
assert cp.all_gather({"rank": r}) == [{"rank": k} for k in range(cp.get_world_size())]
for _ in range(50):
    cp.barrier()
assert cp.broadcast_bytes(b"xyz" if r == 1 else None, root=1) == b"xyz"
cp.close()
print("ok", r)
"""


@pytest.mark.parametrize("world", [2, 3])
def test_tcp_control_plane(tmp_path, world):
    """rank 0 publishes its port in the rendezvous file, the others find it; unique-id hand-over, barrier, max, gather."""
    rdzv = str(tmp_path / "rdzv.json")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="1", LAMP_RDZV_FILE=rdzv)
        procs.append(subprocess.Popen([sys.executable, "-c", _CP_SCRIPT, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p_ in enumerate(procs):
        out, err = p_.communicate(timeout=120)
        assert p_.returncode == 0, err
        assert out.strip() == f"ok {r}"
    assert not os.path.exists(rdzv), "rank 0 removes the rendezvous record when it closes"


def test_control_plane_rejects_a_rank_of_another_launch(tmp_path):
    """a process that connects with the wrong world size / nonce is turned away and the clique still completes"""
    import threading
    rdzv = str(tmp_path / "r.json")
    box = {}
    th = threading.Thread(target=lambda: box.setdefault("cp", D.ControlPlane(0, 2, "127.0.0.1", 0, rdzv, 30.0)))
    th.start()
    import time
    while not os.path.exists(rdzv):
        time.sleep(0.01)
    rec = json.load(open(rdzv))
    s = socket.create_connection(("127.0.0.1", rec["port"]))
    D._send_msg(s, {"rank": 1, "world": 5, "nonce": rec["nonce"]})
    assert D._recv_msg(s)["ok"] is False
    # a rank that gave up waiting and closed its connection after a valid hello is not counted as joined
    gone = socket.create_connection(("127.0.0.1", rec["port"]))
    D._send_msg(gone, {"rank": 1, "world": 2, "nonce": rec["nonce"]})
    gone.close()
    s.close()
    good = D.ControlPlane(1, 2, "127.0.0.1", 0, rdzv, 30.0)
    th.join(30)
    assert box["cp"].peers[1] is not None
    good.close(); box["cp"].close()


def test_control_plane_survives_stray_connections(tmp_path):
    """garbage, an absurd length prefix, a hello without a rank and a client that says nothing must not take rank 0 down,
    make it allocate gigabytes or hold the clique up for the full timeout; the record is owner-only"""
    import threading
    import time
    rdzv = str(tmp_path / "r.json")
    box = {}

    def serve():
        try:
            box["cp"] = D.ControlPlane(0, 2, "127.0.0.1", 0, rdzv, 60.0)
        except Exception as e:                     # noqa: BLE001 - the test reports it
            box["err"] = e
    th = threading.Thread(target=serve)
    th.start()
    while not os.path.exists(rdzv):
        time.sleep(0.01)
    assert (os.stat(rdzv).st_mode & 0o077) == 0, "the nonce must not be readable by other users"
    rec = json.load(open(rdzv))
    strays = []
    for payload in (b"GET / HTTP/1.1\r\n\r\n", b"\xff\xff\xff\xff" + b"x" * 16, None):
        s = socket.create_connection(("127.0.0.1", rec["port"]))
        if payload is not None:
            s.sendall(payload)
        strays.append(s)                           # the silent one stays open: rank 0 drops it after its 5 s handshake timeout
    s = socket.create_connection(("127.0.0.1", rec["port"]))
    D._send_msg(s, {"world": 2})                   # no rank
    strays.append(s)
    t0 = time.monotonic()
    good = D.ControlPlane(1, 2, "127.0.0.1", 0, rdzv, 60.0)
    th.join(60)
    assert "err" not in box, box.get("err")
    assert box["cp"].peers[1] is not None
    assert time.monotonic() - t0 < 30.0
    for s in strays:
        s.close()
    good.close(); box["cp"].close()


def test_bench_dry_launch_two_ranks():
    """`python bench.py --gpus 2` started bare becomes the launcher: two rank processes with the torchrun environment rendezvous."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line == {"dry_launch": True, "ranks": 2, "rank_list": [0, 1], "max_rank": 1.0, "n_gpus": 2}


def test_bench_under_a_launcher_checks_the_world_size():
    """inside a launcher's environment bench.py is ONE rank, and --gpus must agree with WORLD_SIZE"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch"], env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def _visible_gpus():
    n = C.c_int(0)
    from lamp_amd._capi import lib
    lib.load()
    lib.lamp_get_num_gpus(C.byref(n))
    return n.value


def test_bench_multi_gpu_refuses_to_fabricate():
    """a bare `--gpus 2` on a box with fewer than two GPUs must exit non-zero and print no result line (round 1 printed 2 x the
    single-GPU figure)"""
    if _visible_gpus() >= 2:
        pytest.skip("this box really has two GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not any(l.startswith("{") for l in out.stdout.splitlines()), out.stdout


def test_every_nth_sharding():
    assert D.every_nth(10, 4, 1) == [1, 5]            # two full rounds; the tail (8, 9) is dropped on every rank
    shards = [D.every_nth(12, 3, r) for r in range(3)]
    assert sorted(sum(shards, [])) == list(range(12)) and len({len(s) for s in shards}) == 1
    assert D.bucket_layout([4, 6, 2]) == ([0, 4, 10], 13)


@pytest.mark.gpu
def test_flat_bucket_and_single_rank_collectives(gpu):
    from lamp_amd import sten as S
    from lamp_amd._capi import lib, handle_array
    g = [S.STen.from_numpy(np.arange(6, dtype=np.float32).reshape(2, 3)), S.STen.from_numpy(np.ones(4, dtype=np.float32) * 2)]
    offs, total = D.bucket_layout([t.numel for t in g])
    bucket = S.STen.zeros([total], S.F32)
    lib.lamp_flatten_into_(bucket, handle_array([t.h for t in g]), 2, 3.0)
    bucket.slice(0, total - 1, total).fill_(3.0)
    uid = (C.c_uint8 * 128)(); lib.lamp_comm_get_unique_id(uid)
    comm = C.c_void_p(); lib.lamp_comm_init_rank(C.byref(comm), 1, uid, 0)
    lib.lamp_comm_all_reduce(handle_array([bucket.h]), handle_array([comm]), 1, 0)
    lib.lamp_comm_broadcast(handle_array([bucket.h]), handle_array([comm]), 1, 0)
    lib.lamp_comm_reduce(handle_array([bucket.h]), bucket, 0, 0, handle_array([comm]), 1)
    assert np.allclose(bucket.to_numpy(), np.concatenate([np.arange(6) * 3.0, np.ones(4) * 6.0, [3.0]]))
    out = [S.STen.zeros([2, 3]), S.STen.zeros([4])]
    lib.lamp_unflatten_from_(handle_array([t.h for t in out]), 2, bucket, 1)
    assert np.allclose(out[0].to_numpy(), np.arange(6).reshape(2, 3)) and np.allclose(out[1].to_numpy(), 2.0)
    lib.lamp_comm_destroy(comm)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_overlapped_data_parallel_step_single_rank(gpu, dtype_name):
    """lamp_model_train_step with a world-size-1 RCCL communicator runs the whole two-bucket / second-stream exchange
    (pack * n, all-reduce, / sum n, AdamW).  With one rank the averaged gradient is the local one, so three steps must
    land on the parameters of the plain step (n = 64 is a power of two: *n and /n are exact)."""
    from lamp_amd import sten as S, nn
    from lamp_amd._capi import lib
    dt = S.F32 if dtype_name == "f32" else S.BF16
    B = 64
    x = S.STen.from_numpy((np.arange(B * 3 * 32 * 32) * 7919 % 1009 / 1009.0 - 0.5).reshape(B, 3, 32, 32).astype(np.float32), 0, dt)
    target = S.STen.from_numpy((np.arange(B) * 7 % 100).astype(np.int64), 0)
    uid = (C.c_uint8 * 128)(); lib.lamp_comm_get_unique_id(uid)
    comm = C.c_void_p(); lib.lamp_comm_init_rank(C.byref(comm), 1, uid, 0)

    def run(use_comm):
        lib.lamp_manual_seed(99)
        mod = nn.resnet(100, 0.0, dt, 0)
        model = nn.SupervisedModel(mod, nn.SupervisedModel.NLL, S.STen.ones([100], dt, 0))
        opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=(dt == S.BF16))([p.value for p in mod.parameters])
        acc = S.STen.zeros([1], dt, 0)
        for _ in range(3):
            assert model.train_step(opt, x, target, acc, comm if use_comm else None) == B
        lib.lamp_device_synchronize()
        return [s.value.castToDouble().to_numpy() for s in mod.state], acc.castToDouble().to_numpy()

    plain, acc0 = run(False)
    dp, acc1 = run(True)
    lib.lamp_comm_destroy(comm)
    assert np.array_equal(acc0, acc1)
    for a, b in zip(plain, dp):
        assert np.array_equal(a, b), "single-rank data-parallel step must reproduce the plain step bit for bit"


@pytest.mark.gpu
def test_graph_replay_with_gradient_exchange_single_rank(gpu):
    """bench.py's multi-rank step: forward + backprop replayed from a HIP graph, then lamp_model_exchange_and_step (one flat bucket
    * n, all-reduce, / sum n, AdamW) on the replayed gradients.  With a world-size-1 communicator three steps must land on the
    parameters of the plain eager step (n = 64: the scaling is exact)."""
    from lamp_amd import sten as S, nn
    from lamp_amd._capi import lib
    dt = S.BF16
    B = 64
    x = S.STen.from_numpy((np.arange(B * 3 * 32 * 32) * 7919 % 1009 / 1009.0 - 0.5).reshape(B, 3, 32, 32).astype(np.float32), 0, dt)
    target = S.STen.from_numpy((np.arange(B) * 7 % 100).astype(np.int64), 0)
    uid = (C.c_uint8 * 128)(); lib.lamp_comm_get_unique_id(uid)
    comm = C.c_void_p(); lib.lamp_comm_init_rank(C.byref(comm), 1, uid, 0)

    def fresh():
        lib.lamp_manual_seed(99)
        mod = nn.resnet(100, 0.0, dt, 0)
        model = nn.SupervisedModel(mod, nn.SupervisedModel.NLL, S.STen.ones([100], dt, 0))
        opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=True)([p.value for p in mod.parameters])
        return mod, model, opt, S.STen.zeros([1], dt, 0)

    mod, model, opt, acc = fresh()
    for _ in range(3):
        model.train_step(opt, x, target, acc, None)
    lib.lamp_device_synchronize()
    plain = [s.value.castToDouble().to_numpy() for s in mod.state]

    mod, model, opt, acc = fresh()
    lib.lamp_device_synchronize()
    st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(st)); lib.lamp_stream_set_current(st)
    model.sync_state(opt, comm, 0)
    n, grads = model.addTotalLossAndReturnGradientsAndNumExamples(x, target, acc)      # eager once: caches, attributes
    model.exchange_and_step(opt, grads, n, comm)
    lib.lamp_device_synchronize()
    lib.lamp_graph_begin_capture()
    n, grads = model.addTotalLossAndReturnGradientsAndNumExamples(x, target, acc)
    g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
    for _ in range(2):
        lib.lamp_graph_launch(g)
        model.exchange_and_step(opt, grads, n, comm)
    lib.lamp_device_synchronize()
    dflt = C.c_void_p(); lib.lamp_stream_get_default(0, C.byref(dflt)); lib.lamp_stream_set_current(dflt)
    got = [s.value.castToDouble().to_numpy() for s in mod.state]
    lib.lamp_comm_destroy(comm)
    for a, b in zip(plain, got):
        assert np.array_equal(a, b), "graph replay + exchange_and_step must reproduce the plain step bit for bit"


def test_row_shards_cover_every_row_once():
    """host logic of the row-sharded kNN graph: the per-rank blocks partition the rows, and the fixed-size contributions
    of the all-gather (short / empty last blocks re-query the tail rows) can always be cut back to exactly those blocks."""
    for n in (1, 7, 64, 1000, 1001):
        for world in (1, 2, 3, 8):
            per = (n + world - 1) // world
            seen = []
            for r in range(world):
                lo, hi = D.row_shard(n, world, r)
                assert 0 <= lo <= hi <= n and hi - lo <= per
                seen += list(range(lo, hi))
                if hi > lo and per <= n:
                    q = max(min(lo, n - per), 0)                 # first row this rank really queries
                    assert q <= lo and q + per >= hi, "the shifted block still contains the rank's rows"
            assert seen == list(range(n))


@pytest.mark.gpu
def test_sharded_knn_single_rank_and_all_gather(gpu):
    """world size 1: the sharded driver returns the plain graph; lamp_comm_all_gather on a 1-rank RCCL communicator copies."""
    from lamp_amd import sten as S, umap as U
    from lamp_amd._capi import lib
    rng = np.random.default_rng(3)
    X = S.STen.from_numpy(rng.random((500, 64), dtype=np.float32), 0)
    ref = U.knn_search(X, X, 5).to_numpy()
    assert np.array_equal(D.knn_search_sharded(X, 5, None, 1, 0).to_numpy(), ref)
    buf = (C.c_uint8 * 128)()
    lib.lamp_comm_get_unique_id(buf)
    h = C.c_void_p()
    lib.lamp_comm_init_rank(C.byref(h), 1, buf, 0)
    try:
        got = D.knn_search_sharded(X, 5, h, 1, 0).to_numpy()
        assert np.array_equal(got, ref)
        # the sharded UMAP layout driver through a real (1-rank) communicator: its all-reduces run and change nothing
        dist = np.sort(rng.random((500, 5)), 1); dist[:, 0] = 0.0
        ew = U.edge_weights(S.STen.from_numpy(dist, 0, S.F64), S.STen.from_numpy(ref.astype(np.int64), 0))
        l1, v1 = U.optimize(ew, 500, 0.1, 4, 0.0, 5, 42, True, 1.0, 0, 2)
        l2, v2 = U.optimize_sharded(ew, 500, 0.1, 4, 0.0, 5, 42, True, 1.0, 0, 2, h, 1, 0)
        assert abs(v1 - v2) <= 1e-9 * abs(v1) and np.abs(l1.to_numpy() - l2.to_numpy()).max() <= 1e-9
        src = S.STen.from_numpy(np.arange(12, dtype=np.int64).reshape(3, 4), 0)
        dst = S.STen.zeros([3, 4], S.I64, 0)
        lib.lamp_comm_all_gather(dst, src, h)
        assert np.array_equal(dst.to_numpy(), src.to_numpy())
        # the reassembly for several ranks, driven with the single-process pieces a 3-rank job would produce
        n, world, k = 500, 3, 5
        per = (n + world - 1) // world
        blocks = []
        for r in range(world):
            lo, hi = D.row_shard(n, world, r)
            q = max(min(lo, n - per), 0)
            blocks.append(U.knn_search(X, X.slice(0, q, q + per), k).to_numpy())
        gathered = np.concatenate(blocks, 0)
        parts = []
        for r in range(world):
            lo, hi = D.row_shard(n, world, r)
            q = max(min(lo, n - per), 0)
            parts.append(gathered[r * per + (lo - q): r * per + (lo - q) + (hi - lo)])
        assert np.array_equal(np.concatenate(parts, 0), ref)
    finally:
        lib.lamp_comm_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("accumulate", [1, 2])
def test_single_process_data_parallel_step(gpu, accumulate):
    """DataParallel.synchronousStep (lamp-data DataParallel.scala:195-311) with the main model and two replicas (all on the one GPU of
    the test box: same host threads, copies, weighting and reduction as with one GPU each): replicas receive the main state, the
    example-weighted mean gradient and the AdamW step equal the oracle's, the loss accumulators are per model."""
    from lamp_amd import nn, sten as S
    from tests.util import to_sten, to_torch, assert_close, closed_form
    dt = torch.float64
    def oracle_model(): return O.Sequential(O.mlp(12, 3, [8], dt), O.Fun(lambda v: v.logSoftMax(1)))
    def hip_model(): return nn.Sequential(nn.MLP(12, 3, [8], S.F64), nn.Fun("logsoftmax", 1))
    om = oracle_model()
    hmods = [hip_model() for _ in range(3)]
    hmods[0].load([to_sten(v.value) for v in om.state()])            # replicas start with different (random) weights
    cw = torch.ones(3, dtype=dt)
    models = [nn.SupervisedModel(m, nn.SupervisedModel.NLL, to_sten(cw)) for m in hmods]
    hopt = nn.AdamW([p.value for p in hmods[0].parameters], weightDecay=0.01, learningRate=1e-2)
    oopt = O.AdamW([p.value for p in om.parameters()], weightDecay=0.01, learningRate=1e-2)
    sizes = [5, 3, 7]
    accs = [S.STen.zeros([1], S.F64) for _ in range(3)]
    expected_acc = [0.0, 0.0, 0.0]
    for it in range(2 * accumulate):
        xs = [closed_form((n, 12), 100 * it + 10 * i, 2.0, dt) for i, n in enumerate(sizes)]
        ts = [(torch.arange(n) + i + it) % 3 for i, n in enumerate(sizes)]
        zero, step = it % accumulate == 0, it % accumulate == accumulate - 1
        total = nn.dataParallelSynchronousStep(models[0], hopt, models[1:], [(to_sten(x), to_sten(t)) for x, t in zip(xs, ts)], accs,
                                               zeroGrad=zero, step=step)
        assert total == sum(sizes)
        # oracle: every model holds the main state; gradients accumulate over `accumulate` batches per model, then the reference's
        # in-place `grad *= n` of the last batch, sum over models, / total of the last batch
        if zero:
            per_model = [None] * 3
        for i in range(3):
            loss = om.forward(O.const(xs[i])).nllLoss(ts[i], cw, 1, -100)
            expected_acc[i] += loss.value.item() * sizes[i]
            g = [x.clone() for x in om.gradients(loss)]
            per_model[i] = g if per_model[i] is None else [a + b for a, b in zip(per_model[i], g)]
        # the replicas now hold the state the main model had at the start of this step
        for r in hmods[1:]:
            for hv, ov in zip(r.parameters, om.parameters()):        # (the batch-norm running statistics move with every forward)
                assert_close(to_torch(hv.value), ov.value, 1e-12, "replica parameters = main parameters")
        if step:
            avg = [sum(per_model[i][k] * sizes[i] for i in range(3)) / sum(sizes) for k in range(len(per_model[0]))]
            oopt.step(avg, 1.0)
            for hv, ov in zip(hmods[0].parameters, om.parameters()):
                assert_close(to_torch(hv.value), ov.value, 1e-10, "main parameters after the step")
    for a, e in zip(accs, expected_acc):
        assert abs(float(a.to_numpy()[0]) - e) <= 1e-9 * abs(e)


@pytest.mark.gpu
def test_state_broadcast_and_schedule_factor_single_rank(gpu):
    """distributed `broadcast` (distributed/package.scala:683-688) as lamp_model_sync_state, and the schedule factor of the fused
    data-parallel step: with one rank the broadcast changes nothing (incl. AdamW's step count, which travels as state()[0]) and
    lamp_model_train_step_scheduled(comm, factor) lands on the parameters of gradients + optimizer.step(factor)."""
    from lamp_amd import sten as S, nn
    from lamp_amd._capi import lib
    B = 32
    x = S.STen.from_numpy((np.arange(B * 3 * 32 * 32) * 7919 % 1009 / 1009.0 - 0.5).reshape(B, 3, 32, 32).astype(np.float32), 0, S.F32)
    target = S.STen.from_numpy((np.arange(B) * 7 % 100).astype(np.int64), 0)
    uid = (C.c_uint8 * 128)(); lib.lamp_comm_get_unique_id(uid)
    comm = C.c_void_p(); lib.lamp_comm_init_rank(C.byref(comm), 1, uid, 0)
    n = C.c_int(0); lib.lamp_comm_count(comm, C.byref(n)); assert n.value == 1
    r = C.c_int(-1); lib.lamp_comm_user_rank(comm, C.byref(r)); assert r.value == 0

    def run(use_comm):
        lib.lamp_manual_seed(7)
        mod = nn.resnet(100, 0.0, S.F32, 0)
        model = nn.SupervisedModel(mod, nn.SupervisedModel.NLL, S.STen.ones([100], S.F32, 0))
        opt = nn.AdamW_factory(weightDecay=0.01, learningRate=1e-3)([p.value for p in mod.parameters])
        acc = S.STen.zeros([1], S.F64, 0)
        for k, f in enumerate((1.0, 0.5, 0.25)):
            if use_comm:
                if k == 1:
                    model.sync_state(opt, comm, 0)
                assert model.train_step(opt, x, target, acc, comm, f) == B
            else:
                _, g = model.addTotalLossAndReturnGradientsAndNumExamples(x, target, acc)
                opt.step(g, f)
        lib.lamp_device_synchronize()
        return [s.value.to_numpy() for s in mod.state], acc.to_numpy(), opt.state[0].to_numpy()

    plain, acc0, sc0 = run(False)
    dp, acc1, sc1 = run(True)
    lib.lamp_comm_destroy(comm)
    assert sc0 == sc1 == 3.0
    assert np.array_equal(acc0, acc1)
    for a, b in zip(plain, dp):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_allocator_defers_reuse_of_blocks_another_stream_still_uses(gpu):
    """A tensor allocated under the compute stream and consumed on a second stream (lamp's withOtherStream, the gradient exchange):
    after lamp_tensor_record_stream its block is not recycled while the other stream's work is pending.  Stress: the side stream is
    kept busy with a long dependent chain reading `src`; `src` is released and the compute stream immediately allocates and overwrites
    same-sized tensors; the side stream's result must still be computed from the original contents."""
    from lamp_amd import sten as S
    from lamp_amd._capi import lib
    side = C.c_void_p(); lib.lamp_stream_get_from_pool(1, 0, C.byref(side))
    cur = C.c_void_p(); lib.lamp_stream_get_current(0, C.byref(cur))
    d0 = C.c_int64(0); lib.lamp_allocator_deferred_frees(0, C.byref(d0))
    n = 1 << 22
    for trial in range(8):
        src = S.STen.full([n], float(trial + 1), S.F32, 0)
        lib.lamp_stream_wait_stream(side, cur)                       # the fill is complete before the side stream reads
        lib.lamp_tensor_record_stream(src, side)
        lib.lamp_stream_set_current(side)
        try:
            t = src * 1.0
            for _ in range(40):                                      # a long chain on the side stream, all reading src
                t = t * 0.5 + src * 0.5
        finally:
            lib.lamp_stream_set_current(cur)
        src.release()                                                # freed while the side stream still reads it
        junk = [S.STen.full([n], -1e30, S.F32, 0) for _ in range(4)]  # would land in src's block without the deferral
        lib.lamp_stream_synchronize(side)
        got = t.to_numpy()
        assert np.all(got == np.float32(trial + 1)), f"trial {trial}: the side stream read recycled memory ({got[:4]})"
        del junk
    d1 = C.c_int64(0); lib.lamp_allocator_deferred_frees(0, C.byref(d1))
    assert d1.value - d0.value >= 8
    lib.lamp_stream_release(side); lib.lamp_stream_release(cur)


@pytest.mark.gpu
def test_bench_multi_rank_line_verifies_itself(gpu, tmp_path):
    """the N-rank line must prove itself.  One GPU can drive the complete multi-rank code of bench.py with a
    1-rank RCCL communicator (LAMP_BENCH_FORCE_COMM=1): both exchange modes are timed (`value` = the faster, the other under
    `alt_mode`), the all-reduce of each is bracketed (`allreduce_us`), the averaged gradients of one ragged step are checked against
    the example-weighted mean of the all-gathered per-rank gradients (`grad_avg_max_rel_err`) and the replicas' state digests are
    compared (`replicas_identical`)."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               LAMP_BENCH_FORCE_COMM="1", LAMP_RDZV_FILE=str(tmp_path / "r.json"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                          "--batch", "256", "--min-window-s", "0.05"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["replicas_identical"] is True
    assert 0.0 <= line["grad_avg_max_rel_err"] <= 2.0 ** -7
    assert line["grad_avg_examples_per_rank"] == [256]
    assert set(line["allreduce_us"]) == {"graph_single_bucket", "eager_overlapped_two_buckets"}
    assert line["allreduce_us"]["graph_single_bucket"]["launches_per_step"] == 1
    assert line["allreduce_us"]["eager_overlapped_two_buckets"]["launches_per_step"] == 2
    assert all(v["avg_us"] > 0 for v in line["allreduce_us"].values())
    modes = {line["config"]["exchange_mode"].split(":")[0]} | {m["mode"].split(":")[0] for m in line["alt_mode"]}
    assert modes == {"graph_single_bucket", "eager_overlapped_two_buckets"}
    assert all(m["ms_per_step"] >= line["ms_per_step"] for m in line["alt_mode"]), "value is the faster mode"
