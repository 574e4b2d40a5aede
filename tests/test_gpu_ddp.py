"""GPU: two ranks sharing cuda:0 (gloo transport) run the full trainer path -- flat gradient
buffer, bucket hooks firing from the autograd thread, side HIP stream, scalar sync, flat
ScaledAdam -- and must stay bit-identical replicas that follow the same trajectory as a
single process fed the averaged gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import random
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from speech2text_amd.build_task import TaskFactory
    from speech2text_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    cfg = bench.c3_config(64)
    cfg["encoder"]["config"].update({"downsampling_factor": [1, 2], "num_encoder_layers": [1, 1],
                                     "feedforward_dim": [96, 128], "encoder_dim": [48, 64],
                                     "encoder_unmasked_dim": [32, 48], "num_heads": [4, 4],
                                     "query_head_dim": 8, "value_head_dim": 4, "pos_dim": 16,
                                     "cnn_module_kernel": [15, 7]})
    cfg["predictor"]["config"].update({"output_dim": 64, "symbol_embedding_dim": 32})
    cfg["joiner"].update({"input_dim": 64})
    cfg["trainer"]["accumulate_grad_batches"] = 2
    random.seed(5)
    torch.manual_seed(1234 + 17 * rank)            # different init: broadcast must fix it
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    tr = Trainer(bucket_mb=0.05, **cfg["trainer"]).setup(task, dev)
    task.train()
    nb = len(tr.reducer.buckets)
    losses = []
    for i in range(4):
        batch = bench.make_batch(rank * 10 + i, 2, 2.0, 5, 64, dev)
        random.seed(100 + i)                       # same python-random decisions on both ranks
        torch.manual_seed(200 + i)
        losses.append(float(tr.training_step(batch, i)))
    torch.cuda.synchronize()
    # by value (bytes), not as a torch tensor: a tensor travels as a file descriptor the parent has
    # to fetch from THIS process, which may already have exited by then (EOFError in the parent)
    flat = tr.store.p().detach().cpu().numpy().tobytes()
    q.put((rank, losses, flat, nb, {k: float(v) for k, v in task.logged.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_stay_identical(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    (r0, l0, f0, nb, lg0), (r1, l1, f1, _, lg1) = res
    assert nb >= 3
    assert all(x == x for x in l0 + l1)                       # finite
    assert f0 == f1 and len(f0) > 0, "replicas diverged"      # same params after 2 optimizer steps
    assert lg0.keys() == lg1.keys()
    for k in lg0:                                             # synced scalars are the rank mean
        assert abs(lg0[k] - lg1[k]) < 1e-5 * max(1.0, abs(lg0[k])), k


def _nccl_worker(port, q, algo, force):
    """One rank on cuda:0.  force=True: backend "nccl" (= RCCL) with world_size 1 and the reducer
    forced active -- hooks, reverse-order buckets, exchange stream ordered after the
    weight-gradient side stream, reduce-scatter / all-gather or all-reduce on device buffers, the
    packed scalar + late-flag all-reduce -- all of it a numerical no-op on one rank."""
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if force:
        os.environ["S2T_DDP_FORCE"] = "1"
        os.environ["S2T_DDP_ALGO"] = algo
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import bench
    from speech2text_amd.build_task import TaskFactory
    from speech2text_amd.trainer import Trainer
    cfg = bench.c3_config(64)
    cfg["encoder"]["config"].update({"downsampling_factor": [1, 2], "num_encoder_layers": [1, 1],
                                     "feedforward_dim": [96, 128], "encoder_dim": [48, 64],
                                     "encoder_unmasked_dim": [32, 48], "num_heads": [4, 4],
                                     "query_head_dim": 8, "value_head_dim": 4, "pos_dim": 16,
                                     "cnn_module_kernel": [15, 7]})
    cfg["predictor"]["config"].update({"output_dim": 64, "symbol_embedding_dim": 32})
    cfg["joiner"].update({"input_dim": 64})
    random.seed(5)
    torch.manual_seed(1234)
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    tr = Trainer(bucket_mb=0.05, **cfg["trainer"]).setup(task, dev)
    task.train()
    losses = []
    for i in range(4):
        batch = bench.make_batch(i, 2, 2.0, 5, 64, dev)
        random.seed(100 + i)
        torch.manual_seed(200 + i)
        losses.append(float(tr.training_step(batch, i)))
    torch.cuda.synchronize()
    info = (tr.reducer.active, len(tr.reducer.buckets), tr.reducer.poll_dropped(),
            len(tr.reducer._expected or {}), {k: float(v) for k, v in task.logged.items()})
    q.put((losses, tr.store.p().detach().cpu().numpy().tobytes(), info))
    if force:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_rccl_world1_reducer_is_a_noop(dev, algo):
    """RCCL really runs (backend nccl, device buffers, side streams) and changes nothing."""
    import numpy as np
    ctx = mp.get_context("spawn")
    out = []
    for force in (False, True):
        q = ctx.Queue()
        p = ctx.Process(target=_nccl_worker, args=(_free_port(), q, algo, force))
        p.start()
        out.append(q.get(timeout=300))
        p.join(timeout=60)
    (l0, f0, i0), (l1, f1, i1) = out
    assert i0[0] is False and i1[0] is True
    assert i1[1] >= 3 and i1[2] == 0 and i1[3] > 10, i1    # buckets, no dropped step, hooks fired
    a, b = np.frombuffer(f0, np.float32), np.frombuffer(f1, np.float32)
    # (not bitwise: the weight-gradient atomics sum in a run-dependent order, and the two
    # processes time their GEMM plans independently -- library kernel or ours per bucket)
    np.testing.assert_allclose(b, a, rtol=5e-4, atol=2e-5)
    np.testing.assert_allclose(l1, l0, rtol=1e-4)
    assert i0[4].keys() == i1[4].keys()


def _overlap_worker(port, q):
    """RCCL world 1, reducer forced active, the C3 model at the YAML dims (B = 4 x 10 s): which
    buckets go to the exchange stream from INSIDE backward once the per-parameter counts are learned?"""
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["S2T_DDP_FORCE"] = "1"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import bench
    from speech2text_amd import zip_native
    from speech2text_amd.build_task import TaskFactory
    from speech2text_amd.trainer import Trainer
    cfg = bench.c3_config(500)
    random.seed(1234)
    torch.manual_seed(1234)
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    tr = Trainer(**cfg["trainer"]).setup(task, dev)
    task.train()
    batch = bench.make_batch(0, 4, 10.0, 50, 500, dev)
    reports = []
    for i in range(5):
        tr.training_step(batch, i)
        reports.append(tr.reducer.overlap_report())
    torch.cuda.synchronize()
    q.put((reports, tr.reducer.poll_dropped(), list(zip_native.CALLS)))
    dist.barrier()
    dist.destroy_process_group()


def test_buckets_leave_during_backward_on_the_c3_model(dev):
    """DP overlap that one GPU can prove: with the layer executors (Python on the first step, native
    afterwards) telling the reducer which gradients are complete, all buckets but the last leave
    for the exchange stream from inside backward -- not from finish() -- from the second
    synchronised step on, and no step is dropped when the executor changes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_overlap_worker, args=(_free_port(), q))
    p.start()
    reports, dropped, native_calls = q.get(timeout=600)
    p.join(timeout=60)
    print("overlap reports:", reports)
    assert dropped == 0
    assert native_calls[1] >= 12 * 3                       # the native executor served the later steps
    nb = reports[0]["buckets"]
    assert nb >= 3
    assert reports[0]["launched_in_backward"] == 0         # the learning step defers everything
    for r in reports[2:]:
        assert r["launched_in_backward"] >= nb - 1, r
        assert r["first_launch_ms_before_finish"] > 0.0
