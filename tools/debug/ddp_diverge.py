import os, sys, random, socket, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    random.seed(5); torch.manual_seed(1234 + 17 * rank)
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    tr = Trainer(bucket_mb=0.05, **cfg["trainer"]).setup(task, dev)
    task.train()
    names = [n for n, p in task.named_parameters() if p.requires_grad]
    def cmp(tag, t):
        t = t.detach().clone()
        both = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(both, t)
        d = (both[0] - both[1]).abs()
        if rank == 0:
            bad = []
            st = tr.store
            for i, (o, n) in enumerate(zip(st.offsets, st.lengths)):
                m = float(d[o:o + n].max())
                if m > 0: bad.append((names[i] if i < len(names) else i, m))
            print(tag, "max diff", float(d.max()), "ntensors differing", len(bad), bad[:6], flush=True)
    cmp("init params", tr.store.flat_p)
    orig_step = tr.optimizer.step
    def step_dbg(*a, **k):
        torch.cuda.synchronize()
        cmp("grads before opt", tr.store.flat_g)
        r = orig_step(*a, **k)
        torch.cuda.synchronize()
        cmp("params after opt", tr.store.flat_p)
        for gi, s in enumerate(tr.optimizer._gstate):
            cmp(f"  param_rms g{gi}", s["param_rms"]); cmp(f"  fstate g{gi}", s["fstate"])
        cmp("  delta", tr.optimizer._delta); cmp("  eas", tr.optimizer._eas)
        return r
    tr.optimizer.step = step_dbg
    for i in range(4):
        batch = bench.make_batch(rank * 10 + i, 2, 2.0, 5, 64, dev)
        random.seed(100 + i); torch.manual_seed(200 + i)
        tr.training_step(batch, i)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2)
