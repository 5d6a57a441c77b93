"""Parameter sharding schedule of the Wan full fine-tune (finetrainers_amd/wan/fsdp.py) on the CPU over gloo, world size 2: the gathered parameters are the
original ones, two rotating buffers serve the forward / backward order with the expected number of all-gathers, every unit's fp32 gradient is averaged and
lands on its owner, and accumulation over two micro-steps adds up."""
import os

import numpy as np
import torch
import torch.multiprocessing as mp


def _worker(rank, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from finetrainers_amd.wan.fsdp import ParameterSharder

    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        g = torch.Generator().manual_seed(0)
        sizes = [1000, 5000, 5000, 7000 + 37]  # root + three blocks; the last one is not a multiple of anything
        full = [torch.randn(n, generator=g).to(torch.bfloat16) for n in sizes]
        sh = ParameterSharder([t.clone() for t in full], ["root", "blocks.0", "blocks.1", "blocks.2"], 2, rank, "gloo")
        assert all(u.k % 64 == 0 and u.k * 2 >= n for u, n in zip(sh.units, sizes))
        ok = True
        for micro in range(2):
            ok &= torch.equal(sh.acquire(0), full[0])
            groot = sh.grad_buffer(0)
            for i in (1, 2, 3):  # forward order
                p = sh.acquire(i)
                sh.prefetch(i + 1)
                ok &= torch.equal(p, full[i])
            for i in (3, 2, 1):  # backward order
                p = sh.acquire(i)
                sh.prefetch(i - 1)
                ok &= torch.equal(p, full[i])
                gbuf = sh.grad_buffer(i)
                assert float(gbuf.abs().max()) == 0.0  # handed out zeroed
                gbuf.add_(full[i].float() * (rank + 1) * (micro + 1))
                gbuf.add_(1.0)  # a second kernel accumulating into the same buffer
                sh.scatter_grad(i)
            groot.add_(full[0].float() * (rank + 1) * (micro + 1))
            sh.scatter_grad(0)
            sh.finish_gradients()
            if micro == 0:
                gathers_first = sh.gathers_issued
                sh.release_all()
        # forward: root + 3 blocks; backward: blocks 3 and 2 are still resident, only block 1 is gathered again
        assert gathers_first == 5 and sh.scatters_issued == 8
        for u, t in zip(sh.units, full):
            lo, hi = rank * u.k, min((rank + 1) * u.k, u.numel)
            want = torch.zeros(u.k)
            if hi > lo:  # mean over the ranks of (rank + 1) = 1.5, micro-steps 1 + 2 = 3; the blocks also added 1.0 per micro-step
                want[: hi - lo] = t[lo:hi].float() * 1.5 * 3 + (0.0 if u.name == "root" else 2.0)
            ok &= torch.allclose(u.shard_grad, want, rtol=1e-6, atol=1e-6)
            ok &= torch.equal(u.shard[: max(hi - lo, 0)], t[lo:hi])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_parameter_sharder_schedule_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 97
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_parameter_sharder_single_rank_needs_no_copies():
    from finetrainers_amd.wan.fsdp import ParameterSharder

    full = [torch.randn(640).to(torch.bfloat16), torch.randn(1280).to(torch.bfloat16)]
    sh = ParameterSharder([t.clone() for t in full], ["root", "blocks.0"], 1, 0, "none")
    p = sh.acquire(1)
    assert p.data_ptr() == sh.units[1].shard.data_ptr() and torch.equal(p, full[1]) and sh.gathers_issued == 0
    g = sh.grad_buffer(1)
    g.add_(2.0)
    sh.scatter_grad(1)
    sh.finish_gradients()
    assert torch.equal(sh.units[1].shard_grad, torch.full((1280,), 2.0))
    _ = np  # (numpy is only here so that the workers never send torch storages through the queue)


def _ckpt_worker(rank, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from finetrainers_amd.parallel import DataParallelBackend
    from finetrainers_amd.wan import MI355XWanFullFinetuneStep, MI355XWanTransformer3DModel, WanTransformerConfig
    from oracle import wan

    par = DataParallelBackend(backend="gloo", device=torch.device("cpu"))
    try:
        kw = dict(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=64)
        sd = {k.replace("ffn.proj_in.", "ffn.net.0.proj.").replace("ffn.proj_out.", "ffn.net.2."): v for k, v in wan.build_model(wan.WanConfig(**kw), seed=0).state_dict().items()}
        model = MI355XWanTransformer3DModel(WanTransformerConfig(**kw), device=torch.device("cpu"))
        model.load_diffusers_state_dict(sd)
        if rank == 1:  # replicas start from rank 0's weights: a different local copy must not survive
            model.blocks[0].flat.data.zero_()
        step = MI355XWanFullFinetuneStep(model, parallel=par)
        total = model.blocks[0].layout.total
        ok = model.blocks[0].flat.numel() * 2 >= total and model.blocks[0].flat.numel() < total  # the module now holds half of the block
        try:
            model.state_dict_views()
            ok = False  # must refuse: the local tensors are slices
        except RuntimeError:
            pass
        got = step.gathered_state_dict()
        ok &= set(got) == set(sd)
        for k, v in got.items():
            ok &= bool(torch.equal(v.reshape(sd[k].shape), sd[k]))
        q.put((rank, bool(ok)))
    finally:
        par.destroy()


def test_sharded_wan_model_gathers_its_checkpoint_world2_gloo():
    """After ``MI355XWanFullFinetuneStep`` shards the parameters each module holds half of its unit; ``gathered_state_dict`` reassembles the diffusers-named
    tensors bit for bit on every rank (CPU tensors over gloo: the sharding / naming logic has no kernels in it)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 97
    procs = [ctx.Process(target=_ckpt_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
