"""N > 1 path on CPU: two gloo processes exercise exactly what bench.py does at start-up on RCCL — rank 0 owns the
checkpoint, one bucketed broadcast delivers it, the global batch is sliced per rank with no overlap."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edtr_amd import synth
    from edtr_amd.model import ControlLDM
    from edtr_amd.model.params import skip_init
    from edtr_amd.parallel import broadcast_parameters, shard_slice
    from edtr_amd.testing import synthetic_state_dicts
    cfg = synth.tiny_config()
    with skip_init():
        m = ControlLDM(**cfg)
    for p in m.parameters():
        p.data.fill_(float(rank) + 7.0)
    if rank == 0:
        sds = synthetic_state_dicts(cfg)
        m.unet.load_state_dict(sds["unet"])
        m.load_controlnet_from_ckpt(sds["controlnet"])
        m.vae.load_state_dict(sds["vae"])
    from edtr_amd.model.params import params_fingerprint
    fp_before = params_fingerprint(m.unet)
    calls, nbytes = broadcast_parameters(m, src=0, bucket_bytes=1 << 24)
    assert params_fingerprint(m.unet) != fp_before, "the broadcast must invalidate anything packed from the old weights"
    checksum = sum(float(p.double().sum()) for p in m.parameters())
    sl = shard_slice(rank, world, 6)
    noise = synth.synth_normal("dist:noise", (6, 4, 8, 8))[sl]
    gathered = [torch.zeros(3, 4, 8, 8) for _ in range(world)]
    dist.all_gather(gathered, noise.contiguous())
    q.put((rank, calls, nbytes, checksum, float(torch.cat(gathered).sub(synth.synth_normal("dist:noise", (6, 4, 8, 8))).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, calls0, bytes0, sum0, gap0), (r1, calls1, bytes1, sum1, gap1) = res
    assert (r0, r1) == (0, 1)
    assert calls0 == calls1 and calls0 < 40, "bucketed: a handful of collectives, not one per tensor"
    assert bytes0 == bytes1 > 50e6 * 4 * 0.9
    assert sum0 == sum1, "rank 1 must hold rank 0's weights bit for bit"
    assert gap0 == 0.0 and gap1 == 0.0, "per-rank slices tile the global batch"
