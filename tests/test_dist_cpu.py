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


def _worker_packed(rank: int, world: int, port: int, q):
    """broadcast_packed: both ranks pack the same store entries (rank 1 from zeros), one bucketed broadcast overwrites rank 1's
    packed tensors IN PLACE (same addresses: launch records / hipGraphs built on them stay valid)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edtr_amd import synth
    from edtr_amd.model import ControlLDM
    from edtr_amd.model.params import skip_init
    from edtr_amd.parallel import broadcast_packed
    from edtr_amd.testing import synthetic_state_dicts
    cfg = synth.tiny_config()
    with skip_init():
        m = ControlLDM(**cfg)
    m.compute_dtype = torch.bfloat16
    for p in m.parameters():
        p.data.zero_()
    if rank == 0:
        sds = synthetic_state_dicts(cfg)
        m.unet.load_state_dict(sds["unet"])
        m.load_controlnet_from_ckpt(sds["controlnet"])
        m.vae.load_state_dict(sds["vae"])
    store = m._store()
    from edtr_amd.parallel import verify_packed_store
    w, b = store.conv("unet.input_blocks.1.0.in_layers.2.", cin_pad=64)
    w = w.get()
    wl, _ = store.linear(["unet.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight",
                          "unet.input_blocks.1.1.transformer_blocks.0.attn1.to_k.weight"])
    wl = wl.get()
    g = store.vec("vae.encoder.norm_out.weight")
    wg, bg = store.geglu("unet.input_blocks.1.1.transformer_blocks.0.ff.net.0.proj.weight",
                         "unet.input_blocks.1.1.transformer_blocks.0.ff.net.0.proj.bias")
    wg = wg.get()
    _, bz = store.conv("controlnet.zero_convs.0.0.", cin_pad=64, bias_scale=0.5)    # an alpha-scaled bias is a store tensor too
    ptrs = [t.data_ptr() for t in (w, b, wl, g, wg, bg, bz)]
    assert not verify_packed_store(m, src=0)            # before the broadcast rank 1 holds zeros: the check must see it
    calls, nbytes = broadcast_packed(m, src=0, bucket_bytes=1 << 20)
    assert ptrs == [t.data_ptr() for t in (w, b, wl, g, wg, bg, bz)]
    assert verify_packed_store(m, src=0)
    # the receiver's fp32 parameters are still placeholders: a cache miss must raise, never re-pack silently from them
    if rank == 1:
        import pytest
        with pytest.raises(RuntimeError, match="frozen"):
            store.vec("vae.decoder.norm_out.weight")
        with pytest.raises(RuntimeError, match="placeholders"):
            m.release_engines()
    else:
        store.vec("vae.decoder.norm_out.weight")
    q.put((rank, calls, nbytes, [float(t.double().abs().sum()) for t in (w, b, wl, g, wg, bg, bz)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_packed_broadcast_in_place():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, n0, s0), (_, c1, n1, s1) = res
    assert c0 == c1 >= 2 and n0 == n1 > 0          # one bucket per dtype (16-bit matrices, fp32 vectors)
    assert s0 == s1 and all(v > 0 for v in s0), "rank 1 must hold rank 0's packed weights bit for bit"


class _StubCldm(torch.nn.Module):
    """Identity restoration: prepare_condition hands the image through as the 'latent', vae_decode maps it back to [-1, 1]."""

    def __init__(self):
        super().__init__()
        self.unet = torch.nn.Linear(1, 1)

    def prepare_condition(self, clean, prompt):
        return {"c_txt": torch.zeros(clean.size(0), 77, 4), "c_img": clean}

    def vae_decode(self, z):
        return z * 2 - 1


class _StubDiffusion:
    def q_sample(self, x_start, t, noise):
        return x_start


class _StubSampler:
    def manual_sample_with_timesteps(self, model, device, x_T, **kw):
        return x_T


def _worker_dataset(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edtr_amd import evalutil, synth
    sizes = [(40, 56), (64, 64), (33, 47), (64, 20), (50, 50)]
    imgs = [synth.synth_input(f"ds:img{i}", (3, h, w), 0.0, 1.0) for i, (h, w) in enumerate(sizes)]
    gts = [(im + 0.03 * synth.synth_normal(f"ds:n{i}", tuple(im.shape))).clamp(0, 1) for i, im in enumerate(imgs)]
    outs, psnr = evalutil.restore_dataset(_StubCldm(), _StubDiffusion(), _StubSampler(), imgs, gts=gts, img_size=64, batch_size=2,
                                          colour_fix=False)
    q.put((rank, [tuple(o.shape) for o in outs], [float((o - imgs[i]).abs().max()) for o, i in zip(outs, range(*evalutil.shard_slice(rank, world, len(imgs)).indices(len(imgs))))],
           float(psnr)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_restore_dataset_sharding_and_metric():
    """evalutil.restore_dataset on 2 gloo ranks (stub networks: the restoration is the identity): the ragged image list is
    sharded 3 + 2 without overlap, padded / cropped per image, and the PSNR all-reduce gives every rank the dataset mean."""
    from edtr_amd import evalutil, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dataset, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [(40, 56), (64, 64), (33, 47), (64, 20), (50, 50)]
    (_, shp0, gap0, ps0), (_, shp1, gap1, ps1) = res
    assert shp0 == [(3, h, w) for h, w in sizes[:3]] and shp1 == [(3, h, w) for h, w in sizes[3:]]
    assert max(gap0 + gap1) == 0.0
    imgs = [synth.synth_input(f"ds:img{i}", (3, h, w), 0.0, 1.0) for i, (h, w) in enumerate(sizes)]
    gts = [(im + 0.03 * synth.synth_normal(f"ds:n{i}", tuple(im.shape))).clamp(0, 1) for i, im in enumerate(imgs)]
    want = sum(float(evalutil.calculate_psnr_pt(a[None], b[None], 0)[0]) for a, b in zip(imgs, gts)) / len(imgs)
    assert abs(ps0 - want) < 1e-6 and abs(ps1 - want) < 1e-6


def _run_bench_launcher(extra_env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EDTR_BENCH_BACKEND="gloo", **extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_self_launcher_starts_one_rank_per_gpu_and_forwards_rank0():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run must start two ranks itself (before any GPU call in the
    parent), rendezvous on 127.0.0.1, and print exactly rank 0's JSON line with rccl_ranks == 2.
    Dry-run mode: gloo, rendezvous + one all-reduce, no model."""
    import json
    r = _run_bench_launcher({})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dry_run"] is True


def test_bench_self_launcher_fails_when_a_rank_fails():
    r = _run_bench_launcher({"EDTR_BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not r.stdout.strip(), "no result line may be forwarded when a rank failed"


def test_bench_self_launcher_fails_fast_when_a_rank_dies_before_the_rendezvous():
    """a rank that exits BEFORE init_process_group leaves its peer waiting in the rendezvous for minutes; the
    launcher polls all children, kills the survivor and returns that rank's code at once."""
    import time
    t0 = time.time()
    r = _run_bench_launcher({"EDTR_BENCH_DRY_FAIL_EARLY_RANK": "1"})
    assert r.returncode == 4, (r.returncode, r.stderr[-1500:])
    assert not r.stdout.strip()
    assert "rank 1 exited with code 4" in r.stderr
    assert time.time() - t0 < 120, "the launcher waited for the surviving rank's rendezvous timeout"
