"""Multi-GPU layout of the restoration path: one process per GPU, the batch axis sharded across ranks, ONE bucketed
RCCL broadcast of the parameters at start-up, and no collective inside the denoise loop (every image's
encode -> sampler -> decode is independent: GroupNorm / LayerNorm / attention are per sample).

Mirrors what the reference gets from `accelerate` (DataLoaderConfiguration(split_batches=True),
main/cls/test_edtr.py:28, and the DDP parameter broadcast inside accelerator.prepare, main/det/test_edtr.py:95-96)."""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch


def shard_slice(rank: int, world: int, global_batch: int) -> slice:
    """Contiguous, balanced slice of the global batch owned by `rank` (first ranks take the remainder)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def bucketize(tensors: Iterable[torch.Tensor], bucket_bytes: int) -> List[List[torch.Tensor]]:
    """Group tensors (in order) into buckets of >= bucket_bytes: few large messages — an xGMI ring broadcast is
    bound per link (~153 GB/s), so per-message latency is what small tensors would pay 1300 times."""
    buckets, cur, size = [], [], 0
    for t in tensors:
        cur.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    return buckets


def broadcast_parameters(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 1 << 29) -> Tuple[int, int]:
    """Broadcast every parameter/buffer of `module` from rank `src` (torch.distributed must be initialised; backend
    'nccl' == RCCL on ROCm, 'gloo' in the CPU tests).  Returns (number of collectives, bytes moved)."""
    import torch.distributed as dist
    # the parameters themselves (not .data): the in-place copy must bump their version counters, which is what
    # params_fingerprint() watches to invalidate packed weights / engines built before the broadcast
    tensors = list(module.parameters()) + [b for b in module.buffers() if b is not None]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    calls = nbytes = 0
    with torch.no_grad():
        for group in by_dtype.values():
            for bucket in bucketize(group, bucket_bytes):
                flat = torch.cat([t.detach().reshape(-1) for t in bucket])
                dist.broadcast(flat, src=src)
                off = 0
                for t in bucket:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n
                calls += 1
                nbytes += flat.numel() * flat.element_size()
    w = getattr(module, "_weights", None)
    if w is not None and getattr(w, "frozen", None):
        w.frozen = None                                 # real parameters now: packing from them is legitimate again
    return calls, nbytes


def broadcast_packed(model, src: int = 0, bucket_bytes: int = 1 << 29) -> Tuple[int, int]:
    """Start-up weight distribution as SURVEY.md §8(e) specifies it: ONE bucketed broadcast of the PACKED 16-bit weight store
    (plus its small fp32 bias / norm vectors, alpha-scaled biases included) from rank `src`, in place.  Every rank has already
    built the same kernel programs (a warm-up pass packs whatever its parameters hold — placeholders on the receiving ranks),
    so the packed tensors exist at fixed addresses inside the launch records and hipGraphs; overwriting them in place needs
    no re-packing and no re-capture.  About half the bytes of the fp32 parameters and no per-rank packing work.

    The receiving ranks' fp32 parameters are still placeholders afterwards, so their store is FROZEN: a later cache miss (a new
    shape that needs an unpacked weight, `release_engines()` after changing control_scales / precision / compute_dtype) raises
    instead of silently re-packing from placeholders; `broadcast_parameters` lifts the freeze.  Returns (collectives, bytes)."""
    import torch.distributed as dist
    store = model._store()
    by_dtype = {}
    for t in store.tensors():                           # identical order on every rank (same programs -> same keys)
        by_dtype.setdefault(t.dtype, []).append(t)
    calls = nbytes = 0
    with torch.no_grad():
        for dt in sorted(by_dtype, key=str):
            for bucket in bucketize(by_dtype[dt], bucket_bytes):
                flat = torch.cat([t.reshape(-1) for t in bucket])
                dist.broadcast(flat, src=src)
                off = 0
                for t in bucket:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n
                calls += 1
                nbytes += flat.numel() * flat.element_size()
    if dist.get_rank() != src:
        store.frozen = f"packed weights were received from rank {src}"
    model.weights_updated_in_place()
    return calls, nbytes


def store_checksum(model) -> torch.Tensor:
    """fp64 [2] = (sum, sum of squares) over every tensor of the packed store, computed on the device in store order."""
    store = model._store()
    acc = None
    for t in store.tensors():
        v = t.detach().reshape(-1).to(torch.float64)
        part = torch.stack([v.sum(), (v * v).sum()])
        acc = part if acc is None else acc + part
    return acc


def verify_packed_store(model, src: int = 0) -> bool:
    """Every rank compares the checksum of its packed store with rank `src`'s and the verdicts are all-reduced (MIN): True on
    every rank only if every rank holds rank `src`'s weights.  bench.py refuses to report a multi-GPU throughput otherwise."""
    import torch.distributed as dist
    mine = store_checksum(model)
    ref = mine.clone()
    dist.broadcast(ref, src=src)
    ok = torch.tensor([1 if bool(torch.equal(mine, ref)) else 0], dtype=torch.int32, device=mine.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return bool(ok.item())
