"""Kernel-program machinery: device arena, packed-weight store, program (pre-built launch list with
hipGraph capture and per-launch timing) and the primitive emitters used by edtr_amd/nets.py.

Design (MI355X-first): a network evaluation at a fixed shape is compiled ONCE into a flat
list of libedtr_hip launch records whose buffers live at fixed addresses inside an arena (288 GB of HBM
makes a generous, rarely-freed arena the simplest correct allocator).  Steady state = a loop of ctypes
calls on one HIP stream, or one hipGraphLaunch when captured.  There is no tracing compiler and no
PyTorch operator on the data path.
"""
from __future__ import annotations

import collections
import ctypes as ct
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as L
from . import ops
from .ops import Rec, round_up


# ----------------------------------------------------------------------------------------------
# arena
# ----------------------------------------------------------------------------------------------
class Arena:
    """First-fit allocator over large device chunks.  Build-time only: a Program's launch order equals its
    build order on one stream, so a buffer released at build time may be handed to a later op safely."""

    ALIGN = 256

    def __init__(self, device: torch.device, chunk_bytes: int = 1 << 28):
        self.device = device
        self.chunk_bytes = chunk_bytes
        self.chunks: List[torch.Tensor] = []
        self.free_lists: List[List[List[int]]] = []   # per chunk: sorted [offset, size]
        self.live: Dict[int, Tuple[int, int, int, int]] = {}
        self.peak = 0
        self.in_use = 0

    def alloc(self, shape: Sequence[int], dtype: torch.dtype) -> torch.Tensor:
        numel = 1
        for s in shape:
            numel *= int(s)
        item = torch.empty((), dtype=dtype).element_size()
        nbytes = max(self.ALIGN, round_up(numel * item, self.ALIGN))
        for ci, fl in enumerate(self.free_lists):
            for fi, (off, size) in enumerate(fl):
                if size >= nbytes:
                    if size == nbytes:
                        fl.pop(fi)
                    else:
                        fl[fi] = [off + nbytes, size - nbytes]
                    return self._view(ci, off, nbytes, numel, shape, dtype)
        size = max(nbytes, self.chunk_bytes)
        self.chunks.append(torch.empty(size, dtype=torch.uint8, device=self.device))
        self.free_lists.append([[nbytes, size - nbytes]] if size > nbytes else [])
        return self._view(len(self.chunks) - 1, 0, nbytes, numel, shape, dtype)

    def _view(self, ci, off, nbytes, numel, shape, dtype):
        t = self.chunks[ci][off:off + nbytes].view(dtype)[:numel].view(*shape)
        self.live[t.data_ptr()] = (ci, off, nbytes, numel)
        self.in_use += nbytes
        self.peak = max(self.peak, self.in_use)
        return t

    def free(self, t: Optional[torch.Tensor]) -> None:
        if t is None:
            return
        key = t.data_ptr()
        if key not in self.live or self.live[key][3] != t.numel() or not t.is_contiguous():
            return  # a column-slice view / foreign tensor: the owner frees it
        ci, off, nbytes, _ = self.live.pop(key)
        self.in_use -= nbytes
        fl = self.free_lists[ci]
        fl.append([off, nbytes])
        fl.sort()
        merged: List[List[int]] = []
        for o, s in fl:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1][1] += s
            else:
                merged.append([o, s])
        self.free_lists[ci] = merged

    def total_bytes(self) -> int:
        return sum(c.numel() for c in self.chunks)


class EngineCache(collections.OrderedDict):
    """Shape-keyed engines (static buffers + launch programs + hipGraphs) with least-recently-used eviction: a dataset of
    many distinct image sizes must not pin one arena per size for ever.  ``EDTR_ENGINE_CACHE`` overrides the capacity."""

    def __init__(self, release=None, capacity: Optional[int] = None):
        super().__init__()
        self.capacity = max(1, capacity if capacity is not None else int(os.environ.get("EDTR_ENGINE_CACHE", "16")))
        self.release = release

    def fetch(self, key, build):
        if key in self:
            self.move_to_end(key)
            return self[key]
        eng = build()
        self[key] = eng
        while len(self) > self.capacity:
            old_key = next(iter(self))
            old = self.pop(old_key)
            if torch.cuda.is_available():
                torch.cuda.synchronize()          # nothing may still be replaying the evicted engine's graph
            if self.release is not None:
                self.release(old)
        return eng

    def drop_all(self) -> None:
        if len(self) and torch.cuda.is_available():
            torch.cuda.synchronize()              # a replay of one of these graphs may still be queued on another stream
        for eng in list(self.values()):
            if self.release is not None:
                self.release(eng)
        self.clear()


# ----------------------------------------------------------------------------------------------
# packed weights
# ----------------------------------------------------------------------------------------------
class WRef:
    """A packed matrix of the store, resolved per part count: ``get(parts)`` packs on first use (fast mode: parts is 1,
    high mode: always the bf16 split-3 form, mixed mode: fp16 with 1 / 2 / 3 parts as the precision policy asks).
    ``shape`` is the logical (one-part) [N, K]."""

    def __init__(self, store: "WeightStore", key: tuple, build, shape):
        self.store, self.key, self.build, self.shape = store, key, build, tuple(shape)

    def get(self, parts: int = 1) -> torch.Tensor:
        if self.store.dtype != ops.MIXED:
            parts = 3 if self.store.dtype == ops.F32S else 1
        key = self.key + (parts,)
        if key not in self.store.cache:
            self.store.cache[key] = self.build(parts)
        return self.store.cache[key]


class WeightStore:
    """fp32 parameters (reference names/shapes) -> device-resident packed 16-bit matrices + fp32 vectors.
    Packs lazily, caches per (kind, names, parts); ``invalidate()`` after the parameters change."""

    def __init__(self, params: Dict[str, torch.Tensor], dtype, device: torch.device):
        """``dtype``: torch.bfloat16 / torch.float16; ops.F32S for the high-precision mode (bf16 matrices whose K axis is
        the split [hi | hi | lo], three times as wide; see include/edtr_hip.h EDTR_F32_SPLIT); ops.MIXED for the mixed mode
        (fp16 matrices with 1..3 parts per WRef.get)."""
        self.params, self.dtype, self.device = params, dtype, device
        self.cache: Dict[tuple, object] = {}
        self._frozen: Optional[str] = None    # set on ranks whose fp32 parameters are placeholders (parallel.broadcast_packed)
        # the hybrid precision mode packs the SAME parameters in two storage formats (fp16 one-part matrices for the denoiser,
        # the mixed mode's multi-part matrices for the VAE): the formats are sibling stores under one root, so that everything
        # that walks "the" store — the packed broadcast, its checksum, the freeze — sees all of them
        self._root: Optional["WeightStore"] = None
        self._siblings: Dict[object, "WeightStore"] = {}

    def for_dtype(self, dtype) -> "WeightStore":
        """The store of the same parameters in another storage format (created on first use; ``self`` when it already is)."""
        root = self._root or self
        if dtype == self.dtype:
            return self
        if dtype == root.dtype:
            return root
        sib = root._siblings.get(dtype)
        if sib is None:
            sib = WeightStore(root.params, dtype, root.device)
            sib._root = root
            root._siblings[dtype] = sib
        return sib

    @property
    def frozen(self) -> Optional[str]:
        return (self._root or self)._frozen

    @frozen.setter
    def frozen(self, why: Optional[str]) -> None:
        (self._root or self)._frozen = why

    def invalidate(self):
        self.cache.clear()
        for sib in self._siblings.values():
            sib.cache.clear()

    def _p(self, name: str) -> torch.Tensor:
        if self.frozen:
            raise RuntimeError(f"weight store is frozen ({self.frozen}): packing {name!r} now would read placeholder parameters; "
                               "build every program before the packed broadcast, or broadcast the fp32 parameters")
        return self.params[name].detach().to(self.device, torch.float32)

    def shape_of(self, name: str):
        return tuple(self.params[name].shape)

    def vec(self, name: str, n_pad: Optional[int] = None, scale: float = 1.0) -> torch.Tensor:
        """fp32 vector (bias / norm affine), optionally pre-multiplied (a conv's alpha-scaled bias): cached, so that it is part
        of the packed store a multi-GPU start-up broadcasts."""
        key = ("vec", name, n_pad, float(scale))
        if key not in self.cache:
            v = self._p(name).reshape(-1)
            v = ops.pad_bias(v, n_pad or v.numel())
            self.cache[key] = (v * scale if scale != 1.0 else v).contiguous()
        return self.cache[key]

    def conv(self, prefix: str, cin_pad: Optional[int] = None, bias_scale: float = 1.0) -> Tuple[WRef, torch.Tensor]:
        """(WRef of [Np, taps*Cinp], bias f32 [Np]) for ``prefix + 'weight'/'bias'`` (3x3 or 1x1 conv)."""
        co, ci, kh, kw = self.shape_of(prefix + "weight")
        cip, cop = cin_pad or round_up(ci, 8), round_up(co, 8)
        ref = WRef(self, ("conv", prefix, cin_pad),
                   lambda parts: ops.pack_conv_weight(self._p(prefix + "weight"), self.dtype, cin_pad=cin_pad, parts=parts),
                   (cop, kh * kw * cip))
        return ref, self.vec(prefix + "bias", cop, bias_scale)

    def conv_subpixel(self, prefix: str, cin_pad: Optional[int] = None, bias_scale: float = 1.0) -> Tuple[WRef, torch.Tensor]:
        """(WRef of the four pre-summed phase matrices [4 * Np, 4 * Cinp], bias f32 [Np]) of a nearest-2x upsample convolution in
        its sub-pixel form (ops.pack_conv_weight_subpixel)."""
        co, ci, kh, kw = self.shape_of(prefix + "weight")
        cip, cop = cin_pad or round_up(ci, 8), round_up(co, 8)
        ref = WRef(self, ("conv_subpixel", prefix, cin_pad),
                   lambda parts: ops.pack_conv_weight_subpixel(self._p(prefix + "weight"), self.dtype, cin_pad=cin_pad, parts=parts),
                   (4 * cop, 4 * cip))
        return ref, self.vec(prefix + "bias", cop, bias_scale)

    def linear(self, names: Sequence[str], biases: Optional[Sequence[Optional[str]]] = None):
        """Row-concatenation of several [out, in] matrices (fused projections) + matching fp32 bias (or None)."""
        names = tuple(names)
        n_rows = [self.shape_of(n)[0] for n in names]
        k = 1
        for d in self.shape_of(names[0])[1:]:
            k *= d
        npad = round_up(sum(n_rows), 8)

        def build(parts):
            ws = [self._p(n).reshape(self.params[n].shape[0], -1) for n in names]
            return ops.pack_linear_weight(torch.cat(ws, dim=0), self.dtype, parts=parts)

        ref = WRef(self, ("linear", names), build, (npad, round_up(k, 8)))
        b = None
        if biases:
            key = ("linear.bias", names, tuple(biases))
            if key not in self.cache:
                parts_ = [self._p(bn).reshape(-1) if bn else torch.zeros(n_rows[i], device=self.device)
                          for i, bn in enumerate(biases)]
                self.cache[key] = ops.pad_bias(torch.cat(parts_), npad)
            b = self.cache[key]
        return ref, b

    def rows(self, name: str, r0: int, r1: int, bias: Optional[str] = None):
        """Rows r0..r1 of one [out, in] matrix (e.g. the q/k or the v part of a fused in_proj) + the matching bias slice."""
        shp = self.shape_of(name)
        k = 1
        for d in shp[1:]:
            k *= d
        npad = round_up(r1 - r0, 8)

        def build(parts):
            w = self._p(name)
            return ops.pack_linear_weight(w.reshape(w.shape[0], -1)[r0:r1].contiguous(), self.dtype, parts=parts)

        ref = WRef(self, ("rows", name, r0, r1), build, (npad, round_up(k, 8)))
        b = None
        if bias:
            key = ("rows.bias", name, r0, r1, bias)
            if key not in self.cache:
                self.cache[key] = ops.pad_bias(self._p(bias).reshape(-1)[r0:r1].contiguous(), npad)
            b = self.cache[key]
        return ref, b

    def ln_fold(self, kind: str, names: Sequence[str], biases, ln_prefix: str):
        """A projection with the LayerNorm in front of it folded in (include/edtr_hip.h: ln_stats): the packed matrix is
        W . gamma (columns scaled BEFORE the 16-bit rounding), c1[n] = sum_k of the PACKED values (so that the mean term cancels
        exactly what the MFMA multiplied), c2[n] = sum_k beta[k] W[n][k] in fp32.  ``kind``: "linear" (row concatenation of
        ``names``) or "geglu" (value / gate rows interleaved).  Fast modes only.  Returns (WRef, bias or None, c1, c2)."""
        if self.dtype in (ops.F32S, ops.MIXED):
            raise ValueError("the LayerNorm fold is a fast-mode optimisation (the parity modes keep the normalisation launch)")
        names = tuple(names)
        key = ("lnfold", kind, names, ln_prefix)
        if key not in self.cache:
            gamma, beta = self._p(ln_prefix + "weight").reshape(-1), self._p(ln_prefix + "bias").reshape(-1)
            w = torch.cat([self._p(n).reshape(self.params[n].shape[0], -1) for n in names], dim=0)
            b = None
            if biases:
                b = torch.cat([self._p(bn).reshape(-1) if bn else torch.zeros(self.params[n].shape[0], device=self.device)
                               for n, bn in zip(names, biases)])
            if kind == "geglu":
                perm = ops.geglu_perm(w.shape[0] // 2).to(self.device)
                w, b = w[perm], (b[perm] if b is not None else None)
            packed = ops.pack_linear_weight(w * gamma[None, :], self.dtype)
            npad = packed.shape[0]
            c1 = packed[:, : w.shape[1]].float().sum(dim=1).contiguous()
            c2 = ops.pad_bias(w @ beta, npad).contiguous()
            self.cache[key] = (packed, ops.pad_bias(b, npad) if b is not None else None, c1, c2)
        packed, b, c1, c2 = self.cache[key]
        return packed, b, c1, c2

    def ffn(self, w1name: str, b1name: str, w2name: str, b2name: str, ln_prefix: str):
        """Operands of edtr_ffn (include/edtr_hip.h) for one FeedForward: (w1 packed with the LayerNorm gamma folded in and value / gate
        rows interleaved, w2 with its columns permuted, the per-chunk constants, b2).  Fast modes only."""
        key = ("ffn", w1name, w2name, ln_prefix)
        if key not in self.cache:
            w1p, b1p, c1, c2 = self.ln_fold("geglu", [w1name], [b1name], ln_prefix)
            cst = ops.pack_ffn_constants(c2 + b1p)
            w2p = ops.pack_ffn_w2(self._p(w2name).reshape(self.params[w2name].shape[0], -1), self.dtype)
            self.cache[key] = (w1p, w2p, cst, self._p(b2name).reshape(-1).contiguous())
        return self.cache[key]

    def lin320(self, names: Sequence[str], biases, ln_prefix: Optional[str], alpha: float, vt_col0: Optional[int] = None, vt_alpha: float = 1.0):
        """Operands of edtr_lin320 (include/edtr_hip.h) for one K = 320 projection: (the matrix in fragment order — with the LayerNorm's
        gamma folded into its columns when ``ln_prefix`` is given — and the additive row bias + alpha W beta, or None; the rows from
        ``vt_col0`` on carry ``vt_alpha`` instead of ``alpha``: the V part of a fused [Wq; Wk; Wv]).  Fast modes only."""
        names = tuple(names)
        key = ("lin320", names, tuple(biases) if biases else None, ln_prefix, float(alpha), vt_col0, float(vt_alpha))
        if key not in self.cache:
            w = torch.cat([self._p(n).reshape(self.params[n].shape[0], -1) for n in names], dim=0)
            cvec = None
            if biases:
                cvec = torch.cat([self._p(bn).reshape(-1) if bn else torch.zeros(self.params[n].shape[0], device=self.device)
                                  for n, bn in zip(names, biases)]).float()
            if ln_prefix is not None:
                gamma, beta = self._p(ln_prefix + "weight").reshape(-1), self._p(ln_prefix + "bias").reshape(-1)
                shift = float(alpha) * (w @ beta)
                if vt_col0 is not None:
                    shift[vt_col0:] *= float(vt_alpha) / float(alpha)
                cvec = shift if cvec is None else cvec + shift
                w = w * gamma[None, :]
            self.cache[key] = (ops.pack_lin320_w(w, self.dtype), cvec.contiguous() if cvec is not None else None)
        return self.cache[key]

    def raw(self, name: str) -> torch.Tensor:
        """The fp32 parameter itself on the device (embedding tables)."""
        key = ("raw", name)
        if key not in self.cache:
            self.cache[key] = self._p(name).contiguous()
        return self.cache[key]

    def geglu(self, wname: str, bname: str):
        n, k = self.shape_of(wname)[0], self.shape_of(wname)[1]

        def build(parts):
            w = self._p(wname)
            perm = ops.geglu_perm(w.shape[0] // 2).to(self.device)
            return ops.pack_linear_weight(w[perm], self.dtype, parts=parts)

        ref = WRef(self, ("geglu", wname), build, (round_up(n, 8), round_up(k, 8)))
        key = ("geglu.bias", bname)
        if key not in self.cache:
            b = self._p(bname)
            self.cache[key] = b[ops.geglu_perm(b.shape[0] // 2).to(self.device)].contiguous()
        return ref, self.cache[key]

    def tensors(self) -> List[torch.Tensor]:
        """Every device tensor of the store in a deterministic order (same programs -> same keys on every rank)."""
        out: List[torch.Tensor] = []
        seen = set()
        stores = [self] + [self._siblings[k] for k in sorted(self._siblings, key=str)]      # (a root lists its sibling formats too)
        for st in stores:
            for key in sorted(st.cache, key=repr):
                val = st.cache[key]
                for t in (val if isinstance(val, (tuple, list)) else (val,)):
                    if isinstance(t, torch.Tensor) and t.data_ptr() not in seen:
                        seen.add(t.data_ptr())
                        out.append(t)
        return out


# ----------------------------------------------------------------------------------------------
# program
# ----------------------------------------------------------------------------------------------
class Program:
    """A flat launch list.  ``run()`` replays it on torch's current stream; ``capture()`` turns it into a
    hipGraph; ``run_timed()`` brackets every launch with events on the same stream.

    Launches may be tagged with a lane: between ``fork()`` and ``join()`` the lane-1 launches are independent of the
    lane-0 launches that follow them in the list (e.g. ControlNet vs the UNet encoder).  Eager replay ignores lanes
    (list order is a valid serial order); graph capture puts lane 1 on a second stream so the hipGraph gets two
    parallel branches and small kernels of one branch fill the CUs the other leaves idle."""

    def __init__(self, name: str):
        self.name = name
        self.recs: List[Rec] = []
        self.lanes: List[int] = []
        self.marks: Dict[int, str] = {}     # index into recs -> "fork" / "join" placed BEFORE that launch
        self.lane = 0
        self.graph = None
        self.sums_pool: Optional[torch.Tensor] = None   # GroupNorm [B][32][2] fp64 accumulators, zeroed by ONE launch
        self.sums_used = 0

    SUMS_SLOTS = 4096

    def sums_slot(self, arena: "Arena", B: int, count: int = 1) -> torch.Tensor:
        """A never-reused [B, 32, 2] fp64 GroupNorm accumulator ([count, B, 32, 2] for count > 1: the tiled VAE's per-tile
        sums of one GroupNorm) out of a pool that the program's FIRST launch zeroes (a hipMemsetAsync per edtr_gn_stats call
        costs two extra tiny kernels per node inside a hipGraph and far more in eager replay)."""
        if self.sums_pool is None:
            # Its OWN allocation, never arena memory: the pool is live from the program's first launch (which zeroes it), i.e.
            # earlier than the build-time point of this call — an arena hole freed by launches that precede the first
            # pooled GroupNorm would be scribbled over by them after the zeroing (found by the full-size batch-invariance
            # test: wrong results for batch >= 3, where the first GroupNorms are fused and the pool is created late).
            self.sums_pool = torch.zeros((self.SUMS_SLOTS, B, 32, 2), dtype=torch.float64, device=arena.device)
            rec = ops.make_zero(self.sums_pool, name="gn.zero_pool")
            self.recs.insert(0, rec)
            self.lanes.insert(0, 0)
            self.marks = {k + 1: v for k, v in self.marks.items()}
        if self.sums_used + count > self.SUMS_SLOTS or self.sums_pool.shape[1] != B:
            raise RuntimeError("GroupNorm accumulator pool exhausted")
        t = self.sums_pool[self.sums_used] if count == 1 else self.sums_pool[self.sums_used:self.sums_used + count]
        self.sums_used += count
        return t

    def add(self, rec: Rec) -> Rec:
        self.recs.append(rec)
        self.lanes.append(self.lane)
        return rec

    def duplicate_launches(self, substr: str, idempotent=("conv", "qk", "vT", "attn2.q", "geglu", "flash", "layernorm", ".apply",
                                                          "proj_in", "proj_out", "attn.out", "ff.out", "res.skip", "time_embed",
                                                          "emb_layers", "ctx_", "zero_conv")) -> int:
        """Measurement aid for tools / bench.py --dup (never used by the product path): every launch whose name contains
        ``substr`` AND is known to be idempotent (writes only its own output, no in-place residual, no atomics) is issued twice.
        The slowdown of a whole-path run is the MARGINAL wall-clock cost of that kernel class inside the overlapped hipGraph
        execution — what a per-launch event timing cannot show.  Call before capture(); returns the count."""
        if self.graph is not None:
            raise RuntimeError("duplicate_launches() must run before capture()")
        recs, lanes, marks, n = [], [], {}, 0
        for i, (r, ln) in enumerate(zip(self.recs, self.lanes)):
            if i in self.marks:
                marks[len(recs)] = self.marks[i]
            recs.append(r)
            lanes.append(ln)
            if substr in r.name and any(k in r.name for k in idempotent) and not r.name.endswith(".stats"):
                recs.append(r)
                lanes.append(ln)
                n += 1
        if len(self.recs) in self.marks:
            marks[len(recs)] = self.marks[len(self.recs)]
        self.recs, self.lanes, self.marks = recs, lanes, marks
        return n

    def fork(self) -> None:
        self.marks[len(self.recs)] = "fork"

    def join(self) -> None:
        self.marks[len(self.recs)] = "join"
        self.lane = 0

    def set_lane(self, lane: int) -> None:
        self.lane = lane

    def run(self) -> None:
        s = ops.stream_ptr()
        if self.graph is not None:
            L.check(L.load().edtr_graph_launch(self.graph, s), "graph_launch")
            return
        for r in self.recs:
            r.launch(s)

    def capture(self, parallel_lanes: bool = True) -> None:
        """Capture on side streams (hipGraph), then replay with one launch per run()."""
        lib = L.load()
        side = torch.cuda.Stream()
        side2 = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sp, sp2 = side.cuda_stream, side2.cuda_stream
            L.check(lib.edtr_graph_begin(sp), "graph_begin")
            forked = False
            try:
                for i, r in enumerate(self.recs):
                    mark = self.marks.get(i)
                    if mark == "fork" and parallel_lanes:
                        ev = torch.cuda.Event()
                        ev.record(side)
                        side2.wait_event(ev)          # side2 joins the capture
                        forked = True
                    elif mark == "join" and forked:
                        ev = torch.cuda.Event()
                        ev.record(side2)
                        side.wait_event(ev)
                        forked = False
                    r.launch(sp2 if (forked and self.lanes[i] == 1) else sp)
                if forked:
                    ev = torch.cuda.Event()
                    ev.record(side2)
                    side.wait_event(ev)
            finally:
                g = ct.c_void_p()
                code = lib.edtr_graph_end(sp, ct.byref(g))
            L.check(code, "graph_end")
        torch.cuda.current_stream().wait_stream(side)
        self.graph = g

    def release_graph(self) -> None:
        if self.graph is not None:
            L.load().edtr_graph_destroy(self.graph)
            self.graph = None

    def run_timed(self) -> List[Tuple[str, float, float, float]]:
        """[(name, ms, flops, bytes, shape tag)] per launch, HIP events on the launch stream."""
        s = ops.stream_ptr()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(self.recs) + 1)]
        evs[0].record()
        for i, r in enumerate(self.recs):
            r.launch(s)
            evs[i + 1].record()
        torch.cuda.synchronize()
        return [(r.name, evs[i].elapsed_time(evs[i + 1]), r.flops, r.bytes, r.tag) for i, r in enumerate(self.recs)]

    def total_flops(self) -> float:
        return sum(r.flops for r in self.recs)


# ----------------------------------------------------------------------------------------------
# activations + primitive emitters
# ----------------------------------------------------------------------------------------------
class OpN:
    """Multi-part GEMM operand of the high / mixed precision modes: 16-bit ``t`` [rows, parts*C] = [hi | lo | hi][:parts]
    of an fp32 [rows, C] activation (the output of a normalisation, or of an explicit edtr_split_operand launch).
    A consumer that wants fewer parts reads a prefix of the columns."""

    def __init__(self, t: torch.Tensor, C: int, parts: int):
        self.t, self.C, self.parts = t, C, parts

    def stride(self, dim: int) -> int:
        return self.t.stride(dim)


class LNReg:
    """A LayerNorm that is not launched: its raw input rows ``x`` [rows, C]; the consuming projection is an edtr_lin320 launch that
    normalises the rows in its registers (gamma / beta folded into its weights and additive row by the weight store)."""

    def __init__(self, x: torch.Tensor, C: int, prefix: str):
        self.x, self.C, self.prefix = x, C, prefix


class LNRef:
    """A LayerNorm that is not launched: its raw input rows ``x`` [rows, C] and the per-row statistics ``stats`` [rows, C / 32, 2]
    the producing GEMM wrote; the consuming GEMMs fold the normalisation into their weights and epilogue (edtr_hip.h ln_stats)."""

    def __init__(self, x: torch.Tensor, stats: torch.Tensor, C: int, prefix: str, c_valid: int = 0):
        self.x, self.stats, self.C, self.prefix, self.c_valid = x, stats, C, prefix, c_valid or C


@dataclass
class Act:
    """NHWC activation: ``t`` is a 2-D view [B*H*W, C] (row stride ``ld`` >= C) of 16-bit storage — fp32 storage, or an OpN
    (the output of a normalisation, which only ever feeds a GEMM) in the high / mixed precision modes."""
    t: object
    B: int
    H: int
    W: int
    C: int
    gnp: Optional[torch.Tensor] = None    # fused GroupNorm partials written by the producing igemm ([rows / gn_slot][C][2] fp32)
    # a GroupNorm (+ SiLU) that has NOT been applied: ``t`` is the raw tensor and ``gn_in`` the (scale, shift) table [B][C][2] the
    # consuming 3x3 convolution applies while it stages its operand (edtr_hip.h: a_gn).  ``owns``: False = ``t`` still belongs to
    # the Act it was derived from (Emitter.free releases the table only)
    gn_in: Optional[torch.Tensor] = None
    gn_silu: bool = True
    owns: bool = True
    # emits the ordinary apply launch of that GroupNorm and returns the normalised Act: what Emitter.conv falls back to when the
    # convolution that receives the deferred Act turns out not to be one the halo tiles take (the decision is conv()'s)
    gn_apply: Optional[object] = None
    gn_slot: int = 128                    # rows per slot of ``gnp`` (64: the split-K reducer's statistics of 8 x 8 images, ops.gn_slot_rows)

    @property
    def gn_tiles(self) -> int:
        """slots of ``gnp`` per image"""
        return (self.H * self.W) // self.gn_slot

    @property
    def ld(self) -> int:
        return self.t.stride(0)

    @property
    def rows(self) -> int:
        return self.B * self.H * self.W


class Emitter:
    """Primitive emitters.  Precision modes:
      "fast"  : 16-bit activation storage in ``dtype``, one product per GEMM.
      "high"  : the robust parity mode — fp32 activation stream, every GEMM / convolution as a bf16 split-3 product
                (3x the K, fp32 out), attention operands in fp16.
      "mixed" : the fast parity mode — the same fp32 stream, fp16 operands, and a per-layer part count (1 / 2 / 3 products)
                from ``policy`` (edtr_amd/precision.py)."""

    def __init__(self, prog: Program, arena: Arena, store: WeightStore, dtype: torch.dtype, precision: str = "fast",
                 policy=None):
        from .precision import ConstPolicy, mixed_policy
        self.prog, self.arena, self.store = prog, arena, store
        if precision not in ("fast", "high", "mixed"):
            raise ValueError(f"unknown precision mode {precision!r}")
        self.precision = precision
        self.hp = precision != "fast"                              # fp32 activation stream
        want = {"fast": dtype, "high": ops.F32S, "mixed": ops.MIXED}[precision]
        if store.dtype != want:
            raise ValueError("the weight store and the emitter must agree on the precision mode")
        if precision == "high":
            self.dtype, self.policy = torch.bfloat16, ConstPolicy(3)    # MFMA operand type of edtr_igemm
        elif precision == "mixed":
            self.dtype, self.policy = torch.float16, (policy or mixed_policy())
        else:
            self.dtype, self.policy = dtype, ConstPolicy(1)
        self.attn_dtype = torch.float16 if self.hp else dtype      # q / k / v^T / P of the attention kernels
        self.act_dtype = torch.float32 if self.hp else dtype       # storage of the activation stream
        # dtype code of the layout / elementwise / statistics launches (every fp32-stream code means the same to them)
        self.io = ops.F32S if self.hp else dtype
        self.direct16 = precision == "mixed"      # a 16-bit GEMM output IS an attention / one-part operand (same fp16 type)
        # EDTR_AMD_BATCH_INVARIANT=1 (read when a program is emitted): every launch choice that the default path derives from the
        # row count M = B * H * W — tile geometry, split-K, the fused-statistics eligibility of the register-staged tiles — is
        # derived from the layer's per-image shape instead (128 x 128 tiles, no split-K), so an image's result does not depend on
        # the batch it arrives in: bit-identical across batch sizes and therefore across any sharding over GPUs (VERDICT r02,
        # weak 3: the default path holds that only at tolerance level).  Costs throughput (§6 of DESIGN.md).
        self.invariant = ops.batch_invariant()
        # mixed mode: tensors INSIDE a residual branch whose only consumer rounds them to fp16 anyway (conv1 -> GroupNorm -> conv2,
        # ff.out -> proj_out) are stored as fp16, not fp32: only the residual stream and what feeds multi-part products stays wide
        # (VERDICT r03 item 1b).  EDTR_AMD_BRANCH16=0 restores the all-fp32 stream of round 3 (A/B runs).
        self.branch16 = self.direct16 and os.environ.get("EDTR_AMD_BRANCH16", "1") != "0"
        # mixed mode: an fp32 stream tensor that a ONE-part product reads (zero / down / upsample convolutions, the weights-exact 1x1
        # skips) gets an fp16 MIRROR written by its producer's epilogue (edtr_hip.h: out16) instead of a cast launch per consumer.
        # (root fp32 tensor, fp16 mirror) pairs; a producer that writes a column slice of a root writes the same slice of the mirror.
        self.mirrors_on = self.direct16 and os.environ.get("EDTR_AMD_MIRROR", "1") != "0"
        # high mode: the attention operands are hi + lo fp16 PAIRS cut from the fp32 projections and every attention product runs
        # as three MFMA products (edtr_hip.h: q_lo / k_lo / vt_lo): 0 = one fp16 part (rounds 1-3), 1 = q / k split, 2 = q / k and
        # p / v split, fp32 output.  The fp16 rounding of q and k alone cost 2.7e-3 of a denoiser evaluation on the heavy-tailed
        # weight set (tests/heavy_attention_budget.py); the mixed mode's projections write fp16 directly (no low part exists).
        # EDTR_AMD_ATTN_SPLIT overrides (the mixed mode then keeps its attention projections in fp32 and pays the split launches).
        pol_split = getattr(self.policy, "attn_split", None)
        self.attn_split = int(os.environ.get("EDTR_AMD_ATTN_SPLIT", "2" if precision == "high" else str(pol_split or 0))) if self.hp else 0
        self._mirrors: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self.last_gnp = None
        self.last_gn_slot = 128
        self.last_row_stats = None
        self.gnp_into_done = False       # did the last gemm / conv write its GroupNorm partials into the caller's shared slot buffer?
        self.add_stats_done = False      # ... and the last add()

    # -- precision plumbing ---------------------------------------------------------------------
    def parts_for(self, name: str, M: int = 0, N: int = 0, K: int = 0) -> int:
        return self.policy.parts(name, M, N, K) if self.hp else 1

    def feeds_parts(self, feeds, M: int = 0) -> int:
        """Part count a normalisation must write for the GEMM classes it feeds (the widest of them)."""
        if not self.hp:
            return 1
        if not feeds:
            return 3
        # (a class on the weights-exact two-part product reads ONE activation part — but only as a plain GEMM with whole 64-column
        #  K-tiles: a 3 x 3 convolution, or a width that is no multiple of 64, falls back to three parts AFTER this operand was
        #  written (EDTR_AMD_POLICY={"default": 4} died on it in round 5).  Three parts serve both: a consumer reads a prefix.)
        return max(3 if self.parts_for(f, M) == ops.PARTS_2W else ops.op_parts(self.parts_for(f, M)) for f in feeds)

    def op_fmt(self, parts: int):
        """dtype code under which a norm / split launch writes a ``parts``-part operand."""
        return ops.F32S if self.precision == "high" else ops.F32H[parts]

    # -- memory
    def new(self, rows: int, cols: int, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        return self.arena.alloc((rows, cols), dtype or self.act_dtype)

    def free(self, *ts) -> None:
        for t in ts:
            if isinstance(t, Act):
                if t.gn_in is not None:
                    self.arena.free(t.gn_in)
                    if not t.owns:
                        continue
                self.arena.free(t.gnp)
                t = t.t
            if isinstance(t, OpN):
                t = t.t
            if isinstance(t, LNReg):
                continue             # (owns nothing: the raw rows belong to the residual stream)
            if isinstance(t, LNRef):
                t = t.stats          # (the raw rows belong to the residual stream: their owner frees them)
            self._drop_mirror(t)
            self.arena.free(t)

    # -- fp16 mirrors of fp32 stream tensors (mixed mode) -----------------------------------------
    def want_mirror(self, consumer: str) -> bool:
        """Does the GEMM class ``consumer`` read its fp32 input as ONE fp16 part (so that a mirror written by the producer saves the
        cast launch)?"""
        return self.mirrors_on and ops.op_parts(self.parts_for(consumer)) == 1

    def add_mirror(self, root: torch.Tensor) -> torch.Tensor:
        m = self.arena.alloc(tuple(root.shape), self.dtype)
        self._mirrors.append((root, m))
        return m

    def new_stream(self, rows: int, cols: int, mirror: bool = False) -> torch.Tensor:
        """A stream buffer that several producers fill by column slices (the decoder's concat); ``mirror``: with its fp16 mirror."""
        t = self.new(rows, cols)
        if mirror and self.mirrors_on and t.dtype == torch.float32:
            self.add_mirror(t)
        return t

    def mirror_view(self, v) -> Optional[torch.Tensor]:
        """The fp16 mirror of fp32 tensor / column-slice view ``v`` (None when its root has none)."""
        if not self._mirrors or not isinstance(v, torch.Tensor) or v.dtype != torch.float32 or v.dim() != 2:
            return None
        for root, m in self._mirrors:
            off = v.data_ptr() - root.data_ptr()
            if 0 <= off < root.numel() * 4 and v.stride(0) == root.stride(0) and v.stride(1) == 1:
                row, col = divmod(off // 4, root.stride(0))
                if row + v.shape[0] <= root.shape[0] and col + v.shape[1] <= root.shape[1]:
                    return m[row:row + v.shape[0], col:col + v.shape[1]]
        return None

    def _drop_mirror(self, t) -> None:
        if not self._mirrors or not isinstance(t, torch.Tensor):
            return
        for i, (root, m) in enumerate(self._mirrors):
            if root.data_ptr() == t.data_ptr() and root.numel() == t.numel():
                self.arena.free(m)
                del self._mirrors[i]
                return

    # -- multi-part operands --------------------------------------------------------------------
    def _operand(self, a, rows: int, C: int, parts: int):
        """(16-bit operand tensor, temporary to free or None, parts actually used) of an fp32 / 16-bit activation or of a
        ready OpN."""
        ap = ops.op_parts(parts)          # (parts == ops.PARTS_2W: the weights-exact form reads ONE activation part twice)
        if isinstance(a, OpN):
            if a.C != C or a.parts < ap:
                raise ValueError(f"operand has {a.parts} part(s) of {a.C} columns, the GEMM wants {ap} of {C}")
            return a.t, None, parts
        if a.dtype == self.dtype:
            # already the MFMA operand type (mixed mode: an fp16 attention / GEGLU output): its low part is exactly zero, so
            # more activation parts buy nothing — the one-part product (or the weights-exact one)
            return a, None, (parts if parts == ops.PARTS_2W else 1)
        if ap == 1:
            m = self.mirror_view(a)       # the producer's epilogue already wrote the fp16 copy
            if m is not None:
                return m, None, parts
        t = self.arena.alloc((rows, ap * C), self.dtype)
        self.prog.add(ops.make_split_operand(src=a, rows=rows, C=C, dst=t, fmt=self.op_fmt(ap)))
        return t, t, parts

    def to16(self, x: torch.Tensor, rows: int, C: int, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """fp32 stream: a 16-bit copy (attention operand) of an fp32 [rows, C] view."""
        y = self.arena.alloc((rows, C), dtype or self.attn_dtype)
        self.prog.add(ops.make_cast16(dtype=dtype or self.attn_dtype, src=x, rows=rows, C=C, dst=y))
        return y

    @staticmethod
    def _w(w, parts: int) -> torch.Tensor:
        return w.get(parts) if isinstance(w, WRef) else w

    # -- GEMM family ------------------------------------------------------------------------
    def ln_fold_ok(self, C: int, B: int = 0) -> bool:
        """May the LayerNorms (over C columns) of a transformer block that runs on a batch of B images be folded into the GEMMs
        around them (fast modes)?  Built for VERDICT r02 item 4 (launch count): it removes up to 69 launches per denoise step and
        one 16-bit rounding.  With the general epilogue loop it lost 0.7 % at batch 8 and was opt-in; since the producer /
        consumer sides have their own specialised row loops (round 3) it gains where the LayerNorm launches are latency-bound —
        det512s50 (batch 4): **+1.1 %** with every level folded, +0.1 % with the 32x32-and-deeper levels only — and still loses
        where the GEMM epilogues are the longer pole — det512 (batch 8): -1.3 % all levels, -0.8 % / -0.4 % with the levels of
        <= 8192 / <= 2048 rows only (profiles/r03/ab_lnfold_fast_epilogue.log, ab_lnfold_thresholds.log).  So the rule is the
        batch, not the row count: on by default for B <= ops.LN_FOLD_MAX_BATCH.  EDTR_LN_FOLD = 1 / 0 forces it on / off."""
        if self.hp or C % 32 or self.invariant:
            return False
        mode = os.environ.get("EDTR_LN_FOLD", "auto")
        if mode in ("0", "1"):
            return mode == "1"
        return 0 < B <= ops.LN_FOLD_MAX_BATCH

    def _ln_kwargs(self, a, K: int, ln_vec):
        """igemm arguments of a folded LayerNorm for operand ``a`` (an LNRef) -> (raw rows, extra keyword arguments)."""
        if a.C != K or ln_vec is None:
            raise ValueError("folded LayerNorm: the consumer needs K == C and the folded weight's (c1, c2)")
        return a.x, dict(ln_stats=a.stats, ln_C=a.C, ln_valid=a.c_valid, ln_eps=1e-5, ln_c1=ln_vec[0], ln_c2=ln_vec[1])

    def ffn_ok(self, rows: int, C: int) -> bool:
        """May x + ff(norm3(x)) of a transformer block run as ONE edtr_ffn launch?  (fast modes; the shapes edtr_ffn is built for)"""
        return not self.hp and not self.invariant and ops.ffn_ok(rows, C, 4 * C)

    def lin320_ok(self, rows: int, N: int, K: int, ln: bool = False) -> bool:
        """May a K = 320 projection (optionally behind its LayerNorm) run as ONE edtr_lin320 launch?  (fast modes; the shapes it is built for)"""
        return not self.hp and not self.invariant and ops.lin320_ok(rows, N, K, ln)

    def lin320(self, x: torch.Tensor, rows: int, N: int, names, biases=None, *, ln_prefix: Optional[str] = None, alpha: float = 1.0,
               residual: Optional[torch.Tensor] = None, gn_table: Optional[torch.Tensor] = None, rows_per_image: int = 0,
               name: str = "lin320") -> torch.Tensor:
        """out = alpha LayerNorm?(x) W^T + bias (+ residual) on raw 16-bit rows (edtr_lin320: the rows live in registers, the LayerNorm is
        applied there — no normalisation launch, no normalised tensor)."""
        self.last_gnp, self.last_row_stats, self.gnp_into_done = None, None, False
        w, cvec = self.store.lin320(names, biases, ln_prefix, alpha)
        out = self.new(rows, N)
        self.prog.add(ops.make_lin320(dtype=self.dtype, x=x, ldx=x.stride(0), M=rows, N=N, w=w, cvec=cvec, alpha=alpha, ln=ln_prefix is not None,
                                      eps=1e-5, residual=residual, ldr=residual.stride(0) if residual is not None else 0, out=out,
                                      ldo=out.stride(0), gn_table=gn_table, rows_per_image=rows_per_image, name=name))
        return out

    def qkv_lin320(self, x: torch.Tensor, names, *, B: int, N: int, C: int, ln_prefix: str, alpha: float, name: str = "attn1.qkv"):
        """qkv_gemm on the RAW rows with norm1 applied in the launch's registers (edtr_lin320 with its transposed V part): (qk [B*N, 2C]
        row-major with ``alpha`` applied, v^T [B*C, N], its row stride)."""
        M = B * N
        w, cvec = self.store.lin320(names, None, ln_prefix, alpha, vt_col0=2 * C, vt_alpha=1.0)
        qk = self.arena.alloc((M, 2 * C), self.attn_dtype)
        vt = self.arena.alloc((B * C, N), self.attn_dtype)
        self.prog.add(ops.make_lin320(dtype=self.dtype, x=x, ldx=x.stride(0), M=M, N=3 * C, w=w, cvec=cvec, alpha=alpha, ln=True, eps=1e-5, out=qk,
                                      ldo=2 * C, vt_out=vt, vt_col0=2 * C, vt_ld=N, vt_alpha=1.0, rows_per_image=N, name=name))
        return qk, vt, N

    def ffn(self, x: torch.Tensor, rows: int, C: int, tb: str, name: str = "ff.fused") -> torch.Tensor:
        """out = x + W2 GEGLU(W1 LayerNorm(x) + b1) + b2 on the RAW rows x (model/attention.py:233) as one launch."""
        w1, w2, cst, b2 = self.store.ffn(tb + "ff.net.0.proj.weight", tb + "ff.net.0.proj.bias", tb + "ff.net.2.weight", tb + "ff.net.2.bias",
                                         tb + "norm3.")
        out = self.new(rows, C)
        self.prog.add(ops.make_ffn(dtype=self.dtype, x=x, ldx=x.stride(0), M=rows, w1=w1, w2=w2, cst=cst, b2=b2, out=out, ldo=out.stride(0),
                                   name=name))
        self.last_gnp = None
        self.last_row_stats = None
        return out

    def _gnp_into_args(self, gnp_into, M: int, N: int, K: int, hw: int, splitk: int, C2: int = 0):
        """A producer that writes a COLUMN SLICE of a concatenation can still hand the concatenation's GroupNorm its statistics: the
        two halves share one buffer of slots (``gnp_into`` = (buffer [slots][Ctot][2], first column, rows per slot); edtr_hip.h:
        gn_ld).  Returns the extra edtr_igemm arguments, or None where this launch cannot write them (the caller then keeps the
        edtr_gn_stats pass over the concatenated tensor).  Fast modes only."""
        if gnp_into is None or self.hp:
            return None
        buf, col, slot = gnp_into
        want = ops.gn_slot_rows(hw) if splitk > 1 else 128
        if want != slot or not ops.gn_fusable(M, N, K, hw, splitk=splitk, C2=C2, invariant=self.invariant):
            return None
        return dict(gn_partial=buf.reshape(-1)[2 * col:], gn_ld=buf.shape[1], gn_slot_rows=slot if splitk > 1 else 0)

    def gemm(self, a, w, M: int, N: int, K: int, *, bias=None, out=None, act=0,
             residual=None, rowvec=None, rows_per_image=0, out_f32=False, alpha=1.0, name="linear", stats_hw=0,
             out16=False, feeds=None, row_stats=False, ln_vec=None, mirror=False, gnp_into=None, **kw) -> torch.Tensor:
        """out[M, N'] = epilogue(a[M, K] @ w[N, K]^T).  ``a``/``out``/``residual`` are 2-D views (row stride = ld); ``w`` is a
        WRef of the store (or a ready packed tensor).  fp32-stream modes: the output is fp32 unless ``out16`` (an attention
        operand, mixed mode only) or ``feeds`` names a GEMM class that takes it as a one-part operand."""
        n_out = N // 2 if act == L.ACT_GEGLU else N
        self.last_gnp = None     # fused GroupNorm partials of this output (stats_hw = pixels per image), if eligible
        self.last_gn_slot = 128  # ... and the rows each of their slots covers
        self.gnp_into_done = False   # ... or whether they went into the caller's shared buffer (gnp_into)
        self.last_row_stats = None   # per-row statistics of this output for a LayerNorm folded into the next GEMMs (row_stats=True)
        parts = self.parts_for(name, M, N, K)
        if parts == ops.PARTS_2W and (K % 64 or "Z" in kw or isinstance(a, LNRef) or kw.get("C2", 0)):
            parts = 3           # (the weights-exact form needs whole 64-column K-tiles of a plain GEMM)
        tmp = None
        if isinstance(a, LNRef):
            a, lnkw = self._ln_kwargs(a, K, ln_vec)
            kw.update(lnkw)
            kw["tile"] = 0 if "tile" not in kw else kw["tile"]
        if self.hp:
            a, tmp, parts = self._operand(a, M, K, parts)
            if out is not None:
                out_f32 = out.dtype == torch.float32
            else:
                want16 = self.direct16 and (out16 or (feeds is not None and ops.op_parts(self.parts_for(feeds, M)) == 1))
                out_f32 = not want16
        m16 = None
        if out is None:
            out = self.new(M, n_out, torch.float32 if out_f32 else (self.dtype if self.hp else None))
            if mirror and self.mirrors_on and out_f32:
                m16 = self.add_mirror(out)
        elif self.mirrors_on and out_f32:
            m16 = self.mirror_view(out)       # a column slice of a mirrored stream buffer
        if m16 is not None:
            kw["out16"] = m16
        wt = self._w(w, parts)
        Ke = ops.k_mult(parts) * K
        if parts == ops.PARTS_2W:
            kw["a_wrap"] = K
        if "tile" in kw:
            tile, splitk = kw.pop("tile"), 1
        else:
            tile, splitk = ops.choose_splitk(M, N, Ke, kw.get("Z", 1), act)
        if self.invariant and tile == 0:
            tile, splitk = ops.invariant_tile(Ke, kw.get("C2", 0)), 1
        if row_stats and not self.hp and splitk == 1 and act != L.ACT_GEGLU and N % 32 == 0 and "Z" not in kw:
            self.last_row_stats = self.arena.alloc((M, N // 32, 2), torch.float32)
            kw["row_stats"] = self.last_row_stats
        ws = self.arena.alloc((splitk * M * N,), torch.float32) if splitk > 1 else None
        if (stats_hw and act == 0 and (self.hp or not out_f32) and out.stride(0) == N and "Z" not in kw
                and ops.gn_fusable(M, N, Ke, stats_hw, splitk=splitk, C2=kw.get("C2", 0), invariant=self.invariant)):
            self.last_gn_slot = ops.gn_slot_rows(stats_hw) if splitk > 1 else 128
            self.last_gnp = self.arena.alloc((M // self.last_gn_slot, N, 2), torch.float32)
            kw["gn_partial"] = self.last_gnp
            kw["gn_slot_rows"] = self.last_gn_slot if splitk > 1 else 0
        elif gnp_into is not None and stats_hw and act == 0 and not out_f32 and "Z" not in kw:
            into = self._gnp_into_args(gnp_into, M, N, Ke, stats_hw, splitk, kw.get("C2", 0))
            if into is not None:
                kw.update(into)
                self.gnp_into_done = True
        res32 = residual is not None and residual.dtype == torch.float32
        self.prog.add(ops.make_igemm(
            dtype=self.dtype, a1=a, w=wt, out=out, M=M, N=N, C1=Ke, ld1=a.stride(0), ldw=wt.stride(0), ldc=out.stride(0),
            bias_n=bias, act=act, residual=residual, ldr=residual.stride(0) if residual is not None else 0, residual_f32=res32,
            rowvec=rowvec, rowvec_ld=rowvec.stride(0) if rowvec is not None else 0, rows_per_image=rows_per_image,
            out_f32=out_f32, alpha=alpha, name=name, tile=tile, splitk=splitk, workspace=ws, **kw))
        self.arena.free(ws)
        self.arena.free(tmp)
        return out

    def fused_qkv_ok(self, N: int, C: int) -> bool:
        """Can a self-attention's q / k / v^T come out of ONE edtr_igemm launch (transposed second output)?  Needs a 16-bit
        output of the attention operand type (not the high mode, whose MFMA type is bf16 while attention runs on fp16) and V
        columns that start on a column-tile boundary (every SD width: 2C is a multiple of 160 or 128)."""
        if self.hp and (not self.direct16 or self.attn_split):
            return False        # (split attention operands are cut from fp32 projections)
        if self.invariant and (2 * C) % 128:
            return False        # (the invariant mode runs the 128-column tiles only)
        return N % 8 == 0 and C % 64 == 0 and ((2 * C) % 160 == 0 or (2 * C) % 128 == 0) and os.environ.get("EDTR_FUSED_QKV", "1") != "0"

    def qkv_gemm(self, x, w, *, B: int, N: int, C: int, alpha: float, name: str = "attn1.qkv", ln_vec=None):
        """[Q; K; V] = x @ [Wq; Wk; Wv]^T in one launch: (qk [B*N, 2C] row-major with `alpha` applied, v^T [B*C, N] transposed
        by the epilogue, unscaled).  reference model/attention.py:170-178 (to_q / to_k / to_v + the head rearranges)."""
        M = B * N
        parts = self.parts_for(name, M, 3 * C, C)
        parts = 3 if parts == ops.PARTS_2W else parts
        tmp, lnkw = None, {}
        if isinstance(x, LNRef):
            x, lnkw = self._ln_kwargs(x, C, ln_vec)
        if self.hp:
            x, tmp, parts = self._operand(x, M, C, parts)
        wt = self._w(w, parts)
        qk = self.arena.alloc((M, 2 * C), self.attn_dtype)
        vt = self.arena.alloc((B * C, N), self.attn_dtype)
        self.prog.add(ops.make_igemm(dtype=self.dtype, a1=x, w=wt, out=qk, M=M, N=3 * C, C1=parts * C, ld1=x.stride(0),
                                     ldw=wt.stride(0), ldc=2 * C, alpha=alpha, rows_per_image=N, vt_out=vt, vt_col0=2 * C, vt_ld=N,
                                     vt_alpha=1.0, name=name, **lnkw))
        self.arena.free(tmp)
        return qk, vt, N

    def conv(self, x: Act, prefix: str, *, taps=9, stride=1, pad_tl=1, ups=False, rowvec=None, residual=None,
             out=None, out_f32=False, alpha=1.0, name=None, stats=False, feeds=None, mirror=False, gnp_into=None) -> Act:
        """3x3 (or 1x1) convolution of an NHWC activation with the packed weight ``prefix``.  ``feeds`` (mixed mode): the output is
        BRANCH-INTERNAL — its only consumer is a normalisation that feeds the named one-part GEMM classes, which round it to fp16
        anyway — so it is stored as fp16 instead of joining the fp32 stream (Emitter.branch16)."""
        name = name or ("conv3x3" if taps == 9 else "conv1x1")
        w, bias = self.store.conv(prefix, cin_pad=x.C, bias_scale=alpha)      # the epilogue applies alpha before the bias
        N = w.shape[0]
        if taps == 9:
            LH, LW = (x.H * 2, x.W * 2) if ups else (x.H, x.W)
            pad_br = 1  # bottom/right halo always exists (zero); only the top/left pad differs (VAE downsample)
            OH = (LH + pad_tl + pad_br - 3) // stride + 1
            OW = (LW + pad_tl + pad_br - 3) // stride + 1
            spatial = (x.H, x.W, OH, OW, stride, pad_tl, pad_tl, int(ups))
        else:
            OH, OW, spatial = x.H, x.W, None
        M = x.B * OH * OW
        parts = self.parts_for(name, M, N, taps * x.C)
        if parts == ops.PARTS_2W and (taps != 1 or x.C % 64):
            parts = 3           # (the weights-exact form exists for 1x1 convolutions = plain GEMMs)
        if x.gn_in is not None:          # a deferred GroupNorm: is this one of the convolutions that apply it while staging their operand?
            img8_ = taps == 9 and stride == 1 and pad_tl == 1 and not ups and (x.H, x.W) == (8, 8) and x.C % 64 == 0
            fusable = (taps == 9 and stride == 1 and pad_tl == 1 and not ups and not self.invariant
                       and (not self.hp or (x.t.dtype == self.dtype and parts == 1))
                       and ops.gn_in_conv_ok(x.B, x.H, x.W, x.C, N, ops.choose_splitk(M, N, taps * x.C, img8=img8_)[1], x.ld))
            if not fusable:
                # the caller's conv_n promised another convolution than the one that arrived: the GroupNorm gets its
                # own apply launch after all, on the same statistics, and the convolution runs on the normalised tensor
                if x.gn_apply is None:
                    raise RuntimeError(f"{name}: a deferred GroupNorm reached a convolution that cannot apply it")
                xn = x.gn_apply()
                y = self.conv(xn, prefix, taps=taps, stride=stride, pad_tl=pad_tl, ups=ups, rowvec=rowvec, residual=residual, out=out,
                              out_f32=out_f32, alpha=alpha, name=name, stats=stats, feeds=feeds, mirror=mirror, gnp_into=gnp_into)
                self.free(xn)
                return y
        a, tmp = x.t, None
        if self.hp:
            a, tmp, parts = self._operand(x.t, x.rows, x.C, parts)
            out_f32 = True if out is None else out.dtype == torch.float32
            if out is None and self.branch16 and feeds and self.feeds_parts(feeds, M) == 1:
                out_f32 = False
        Ce = ops.k_mult(parts) * x.C
        # nearest-2x upsample convolutions run in their sub-pixel form (four 2x2 convolutions of the source image with pre-summed
        # weights: 4 instead of 9 multiply-adds per output element) wherever the halo kernel's geometry takes them
        subpix = (ups and taps == 9 and stride == 1 and pad_tl == 1 and not self.invariant
                  and ops.subpixel_ok(x.H, x.W, Ce, N, x.B, a.stride(0)))
        if subpix:
            w, bias = self.store.conv_subpixel(prefix, cin_pad=x.C, bias_scale=alpha)
            spatial = spatial[:7] + (2,)
        wt = self._w(w, parts)
        m16 = None
        if out is None:
            out = self.new(M, N, torch.float32 if out_f32 else (self.dtype if self.hp else None))
            if mirror and self.mirrors_on and out_f32:
                m16 = self.add_mirror(out)
        elif self.mirrors_on and out_f32:
            m16 = self.mirror_view(out)
        img8 = taps == 9 and stride == 1 and pad_tl == 1 and not ups and (x.H, x.W) == (8, 8) and Ce % 64 == 0
        tile, splitk = ops.choose_splitk(M, N, taps * Ce, img8=img8)
        if self.invariant:
            tile, splitk = ops.invariant_tile(Ce, 0), 1
        if subpix:
            tile, splitk = 16, 1
        if x.gn_in is not None:          # the input's GroupNorm rides in this convolution's patch staging (group_norm(..., conv_n=N))
            if (taps != 9 or stride != 1 or pad_tl != 1 or ups or (self.hp and (a.dtype != self.dtype or parts != 1 or a is not x.t))
                    or not ops.gn_in_conv_ok(x.B, x.H, x.W, x.C, N, splitk, x.ld)):
                raise RuntimeError(f"{name}: a deferred GroupNorm reached a convolution that cannot apply it")      # (checked above)
            tile = 0             # (edtr_igemm picks the halo geometry itself: tile 17 from 256 units of 512 pixels, tile 16 below)
        ws = self.arena.alloc((splitk * M * N,), torch.float32) if splitk > 1 else None
        gnp, gn_slot, into = None, 128, None
        self.gnp_into_done = False
        if stats and (self.hp or not out_f32) and out.stride(0) == N and ops.gn_fusable(M, N, Ce, OH * OW, splitk=splitk, invariant=self.invariant):
            gn_slot = ops.gn_slot_rows(OH * OW) if splitk > 1 else 128
            gnp = self.arena.alloc((M // gn_slot, N, 2), torch.float32)
        elif gnp_into is not None and not out_f32:
            into = self._gnp_into_args(gnp_into, M, N, Ce, OH * OW, splitk)
            self.gnp_into_done = into is not None
        res32 = residual is not None and residual.dtype == torch.float32
        self.prog.add(ops.make_igemm(
            dtype=self.dtype, a1=a, w=wt, out=out, taps=taps, M=M, N=N, C1=Ce, ld1=a.stride(0), ldw=wt.stride(0),
            ldc=out.stride(0), spatial=spatial, bias_n=bias, rowvec=rowvec,
            rowvec_ld=rowvec.stride(0) if rowvec is not None else 0, rows_per_image=OH * OW, residual=residual,
            ldr=residual.stride(0) if residual is not None else 0, residual_f32=res32, out_f32=out_f32, alpha=alpha, tile=tile,
            splitk=splitk, workspace=ws, name=name,
            **(into if into is not None else dict(gn_partial=gnp, gn_slot_rows=gn_slot if splitk > 1 else 0)),
            w_phase_stride=(N * wt.stride(0)) if subpix else 0,
            out16=m16, a_wrap=x.C if parts == ops.PARTS_2W else 0, a_gn=x.gn_in, a_gn_silu=x.gn_silu))
        self.arena.free(ws)
        self.arena.free(tmp)
        return Act(out, x.B, OH, OW, N, gnp, gn_slot=gn_slot)

    def conv128_out(self, x: Act, prefix: str, out_nchw: torch.Tensor, n_valid: int, alpha: float = 1.0, name="vae.conv_out") -> None:
        """The VAE decoder's conv_out on edtr_conv128_out: ``x`` carries its deferred GroupNorm (group_norm / gn_stats_into with
        conv_n = -n_valid), the result goes straight into the fp32 NCHW tensor."""
        key = ("conv128_out", prefix, alpha)
        if key not in self.store.cache:
            w = self.store._p(prefix + "weight")
            self.store.cache[key] = (ops.pack_conv128_out_weight(w, self.dtype), ops.pad_bias(self.store._p(prefix + "bias") * alpha, 32))
        w, b = self.store.cache[key]
        self.prog.add(ops.make_conv128_out(dtype=self.dtype, x=x.t, ldx=x.ld, w=w, bias=b, out=out_nchw, B=x.B, H=x.H, W=x.W, n_valid=n_valid,
                                           gn_table=x.gn_in, alpha=alpha, name=name))

    # -- norms --------------------------------------------------------------------------------
    def _gn_recs(self, x: Act, prefix: str, eps: float, silu: bool, sums: Optional[torch.Tensor], y: torch.Tensor,
                 sums_zeroed: bool = False, parts: int = 1, partial=None):
        gamma, beta = self.store.vec(prefix + "weight", x.C), self.store.vec(prefix + "bias", x.C)
        c_real = self.store.params[prefix + "weight"].numel()
        if c_real != x.C:
            raise ValueError(f"GroupNorm {prefix}: activation has {x.C} channels, parameter has {c_real}")
        in32 = x.t.dtype == torch.float32
        return ops.make_gn(dtype=(self.op_fmt(parts) if in32 else self.dtype) if self.hp else self.io, x=x.t, ldx=x.ld, B=x.B, HW=x.H * x.W, C=x.C,
                           sums=sums, gamma=gamma, beta=beta, eps=eps, silu=silu, y=y, ldy=y.stride(0), sums_zeroed=sums_zeroed,
                           partial=partial, tiles_per_image=x.gn_tiles)

    def _norm_out(self, rows: int, C: int, parts: int):
        """(buffer the apply launch writes, what the caller carries): fp32-stream modes = the multi-part operand itself."""
        if self.hp:
            t = self.arena.alloc((rows, parts * C), self.dtype)
            return t, OpN(t, C, parts)
        y = self.new(rows, C)
        return y, y

    def gn_deferrable(self, x: Act, conv_n: int, feeds=None) -> bool:
        """Can the 3x3 / stride 1 / pad 1 convolution with ``conv_n`` output channels that consumes this GroupNorm apply it itself
        (halo tiles)?  conv_n < 0: the consumer is edtr_conv128_out with -conv_n channels.  Fast modes: any 16-bit tensor.  fp32-stream
        modes (round 5): only a BRANCH-INTERNAL tensor that is already stored in the operand format (conv1 -> norm2 -> conv2 of a
        ResBlock under `branch16`) whose consumer runs the one-part product — the same arithmetic and rounding as the apply launch."""
        if not conv_n or self.invariant or x.t.dtype == torch.float32:
            return False
        if self.hp and (conv_n < 0 or x.t.dtype != self.dtype or feeds is None or self.feeds_parts(feeds, x.rows) != 1
                        or os.environ.get("EDTR_GN_IN_CONV_HP", "1") == "0"):
            return False
        if conv_n < 0:
            return ops.conv128_out_ok(x.H, x.W, x.C, -conv_n)
        _, splitk = ops.choose_splitk(x.rows, conv_n, 9 * x.C)
        return ops.gn_in_conv_ok(x.B, x.H, x.W, x.C, conv_n, splitk, x.ld)

    def _deferred(self, x: Act, table: torch.Tensor, silu: bool, take: bool, apply=None) -> Act:
        y = Act(x.t, x.B, x.H, x.W, x.C, x.gnp if take else None, gn_in=table, gn_silu=silu, owns=take, gn_apply=apply)
        if take:                 # the raw tensor now belongs to the deferred activation: the caller's em.free(x) is a no-op
            x.t, x.gnp = None, None
        return y

    def group_norm(self, x: Act, prefix: str, eps: float, silu: bool, out=None, feeds=None, conv_n: int = 0, take: bool = False,
                   lin_ok: bool = False) -> Act:
        """``feeds``: the GEMM classes that consume the result (their precision policy decides how many operand parts the
        apply launch writes in the fp32-stream modes).  ``conv_n``: the result's ONLY consumer is a 3x3 / stride 1 / pad 1
        convolution with that many output channels — where the halo tile takes it, no apply launch is emitted: one small launch
        turns the statistics into a (scale, shift) table and the convolution normalises its operand while staging it (the returned
        Act carries the RAW tensor; ``take``: it takes over x's storage, i.e. the caller is done with x).  ``lin_ok``: the ONLY consumer
        is an edtr_lin320 projection (proj_in of a SpatialTransformer at the 64 x 64-latent level), which applies the table to the rows
        it holds in registers (no SiLU) — deferred the same way."""
        lin_defer = (lin_ok and not silu and not self.hp and not self.invariant and x.t.dtype != torch.float32 and (x.H * x.W) % ops.LIN320_ROWS == 0
                     and os.environ.get("EDTR_LIN320_GN", "1") != "0")
        if out is None and (self.gn_deferrable(x, conv_n, feeds) or lin_defer):
            gamma, beta = self.store.vec(prefix + "weight", x.C), self.store.vec(prefix + "bias", x.C)
            table = self.arena.alloc((x.B, x.C, 2), torch.float32)
            hw = x.H * x.W
            if x.gnp is not None:
                self.prog.add(ops.make_gn_table(partial=x.gnp, tiles_per_image=x.gn_tiles, sums=None, B=x.B, C=x.C, HW=hw, gamma=gamma,
                                                beta=beta, eps=eps, table=table))
            else:
                sums = self.prog.sums_slot(self.arena, x.B)
                st, _ = self._gn_recs(x, prefix, eps, silu, sums, x.t, sums_zeroed=True)
                self.prog.add(st)
                self.prog.add(ops.make_gn_table(partial=None, tiles_per_image=0, sums=sums, B=x.B, C=x.C, HW=hw, gamma=gamma, beta=beta,
                                                eps=eps, table=table))
            raw, gnp, geo, slot = x.t, x.gnp, (x.B, x.H, x.W, x.C), x.gn_slot
            stat_sums = None if x.gnp is not None else sums

            def apply_now() -> Act:          # the launch(es) group_norm would have emitted without the deferral, on the same statistics
                xr = Act(raw, *geo, gnp, gn_slot=slot)
                y, carried = self._norm_out(xr.rows, xr.C, 1)
                if gnp is not None and ops.gn_foldable(xr.H * xr.W, xr.C, tiles=xr.gn_tiles):
                    _, ap = self._gn_recs(xr, prefix, eps, silu, None, y, partial=gnp)
                elif gnp is not None:
                    fs = self.arena.alloc((xr.B, 32, 2), torch.float64)
                    _, ap = self._gn_recs(xr, prefix, eps, silu, fs, y)
                    self.prog.add(ops.make_gn_finalize(partial=gnp, tiles_per_image=xr.gn_tiles, B=xr.B, C=xr.C, sums=fs))
                    self.arena.free(fs)
                else:
                    _, ap = self._gn_recs(xr, prefix, eps, silu, stat_sums, y, sums_zeroed=True)
                self.prog.add(ap)
                return Act(carried, *geo)
            return self._deferred(x, table, silu, take, apply_now)
        parts = self.feeds_parts(feeds, x.rows)
        if self.hp and x.t.dtype != torch.float32:
            parts = 1            # (a branch-internal fp16 tensor: its low parts are exactly zero)
        if out is not None:
            y, carried = out, out
        else:
            y, carried = self._norm_out(x.rows, x.C, parts)
        sums = None
        if x.gnp is not None and ops.gn_foldable(x.H * x.W, x.C, tiles=x.gn_tiles):
            # the producer's epilogue already reduced this tensor per 128-row tile, and the tiles are few: the apply launch folds
            # them itself (one launch per GroupNorm instead of two)
            _, ap = self._gn_recs(x, prefix, eps, silu, None, y, parts=parts, partial=x.gnp)
            st = None
        elif x.gnp is not None:  # many tiles (the VAE's large levels): a finalize launch folds them once for all workgroups
            sums = self.arena.alloc((x.B, 32, 2), torch.float64)
            _, ap = self._gn_recs(x, prefix, eps, silu, sums, y, parts=parts)
            st = ops.make_gn_finalize(partial=x.gnp, tiles_per_image=x.gn_tiles, B=x.B, C=x.C, sums=sums)
        else:                    # atomically accumulated statistics: a pre-zeroed pool slot, never reused in this program
            sums = self.prog.sums_slot(self.arena, x.B)
            st, ap = self._gn_recs(x, prefix, eps, silu, sums, y, sums_zeroed=True, parts=parts)
        if st is not None:
            self.prog.add(st)
        self.prog.add(ap)
        if x.gnp is not None and sums is not None:
            self.arena.free(sums)
        return Act(carried, x.B, x.H, x.W, x.C)

    def gn_stats_into(self, x: Act, prefix: str, eps: float, silu: bool, sums: torch.Tensor, sums_zeroed: bool = False,
                      feeds=None, conv_n: int = 0, take: bool = False):
        """Statistics half only (tiled VAE: the caller pools `sums` across tiles before the apply half).
        Returns a closure that emits the apply half and yields the normalised activation.  ``conv_n`` / ``take``: as group_norm —
        the apply half is then the (scale, shift) table launch and the activation stays raw."""
        if self.gn_deferrable(x, conv_n, feeds):
            if x.gnp is not None:
                self.prog.add(ops.make_gn_finalize(partial=x.gnp, tiles_per_image=x.gn_tiles, B=x.B, C=x.C, sums=sums))
            else:
                st, _ = self._gn_recs(x, prefix, eps, silu, sums, x.t, sums_zeroed=sums_zeroed)
                self.prog.add(st)

            def table_apply() -> Act:
                gamma, beta = self.store.vec(prefix + "weight", x.C), self.store.vec(prefix + "bias", x.C)
                table = self.arena.alloc((x.B, x.C, 2), torch.float32)
                self.prog.add(ops.make_gn_table(partial=None, tiles_per_image=0, sums=sums, B=x.B, C=x.C, HW=x.H * x.W, gamma=gamma,
                                                beta=beta, eps=eps, table=table))
                raw, geo = x.t, (x.B, x.H, x.W, x.C)

                def apply_now() -> Act:
                    xr = Act(raw, *geo)
                    y, carried = self._norm_out(xr.rows, xr.C, 1)
                    _, ap = self._gn_recs(xr, prefix, eps, silu, sums, y, sums_zeroed=True)
                    self.prog.add(ap)
                    return Act(carried, *geo)
                return self._deferred(x, table, silu, take, apply_now)
            return table_apply
        parts = self.feeds_parts(feeds, x.rows)
        if self.hp and x.t.dtype != torch.float32:
            parts = 1
        y, carried = self._norm_out(x.rows, x.C, parts)
        st, ap = self._gn_recs(x, prefix, eps, silu, sums, y, sums_zeroed=sums_zeroed, parts=parts)
        if x.gnp is not None:
            st = ops.make_gn_finalize(partial=x.gnp, tiles_per_image=x.gn_tiles, B=x.B, C=x.C, sums=sums)
        self.prog.add(st)

        def apply() -> Act:
            self.prog.add(ap)
            return Act(carried, x.B, x.H, x.W, x.C)
        return apply

    def layer_norm(self, x: torch.Tensor, rows: int, C: int, prefix: str, feeds=None, stats: Optional[torch.Tensor] = None):
        """``stats``: the producing GEMM's per-row statistics of ``x`` (gemm(..., row_stats=True)) — no launch then: an LNRef
        that the consuming GEMMs fold into their weights and epilogues."""
        if stats is not None:
            return LNRef(x, stats, C, prefix)
        parts = self.feeds_parts(feeds, rows)
        y, carried = self._norm_out(rows, C, parts)
        self.prog.add(ops.make_layernorm(dtype=self.op_fmt(parts) if self.hp else self.io, x=x, rows=rows, C=C, ldx=x.stride(0),
                                         gamma=self.store.vec(prefix + "weight"), beta=self.store.vec(prefix + "bias"),
                                         eps=1e-5, y=y, ldy=y.stride(0)))
        return carried

    # -- elementwise --------------------------------------------------------------------------
    def add(self, a: torch.Tensor, b: Optional[torch.Tensor], rows: int, C: int, out=None, stats_into=None) -> torch.Tensor:
        """``stats_into`` = (buffer [slots][Ctot][2], first column, rows per slot): the launch also writes the result's GroupNorm partials
        into the shared slot buffer of the concatenation it fills a column slice of (fast modes; Emitter._gnp_into_args)."""
        if out is None:
            out = self.new(rows, C)
        if stats_into is not None and not self.hp and C % 32 == 0 and rows % stats_into[2] == 0:
            buf, col, slot = stats_into
            self.prog.add(ops.make_add_stats(dtype=self.io, a=a, lda=a.stride(0), b=b, ldb=b.stride(0) if b is not None else 0, out=out,
                                             ldo=out.stride(0), rows=rows, C=C, gn_partial=buf.reshape(-1)[2 * col:], gn_ld=buf.shape[1],
                                             slot_rows=slot))
            self.add_stats_done = True
            return out
        self.add_stats_done = False
        m16 = self.mirror_view(out) if self.mirrors_on else None
        if m16 is not None and C % 8:
            raise ValueError(f"add into a mirrored stream buffer needs C % 8 == 0 (got {C}): its fp16 mirror would go stale")
        if m16 is not None:
            self.prog.add(ops.make_add_mirror(a=a, lda=a.stride(0), b=b, ldb=b.stride(0) if b is not None else 0, out=out,
                                              ldo=out.stride(0), out16=m16, rows=rows, C=C))
            return out
        self.prog.add(ops.make_add(dtype=self.io, a=a, lda=a.stride(0), b=b, ldb=b.stride(0) if b is not None else 0,
                                   out=out, ldo=out.stride(0), rows=rows, C=C))
        return out

    def to_nhwc(self, src: torch.Tensor, B: int, C: int, HW: int, dst: torch.Tensor, coff=0, pad_to=0, scale=1.0,
                shift=0.0) -> None:
        self.prog.add(ops.make_nchw_to_nhwc(dtype=self.io, src=src, B=B, C=C, HW=HW, dst=dst, ld=dst.stride(0),
                                            coff=coff, zero_pad_to=pad_to, scale=scale, shift=shift))

    def to_nchw(self, src: torch.Tensor, B: int, C: int, HW: int, dst: torch.Tensor, scale=1.0) -> None:
        self.prog.add(ops.make_nhwc_to_nchw(dtype=self.dtype, src=src, src_f32=(src.dtype == torch.float32), B=B, C=C,
                                            HW=HW, ld=src.stride(0), dst=dst, scale=scale))

    def cast_flat(self, src_f32: torch.Tensor, n: int) -> torch.Tensor:
        """fp32 -> 16-bit cast of n contiguous elements (inputs such as c_txt); the high-precision mode keeps fp32."""
        if self.hp:
            return src_f32.reshape(n, 1)
        dst = self.new(n, 1)
        self.prog.add(ops.make_nchw_to_nhwc(dtype=self.dtype, src=src_f32, B=1, C=1, HW=n, dst=dst, ld=1, name="cast16"))
        return dst

    # -- attention ------------------------------------------------------------------------------
    def split16(self, x: torch.Tensor, rows: int, C: int) -> torch.Tensor:
        """fp32 [rows, C] view -> fp16 [rows, 2C] = [hi | lo] (x = hi + lo to ~22 bits): a split attention operand."""
        y = self.arena.alloc((rows, 2 * C), self.attn_dtype)
        self.prog.add(ops.make_split_operand(src=x, rows=rows, C=C, dst=y, fmt=ops.F32H[2], name="attn.split"))
        return y

    def flash(self, q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, *, B, H, Nq, Nk, k_bs, vt_bs, vt_ld,
              out=None, causal: bool = False, prescaled: bool = False, k_lo=None, vt_lo=None) -> torch.Tensor:
        """``k_lo`` / ``vt_lo``: low halves of operands that arrive already split (the per-prompt context keys / values)."""
        C = H * 64
        tmp = []
        q_lo = None
        split = self.attn_split if self.hp else 0
        if split and q.dtype == torch.float32:      # high mode: hi + lo pairs, three MFMA products per attention product
            q2 = self.split16(q, B * Nq, C)
            tmp.append(q2)
            q, q_lo = q2[:, :C], q2[:, C:]
            if k.dtype == torch.float32:
                k2 = self.split16(k, B * Nk, C)
                tmp.append(k2)
                k, k_lo, k_bs = k2[:, :C], k2[:, C:], Nk * 2 * C
            elif k_lo is None:
                raise ValueError("split attention: k must be fp32 or arrive with its low half")
            if split >= 2:
                if vt.dtype == torch.float32:
                    v2 = self.split16(vt, B * C, vt_ld)
                    tmp.append(v2)
                    vt, vt_lo, vt_bs, vt_ld = v2[:, :vt_ld], v2[:, vt_ld:], C * 2 * vt_ld, 2 * vt_ld
                elif vt_lo is None:
                    raise ValueError("split attention: v^T must be fp32 or arrive with its low half")
            else:
                vt_lo = None
        else:
            k_lo = vt_lo = None
        if self.hp:      # fp32 projections -> fp16 operands (mixed mode: the projections wrote fp16 already; k / v^T of the context arrive cast)
            if q.dtype == torch.float32:
                q = self.to16(q, B * Nq, C)
                tmp.append(q)
            if k.dtype == torch.float32:
                k = self.to16(k, B * Nk, C)
                tmp.append(k)
                k_bs = Nk * C
            if vt.dtype == torch.float32:
                vt = self.to16(vt, B * C, vt_ld)
                tmp.append(vt)
                vt_bs = C * vt_ld
        out32 = q_lo is not None and vt_lo is not None      # fully split: the result goes to a multi-part GEMM unrounded
        if out is None:
            out = self.arena.alloc((B * Nq, C), torch.float32 if out32 else self.attn_dtype)
        self.prog.add(ops.make_flash_attn(dtype=self.attn_dtype, q=q, k=k, vt=vt, out=out, B=B, H=H, Nq=Nq, Nk=Nk,
                                          q_bs=Nq * q.stride(0), q_ld=q.stride(0), k_bs=k_bs, k_ld=k.stride(0),
                                          vt_bs=vt_bs, vt_ld=vt_ld, o_bs=Nq * out.stride(0), o_ld=out.stride(0),
                                          scale=1.0 / math.sqrt(64.0), causal=causal, prescaled=prescaled,
                                          q_lo=q_lo, k_lo=k_lo, vt_lo=vt_lo, out_f32=out.dtype == torch.float32))
        self.free(*tmp)
        return out

    def vt_gemm(self, wv, x, *, B, Ntok, Cin, bias_m=None, name="v_transposed", alpha=1.0, out16=True) -> Tuple[torch.Tensor, int]:
        """V^T[b] = Wv @ x[b]^T  ->  [B, Cout, roundup8(Ntok)] (padding keys exactly zero when bias_m is None).
        fp32-stream modes: A = the packed weight [Wh | Wh | Wl][:parts], B operand = x as [hi | lo | hi][:parts]
        (wh*xh + wh*xl + wl*xh); the mixed mode writes the fp16 attention operand directly (``out16``), the high mode
        writes fp32 (its MFMA type is bf16) and the attention emitter casts."""
        Cout = wv.shape[0]
        ldv = round_up(Ntok, 8)
        parts = self.parts_for(name, Cout, ldv, Cin)
        parts = 3 if parts == ops.PARTS_2W else parts
        tmp = None
        if self.hp:
            x, tmp, parts = self._operand(x, B * Ntok, Cin, parts)
        f32 = self.hp and not (self.direct16 and out16)
        wt = self._w(wv, parts)
        vt = self.arena.alloc((B * Cout, ldv), torch.float32 if f32 else self.dtype)
        self.prog.add(ops.make_igemm(dtype=self.dtype, a1=wt, w=x, out=vt, M=Cout, N=ldv, n_valid=Ntok, C1=parts * Cin,
                                     ld1=wt.stride(0), ldw=x.stride(0), ldc=ldv, Z=B, a_zs=(0, 0),
                                     w_zs=(Ntok * x.stride(0), 0), o_zs=(Cout * ldv, 0), bias_m=bias_m, out_f32=f32,
                                     alpha=alpha, name=name))
        self.arena.free(tmp)
        return vt, ldv
