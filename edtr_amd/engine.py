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
class WeightStore:
    """fp32 parameters (reference names/shapes) -> device-resident packed 16-bit matrices + fp32 vectors.
    Packs lazily, caches per (kind, names); ``invalidate()`` after the parameters change."""

    def __init__(self, params: Dict[str, torch.Tensor], dtype, device: torch.device):
        """``dtype``: torch.bfloat16 / torch.float16, or ops.F32S for the high-precision mode (bf16 matrices whose K axis is
        the split [hi | hi | lo], three times as wide; see include/edtr_hip.h EDTR_F32_SPLIT)."""
        self.params, self.dtype, self.device = params, dtype, device
        self.cache: Dict[tuple, object] = {}

    def invalidate(self):
        self.cache.clear()

    def _p(self, name: str) -> torch.Tensor:
        return self.params[name].detach().to(self.device, torch.float32)

    def vec(self, name: str, n_pad: Optional[int] = None) -> torch.Tensor:
        key = ("vec", name, n_pad)
        if key not in self.cache:
            v = self._p(name).reshape(-1)
            self.cache[key] = ops.pad_bias(v, n_pad or v.numel()).contiguous()
        return self.cache[key]

    def conv(self, prefix: str, cin_pad: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(w16 [Np, taps*Cinp], bias f32 [Np]) for ``prefix + 'weight'/'bias'`` (3x3 or 1x1 conv)."""
        key = ("conv", prefix, cin_pad)
        if key not in self.cache:
            w = self._p(prefix + "weight")
            wp = ops.pack_conv_weight(w, self.dtype, cin_pad=cin_pad)
            self.cache[key] = (wp, self.vec(prefix + "bias", wp.shape[0]))
        return self.cache[key]

    def linear(self, names: Sequence[str], biases: Optional[Sequence[Optional[str]]] = None):
        """Row-concatenation of several [out, in] matrices (fused projections) + matching fp32 bias (or None)."""
        key = ("linear", tuple(names), tuple(biases) if biases else None)
        if key not in self.cache:
            ws = [self._p(n).reshape(self.params[n].shape[0], -1) for n in names]
            w = torch.cat(ws, dim=0)
            wp = ops.pack_linear_weight(w, self.dtype)
            b = None
            if biases:
                parts = [self._p(bn).reshape(-1) if bn else torch.zeros(ws[i].shape[0], device=self.device)
                         for i, bn in enumerate(biases)]
                b = ops.pad_bias(torch.cat(parts), wp.shape[0])
            self.cache[key] = (wp, b)
        return self.cache[key]

    def rows(self, name: str, r0: int, r1: int, bias: Optional[str] = None):
        """Rows r0..r1 of one [out, in] matrix (e.g. the q/k or the v part of a fused in_proj) + the matching bias slice."""
        key = ("rows", name, r0, r1, bias)
        if key not in self.cache:
            w = self._p(name)
            wp = ops.pack_linear_weight(w.reshape(w.shape[0], -1)[r0:r1].contiguous(), self.dtype)
            b = ops.pad_bias(self._p(bias).reshape(-1)[r0:r1].contiguous(), wp.shape[0]) if bias else None
            self.cache[key] = (wp, b)
        return self.cache[key]

    def raw(self, name: str) -> torch.Tensor:
        """The fp32 parameter itself on the device (embedding tables)."""
        key = ("raw", name)
        if key not in self.cache:
            self.cache[key] = self._p(name).contiguous()
        return self.cache[key]

    def geglu(self, wname: str, bname: str):
        key = ("geglu", wname)
        if key not in self.cache:
            w, b = self._p(wname), self._p(bname)
            perm = ops.geglu_perm(w.shape[0] // 2).to(self.device)
            self.cache[key] = (ops.pack_linear_weight(w[perm], self.dtype), b[perm].contiguous())
        return self.cache[key]


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
        # Measurement aid (EDTR_EXP_DUP=<substring of a launch name>): idempotent launches whose name matches are issued TWICE.
        # The slowdown of a whole-path run is the MARGINAL wall-clock cost of that kernel class inside the overlapped hipGraph
        # execution — what a per-launch event timing cannot show.  Never set in production.
        dup = os.environ.get("EDTR_EXP_DUP")
        if dup and dup in rec.name and not rec.name.endswith(".stats"):
            self.recs.append(rec)
            self.lanes.append(self.lane)
        return rec

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
class Op3:
    """High-precision GEMM operand: bf16 ``t3`` [rows, 3*C] = [hi | lo | hi] of an fp32 [rows, C] activation."""

    def __init__(self, t3: torch.Tensor, C: int):
        self.t3, self.C = t3, C

    def stride(self, dim: int) -> int:
        return self.t3.stride(dim)


@dataclass
class Act:
    """NHWC activation: ``t`` is a 2-D view [B*H*W, C] (row stride ``ld`` >= C) of 16-bit storage — fp32 storage, or an Op3
    (the output of a normalisation, which only ever feeds a GEMM) in the high-precision mode."""
    t: object
    B: int
    H: int
    W: int
    C: int
    gnp: Optional[torch.Tensor] = None    # fused GroupNorm partials written by the producing igemm ([rows/128][C][2] fp32)

    @property
    def ld(self) -> int:
        return self.t.stride(0)

    @property
    def rows(self) -> int:
        return self.B * self.H * self.W


class Emitter:
    """Primitive emitters.  ``precision="high"`` selects the parity mode: fp32 activation stream, every GEMM / convolution
    as a bf16 split-3 product (3x the K, fp32 out), residual adds as fp32 launches, attention operands in fp16."""

    def __init__(self, prog: Program, arena: Arena, store: WeightStore, dtype: torch.dtype, precision: str = "fast"):
        self.prog, self.arena, self.store = prog, arena, store
        self.hp = precision == "high"
        if self.hp != (store.dtype == ops.F32S):
            raise ValueError("the weight store and the emitter must agree on the precision mode")
        self.dtype = torch.bfloat16 if self.hp else dtype          # MFMA operand type of edtr_igemm
        self.attn_dtype = torch.float16 if self.hp else dtype      # q / k / v^T / P of the attention kernels
        self.act_dtype = torch.float32 if self.hp else dtype       # storage of the activation stream
        self.io = ops.F32S if self.hp else dtype                    # dtype code of the norm / layout / elementwise launches
        self.last_gnp = None

    # -- memory
    def new(self, rows: int, cols: int, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        return self.arena.alloc((rows, cols), dtype or self.act_dtype)

    def free(self, *ts) -> None:
        for t in ts:
            if isinstance(t, Act):
                self.arena.free(t.gnp)
                t = t.t
            if isinstance(t, Op3):
                t = t.t3
            self.arena.free(t)

    # -- high-precision operands ------------------------------------------------------------------
    def _operand(self, a, rows: int, C: int, pattern: int = 0):
        """(bf16 [rows, 3C] operand, temporary to free or None) of an fp32 / fp16 activation or of a ready Op3."""
        if isinstance(a, Op3):
            if a.C != C or pattern != 0:
                raise ValueError("Op3 operand does not match the GEMM")
            return a.t3, None
        t3 = self.arena.alloc((rows, 3 * C), torch.bfloat16)
        self.prog.add(ops.make_split3(src=a, rows=rows, C=C, dst=t3, pattern=pattern))
        return t3, t3

    def to16(self, x: torch.Tensor, rows: int, C: int, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """High-precision mode: a 16-bit copy (attention operand) of an fp32 [rows, C] view."""
        y = self.arena.alloc((rows, C), dtype or self.attn_dtype)
        self.prog.add(ops.make_cast16(dtype=dtype or self.attn_dtype, src=x, rows=rows, C=C, dst=y))
        return y

    # -- GEMM family ------------------------------------------------------------------------
    def gemm(self, a, w: torch.Tensor, M: int, N: int, K: int, *, bias=None, out=None, act=0,
             residual=None, rowvec=None, rows_per_image=0, out_f32=False, alpha=1.0, name="linear", stats_hw=0,
             **kw) -> torch.Tensor:
        """out[M, N'] = epilogue(a[M, K] @ w[N, K]^T).  ``a``/``out``/``residual`` are 2-D views (row stride = ld)."""
        n_out = N // 2 if act == L.ACT_GEGLU else N
        self.last_gnp = None     # fused GroupNorm partials of this output (stats_hw = pixels per image), if eligible
        if self.hp:
            if out is None:
                out = self.new(M, n_out, torch.float32)
            a3, tmp = self._operand(a, M, K)
            self.prog.add(ops.make_igemm(
                dtype=self.dtype, a1=a3, w=w, out=out, M=M, N=N, C1=3 * K, ld1=a3.stride(0), ldw=w.stride(0),
                ldc=out.stride(0), bias_n=bias, act=act, rowvec=rowvec, rowvec_ld=rowvec.stride(0) if rowvec is not None else 0,
                rows_per_image=rows_per_image, out_f32=True, alpha=alpha, name=name, **kw))
            self.arena.free(tmp)
            if residual is not None:
                self.add(out, residual, M, n_out, out=out)
            return out
        if out is None:
            out = self.new(M, n_out, torch.float32 if out_f32 else None)
        tile, splitk = ops.choose_splitk(M, N, K, kw.get("Z", 1), act) if "tile" not in kw else (kw.pop("tile"), 1)
        ws = self.arena.alloc((splitk * M * N,), torch.float32) if splitk > 1 else None
        if (stats_hw and act == 0 and not out_f32 and out.stride(0) == N and "Z" not in kw
                and ops.gn_fusable(M, N, K, stats_hw, splitk=splitk)):
            self.last_gnp = self.arena.alloc((M // 128, N, 2), torch.float32)
            kw["gn_partial"] = self.last_gnp
        self.prog.add(ops.make_igemm(
            dtype=self.dtype, a1=a, w=w, out=out, M=M, N=N, C1=K, ld1=a.stride(0), ldw=w.stride(0), ldc=out.stride(0),
            bias_n=bias, act=act, residual=residual, ldr=residual.stride(0) if residual is not None else 0,
            rowvec=rowvec, rowvec_ld=rowvec.stride(0) if rowvec is not None else 0, rows_per_image=rows_per_image,
            out_f32=out_f32, alpha=alpha, name=name, tile=tile, splitk=splitk, workspace=ws, **kw))
        self.arena.free(ws)
        return out

    def conv(self, x: Act, prefix: str, *, taps=9, stride=1, pad_tl=1, ups=False, rowvec=None, residual=None,
             out=None, out_f32=False, alpha=1.0, name=None, stats=False) -> Act:
        """3x3 (or 1x1) convolution of an NHWC activation with the packed weight ``prefix``."""
        w, bias = self.store.conv(prefix, cin_pad=x.C)
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
        if alpha != 1.0:
            bias = bias * alpha  # epilogue applies alpha before the bias
        if self.hp:
            if out is None:
                out = self.new(M, N, torch.float32)
            a3, tmp = self._operand(x.t, x.rows, x.C)
            self.prog.add(ops.make_igemm(
                dtype=self.dtype, a1=a3, w=w, out=out, taps=taps, M=M, N=N, C1=3 * x.C, ld1=a3.stride(0), ldw=w.stride(0),
                ldc=out.stride(0), spatial=spatial, bias_n=bias, rowvec=rowvec,
                rowvec_ld=rowvec.stride(0) if rowvec is not None else 0, rows_per_image=OH * OW, out_f32=True, alpha=alpha,
                name=name or ("conv3x3" if taps == 9 else "conv1x1")))
            self.arena.free(tmp)
            if residual is not None:
                self.add(out, residual, M, N, out=out)
            return Act(out, x.B, OH, OW, N, None)
        if out is None:
            out = self.new(M, N, torch.float32 if out_f32 else None)
        tile, splitk = ops.choose_splitk(M, N, taps * x.C)
        ws = self.arena.alloc((splitk * M * N,), torch.float32) if splitk > 1 else None
        gnp = None
        if stats and not out_f32 and out.stride(0) == N and ops.gn_fusable(M, N, x.C, OH * OW, splitk=splitk):
            gnp = self.arena.alloc((M // 128, N, 2), torch.float32)
        self.prog.add(ops.make_igemm(
            dtype=self.dtype, a1=x.t, w=w, out=out, taps=taps, M=M, N=N, C1=x.C, ld1=x.ld, ldw=w.stride(0),
            ldc=out.stride(0), spatial=spatial, bias_n=bias, rowvec=rowvec,
            rowvec_ld=rowvec.stride(0) if rowvec is not None else 0, rows_per_image=OH * OW, residual=residual,
            ldr=residual.stride(0) if residual is not None else 0, out_f32=out_f32, alpha=alpha, tile=tile, splitk=splitk,
            workspace=ws, gn_partial=gnp, name=name or ("conv3x3" if taps == 9 else "conv1x1")))
        self.arena.free(ws)
        return Act(out, x.B, OH, OW, N, gnp)

    # -- norms --------------------------------------------------------------------------------
    def _gn_recs(self, x: Act, prefix: str, eps: float, silu: bool, sums: torch.Tensor, y: torch.Tensor,
                 sums_zeroed: bool = False):
        gamma, beta = self.store.vec(prefix + "weight", x.C), self.store.vec(prefix + "bias", x.C)
        c_real = self.store.params[prefix + "weight"].numel()
        if c_real != x.C:
            raise ValueError(f"GroupNorm {prefix}: activation has {x.C} channels, parameter has {c_real}")
        return ops.make_gn(dtype=self.io, x=x.t, ldx=x.ld, B=x.B, HW=x.H * x.W, C=x.C, sums=sums, gamma=gamma,
                           beta=beta, eps=eps, silu=silu, y=y, ldy=y.stride(0), sums_zeroed=sums_zeroed)

    def _gn_out(self, x: Act):
        """(buffer the apply launch writes, what the Act carries): high-precision mode = the split-3 operand itself."""
        if self.hp:
            y3 = self.arena.alloc((x.rows, 3 * x.C), torch.bfloat16)
            return y3, Op3(y3, x.C)
        y = self.new(x.rows, x.C)
        return y, y

    def group_norm(self, x: Act, prefix: str, eps: float, silu: bool, out=None) -> Act:
        if out is not None:
            y, carried = out, out
        else:
            y, carried = self._gn_out(x)
        if x.gnp is not None:    # the producer's epilogue already reduced this tensor per 128-row tile
            sums = self.arena.alloc((x.B, 32, 2), torch.float64)
            _, ap = self._gn_recs(x, prefix, eps, silu, sums, y)
            st = ops.make_gn_finalize(partial=x.gnp, tiles_per_image=(x.H * x.W) // 128, B=x.B, C=x.C, sums=sums)
        else:                    # atomically accumulated statistics: a pre-zeroed pool slot, never reused in this program
            sums = self.prog.sums_slot(self.arena, x.B)
            st, ap = self._gn_recs(x, prefix, eps, silu, sums, y, sums_zeroed=True)
        self.prog.add(st)
        self.prog.add(ap)
        if x.gnp is not None:
            self.arena.free(sums)
        return Act(carried, x.B, x.H, x.W, x.C)

    def gn_stats_into(self, x: Act, prefix: str, eps: float, silu: bool, sums: torch.Tensor, sums_zeroed: bool = False):
        """Statistics half only (tiled VAE: the caller pools `sums` across tiles before the apply half).
        Returns a closure that emits the apply half and yields the normalised activation."""
        y, carried = self._gn_out(x)
        st, ap = self._gn_recs(x, prefix, eps, silu, sums, y, sums_zeroed=sums_zeroed)
        if x.gnp is not None:
            st = ops.make_gn_finalize(partial=x.gnp, tiles_per_image=(x.H * x.W) // 128, B=x.B, C=x.C, sums=sums)
        self.prog.add(st)

        def apply() -> Act:
            self.prog.add(ap)
            return Act(carried, x.B, x.H, x.W, x.C)
        return apply

    def layer_norm(self, x: torch.Tensor, rows: int, C: int, prefix: str):
        if self.hp:
            y3 = self.arena.alloc((rows, 3 * C), torch.bfloat16)
            y, carried = y3, Op3(y3, C)
        else:
            y = carried = self.new(rows, C)
        self.prog.add(ops.make_layernorm(dtype=self.io, x=x, rows=rows, C=C, ldx=x.stride(0),
                                         gamma=self.store.vec(prefix + "weight"), beta=self.store.vec(prefix + "bias"),
                                         eps=1e-5, y=y, ldy=y.stride(0)))
        return carried

    # -- elementwise --------------------------------------------------------------------------
    def add(self, a: torch.Tensor, b: Optional[torch.Tensor], rows: int, C: int, out=None) -> torch.Tensor:
        if out is None:
            out = self.new(rows, C)
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
    def flash(self, q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, *, B, H, Nq, Nk, k_bs, vt_bs, vt_ld,
              out=None, causal: bool = False, prescaled: bool = False) -> torch.Tensor:
        C = H * 64
        tmp = []
        if self.hp:      # fp32 projections -> fp16 operands (k / v^T of the context arrive already cast)
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
        if out is None:
            out = self.arena.alloc((B * Nq, C), self.attn_dtype)
        self.prog.add(ops.make_flash_attn(dtype=self.attn_dtype, q=q, k=k, vt=vt, out=out, B=B, H=H, Nq=Nq, Nk=Nk,
                                          q_bs=Nq * q.stride(0), q_ld=q.stride(0), k_bs=k_bs, k_ld=k.stride(0),
                                          vt_bs=vt_bs, vt_ld=vt_ld, o_bs=Nq * out.stride(0), o_ld=out.stride(0),
                                          scale=1.0 / math.sqrt(64.0), causal=causal, prescaled=prescaled))
        self.free(*tmp)
        return out

    def vt_gemm(self, wv: torch.Tensor, x, *, B, Ntok, Cin, bias_m=None, name="v_transposed", alpha=1.0) -> Tuple[torch.Tensor, int]:
        """V^T[b] = Wv @ x[b]^T  ->  [B, Cout, roundup8(Ntok)] (padding keys exactly zero when bias_m is None)."""
        Cout = wv.shape[0]
        ldv = round_up(Ntok, 8)
        if self.hp:      # A = [Wh | Wh | Wl] (the packed weight), B operand = x as [hi | lo | hi]: wh*xh + wh*xl + wl*xh
            vt = self.arena.alloc((B * Cout, ldv), torch.float32)
            x3, tmp = self._operand(x, B * Ntok, Cin)
            self.prog.add(ops.make_igemm(dtype=self.dtype, a1=wv, w=x3, out=vt, M=Cout, N=ldv, n_valid=Ntok, C1=3 * Cin,
                                         ld1=wv.stride(0), ldw=x3.stride(0), ldc=ldv, Z=B, a_zs=(0, 0),
                                         w_zs=(Ntok * x3.stride(0), 0), o_zs=(Cout * ldv, 0), bias_m=bias_m, out_f32=True,
                                         alpha=alpha, name=name))
            self.arena.free(tmp)
            return vt, ldv
        vt = self.arena.alloc((B * Cout, ldv), self.dtype)
        self.prog.add(ops.make_igemm(dtype=self.dtype, a1=wv, w=x, out=vt, M=Cout, N=ldv, n_valid=Ntok, C1=Cin,
                                     ld1=wv.stride(0), ldw=x.stride(0), ldc=ldv, Z=B, a_zs=(0, 0),
                                     w_zs=(Ntok * x.stride(0), 0), o_zs=(Cout * ldv, 0), bias_m=bias_m, alpha=alpha, name=name))
        return vt, ldv
