"""Host-side mirror of the reference's `FrozenOpenCLIPEmbedder` (reference model/clip.py:12-65): the OpenCLIP text tower
(token + positional embedding, causal pre-LN transformer, `ln_final`; output of the penultimate or last block) with the
SAME constructor arguments and state-dict keys (`model.positional_embedding`, `model.text_projection`,
`model.logit_scale`, `model.token_embedding.weight`, `model.transformer.resblocks.N.{ln_1,attn.in_proj_*,attn.out_proj,
ln_2,mlp.c_fc,mlp.c_proj}.*`, `model.ln_final.*`), so `load_pretrained_sd` can fill it from `cond_stage_model.*`.

Every forward is a program of libedtr_hip launches: edtr_embed_tokens, edtr_layernorm, edtr_igemm
(fused [Wq;Wk] projection, operand-swapped V^T projection, out projection + residual, c_fc + exact GELU, c_proj +
residual) and edtr_flash_attn64 with the causal flag.  Head width must be 64 (ViT-H text tower: 1024 / 16).

`encode([""] * n)` needs no vocabulary (start / end tokens only); other prompts need the reference's BPE vocabulary
file (`bpe_path=`), tokenised by the restatement of model/open_clip/tokenizer.py below."""
from __future__ import annotations

import gzip
import html
import os
import re
from functools import lru_cache
from typing import Tuple, Dict, List, Optional, Sequence

import torch

from .. import lib as L
from .. import ops as ops_mod
from ..engine import Arena, Emitter, Program, WeightStore
from .params import ParamTree, params_fingerprint

SOT, EOT = 49406, 49407


def clip_text_param_spec(embed_dim: int, text_cfg: dict):
    """(key, shape) in the reference module's state_dict order (model/open_clip/model.py CLIP with `visual` deleted)."""
    W, Lc, V, n = text_cfg["width"], text_cfg["context_length"], text_cfg["vocab_size"], text_cfg["layers"]
    spec = [("model.positional_embedding", (Lc, W)), ("model.text_projection", (W, embed_dim)), ("model.logit_scale", ())]
    for i in range(n):
        p = f"model.transformer.resblocks.{i}."
        spec += [(p + "ln_1.weight", (W,)), (p + "ln_1.bias", (W,)),
                 (p + "attn.in_proj_weight", (3 * W, W)), (p + "attn.in_proj_bias", (3 * W,)),
                 (p + "attn.out_proj.weight", (W, W)), (p + "attn.out_proj.bias", (W,)),
                 (p + "ln_2.weight", (W,)), (p + "ln_2.bias", (W,)),
                 (p + "mlp.c_fc.weight", (4 * W, W)), (p + "mlp.c_fc.bias", (4 * W,)),
                 (p + "mlp.c_proj.weight", (W, 4 * W)), (p + "mlp.c_proj.bias", (W,))]
    spec += [("model.token_embedding.weight", (V, W)), ("model.ln_final.weight", (W,)), ("model.ln_final.bias", (W,))]
    return spec


# ----------------------------------------------------------------------------------------------
# tokenizer: byte-level BPE with the OpenCLIP vocabulary (the algorithm of reference model/open_clip/tokenizer.py:26-157,
# which is GPT-2's; token ids are pinned by tests/golden/clip_tokens.json, generated with the reference tokenizer).  ftfy is
# skipped like in tools/ref_import.py — it only repairs mojibake.
# ----------------------------------------------------------------------------------------------
N_MERGES = 49152 - 256 - 2                     # merge rules kept from the vocabulary file (tokenizer.py:64)
_WORD_END = "</w>"
_PRINTABLE = [(0x21, 0x7E), (0xA1, 0xAC), (0xAE, 0xFF)]     # byte values that stand for themselves


@lru_cache()
def _byte_symbols() -> List[str]:
    """One printable unicode character per byte value: printable latin-1 bytes map to themselves, the other 68 to U+0100...
    in ascending byte order."""
    own = [any(lo <= b <= hi for lo, hi in _PRINTABLE) for b in range(256)]
    table, spare = [], 0
    for b in range(256):
        if own[b]:
            table.append(chr(b))
        else:
            table.append(chr(256 + spare))
            spare += 1
    return table


def _vocabulary_order(symbols: List[str]) -> List[str]:
    """Single-byte tokens are numbered in the order: printable bytes (ascending), then the remapped ones (ascending)."""
    own = [any(lo <= b <= hi for lo, hi in _PRINTABLE) for b in range(256)]
    return [symbols[b] for b in range(256) if own[b]] + [symbols[b] for b in range(256) if not own[b]]


class SimpleTokenizer:
    def __init__(self, bpe_path: str):
        try:
            import regex
        except ImportError as exc:       # \p{L} / \p{N} have no exact stdlib equivalent: no silent approximation
            raise RuntimeError("tokenising a non-empty prompt needs the `regex` package (unicode letter / number classes)") from exc
        self.byte_symbol = _byte_symbols()
        with gzip.open(bpe_path) as f:
            rows = f.read().decode("utf-8").split("\n")[1:N_MERGES + 1]
        self.rank: Dict[Tuple[str, str], int] = {}
        singles = _vocabulary_order(self.byte_symbol)
        names = singles + [c + _WORD_END for c in singles]
        for r, row in enumerate(rows):
            left, right = row.split()
            self.rank[(left, right)] = r
            names.append(left + right)
        names += ["<start_of_text>", "<end_of_text>"]
        self.token_id = {name: i for i, name in enumerate(names)}
        self.split = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                   regex.IGNORECASE)
        self._memo: Dict[str, List[str]] = {}

    def _merge_word(self, chars: str) -> List[str]:
        """Greedy BPE of one pre-token: repeatedly fuse every occurrence (left to right) of the adjacent pair with the lowest
        merge rank until no adjacent pair has a rank."""
        hit = self._memo.get(chars)
        if hit is not None:
            return hit
        parts = list(chars)
        parts[-1] += _WORD_END
        inf = len(self.rank)
        while len(parts) > 1:
            best, best_rank = -1, inf
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]), inf)
                if r < best_rank:
                    best, best_rank = i, r
            if best < 0:
                break
            left, right = parts[best], parts[best + 1]
            fused, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == left and parts[i + 1] == right:
                    fused.append(left + right)
                    i += 2
                else:
                    fused.append(parts[i])
                    i += 1
            parts = fused
        self._memo[chars] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        text = " ".join(html.unescape(html.unescape(text)).split()).lower()       # whitespace_clean(basic_clean(text)).lower()
        ids: List[int] = []
        for piece in self.split.findall(text):
            if piece in ("<start_of_text>", "<end_of_text>"):
                ids.append(self.token_id[piece])
                continue
            chars = "".join(self.byte_symbol[b] for b in piece.encode("utf-8"))
            ids.extend(self.token_id[t] for t in self._merge_word(chars))
        return ids


def tokenize(texts: Sequence[str], context_length: int = 77, bpe_path: Optional[str] = None) -> torch.Tensor:
    """int64 [len(texts), context_length]: <start> ids <end>, zero padded, truncated with <end> kept (tokenizer.py:159-188)."""
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros((len(texts), context_length), dtype=torch.int64)
    tok = None
    for i, text in enumerate(texts):
        ids: List[int] = []
        if text.strip():
            if tok is None:
                path = bpe_path or os.environ.get("EDTR_CLIP_BPE")
                if not path or not os.path.exists(path):
                    raise RuntimeError("tokenising a non-empty prompt needs the OpenCLIP BPE vocabulary "
                                       "(bpe_simple_vocab_16e6.txt.gz): pass bpe_path= or set EDTR_CLIP_BPE")
                tok = _tokenizer(path)
            ids = tok.encode(text)
        ids = [SOT] + ids + [EOT]
        if len(ids) > context_length:
            ids = ids[:context_length]
            ids[-1] = EOT
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int64)
    return out


@lru_cache()
def _tokenizer(path: str) -> SimpleTokenizer:
    return SimpleTokenizer(path)


# ----------------------------------------------------------------------------------------------
# the module
# ----------------------------------------------------------------------------------------------
class _TextEngine:
    """The text tower for a fixed batch size: static token input, fp32 [B, L, W] output, one program."""

    def __init__(self, owner: "FrozenOpenCLIPEmbedder", B: int):
        dev = owner._device()
        dt = owner.compute_dtype
        cfg = owner.text_cfg
        W, Lc, heads, layers = cfg["width"], cfg["context_length"], cfg["heads"], cfg["layers"]
        if W != heads * 64:
            raise NotImplementedError(f"CLIP text tower: head width {W // heads} (edtr_flash_attn64 needs 64)")
        self.tokens = torch.zeros((B, Lc), dtype=torch.int64, device=dev)
        self.out = torch.zeros((B, Lc, W), dtype=torch.float32, device=dev)
        self.arena = Arena(dev)
        store = WeightStore(owner.flat_params(""), dt, dev)
        self.prog = Program("clip.text")
        em = Emitter(self.prog, self.arena, store, dt)
        rows = B * Lc
        x = em.new(rows, W)
        self.prog.add(ops_mod.make_embed_tokens(dtype=dt, tokens=self.tokens, table=store.raw("model.token_embedding.weight"),
                                                pos=store.raw("model.positional_embedding"), rows=rows, L_ctx=Lc, D=W, out=x, ld=W))
        n_run = layers - owner.layer_idx          # "penultimate": the last block is skipped (model/clip.py:50-58)
        for i in range(n_run):
            p = f"model.transformer.resblocks.{i}."
            h = em.layer_norm(x, rows, W, p + "ln_1.")
            wqk, bqk = store.rows(p + "attn.in_proj_weight", 0, 2 * W, p + "attn.in_proj_bias")
            qk = em.gemm(h, wqk, rows, 2 * W, W, bias=bqk, name="clip.qk")
            wv, _ = store.rows(p + "attn.in_proj_weight", 2 * W, 3 * W)
            bv = store.raw(p + "attn.in_proj_bias")[2 * W:3 * W]
            vt, ldv = em.vt_gemm(wv, h, B=B, Ntok=Lc, Cin=W, bias_m=bv, name="clip.vT")
            o = em.flash(qk[:, :W], qk[:, W:], vt, B=B, H=heads, Nq=Lc, Nk=Lc, k_bs=Lc * qk.stride(0), vt_bs=W * ldv,
                         vt_ld=ldv, causal=True)
            em.free(h, qk, vt)
            wo, bo = store.linear([p + "attn.out_proj.weight"], [p + "attn.out_proj.bias"])
            x1 = em.gemm(o, wo, rows, W, W, bias=bo, residual=x, name="clip.attn_out")
            em.free(o, x)
            h2 = em.layer_norm(x1, rows, W, p + "ln_2.")
            wf, bf = store.linear([p + "mlp.c_fc.weight"], [p + "mlp.c_fc.bias"])
            g = em.gemm(h2, wf, rows, 4 * W, W, bias=bf, act=L.ACT_GELU, name="clip.c_fc")
            em.free(h2)
            wp, bp = store.linear([p + "mlp.c_proj.weight"], [p + "mlp.c_proj.bias"])
            x = em.gemm(g, wp, rows, W, 4 * W, bias=bp, residual=x1, name="clip.c_proj")
            em.free(g, x1)
        y = em.layer_norm(x, rows, W, "model.ln_final.")
        self.prog.add(ops_mod.make_nhwc_to_nchw(dtype=dt, src=y, src_f32=False, B=1, C=1, HW=rows * W, ld=1, dst=self.out,
                                                name="clip.out_f32"))

    def run(self, tokens: torch.Tensor) -> torch.Tensor:
        self.tokens.copy_(tokens)
        self.prog.run()
        return self.out.clone()


class FrozenOpenCLIPEmbedder(ParamTree):
    """reference model/clip.py:12-65."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, embed_dim, vision_cfg, text_cfg, layer="last"):
        assert layer in self.LAYERS
        self.embed_dim, self.vision_cfg, self.text_cfg = embed_dim, dict(vision_cfg), dict(text_cfg)
        super().__init__(clip_text_param_spec(embed_dim, self.text_cfg), unet_like=False)
        self.layer = layer
        self.layer_idx = 0 if layer == "last" else 1
        self.compute_dtype = None            # set by ControlLDM (or default bf16)
        self.bpe_path: Optional[str] = None
        self._engines: Dict[int, _TextEngine] = {}
        self._fingerprint = None
        self._fixed: Optional[torch.Tensor] = None

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def set_embedding(self, emb: Optional[torch.Tensor]) -> None:
        """Optional override: a precomputed [1, L, W] embedding returned by ``encode`` for every prompt (e.g. when no CLIP
        weights are loaded and the constant embedding of the fixed prompt "" comes from elsewhere)."""
        self._fixed = emb

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        if tokens.device.type != "cuda":
            raise RuntimeError("FrozenOpenCLIPEmbedder: the MI355X path runs only on a ROCm GPU; there is no CPU fallback")
        if self.compute_dtype is None:
            self.compute_dtype = torch.bfloat16
        fp = (params_fingerprint(self), self.compute_dtype)
        if fp != self._fingerprint:
            self._engines.clear()
            self._fingerprint = fp
        B = tokens.shape[0]
        if B not in self._engines:
            self._engines[B] = _TextEngine(self, B)
        return self._engines[B].run(tokens)

    encode_with_transformer = forward

    def encode(self, text: List[str]) -> torch.Tensor:
        n = len(text) if isinstance(text, (list, tuple)) else 1
        if self._fixed is not None:
            return self._fixed.expand(n, -1, -1).contiguous()
        tokens = tokenize(text, self.text_cfg["context_length"], self.bpe_path).to(self._device())
        return self(tokens)
