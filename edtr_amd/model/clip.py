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
from typing import Dict, List, Optional, Sequence

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
# tokenizer (restatement of reference model/open_clip/tokenizer.py:20-188; ftfy is skipped like in
# tools/ref_import.py — it only repairs mojibake)
# ----------------------------------------------------------------------------------------------
@lru_cache()
def _bytes_to_unicode() -> Dict[int, str]:
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class SimpleTokenizer:
    def __init__(self, bpe_path: str):
        self.byte_encoder = _bytes_to_unicode()
        merges = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in merges[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<start_of_text>", "<end_of_text>"]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        # \p{L} / \p{N} of the `regex` module restated with the stdlib `re` classes (letters / digits, unicode aware)
        try:        # the reference's pattern needs \p{L} / \p{N} (third-party `regex`); stdlib classes are the fallback
            import regex
            self.pat = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                     regex.IGNORECASE)
        except ImportError:
            self.pat = re.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+|_+", re.IGNORECASE)

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda pr: self.bpe_ranks.get(pr, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new_word, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                    new_word.extend(word[i:j])
                    i = j
                except ValueError:
                    new_word.extend(word[i:])
                    break
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new_word.append(first + second)
                    i += 2
                else:
                    new_word.append(word[i])
                    i += 1
            word = tuple(new_word)
            if len(word) == 1:
                break
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        text = re.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()
        ids: List[int] = []
        for token in self.pat.findall(text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
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
