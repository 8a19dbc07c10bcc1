"""Per-kernel parity tests (run on the MI355X box: pytest -m gpu).  Every HIP kernel is called through the
C ABI (edtr_amd.ops -> libedtr_hip.so) and compared with a plain torch fp32 CPU computation of the same op
on the same 16-bit-rounded inputs.  Stated tolerances (relative L2 error of the output tensor):
  bf16 storage: 3.5e-3     fp16 storage: 4.4e-4
(<= 1.5 x the largest error any kernel measures, 2.34e-3 / 2.95e-4 — the attention kernels; test_zz_measured_error_envelope
prints the current envelope; a 2x regression of any kernel's numerics fails)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
TOL = {torch.bfloat16: 3.5e-3, torch.float16: 4.4e-4}


def _ops():
    from edtr_amd import ops
    return ops


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


MEASURED = {}      # test id -> largest relative error it computed (printed by the last test of this file; EDTR_TEST_ERRLOG=path dumps it)


def rel(a, b):
    import os
    a, b = a.double().cpu(), b.double().cpu()
    v = float((a - b).norm() / b.norm().clamp_min(1e-30))
    key = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    MEASURED[key] = max(MEASURED.get(key, 0.0), v)
    return v


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def nhwc16(x, dtype, pad_to=None):
    """NCHW fp32 cpu -> NHWC 16-bit cuda (optionally zero-padded channels); also returns the rounded fp32 NCHW."""
    xr = x.to(dtype).float()
    t = xr.permute(0, 2, 3, 1).contiguous()
    if pad_to and pad_to > t.shape[-1]:
        t = F.pad(t, (0, pad_to - t.shape[-1]))
    return t.to(dtype).to(dev()).contiguous(), xr


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,tile", [(256, 128, 64, 1), (300, 72, 200, 2), (4096, 320, 320, 0), (77, 640, 1024, 0),
                                        (8, 1280, 320, 2), (130, 136, 72, 1),
                                        # tile 3 = LDS-DMA main loop (K % 64 == 0)
                                        (256, 128, 64, 3), (300, 72, 192, 3), (4096, 320, 320, 3), (130, 136, 128, 3),
                                        (77, 640, 1024, 3), (8, 1280, 320, 3),
                                        # tile 6 = 256x256 tile, 8 waves, ping-pong 8-phase schedule (odd K-tile counts too)
                                        (256, 128, 64, 6), (300, 72, 192, 6), (4096, 320, 320, 6), (130, 136, 128, 6),
                                        (77, 640, 1024, 6), (520, 1280, 320, 6), (1000, 520, 1152, 6),
                                        # tile 8 = 128x160 tile (N = 320 / 640 / 1280 without column padding)
                                        (256, 128, 64, 8), (300, 72, 192, 8), (4096, 320, 320, 8), (130, 136, 128, 8),
                                        (77, 640, 1024, 8), (520, 1280, 320, 8), (1000, 520, 1152, 8), (700, 160, 256, 8)])
def test_gemm_bias_residual(dtype, M, N, K, tile):
    ops = _ops()
    a = rnd((M, K), 1).to(dtype)
    w = rnd((N, K), 2, 1 / math.sqrt(K)).to(dtype)
    bias = rnd((N,), 3)
    res = rnd((M, N), 4).to(dtype)
    ref = a.float() @ w.float().t() + bias + res.float()
    d = dev()
    ad, wd, bd, rd = a.to(d), w.to(d), bias.to(d), res.to(d)
    out = torch.empty((M, N), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=ad, w=wd, out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N, bias_n=bd,
                              residual=rd, ldr=N, tile=tile))
    torch.cuda.synchronize()
    assert rel(out.float(), ref) < TOL[dtype]
    # fp32 output, alpha, SiLU epilogue, no residual
    out32 = torch.empty((M, N), dtype=torch.float32, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=ad, w=wd, out=out32, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N, bias_n=bd,
                              alpha=0.5, act=2, out_f32=True, tile=tile))
    torch.cuda.synchronize()
    ref2 = F.silu(0.5 * (a.float() @ w.float().t()) + bias)
    assert rel(out32, ref2) < 5e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,S,tile", [(512, 256, 1152, 4, 1), (100, 72, 640, 10, 2), (64, 1280, 2880, 7, 1),
                                          (512, 256, 1152, 4, 3), (100, 72, 640, 10, 3), (64, 1280, 2880, 45, 3)])
def test_gemm_splitk(dtype, M, N, K, S, tile):
    """Split-K: fp32 partial slabs + reducer with the full epilogue (bias, row vector, SiLU-free residual)."""
    ops = _ops()
    d = dev()
    a = rnd((M, K), 1).to(dtype)
    w = rnd((N, K), 2, 1 / math.sqrt(K)).to(dtype)
    bias, res, rv = rnd((N,), 3), rnd((M, N), 4).to(dtype), rnd((2, N), 5)
    ref = a.float() @ w.float().t() + bias + rv.repeat_interleave(M // 2, dim=0) + res.float()
    out = torch.empty((M, N), dtype=dtype, device=d)
    ws = torch.empty(S * M * N, dtype=torch.float32, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=a.to(d), w=w.to(d), out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N,
                              bias_n=bias.to(d), residual=res.to(d), ldr=N, rowvec=rv.to(d), rowvec_ld=N,
                              rows_per_image=M // 2, tile=tile, splitk=S, workspace=ws))
    torch.cuda.synchronize()
    assert rel(out.float(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_splitk_matches_plain(dtype):
    ops = _ops()
    d = dev()
    B, H, W, cin, cout = 2, 8, 8, 128, 136
    x = rnd((B, cin, H, W), 20)
    w = rnd((cout, cin, 3, 3), 21, 1 / math.sqrt(9 * cin))
    x16, xr = nhwc16(x, dtype)
    wp = ops.pack_conv_weight(w, dtype).to(d)
    bias = rnd((cout,), 22).to(d)
    ref = F.conv2d(xr, w.to(dtype).float(), bias.cpu(), padding=1)
    outs = []
    for S in (1, 6):
        out = torch.empty((B, H, W, cout), dtype=dtype, device=d)
        ws = torch.empty(S * B * H * W * cout, dtype=torch.float32, device=d) if S > 1 else None
        ops.launch(ops.make_igemm(dtype=dtype, a1=x16, w=wp, out=out, taps=9, M=B * H * W, N=cout, C1=cin, ld1=cin,
                                  ldw=9 * cin, ldc=cout, spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, tile=1, splitk=S,
                                  workspace=ws))
        torch.cuda.synchronize()
        outs.append(out.float().cpu().permute(0, 3, 1, 2))
    assert rel(outs[0], ref) < TOL[dtype]
    assert rel(outs[1], ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_geglu_and_concat(dtype):
    ops = _ops()
    M, d_model, inner = 200, 128, 512
    x = rnd((M, d_model), 5).to(dtype)
    w = rnd((2 * inner, d_model), 6, 1 / math.sqrt(d_model))
    b = rnd((2 * inner,), 7, 0.1)
    perm = ops.geglu_perm(inner)
    wp = w[perm].to(dtype)
    bp = b[perm].contiguous()
    h = x.float() @ w.to(dtype).float().t() + b
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    d = dev()
    out = torch.empty((M, inner), dtype=dtype, device=d)
    for tile in (1, 3):   # register-staged and LDS-DMA main loops
        out.zero_()
        ops.launch(ops.make_igemm(dtype=dtype, a1=x.to(d), w=wp.to(d), out=out, M=M, N=2 * inner, C1=d_model,
                                  ld1=d_model, ldw=d_model, ldc=inner, bias_n=bp.to(d), act=1, tile=tile))
        torch.cuda.synchronize()
        assert rel(out.float(), ref) < TOL[dtype], tile
    # K-concat of two sources (1x1 conv over cat([h, skip]))
    x2 = rnd((M, 64), 8).to(dtype)
    wc = rnd((72, d_model + 64), 9, 0.1).to(dtype)
    refc = torch.cat([x.float(), x2.float()], dim=1) @ wc.float().t()
    outc = torch.empty((M, 72), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=x.to(d), a2=x2.to(d), w=wc.to(d), out=outc, M=M, N=72, C1=d_model, C2=64,
                              ld1=d_model, ld2=64, ldw=d_model + 64, ldc=72))
    torch.cuda.synchronize()
    assert rel(outc.float(), refc) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_strided(dtype):
    """z-batched forms used by the VAE attention: S = alpha Q K^T (fp32 out), V^T = Wv X^T + bias_m, O = P V."""
    ops = _ops()
    B, Ntok, Cc = 2, 136, 64
    d = dev()
    q = rnd((B, Ntok, Cc), 10).to(dtype)
    k = rnd((B, Ntok, Cc), 11).to(dtype)
    s = torch.empty((B, Ntok, Ntok), dtype=torch.float32, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=q.to(d), w=k.to(d), out=s, M=Ntok, N=Ntok, C1=Cc, ld1=Cc, ldw=Cc, ldc=Ntok,
                              Z=B, a_zs=(Ntok * Cc, 0), w_zs=(Ntok * Cc, 0), o_zs=(Ntok * Ntok, 0), alpha=0.125,
                              out_f32=True))
    torch.cuda.synchronize()
    assert rel(s, 0.125 * q.float() @ k.float().transpose(1, 2)) < 2e-5
    # V^T with a shared weight (zero z-stride on A) and per-row bias; Ntok not a multiple of 8 -> n_valid
    Nt2 = 77
    x = rnd((B, Nt2, Cc), 12).to(dtype)
    wv = rnd((96, Cc), 13, 0.2).to(dtype)
    bm = rnd((96,), 14)
    ldv = 80
    vt = torch.full((B, 96, ldv), 7.0, dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=wv.to(d), w=x.to(d), out=vt, M=96, N=ldv, n_valid=Nt2, C1=Cc, ld1=Cc, ldw=Cc,
                              ldc=ldv, Z=B, a_zs=(0, 0), w_zs=(Nt2 * Cc, 0), o_zs=(96 * ldv, 0), bias_m=bm.to(d)))
    torch.cuda.synchronize()
    refv = wv.float() @ x.float().transpose(1, 2) + bm[None, :, None]
    assert rel(vt[:, :, :Nt2].float(), refv) < TOL[dtype]
    assert float((vt[:, :, Nt2:].float() - bm.to(d)[None, :, None]).abs().max()) < 2e-2  # zero rows + bias only


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["s1", "s2", "vae_down", "up", "concat", "small_cin", "small_cout",
                                  "s1_dma", "s2_dma", "vae_down_dma", "up_dma", "small_cout_dma",
                                  "s1_256", "s2_256", "vae_down_256", "up_256", "small_cout_256",
                                  "s1_n160", "s2_n160", "vae_down_n160", "up_n160", "small_cout_n160"])
def test_conv3x3(dtype, case):
    tile = 0
    if case.endswith("_dma"):
        case, tile = case[:-4], 3
    if case.endswith("_256"):
        case, tile = case[:-4], 6
    if case.endswith("_n160"):
        case, tile = case[:-5], 8
    ops = _ops()
    d = dev()
    B, H, W = 2, 12, 20
    cin, cout = 64, 72
    stride, pad, ups, pad_tl = 1, 1, False, 1
    if case == "small_cin":
        cin = 4
    if case == "small_cout":
        cout = 3
    x = rnd((B, cin, H, W), 20)
    w = rnd((cout, cin, 3, 3), 21, 1 / math.sqrt(9 * cin))
    bias = rnd((cout,), 22)
    xr16, xr = nhwc16(x, dtype, pad_to=ops.round_up(cin, 8))
    wr = w.to(dtype).float()
    x2 = None
    if case == "s1" or case.startswith("small"):
        ref = F.conv2d(xr, wr, bias, padding=1)
    elif case == "s2":
        stride = 2
        ref = F.conv2d(xr, wr, bias, stride=2, padding=1)
    elif case == "vae_down":   # pad (0,1,0,1) then stride 2, pad 0
        stride, pad_tl = 2, 0
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, bias, stride=2, padding=0)
    elif case == "up":
        ups = True
        ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wr, bias, padding=1)
    elif case == "concat":
        x2 = rnd((B, 40, H, W), 23)
        w = rnd((cout, cin + 40, 3, 3), 24, 1 / math.sqrt(9 * (cin + 40)))
        wr = w.to(dtype).float()
        x216, x2r = nhwc16(x2, dtype)
        ref = F.conv2d(torch.cat([xr, x2r], dim=1), wr, bias, padding=1)
    OH, OW = ref.shape[2], ref.shape[3]
    cinp = ops.round_up(cin, 8)
    c2 = 40 if case == "concat" else 0
    if case == "concat":
        wp = ops.pack_conv_weight(w, dtype).to(d)
    else:
        wp = ops.pack_conv_weight(w, dtype, cin_pad=cinp).to(d)
    N = wp.shape[0]
    bp = ops.pad_bias(bias, N).to(d)
    emb = rnd((B, N), 25).to(d)
    res = rnd((B, OH, OW, N), 26).to(dtype).to(d)
    out = torch.empty((B, OH, OW, N), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=xr16, a2=(x216 if case == "concat" else None), w=wp, out=out, taps=9,
                              M=B * OH * OW, N=N, C1=cinp, C2=c2, ld1=cinp, ld2=c2, ldw=wp.shape[1], ldc=N,
                              spatial=(H, W, OH, OW, stride, pad_tl, pad_tl, int(ups)), bias_n=bp, rowvec=emb,
                              rowvec_ld=N, rows_per_image=OH * OW, residual=res, ldr=N, tile=tile))
    torch.cuda.synchronize()
    got = out.float().cpu()[..., :cout].permute(0, 3, 1, 2)
    full_ref = ref + emb.cpu()[:, :cout, None, None] + res.float().cpu()[..., :cout].permute(0, 3, 1, 2)
    assert rel(got, full_ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 5, 256, 256), (1, 2, 1024, 77), (2, 1, 64, 64), (1, 3, 4, 4), (1, 1, 200, 333)])
def test_flash_attn64(dtype, B, H, Nq, Nk):
    ops = _ops()
    d = dev()
    Cc = H * 64
    q = rnd((B, Nq, Cc), 30).to(dtype)
    k = rnd((B, Nk, Cc), 31).to(dtype)
    v = rnd((B, Nk, Cc), 32).to(dtype)
    ldv = ops.round_up(Nk, 8)
    vt = torch.zeros((B, Cc, ldv), dtype=dtype)
    vt[:, :, :Nk] = v.transpose(1, 2)
    out = torch.empty((B, Nq, Cc), dtype=dtype, device=d)
    scale = 0.125
    ops.launch(ops.make_flash_attn(dtype=dtype, q=q.to(d), k=k.to(d), vt=vt.to(d), out=out, B=B, H=H, Nq=Nq, Nk=Nk,
                                   q_bs=Nq * Cc, q_ld=Cc, k_bs=Nk * Cc, k_ld=Cc, vt_bs=Cc * ldv, vt_ld=ldv, o_bs=Nq * Cc,
                                   o_ld=Cc, scale=scale))
    torch.cuda.synchronize()

    def heads(t):
        return t.float().reshape(B, -1, H, 64).transpose(1, 2)

    p = torch.softmax(heads(q) @ heads(k).transpose(-1, -2) * scale, dim=-1)
    ref = (p @ heads(v)).transpose(1, 2).reshape(B, Nq, Cc)
    assert rel(out.float(), ref) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk,prescaled,pipeline_layout", [
    (2, 5, 4096, 77, True, True),       # the 64x64-latent cross-attention: two 32-query blocks per wave, keys 77 of 128
    (8, 20, 256, 77, True, True),       # 16x16 latents: one block per wave
    (3, 2, 1000, 128, False, False),    # Nk = 128 exactly (all four key blocks), ragged Nq, unscaled scores
    (1, 3, 70, 64, False, False),       # one key tile; a workgroup whose last waves have no block
    (2, 1, 33, 1, True, False),         # a single key: softmax == 1
    (1, 20, 16384, 33, True, False),    # four blocks per wave (the per-workgroup walk), keys end inside the second block of 32
])
def test_flash_attn64_small_nk_resident_kv(dtype, B, H, Nq, Nk, prescaled, pipeline_layout):
    """Nk <= 128 (the cross-attention over the 77 context tokens, reference model/attention.py:171-203): K / V^T resident in LDS,
    two-pass softmax in registers.  pipeline_layout: keys are a column slice of a wider [B * Nk, sumC] matrix and V^T a row slice
    of [B, sumC, ldv], as nets.emit_context_kv lays them out; EDTR_ATTN_SMALLK=0 is the generic kernel (A/B)."""
    ops = _ops()
    d = dev()
    Cc = H * 64
    sumC = Cc * 2 + 64 if pipeline_layout else Cc
    off = 64 if pipeline_layout else 0
    q = rnd((B, Nq, Cc), 230, 0.42 if prescaled else 1.0).to(dtype)
    kall = rnd((B, Nk, sumC), 231, 0.42 if prescaled else 1.0).to(dtype)
    vall = rnd((B, Nk, sumC), 232).to(dtype)
    ldv = ops.round_up(Nk, 8)
    vt = torch.zeros((B, sumC, ldv), dtype=dtype)
    vt[:, :, :Nk] = vall.transpose(1, 2)
    kd, vtd = kall.to(d), vt.to(d)
    out = torch.full((B, Nq, Cc), float("nan"), dtype=dtype, device=d)
    scale = 0.125
    ops.launch(ops.make_flash_attn(dtype=dtype, q=q.to(d), k=kd[:, :, off:off + Cc], vt=vtd[:, off:off + Cc], out=out, B=B, H=H, Nq=Nq, Nk=Nk,
                                   q_bs=Nq * Cc, q_ld=Cc, k_bs=Nk * sumC, k_ld=sumC, vt_bs=sumC * ldv, vt_ld=ldv, o_bs=Nq * Cc,
                                   o_ld=Cc, scale=scale, prescaled=prescaled))
    torch.cuda.synchronize()
    k, v = kall[:, :, off:off + Cc], vall[:, :, off:off + Cc]

    def heads(t):
        return t.float().reshape(B, -1, H, 64).transpose(1, 2)

    logits = heads(q) @ heads(k).transpose(-1, -2) * (math.log(2.0) if prescaled else scale)
    ref = (torch.softmax(logits, dim=-1) @ heads(v)).transpose(1, 2).reshape(B, Nq, Cc)
    assert torch.isfinite(out.float()).all()
    assert rel(out.float(), ref) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk,growth", [(1, 5, 4096, 4096, 1.0),     # the hot shape: 64x64 latents, 5 heads (large-N kernel v3)
                                             (2, 3, 2100, 512, 1.0),      # ragged Nq (store predicate), 2 unrolled trips
                                             (1, 2, 2048, 1024, 5.0),     # later keys beat the first tile's row maximum by > 14 octaves
                                             (1, 2, 2048, 320, 1.0),      # Nk % 256 != 0 -> v1 kernel on prescaled scores
                                             (2, 2, 300, 256, 1.0)])      # below the large-N threshold -> v1 on prescaled scores
def test_flash_attn64_prescaled(dtype, B, H, Nq, Nk, growth):
    """q_prescaled: q.k already carries scale * log2(e) (folded into the projection GEMMs' epilogues by nets.py); the kernels
    evaluate exp2(q.k - max).  Covers the generated-asm large-N kernel, its re-maximise slow path and the v1 fallbacks."""
    ops = _ops()
    d = dev()
    Cc = H * 64
    q = (rnd((B, Nq, Cc), 36) * 0.42).to(dtype)
    k = rnd((B, Nk, Cc), 37) * 0.42
    k[:, Nk // 2:] *= growth
    k = k.to(dtype)
    v = rnd((B, Nk, Cc), 38).to(dtype)
    vt = v.transpose(1, 2).contiguous()
    out = torch.empty((B, Nq, Cc), dtype=dtype, device=d)
    ops.launch(ops.make_flash_attn(dtype=dtype, q=q.to(d), k=k.to(d), vt=vt.to(d), out=out, B=B, H=H, Nq=Nq, Nk=Nk,
                                   q_bs=Nq * Cc, q_ld=Cc, k_bs=Nk * Cc, k_ld=Cc, vt_bs=Cc * Nk, vt_ld=Nk, o_bs=Nq * Cc,
                                   o_ld=Cc, scale=0.125, prescaled=True))
    torch.cuda.synchronize()

    def heads(t):
        return t.double().reshape(B, -1, H, 64).transpose(1, 2)

    p = torch.softmax(heads(q) @ heads(k).transpose(-1, -2) * math.log(2.0), dim=-1)
    ref = (p @ heads(v)).transpose(1, 2).reshape(B, Nq, Cc)
    assert torch.isfinite(out.float()).all()
    assert rel(out.float(), ref) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", DTYPES)
def test_flash_attn64_forced_rescale(dtype):
    """Online-softmax rescale branch: a late key dominates one query row (max jumps in the last tile)."""
    ops = _ops()
    d = dev()
    B, H, Nq, Nk, Cc = 1, 1, 128, 256, 64
    q = rnd((B, Nq, Cc), 33).to(dtype)
    k = rnd((B, Nk, Cc), 34).to(dtype)
    v = rnd((B, Nk, Cc), 35).to(dtype)
    k[0, 250] = (q[0, 17].float() * 4).to(dtype)   # spike for query 17 in the 4th tile
    k[0, 3] = (q[0, 90].float() * 4).to(dtype)     # and one in the 1st tile
    vt = v.transpose(1, 2).contiguous()
    out = torch.empty((B, Nq, Cc), dtype=dtype, device=d)
    ops.launch(ops.make_flash_attn(dtype=dtype, q=q.to(d), k=k.to(d), vt=vt.to(d), out=out, B=B, H=H, Nq=Nq, Nk=Nk,
                                   q_bs=Nq * Cc, q_ld=Cc, k_bs=Nk * Cc, k_ld=Cc, vt_bs=Cc * Nk, vt_ld=Nk, o_bs=Nq * Cc,
                                   o_ld=Cc, scale=0.125))
    torch.cuda.synchronize()
    p = torch.softmax(q.float() @ k.float().transpose(1, 2) * 0.125, dim=-1)
    ref = p @ v.float()
    assert rel(out.float(), ref) < TOL[dtype] * 1.5
    assert (out.float().cpu()[0, 17] - ref[0, 17]).abs().max() < 0.05


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,HW,eps,silu", [(320, 1024, 1e-5, True), (32, 300, 1e-6, True), (128, 4096, 1e-6, False),
                                          (2560, 64, 1e-5, True), (1920, 256, 1e-5, True), (64, 4, 1e-5, True)])
def test_groupnorm(dtype, C, HW, eps, silu):
    ops = _ops()
    d = dev()
    B = 2
    x = rnd((B, C, HW, 1), 40) * 2 + 0.7
    gamma, beta = 1 + 0.1 * rnd((C,), 41), 0.1 * rnd((C,), 42)
    x16, xr = nhwc16(x, dtype)
    sums = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    y = torch.empty_like(x16)
    st, ap = ops.make_gn(dtype=dtype, x=x16, ldx=C, B=B, HW=HW, C=C, sums=sums, gamma=gamma.to(d), beta=beta.to(d),
                         eps=eps, silu=silu, y=y, ldy=C)
    ops.launch(st)
    ops.launch(ap)
    torch.cuda.synchronize()
    ref = F.group_norm(xr, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert rel(got, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 1280), (5, 64), (300, 640)])
def test_layernorm(dtype, rows, C):
    ops = _ops()
    d = dev()
    x = (rnd((rows, C), 50) * 1.5 + 0.3).to(dtype)
    gamma, beta = 1 + 0.1 * rnd((C,), 51), 0.1 * rnd((C,), 52)
    y = torch.empty((rows, C), dtype=dtype, device=d)
    ops.launch(ops.make_layernorm(dtype=dtype, x=x.to(d), rows=rows, C=C, ldx=C, gamma=gamma.to(d), beta=beta.to(d),
                                  eps=1e-5, y=y, ldy=C))
    torch.cuda.synchronize()
    assert rel(y.float(), F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(64, 4096), (10, 136), (3, 16384)])
def test_softmax_rows(dtype, rows, cols):
    ops = _ops()
    d = dev()
    s = rnd((rows, cols), 60) * 3
    p = torch.empty((rows, cols), dtype=dtype, device=d)
    ops.launch(ops.make_softmax_rows(dtype=dtype, s=s.to(d), rows=rows, cols=cols, ld_s=cols, p=p, ld_p=cols))
    torch.cuda.synchronize()
    assert rel(p.float(), torch.softmax(s, dim=-1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_and_elementwise(dtype):
    ops = _ops()
    d = dev()
    B, Cc, H, W = 2, 4, 10, 13
    x = rnd((B, Cc, H, W), 70)
    dst = torch.full((B, H * W, 16), 9.0, dtype=dtype, device=d)
    ops.launch(ops.make_nchw_to_nhwc(dtype=dtype, src=x.to(d), B=B, C=Cc, HW=H * W, dst=dst, ld=16, coff=8,
                                     zero_pad_to=8, scale=2.0, shift=-1.0))
    torch.cuda.synchronize()
    ref = (x * 2 - 1).permute(0, 2, 3, 1).reshape(B, H * W, Cc)
    assert rel(dst[:, :, 8:12].float(), ref) < TOL[dtype]
    assert float(dst[:, :, 12:16].float().abs().max()) == 0.0
    assert float((dst[:, :, :8].float() - 9.0).abs().max()) == 0.0
    back = torch.empty((B, Cc, H * W), dtype=torch.float32, device=d)
    ops.launch(ops.make_nhwc_to_nchw(dtype=dtype, src=dst[:, :, 8:], src_f32=False, B=B, C=Cc, HW=H * W, ld=16, dst=back,
                                     scale=0.5))
    torch.cuda.synchronize()
    assert rel(back.reshape(B, Cc, H, W), 0.5 * (x * 2 - 1)) < TOL[dtype]
    # fp32 NHWC source
    src32 = rnd((B, H * W, 8), 71).to(d)
    back2 = torch.empty((B, 3, H * W), dtype=torch.float32, device=d)
    ops.launch(ops.make_nhwc_to_nchw(dtype=dtype, src=src32, src_f32=True, B=B, C=3, HW=H * W, ld=8, dst=back2))
    torch.cuda.synchronize()
    assert rel(back2, src32[:, :, :3].permute(0, 2, 1)) < 1e-7
    # add / strided copy into a column slice
    a, b = rnd((1000, 24), 72).to(dtype), rnd((1000, 24), 73).to(dtype)
    o = torch.zeros((1000, 40), dtype=dtype, device=d)
    ops.launch(ops.make_add(dtype=dtype, a=a.to(d), lda=24, b=b.to(d), ldb=24, out=o[:, 16:], ldo=40, rows=1000, C=24))
    ops.launch(ops.make_add(dtype=dtype, a=a.to(d)[:, 8:], lda=24, b=None, ldb=0, out=o, ldo=40, rows=1000, C=16))
    torch.cuda.synchronize()
    assert rel(o[:, 16:].float(), a.float() + b.float()) < TOL[dtype]
    assert rel(o[:, :16].float(), a[:, 8:].float()) == 0.0
    # timestep embedding
    t = torch.tensor([50, 100, 150, 200, 999], dtype=torch.int64)
    e = torch.empty((5, 320), dtype=dtype, device=d)
    ops.launch(ops.make_timestep_embedding(dtype=dtype, t=t.to(d), B=5, dim=320, out=e, ld=320))
    torch.cuda.synchronize()
    freqs = torch.exp(-math.log(10000.0) * torch.arange(160, dtype=torch.float32) / 160)
    args = t[:, None].float() * freqs[None]
    assert rel(e.float(), torch.cat([torch.cos(args), torch.sin(args)], dim=-1)) < TOL[dtype]


def test_sampler_kernels():
    ops = _ops()
    d = dev()
    n = 2 * 4 * 16 * 16
    x, eps, noise = rnd((n,), 80), rnd((n,), 81), rnd((n,), 82)
    xp = torch.empty(n, device=d)
    p0 = torch.empty(n, device=d)
    coefs = (1.0990925, 0.45607486, 0.40786713, 0.59105539, math.sqrt(0.045623116))
    ops.launch(ops.make_sampler_update(x=x.to(d), eps=eps.to(d), noise=noise.to(d), coefs=coefs, x_prev=xp, pred_x0=p0, n=n))
    o = torch.empty(n, device=d)
    ops.launch(ops.make_axpby(x=x.to(d), y=noise.to(d), a=0.86815441, b=0.49629423, out=o, n=n))
    torch.cuda.synchronize()
    rp0 = coefs[0] * x - coefs[1] * eps
    assert rel(p0, rp0) < 1e-6
    assert rel(xp, coefs[2] * rp0 + coefs[3] * x + coefs[4] * noise) < 1e-6
    assert rel(o, 0.86815441 * x + 0.49629423 * noise) < 1e-6
    # tile accumulate + divide
    out = torch.zeros((1, 4, 16, 24), device=d)
    cnt = torch.zeros_like(out)
    tile = rnd((1, 4, 8, 8), 83)
    wts = rnd((8, 8), 84).abs() + 0.1
    ops.launch(ops.make_tile_accumulate(tile=tile.to(d), wts=wts.to(d), out=out, count=cnt, B=1, C=4, H=16, W=24, th=8, tw=8,
                                        hi=4, wi=16))
    torch.cuda.synchronize()
    ref = torch.zeros(1, 4, 16, 24)
    ref[..., 4:12, 16:24] = tile * wts
    assert rel(out, ref) < 1e-6
    assert rel(cnt[0, 0, 4:12, 16:24], wts) < 1e-6


def test_error_codes():
    ops = _ops()
    from edtr_amd import lib as L
    d = dev()
    a = torch.zeros((16, 12), dtype=torch.bfloat16, device=d)
    with pytest.raises(RuntimeError, match="EDTR_E_ALIGN"):
        ops.launch(ops.make_igemm(dtype=torch.bfloat16, a1=a, w=a, out=a, M=16, N=16, C1=12, ld1=12, ldw=12, ldc=16))
    with pytest.raises(RuntimeError, match="EDTR_E_NULL"):
        L.check(L.load().edtr_igemm(None, None), "igemm")


def test_graph_capture_replay():
    """A launch sequence captured into a hipGraph replays with the same result."""
    ops = _ops()
    from edtr_amd import lib as L
    import ctypes as C
    d = dev()
    n = 4096
    x = torch.ones(n, device=d)
    y = torch.full((n,), 2.0, device=d)
    o1 = torch.empty(n, device=d)
    o2 = torch.empty(n, device=d)
    r1 = ops.make_axpby(x=x, y=y, a=1.0, b=1.0, out=o1, n=n)
    r2 = ops.make_axpby(x=o1, y=y, a=2.0, b=1.0, out=o2, n=n)
    s = torch.cuda.Stream()
    lib = L.load()
    with torch.cuda.stream(s):
        L.check(lib.edtr_graph_begin(s.cuda_stream), "graph_begin")
        r1.launch(s.cuda_stream)
        r2.launch(s.cuda_stream)
        g = C.c_void_p()
        L.check(lib.edtr_graph_end(s.cuda_stream, C.byref(g)), "graph_end")
        x.fill_(3.0)
        L.check(lib.edtr_graph_launch(g, s.cuda_stream), "graph_launch")
    s.synchronize()
    assert float(o2[0]) == (3.0 + 2.0) * 2 + 2.0
    L.check(lib.edtr_graph_destroy(g), "graph_destroy")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,tile", [(64, 0), (64, 3), (40, 1), (64, 6), (64, 8)])
def test_fused_groupnorm_partials(dtype, cin, tile):
    """The igemm epilogue's per-tile column sums + edtr_gn_finalize reproduce edtr_gn_stats on the stored tensor."""
    ops = _ops()
    d = dev()
    B, H, W, cout = 2, 16, 24, 96          # HW = 384 = 3 row tiles per image
    x = rnd((B, cin, H, W), 90)
    w = rnd((cout, cin, 3, 3), 91, 1 / math.sqrt(9 * cin))
    x16, _ = nhwc16(x, dtype)
    wp = ops.pack_conv_weight(w, dtype).to(d)
    bias = rnd((cout,), 92).to(d)
    M = B * H * W
    out = torch.empty((M, cout), dtype=dtype, device=d)
    gnp = torch.full((M // 128, cout, 2), 123.0, dtype=torch.float32, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=x16, w=wp, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin,
                              ldc=cout, spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, tile=tile, gn_partial=gnp))
    s_fused = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    ops.launch(ops.make_gn_finalize(partial=gnp, tiles_per_image=(H * W) // 128, B=B, C=cout, sums=s_fused))
    s_ref = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    g = torch.ones(cout, device=d)
    st, _ = ops.make_gn(dtype=dtype, x=out, ldx=cout, B=B, HW=H * W, C=cout, sums=s_ref, gamma=g, beta=g, eps=1e-5,
                        silu=False, y=torch.empty_like(out), ldy=cout)
    ops.launch(st)
    torch.cuda.synchronize()
    # the fused sums use the values BEFORE the 16-bit rounding of the store: agreement to rounding noise
    assert rel(s_fused[..., 1], s_ref[..., 1]) < (2e-3 if dtype == torch.bfloat16 else 3e-4)
    assert float((s_fused[..., 0] - s_ref[..., 0]).abs().max()) < (0.5 if dtype == torch.bfloat16 else 0.06)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,cin,cout,sk,res", [(4, 8, 8, 128, 160, 2, True), (8, 8, 8, 1280, 1280, 10, True), (2, 16, 16, 192, 128, 3, False),
                                                   (2, 16, 16, 1280, 1280, 3, True), (1, 16, 24, 64, 96, 2, False)])
def test_split_k_reducer_writes_the_groupnorm_partials(dtype, B, H, W, cin, cout, sk, res):
    """Round 6: with split-K the REDUCER writes the output's GroupNorm statistics (gn_slot_rows: 128-row slots, 64-row slots for the
    8 x 8 images of the deepest latent level) — the finalize of its slots reproduces edtr_gn_stats on the stored tensor, the tensor
    itself is the one the plain reducer writes (bit for bit), and the apply launch folds the slots itself where they are few."""
    ops = _ops()
    d = dev()
    hw = H * W
    sr = ops.gn_slot_rows(hw)
    assert sr in (64, 128)
    x = rnd((B, cin, H, W), 190)
    w = rnd((cout, cin, 3, 3), 191, 1 / math.sqrt(9 * cin))
    x16, _ = nhwc16(x, dtype)
    wp = ops.pack_conv_weight(w, dtype).to(d)
    bias = rnd((cout,), 192).to(d)
    M = B * hw
    r16 = rnd((M, cout), 193).to(dtype).to(d) if res else None
    ws = torch.empty((sk * M * cout,), dtype=torch.float32, device=d)
    kw = dict(dtype=dtype, a1=x16, w=wp, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout, spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias,
              residual=r16, ldr=cout, splitk=sk, workspace=ws, rows_per_image=hw)
    plain = torch.empty((M, cout), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(out=plain, **kw))
    out = torch.empty((M, cout), dtype=dtype, device=d)
    gnp = torch.full((M // sr, cout, 2), 123.0, dtype=torch.float32, device=d)
    assert ops.gn_fusable(M, cout, cin, hw, splitk=sk)
    ops.launch(ops.make_igemm(out=out, gn_partial=gnp, gn_slot_rows=sr, **kw))
    torch.cuda.synchronize()
    assert torch.equal(out, plain), "the statistics reducer must store what the plain reducer stores"
    s_fused = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    ops.launch(ops.make_gn_finalize(partial=gnp, tiles_per_image=hw // sr, B=B, C=cout, sums=s_fused))
    s_ref = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    g1 = (1 + 0.1 * rnd((cout,), 194)).to(d)
    b1 = (0.1 * rnd((cout,), 195)).to(d)
    y_ref = torch.empty_like(out)
    st, ap = ops.make_gn(dtype=dtype, x=out, ldx=cout, B=B, HW=hw, C=cout, sums=s_ref, gamma=g1, beta=b1, eps=1e-5, silu=True, y=y_ref, ldy=cout)
    ops.launch(st)
    ops.launch(ap)
    torch.cuda.synchronize()
    # the fused sums use the values BEFORE the 16-bit rounding of the store: agreement to rounding noise
    assert rel(s_fused[..., 1], s_ref[..., 1]) < (2e-3 if dtype == torch.bfloat16 else 3e-4)
    assert float((s_fused[..., 0] - s_ref[..., 0]).abs().max()) < (0.5 if dtype == torch.bfloat16 else 0.06) * max(1.0, hw * (cout // 32) / 1152.0)
    if ops.gn_foldable(hw, cout, tiles=hw // sr):      # the apply launch folds the slots itself
        y = torch.empty_like(out)
        _, ap2 = ops.make_gn(dtype=dtype, x=out, ldx=cout, B=B, HW=hw, C=cout, sums=None, gamma=g1, beta=b1, eps=1e-5, silu=True, y=y, ldy=cout,
                             partial=gnp, tiles_per_image=hw // sr)
        ops.launch(ap2)
        torch.cuda.synchronize()
        assert rel(y.float(), y_ref.float()) < TOL[dtype]
    # slot sizes the library does not know, or a main-loop launch asked for 64-row slots: refused, not mis-indexed
    with pytest.raises(RuntimeError):
        ops.launch(ops.make_igemm(out=out, gn_partial=gnp, gn_slot_rows=32, **kw))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C1,C2,sk", [(2, 16, 16, 128, 64, 1), (4, 8, 8, 1280, 1280, 10), (2, 32, 32, 320, 640, 1), (2, 16, 16, 1280, 640, 3)])
def test_concat_halves_share_one_buffer_of_groupnorm_slots(dtype, B, H, W, C1, C2, sk):
    """Round 6: the two halves of torch.cat([h, skip + control]) (reference model/controlnet.py:35-37) are written by different launches — a
    convolution into the left column slice (edtr_igemm gn_ld: its partials go to ITS columns of a shared slot buffer; with split-K the
    reducer writes them) and edtr_add_stats into the right one — and the GroupNorm of the concatenation reads the statistics from that
    buffer: finalize(shared slots) == edtr_gn_stats on the concatenated tensor."""
    ops = _ops()
    d = dev()
    hw, Ct = H * W, C1 + C2
    M = B * hw
    slot = ops.gn_slot_rows(hw) if sk > 1 else 128
    if hw % slot:
        pytest.skip("main-loop statistics need whole 128-row slots per image")
    cat = torch.full((M, Ct), float("nan"), dtype=dtype, device=d)
    slots = torch.full((M // slot, Ct, 2), 321.0, dtype=torch.float32, device=d)
    # left half: a 3 x 3 convolution writing columns [0, C1)
    cin = 128
    x16, _ = nhwc16(rnd((B, cin, H, W), 210), dtype)
    wp = ops.pack_conv_weight(rnd((C1, cin, 3, 3), 211, 1 / math.sqrt(9 * cin)), dtype).to(d)
    ws = torch.empty((sk * M * C1,), dtype=torch.float32, device=d) if sk > 1 else None
    ops.launch(ops.make_igemm(dtype=dtype, a1=x16, w=wp, out=cat[:, :C1], taps=9, M=M, N=C1, C1=cin, ld1=cin, ldw=9 * cin, ldc=Ct,
                              spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=rnd((C1,), 212).to(d), splitk=sk, workspace=ws, rows_per_image=hw,
                              gn_partial=slots.reshape(-1), gn_ld=Ct, gn_slot_rows=slot if sk > 1 else 0))
    # right half: skip + control into columns [C1, Ct)
    a = rnd((M, C2), 213).to(dtype).to(d)
    b = rnd((M, C2), 214).to(dtype).to(d)
    ops.launch(ops.make_add_stats(dtype=dtype, a=a, lda=C2, b=b, ldb=C2, out=cat[:, C1:], ldo=Ct, rows=M, C=C2,
                                  gn_partial=slots.reshape(-1)[2 * C1:], gn_ld=Ct, slot_rows=slot))
    torch.cuda.synchronize()
    assert torch.isfinite(cat.float()).all()
    assert rel(cat[:, C1:].float(), (a.float() + b.float())) < TOL[dtype]
    s_fused = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    ops.launch(ops.make_gn_finalize(partial=slots, tiles_per_image=hw // slot, B=B, C=Ct, sums=s_fused))
    s_ref = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    g = torch.ones(Ct, device=d)
    st, _ = ops.make_gn(dtype=dtype, x=cat, ldx=Ct, B=B, HW=hw, C=Ct, sums=s_ref, gamma=g, beta=g, eps=1e-5, silu=False, y=torch.empty_like(cat), ldy=Ct)
    ops.launch(st)
    torch.cuda.synchronize()
    assert rel(s_fused[..., 1], s_ref[..., 1]) < (2e-3 if dtype == torch.bfloat16 else 3e-4)
    assert float((s_fused[..., 0] - s_ref[..., 0]).abs().max()) < (0.5 if dtype == torch.bfloat16 else 0.06) * max(1.0, hw * (Ct // 32) / 1152.0)
    with pytest.raises(RuntimeError):          # a slot stride narrower than the launch's own columns
        ops.launch(ops.make_igemm(dtype=dtype, a1=x16, w=wp, out=cat[:, :C1], taps=9, M=M, N=C1, C1=cin, ld1=cin, ldw=9 * cin, ldc=Ct,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), splitk=sk, workspace=ws, rows_per_image=hw, gn_partial=slots.reshape(-1),
                                  gn_ld=C1 - 32, gn_slot_rows=slot if sk > 1 else 0))
    with pytest.raises(RuntimeError):          # rows that are not whole slots
        ops.launch(ops.make_add_stats(dtype=dtype, a=a, lda=C2, b=None, ldb=0, out=cat[:, C1:], ldo=Ct, rows=M - 8, C=C2,
                                      gn_partial=slots.reshape(-1)[2 * C1:], gn_ld=Ct, slot_rows=slot))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,C,tile", [(2, 1024, 320, 0), (2, 256, 640, 0), (3, 64, 1280, 0), (1, 4096, 320, 8), (2, 64, 128, 3), (2, 200, 128, 1),
                                        (1, 72, 64, 0)])
def test_fused_qkv_projection_with_transposed_v(dtype, B, N, C, tile):
    """One edtr_igemm launch for [Wq; Wk; Wv]: q / k row-major (scaled by alpha), v^T[image][channel][token] through the
    transposed second output — against the separate products."""
    ops = _ops()
    d = dev()
    M = B * N
    x = rnd((M, C), 150).to(dtype)
    w = rnd((3 * C, C), 151, 1 / math.sqrt(C))
    wp = ops.pack_linear_weight(w, dtype).to(d)
    qk = torch.full((M, 2 * C), 9.0, dtype=dtype, device=d)
    vt = torch.full((B * C, N), 9.0, dtype=dtype, device=d)
    rec = ops.make_igemm(dtype=dtype, a1=x.to(d), w=wp, out=qk, M=M, N=3 * C, C1=C, ld1=C, ldw=C, ldc=2 * C, alpha=0.37, rows_per_image=N,
                         vt_out=vt, vt_col0=2 * C, vt_ld=N, vt_alpha=1.0, tile=tile)
    if (2 * C) % 128 != 0 and (2 * C) % 160 != 0:
        with pytest.raises(RuntimeError):
            ops.launch(rec)
        return
    ops.launch(rec)
    torch.cuda.synchronize()
    full = x.float() @ w.to(dtype).float().t()
    assert rel(qk.float(), 0.37 * full[:, :2 * C]) < TOL[dtype]
    v_ref = full[:, 2 * C:].reshape(B, N, C).permute(0, 2, 1).reshape(B * C, N)
    assert rel(vt.float(), v_ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,tile", [(300, 320, 128, 0), (32768, 640, 320, 8), (4096, 320, 1280, 8), (512, 1280, 192, 3), (130, 96, 64, 1), (2048, 512, 1024, 0), (77, 160, 64, 2)])
def test_gemm_writes_row_statistics_of_its_output(dtype, M, N, K, tile):
    """Producer side of the folded LayerNorm: per row, sum and sum of squares of the stored values in N / 32 slots (a column
    tile fills its first slot and zeroes the rest it covers), whatever tile runs; bias + residual included."""
    ops = _ops()
    d = dev()
    x, w = rnd((M, K), 160).to(dtype), rnd((N, K), 161, 1 / math.sqrt(K)).to(dtype)
    bias, res = rnd((N,), 162), rnd((M, N), 163).to(dtype)
    out = torch.empty((M, N), dtype=dtype, device=d)
    stats = torch.full((M, N // 32, 2), float("nan"), dtype=torch.float32, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=x.to(d), w=w.to(d), out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N, bias_n=bias.to(d),
                              residual=res.to(d), ldr=N, tile=tile, row_stats=stats))
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + bias + res.float()
    assert rel(out.float(), ref) < TOL[dtype]
    tot = stats.double().sum(1).cpu()
    assert torch.isfinite(tot).all()
    # (round 4) the statistics are those of the STORED 16-bit values — what the consuming GEMM multiplies — to fp32 summation accuracy
    st = out.double().cpu()
    assert rel(tot[:, 0], st.sum(1)) < 2e-6 and rel(tot[:, 1], (st ** 2).sum(1)) < 2e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C,N,kind", [(1024, 320, 320, "plain"), (512, 640, 5120, "geglu"), (2 * 256, 320, 960, "qkv"), (300, 1280, 1280, "plain"),
                                        (2 * 64, 1280, 3840, "qkv"), (4096, 320, 2560, "geglu")])
def test_layernorm_folded_into_the_consumer_gemm(dtype, M, C, N, kind):
    """Consumer side: the GEMM runs on the RAW rows against gamma-scaled weights and its epilogue applies
    rstd (alpha acc - mean alpha c1) + alpha c2 + bias — plain, GEGLU and the fused [Q; K; V] projection (transposed V) —
    against torch LayerNorm -> linear on the same 16-bit inputs.  The statistics come from a producing GEMM's row_stats."""
    from edtr_amd import lib as L
    ops = _ops()
    d = dev()
    x = (rnd((M, C), 170, 2.0) + 0.7).to(dtype)                       # non-zero mean: the mean term must cancel
    gamma, beta = 1 + 0.2 * rnd((C,), 171), 0.3 * rnd((C,), 172)
    w = rnd((N, C), 173, 1 / math.sqrt(C))
    bias = rnd((N,), 174)
    # statistics exactly as a producer writes them: here from an identity-like GEMM is overkill — build them from x directly
    xf = x.float()
    stats = torch.zeros((M, C // 32, 2), dtype=torch.float32)
    stats[:, 0, 0], stats[:, 0, 1] = xf.sum(1), (xf * xf).sum(1)
    if kind == "geglu":
        perm = ops.geglu_perm(N // 2)
        wq, bq = w[perm], bias[perm]
    else:
        wq, bq = w, bias
    packed = ops.pack_linear_weight(wq * gamma[None, :], dtype)
    c1 = packed.float().sum(1).contiguous()
    c2 = (wq @ beta).contiguous()
    alpha = 0.61
    ln = torch.nn.functional.layer_norm(xf, (C,), gamma, beta, 1e-5)
    full = alpha * (ln @ w.t()) + bias
    kw = dict(dtype=dtype, a1=x.to(d), w=packed.to(d), M=M, N=N, C1=C, ld1=C, ldw=C, alpha=alpha, bias_n=bq.to(d), ln_stats=stats.to(d), ln_C=C,
              ln_c1=c1.to(d), ln_c2=c2.to(d))
    if kind == "plain":
        out = torch.empty((M, N), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(out=out, ldc=N, **kw))
        torch.cuda.synchronize()
        assert rel(out.float(), full) < TOL[dtype]
    elif kind == "geglu":
        out = torch.empty((M, N // 2), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(out=out, ldc=N // 2, act=L.ACT_GEGLU, **kw))
        torch.cuda.synchronize()
        val, gate = full[:, : N // 2], full[:, N // 2:]
        assert rel(out.float(), val * F.gelu(gate)) < TOL[dtype]
    else:
        Cq = N // 3
        Bn, Ntok = 2, M // 2
        qk = torch.empty((M, 2 * Cq), dtype=dtype, device=d)
        vt = torch.empty((Bn * Cq, Ntok), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(out=qk, ldc=2 * Cq, rows_per_image=Ntok, vt_out=vt, vt_col0=2 * Cq, vt_ld=Ntok, vt_alpha=1.0, **kw))
        torch.cuda.synchronize()
        assert rel(qk.float(), full[:, : 2 * Cq]) < TOL[dtype]
        v_ref = ((ln @ w[2 * Cq:].t()) + bias[2 * Cq:]).reshape(Bn, Ntok, Cq).permute(0, 2, 1).reshape(Bn * Cq, Ntok)
        assert rel(vt.float(), v_ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,HW", [(320, 4096), (640, 1024), (1280, 256), (96, 384), (2560, 256), (512, 8192)])
def test_gn_apply_folds_the_tile_partials_itself(dtype, C, HW):
    """edtr_gn_apply with `partial`: the <= 64 per-tile column partials of the producing igemm are folded inside the apply
    launch (no edtr_gn_finalize launch) — identical to finalize + apply, incl. widths whose groups (10 / 20 / 40 / 80 channels)
    straddle the kernel's 256-channel chunks."""
    ops = _ops()
    d = dev()
    B = 2
    tiles = HW // 128
    x = (rnd((B * HW, C), 140, 1.5) + 0.3).to(dtype).to(d)
    xf = x.float().reshape(B, tiles, 128, C)
    gnp = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=-1).reshape(B * tiles, C, 2).contiguous()     # what an igemm epilogue writes
    gamma, beta = (1 + 0.1 * rnd((C,), 141)).to(d), (0.1 * rnd((C,), 142)).to(d)
    if C // 32 > 64:
        assert not ops.gn_foldable(HW, C)
        return
    assert ops.gn_foldable(HW, C) == (HW // 128 <= ops.GN_FOLD_MAX_TILES)      # the emitter's policy; the kernel itself takes <= 64 tiles
    sums = torch.empty((B, 32, 2), dtype=torch.float64, device=d)
    ops.launch(ops.make_gn_finalize(partial=gnp, tiles_per_image=tiles, B=B, C=C, sums=sums))
    y_ref, y = torch.empty_like(x), torch.empty_like(x)
    _, ap_ref = ops.make_gn(dtype=dtype, x=x, ldx=C, B=B, HW=HW, C=C, sums=sums, gamma=gamma, beta=beta, eps=1e-5, silu=True, y=y_ref, ldy=C)
    _, ap = ops.make_gn(dtype=dtype, x=x, ldx=C, B=B, HW=HW, C=C, sums=None, gamma=gamma, beta=beta, eps=1e-5, silu=True, y=y, ldy=C,
                        partial=gnp)
    ops.launch(ap_ref)
    ops.launch(ap)
    torch.cuda.synchronize()
    assert rel(y.float(), y_ref.float()) < 1e-6          # same fp64 totals up to summation order
    want = F.silu(F.group_norm(x.float().reshape(B, HW, C).permute(0, 2, 1).cpu(), 32, gamma.cpu(), beta.cpu(), 1e-5)).permute(0, 2, 1).reshape(B * HW, C)
    assert rel(y.float(), want) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,N", [(2, 2, 77), (1, 3, 200), (1, 1, 64), (2, 1, 129)])
def test_flash_attn64_causal(dtype, B, H, N):
    """Causal mask of the CLIP text tower (key j <= query i), incl. ragged last tiles and fully masked key tiles."""
    ops = _ops()
    d = dev()
    Cc = H * 64
    q = rnd((B, N, Cc), 130).to(dtype)
    k = rnd((B, N, Cc), 131).to(dtype)
    v = rnd((B, N, Cc), 132).to(dtype)
    ldv = ops.round_up(N, 8)
    vt = torch.zeros((B, Cc, ldv), dtype=dtype)
    vt[:, :, :N] = v.transpose(1, 2)
    out = torch.empty((B, N, Cc), dtype=dtype, device=d)
    ops.launch(ops.make_flash_attn(dtype=dtype, q=q.to(d), k=k.to(d), vt=vt.to(d), out=out, B=B, H=H, Nq=N, Nk=N,
                                   q_bs=N * Cc, q_ld=Cc, k_bs=N * Cc, k_ld=Cc, vt_bs=Cc * ldv, vt_ld=ldv, o_bs=N * Cc,
                                   o_ld=Cc, scale=0.125, causal=True))
    torch.cuda.synchronize()
    qh, kh, vh = (t.float().reshape(B, N, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True, scale=0.125).transpose(1, 2).reshape(B, N, Cc)
    assert rel(out.float().cpu(), ref) < TOL[dtype]
    with pytest.raises(RuntimeError):          # causal needs Nq == Nk
        ops.launch(ops.make_flash_attn(dtype=dtype, q=q.to(d), k=k.to(d), vt=vt.to(d), out=out, B=B, H=H, Nq=N, Nk=N - 1,
                                       q_bs=N * Cc, q_ld=Cc, k_bs=N * Cc, k_ld=Cc, vt_bs=Cc * ldv, vt_ld=ldv,
                                       o_bs=N * Cc, o_ld=Cc, scale=0.125, causal=True))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 3, 8, 6])
def test_gemm_gelu_epilogue(dtype, tile):
    """Exact (erf) GELU epilogue (the CLIP MLP, act = 3) on every main-loop family, with and without split-K."""
    ops = _ops()
    d = dev()
    M, N, K = 308, 320, 256
    a = rnd((M, K), 140).to(dtype)
    w = rnd((N, K), 141, 1 / math.sqrt(K)).to(dtype)
    b = rnd((N,), 142)
    ref = F.gelu(a.float() @ w.float().t() + b)
    out = torch.empty((M, N), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=a.to(d), w=w.to(d), out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N,
                              bias_n=b.to(d), act=3, tile=tile))
    torch.cuda.synchronize()
    assert rel(out.float().cpu(), ref) < TOL[dtype]
    if tile in (0, 3):
        ws = torch.empty(2 * M * N, dtype=torch.float32, device=d)
        out2 = torch.empty((M, N), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=a.to(d), w=w.to(d), out=out2, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N,
                                  bias_n=b.to(d), act=3, tile=tile, splitk=2, workspace=ws))
        torch.cuda.synchronize()
        assert rel(out2.float().cpu(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_tokens_and_zero_bytes(dtype):
    ops = _ops()
    d = dev()
    V, Lc, D, B = 1000, 77, 128, 3
    table, pos = rnd((V, D), 150), rnd((Lc, D), 151)
    tokens = torch.randint(0, V, (B, Lc), generator=torch.Generator().manual_seed(5))
    tokens[0, 3], tokens[1, 0] = V + 7, -2          # out-of-range ids are clamped, never read out of bounds
    out = torch.empty((B * Lc, D), dtype=dtype, device=d)
    ops.launch(ops.make_embed_tokens(dtype=dtype, tokens=tokens.to(d), table=table.to(d), pos=pos.to(d), rows=B * Lc,
                                     L_ctx=Lc, D=D, out=out, ld=D))
    torch.cuda.synchronize()
    ref = (table[tokens.clamp(0, V - 1)] + pos[None]).reshape(B * Lc, D)
    assert rel(out.float().cpu(), ref.to(dtype).float()) < 1e-6
    buf = torch.full((4096 + 16,), 3.0, dtype=torch.float64, device=d)
    ops.launch(ops.make_zero(buf[2:2 + 4096]))
    torch.cuda.synchronize()
    assert float(buf[2:2 + 4096].abs().max()) == 0.0 and float(buf[:2].min()) == 3.0 and float(buf[2 + 4096:].min()) == 3.0


# ---------------------------------------------------------------------------------------------------------------------
# SwinIR kernels
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("r,H,W,ld", [(8, 64, 96, 200), (4, 32, 40, 56), (2, 6, 8, 16)])
def test_pixel_unshuffle(dtype, r, H, W, ld):
    """(x - mean) * range + nn.PixelUnshuffle(r) into NHWC 16-bit rows with zeroed pad columns (model/swinir.py:861,700-704)."""
    ops = _ops()
    d = dev()
    B, C = 2, 3
    x = torch.rand((B, C, H, W), generator=torch.Generator().manual_seed(160))
    sub = torch.tensor([0.4488, 0.4371, 0.4040])
    out = torch.full((B * (H // r) * (W // r), ld), float("nan"), dtype=dtype, device=d)
    ops.launch(ops.make_pixel_unshuffle(dtype=dtype, src=x.to(d), B=B, C=C, H=H, W=W, r=r, dst=out, ld=ld, sub=sub.to(d), scale=2.0,
                                        zero_pad_to=ld))
    torch.cuda.synchronize()
    ref = F.pixel_unshuffle((x - sub.view(1, 3, 1, 1)) * 2.0, r).permute(0, 2, 3, 1).reshape(-1, C * r * r)
    got = out.float().cpu()
    assert torch.equal(got[:, :C * r * r], ref.to(dtype).float()) and float(got[:, C * r * r:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C,cv", [(1000, 192, 180), (37, 64, 60), (5, 256, 255)])
def test_layernorm_padded_columns(dtype, rows, C, cv):
    """c_valid < C: statistics over the real columns only (garbage in the pad columns is ignored), pad columns of y zeroed."""
    ops = _ops()
    d = dev()
    x = (rnd((rows, C), 161) * 2 + 0.5).to(dtype)
    x[:, cv:] = 7.0
    gamma, beta = torch.zeros(C), torch.zeros(C)
    gamma[:cv], beta[:cv] = 1 + 0.1 * rnd((cv,), 162), 0.1 * rnd((cv,), 163)
    y = torch.full((rows, C), float("nan"), dtype=dtype, device=d)
    ops.launch(ops.make_layernorm(dtype=dtype, x=x.to(d), rows=rows, C=C, ldx=C, gamma=gamma.to(d), beta=beta.to(d), eps=1e-5,
                                  y=y, ldy=C, c_valid=cv))
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert rel(got[:, :cv], F.layer_norm(x.float()[:, :cv], (cv,), gamma[:cv], beta[:cv], 1e-5)) < TOL[dtype]
    assert float(got[:, cv:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 6, 8])
def test_gemm_leaky_relu_epilogue(dtype, tile):
    """LeakyReLU epilogue (act = 4, SwinIR reconstruction convs) on every main-loop family and through the split-K reducer."""
    ops = _ops()
    d = dev()
    M, N, K = 308, 320, 256
    a = rnd((M, K), 164).to(dtype)
    w = rnd((N, K), 165, 1 / math.sqrt(K)).to(dtype)
    b = rnd((N,), 166)
    for slope in (0.2, 0.01):
        ref = F.leaky_relu(a.float() @ w.float().t() + b, slope)
        out = torch.empty((M, N), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=a.to(d), w=w.to(d), out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N,
                                  bias_n=b.to(d), act=4, act_slope=slope, tile=tile))
        torch.cuda.synchronize()
        assert rel(out.float().cpu(), ref) < TOL[dtype]
    if tile in (0, 3):
        ws = torch.empty(2 * M * N, dtype=torch.float32, device=d)
        out2 = torch.empty((M, N), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=a.to(d), w=w.to(d), out=out2, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N,
                                  bias_n=b.to(d), act=4, act_slope=0.01, tile=tile, splitk=2, workspace=ws))
        torch.cuda.synchronize()
        assert rel(out2.float().cpu(), ref) < TOL[dtype]
    p_bad = ops.make_igemm(dtype=dtype, a1=a.to(d), w=w.to(d), out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N, act=4, act_slope=1.5)
    with pytest.raises(RuntimeError):
        ops.launch(p_bad)


def _window_attn_reference(qkv, table_bias, H, W, heads, d, shift):
    """plain torch fp32 restatement of model/swinir.py:254-279 + :120-148 without the two linears: qkv [B, H, W, 3, heads, d]."""
    from edtr_amd.model import swinir as S
    B = qkv.shape[0]
    C = heads * d
    h = qkv.reshape(B, H, W, 3 * C)
    if shift:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    win = h.reshape(B, H // 8, 8, W // 8, 8, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64, 3, heads, d)
    q, k, v = win.permute(2, 0, 3, 1, 4)
    att = (q * d ** -0.5) @ k.transpose(-1, -2) + table_bias
    if shift:
        m = torch.from_numpy(S.shift_mask(H, W, 8, shift))
        att = (att.reshape(B, -1, heads, 64, 64) + m[None, :, None]).reshape(-1, heads, 64, 64)
    o = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(B, H // 8, W // 8, 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    return torch.roll(o, (shift, shift), (1, 2)) if shift else o


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,heads,d,cp,shift", [(2, 16, 24, 6, 30, 192, 0), (2, 16, 24, 6, 30, 192, 4), (1, 8, 8, 2, 30, 64, 4),
                                                    (3, 24, 8, 4, 32, 128, 3), (1, 64, 64, 6, 30, 192, 4), (1, 16, 16, 3, 8, 64, 0)])
def test_window_attention(dtype, B, H, W, heads, d, cp, shift):
    """edtr_window_attn vs the windowed / shifted / masked / biased softmax attention written out in torch fp32: non-square
    token grids, one-window images (every region label inside one window), head widths 8 / 30 / 32, odd shift."""
    from edtr_amd.model import swinir as S
    ops = _ops()
    dv = dev()
    qkv = (rnd((B, H, W, 3, heads, d), 170) * 1.5).to(dtype)
    table = rnd((225, heads), 171, 0.7)
    bias = S.expand_bias(table, 8)
    packed = torch.zeros((B * H * W, 3, heads, 32), dtype=dtype)
    packed[..., :d] = qkv.reshape(B * H * W, 3, heads, d)
    out = torch.full((B * H * W, cp), float("nan"), dtype=dtype, device=dv)
    lab = torch.from_numpy(S.region_labels(H, W, 8, shift)).to(dv) if shift else None
    ops.launch(ops.make_window_attn(dtype=dtype, qkv=packed.reshape(B * H * W, -1).to(dv), ld_qkv=3 * heads * 32, out=out, ld_out=cp,
                                    B=B, H=H, W=W, heads=heads, head_dim=d, c_pad=cp, shift=shift, bias=bias.to(dv), labels=lab,
                                    scale=d ** -0.5))
    torch.cuda.synchronize()
    ref = _window_attn_reference(qkv.float(), bias, H, W, heads, d, shift).reshape(B * H * W, heads * d)
    got = out.float().cpu()
    assert rel(got[:, :heads * d], ref) < TOL[dtype]
    assert float(got[:, heads * d:].abs().max()) == 0.0 if cp > heads * d else True


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,ups,cout,nchw", [(2, 32, 48, False, 64, False), (1, 64, 32, True, 64, False), (3, 16, 16, False, 64, False),
                                                  (2, 48, 32, False, 3, True), (1, 256, 320, True, 64, False), (1, 512, 512, False, 64, False)])
def test_conv64_persistent(dtype, B, H, W, ups, cout, nchw):
    """edtr_conv64 (SwinIR's 64-channel reconstruction convolutions, reference model/swinir.py:878-886) vs torch fp32 conv2d on the
    same 16-bit inputs: plain and behind a nearest-2x upsample, LeakyReLU, bias, the alpha scaling; 16-bit NHWC output and the
    fp32 NCHW planes of the network's last convolution; one-patch images, more patches than workgroups (the persistent loop)."""
    ops = _ops()
    from edtr_amd import lib as L
    d = dev()
    SH, SW = (H // 2, W // 2) if ups else (H, W)
    x = rnd((B, SH, SW, 64), 210, 1.2).to(dtype)
    w = rnd((cout, 64, 3, 3), 211, 1 / math.sqrt(576))
    bias = torch.zeros(64)
    bias[:cout] = 0.3 * rnd((cout,), 212)
    img = ops.pack_conv64_weight(w, dtype)
    alpha, slope = (0.7, 0.0) if nchw else (1.0, 0.2)
    if nchw:
        out = torch.full((B, cout, H, W), float("nan"), dtype=torch.float32, device=d)
    else:
        out = torch.full((B * H * W, 64), float("nan"), dtype=dtype, device=d)
    ops.launch(ops.make_conv64(dtype=dtype, x=x.reshape(-1, 64).to(d), ldx=64, w=img.to(d), bias=bias.to(d), out=out, B=B, H=H, W=W, upsample2x=ups,
                               act=L.ACT_NONE if nchw else L.ACT_LRELU, act_slope=slope, alpha=alpha, ldo=0 if nchw else 64, out_nchw_f32=nchw,
                               n_valid=cout if nchw else 0))
    torch.cuda.synchronize()
    xin = x.float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = alpha * F.conv2d(xin, w.to(dtype).float(), None, padding=1) + bias[:cout, None, None]
    if nchw:
        got = out.cpu()
    else:
        ref = F.leaky_relu(ref, slope)
        got = out.float().cpu().reshape(B, H, W, 64).permute(0, 3, 1, 2)
        assert float(got[:, cout:].abs().max()) == 0.0 if cout < 64 else True
        got = got[:, :cout]
    assert torch.isfinite(got).all()
    assert rel(got, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,gn", [(2, 32, 48, True), (1, 16, 16, True), (3, 64, 32, False), (1, 512, 512, True)])
def test_conv128_out_one_launch(dtype, B, H, W, gn):
    """edtr_conv128_out: the VAE decoder's norm_out -> SiLU -> conv_out -> NCHW (reference model/vae.py:553-560) in one launch vs
    torch fp32 on the same 16-bit input; with and without the normalisation, one-patch images, more patches than workgroups."""
    ops = _ops()
    d = dev()
    HW = H * W
    x = (rnd((B * HW, 128), 240, 1.3) + 0.3).to(dtype)
    w = rnd((3, 128, 3, 3), 241, 1 / math.sqrt(1152))
    bias = torch.zeros(32)
    bias[:3] = 0.3 * rnd((3,), 242)
    gamma, beta = (1 + 0.2 * rnd((128,), 243)).to(d), (0.2 * rnd((128,), 244)).to(d)
    xd = x.to(d)
    table = None
    if gn:
        sums = torch.zeros((B, 32, 2), dtype=torch.float64, device=d)
        st, _ = ops.make_gn(dtype=dtype, x=xd, ldx=128, B=B, HW=HW, C=128, sums=sums, gamma=gamma, beta=beta, eps=1e-6, silu=True, y=xd, ldy=128,
                            sums_zeroed=True)
        ops.launch(st)
        table = torch.empty((B, 128, 2), dtype=torch.float32, device=d)
        ops.launch(ops.make_gn_table(partial=None, tiles_per_image=0, sums=sums, B=B, C=128, HW=HW, gamma=gamma, beta=beta, eps=1e-6, table=table))
    out = torch.full((B, 3, H, W), float("nan"), dtype=torch.float32, device=d)
    ops.launch(ops.make_conv128_out(dtype=dtype, x=xd, ldx=128, w=ops.pack_conv128_out_weight(w, dtype).to(d), bias=bias.to(d), out=out, B=B, H=H, W=W,
                                    n_valid=3, gn_table=table, alpha=0.9))
    torch.cuda.synchronize()
    xin = x.float().reshape(B, H, W, 128).permute(0, 3, 1, 2)
    if gn:
        xin = F.silu(F.group_norm(xin, 32, gamma.cpu(), beta.cpu(), 1e-6)).to(dtype).float()      # (the kernel rounds the normalised operand to 16 bits)
    ref = 0.9 * F.conv2d(xin, w.to(dtype).float(), None, padding=1) + bias[:3, None, None]
    assert torch.isfinite(out).all()
    assert rel(out.cpu(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,shift", [(2, 16, 24, 0), (2, 16, 24, 4), (1, 8, 8, 4), (3, 24, 8, 3), (1, 64, 64, 4), (1, 8, 24, 0)])
def test_swin_attention_half_one_launch(dtype, B, H, W, shift):
    """edtr_swin_attn: x + proj(WindowAttention(LayerNorm(x))) of a Swin layer in one launch (reference model/swinir.py:254-279,
    :120-148) vs torch fp32 on the same 16-bit inputs: LayerNorm, the qkv linear, q * scale, relative-position bias, the shift
    mask, softmax, proj and the residual; non-square token grids, one-window images, an odd window count (the last workgroup
    owns one window), odd shift; pad columns stay exactly zero."""
    from edtr_amd.model import swinir as S
    ops = _ops()
    dv = dev()
    heads, d, C, CP = 6, 30, 180, ops.SWIN_MLP_C
    rows = B * H * W
    x = torch.zeros((rows, CP))
    x[:, :C] = rnd((rows, C), 200, 1.5) + 0.3
    x = x.to(dtype)
    gamma, beta = 1 + 0.2 * rnd((C,), 201), 0.3 * rnd((C,), 202)
    wqkv, bqkv = rnd((3 * C, C), 203, 1.5 / math.sqrt(C)), 0.2 * rnd((3 * C,), 204)
    wp, bp = rnd((C, C), 205, 1 / math.sqrt(C)), 0.2 * rnd((C,), 206)
    table = rnd((225, heads), 207, 0.7)
    bias = S.expand_bias(table, 8)
    # packing exactly as model/swinir.py does it
    wq32, bq32 = S.pack_qkv(wqkv, bqkv, heads, CP)                  # [(s, h, e), CP] with zero pad rows / columns
    g_pad, b_pad = torch.zeros(CP), torch.zeros(CP)
    g_pad[:C], b_pad[:C] = gamma, beta
    scale = torch.ones(3 * heads * 32)
    scale[: heads * 32] = d ** -0.5
    wg = wq32 * g_pad[None, :] * scale[:, None]
    c2b = ((wq32 @ b_pad) + bq32) * scale
    wpp = torch.zeros((CP, heads, 32))
    wpp[:C, :, :d] = wp.reshape(C, heads, d)
    img_qkv, img_proj = ops.pack_swin_attn_weights(wg, wpp.reshape(CP, heads * 32), dtype)
    c1 = wg.to(dtype).float().sum(1).contiguous()
    bpp = torch.zeros(CP)
    bpp[:C] = bp
    out = torch.full((rows, CP), float("nan"), dtype=dtype, device=dv)
    lab = torch.from_numpy(S.region_labels(H, W, 8, shift)).to(dv) if shift else None
    ops.launch(ops.make_swin_attn(dtype=dtype, x=x.to(dv), ldx=CP, out=out, ldo=CP, B=B, H=H, W=W, head_dim=d, shift=shift, c_valid=C, eps=1e-5,
                                  wqkv=img_qkv.to(dv), wproj=img_proj.to(dv), c1=c1.to(dv), c2b=c2b.contiguous().to(dv), bproj=bpp.to(dv),
                                  bias=ops.swin_attn_bias(bias).to(dv), labels=lab))
    torch.cuda.synchronize()
    xf = x.float()[:, :C]
    qkv = F.layer_norm(xf, (C,), gamma, beta, 1e-5) @ wqkv.t() + bqkv
    att = _window_attn_reference(qkv.reshape(B, H, W, 3, heads, d), bias, H, W, heads, d, shift).reshape(rows, C)
    ref = xf + att @ wp.t() + bp
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert rel(got[:, :C], ref) < TOL[dtype]
    assert float(got[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,shift", [(2, 16, 24, 4), (1, 8, 24, 0), (1, 64, 64, 4)])
def test_swin_layer_one_launch(dtype, B, H, W, shift):
    """edtr_swin_layer (both halves of a Swin layer on one LDS tile) == edtr_swin_attn followed by edtr_swin_mlp, bit for bit: the
    two forms run the same arithmetic on the same 16-bit tile."""
    from edtr_amd.model import swinir as S
    ops = _ops()
    dv = dev()
    heads, d, C, HID, CP, HP = 6, 30, 180, 360, ops.SWIN_MLP_C, ops.SWIN_MLP_HIDDEN
    rows = B * H * W
    x = torch.zeros((rows, CP))
    x[:, :C] = rnd((rows, C), 220, 1.5) + 0.3
    x = x.to(dtype).to(dv)
    wq32, bq32 = S.pack_qkv(rnd((3 * C, C), 221, 1.5 / math.sqrt(C)), 0.2 * rnd((3 * C,), 222), heads, CP)
    g1, b1 = torch.zeros(CP), torch.zeros(CP)
    g1[:C], b1[:C] = 1 + 0.2 * rnd((C,), 223), 0.3 * rnd((C,), 224)
    scale = torch.ones(3 * heads * 32)
    scale[: heads * 32] = d ** -0.5
    wg = wq32 * g1[None, :] * scale[:, None]
    wpp = torch.zeros((CP, heads, 32))
    wpp[:C, :, :d] = rnd((C, C), 225, 1 / math.sqrt(C)).reshape(C, heads, d)
    img_qkv, img_proj = ops.pack_swin_attn_weights(wg, wpp.reshape(CP, heads * 32), dtype)
    bpp = torch.zeros(CP)
    bpp[:C] = 0.2 * rnd((C,), 226)
    bias = ops.swin_attn_bias(S.expand_bias(rnd((225, heads), 227, 0.7), 8))
    lab = torch.from_numpy(S.region_labels(H, W, 8, shift)).to(dv) if shift else None
    w1g, w2p, c2b, b2p = torch.zeros((HP, CP)), torch.zeros((CP, HP)), torch.zeros(HP), torch.zeros(CP)
    w1g[:HID, :C] = rnd((HID, C), 228, 1 / math.sqrt(C))
    w2p[:C, :HID] = rnd((C, HID), 229, 1 / math.sqrt(HID))
    c2b[:HID], b2p[:C] = 0.2 * rnd((HID,), 230), 0.2 * rnd((C,), 231)
    img1, img2 = ops.pack_swin_mlp_weights(w1g, w2p, dtype)

    def recs(x1, y):
        a = ops.make_swin_attn(dtype=dtype, x=x, ldx=CP, out=x1, ldo=CP, B=B, H=H, W=W, head_dim=d, shift=shift, c_valid=C, eps=1e-5,
                               wqkv=img_qkv.to(dv), wproj=img_proj.to(dv), c1=wg.to(dtype).float().sum(1).contiguous().to(dv),
                               c2b=((wq32 @ b1 + bq32) * scale).contiguous().to(dv), bproj=bpp.to(dv), bias=bias.to(dv), labels=lab)
        m = ops.make_swin_mlp(dtype=dtype, x=x1, ldx=CP, rows=rows, c_valid=C, eps=1e-5, w1=img1.to(dv), w2=img2.to(dv),
                              c1=w1g.to(dtype).float().sum(1).contiguous().to(dv), c2b=c2b.to(dv), b2=b2p.to(dv), out=y, ldo=CP)
        return a, m
    x1 = torch.empty((rows, CP), dtype=dtype, device=dv)
    two = torch.full((rows, CP), float("nan"), dtype=dtype, device=dv)
    a, m = recs(x1, two)
    ops.launch(a)
    ops.launch(m)
    one = torch.full((rows, CP), float("nan"), dtype=dtype, device=dv)
    a2, m2 = recs(one, one)
    ops.launch(ops.make_swin_layer(a2, m2))
    torch.cuda.synchronize()
    assert torch.isfinite(one.float()).all()
    assert torch.equal(one, two)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows", [128, 4096 + 64, 300, 32768])
def test_swin_mlp_one_launch(dtype, rows):
    """edtr_swin_mlp: x + fc2(GELU(fc1(LayerNorm(x)))) of a Swin layer in one launch (reference model/swinir.py:24-37, :281-283)
    vs torch fp32 on the same 16-bit inputs: 180 real of 192 columns, 360 real of 384 hidden units, ragged row counts
    (a partial last workgroup, a partial 32-token tile); pad columns stay exactly zero; the row statistics it writes for the
    next layer's folded LayerNorm are those of the stored values."""
    ops = _ops()
    d = dev()
    C, HID, CP, HP = 180, 360, ops.SWIN_MLP_C, ops.SWIN_MLP_HIDDEN
    x = torch.zeros((rows, CP))
    x[:, :C] = rnd((rows, C), 190, 1.5) + 0.4
    x = x.to(dtype)
    gamma, beta = 1 + 0.2 * rnd((C,), 191), 0.3 * rnd((C,), 192)
    w1, b1 = rnd((HID, C), 193, 1 / math.sqrt(C)), 0.2 * rnd((HID,), 194)
    w2, b2 = rnd((C, HID), 195, 1 / math.sqrt(HID)), 0.2 * rnd((C,), 196)
    w1g = torch.zeros((HP, CP))
    w1g[:HID, :C] = w1 * gamma[None, :]
    w2p = torch.zeros((CP, HP))
    w2p[:C, :HID] = w2
    img1, img2 = ops.pack_swin_mlp_weights(w1g, w2p, dtype)
    c1 = w1g.to(dtype).float().sum(1).contiguous()
    c2b = torch.zeros(HP)
    c2b[:HID] = w1 @ beta + b1
    b2p = torch.zeros(CP)
    b2p[:C] = b2
    out = torch.full((rows, CP), float("nan"), dtype=dtype, device=d)
    stats = torch.full((rows, CP // 32, 2), float("nan"), dtype=torch.float32, device=d)
    ops.launch(ops.make_swin_mlp(dtype=dtype, x=x.to(d), ldx=CP, rows=rows, c_valid=C, eps=1e-5, w1=img1.to(d), w2=img2.to(d), c1=c1.to(d),
                                 c2b=c2b.to(d), b2=b2p.to(d), out=out, ldo=CP, row_stats=stats))
    torch.cuda.synchronize()
    xf = x.float()[:, :C]
    h = F.gelu(F.layer_norm(xf, (C,), gamma, beta, 1e-5) @ w1.t() + b1)
    ref = xf + h @ w2.t() + b2
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert rel(got[:, :C], ref) < TOL[dtype]
    assert float(got[:, C:].abs().max()) == 0.0
    tot = stats.double().sum(1).cpu()
    st = out.double().cpu()
    assert rel(tot[:, 0], st.sum(1)) < 2e-6 and rel(tot[:, 1], (st ** 2).sum(1)) < 2e-6


# ---------------------------------------------------------------------------------------------------------------------
# tile 14 (256 x 32, skinny-N convolutions: the VAE decoder's 3-channel output conv) is selected automatically for N <= 32 at
# large M; tiles 11-13 (round-1 experiments) were measured on the MI355X without gain and removed.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_skinny_n_conv_tile14_matches_tile3(dtype):
    ops = _ops()
    d = dev()
    B, H, W, cin, cout = 2, 192, 256, 128, 8               # M = 98304 >= 65536: the automatic choice is tile 14
    x = rnd((B * H * W, cin), 183).to(dtype).to(d)
    w = rnd((cout, 9 * cin), 184, 1 / math.sqrt(9 * cin)).to(dtype).to(d)
    b = rnd((cout,), 185).to(d)
    outs = []
    for t in (3, 14, 0):
        out = torch.empty((B * H * W, cout), dtype=torch.float32, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=out, taps=9, M=B * H * W, N=cout, C1=cin, ld1=cin, ldw=9 * cin,
                                  ldc=cout, spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=b, out_f32=True, tile=t))
        outs.append(out)
    torch.cuda.synchronize()
    assert rel(outs[1], outs[0]) < 1e-5                     # other MFMA shape, same products: fp32 summation order only
    assert torch.equal(outs[2], outs[1])                    # tile 0 (auto) picked tile 14
    for gone in (4, 5, 7, 9, 10, 11, 15, 18, 19, 21):                   # the removed experiments are rejected, not silently remapped
        with pytest.raises(RuntimeError):
            ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=outs[0], taps=9, M=B * H * W, N=cout, C1=cin, ld1=cin, ldw=9 * cin,
                                      ldc=cout, spatial=(H, W, H, W, 1, 1, 1, 0), out_f32=True, tile=gone))


# ---------------------------------------------------------------------------------------------------------------------
# Halo tile (tile 16): 3x3 / stride 1 convolutions keep the 18 x 18 input patch of a 16 x 16 output patch in LDS and read the
# nine taps from it.  Same products as tile 3 in another summation order (chunk-major K, two K halves per chunk): fp32
# reference for the absolute bound, tile 3 for the tight one; every epilogue path the UNet / VAE convolutions use.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # B, H, W, cin, cout, bias, rowvec, SiLU, residual, gn_partial, out_f32
    (2, 32, 48, 128, 128, True, False, 0, False, True, False),     # VAE ResBlock conv1: bias + fused GroupNorm partials
    (3, 16, 16, 64, 256, True, True, 0, False, True, False),       # UNet ResBlock conv1: + time-embedding row per image
    (1, 48, 32, 192, 320, True, False, 0, True, True, False),      # conv2 + skip; N = 320: a 64-wide last column tile
    (2, 16, 32, 128, 128, True, False, 1, False, False, False),    # SiLU epilogue
    (1, 32, 32, 384, 128, False, False, 0, False, False, True),    # fp32 output (the parity mode's convs: K = 9 * 3C)
])
def test_halo_conv_tile16(dtype, case):
    _halo_case(dtype, case, ups=False)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    (2, 16, 24, 128, 128, True, False, 0, False, True, False),     # Upsample: nearest-2x fused into the gather (source 16 x 24 -> 32 x 48)
    (1, 8, 8, 192, 256, True, False, 0, False, False, False),      # one 16 x 16 output patch per image: every border
    (3, 24, 16, 64, 320, True, False, 0, True, False, False),      # + residual, ragged last column tile
])
def test_halo_conv_tile16_upsample2x(dtype, case):
    _halo_case(dtype, case, ups=True)


# ---------------------------------------------------------------------------------------------------------------------
# Tile 17 (halo512.hip, round 5): the halo tile on 32 x 16-pixel units with 32-channel chunks and an epilogue that works
# straight from the accumulators (weight rows permuted so that a lane holds runs of eight consecutive channels; whole-line
# stores through a row_ror:8 exchange; a 16-bit residual fetched by LDS-DMA).  Same products as tiles 3 / 16 in another
# summation order.  One / three / five chunks, every image border inside a unit and between units, several column tiles,
# every epilogue path the VAE convolutions use in the three precision modes.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # B, H, W, cin, cout, bias, rowvec, SiLU, residual, gn_partial, out_f32
    (2, 16, 32, 32, 128, True, False, 0, False, True, False),      # ONE unit per image, ONE chunk: every border, the next chunk's staging is out-of-range loads
    (1, 32, 64, 96, 128, True, False, 0, True, True, False),       # 2 x 2 units, 3 chunks, 16-bit residual (LDS-DMA) + statistics: VAE conv2
    (3, 16, 64, 160, 256, True, True, 0, False, True, False),      # 5 chunks, two column tiles, time-embedding row per image
    (1, 48, 32, 64, 384, False, False, 0, True, False, False),     # three column tiles, no bias, residual without statistics
    (2, 32, 32, 128, 128, True, False, 0, False, True, True),      # fp32 output (the parity modes) + statistics
    (1, 16, 32, 64, 128, True, False, 0, True, False, True),       # fp32 output + 16-bit residual
])
def test_halo_conv_tile17(dtype, case):
    _halo_case(dtype, case, ups=False, tile=17)


@pytest.mark.parametrize("dtype", DTYPES)
def test_halo_conv_tile17_fp32_residual_and_mirror(dtype):
    """The mixed mode's conv2: fp32 output, fp32 residual, 16-bit mirror of the sum (edtr_igemm_params.out16), statistics."""
    ops = _ops()
    d = dev()
    B, H, W, cin, cout = 2, 16, 64, 64, 128
    M = B * H * W
    x = rnd((M, cin), 501).to(dtype)
    w = rnd((cout, 9 * cin), 502, 1 / math.sqrt(9 * cin)).to(dtype)
    bias = rnd((cout,), 503).to(d)
    res = rnd((M, cout), 504).to(d)
    outs = {}
    for t in (3, 17):
        out = torch.full((M, cout), float("nan"), dtype=torch.float32, device=d)
        mir = torch.full((M, cout), float("nan"), dtype=dtype, device=d)
        gn = torch.full((M // 128, cout, 2), float("nan"), dtype=torch.float32, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=x.to(d), w=w.to(d), out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, residual=res, ldr=cout, residual_f32=True, out_f32=True,
                                  out16=mir, gn_partial=gn, tile=t))
        outs[t] = (out, mir, gn)
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().reshape(B, H, W, cin).permute(0, 3, 1, 2), w.float().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2), bias.cpu(),
                   padding=1).permute(0, 2, 3, 1).reshape(M, cout) + res.cpu()
    assert rel(outs[17][0], ref) < 2e-5 and rel(outs[17][0], outs[3][0]) < 2e-6
    assert torch.equal(outs[17][1], outs[17][0].to(dtype))                                  # the mirror is the stored sum, rounded once
    s17, s3 = (g.double().reshape(B, -1, cout, 2).sum(1) for g in (outs[17][2], outs[3][2]))
    assert float((s17 - s3).abs().max() / s3.abs().max()) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,cin,cout,from_partial", [(1, 16, 32, 32, 128, False), (2, 32, 32, 96, 128, True), (1, 32, 64, 160, 256, False), (2, 16, 64, 128, 128, True)])
def test_halo_conv_tile17_with_groupnorm_of_its_input(dtype, B, H, W, cin, cout, from_partial):
    """a_gn on tile 17 (prologue-only path with one chunk, the in-loop normalisation with 3 / 5 / 4 chunks; image borders on every side
    of a unit: the padding must stay zero AFTER the normalisation) — against edtr_gn_apply + the same convolution, and torch fp32."""
    ops = _ops()
    d = dev()
    M, HW = B * H * W, H * W
    x = (rnd((M, cin), 401, 1.3) + 0.4).to(dtype)
    w = rnd((cout, 9 * cin), 402, 1 / math.sqrt(9 * cin)).to(dtype)
    bias = rnd((cout,), 403).to(d)
    gamma, beta = (1 + 0.2 * rnd((cin,), 404)).to(d), (0.2 * rnd((cin,), 405)).to(d)
    xd, wd = x.to(d), w.to(d)
    sums = torch.zeros((B, 32, 2), dtype=torch.float64, device=d)
    y = torch.empty((M, cin), dtype=dtype, device=d)
    st, ap = ops.make_gn(dtype=dtype, x=xd, ldx=cin, B=B, HW=HW, C=cin, sums=sums, gamma=gamma, beta=beta, eps=1e-6, silu=True, y=y, ldy=cin,
                         sums_zeroed=True)
    ops.launch(st)
    ops.launch(ap)
    table = torch.full((B, cin, 2), float("nan"), dtype=torch.float32, device=d)
    partial = None
    if from_partial:
        xf = xd.float().reshape(B * HW // 128, 128, cin)
        partial = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=-1).contiguous()
    ops.launch(ops.make_gn_table(partial=partial, tiles_per_image=HW // 128, sums=None if from_partial else sums, B=B, C=cin, HW=HW, gamma=gamma,
                                 beta=beta, eps=1e-6, table=table))
    outs = {}
    for fused in (False, True):
        out = torch.full((M, cout), float("nan"), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=xd if fused else y, w=wd, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, rows_per_image=HW, tile=17,
                                  a_gn=table if fused else None, a_gn_silu=True))
        outs[fused] = out
    torch.cuda.synchronize()
    xr = F.silu(F.group_norm(x.float().reshape(B, H, W, cin).permute(0, 3, 1, 2), 32, gamma.cpu(), beta.cpu(), 1e-6))
    ref = F.conv2d(xr, w.float().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2), bias.cpu(), padding=1).permute(0, 2, 3, 1).reshape(M, cout)
    assert torch.isfinite(outs[True].float()).all()
    assert rel(outs[True].float(), ref) < TOL[dtype]
    assert rel(outs[True].float(), outs[False].float()) < 0.1 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,cin,cout,gnin,bias", [(1, 512, 512, 64, 128, False, True),       # 512 units: two per workgroup, two chunks
                                                      (3, 256, 256, 128, 256, True, True),       # 768 units in two column tiles: three per workgroup, GroupNorm + SiLU of the input
                                                      (1, 272, 512, 64, 128, True, False),       # 272 units: 16 workgroups walk two, the others one
                                                      (1, 32, 64, 64, 128, False, True)])        # four units: a grid of eight, half of it without work
def test_halo_conv_tile21_persistent_form_is_bit_identical_to_tile17(dtype, B, H, W, cin, cout, gnin, bias):
    """Tile 21 (round 6, opt-in): tile 17's units walked by a persistent grid, the finished 16-bit output of unit i held in registers and
    stored under unit i + 1's multiply loop, its GroupNorm sums published behind that unit's first phase.  Same products in the same
    order, same single rounding: the stored tensor and the statistics must EQUAL tile 17's."""
    ops = _ops()
    d = dev()
    M = B * H * W
    g = torch.Generator().manual_seed(2100 + cin + H)
    x = (torch.randn((M, cin), generator=g) * 1.2 + 0.3).to(dtype).to(d)
    w = (torch.randn((cout, 9 * cin), generator=g) / math.sqrt(9 * cin)).to(dtype).to(d)
    bv = torch.randn((cout,), generator=g).to(d) if bias else None
    table = None
    if gnin:
        table = torch.stack([1 + 0.3 * torch.randn((B, cin), generator=g), 0.2 * torch.randn((B, cin), generator=g)], dim=-1).contiguous().to(d)
    outs = {}
    for t in (17, 21):
        out = torch.full((M, cout), float("nan"), dtype=dtype, device=d)
        gn = torch.full((M // 128, cout, 2), float("nan"), dtype=torch.float32, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bv, rows_per_image=H * W, tile=t, gn_partial=gn,
                                  a_gn=table, a_gn_silu=gnin))
        outs[t] = (out, gn)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[21][0].float()).all()
    assert torch.equal(outs[21][0], outs[17][0])
    assert torch.equal(outs[21][1], outs[17][1])


# ---------------------------------------------------------------------------------------------------------------------
# Tile 20 (halo512.hip, round 5): the halo tile on 16 x 16 pixels x 160 channels for N % 160 == 0 (the 64 x 64-latent ResBlock
# convolutions of the UNet / ControlNet: N = 320).  One / three / ten chunks, one / two / four column tiles, time-embedding row,
# residual (16-bit and fp32), statistics, fp32 output + mirror.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # B, H, W, cin, cout, bias, rowvec, SiLU, residual, gn_partial, out_f32
    (2, 16, 16, 32, 160, True, False, 0, False, True, False),      # ONE unit per image, ONE chunk, one column tile: every border
    (1, 32, 48, 96, 320, True, True, 0, False, True, False),       # 2 x 3 units, 3 chunks, two column tiles, time-embedding row: UNet conv1
    (3, 16, 32, 320, 320, True, False, 0, True, True, False),      # 10 chunks (the unrolled-by-four chunk loop wraps), 16-bit residual: UNet conv2
    (2, 32, 16, 64, 640, False, False, 0, True, False, False),     # four column tiles, no bias
    (1, 16, 32, 128, 160, True, True, 0, False, True, True),       # fp32 output + statistics (parity modes)
])
def test_halo_conv_tile20(dtype, case):
    _halo_case(dtype, case, ups=False, tile=20)


def test_halo160_fp32_residual_mirror_and_automatic_choice():
    ops = _ops()
    d = dev()
    dtype = torch.float16
    B, H, W, cin, cout = 2, 32, 32, 64, 320
    M = B * H * W
    x = rnd((M, cin), 521).to(dtype).to(d)
    w = rnd((cout, 9 * cin), 522, 1 / math.sqrt(9 * cin)).to(dtype).to(d)
    bias = rnd((cout,), 523).to(d)
    res = rnd((M, cout), 524).to(d)
    outs = {}
    for t in (3, 20):
        out = torch.full((M, cout), float("nan"), dtype=torch.float32, device=d)
        mir = torch.full((M, cout), float("nan"), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, residual=res, ldr=cout, residual_f32=True, out_f32=True,
                                  out16=mir, tile=t))
        outs[t] = (out, mir)
    torch.cuda.synchronize()
    assert rel(outs[20][0], outs[3][0]) < 2e-6 and torch.equal(outs[20][1], outs[20][0].to(dtype))
    # automatic: 256 units (8 images of 64 x 64, N = 320) -> tile 20; 128 units -> not
    from edtr_amd import lib as L
    import ctypes as C
    def plan(Bn, Hn, N):
        p = L.IgemmParams()
        p.dtype, p.taps, p.M, p.N, p.K, p.Z, p.zdiv = 0, 9, Bn * Hn * Hn, N, 9 * 320, 1, 1
        p.a1, p.C1, p.ld1, p.w, p.ldw = x.data_ptr(), 320, 320, w.data_ptr(), 9 * 320
        p.IH, p.IW, p.OH, p.OW, p.stride, p.pad_t, p.pad_l = Hn, Hn, Hn, Hn, 1, 1, 1
        p.alpha, p.out, p.ldc = 1.0, outs[3][0].data_ptr(), N
        return L.load().edtr_igemm_plan(C.byref(p))
    assert plan(8, 64, 320) == 20 and plan(4, 64, 320) != 20 and plan(8, 64, 640) != 20
    with pytest.raises(RuntimeError):                        # N = 128 is not a multiple of 160
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w[:128], out=torch.empty((M, 128), dtype=dtype, device=d), taps=9, M=M, N=128, C1=cin, ld1=cin,
                                  ldw=9 * cin, ldc=128, spatial=(H, W, H, W, 1, 1, 1, 0), tile=20))


def test_halo512_is_the_automatic_choice_from_256_units_and_rejects_other_shapes():
    """tile 0 picks tile 17 where tile 16 would run and there are >= 256 units of 512 pixels x 128 channels; below that tile 16; an
    explicit tile 17 on a shape it cannot run (width not a multiple of 32, split-K, an activation) is an error."""
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    B, H, W, cin, cout = 1, 256, 512, 64, 128                    # 256 units
    M = B * H * W
    x = rnd((M, cin), 511).to(dtype).to(d)
    w = rnd((cout, 9 * cin), 512, 1 / math.sqrt(9 * cin)).to(dtype).to(d)
    outs = []
    for t in (17, 0, 16):
        out = torch.empty((M, cout), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), tile=t))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])                         # tile 0 (auto) ran tile 17 ...
    assert rel(outs[0], outs[2]) < TOL[dtype] and not torch.equal(outs[0], outs[2])      # ... not tile 16 (another summation order)
    Ms = 128 * 512
    outs = []
    for t in (16, 0):                                            # 128 units: the 256-pixel halo tile stays
        out = torch.empty((Ms, cout), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=out, taps=9, M=Ms, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(128, 512, 128, 512, 1, 1, 1, 0), tile=t))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    for kw in (dict(spatial=(64, 48, 64, 48, 1, 1, 1, 0), M=64 * 48), dict(spatial=(64, 64, 64, 64, 1, 1, 1, 0), M=4096, act=1)):
        m = kw.pop("M")
        from edtr_amd import lib as L
        act = L.ACT_SILU if kw.pop("act", 0) else L.ACT_NONE
        with pytest.raises(RuntimeError):
            ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=torch.empty((m, cout), dtype=dtype, device=d), taps=9, M=m, N=cout,
                                      C1=cin, ld1=cin, ldw=9 * cin, ldc=cout, act=act, tile=17, **kw))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,cin,cout,splitk,from_partial", [(1, 32, 48, 128, 128, 1, False), (2, 16, 16, 192, 256, 1, True), (1, 64, 64, 64, 128, 1, False),
                                                               (2, 16, 32, 320, 128, 1, True), (1, 16, 16, 1280, 256, 3, False), (8, 32, 32, 640, 640, 1, True)])
def test_halo_conv_with_groupnorm_of_its_input(dtype, B, H, W, cin, cout, splitk, from_partial):
    """edtr_igemm with a_gn: GroupNorm apply + SiLU of the INPUT inside the halo tile's patch staging (reference model/vae.py:103-114,
    model/unet.py:203-218: `conv(silu(norm(x)))`) — against the two-launch form (edtr_gn_apply, then the same convolution) and against
    torch fp32.  One / three / five / twenty 64-channel chunks (the prologue-only path, the in-loop normalisation, odd chunk counts),
    split-K, statistics from tile partials and from fp64 sums, image borders on every side of a patch (the padding must stay zero
    AFTER the normalisation)."""
    from edtr_amd import lib as L
    ops = _ops()
    d = dev()
    M, HW = B * H * W, H * W
    x = (rnd((M, cin), 401, 1.3) + 0.4).to(dtype)
    w = rnd((cout, 9 * cin), 402, 1 / math.sqrt(9 * cin)).to(dtype)
    bias = rnd((cout,), 403).to(d)
    gamma, beta = (1 + 0.2 * rnd((cin,), 404)).to(d), (0.2 * rnd((cin,), 405)).to(d)
    xd, wd = x.to(d), w.to(d)
    sums = torch.zeros((B, 32, 2), dtype=torch.float64, device=d)
    y = torch.empty((M, cin), dtype=dtype, device=d)
    st, ap = ops.make_gn(dtype=dtype, x=xd, ldx=cin, B=B, HW=HW, C=cin, sums=sums, gamma=gamma, beta=beta, eps=1e-6, silu=True, y=y, ldy=cin,
                         sums_zeroed=True)
    ops.launch(st)
    ops.launch(ap)
    table = torch.full((B, cin, 2), float("nan"), dtype=torch.float32, device=d)
    partial = None
    if from_partial:            # the statistics as a producing igemm's epilogue leaves them: per 128-row tile and channel
        xf = xd.float().reshape(B * HW // 128, 128, cin)
        partial = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=-1).contiguous()
    ops.launch(ops.make_gn_table(partial=partial, tiles_per_image=HW // 128, sums=None if from_partial else sums, B=B, C=cin, HW=HW, gamma=gamma,
                                 beta=beta, eps=1e-6, table=table))
    outs = {}
    for fused in (False, True):
        out = torch.full((M, cout), float("nan"), dtype=dtype, device=d)
        ws = torch.empty((splitk * M * cout,), dtype=torch.float32, device=d) if splitk > 1 else None
        ops.launch(ops.make_igemm(dtype=dtype, a1=xd if fused else y, w=wd, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, rows_per_image=HW, tile=16, splitk=splitk, workspace=ws,
                                  a_gn=table if fused else None, a_gn_silu=True))
        outs[fused] = out
    torch.cuda.synchronize()
    xr = F.silu(F.group_norm(x.float().reshape(B, H, W, cin).permute(0, 3, 1, 2), 32, gamma.cpu(), beta.cpu(), 1e-6))
    ref = F.conv2d(xr, w.float().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2), bias.cpu(), padding=1).permute(0, 2, 3, 1).reshape(M, cout)
    assert torch.isfinite(outs[True].float()).all()
    assert rel(outs[True].float(), ref) < TOL[dtype]
    # the fused form rounds the normalised activation exactly as edtr_gn_apply stores it: the two forms agree to the order of the
    # statistics' summation (fp32 tile partials against fp64 sums) — an order of magnitude below the kernels' own rounding error
    assert rel(outs[True].float(), outs[False].float()) < 0.1 * TOL[dtype]


def _halo_case(dtype, case, ups, tile=16):
    import torch.nn.functional as F
    from edtr_amd import lib as L
    ops = _ops()
    d = dev()
    B, IH, IW, cin, cout, use_bias, use_rv, act, use_res, use_gn, out_f32 = case
    H, W = (2 * IH, 2 * IW) if ups else (IH, IW)              # output size
    M = B * H * W
    x = rnd((B * IH * IW, cin), 301).to(dtype)
    w = rnd((cout, 9 * cin), 302, 1 / math.sqrt(9 * cin)).to(dtype)
    bias = rnd((cout,), 303).to(d) if use_bias else None
    rv = rnd((B, cout), 304).to(d) if use_rv else None
    res = rnd((M, cout), 305).to(dtype).to(d) if use_res else None
    xd, wd = x.to(d), w.to(d)
    outs, gns = {}, {}
    ref_tile = 3 if cin % 64 == 0 else 1                     # (the LDS-DMA 128x128 loop needs whole 64-channel K-tiles)
    for t in (ref_tile, tile):
        out = torch.full((M, cout), float("nan"), dtype=torch.float32 if out_f32 else dtype, device=d)
        gn = torch.full((M // 128, cout, 2), float("nan"), dtype=torch.float32, device=d) if use_gn else None
        ops.launch(ops.make_igemm(dtype=dtype, a1=xd, w=wd, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(IH, IW, H, W, 1, 1, 1, int(ups)), bias_n=bias, rowvec=rv, rowvec_ld=cout if use_rv else 0,
                                  rows_per_image=H * W, act=L.ACT_SILU if act else L.ACT_NONE, residual=res, ldr=cout, out_f32=out_f32, gn_partial=gn, tile=t))
        outs[t], gns[t] = out, gn
    torch.cuda.synchronize()
    # fp32 reference on the rounded operands (weights are packed [cout][ky][kx][cin])
    xr = x.float().reshape(B, IH, IW, cin).permute(0, 3, 1, 2)
    if ups:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    wr = w.float().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xr, wr, bias.cpu() if use_bias else None, padding=1)
    if use_rv:
        ref = ref + rv.cpu()[:, :, None, None]
    if act == 1:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1).reshape(M, cout)
    if use_res:
        ref = ref + res.float().cpu()
    assert torch.isfinite(outs[tile].float()).all()
    assert rel(outs[tile], ref) < (2e-5 if out_f32 else TOL[dtype])
    assert rel(outs[tile], outs[ref_tile]) < (2e-6 if out_f32 else TOL[dtype])           # same products, other summation order
    if use_gn:
        # the per-tile partials sit in other slots (one 256-pixel patch = two 128-row slots, one 512-pixel unit = four): compare the per-image sums
        s16, s3 = (g.double().reshape(B, -1, cout, 2).sum(1) for g in (gns[tile], gns[ref_tile]))
        assert torch.isfinite(s16).all()
        assert float((s16 - s3).abs().max() / s3.abs().max()) < 1e-5
        want = torch.stack([ref.double().reshape(B, H * W, cout).sum(1), (ref.double() ** 2).reshape(B, H * W, cout).sum(1)], -1)
        assert float((s16.cpu() - want).abs().max() / want.abs().max()) < (1e-5 if out_f32 else 5e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # B, IH, IW, cin, cout, bias, residual, gn_partial, out_f32, parts, splitk
    (2, 16, 32, 128, 128, True, False, True, False, 1, 1),      # two source blocks per image, every image border, fused GroupNorm partials
    (1, 16, 16, 192, 256, False, True, False, False, 1, 1),     # 3 chunks (the ring of three slices wraps mid-unit), residual, two column tiles
    (3, 32, 16, 64, 320, True, False, False, True, 1, 1),       # ONE chunk (the next chunk's slices are out-of-range loads), ragged last column tile, fp32 output
    (1, 48, 32, 320, 128, True, False, True, True, 1, 1),       # 5 chunks
    (2, 16, 16, 128, 128, True, False, False, True, 3, 1),      # the parity modes' three-part product: K = 4 * 3 C per phase
    (1, 32, 32, 256, 128, True, False, False, False, 1, 2),     # split-K over the chunks
])
def test_halo_conv_subpixel_upsample2x(dtype, case):
    """upsample2x == 2: nearest-2x upsample + 3x3 conv as four 2x2 convolutions of the source image with pre-summed weights
    (include/edtr_hip.h: w_phase_stride; reference model/unet.py:70-79, model/vae.py:35-39).  Checked against (a) torch's
    F.interpolate + F.conv2d in fp32 on the unsummed fp32 kernel, (b) the 9-tap UP2 gather of the same kernel family."""
    from edtr_amd import lib as L
    ops = _ops()
    d = dev()
    B, IH, IW, cin, cout, use_bias, use_res, use_gn, out_f32, parts, splitk = case
    H, W = 2 * IH, 2 * IW
    M = B * H * W
    x32 = rnd((B * IH * IW, cin), 311)
    w4 = rnd((cout, cin, 3, 3), 312, 1 / math.sqrt(9 * cin))
    bias = rnd((cout,), 313).to(d) if use_bias else None
    res = rnd((M, cout), 314).to(dtype).to(d) if use_res else None
    if parts == 3:        # [hi | lo | hi] activation against [Wh | Wh | Wl] weights: the product is exact to ~16 / 22 bits
        hi = x32.to(dtype)
        lo = (x32 - hi.float()).to(dtype)
        xd = torch.cat([hi, lo, hi], dim=1).contiguous().to(d)
        x_eff = x32
        store_dt = ops.F32S if dtype == torch.bfloat16 else ops.MIXED
    else:
        xd = x32.to(dtype).to(d)
        x_eff = x32.to(dtype).float()
        store_dt = dtype
    Ce = parts * cin
    wsp = ops.pack_conv_weight_subpixel(w4, store_dt, cin_pad=cin, parts=parts).to(d)
    assert wsp.shape == (4 * cout, 4 * Ce)
    w9 = ops.pack_conv_weight(w4, store_dt, cin_pad=cin, parts=parts).to(d)
    outs, gns = {}, {}
    for form in ("subpixel", "gather"):
        out = torch.full((M, cout), float("nan"), dtype=torch.float32 if out_f32 else dtype, device=d)
        gn = torch.full((M // 128, cout, 2), float("nan"), dtype=torch.float32, device=d) if use_gn else None
        sk = splitk if form == "subpixel" else 1
        ws = torch.empty((sk * M * cout,), dtype=torch.float32, device=d) if sk > 1 else None
        kw = dict(w=wsp, ldw=4 * Ce, w_phase_stride=cout * 4 * Ce, spatial=(IH, IW, H, W, 1, 1, 1, 2), tile=16) if form == "subpixel" else \
            dict(w=w9, ldw=9 * Ce, spatial=(IH, IW, H, W, 1, 1, 1, 1), tile=3)
        ops.launch(ops.make_igemm(dtype=dtype, a1=xd, out=out, taps=9, M=M, N=cout, C1=Ce, ld1=Ce, ldc=cout, bias_n=bias,
                                  rows_per_image=H * W, residual=res, ldr=cout, out_f32=out_f32, gn_partial=gn if sk == 1 else None,
                                  splitk=sk, workspace=ws, **kw))
        outs[form], gns[form] = out, gn
    torch.cuda.synchronize()
    xs = x_eff.reshape(B, IH, IW, cin).permute(0, 3, 1, 2)
    # (a) the kernel's own arithmetic in fp32: four 2x2 convolutions of the source image with the PACKED (rounded) phase weights
    p6 = wsp.float().cpu().reshape(4, cout, 2, 2, parts, cin)
    w_ph = p6[..., 0, :] + (p6[..., 2, :] if parts == 3 else 0.0)                      # parts 3: [Wh | Wh | Wl] -> Wh + Wl
    xp = F.pad(xs, (1, 1, 1, 1))
    ref = torch.zeros((B, cout, H, W))
    for py in (0, 1):
        for px in (0, 1):
            win = xp[:, :, py:py + IH + 1, px:px + IW + 1]                              # source rows s + py - 1 .. s + py
            ref[:, :, py::2, px::2] = F.conv2d(win, w_ph[2 * py + px].permute(0, 3, 1, 2))
    if use_bias:
        ref = ref + bias.cpu()[None, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(M, cout)
    # (b) the reference's formulation on the unsummed fp32 kernel
    ref_up = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w4, bias.cpu() if use_bias else None, padding=1)
    ref_up = ref_up.permute(0, 2, 3, 1).reshape(M, cout)
    if use_res:
        ref, ref_up = ref + res.float().cpu(), ref_up + res.float().cpu()
    assert torch.isfinite(outs["subpixel"].float()).all()
    tol = (3e-5 if parts == 3 else 2e-5) if out_f32 else TOL[dtype]
    assert rel(outs["subpixel"], ref) < tol
    # against the unsummed fp32 kernel the one-part form carries its weights' 16-bit rounding (as does the 9-tap gather)
    wr = {torch.bfloat16: 4e-3, torch.float16: 5e-4}[dtype]
    assert rel(outs["subpixel"], ref_up) < (3e-5 if parts == 3 else wr) + (0 if out_f32 else TOL[dtype])
    assert rel(outs["gather"], ref_up) < (3e-5 if parts == 3 else wr) + (0 if out_f32 else TOL[dtype])
    if use_gn and splitk == 1:
        s_sp, s_ga = (g.double().reshape(B, -1, cout, 2).sum(1) for g in (gns["subpixel"], gns["gather"]))
        assert torch.isfinite(s_sp).all()
        want = torch.stack([ref.double().reshape(B, H * W, cout).sum(1), (ref.double() ** 2).reshape(B, H * W, cout).sum(1)], -1)
        assert float((s_sp.cpu() - want).abs().max() / want.abs().max()) < (1e-4 if out_f32 else 5e-3)
    # shapes the geometry does not take are refused, not silently remapped
    with pytest.raises(RuntimeError):
        ops.launch(ops.make_igemm(dtype=dtype, a1=xd, w=wsp, out=outs["subpixel"], taps=9, M=M, N=cout, C1=Ce, ld1=Ce, ldw=4 * Ce, ldc=cout,
                                  w_phase_stride=cout * 4 * Ce, spatial=(IH, IW, H, W, 1, 1, 1, 2), out_f32=out_f32, tile=3))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,cin,cout,S,extras", [(4, 128, 128, 1, True), (8, 320, 256, 3, True), (4, 1280, 128, 10, False), (12, 192, 320, 2, True),
                                                 (8, 384, 128, 6, False)])
def test_halo_conv_8x8_images(dtype, B, cin, cout, S, extras):
    """The halo kernel's third geometry: FOUR WHOLE 8 x 8 images per workgroup (the 8x8 latent level, where the weights are 20x
    the activations): every border pixel of every image, split-K over chunks incl. uneven chunk counts, time-embedding row +
    residual + SiLU in the epilogue / the split-K reducer — against conv2d on the rounded operands and against tile 3."""
    import torch.nn.functional as F
    from edtr_amd import lib as L
    ops = _ops()
    d = dev()
    H = W = 8
    M = B * 64
    x = rnd((M, cin), 331).to(dtype)
    w = rnd((cout, 9 * cin), 332, 1 / math.sqrt(9 * cin)).to(dtype)
    bias = rnd((cout,), 333).to(d)
    rv = rnd((B, cout), 334).to(d) if extras else None
    res = rnd((M, cout), 335).to(dtype).to(d) if extras else None
    outs = {}
    for t in (3, 16):
        out = torch.full((M, cout), float("nan"), dtype=dtype, device=d)
        ws = torch.empty(S * M * cout, dtype=torch.float32, device=d) if S > 1 else None
        ops.launch(ops.make_igemm(dtype=dtype, a1=x.to(d), w=w.to(d), out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, rowvec=rv, rowvec_ld=cout if extras else 0,
                                  rows_per_image=64, act=L.ACT_SILU if extras else L.ACT_NONE, residual=res, ldr=cout, tile=t, splitk=S,
                                  workspace=ws))
        outs[t] = out
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().reshape(B, H, W, cin).permute(0, 3, 1, 2), w.float().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2), bias.cpu(),
                   padding=1)
    if extras:
        ref = F.silu(ref + rv.cpu()[:, :, None, None])
    ref = ref.permute(0, 2, 3, 1).reshape(M, cout)
    if extras:
        ref = ref + res.float().cpu()
    assert torch.isfinite(outs[16].float()).all()
    assert rel(outs[16], ref) < TOL[dtype] and rel(outs[16], outs[3]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_halo_conv_splitk(dtype):
    """Split-K over the 64-channel chunks (the 16x16-latent convolutions: 8 patches x 10 column tiles fill a third of the chip):
    fp32 partial slabs + the shared reducer, with bias / residual applied there."""
    ops = _ops()
    d = dev()
    B, H, W, cin, cout, S = 2, 16, 16, 640, 256, 3
    M = B * H * W
    x = rnd((M, cin), 321).to(dtype).to(d)
    w = rnd((cout, 9 * cin), 322, 1 / math.sqrt(9 * cin)).to(dtype).to(d)
    bias = rnd((cout,), 323).to(d)
    res = rnd((M, cout), 324).to(dtype).to(d)
    outs = []
    for t, sk in ((3, 1), (16, S), (3, S)):
        out = torch.full((M, cout), float("nan"), dtype=dtype, device=d)
        ws = torch.empty(sk * M * cout, dtype=torch.float32, device=d) if sk > 1 else None
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), bias_n=bias, residual=res, ldr=cout, tile=t, splitk=sk, workspace=ws))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[1].float()).all()
    assert rel(outs[1], outs[0]) < TOL[dtype] and rel(outs[1], outs[2]) < TOL[dtype]


def test_halo_conv_is_the_automatic_choice_and_rejects_other_shapes():
    """tile 0 picks the halo tile for >= 48 units of 16x16 pixels x 128 channels (N a multiple of 128); an explicit tile 16
    on a shape it cannot run (stride 2, image not 16-pixel aligned) is an error, not a silent fallback."""
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    B, H, W, cin, cout = 3, 64, 64, 64, 128                      # 3 * 16 patches = 48 units
    M = B * H * W
    x = rnd((M, cin), 311).to(dtype).to(d)
    w = rnd((cout, 9 * cin), 312, 1 / math.sqrt(9 * cin)).to(dtype).to(d)
    outs = []
    for t in (16, 0, 3):
        out = torch.empty((M, cout), dtype=dtype, device=d)
        ops.launch(ops.make_igemm(dtype=dtype, a1=x, w=w, out=out, taps=9, M=M, N=cout, C1=cin, ld1=cin, ldw=9 * cin, ldc=cout,
                                  spatial=(H, W, H, W, 1, 1, 1, 0), tile=t))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])                         # tile 0 (auto) ran the halo tile
    assert rel(outs[0], outs[2]) < TOL[dtype]
    for spatial, m in (((64, 64, 32, 32, 2, 1, 1, 0), B * 32 * 32), ((24, 24, 24, 24, 1, 1, 1, 0), B * 24 * 24)):
        xs = rnd((B * spatial[0] * spatial[1], cin), 313).to(dtype).to(d)
        with pytest.raises(RuntimeError):
            ops.launch(ops.make_igemm(dtype=dtype, a1=xs, w=w, out=torch.empty((m, cout), dtype=dtype, device=d), taps=9, M=m, N=cout,
                                      C1=cin, ld1=cin, ldw=9 * cin, ldc=cout, spatial=spatial, tile=16))


# ---------------------------------------------------------------------------------------------------------------------
# Tile order (`tile_coords` in igemm.hip): the blockIdx -> (row tile, column tile) map switches with the operand sizes — column tile
# fastest, row tile fastest (weights > 4x the activations) or 8x8 super-blocks (neither operand fits an L2, both tile counts
# multiples of 8).  Every order must cover every tile exactly once: full-output comparison on one shape per rule and per kernel.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [3, 6, 8, 1])
@pytest.mark.parametrize("M,N,K", [(1024, 1024, 128),      # 8 x 8 tiles of 128: the super-block order (model: traffic halves)
                                   (2048, 2048, 128),      # 16 x 16 tiles: four super-blocks
                                   (128, 1024, 2048),      # weights 16x the activations: row tile fastest
                                   (384, 1280, 1024),      # row tile fastest with ragged counts (3 x 10 tiles)
                                   (1536, 320, 256)])      # default order
def test_gemm_tile_orders_cover_every_tile(M, N, K, tile):
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    a = rnd((M, K), 331).to(dtype)
    w = rnd((N, K), 332, 1 / math.sqrt(K)).to(dtype)
    ref = a.float() @ w.float().t()
    out = torch.full((M, N), float("nan"), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=a.to(d), w=w.to(d), out=out, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N, tile=tile))
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()                # an uncovered tile stays NaN
    assert rel(out.float(), ref) < TOL[dtype]


# ---------------------------------------------------------------------------------------------------------------------
# edtr_flash_attn512 (attn512.hip, round 5): the VAE AttnBlock's single-head attention with head width 512 in one launch
# (reference model/vae.py:279-308) against fp32 softmax attention of the same 16-bit operands.  The bench shape (B, 4096, 512), the
# untiled 1024^2 shape (1, 16384, 512), the tiled VAE's 40 x 40 tile (N = 1600: a ragged last query block), one tile only, odd
# tile counts, sharp rows (the deferred-rescale path), fp32 output.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,sharp,out_f32", [(2, 4096, 1.0, False), (1, 16384, 1.0, False), (3, 1600, 1.0, False), (2, 32, 1.0, False), (1, 96, 1.0, True),
                                               (2, 1024, 12.0, False), (1, 2080, 40.0, True)])
def test_flash_attn512_vs_fp32_softmax(dtype, B, N, sharp, out_f32):
    ops = _ops()
    d = dev()
    C = 512
    ld = 2 * C                                              # q | k side by side in one projection buffer, as the emitter lays them out
    ldv = N + 8                                             # V^T rows with a pad
    qk = rnd((B * N, ld), 601).to(dtype)
    qk[:, :C] *= sharp                                      # sharper rows: scores of +-50 and more, the running maximum moves late
    vt = rnd((B * C, ldv), 602).to(dtype)
    out = torch.full((B * N, C), float("nan"), dtype=torch.float32 if out_f32 else dtype, device=d)
    qd, vd = qk.to(d), vt.to(d)
    ops.launch(ops.make_flash_attn512(dtype=dtype, q=qd[:, :C], k=qd[:, C:], vt=vd, out=out, B=B, N=N, q_bs=N * ld, q_ld=ld, k_bs=N * ld, k_ld=ld,
                                      vt_bs=C * ldv, vt_ld=ldv, o_bs=N * C, o_ld=C, scale=1.0 / math.sqrt(C), out_f32=out_f32))
    torch.cuda.synchronize()
    q = qd[:, :C].float().reshape(B, N, C)
    k = qd[:, C:].float().reshape(B, N, C)
    v = vd.float().reshape(B, C, ldv)[:, :, :N].transpose(1, 2)
    worst = 0.0
    for b in range(B):                                       # (row blocks: the 16384^2 score matrix is the thing this kernel avoids)
        for r0 in range(0, N, 4096):
            w = torch.softmax((q[b, r0:r0 + 4096].double() @ k[b].double().t()) / math.sqrt(C), dim=-1)
            ref = (w @ v[b].double()).float()
            got = out[b * N + r0:b * N + min(r0 + 4096, N)].float()
            assert torch.isfinite(got).all()
            worst = max(worst, rel(got, ref))
    MEASURED[f"flash_attn512[{dtype}-{B}-{N}-{sharp}]"] = worst
    # measured (round 6, profiles/r06/per_kernel_errors.json): bf16 <= 2.36e-3, fp16 <= 2.95e-4 over these shapes — the file's envelope (<= 1.5 x)
    assert worst < TOL[dtype], worst


def test_flash_attn512_rejects_what_it_cannot_run():
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    x = torch.zeros((40, 1024), dtype=dtype, device=d)
    vt = torch.zeros((512, 48), dtype=dtype, device=d)
    out = torch.zeros((40, 512), dtype=dtype, device=d)
    with pytest.raises(RuntimeError):                        # 40 keys: not whole 32-key tiles
        ops.launch(ops.make_flash_attn512(dtype=dtype, q=x[:, :512], k=x[:, 512:], vt=vt, out=out, B=1, N=40, q_bs=40 * 1024, q_ld=1024, k_bs=40 * 1024,
                                          k_ld=1024, vt_bs=512 * 48, vt_ld=48, o_bs=40 * 512, o_ld=512, scale=1.0))
    assert not ops.flash_attn512_ok(40, 512) and not ops.flash_attn512_ok(4096, 256) and ops.flash_attn512_ok(1600, 512)


# ---------------------------------------------------------------------------------------------------------------------
# edtr_ffn (ffn.hip, round 6): x + W2 GEGLU(W1 LayerNorm(x) + b1) + b2 of a BasicTransformerBlock in one launch (reference
# model/attention.py:20-47, 233) against torch LayerNorm -> linear -> chunk -> x * gelu(gate) -> linear -> + x in fp32 on the same
# 16-bit rows.  Row counts: one workgroup, the bench level (32768 = 8 x 64 x 64), a non-power-of-two multiple of 128; rows with a
# large mean (the folded mean term must cancel) and a padded row stride.
# ---------------------------------------------------------------------------------------------------------------------
def _ffn_operands(ops, dtype, seed=0):
    D, H = ops.FFN_D, ops.FFN_H
    gamma, beta = 1 + 0.2 * rnd((D,), 700 + seed), 0.3 * rnd((D,), 701 + seed)
    w1 = rnd((2 * H, D), 702 + seed, 1 / math.sqrt(D))
    b1 = 0.5 * rnd((2 * H,), 703 + seed)
    w2 = rnd((D, H), 704 + seed, 1 / math.sqrt(H))
    b2 = 0.5 * rnd((D,), 705 + seed)
    perm = ops.geglu_perm(H)
    w1p = ops.pack_linear_weight(w1[perm] * gamma[None, :], dtype)
    c1 = w1p.float().sum(1)
    c2b = w1[perm] @ beta + b1[perm]
    return dict(gamma=gamma, beta=beta, w1=w1, b1=b1, w2=w2, b2=b2, w1p=w1p, w2p=ops.pack_ffn_w2(w2, dtype), cst=ops.pack_ffn_constants(c2b))


def _ffn_reference(x16, o):
    xf = x16.float()
    ln = F.layer_norm(xf, (xf.shape[1],), o["gamma"], o["beta"], 1e-5)
    hcat = ln @ o["w1"].t() + o["b1"]
    val, gate = hcat.chunk(2, dim=-1)
    return xf + (val * F.gelu(gate)) @ o["w2"].t() + o["b2"]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,ldx,shift", [(128, 320, 0.0), (32768, 320, 0.7), (640, 328, 3.0)])
def test_ffn_fused_vs_fp32_reference(dtype, M, ldx, shift):
    ops = _ops()
    d = dev()
    o = _ffn_operands(ops, dtype)
    D = ops.FFN_D
    xs = torch.zeros((M, ldx), dtype=dtype)
    xs[:, :D] = (rnd((M, D), 710, 1.5) + shift).to(dtype)
    xd = xs.to(d)
    out = torch.full((M, ldx), float("nan"), dtype=dtype, device=d)
    assert ops.ffn_ok(max(M, 16384), D, 4 * D)      # (the host rule also wants enough rows to fill half the chip)
    ops.launch(ops.make_ffn(dtype=dtype, x=xd[:, :D], ldx=ldx, M=M, w1=o["w1p"].to(d), w2=o["w2p"].to(d), cst=o["cst"].to(d), b2=o["b2"].to(d),
                            out=out[:, :D], ldo=ldx))
    torch.cuda.synchronize()
    got = out[:, :D].float().cpu()
    assert torch.isfinite(got).all()
    ref = _ffn_reference(xs[:, :D], o)
    # the branch alone (residual removed): the sum is dominated by x, which would hide an error of the feed-forward part
    xb = xs[:, :D].float()
    e_all = rel(got, ref)
    e_branch = float(((got - xb) - (ref - xb)).double().norm() / (ref - xb).double().norm())      # (not through rel(): the envelope is about stored tensors)
    MEASURED[f"ffn_branch_only[{'b16' if dtype == torch.bfloat16 else 'h16'}-{M}]"] = e_branch
    assert e_all < TOL[dtype], (e_all, e_branch)
    assert e_branch < (1.2e-2 if dtype == torch.bfloat16 else 1.6e-3), (e_all, e_branch)      # (the branch is a rounded 16-bit sum with x: its own error is ~ 2^-9 / 2^-12 of |x| / |branch|)
    if ldx > D:
        assert bool(torch.isnan(out[:, D:].float()).all())      # pad columns untouched


def test_ffn_matches_the_two_gemm_form():
    """The fused launch against the product path it replaces (LayerNorm launch -> GEGLU edtr_igemm -> edtr_igemm + residual) on the same
    operands: the two differ by 16-bit roundings of the hidden tensor only."""
    from edtr_amd import lib as L
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    o = _ffn_operands(ops, dtype, seed=20)
    M, D, H = 1024, ops.FFN_D, ops.FFN_H
    x = (rnd((M, D), 730, 1.5) + 0.4).to(dtype).to(d)
    fused = torch.empty((M, D), dtype=dtype, device=d)
    ops.launch(ops.make_ffn(dtype=dtype, x=x, ldx=D, M=M, w1=o["w1p"].to(d), w2=o["w2p"].to(d), cst=o["cst"].to(d), b2=o["b2"].to(d), out=fused, ldo=D))
    xn = torch.empty_like(x)
    ops.launch(ops.make_layernorm(dtype=dtype, x=x, rows=M, C=D, ldx=D, gamma=o["gamma"].to(d), beta=o["beta"].to(d), eps=1e-5, y=xn, ldy=D))
    perm = ops.geglu_perm(H)
    g = torch.empty((M, H), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=xn, w=ops.pack_linear_weight(o["w1"][perm], dtype).to(d), out=g, M=M, N=2 * H, C1=D, ld1=D, ldw=D, ldc=H,
                              bias_n=o["b1"][perm].to(d), act=L.ACT_GEGLU))
    two = torch.empty((M, D), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=g, w=ops.pack_linear_weight(o["w2"], dtype).to(d), out=two, M=M, N=D, C1=H, ld1=H, ldw=H, ldc=D,
                              bias_n=o["b2"].to(d), residual=x, ldr=D))
    torch.cuda.synchronize()
    ref = _ffn_reference(x.cpu(), o)
    e_f, e_t = rel(fused.float().cpu(), ref), rel(two.float().cpu(), ref)
    assert e_f < TOL[dtype] and e_f < 1.3 * e_t + 1e-4, (e_f, e_t)


def test_ffn_rejects_what_it_cannot_run():
    from edtr_amd import lib as L
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    o = _ffn_operands(ops, dtype)
    x = torch.zeros((192, 320), dtype=dtype, device=d)
    out = torch.zeros_like(x)
    with pytest.raises(RuntimeError):                        # 192 rows: not whole 128-row workgroups
        ops.launch(ops.make_ffn(dtype=dtype, x=x, ldx=320, M=192, w1=o["w1p"].to(d), w2=o["w2p"].to(d), cst=o["cst"].to(d), b2=o["b2"].to(d), out=out, ldo=320))
    with pytest.raises(RuntimeError):                        # in place: rows are re-read as the residual
        ops.launch(ops.make_ffn(dtype=dtype, x=x[:128], ldx=320, M=128, w1=o["w1p"].to(d), w2=o["w2p"].to(d), cst=o["cst"].to(d), b2=o["b2"].to(d), out=x[:128], ldo=320))
    assert not ops.ffn_ok(192, 320, 1280) and not ops.ffn_ok(8192, 640, 2560) and ops.ffn_ok(16384, 320, 1280)


# ---------------------------------------------------------------------------------------------------------------------
# edtr_lin320 (lin320.hip, round 6): the K = 320 linear layers of the 64 x 64-latent transformer blocks as a row-resident product — a wave
# keeps 32 token rows in registers (normalised in place where a LayerNorm is asked for), the weights stream through LDS in fragment
# order, the 64-column groups leave row-major through a wave-private fp32 tile.  Reference: model/attention.py:171, 195, 224-232, 283-302.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,ln,res,pad", [(128, 320, False, False, 0), (384, 320, True, False, 0), (256, 320, False, True, 64), (128, 960, True, False, 0),
                                            (640, 64, True, True, 8), (128, 1024, False, True, 0)])
def test_lin320_vs_fp32_reference(dtype, M, N, ln, res, pad):
    ops = _ops()
    d = dev()
    K = ops.LIN320_K
    x = torch.full((M, K + pad), float("nan"), dtype=dtype)
    x[:, :K] = (rnd((M, K), 801, 1.4) + 0.3).to(dtype)
    w = rnd((N, K), 802, 1 / math.sqrt(K))
    gamma, beta = 1 + 0.2 * rnd((K,), 803), 0.2 * rnd((K,), 804)
    bias = rnd((N,), 805)
    alpha = 0.31
    r = rnd((M, N), 806).to(dtype) if res else None
    if ln:        # gamma into the columns, alpha * W beta (+ bias) into the additive row: what the engine's weight store hands the kernel
        wp = ops.pack_lin320_w(w * gamma[None, :], dtype)
        cvec = alpha * (w @ beta) + bias
    else:
        wp = ops.pack_lin320_w(w, dtype)
        cvec = bias
    out = torch.full((M, N + pad), float("nan"), dtype=dtype, device=d)
    xd = x.to(d)
    ops.launch(ops.make_lin320(dtype=dtype, x=xd[:, :K], ldx=K + pad, M=M, N=N, w=wp.to(d), cvec=cvec.to(d), alpha=alpha, ln=ln, eps=1e-5,
                               residual=None if r is None else r.to(d), ldr=N, out=out[:, :N], ldo=N + pad))
    torch.cuda.synchronize()
    xf = x[:, :K].float()
    if ln:
        xn = F.layer_norm(xf, (K,), None, None, 1e-5)
        # the kernel rounds the normalised rows to 16 bits (what the LayerNorm launch hands its GEMM) and multiplies 16-bit weights
        ref = alpha * (xn @ (w * gamma[None, :]).T) + cvec
    else:
        ref = alpha * (xf @ w.T) + cvec
    if res:
        ref = ref + r.float()
    got = out[:, :N].float().cpu()
    assert torch.isfinite(got).all()
    assert rel(got, ref) < TOL[dtype]
    if pad:
        assert bool(torch.isnan(out[:, N:].float()).all())      # pad columns untouched


@pytest.mark.parametrize("dtype", DTYPES)
def test_lin320_fused_qkv_with_transposed_v(dtype):
    """The [Wq; Wk; Wv] projection behind norm1: q / k row-major with alpha, v^T transposed per image with vt_alpha (edtr_igemm's vt_out layout)."""
    ops = _ops()
    d = dev()
    K, C, B, Ntok = ops.LIN320_K, 320, 3, 128
    M, N = B * Ntok, 3 * C
    x = (rnd((M, K), 821, 1.4) + 0.3).to(dtype)
    w = rnd((N, K), 822, 1 / math.sqrt(K))
    gamma, beta = 1 + 0.2 * rnd((K,), 823), 0.2 * rnd((K,), 824)
    alpha = 0.6
    acol = torch.full((N,), alpha)
    acol[2 * C:] = 1.0
    cvec = acol * (w @ beta)
    qk = torch.full((M, 2 * C), float("nan"), dtype=dtype, device=d)
    vt = torch.full((B * C, Ntok + 8), float("nan"), dtype=dtype, device=d)
    ops.launch(ops.make_lin320(dtype=dtype, x=x.to(d), ldx=K, M=M, N=N, w=ops.pack_lin320_w(w * gamma[None, :], dtype).to(d), cvec=cvec.to(d), alpha=alpha,
                               ln=True, eps=1e-5, out=qk, ldo=2 * C, vt_out=vt, vt_col0=2 * C, vt_ld=Ntok + 8, vt_alpha=1.0, rows_per_image=Ntok))
    torch.cuda.synchronize()
    ref = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.T            # [M, 3C]
    assert rel(qk.float().cpu(), alpha * ref[:, :2 * C]) < TOL[dtype]
    vref = ref[:, 2 * C:].reshape(B, Ntok, C).permute(0, 2, 1).reshape(B * C, Ntok)
    assert rel(vt[:, :Ntok].float().cpu(), vref) < TOL[dtype]
    assert bool(torch.isnan(vt[:, Ntok:].float()).all())                     # pad columns untouched


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("res", [False, True])
def test_lin320_applies_a_groupnorm_table_to_its_rows(dtype, res):
    """proj_in(norm(x)): the (scale, shift) table of edtr_gn_table applied to the rows in registers, per image (two images of 256 rows, a
    workgroup never leaves its image), against x * scale + shift rounded to 16 bits (what edtr_gn_apply stores) and torch fp32."""
    ops = _ops()
    d = dev()
    K, N, B, HW = ops.LIN320_K, 320, 2, 256
    M = B * HW
    x = (rnd((M, K), 831, 1.4) + 0.3).to(dtype)
    w = rnd((N, K), 832, 1 / math.sqrt(K))
    bias = rnd((N,), 833)
    table = torch.stack([1 + 0.3 * rnd((B, K), 834), 0.3 * rnd((B, K), 835)], dim=-1).contiguous()      # [B][K][2]
    r = rnd((M, N), 836).to(dtype) if res else None
    out = torch.full((M, N), float("nan"), dtype=dtype, device=d)
    ops.launch(ops.make_lin320(dtype=dtype, x=x.to(d), ldx=K, M=M, N=N, w=ops.pack_lin320_w(w, dtype).to(d), cvec=bias.to(d), gn_table=table.to(d),
                               rows_per_image=HW, residual=None if r is None else r.to(d), ldr=N, out=out, ldo=N))
    torch.cuda.synchronize()
    xn = (x.float().reshape(B, HW, K) * table[:, None, :, 0] + table[:, None, :, 1]).reshape(M, K)
    ref = xn.to(dtype).float() @ w.to(dtype).float().T + bias + (r.float() if res else 0)
    assert rel(out.float().cpu(), ref.to(dtype).float()) < 0.3 * TOL[dtype]      # same roundings as the reference composition: only the fp32 summation order differs
    assert rel(out.float().cpu(), xn @ w.T + bias + (r.float() if res else 0)) < TOL[dtype]
    with pytest.raises(RuntimeError):                               # 192 rows per image: a workgroup would straddle two images
        ops.launch(ops.make_lin320(dtype=dtype, x=x.to(d), ldx=K, M=384, N=N, w=ops.pack_lin320_w(w, dtype).to(d), gn_table=table.to(d), rows_per_image=192,
                                   out=out, ldo=N))


def test_lin320_matches_the_layernorm_plus_igemm_form():
    """Against the product path it replaces on the same operands (edtr_layernorm -> edtr_igemm with alpha / bias / residual): both round
    the normalised rows to 16 bits and accumulate in fp32; only the summation order differs."""
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    M, N, K = 2048, 320, ops.LIN320_K
    x = (rnd((M, K), 811, 1.4) + 0.3).to(dtype).to(d)
    w = rnd((N, K), 812, 1 / math.sqrt(K))
    gamma, beta = (1 + 0.2 * rnd((K,), 813)).to(d), (0.2 * rnd((K,), 814)).to(d)
    r = rnd((M, N), 815).to(dtype).to(d)
    alpha = 0.5
    xn = torch.empty_like(x)
    ops.launch(ops.make_layernorm(dtype=dtype, x=x, rows=M, C=K, ldx=K, gamma=gamma, beta=beta, eps=1e-5, y=xn, ldy=K))
    two = torch.empty((M, N), dtype=dtype, device=d)
    ops.launch(ops.make_igemm(dtype=dtype, a1=xn, w=ops.pack_linear_weight(w, dtype).to(d), out=two, M=M, N=N, C1=K, ld1=K, ldw=K, ldc=N, alpha=alpha,
                              residual=r, ldr=N))
    one = torch.empty((M, N), dtype=dtype, device=d)
    cvec = (alpha * (w @ beta.cpu())).to(d)
    ops.launch(ops.make_lin320(dtype=dtype, x=x, ldx=K, M=M, N=N, w=ops.pack_lin320_w(w * gamma.cpu()[None, :], dtype).to(d), cvec=cvec, alpha=alpha, ln=True,
                               eps=1e-5, residual=r, ldr=N, out=one, ldo=N))
    torch.cuda.synchronize()
    ref = alpha * (F.layer_norm(x.float().cpu(), (K,), gamma.cpu(), beta.cpu(), 1e-5) @ w.T) + r.float().cpu()
    e1, e2 = rel(one.float().cpu(), ref), rel(two.float().cpu(), ref)
    assert e1 < TOL[dtype] and e1 < 1.3 * e2 + 1e-4, (e1, e2)


def test_lin320_rejects_what_it_cannot_run():
    ops = _ops()
    d = dev()
    dtype = torch.bfloat16
    x = torch.zeros((256, 320), dtype=dtype, device=d)
    w = torch.zeros((320 * 320,), dtype=dtype, device=d)
    out = torch.zeros((256, 320), dtype=dtype, device=d)
    with pytest.raises(RuntimeError):                        # 192 rows: not whole 128-row workgroups
        ops.launch(ops.make_lin320(dtype=dtype, x=x, ldx=320, M=192, N=320, w=w, out=out, ldo=320))
    with pytest.raises(RuntimeError):                        # N = 96: not whole 64-column groups
        ops.launch(ops.make_lin320(dtype=dtype, x=x, ldx=320, M=256, N=96, w=w, out=out, ldo=320))
    with pytest.raises(RuntimeError):                        # in place
        ops.launch(ops.make_lin320(dtype=dtype, x=x, ldx=320, M=256, N=320, w=w, out=x, ldo=320))
    assert not ops.lin320_ok(32768, 320, 640) and not ops.lin320_ok(192, 320, 320) and ops.lin320_ok(32768, 320, 320) and ops.lin320_ok(32768, 960, 320)
    assert ops.lin320_ok(16384, 320, 320, ln=True) and not ops.lin320_ok(16384, 320, 320) and not ops.lin320_ok(16384, 960, 320, ln=True)


def test_zz_measured_error_envelope():
    """Bookkeeping (runs last in this file): the largest error each dtype's kernels measured against their torch references —
    TOL above is held to <= 1.5 x these."""
    import json
    import os
    worst = {}
    for key, v in MEASURED.items():
        dt = "bf16" if "bfloat16" in key or "dtype0" in key else ("fp16" if "float16" in key or "dtype1" in key else "other")
        if v > worst.get(dt, ("", 0.0))[1]:
            worst[dt] = (key, v)
    print("\n[per-kernel error envelope] " + "; ".join(f"{k}: {v[1]:.2e} ({v[0].split('::')[-1]})" for k, v in sorted(worst.items())))
    path = os.environ.get("EDTR_TEST_ERRLOG")
    if path:
        with open(path, "w") as f:
            json.dump({"worst": worst, "all": MEASURED}, f, indent=1)
