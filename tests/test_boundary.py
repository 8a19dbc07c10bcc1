"""The Python side of the drop-in boundary: the reference's dotted import paths, its
`instantiate_from_config` resolution, and survival of `accelerate`'s prepare / unwrap_model / autocast wrapping
(main/det/test_edtr.py:29-31,95-96).  CPU tests construct and resolve only; the GPU tests run forwards."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def _run(code: str, pythonpath: str) -> str:
    env = dict(os.environ, PYTHONPATH=pythonpath)
    res = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return res.stdout


def test_standalone_shim_resolves_reference_import_paths():
    """`shim/` in front of sys.path: the YAML `target:` strings (configs/det/demo.yaml:21) and the scripts' import lines
    resolve to the edtr_amd classes through the reference's own resolution algorithm (utils/common.py:23-34)."""
    out = _run("""
        import edtr_amd.model.cldm as ours, edtr_amd.sampler, edtr_amd.diffusion
        from edtr_amd import synth
        from utils.common import instantiate_from_config, get_obj_from_str, pad_if_smaller, pad_to_multiples_of
        from model import ControlLDM, Diffusion, SwinIR, FrozenOpenCLIPEmbedder, AutoencoderKL, ControlNet, ControlledUnetModel
        from utils.sampler import SpacedSampler
        assert get_obj_from_str("model.cldm.ControlLDM") is ours.ControlLDM
        assert get_obj_from_str("model.gaussian_diffusion.Diffusion") is edtr_amd.diffusion.Diffusion
        assert get_obj_from_str("utils.sampler.SpacedSampler") is edtr_amd.sampler.SpacedSampler
        m = instantiate_from_config({"target": "model.cldm.ControlLDM", "params": synth.tiny_config()})
        assert isinstance(m, ours.ControlLDM) and len(m.control_scales) == 13
        d = instantiate_from_config({"target": "model.gaussian_diffusion.Diffusion",
                                     "params": dict(linear_start=0.00085, linear_end=0.0120, timesteps=1000)})
        s = SpacedSampler(d.betas)
        try:
            instantiate_from_config({"params": {}})
        except KeyError as e:
            assert "target" in str(e)
        else:
            raise AssertionError("missing target must raise KeyError")
        import torch
        assert tuple(pad_if_smaller(torch.zeros(1, 3, 100, 600), 512).shape) == (1, 3, 512, 600)
        assert tuple(pad_to_multiples_of(torch.zeros(1, 3, 513, 640), 64).shape) == (1, 3, 576, 640)
        print("OK")
    """, os.path.join(ROOT, "shim") + os.pathsep + ROOT)
    assert "OK" in out


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (build container only)")
def test_overlay_shim_on_a_reference_checkout():
    """edtr_amd.shim.install() on top of the real reference tree: the REFERENCE's `instantiate_from_config` resolves
    `model.cldm.ControlLDM` / `utils.sampler.SpacedSampler` to the MI355X classes while everything else (`model.resnet`,
    `utils.common` itself) still comes from the reference."""
    out = _run(f"""
        import sys
        sys.dont_write_bytecode = True
        sys.path.insert(0, {os.path.join(ROOT, "tools")!r})
        import ref_import
        ref_import.install_stubs()
        sys.path.insert(0, {REFERENCE!r})
        import edtr_amd.shim as shim
        shim.install()
        import edtr_amd.model.cldm as ours, edtr_amd.sampler
        from edtr_amd import synth
        from utils.common import instantiate_from_config          # the reference's function
        import utils.common
        assert utils.common.__file__.startswith({REFERENCE!r})
        m = instantiate_from_config({{"target": "model.cldm.ControlLDM", "params": synth.tiny_config()}})
        assert type(m) is ours.ControlLDM
        from model import ControlLDM, Diffusion
        from utils.sampler import SpacedSampler
        assert ControlLDM is ours.ControlLDM and SpacedSampler is edtr_amd.sampler.SpacedSampler
        import model.resnet
        assert model.resnet.__file__.startswith({REFERENCE!r})
        print("OK")
    """, ROOT)
    assert "OK" in out


def _tiny(dev, dtype=torch.float16):
    from edtr_amd import synth
    from edtr_amd.testing import build_synthetic_cldm
    return build_synthetic_cldm(synth.tiny_config(), dev, dtype)


def test_accelerate_prepare_and_unwrap_keep_the_module():
    """`accelerator.prepare(cldm)` / `unwrap_model` (main/det/test_edtr.py:95-96) on one process: the module keeps its
    class, its engine caches and the re-assignable `forward`."""
    from accelerate import Accelerator
    from edtr_amd.model import ControlLDM
    acc = Accelerator(cpu=True)
    cldm = _tiny("cpu")
    prepared = acc.prepare(cldm)
    pure = acc.unwrap_model(prepared)
    assert isinstance(pure, ControlLDM) and pure is cldm
    assert pure._cldm_engines is cldm._cldm_engines and len(pure._cldm_engines) == 0
    marker = lambda *a, **k: "patched"       # noqa: E731  (the tiled sampler re-assigns forward, utils/sampler.py:290)
    prepared.forward = marker
    assert prepared(None, None, None) == "patched"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pure.vae_encode(torch.zeros(1, 3, 64, 64), sample=False)


@pytest.mark.gpu
def test_accelerate_mixed_precision_wrapping_on_gpu():
    """Accelerator(mixed_precision='fp16').prepare wraps `forward` in autocast + fp32 output conversion; the HIP path ignores
    autocast (its dtype is compute_dtype), returns the same fp32 tensors, and keeps one cached engine per shape."""
    from accelerate import Accelerator
    from edtr_amd import synth
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    cldm = _tiny(dev)
    x = synth.synth_normal("acc:x", (2, 4, 16, 16)).to(dev)
    c_img = synth.synth_normal("acc:c", (2, 4, 16, 16)).to(dev)
    c_txt = synth.synth_input("acc:t", (2, 77, 64), -1.0, 1.0).to(dev)
    t = torch.tensor([200, 200], device=dev)
    ref = cldm(x, t, {"c_txt": c_txt, "c_img": c_img}).clone()
    acc = Accelerator(mixed_precision="fp16")
    prepared = acc.prepare(cldm)
    pure = acc.unwrap_model(prepared)
    with torch.no_grad(), acc.autocast():
        out = prepared(x, t, {"c_txt": c_txt, "c_img": c_img})
        z = pure.vae_encode(synth.synth_input("acc:img", (1, 3, 64, 64), -1.0, 1.0).to(dev), sample=False)
    torch.cuda.synchronize()
    assert out.dtype == torch.float32 and torch.equal(out, ref)
    assert z.dtype == torch.float32 and tuple(z.shape) == (1, 4, 8, 8)
    assert len(pure._cldm_engines) == 1


@pytest.mark.gpu
def test_context_cache_is_keyed_on_tensor_identity():
    """A different prompt embedding that happens to reuse the freed storage of the previous one (same address, same shape,
    version 0 — what the caching allocator does with clip.encode() results) must NOT hit the cross-attention K/V cache."""
    from edtr_amd import synth
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    cldm = _tiny(dev)
    x = synth.synth_normal("ctx:x", (1, 4, 16, 16)).to(dev)
    c_img = synth.synth_normal("ctx:c", (1, 4, 16, 16)).to(dev)
    t = torch.tensor([100], device=dev)
    host = [synth.synth_input(f"ctx:t{i}", (1, 77, 64), -1.0, 1.0) for i in range(2)]
    outs, ptrs = [], []
    for h in host:
        c_txt = h.to(dev)
        ptrs.append(c_txt.data_ptr())
        outs.append(cldm(x, t, {"c_txt": c_txt, "c_img": c_img}).clone())
        del c_txt
    fresh = _tiny(dev)
    want = fresh(x, t, {"c_txt": host[1].to(dev), "c_img": c_img})
    torch.cuda.synchronize()
    assert not torch.equal(outs[0], outs[1])
    assert torch.equal(outs[1], want), f"stale context reused (storage recycled: {ptrs[0] == ptrs[1]})"
    # same tensor object again -> cache hit (the context program does not re-run)
    eng = next(iter(cldm._cldm_engines.values()))
    c_keep = host[1].to(dev)
    cldm(x, t, {"c_txt": c_keep, "c_img": c_img})
    key = eng.ctx_key
    cldm(x, t, {"c_txt": c_keep, "c_img": c_img})
    assert eng.ctx_key is key


@pytest.mark.gpu
def test_q_sample_with_device_timesteps_matches_host_timesteps():
    """A GPU-resident `t` (demo.py:107-108) is read on the device; per-image timesteps included."""
    from edtr_amd import synth
    from edtr_amd.diffusion import Diffusion
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    d = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    x = synth.synth_normal("qs:x", (3, 4, 8, 8)).to(dev)
    n = synth.synth_normal("qs:n", (3, 4, 8, 8)).to(dev)
    for tl in ([200, 200, 200], [0, 500, 999]):
        th = torch.tensor(tl, dtype=torch.int64)
        a = d.q_sample(x, th, n)
        b = d.q_sample(x, th.to(dev), n)
        want = (d.sqrt_alphas_cumprod[th.to(dev)].view(3, 1, 1, 1) * x + d.sqrt_one_minus_alphas_cumprod[th.to(dev)].view(3, 1, 1, 1) * n)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        assert torch.allclose(b, want, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_tiled_vae_nan_guard():
    """NansException("vae") like utils/tilevae/tilevae.py:62-69,548 when a tile comes out NaN."""
    from edtr_amd import synth
    from edtr_amd.model.cldm import NansException
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    cldm = _tiny(dev)
    img = synth.synth_input("nan:img", (1, 3, 192, 256), -1.0, 1.0).to(dev)
    cldm.vae_encode(img, sample=False, tiled=True, tile_size=64)       # healthy input passes
    img[0, 0, 3, 5] = float("nan")
    with pytest.raises(NansException):
        cldm.vae_encode(img, sample=False, tiled=True, tile_size=64)


@pytest.mark.gpu
def test_synthetic_weights_hashed_on_the_device_equal_the_host():
    """edtr_amd.synth on the GPU (int64 hash + exactly representable fp32 results) reproduces the host bits: the goldens
    were generated from host-hashed weights, the GPU tests and bench.py hash on the device."""
    from edtr_amd import synth
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    for key, shape in (("unet.input_blocks.1.0.in_layers.2.weight", (320, 320, 3, 3)), ("vae.decoder.norm_out.weight", (128,)),
                       ("controlnet.time_embed.0.bias", (1280,)), ("unet.out.2.weight", (4, 320, 3, 3))):
        assert torch.equal(synth.synth_param(key, shape, device=dev).cpu(), synth.synth_param(key, shape))


@pytest.mark.gpu
def test_bench_rccl_path_on_one_rank():
    """bench.py with EDTR_BENCH_DIST=1 in a fresh child process: RCCL (backend "nccl") initialises on one rank, the packed
    weight store is broadcast in place, the timed region uses the barrier / all-reduce path, and ONE JSON line comes out."""
    import json
    import socket
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EDTR_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--size", "128", "--batch", "2",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["weight_broadcast"]["collectives"] >= 2 and out["weight_broadcast"]["GiB"] > 0
    assert "packed weights broadcast over RCCL" in res.stderr
