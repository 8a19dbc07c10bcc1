"""Parameter trees with the reference's exact state-dict key names / shapes / order, built from the flat specs
of edtr_amd/arch.py (so `load_state_dict(strict=True)` of SD-2.1 / EDTR checkpoints works)."""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import torch
from torch import nn

# parameters the reference zero-initialises (zero_module: model/unet.py:177,678; model/controlnet.py:261;
# model/attention.py:280)
_ZERO_INIT_SUFFIXES = ("out_layers.3.weight", "out_layers.3.bias", "proj_out.weight", "proj_out.bias")
_ZERO_INIT_PREFIXES = ("zero_convs.", "middle_block_out.", "out.2.")


_STRUCT_EPOCH = [0]      # bumped whenever a Parameter OBJECT is (re)registered anywhere in a ParamTree (params_fingerprint)


class _Node(nn.Module):
    def register_parameter(self, name, param):      # (nn.Module.__setattr__ of a Parameter and load_state_dict(assign=True) end here)
        _STRUCT_EPOCH[0] += 1
        super().register_parameter(name, param)


_SKIP_INIT = False


class skip_init:
    """Context manager: construct ParamTrees with uninitialised storage (the caller loads a checkpoint next)."""

    def __enter__(self):
        global _SKIP_INIT
        self.prev, _SKIP_INIT = _SKIP_INIT, True

    def __exit__(self, *exc):
        global _SKIP_INIT
        _SKIP_INIT = self.prev


def _is_zero_init(key: str, is_unet_like: bool) -> bool:
    if not is_unet_like:
        return False
    return key.endswith(_ZERO_INIT_SUFFIXES) or key.startswith(_ZERO_INIT_PREFIXES)


class ParamTree(_Node):
    def __init__(self, spec: Iterable[Tuple[str, Tuple[int, ...]]], unet_like: bool = False):
        super().__init__()
        gen = torch.Generator().manual_seed(0)
        for key, shape in spec:
            parts = key.split(".")
            node: nn.Module = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            if _SKIP_INIT:
                val = torch.empty(shape)
            elif _is_zero_init(key, unet_like):
                val = torch.zeros(shape)
            elif len(shape) >= 2:
                fan_in = int(math.prod(shape[1:]))
                bound = 1.0 / math.sqrt(fan_in)
                val = (torch.rand(shape, generator=gen) * 2 - 1) * bound
            elif parts[-1] == "weight":
                val = torch.ones(shape)
            else:
                val = torch.zeros(shape)
            node.register_parameter(parts[-1], nn.Parameter(val, requires_grad=False))

    def flat_params(self, prefix: str = "") -> Dict[str, torch.Tensor]:
        return {prefix + k: v for k, v in self.named_parameters()}


def params_fingerprint(module: nn.Module) -> Tuple:
    """Changes whenever any parameter is rewritten in place (load_state_dict / copy_ on the parameter itself), replaced, or
    moved.  It is built from the tensors' version counters, so a write that goes through ``.data`` (``p.data.copy_(...)``, EMA
    swaps) is NOT seen: after such a write call ``ControlLDM.release_engines()`` (packed weights, programs and hipGraphs are
    rebuilt on the next forward).  The parameter list is collected once and reused while no Parameter OBJECT has been
    (re)registered in any ParamTree since (`_STRUCT_EPOCH`: `sub.weight = nn.Parameter(...)`, `load_state_dict(assign=True)` on
    any sub-module — ADVICE r03: a cache keyed on the first parameter alone kept reading a replaced parameter's old version
    counter), so the per-forward cost stays one pass over a cached list (~1300 integer reads), not a walk of the module tree.
    (`module.to(device)` replaces ``.data`` in place, not the objects; the device of the first parameter is part of the result.)"""
    hit = module.__dict__.get("_fp_params")
    if hit is not None and hit[0] == _STRUCT_EPOCH[0]:
        # replacements that bypass register_parameter — `module._parameters[name] = p` (accelerate's set_module_tensor_to_device),
        # torch.__future__.set_overwrite_module_params_on_conversion — do not bump the epoch: a cheap spot check of the
        # first, the last and one middle parameter's identity catches whole-model conversions; a single `_parameters[...]` write
        # elsewhere still needs release_engines()
        plist = hit[1]
        first = next(iter(module.parameters()), None)
        if (first is None) != (not plist) or (plist and first is not plist[0]):
            hit = None
        elif len(plist) > 2:
            # last and middle: the last parameter of the last non-None child that has any (a reversed walk, no full traversal), and
            # the first parameter of the middle child; both compared by MEMBERSHIP in the cached identity set — a tied parameter is
            # deduplicated by module.parameters() and need not sit at plist[-1]
            ids = hit[3]
            kids = [m for m in module._modules.values() if m is not None]
            probes = []
            for sub in reversed(kids):
                last = None
                for last in sub.parameters():
                    pass
                if last is not None:
                    probes.append(last)
                    break
            if kids:
                mid = next(iter(kids[len(kids) // 2].parameters()), None)
                if mid is not None:
                    probes.append(mid)
            if any(id(p) not in ids for p in probes):
                hit = None
    if hit is None or hit[0] != _STRUCT_EPOCH[0]:
        plist = list(module.parameters())
        ids = frozenset(id(p) for p in plist)                                  # (the objects' identities: a replaced parameter may
        hit = (_STRUCT_EPOCH[0], plist, hash(tuple(id(p) for p in plist)), ids)  #  carry the same version counter as the old one)
        module.__dict__["_fp_params"] = hit
    ver, dev = 0, None
    for p in hit[1]:
        ver += p._version
        dev = p.device
    return (str(dev), ver, hit[2])
