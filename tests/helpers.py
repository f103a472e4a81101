"""Shared helpers of the GPU parity tests."""
import os

import numpy as np
import torch

# 16-bit storage dtype of the library under test: fp16 (the default build) or bf16 (EVK_STORE=bf16, evoke_amd/hip.py)
F16_BUILD = os.environ.get('EVK_STORE', 'f16').lower() == 'f16'
STORE_DTYPE = torch.float16 if F16_BUILD else torch.bfloat16
# the fp16-storage build scales the LOSS (ops.LossScaler) inside FineTune / Pretrain; a unit test that back-propagates an
# unscaled loss through single kernels sees gradients of ~1e-7 in fp16's subnormal range, so such legs run in bf16 only
NO_F16_GRADS = 'fp16-storage build: this unit-level backward is not loss-scaled (gradients ~1e-7 underflow); checked in the bf16 build'

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
V = 1444

from evoke_amd.config import ARGS  # noqa: E402,F401  (the package owns the default argument dictionary; tests start from it)


def load_tokenizer():
    from evoke_amd.tokenizer import load_tokenizer as lt
    return lt(os.path.join(GOLDEN, 'iu_xray_wordlevel_uncased_tokenizer.json'))


def load_procedural(model, spec, device='cuda'):
    from oracle import spec as S
    sd = S.procedural_state(spec)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:5]
    assert all(k.endswith('position_ids') for k in res.missing_keys), res.missing_keys[:5]
    return model.to(device)


def rel_err(got, want):
    got, want = got.detach().double().cpu().reshape(-1), want.detach().double().cpu().reshape(-1)
    return float((got - want).norm() / (want.norm() + 1e-30))
