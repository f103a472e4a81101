"""CPU test: the C-ABI library builds, loads, and exports every symbol include/evoke_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(REPO, 'include', 'evoke_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(evk_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    """Both builds of the sources: fp16 storage (the default) and bf16 storage (EVK_STORE=bf16)."""
    from evoke_amd import build
    path = build.build()
    names = _declared()
    assert len(names) >= 10
    for lib_path, fmt in ((path, 16), (build.LIB_BF16, 0)):
        lib = ctypes.CDLL(lib_path)
        missing = [n for n in names if not hasattr(lib, n)]
        assert not missing, (lib_path, missing)
        lib.evk_last_error.restype = ctypes.c_char_p
        assert lib.evk_version() >= 100
        assert lib.evk_storage_format() == fmt
        assert isinstance(lib.evk_last_error(), bytes)


def test_argument_validation_without_gpu():
    """EINVAL paths return before any HIP call, so they can be exercised on the CPU box."""
    from evoke_amd import hip as H
    d = H.Gemm()
    assert H.lib.evk_gemm_launch(ctypes.byref(d), None) == -1
    assert b'null operand' in H.lib.evk_last_error()
