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


def test_built_kernels_avoid_the_cross_half_packed_f32_form():
    """A property of the BUILT code objects, checked on the CPU box by disassembly (tools/check_packed_opsel.py): no
    v_pk_{add,mul,fma}_f32 whose low result reads the high half of a VGPR operand (op_sel bit set).  The compiler's vectoriser produced
    that form for the gate-statistic accumulators of csrc/gemm.hip and on MI355X it returned a stale value for lanes 48-63 about once
    per 10^6 results (same launch, same inputs, a different partial sum every few launches) -- the kernels now spell those
    accumulations as scalar instructions; this keeps a compiler or source change from bringing the form back unnoticed."""
    import importlib.util
    from evoke_amd import build
    spec = importlib.util.spec_from_file_location('check_packed_opsel', os.path.join(REPO, 'tools', 'check_packed_opsel.py'))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    for lib in (build.build(), build.LIB_BF16):
        assert len(chk.code_objects(lib)) >= 8, lib
        bad = chk.suspicious(lib)
        assert not bad, (lib, bad[:5])
