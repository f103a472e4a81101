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


def test_routing_of_the_resnet101_convolutions_is_decided_on_the_host():
    """The round-3 kernels (halo-tile 3x3 forward / data gradient / weight gradient, strip GEMM, stem halo kernels) are chosen by
    pure host functions of the geometry; their answers for the trunk of the headline workload (64 images of 384^2) and for 224^2
    are part of the contract: every statistics / slab buffer the tile path sizes is large enough for the kernel that takes over."""
    import ctypes as C
    from evoke_amd import hip as H
    L = H.lib
    planes, sizes384, sizes224 = (64, 128, 256, 512), (96, 48, 24, 12), (56, 28, 14, 7)
    for N, sizes in ((64, sizes384), (8, sizes224)):
        for C_, S in zip(planes, sizes):
            g = H.conv_geom(N, S, S, C_, C_, 3, 3, 1, 1)
            # Bottleneck.conv2 (stride 1): forward / flipped data gradient / weight gradient
            assert L.evk_conv3x3_halo_supported(N, S, S, C_, C_) == 1, (N, S, C_)
            assert L.evk_conv3x3_halo_routes(C.byref(g), C_, C_, L.evk_conv_stats_bytes(N * S * S, C_), 1) == 1
            assert 0 < L.evk_conv3x3_halo_part_bytes(N, S, S, C_) <= L.evk_conv_stats_bytes(N * S * S, C_)
            assert L.evk_conv3x3_wgrad_halo_supported(N, S, S, C_, C_) == 1
            assert L.evk_conv3x3_wgrad_halo_routes(C.byref(g)) == 1
            assert L.evk_conv2d_wgrad_ws_bytes(C.byref(g)) >= L.evk_conv3x3_wgrad_halo_ws_bytes(N, S, S, C_, C_) > 0
            # a stride-2 3x3 (first block of layers 2-4) stays on the tile path
            g2 = H.conv_geom(N, 2 * S, 2 * S, C_, C_, 3, 3, 2, 1)
            assert L.evk_conv3x3_halo_routes(C.byref(g2), C_, C_, 0, 0) == 0 and L.evk_conv3x3_wgrad_halo_routes(C.byref(g2)) == 0
    # strip GEMM: the contracting 1x1 convolutions where the one-workgroup-per-CU grid fills the chip (layers 2 and 3 at 384^2)
    M = lambda S: 64 * S * S
    assert L.evk_gemm_strip_routes(M(24), 256, 1024, 0, 0) == 1 and L.evk_gemm_strip_routes(M(48), 128, 512, 0, 0) == 1
    assert L.evk_gemm_strip_routes(M(96), 64, 256, 0, 0) == 0          # 64 output channels
    assert L.evk_gemm_strip_routes(M(12), 512, 2048, 0, 0) == 0        # 128 workgroups: the tile path keeps layer4
    assert L.evk_gemm_strip_part_bytes(M(24), 256) <= L.evk_conv_stats_bytes(M(24), 256)
    # stem: halo kernels for 384^2 (half width a multiple of 64), tile path for 224^2
    assert L.evk_stem_halo_supported(64, 384, 384) == 1 and L.evk_stem_halo_supported(8, 224, 224) == 0
    assert L.evk_stem_wgrad_ws_bytes(64, 384, 384) >= L.evk_stem_halo_wgrad_ws_bytes(64, 384, 384) > 0
    assert L.evk_stem_halo_part_bytes(64, 384, 384) <= L.evk_conv_stats_bytes(64 * 192 * 192, 64)
