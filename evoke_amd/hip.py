"""ctypes binding of libevoke_hip.so (the C ABI in include/evoke_hip.h).

The product path has NO fallback: importing this module without the built library raises, and every
entry point raises RuntimeError with evk_last_error() on a non-zero status."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# The default library stores activations / operands in IEEE fp16 (11-bit mantissa: meets the 1e-3 loss parity of the north star;
# its backward runs under the dynamic loss scale of ops.LossScaler).  EVK_STORE=bf16 selects the bf16-storage build of the same
# kernels (fp32's exponent range, no loss scaling, 3.5e-3 loss parity).  One format per process.
STORE = os.environ.get('EVK_STORE', 'f16').lower()
if STORE not in ('bf16', 'f16'):
    raise RuntimeError('EVK_STORE must be bf16 or f16, not %r' % STORE)
STORE_DTYPE = torch.float16 if STORE == 'f16' else torch.bfloat16
LIB_PATH = os.path.join(_HERE, 'libevoke_hip.so' if STORE == 'f16' else 'libevoke_hip_bf16.so')

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH, ACT_SIGMOID, ACT_GELU_NEW = 0, 1, 2, 3, 4, 5
A_PLAIN, A_CONV, A_DGRAD, A_KSTR = 0, 1, 2, 3
B_PLAIN, B_KSTR, B_WGATHER = 0, 1, 2
FAMILIES = ('gemm', 'norm', 'eltwise', 'reduce', 'optim')


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('N', 'Hi', 'Wi', 'Ci', 'Ho', 'Wo', 'Co', 'KH', 'KW', 'stride_h', 'stride_w',
                                         'pad_h', 'pad_w')] + [(n, C.c_int64) for n in ('sN', 'sH', 'sW')]


class Gemm(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('bias', C.c_void_p), ('resid', C.c_void_p),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('a_mode', C.c_int32), ('b_mode', C.c_int32),
                ('lda', C.c_int64), ('ldb', C.c_int64), ('ldc', C.c_int64), ('ldr', C.c_int64),
                ('batch_outer', C.c_int32), ('batch_inner', C.c_int32),
                ('sAo', C.c_int64), ('sAi', C.c_int64), ('sBo', C.c_int64), ('sBi', C.c_int64),
                ('sCo', C.c_int64), ('sCi', C.c_int64), ('sRo', C.c_int64), ('sRi', C.c_int64),
                ('alpha', C.c_float), ('act', C.c_int32), ('c_dtype', C.c_int32), ('r_dtype', C.c_int32),
                ('accumulate', C.c_int32), ('splitk', C.c_int32), ('b_klog', C.c_int32), ('b_tapstride', C.c_int64),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64), ('bias_stride_inner', C.c_int64), ('relu_gate', C.c_void_p),
                ('ldg', C.c_int64),
                ('colstats', C.c_void_p), ('gatestats', C.c_void_p), ('g', ConvGeom)]


class PreprocDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('src_h', C.c_int32), ('src_w', C.c_int32), ('channels', C.c_int32), ('resize_h', C.c_int32),
                ('resize_w', C.c_int32), ('crop_top', C.c_int32), ('crop_left', C.c_int32), ('out_size', C.c_int32),
                ('flip', C.c_int32), ('rotate', C.c_int32), ('affine', C.c_int32 * 6), ('mean', C.c_float * 3), ('std', C.c_float * 3)]


class TrunkCfg(C.Structure):
    _fields_ = [('blocks', C.c_int32 * 4), ('planes', C.c_int32 * 4), ('stride', C.c_int32 * 4), ('eps', C.c_float),
                ('momentum', C.c_float)]


class TrunkLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('w', 'dw', 'gamma', 'beta', 'running_mean', 'running_var', 'dgamma', 'dbeta')]


_CTYPES = {'int': C.c_int32, 'int32_t': C.c_int32, 'int64_t': C.c_int64, 'uint64_t': C.c_uint64, 'float': C.c_float,
           'evk_stream_t': C.c_void_p}


def _prototypes():
    """argtypes for every `int evk_*(...)` prototype of include/evoke_hip.h (pointers -> void*)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), 'include', 'evoke_hip.h')
    txt = re.sub(r'/\*.*?\*/', '', open(hdr).read(), flags=re.S)
    out = {}
    for m in re.finditer(r'\bint(?:64_t)?\s+(evk_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', txt):
        args = []
        for a in m.group(2).split(','):
            a = a.strip()
            if a in ('void', ''):
                continue
            if '*' in a:
                args.append(C.c_void_p)
            else:
                args.append(_CTYPES[a.replace('const ', '').split()[0]])
        out[m.group(1)] = args
    return out


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('evoke_amd: %s is missing -- build it with `python -m evoke_amd.build` '
                           '(the HIP engine has no CPU/PyTorch fallback)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.evk_last_error.restype = C.c_char_p
    for name, args in _prototypes().items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int64 if name.endswith('_bytes') else C.c_int
    lib.evk_replay_build.restype = C.c_void_p
    lib.evk_replay_build.argtypes = [C.c_void_p, C.c_int32]
    lib.evk_replay_build_streams.restype = C.c_void_p
    lib.evk_replay_build_streams.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    want = 16 if STORE == 'f16' else 0
    if lib.evk_storage_format() != want:
        raise RuntimeError('evoke_amd: %s stores format %d, EVK_STORE=%s needs %d -- rebuild' % (LIB_PATH, lib.evk_storage_format(), STORE, want))
    return lib


lib = _load()


def check(status, what=''):
    if status != 0:
        raise RuntimeError('libevoke_hip %s failed (%d): %s' % (what, status, lib.evk_last_error().decode()))


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def stream():
    """raw hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream() walks five Python frames per call --
    ~5 us, 1400 calls per training step: a tenth of the step's host time -- the two C entry points it ends in take 0.3 us.)"""
    return _raw_stream(_cur_device())


_stream_objs = {}


def current_stream():
    """torch.cuda.current_stream() for callers that need the Stream OBJECT (wait_stream / record_stream), cached by raw handle"""
    dev = _cur_device()
    raw = _raw_stream(dev)
    st = _stream_objs.get((dev, raw))
    if st is None:
        st = _stream_objs[(dev, raw)] = torch.cuda.current_stream()
    return st


_masked_streams = {}


def masked_stream(device, reserve):
    """torch stream (process lifetime, cached) whose kernels stay off `reserve` of every 32 compute units: the stream for full-GPU grids
    that must leave room for another stream's small latency-bound kernels (evk_stream_create_cu_mask).  The cleared bits are the top
    `reserve` / 8 of every 4 consecutive groups of 8 CU indices: the same share of every XCD whether the runtime enumerates the CUs
    XCD-major or round-robin over the 8 XCDs.  reserve: 0 (no mask), 8, 16 or 24."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if reserve not in (8, 16, 24):
        raise ValueError('masked_stream: reserve must be 8, 16 or 24 of every 32 CUs')
    key = (idx, reserve)
    if key not in _masked_streams:
        n = C.c_int32(0)
        check(lib.evk_device_cu_count(idx, C.byref(n)), 'device_cu_count')
        words = (n.value + 31) // 32
        mask = (C.c_uint32 * words)()
        keep_groups = 4 - reserve // 8
        for i in range(n.value):
            if (i // 8) % 4 < keep_groups:
                mask[i // 32] |= 1 << (i % 32)
        out = C.c_void_p()
        with torch.cuda.device(idx):
            check(lib.evk_stream_create_cu_mask(mask, words, C.byref(out)), 'stream_create_cu_mask')
        _masked_streams[key] = torch.cuda.ExternalStream(out.value, device=idx)
    return _masked_streams[key]


_rejected_streams = []          # streams the hardware-queue check turned down: kept alive so that the runtime's assignment moves on


def streams_concurrent(a, b):
    """True when kernels launched on the two torch streams overlap; False when the HIP runtime serialises them -- it multiplexes streams
    onto a few hardware queues per priority (GPU_MAX_HW_QUEUES, default 4), and which queue a new stream gets depends on how many streams the
    process created before (evk_streams_concurrent: two ~150 us single-wave kernels, both streams synchronised first)."""
    out = C.c_int32(0)
    check(lib.evk_streams_concurrent(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.byref(out)), 'streams_concurrent')
    return bool(out.value)


def concurrent_streams(count, priority=0, device=None, against=(), tries=16):
    """`count` new torch streams of `priority` that are pairwise concurrent with each other and with the streams in `against`: for work that
    is meant to overlap (the serving loop's searches in flight and its encoder stream).  A candidate that shares a hardware queue with an
    accepted stream is set aside (alive, see _rejected_streams) and another one is created, at most `tries` times per stream."""
    chosen = list(against)
    out = []
    for _ in range(count):
        cand = torch.cuda.Stream(device=device, priority=priority)
        for _t in range(tries):
            if all(streams_concurrent(c, cand) for c in chosen):
                break
            _rejected_streams.append(cand)
            cand = torch.cuda.Stream(device=device, priority=priority)
        chosen.append(cand)
        out.append(cand)
    return out


def ptr(t):
    return t.data_ptr() if t is not None else None


def dt(t):
    if t.dtype == STORE_DTYPE:
        return BF16          # "the library's 16-bit storage format"
    if t.dtype == torch.float32:
        return F32
    raise TypeError('unsupported dtype %s' % t.dtype)


def gemm_launch(desc):
    check(lib.evk_gemm_launch(C.byref(desc), stream()), 'evk_gemm_launch')


def conv_geom(N, Hi, Wi, Ci, Co, KH, KW, stride, pad):
    g = ConvGeom()
    g.N, g.Hi, g.Wi, g.Ci, g.Co, g.KH, g.KW = N, Hi, Wi, Ci, Co, KH, KW
    g.stride_h = g.stride_w = stride
    g.pad_h = g.pad_w = pad
    g.Ho = (Hi + 2 * pad - KH) // stride + 1
    g.Wo = (Wi + 2 * pad - KW) // stride + 1
    return g


def prof_enable(on):
    check(lib.evk_prof_enable(int(on)))


def prof_collect():
    ms = (C.c_double * len(FAMILIES))()
    n = (C.c_int64 * len(FAMILIES))()
    fl = C.c_double(0.0)
    check(lib.evk_prof_collect(ms, n, C.byref(fl)))
    return {f: (ms[i], n[i]) for i, f in enumerate(FAMILIES)}, fl.value
