"""Whole-step HIP-graph capture (the reference's step: modules/trainer_v0401.py:426-435 -- zero_grad, forward, backward,
clip_grad_value_, optimizer.step).

A training step of this engine is ~5k kernel launches (1.7k GEMMs, the 104-conv trunk, ~2k relational-memory launches) issued
from Python through ctypes: ~50 ms of host time per 58 ms step, i.e. no slack for faster kernels.  The launch sequence of a step
depends only on the STRUCTURE of the batch -- tensor shapes and the multi-view grouping of `patient_ids` (which anchors have
which siblings) -- not on its data, so it is captured once per structure in a HIP graph (all four streams: main, weight
gradients, relational memory, text encoder; their event joins become graph edges) and replayed with one host call.

What makes the step capturable:
  * no host read-back anywhere in it: loss scale, overflow skip and optimizer step counts live on the device
    (ops.LossScaler, csrc/eltwise.hip), index uploads are memcpy nodes from dedicated pinned buffers (ops._Uploader);
  * dropout masks: seeds are frozen kernel arguments, the device-side seed epoch advanced by a captured op varies them per replay;
  * inputs are copied into the static device buffers of the captured structure before each replay.
Memory: the capture's intermediates live in a private pool (one step's peak, ~40 GB for the 384^2 / 64-image batch) for as long
as the graph exists -- sized for 288 GB of HBM.
"""
import ctypes as C
import os
from .config import tunable

import numpy as np
import torch

from . import hip as H
from . import ops

# 'replay': capture once, re-issue the recorded launches from C++ (csrc/replay.hip); 'hipgraph': hipGraphLaunch (7-12 us of host
# time per node on ROCm 7.2: slower than eager Python for the 6k-node training step, kept for comparison)
MODE = tunable('EVK_STEP_GRAPH_MODE', 'replay')
# HIP queue priority of the replay lane of a named side stream (ops.side_stream) where it should differ from its capture stream's.  None by
# default: every priority assignment tried either changed nothing or put the step into a 72-110 ms mode (profiles/r05_hw_queues.txt -- the
# runtime gives each priority class hardware queues of its own, and more than four active queues per process is where that mode begins);
# EVK_REPLAY_RM_PRIO / EVK_REPLAY_LANE_PRIO reproduce those measurements.
LANE_PRIORITY = {}
if tunable('EVK_REPLAY_RM_PRIO', '') not in ('', '0'):
    LANE_PRIORITY['rm'] = int(tunable('EVK_REPLAY_RM_PRIO', '0'))
for _kv in tunable('EVK_REPLAY_LANE_PRIO', '').split(','):          # experiments: "wgrad:1,text:-1" (HIP: -1 high, 0 default, 1 low)
    if ':' in _kv:
        LANE_PRIORITY[_kv.split(':')[0].strip()] = int(_kv.split(':')[1])


def structure_key(tensors, patient_ids, extra=()):
    """Hashable signature of everything the launch sequence depends on."""
    pid = np.asarray(patient_ids)
    _, inv = np.unique(pid, return_inverse=True)
    first = {}
    canon = tuple(first.setdefault(int(g), len(first)) for g in inv)          # grouping pattern, independent of the id strings
    return (tuple((tuple(t.shape), str(t.dtype)) if torch.is_tensor(t) else None for t in tensors), canon, tuple(extra))


class StepGraph:
    """fn() -> tensor(s), run eagerly `warmup` times, then captured and replayed.  `fn` must read its inputs from tensors that
    keep their addresses (static buffers) and must not synchronise with the host."""

    _shared_pool = {}

    def __init__(self, fn, warmup=2, make_on_replay=None):
        """make_on_replay(): called once right after a successful capture, returns the host-side hook to run after every
        replay (e.g. FusedOptimizer.replay_hook: the optimizer's host bookkeeping of the captured step)."""
        self.fn, self.warmup, self.make_on_replay, self.on_replay = fn, warmup, make_on_replay, None
        self.calls = 0
        self.graph = None
        self.plan = None
        self.info = None
        self.out = None
        self.failed = None

    def _replay(self):
        if self.plan is not None:
            H.check(H.lib.evk_replay_run(self.plan, H.stream()), 'replay_run')
            if self.info is not None and 'checked' not in self.info:          # (the first run separates lanes that share a hardware queue)
                info = (C.c_int64 * 7)()
                H.check(H.lib.evk_replay_info(self.plan, info), 'replay_info')
                self.info.update(lane_streams_replaced=int(info[6]), checked=True)
        else:
            self.graph.replay()
        if self.on_replay is not None:
            self.on_replay()
        return self.out

    def __del__(self):
        if getattr(self, 'plan', None) is not None:
            try:
                torch.cuda.synchronize()
                H.lib.evk_replay_destroy(self.plan)
            except Exception:          # noqa: BLE001 -- interpreter shutdown
                pass

    def __call__(self):
        if self.graph is not None:
            return self._replay()
        self.calls += 1
        if self.failed is not None or self.calls <= self.warmup:
            return self.fn()
        # capture on a fresh stream that inherits nothing from the surrounding one (torch.cuda.graph switches streams itself)
        torch.cuda.synchronize()
        keep0 = len(ops.CAPTURE_KEEPALIVE)
        g = torch.cuda.CUDAGraph(keep_graph=True) if MODE == 'replay' else torch.cuda.CUDAGraph()
        dev = torch.cuda.current_device()
        pool = StepGraph._shared_pool.get(dev)          # step graphs replay one at a time on one stream: one private pool for all
        if pool is None or tunable('EVK_STEP_GRAPH_SHARED_POOL', '1') == '0':
            pool = StepGraph._shared_pool[dev] = torch.cuda.graph_pool_handle()
        origin = [None]
        try:
            if MODE == 'replay':
                H.check(H.lib.evk_capture_probe(1), 'capture_probe')          # every library launch notes (graph node -> stream)
            with torch.cuda.graph(g, pool=pool, capture_error_mode='relaxed'):
                origin[0] = H.stream()                                        # the stream the capture runs on: lane 0 of the plan
                out = self.fn()
        except Exception as e:          # noqa: BLE001 -- stay on the (eager) HIP path
            import warnings
            H.lib.evk_capture_probe(0)
            self.failed = e
            del ops.CAPTURE_KEEPALIVE[keep0:]
            warnings.warn('step graph capture failed (%r); running eagerly' % (e,))
            torch.cuda.synchronize()
            return self.fn()
        self.keep = ops.CAPTURE_KEEPALIVE[keep0:]          # pinned staging buffers the graph's upload kernels read: freed with this object
        del ops.CAPTURE_KEEPALIVE[keep0:]
        if MODE == 'replay':
            if tunable('EVK_REPLAY_LANES', 'streams') == 'streams':
                for name, prio in LANE_PRIORITY.items():
                    st = ops.existing_side_stream(name)
                    if st is not None:
                        H.check(H.lib.evk_replay_lane_priority(C.c_void_p(st.cuda_stream), prio), 'replay_lane_priority')
                plan = H.lib.evk_replay_build_streams(C.c_void_p(g.raw_cuda_graph()), 16, C.c_void_p(origin[0]))
            else:                                       # the minimum path cover (rounds 2-4)
                plan = H.lib.evk_replay_build(C.c_void_p(g.raw_cuda_graph()), 16)
            H.lib.evk_capture_probe(0)
            if not plan:
                import warnings
                self.failed = RuntimeError(H.lib.evk_last_error().decode())
                self.keep = None
                warnings.warn('step replay plan failed (%s); running eagerly' % self.failed)
                return self.fn()
            self.plan = plan
            info = (C.c_int64 * 7)()
            H.check(H.lib.evk_replay_info(self.plan, info), 'replay_info')
            self.info = dict(zip(('nodes', 'kernels', 'memcpys', 'memsets', 'lanes', 'cross_lane_edges', 'lane_streams_replaced'), list(info)))
        self.graph, self.out = g, out
        if self.make_on_replay is not None:
            self.on_replay = self.make_on_replay()
        return self._replay()           # the capture pass itself executed nothing
