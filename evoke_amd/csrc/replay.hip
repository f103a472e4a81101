// replay.hip -- step replayer: re-issues the launch sequence of a stream-CAPTURED training step from C++.
//
// A training step of the engine is ~5k kernel launches on four HIP streams, issued from Python at ~10 us each (50 ms of host
// time per 58 ms step).  Capturing the step in a HIP graph records exactly what has to run (our kernels, torch's glue kernels,
// autograd's accumulation adds, memcpy / memset nodes, cross-stream dependencies), but hipGraphLaunch on ROCm 7.2 spends
// 7-12 us of HOST time per node (measured: 75 ms per replay of the 6.2k-node step graph, 43 ms with the step forced onto one
// stream) -- slower than the Python loop it replaces.  So the graph is used as a RECORDING only: this file walks the captured
// hipGraph_t once (nodes, edges, kernel / memcpy / memset parameters), splits it into chains ("lanes" = HIP streams: a node
// continues the lane of a predecessor when it can, which recovers the capture's stream structure), and replays it with plain
// hipLaunchKernel / hipMemcpyAsync / hipMemsetAsync calls in topological order -- ~3.5 us per launch, cross-lane edges as
// hipEventRecord / hipStreamWaitEvent pairs.  The kernel argument blocks stay owned by the hipGraph_t, which the caller keeps
// alive for the life of the plan.  The reference has no counterpart (its step is eager PyTorch, modules/trainer_v0401.py:426-435).
#include <stdlib.h>
#include <algorithm>
#include <queue>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <string.h>
#include "common.h"

// ---- capture probe (common.h): node -> the HIP stream whose launch created it, noted by every launch of the library while armed
bool g_evk_capture_probe = false;
static std::mutex g_probe_mu;
static std::unordered_map<hipGraphNode_t, hipStream_t> g_probe_map;
static std::unordered_map<hipStream_t, int> g_lane_prio;          // capture stream -> priority of the lane that replays it (evk_replay_lane_priority)

void evk_capture_note(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t n = 0;
  if (hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &n) != hipSuccess) { (void)hipGetLastError(); return; }
  if (st != hipStreamCaptureStatusActive || n != 1 || !deps) return;          // right after a launch the stream's dependency set is that node
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe_map[deps[0]] = s;
}

namespace {

struct RNode {
  int type = 3;                 // 0 kernel, 1 memcpy (1-D), 2 memset, 3 nothing to issue
  void* func = nullptr; bool module_fn = false;
  dim3 grid, block; unsigned shmem = 0; void** args = nullptr; void** extra = nullptr;
  void* dst = nullptr; const void* src = nullptr; size_t bytes = 0; hipMemcpyKind kind = hipMemcpyDefault;
  hipMemsetParams ms{};
  int lane = 0;
  std::vector<int> waits;       // events to wait for before issuing
  int record = -1;              // event to record after issuing
};

struct Plan {
  std::vector<RNode> nodes;                 // in issue order
  std::vector<hipStream_t> lanes;           // lanes[0] is the caller's stream at run time
  std::vector<hipEvent_t> events;
  int begin_event = -1;
  std::vector<int> tail_events;             // events recorded at the end of the side lanes, joined by lane 0
  size_t n_kernels = 0, n_copies = 0, n_sets = 0, n_cross = 0;
  std::vector<int> lane_prio;               // HIP priority each side lane was created with
  std::vector<hipStream_t> spare;           // streams rejected by the hardware-queue check (kept alive: they hold their queue slot)
  bool queues_checked = false;
  int n_realiased = 0;
};

inline bool replay_debug();

// ---- do two streams share a hardware queue?
// The HIP runtime multiplexes streams onto a few hardware queues per priority (GPU_MAX_HW_QUEUES, default 4); which queue a new stream gets
// depends on how many streams the process created before.  Two LANES on one hardware queue are serialised kernel by kernel, and the replayed
// FineTune step then costs 72-80 ms instead of 47 (profiles/r05_hw_queues.txt: the step time over GPU_MAX_HW_QUEUES = 4 / 6 / 8 / 12 is
// 48.7 / 72 / 47.0 / 76 ms with nothing else changed).  The check: one single-wave kernel that spins for ~150 us of wall clock on each of the two
// streams, launched back to back from idle streams -- concurrent streams finish both in ~one spin, streams on one queue in two.
__global__ void replay_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  for (int i = 0; i < (1 << 20) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);          // bounded: ends whatever the clock does
}

double spin_ms(hipStream_t a, hipStream_t b, long long ticks) {
  (void)hipStreamSynchronize(a);
  if (b) (void)hipStreamSynchronize(b);
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return -1.0;
  }
  (void)hipEventRecord(e0, a);
  hipLaunchKernelGGL(replay_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
  (void)hipEventRecord(e1, a);
  if (b) {
    hipLaunchKernelGGL(replay_spin_kernel, dim3(1), dim3(64), 0, b, ticks);
    (void)hipEventRecord(e2, b);
  }
  (void)hipStreamSynchronize(a);
  if (b) (void)hipStreamSynchronize(b);
  float ta = 0.f, tb = 0.f;
  (void)hipEventElapsedTime(&ta, e0, e1);
  if (b) (void)hipEventElapsedTime(&tb, e0, e2);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  (void)hipGetLastError();
  return (double)(ta > tb ? ta : tb);
}

// 1: kernels on a and b overlap, 0: they are serialised (one hardware queue), -1: could not measure
int streams_concurrent(hipStream_t a, hipStream_t b) {
  if (a == b) return 0;
  if (!b) std::swap(a, b);          // (spin_ms reads a null second stream as "none": the legacy default stream goes first)
  int dev = 0, khz = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  const long long ticks = (long long)khz * 150 / 1000;          // 150 us
  // Other work on the device (another process, another stream) can only DELAY a spin kernel, never overlap two that share a queue: the pair
  // counts as concurrent as soon as one of three trials shows it, and as serialised only when all three do.
  bool measured = false;
  for (int trial = 0; trial < 3; ++trial) {
    const double one = spin_ms(a, nullptr, ticks);
    const double two = spin_ms(a, b, ticks);                     // b's kernel is queued right behind a's: end of b measured from a's start
    if (one <= 0.0 || two <= 0.0) continue;
    measured = true;
    if (two < 1.6 * one) return 1;
  }
  return measured ? 0 : -1;
}

// make every side lane of the plan concurrent with lane 0 (`s0`) and with the other side lanes: a lane that shares a hardware queue is replaced
// by a newly created stream of the same priority (the rejected one stays alive, so the runtime's assignment moves on), up to 16 attempts per lane
void separate_lanes(Plan* p, hipStream_t s0) {
  std::vector<hipStream_t> chosen{s0};
  for (size_t l = 1; l < p->lanes.size(); ++l) {
    hipStream_t cand = p->lanes[l];
    for (int tries = 0; tries < 16; ++tries) {
      bool clash = false;
      for (hipStream_t c : chosen) if (streams_concurrent(c, cand) == 0) { clash = true; break; }
      if (!clash) break;
      hipStream_t fresh = nullptr;
      if (hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, l < p->lane_prio.size() ? p->lane_prio[l] : 0) != hipSuccess) { (void)hipGetLastError(); break; }
      p->spare.push_back(cand);
      cand = fresh;
      ++p->n_realiased;
    }
    p->lanes[l] = cand;
    chosen.push_back(cand);
  }
  if (replay_debug()) fprintf(stderr, "[replay] hardware-queue check: %d lane streams replaced\n", p->n_realiased);
}

inline bool replay_debug() { static const int on = evk_tunable("EVK_REPLAY_DEBUG", 0); return on != 0; }

int new_event(Plan* p) {
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
  p->events.push_back(e);
  return (int)p->events.size() - 1;
}

}  // namespace

extern "C" {

/* Arms (1) / disarms (0) the capture probe; arming forgets what an earlier capture noted.  Arm it right before the capture begins and build
 * the plan with evk_replay_build_streams before disarming. */
int evk_capture_probe(int32_t on) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  if (on) g_probe_map.clear();
  else g_lane_prio.clear();          // (lane priorities are set between the capture and the build: they belong to that one plan)
  g_evk_capture_probe = on != 0;
  return EVK_OK;
}

void* evk_replay_build_streams(void* graph_handle, int32_t max_lanes, evk_stream_t origin);

/* The lane that replays capture stream `captured` runs at HIP queue priority `prio` (-1 above the default, 0 default) instead of that stream's
 * own priority; applies to plans built afterwards.  Measured (profiles/r05_stream_priorities.txt): with the whole step queued at once the
 * relational-memory chain -- ~1400 dependent launches of a few microseconds -- wants the higher priority (replayed step 48.9 -> 48.1 ms at
 * 384^2, 31.4 -> 30.0 ms at 224^2), while the SAME priority on the eager step's stream makes that step 60 % slower (78 ms). */
int evk_replay_lane_priority(evk_stream_t captured, int32_t prio) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_lane_prio[reinterpret_cast<hipStream_t>(captured)] = prio;
  return EVK_OK;
}

/* Builds a replay plan from a captured (not necessarily instantiated) hipGraph_t.  Returns NULL on failure (evk_last_error).
 * Lanes = a minimum path cover of the dependency graph (no knowledge of the capture's streams). */
void* evk_replay_build(void* graph_handle, int32_t max_lanes) { return evk_replay_build_streams(graph_handle, max_lanes, nullptr); }

/* The same with the capture probe's notes: `origin` = the stream the capture began on (its nodes become lane 0 = the caller's stream at
 * run time); every other capture stream becomes a lane of its own, created at the default priority like the eager step's side streams.  Nodes
 * the library did not launch itself (torch's glue kernels, memsets) continue the lane of their FIRST dependency -- a capturing stream lists
 * its own previous node first, the nodes of the events it waited for after it.  origin == NULL or no notes: the path cover. */
void* evk_replay_build_streams(void* graph_handle, int32_t max_lanes, evk_stream_t origin) {
  hipGraph_t graph = reinterpret_cast<hipGraph_t>(graph_handle);
  if (!graph || max_lanes < 1) { evk_set_error("replay_build: bad args"); return nullptr; }
  size_t n = 0, ne = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess || n == 0) { evk_set_error("replay_build: hipGraphGetNodes failed / empty graph"); return nullptr; }
  std::vector<hipGraphNode_t> nodes(n);
  if (hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) { evk_set_error("replay_build: hipGraphGetNodes failed"); return nullptr; }
  if (hipGraphGetEdges(graph, nullptr, nullptr, &ne) != hipSuccess) { evk_set_error("replay_build: hipGraphGetEdges failed"); return nullptr; }
  std::vector<hipGraphNode_t> ef(ne), et(ne);
  if (ne && hipGraphGetEdges(graph, ef.data(), et.data(), &ne) != hipSuccess) { evk_set_error("replay_build: hipGraphGetEdges failed"); return nullptr; }
  std::unordered_map<hipGraphNode_t, int> id;
  id.reserve(n * 2);
  for (size_t i = 0; i < n; ++i) id[nodes[i]] = (int)i;
  std::vector<std::vector<int>> succ(n), pred(n);
  for (size_t e = 0; e < ne; ++e) {
    auto a = id.find(ef[e]), b = id.find(et[e]);
    if (a == id.end() || b == id.end()) { evk_set_error("replay_build: edge references an unknown node"); return nullptr; }
    succ[a->second].push_back(b->second);
    pred[b->second].push_back(a->second);
  }
  // topological order, creation order as the priority (the capture's own issue order is one valid schedule)
  std::vector<int> indeg(n), order;
  order.reserve(n);
  std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
  for (size_t i = 0; i < n; ++i) { indeg[i] = (int)pred[i].size(); if (!indeg[i]) ready.push((int)i); }
  while (!ready.empty()) {
    const int v = ready.top(); ready.pop();
    order.push_back(v);
    for (int s : succ[v]) if (--indeg[s] == 0) ready.push(s);
  }
  if (order.size() != n) { evk_set_error("replay_build: the graph has a cycle"); return nullptr; }

  Plan* p = new Plan();
  p->nodes.resize(n);
  std::vector<int> pos(n);                    // node id -> index in issue order
  for (size_t k = 0; k < n; ++k) pos[order[k]] = (int)k;
  // node parameters
  size_t n_bad = 0;                           // EVK_REPLAY_DEBUG: unreplayable nodes are all listed before the build fails
  for (size_t k = 0; k < n; ++k) {
    RNode& r = p->nodes[k];
    hipGraphNode_t nd = nodes[order[k]];
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(nd, &ty) != hipSuccess) { evk_set_error("replay_build: hipGraphNodeGetType failed"); delete p; return nullptr; }
    if (ty == hipGraphNodeTypeKernel) {
      hipKernelNodeParams kp{};
      if (hipGraphKernelNodeGetParams(nd, &kp) != hipSuccess) { evk_set_error("replay_build: hipGraphKernelNodeGetParams failed"); delete p; return nullptr; }
      r.type = 0; r.func = kp.func; r.grid = kp.gridDim; r.block = kp.blockDim; r.shmem = kp.sharedMemBytes; r.args = kp.kernelParams; r.extra = kp.extra;
      hipFuncAttributes fa;
      r.module_fn = hipFuncGetAttributes(&fa, kp.func) != hipSuccess;      // not a host-side kernel symbol: a hipFunction_t
      (void)hipGetLastError();
      ++p->n_kernels;
    } else if (ty == hipGraphNodeTypeMemcpy) {
      hipMemcpy3DParms mp{};
      if (hipGraphMemcpyNodeGetParams(nd, &mp) != hipSuccess) { evk_set_error("replay_build: hipGraphMemcpyNodeGetParams failed"); delete p; return nullptr; }
      if (mp.srcArray || mp.dstArray || mp.extent.height > 1 || mp.extent.depth > 1) {
        // 2-D / 3-D copies (torch issues hipMemcpy2DAsync for some strided slice copies): the public query does not return
        // their parameters faithfully on ROCm 7.2 and cloning + pruning the graph to isolate such a node corrupts the
        // original's kernel arguments -- the plan is refused and the caller stays on eager launches.  Keep such copies out of
        // the step (the engine's own code has none; EVK_REPLAY_DEBUG=1 names the neighbours of the offending node).
        if (replay_debug()) {
          fprintf(stderr, "[replay] multi-dimensional memcpy at node %zu: width %zu height %zu depth %zu, src pitch %zu dst pitch %zu\n", k, (size_t)mp.extent.width,
                  (size_t)mp.extent.height, (size_t)mp.extent.depth, (size_t)mp.srcPtr.pitch, (size_t)mp.dstPtr.pitch);
          for (size_t j = (k > 4 ? k - 4 : 0); j < k + 3 && j < n; ++j) {
            hipGraphNodeType tj; (void)hipGraphNodeGetType(nodes[order[j]], &tj);
            const char* nm = "";
            if (tj == hipGraphNodeTypeKernel) { hipKernelNodeParams kq{}; if (hipGraphKernelNodeGetParams(nodes[order[j]], &kq) == hipSuccess) nm = hipKernelNameRefByPtr(kq.func, nullptr); }
            fprintf(stderr, "[replay] node %zu type %d %.90s\n", j, (int)tj, nm ? nm : "?");
          }
        }
        evk_set_error("replay_build: node %zu is a multi-dimensional memcpy (not replayable)", k);
        if (replay_debug()) { ++n_bad; continue; }          // list them all, fail after the walk
        delete p;
        return nullptr;
      }
      if (replay_debug()) {
        fprintf(stderr, "[replay] 1-D memcpy at node %zu: %zu bytes kind %d\n", k, (size_t)mp.extent.width, (int)mp.kind);
        for (size_t j = (k > 3 ? k - 3 : 0); j < k + 3 && j < n; ++j) {
          hipGraphNodeType tj; (void)hipGraphNodeGetType(nodes[order[j]], &tj);
          const char* nm = "";
          if (tj == hipGraphNodeTypeKernel) { hipKernelNodeParams kq{}; if (hipGraphKernelNodeGetParams(nodes[order[j]], &kq) == hipSuccess) nm = hipKernelNameRefByPtr(kq.func, nullptr); }
          fprintf(stderr, "[replay]    node %zu type %d %.90s\n", j, (int)tj, nm ? nm : "?");
        }
      }
      r.type = 1;
      r.dst = (char*)mp.dstPtr.ptr + mp.dstPos.x; r.src = (const char*)mp.srcPtr.ptr + mp.srcPos.x; r.bytes = mp.extent.width; r.kind = mp.kind;
      ++p->n_copies;
    } else if (ty == hipGraphNodeTypeMemset) {
      if (hipGraphMemsetNodeGetParams(nd, &r.ms) != hipSuccess) { evk_set_error("replay_build: hipGraphMemsetNodeGetParams failed"); delete p; return nullptr; }
      if (r.ms.height > 1 || (r.ms.elementSize != 1 && r.ms.elementSize != 2 && r.ms.elementSize != 4)) {
        evk_set_error("replay_build: node %zu is a 2-D memset (not replayable)", k);
        delete p;
        return nullptr;
      }
      r.type = 2;
      ++p->n_sets;
    } else if (ty == hipGraphNodeTypeEmpty || ty == hipGraphNodeTypeEventRecord || ty == hipGraphNodeTypeWaitEvent) {
      r.type = 3;
    } else {
      evk_set_error("replay_build: unsupported node type %d", (int)ty);
      delete p;
      return nullptr;
    }
  }
  if (n_bad) { fprintf(stderr, "[replay] %zu multi-dimensional memcpy nodes\n", n_bad); delete p; return nullptr; }
  // Lanes = a MINIMUM PATH COVER of the dependency DAG (maximum bipartite matching between "node as predecessor" and "node as
  // successor" along the edges): every chain follows real edges only, so putting a chain on one stream adds no false
  // dependency, and the cover needs no more chains than the capture had streams (main, weight gradients, relational memory,
  // text encoder, copies -- the capture's own streams are one valid cover).  Local rules do not recover them: at a fork the
  // side branch may be issued before the forking stream continues (side.wait(main); launch side; launch main) or long after
  // (autograd records an event when a node finishes and the consumer's stream waits for it when it runs); "first successor
  // continues the lane" puts weight-gradient GEMM i and data-gradient GEMM i+2 on one lane and serialises the step (75 ms
  // instead of 57), "last successor continues" breaks the main stream into 13 chains.
  std::vector<int> nxt(n, -1), prv(n, -1);       // in issue indices
  std::vector<std::vector<int>> sk(n);          // successors in issue indices, ascending
  for (size_t k = 0; k < n; ++k) { for (int sc : succ[order[k]]) sk[k].push_back(pos[sc]); std::sort(sk[k].begin(), sk[k].end()); }
  for (size_t k = 0; k < n; ++k)                // greedy start: the earliest free successor
    for (int sc : sk[k]) if (prv[sc] < 0) { nxt[k] = sc; prv[sc] = (int)k; break; }
  {
    std::vector<int> seen(n, -1);
    // Kuhn's augmenting paths, iterative (chains are thousands of nodes long)
    for (size_t root = 0; root < n; ++root) {
      if (nxt[root] >= 0 || sk[root].empty()) continue;
      std::vector<std::pair<int, size_t>> st;     // (left node, next successor slot to try)
      std::vector<int> via;                       // right node through which each stacked left node was reached
      st.push_back({(int)root, 0});
      via.push_back(-1);
      bool found = false;
      while (!st.empty() && !found) {
        auto& top = st.back();
        const int u = top.first;
        if (top.second >= sk[u].size()) { st.pop_back(); via.pop_back(); continue; }
        const int v = sk[u][top.second++];
        if (seen[v] == (int)root) continue;
        seen[v] = (int)root;
        if (prv[v] < 0) {                         // free right node: flip the path
          int right = v;
          for (int i = (int)st.size() - 1; i >= 0; --i) {
            const int left = st[i].first;
            const int old_right = nxt[left];
            nxt[left] = right; prv[right] = left;
            right = old_right;                    // the right node this left node gives up goes to the previous left node
            (void)via;
          }
          found = true;
        } else {
          st.push_back({prv[v], 0});
          via.push_back(v);
        }
      }
    }
  }
  // lanes by capture stream (probe notes), when there are any
  std::vector<int> forced(n, -1);
  std::vector<hipStream_t> lane_stream;       // lane -> the capture stream it stands for (stream lanes only)
  size_t n_streams = 0, n_noted = 0;
  if (origin) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    std::unordered_map<hipStream_t, int> lane_id;
    lane_id[reinterpret_cast<hipStream_t>(origin)] = 0;
    for (size_t k = 0; k < n && !g_probe_map.empty(); ++k) {
      auto it = g_probe_map.find(nodes[order[k]]);
      if (it != g_probe_map.end()) {
        auto ins = lane_id.emplace(it->second, (int)lane_id.size());
        forced[k] = ins.first->second;
        ++n_noted;
      } else {
        size_t nd = 0;
        int lane = 0;
        if (hipGraphNodeGetDependencies(nodes[order[k]], nullptr, &nd) == hipSuccess && nd > 0) {
          std::vector<hipGraphNode_t> deps(nd);
          if (hipGraphNodeGetDependencies(nodes[order[k]], deps.data(), &nd) == hipSuccess && nd > 0) {
            auto f = id.find(deps[0]);
            if (f != id.end() && forced[pos[f->second]] >= 0) lane = forced[pos[f->second]];
          }
        }
        forced[k] = lane;
      }
    }
    n_streams = lane_id.size();
    if (n_noted == 0 || (int)n_streams > max_lanes) { std::fill(forced.begin(), forced.end(), -1); n_streams = 0; }
    else { lane_stream.assign(n_streams, nullptr); for (auto& kv : lane_id) lane_stream[kv.second] = kv.first; }
  }
  std::vector<int> lane_tail;                 // per lane: issue index of its last node
  if (n_streams) lane_tail.assign(n_streams, -1);
  for (size_t k = 0; k < n; ++k) {
    RNode& r = p->nodes[k];
    const std::vector<int>& pr = pred[order[k]];
    int lane = forced[k];
    if (lane < 0 && prv[k] >= 0) lane = p->nodes[prv[k]].lane;
    if (lane < 0) {
      if (lane_tail.empty()) { lane = 0; lane_tail.push_back(-1); }
      else if ((int)lane_tail.size() < max_lanes) { lane = (int)lane_tail.size(); lane_tail.push_back(-1); }
      else {
        for (size_t l = 1; l < lane_tail.size(); ++l)            // out of lanes: re-use one whose chain has ended
          if (lane_tail[l] >= 0 && nxt[lane_tail[l]] < 0) { lane = (int)l; break; }
        if (lane < 0) lane = 0;
      }
    }
    r.lane = lane;
    for (int q : pr) {
      const int qi = pos[q];
      RNode& rq = p->nodes[qi];
      if (rq.lane == lane && qi <= lane_tail[lane]) continue;    // ordered by the lane itself
      if (rq.record < 0) { rq.record = new_event(p); if (rq.record < 0) { evk_set_error("replay_build: hipEventCreate failed"); delete p; return nullptr; } }
      r.waits.push_back(rq.record);
      ++p->n_cross;
    }
    lane_tail[lane] = (int)k;
  }
  if (replay_debug()) {
    fprintf(stderr, "[replay] %zu nodes, %zu noted by the capture probe, %zu capture streams%s\n", n, n_noted, n_streams, n_streams ? "" : " (path-cover lanes)");
    std::vector<int> cnt(lane_tail.size(), 0);
    for (RNode& r : p->nodes) ++cnt[r.lane];
    for (size_t l = 0; l < cnt.size(); ++l) {
      fprintf(stderr, "[replay] lane %zu: %d nodes;", l, cnt[l]);
      int shown = 0;
      for (size_t k = 0; k < n && shown < 4; ++k)
        if (p->nodes[k].lane == (int)l && p->nodes[k].type == 0) { const char* nm = hipKernelNameRefByPtr(p->nodes[k].func, nullptr); fprintf(stderr, " [%zu]%.48s", k, nm ? nm : "?"); ++shown; }
      fprintf(stderr, "\n");
    }
  }
  p->lanes.assign(lane_tail.size(), nullptr);
  // side lanes: the priority of the capture stream each stands for (the eager step's own assignment: all default), unless
  // evk_replay_lane_priority names another one for this plan
  const int side_prio = 0;
  for (size_t l = 1; l < p->lanes.size(); ++l) {
    int prio = side_prio;
    if (l < lane_stream.size() && lane_stream[l]) {
      std::lock_guard<std::mutex> lk(g_probe_mu);
      auto it = g_lane_prio.find(lane_stream[l]);
      if (it != g_lane_prio.end()) prio = it->second;
      else if (hipStreamGetPriority(lane_stream[l], &prio) != hipSuccess) { prio = side_prio; (void)hipGetLastError(); }
    }
    if (hipStreamCreateWithPriority(&p->lanes[l], hipStreamNonBlocking, prio) != hipSuccess) { evk_set_error("replay_build: hipStreamCreate failed"); delete p; return nullptr; }
    p->lane_prio.resize(p->lanes.size(), 0);
    p->lane_prio[l] = prio;
  }
  p->begin_event = new_event(p);
  for (size_t l = 1; l < p->lanes.size(); ++l) {
    const int e = new_event(p);
    if (e < 0 || p->begin_event < 0) { evk_set_error("replay_build: hipEventCreate failed"); delete p; return nullptr; }
    p->tail_events.push_back(e);
  }
  return p;
}

/* *concurrent = 1 when kernels launched on the two streams overlap, 0 when the runtime serialises them (the streams share a hardware queue);
 * both streams are synchronised and two ~150 us single-wave kernels run on each.  For callers that drive several streams at once (the serving
 * loop's searches in flight): pick streams that do not share a queue. */
int evk_streams_concurrent(evk_stream_t a, evk_stream_t b, int32_t* concurrent) {
  EVK_REQUIRE(concurrent, "streams_concurrent: null result");
  const int r = streams_concurrent(reinterpret_cast<hipStream_t>(a), reinterpret_cast<hipStream_t>(b));
  if (r < 0) { evk_set_error("streams_concurrent: could not time the streams"); return EVK_ELAUNCH; }
  *concurrent = r;
  return EVK_OK;
}

/* nodes / kernels / memcpys / memsets / lanes / cross-lane edges / isolated one-node sub-graphs of a plan */
int evk_replay_info(void* plan, int64_t* out6) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  EVK_REQUIRE(p && out6, "replay_info: bad args");
  out6[0] = (int64_t)p->nodes.size(); out6[1] = (int64_t)p->n_kernels; out6[2] = (int64_t)p->n_copies; out6[3] = (int64_t)p->n_sets;
  out6[4] = (int64_t)p->lanes.size(); out6[5] = (int64_t)p->n_cross; out6[6] = (int64_t)p->n_realiased;
  return EVK_OK;
}

/* Issues the recorded step: lane 0 = `stream` (everything is ordered after what is already queued there, and `stream` waits
 * for every lane at the end), the other lanes are streams the plan owns. */
int evk_replay_run(void* plan, evk_stream_t stream) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  EVK_REQUIRE(p, "replay_run: null plan");
  hipStream_t s0 = reinterpret_cast<hipStream_t>(stream);
  p->lanes[0] = s0;
  if (!p->queues_checked && p->lanes.size() > 1) {
    // first run of a multi-lane plan: no two lanes on one hardware queue (once; the device is drained first so that the check times idle streams)
    p->queues_checked = true;
    static const int on = evk_tunable("EVK_REPLAY_SEPARATE_LANES", 1);
    if (on) {
      (void)hipDeviceSynchronize();
      separate_lanes(p, s0);
    }
  }
  if (p->lanes.size() > 1) {
    if (hipEventRecord(p->events[p->begin_event], s0) != hipSuccess) { evk_set_error("replay_run: hipEventRecord failed"); return EVK_ELAUNCH; }
    for (size_t l = 1; l < p->lanes.size(); ++l) (void)hipStreamWaitEvent(p->lanes[l], p->events[p->begin_event], 0);
  }
  for (RNode& r : p->nodes) {
    hipStream_t s = p->lanes[r.lane];
    for (int e : r.waits) (void)hipStreamWaitEvent(s, p->events[e], 0);
    hipError_t err = hipSuccess;
    switch (r.type) {
      case 0:
        if (r.module_fn)
          err = hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(r.func), r.grid.x, r.grid.y, r.grid.z, r.block.x, r.block.y, r.block.z, r.shmem, s, r.args, r.extra);
        else
          err = hipLaunchKernel(r.func, r.grid, r.block, r.args, r.shmem, s);
        break;
      case 1: err = hipMemcpyAsync(r.dst, r.src, r.bytes, r.kind, s); break;
      case 2:
        if (r.ms.elementSize == 4) err = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(r.ms.dst), (int)r.ms.value, r.ms.width, s);
        else if (r.ms.elementSize == 2) err = hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(r.ms.dst), (unsigned short)r.ms.value, r.ms.width, s);
        else err = hipMemsetAsync(r.ms.dst, (int)r.ms.value, r.ms.width, s);
        break;
      default: break;
    }
    if (err != hipSuccess) {
      evk_set_error("replay_run: node %ld (type %d, lane %d of %zu, %zu waits) failed: %s", (long)(&r - p->nodes.data()), r.type, r.lane, p->lanes.size(),
                    r.waits.size(), hipGetErrorString(err));
      return EVK_ELAUNCH;
    }
    if (r.record >= 0) (void)hipEventRecord(p->events[r.record], s);
  }
  for (size_t l = 1; l < p->lanes.size(); ++l) {
    (void)hipEventRecord(p->events[p->tail_events[l - 1]], p->lanes[l]);
    (void)hipStreamWaitEvent(s0, p->events[p->tail_events[l - 1]], 0);
  }
  return evk_check_launch("replay_run");
}

/* `n` consecutive replays of the step from ONE call: the decode's token loop needs no host decision between steps (the beam bookkeeping
 * is a kernel of the step), so a host thread hands the whole remaining search to the launch queue here and the interpreter is not part
 * of the per-token path.  Blocks at the pace of the GPU once the launch queue is full -- call it from a thread that may wait. */
int evk_replay_run_n(void* plan, evk_stream_t stream, int32_t n) {
  for (int32_t i = 0; i < n; ++i) {
    const int rc = evk_replay_run(plan, stream);
    if (rc != EVK_OK) return rc;
  }
  return EVK_OK;
}

int evk_replay_destroy(void* plan) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p) return EVK_OK;
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  for (size_t l = 1; l < p->lanes.size(); ++l) if (p->lanes[l]) (void)hipStreamDestroy(p->lanes[l]);
  for (hipStream_t st : p->spare) (void)hipStreamDestroy(st);
  delete p;
  return EVK_OK;
}

}  // extern "C"
