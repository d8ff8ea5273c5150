// Launch plans: a stream-captured hipGraph replayed as PLAIN stream launches from one C loop.
//
// Why not hipGraphLaunch: on ROCm 7.2 the replay of the captured RVT-S training step costs the host 8 us per node (5.2 ms for
// the single-stream capture of ~650 kernels) and, as soon as the graph has parallel branches (the weight-gradient fork / join of
// functions.WgradSide, the per-level head streams), as much as launching eagerly from Python (14 ms, profiles/r04_a_graph_ab.txt) --
// while a single-stream graph gives up the overlap those branches exist for.  A plan keeps what capture is good at (one recording of
// every launch of the step with its arguments, ATen's included, in a private memory pool at static addresses, the fork / join
// structure as graph edges) and replaces the executor: the nodes are sorted topologically, chains of the graph become lanes (lane 0
// = the caller's stream, the others streams owned by the plan), edges that cross lanes become event record / wait pairs, and a
// replay is ~3 us per kernel: hipLaunchKernel with the node's own argument block.
//
// The graph must stay alive (the plan borrows the kernel-argument blocks of its nodes) and must come from stream capture of
// kernel / memset / 1-D memcpy work (no host nodes, child graphs or memory nodes).
#include "common.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

extern "C" int leod_comm_allreduce(void* buf, long count, int dtype, hipStream_t stream);      // k_comm.hip

enum OpType { OP_KERNEL = 0, OP_MEMSET = 1, OP_MEMCPY = 2, OP_NOP = 3 };

struct PlanOp {
    int type = OP_NOP;
    int lane = 0;
    hipKernelNodeParams kp{};
    hipMemsetParams ms{};
    void* cp_dst = nullptr;
    const void* cp_src = nullptr;
    size_t cp_bytes = 0;
    hipMemcpyKind cp_kind = hipMemcpyDeviceToDevice;
    std::vector<int> waits;      // events this op's lane waits for before the op
    int record = -1;             // event recorded on the op's lane after the op
    // input kernels (leod_register_input_kernel): a plan-owned copy of the argument pointer array whose input slot points at in_value
    std::vector<void*> own_params;
    int in_index = -1;
    // a collective recorded as leod_comm_marker_kernel (k_comm.hip): the replay issues the all-reduce on the op's lane instead of the marker
    bool coll = false; void* coll_buf = nullptr; long coll_count = 0; int coll_dtype = 0;
    void* in_captured = nullptr; // the input pointer the kernel was captured with
    void* in_value = nullptr;    // the input pointer of the next replay
};

struct Plan {
    std::vector<PlanOp> ops;
    std::vector<hipStream_t> lanes;          // [0] unused (the caller's stream), [k > 0] owned
    std::vector<hipEvent_t> events;
    std::vector<int> lane_first_wait;        // per lane > 0: 1 when the lane has ops (it then waits for the start event)
    int start_event = -1;                    // recorded on the caller's stream before anything else
    std::vector<int> tail_event;             // per lane > 0: event recorded after its last op (-1: lane unused)
    int n_kernel = 0, n_memset = 0, n_memcpy = 0, n_nop = 0, n_waits = 0, n_hoisted = 0, n_input = 0, n_coll = 0;
};

std::mutex g_mu;
std::mutex g_in_mu;
std::unordered_map<const void*, std::pair<int, int>> g_input_kernels;    // host function pointer -> (input argument index, argument count)
std::unordered_map<long, Plan*> g_plans;
long g_next = 1;
std::string g_err;

int new_event(Plan& p) {
    hipEvent_t e;
    // the events only order lanes of ONE device against each other (hipStreamWaitEvent; the host never inspects them), so the system-scope
    // fence of a default event is left out: 15.84 -> 15.70 ms per step (profiles/r04_a_graph_ab.txt).
    static const unsigned extra = (unsigned)hipEventDisableSystemFence;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | extra) != hipSuccess) return -1;
    p.events.push_back(e);
    return (int)p.events.size() - 1;
}

void free_plan(Plan* p) {
    for (auto e : p->events) hipEventDestroy(e);
    for (size_t k = 1; k < p->lanes.size(); ++k)
        if (p->lanes[k]) hipStreamDestroy(p->lanes[k]);
    delete p;
}

}  // namespace

LEOD_API const char* leod_plan_last_error() { return g_err.c_str(); }

void leod_register_input_kernel(const void* func, int arg_index, int nargs) {
    std::lock_guard<std::mutex> lk(g_in_mu);
    g_input_kernels[func] = std::make_pair(arg_index, nargs);
}

// persistent weight-pack buffers (see the header): a pack kernel's destination is its SECOND argument (conv3_pack_kernel(w, out, ..),
// lstm_pack_kernel(W, wpf, wpb, ..): one buffer, wpf first)
namespace {
std::vector<std::pair<uintptr_t, uintptr_t>> g_hoist_ranges;
bool pack_dest_persistent(const hipKernelNodeParams& kp) {
    if (!kp.kernelParams || !kp.kernelParams[1]) return false;
    const uintptr_t dst = reinterpret_cast<uintptr_t>(*reinterpret_cast<void* const*>(kp.kernelParams[1]));
    for (const auto& r : g_hoist_ranges)
        if (dst >= r.first && dst < r.first + r.second) return true;
    return false;
}
}  // namespace
LEOD_API int leod_plan_set_hoist_ranges(const long* starts, const long* bytes, int n) {
    if (n < 0 || (n > 0 && (!starts || !bytes))) return LEOD_ERR_ARG;
    g_hoist_ranges.clear();
    for (int k = 0; k < n; ++k)
        if (bytes[k] > 0) g_hoist_ranges.emplace_back((uintptr_t)starts[k], (uintptr_t)bytes[k]);
    return LEOD_OK;
}

// hip_graph: hipGraph_t of a finished stream capture.  max_lanes: streams the plan may use (1 = everything on the caller's
// stream in a topological order).  Returns a handle > 0, or LEOD_ERR_ARG / LEOD_ERR_UNSUPPORTED (leod_plan_last_error() says why).
LEOD_API long leod_plan_create(void* hip_graph, int max_lanes) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_err.clear();
    hipGraph_t graph = (hipGraph_t)hip_graph;
    if (!graph || max_lanes < 1 || max_lanes > 16) { g_err = "bad argument"; return LEOD_ERR_ARG; }
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) { g_err = "hipGraphGetNodes failed"; return LEOD_ERR_ARG; }
    if (n == 0) { g_err = "empty graph"; return LEOD_ERR_ARG; }
    std::vector<hipGraphNode_t> nodes(n);
    if (hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) { g_err = "hipGraphGetNodes failed"; return LEOD_ERR_ARG; }
    std::unordered_map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[nodes[i]] = (int)i;
    size_t ne = 0;
    if (hipGraphGetEdges(graph, nullptr, nullptr, &ne) != hipSuccess) { g_err = "hipGraphGetEdges failed"; return LEOD_ERR_ARG; }
    std::vector<hipGraphNode_t> from(ne ? ne : 1), to(ne ? ne : 1);
    if (ne && hipGraphGetEdges(graph, from.data(), to.data(), &ne) != hipSuccess) { g_err = "hipGraphGetEdges failed"; return LEOD_ERR_ARG; }
    std::vector<std::vector<int>> succ(n), pred(n);
    for (size_t e = 0; e < ne; ++e) {
        auto a = index.find(from[e]), b = index.find(to[e]);
        if (a == index.end() || b == index.end()) { g_err = "edge to a node outside the graph"; return LEOD_ERR_ARG; }
        succ[a->second].push_back(b->second);
        pred[b->second].push_back(a->second);
    }
    Plan* p = new Plan();
    std::vector<PlanOp> ops(n);
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) { g_err = "hipGraphNodeGetType failed"; delete p; return LEOD_ERR_ARG; }
        PlanOp& o = ops[i];
        if (t == hipGraphNodeTypeKernel) {
            o.type = OP_KERNEL;
            if (hipGraphKernelNodeGetParams(nodes[i], &o.kp) != hipSuccess) { g_err = "hipGraphKernelNodeGetParams failed"; delete p; return LEOD_ERR_ARG; }
            if (!o.kp.func || (!o.kp.kernelParams && !o.kp.extra)) { g_err = "kernel node without function / arguments"; delete p; return LEOD_ERR_UNSUPPORTED; }
            {
                const char* knm = hipKernelNameRefByPtr(o.kp.func, nullptr);
                if (knm && strstr(knm, "leod_comm_marker_kernel")) {
                    if (!o.kp.kernelParams) { g_err = "collective marker without an argument array"; delete p; return LEOD_ERR_UNSUPPORTED; }
                    o.coll = true;
                    o.coll_buf = *reinterpret_cast<void* const*>(o.kp.kernelParams[0]);
                    o.coll_count = *reinterpret_cast<const long*>(o.kp.kernelParams[1]);
                    o.coll_dtype = *reinterpret_cast<const int*>(o.kp.kernelParams[2]);
                    ++p->n_coll;
                }
            }
            ++p->n_kernel;
        } else if (t == hipGraphNodeTypeMemset) {
            o.type = OP_MEMSET;
            if (hipGraphMemsetNodeGetParams(nodes[i], &o.ms) != hipSuccess) { g_err = "hipGraphMemsetNodeGetParams failed"; delete p; return LEOD_ERR_ARG; }
            if (o.ms.height > 1 || (o.ms.elementSize != 1 && o.ms.elementSize != 4)) { g_err = "2-D / 2-byte memset node"; delete p; return LEOD_ERR_UNSUPPORTED; }
            ++p->n_memset;
        } else if (t == hipGraphNodeTypeMemcpy) {
            o.type = OP_MEMCPY;
            hipMemcpy3DParms m{};
            if (hipGraphMemcpyNodeGetParams(nodes[i], &m) != hipSuccess) { g_err = "hipGraphMemcpyNodeGetParams failed"; delete p; return LEOD_ERR_UNSUPPORTED; }
            if (m.extent.height > 1 || m.extent.depth > 1 || m.srcArray || m.dstArray || !m.srcPtr.ptr || !m.dstPtr.ptr || m.extent.width == 0 ||
                m.srcPos.x || m.srcPos.y || m.srcPos.z || m.dstPos.x || m.dstPos.y || m.dstPos.z) {
                g_err = "memcpy node that is not a plain 1-D copy: extent " + std::to_string(m.extent.width) + " x " + std::to_string(m.extent.height) + " x " +
                        std::to_string(m.extent.depth) + ", kind " + std::to_string((int)m.kind) + ", src " + std::to_string((size_t)m.srcPtr.ptr) + " pitch " +
                        std::to_string(m.srcPtr.pitch) + ", dst " + std::to_string((size_t)m.dstPtr.ptr) + ", arrays " + std::to_string((size_t)m.srcArray) + " " +
                        std::to_string((size_t)m.dstArray);
                // name the neighbours so that the copy can be found in the host code
                auto nm = [&](int j) -> std::string {
                    hipGraphNodeType tj; hipGraphNodeGetType(nodes[j], &tj);
                    if (tj != hipGraphNodeTypeKernel) return "node type " + std::to_string((int)tj);
                    hipKernelNodeParams kj{}; hipGraphKernelNodeGetParams(nodes[j], &kj);
                    const char* s_ = hipKernelNameRefByPtr(kj.func, nullptr);
                    return s_ ? std::string(s_).substr(0, 90) : "?";
                };
                g_err += "; after";
                for (int q : pred[i]) g_err += " [" + nm(q) + "]";
                g_err += " before";
                for (int q : succ[i]) g_err += " [" + nm(q) + "]";
                delete p; return LEOD_ERR_UNSUPPORTED;
            }
            o.cp_dst = m.dstPtr.ptr; o.cp_src = m.srcPtr.ptr; o.cp_bytes = m.extent.width; o.cp_kind = m.kind;
            ++p->n_memcpy;
        } else if (t == hipGraphNodeTypeEmpty) {
            o.type = OP_NOP;
            ++p->n_nop;
        } else {
            g_err = "graph node type " + std::to_string((int)t) + " (host / child graph / memory / event nodes are not plan material)";
            delete p;
            return LEOD_ERR_UNSUPPORTED;
        }
    }
    // ---- weight packs leave the critical chain -----------------------------------------------------------------------------------------
    // conv3_pack_kernel / lstm_pack_kernel read parameters only (constant while a step runs) and write a buffer nobody else writes; the capture
    // ordered each behind the kernel that happened to precede it on the stream (41 of them in a training step, ~5 us + a launch gap each, in the
    // middle of the neck / head chains).  Their incoming edges are dropped (predecessors are linked to their successors instead), they are chained
    // among themselves and put on lane 1 at the START of the plan, and every consumer waits for the LAST of them -- one wait on the critical lane.
    std::vector<char> hoisted(n, 0);
    static const int hoist_on = 1;
    if (hoist_on && max_lanes >= 2) {
        std::vector<int> hs;
        for (size_t i = 0; i < n; ++i) {
            if (ops[i].type != OP_KERNEL) continue;
            const char* nm = hipKernelNameRefByPtr(ops[i].kp.func, nullptr);
            if (nm && (strstr(nm, "conv3_pack_kernel") || strstr(nm, "lstm_pack_kernel")) && pack_dest_persistent(ops[i].kp)) hs.push_back((int)i);
        }
        auto erase = [](std::vector<int>& v, int x) { v.erase(std::remove(v.begin(), v.end(), x), v.end()); };
        auto add_edge = [&](int a, int b) {
            if (a == b || std::find(succ[a].begin(), succ[a].end(), b) != succ[a].end()) return;
            succ[a].push_back(b); pred[b].push_back(a);
        };
        if (hs.size() >= 2) {
            for (int v : hs) hoisted[v] = 1;
            for (int v : hs) {
                const std::vector<int> ps = pred[v], ss = succ[v];
                for (int q : ps) { erase(succ[q], v); for (int s2 : ss) add_edge(q, s2); }
                pred[v].clear();
            }
            const int last = hs.back();
            for (int v : hs) {
                if (v == last) continue;
                const std::vector<int> ss = succ[v];
                for (int s2 : ss) { erase(succ[v], s2); erase(pred[s2], v); if (!hoisted[s2]) add_edge(last, s2); }
            }
            for (size_t k = 0; k + 1 < hs.size(); ++k) add_edge(hs[k], hs[k + 1]);
            p->n_hoisted = (int)hs.size();
        }
    }
    // topological order, capture order (node index) first among the ready nodes
    std::vector<int> indeg(n), order;
    order.reserve(n);
    // (hoisted weight packs come first: keys below every node index)
    std::priority_queue<long, std::vector<long>, std::greater<long>> ready;
    auto key = [&](int v) { return hoisted[v] ? (long)v - (long)n : (long)v; };
    for (size_t i = 0; i < n; ++i) { indeg[i] = (int)pred[i].size(); if (!indeg[i]) ready.push(key((int)i)); }
    while (!ready.empty()) {
        const long kv = ready.top(); ready.pop();
        const int v = (int)(kv < 0 ? kv + (long)n : kv);
        order.push_back(v);
        for (int s : succ[v]) if (--indeg[s] == 0) ready.push(key(s));
    }
    if (order.size() != n) { g_err = "graph has a cycle"; delete p; return LEOD_ERR_ARG; }
    // height = kernels on the longest path from the node to a sink; the successor with the greatest height inherits a node's lane
    std::vector<double> height(n, 0.0);
    for (int k = (int)n - 1; k >= 0; --k) {
        int v = order[k];
        double h = 0.0;
        for (int s : succ[v]) h = std::max(h, height[s]);
        height[v] = h + (ops[v].type == OP_NOP ? 0.01 : 1.0);
    }
    std::vector<int> heir(n, -1);
    for (size_t v = 0; v < n; ++v) {
        double best = -1.0;
        for (int s : succ[v]) if (height[s] > best || (height[s] == best && s < heir[v])) { best = height[s]; heir[v] = s; }
    }
    std::vector<int> lane(n, -1);
    std::vector<int> lane_tail;              // last node placed on each lane
    std::vector<double> lane_load;
    int root = order[0];
    for (int v : order) if (hoisted[root] || (pred[v].empty() && !hoisted[v] && height[v] > height[root])) root = v;
    for (int v : order) {
        int l = -1;
        if (max_lanes == 1) l = 0;
        else if (hoisted[v]) l = 1;
        else if (v == root) l = 0;
        else {
            double best = -1.0;
            for (int q : pred[v]) if (!hoisted[q] && heir[q] == v && lane_tail[lane[q]] == q && height[q] > best) { best = height[q]; l = lane[q]; }
            // not the heir of any predecessor, but one of them is still the tail of its lane (its heir went on with another lane):
            // go on there rather than open a lane
            if (l < 0)
                for (int q : pred[v]) if (!hoisted[q] && lane_tail[lane[q]] == q && height[q] > best) { best = height[q]; l = lane[q]; }
        }
        if (l < 0) {
            // a branch starts here: a lane of its own while there are lanes left (lane 0 is kept for the root's chain), else the
            // lane with the least work queued
            if ((int)lane_tail.size() < max_lanes && !(lane_tail.empty())) { l = (int)lane_tail.size(); }
            else if (lane_tail.empty()) { l = 0; }
            else {
                l = max_lanes > 1 ? 1 : 0;
                for (int k = 1; k < (int)lane_tail.size(); ++k) if (lane_load[k] < lane_load[l]) l = k;
            }
        }
        while ((int)lane_tail.size() <= l) { lane_tail.push_back(-1); lane_load.push_back(0.0); }
        lane[v] = l;
        lane_tail[l] = v;
        lane_load[l] += 1.0;
    }
    int nl = (int)lane_tail.size();
    p->lanes.assign(nl, nullptr);
    // Side lanes are created WITHOUT a priority: any non-default HIP stream priority (low or high) doubled the step on MI355X / ROCm 7.2
    // (16.5 -> 30.8 ms, profiles/r04_a_graph_ab.txt).
    for (int k = 1; k < nl; ++k)
        if (hipStreamCreateWithFlags(&p->lanes[k], hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; free_plan(p); return LEOD_ERR_LAUNCH; }
    // cross-lane edges -> events.  pos[v] = position in launch order; an edge q -> v inside one lane is ordered by the stream itself
    // (q is launched before v: topological order).  A wait is dropped when an earlier op of v's lane already waited for an event
    // recorded at or after q on q's lane (events of a lane are ordered).
    std::vector<int> pos(n);
    for (size_t k = 0; k < n; ++k) pos[order[k]] = (int)k;
    std::vector<int> rec_event(n, -1);
    std::vector<std::vector<int>> seen(nl, std::vector<int>(nl, -1));   // seen[a][b] = latest launch position on lane b that lane a has waited for
    for (int v : order) {
        PlanOp& o = ops[v];
        o.lane = lane[v];
        for (int q : pred[v]) {
            if (lane[q] == lane[v]) continue;
            if (seen[lane[v]][lane[q]] >= pos[q]) continue;
            if (rec_event[q] < 0) {
                rec_event[q] = new_event(*p);
                if (rec_event[q] < 0) { g_err = "hipEventCreate failed"; free_plan(p); return LEOD_ERR_LAUNCH; }
                ops[q].record = rec_event[q];
            }
            o.waits.push_back(rec_event[q]);
            seen[lane[v]][lane[q]] = pos[q];
            ++p->n_waits;
        }
    }
    p->start_event = new_event(*p);
    p->tail_event.assign(nl, -1);
    for (int k = 1; k < nl; ++k) p->tail_event[k] = new_event(*p);
    p->ops.reserve(n);
    for (int v : order) p->ops.push_back(std::move(ops[v]));
    // kernels that read the step's input: the plan launches them with its own copy of the argument pointers, the input slot pointing at a
    // value the caller may re-point per replay (leod_plan_rebase_input)
    {
        std::lock_guard<std::mutex> lk2(g_in_mu);
        for (PlanOp& o : p->ops) {
            if (o.type != OP_KERNEL || !o.kp.kernelParams) continue;
            auto it = g_input_kernels.find(o.kp.func);
            if (it == g_input_kernels.end()) continue;
            const int idx = it->second.first, na = it->second.second;
            o.own_params.assign(o.kp.kernelParams, o.kp.kernelParams + na);
            o.in_index = idx;
            o.in_captured = o.in_value = *reinterpret_cast<void* const*>(o.kp.kernelParams[idx]);
        }
        for (PlanOp& o : p->ops)
            if (o.in_index >= 0) { o.own_params[o.in_index] = &o.in_value; o.kp.kernelParams = o.own_params.data(); ++p->n_input; }
    }
    long h = g_next++;
    g_plans[h] = p;
    return h;
}

// Replay on `stream` (lane 0); the other lanes start after everything enqueued on `stream` so far and are joined back into it.
static int plan_launch(long handle, hipStream_t stream, bool join) {
    Plan* p;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_plans.find(handle);
        if (it == g_plans.end()) return LEOD_ERR_ARG;
        p = it->second;
    }
    const int nl = (int)p->lanes.size();
    if (nl > 1) {
        if (hipEventRecord(p->events[p->start_event], stream) != hipSuccess) return LEOD_ERR_LAUNCH;
        for (int k = 1; k < nl; ++k)
            if (hipStreamWaitEvent(p->lanes[k], p->events[p->start_event], 0) != hipSuccess) return LEOD_ERR_LAUNCH;
    }
    for (PlanOp& o : p->ops) {
        hipStream_t s = o.lane ? p->lanes[o.lane] : stream;
        for (int e : o.waits)
            if (hipStreamWaitEvent(s, p->events[e], 0) != hipSuccess) return LEOD_ERR_LAUNCH;
        hipError_t rc = hipSuccess;
        switch (o.type) {
            case OP_KERNEL:
                if (o.coll) { if (leod_comm_allreduce(o.coll_buf, o.coll_count, o.coll_dtype, s) != LEOD_OK) return LEOD_ERR_LAUNCH; }
                else if (o.kp.kernelParams) rc = hipLaunchKernel(o.kp.func, o.kp.gridDim, o.kp.blockDim, o.kp.kernelParams, o.kp.sharedMemBytes, s);
                else rc = hipModuleLaunchKernel((hipFunction_t)o.kp.func, o.kp.gridDim.x, o.kp.gridDim.y, o.kp.gridDim.z, o.kp.blockDim.x,
                                                o.kp.blockDim.y, o.kp.blockDim.z, o.kp.sharedMemBytes, s, nullptr, o.kp.extra);
                break;
            case OP_MEMSET:
                if (o.ms.elementSize == 4) rc = hipMemsetD32Async((hipDeviceptr_t)o.ms.dst, (int)o.ms.value, o.ms.width, s);
                else rc = hipMemsetAsync(o.ms.dst, (int)o.ms.value, o.ms.width, s);
                break;
            case OP_MEMCPY:
                rc = hipMemcpyAsync(o.cp_dst, o.cp_src, o.cp_bytes, o.cp_kind, s);
                break;
            default: break;
        }
        if (rc != hipSuccess) return LEOD_ERR_LAUNCH;
        if (o.record >= 0 && hipEventRecord(p->events[o.record], s) != hipSuccess) return LEOD_ERR_LAUNCH;
    }
    for (int k = 1; k < nl; ++k) {
        if (hipEventRecord(p->events[p->tail_event[k]], p->lanes[k]) != hipSuccess) return LEOD_ERR_LAUNCH;
        if (join && hipStreamWaitEvent(stream, p->events[p->tail_event[k]], 0) != hipSuccess) return LEOD_ERR_LAUNCH;
    }
    return LEOD_OK;
}
LEOD_API int leod_plan_launch(long handle, hipStream_t stream) { return plan_launch(handle, stream, true); }
// The same without the closing join: `stream` does not wait for the plan's side lanes (their work -- weight gradients -- feeds nothing the
// caller enqueues next).  leod_plan_join(plan, stream) makes `stream` wait for them later; it must be called before the plan is launched
// again and before anything reads what the side lanes wrote.
LEOD_API int leod_plan_launch_nojoin(long handle, hipStream_t stream) { return plan_launch(handle, stream, false); }
// The step's input without a copy: every kernel of the plan that reads the input tensor (leod_register_input_kernel: the stem convolution and
// its weight gradient) and was captured with a pointer inside [captured_base, captured_base + bytes) reads new_base + the same offset from
// the next launch on.  Returns the number of kernels re-pointed (0: the plan holds none -- the caller copies instead).  The caller keeps
// the new buffer alive and unchanged until the launches that read it have completed (the backward plan reads it last).
LEOD_API int leod_plan_rebase_input(long handle, const void* captured_base, long bytes, const void* new_base) {
    Plan* p;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_plans.find(handle);
        if (it == g_plans.end()) return LEOD_ERR_ARG;
        p = it->second;
    }
    if (!captured_base || !new_base || bytes <= 0) return LEOD_ERR_ARG;
    int n = 0;
    const char* lo = static_cast<const char*>(captured_base);
    for (PlanOp& o : p->ops) {
        if (o.in_index < 0) continue;
        const char* c = static_cast<const char*>(o.in_captured);
        if (c < lo || c >= lo + bytes) continue;
        o.in_value = const_cast<char*>(static_cast<const char*>(new_base) + (c - lo));
        ++n;
    }
    return n;
}

LEOD_API int leod_plan_join(long handle, hipStream_t stream) {
    Plan* p;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_plans.find(handle);
        if (it == g_plans.end()) return LEOD_ERR_ARG;
        p = it->second;
    }
    for (int k = 1; k < (int)p->lanes.size(); ++k)
        if (hipStreamWaitEvent(stream, p->events[p->tail_event[k]], 0) != hipSuccess) return LEOD_ERR_LAUNCH;
    return LEOD_OK;
}

// info[8] = kernels, memsets, memcpys, empty nodes, lanes, events, cross-lane waits, ops
LEOD_API int leod_plan_info(long handle, int* info) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(handle);
    if (it == g_plans.end() || !info) return LEOD_ERR_ARG;
    Plan* p = it->second;
    info[0] = p->n_kernel; info[1] = p->n_memset; info[2] = p->n_memcpy; info[3] = p->n_nop;
    info[4] = (int)p->lanes.size(); info[5] = (int)p->events.size(); info[6] = p->n_waits; info[7] = (int)p->ops.size(); info[8] = p->n_coll;
    return LEOD_OK;
}

// Debug listing of a plan: one line per op in launch order -- lane, type, kernel name, events waited for, event recorded.
LEOD_API int leod_plan_dump(long handle, const char* path) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(handle);
    if (it == g_plans.end() || !path) return LEOD_ERR_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return LEOD_ERR_ARG;
    int k = 0;
    for (const PlanOp& o : it->second->ops) {
        const char* nm = o.type == OP_KERNEL ? hipKernelNameRefByPtr(o.kp.func, nullptr) : (o.type == OP_MEMSET ? "memset" : (o.type == OP_MEMCPY ? "memcpy" : "nop"));
        fprintf(f, "%5d lane %d  %-.100s  waits", k++, o.lane, nm ? nm : "?");
        for (int e : o.waits) fprintf(f, " %d", e);
        fprintf(f, "  record %d\n", o.record);
    }
    fclose(f);
    return LEOD_OK;
}

LEOD_API int leod_plan_destroy(long handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(handle);
    if (it == g_plans.end()) return LEOD_ERR_ARG;
    free_plan(it->second);
    g_plans.erase(it);
    return LEOD_OK;
}
