// Launch plans: the train step recorded once, replayed from C.
//
// The eager step is ~370 kernel launches on three HIP streams, each one a Python -> ctypes -> VIAI_LAUNCH trip plus the
// autograd bookkeeping around it: 5.9 ms of host time for a 7.5 ms step.  A hipGraph of the same capture removes the host
// time but its executor serialises the side branches (measured 9.6 ms against 8.8 ms eager).  A plan keeps
// the launches exactly as the eager step issues them -- same kernels, same arguments, same streams, same cross-stream
// dependencies -- and only replaces WHO issues them: the step is stream-captured once (the capture is used as a recorder,
// the hipGraph is never instantiated), its nodes and edges are read back, and viai_plan_replay() walks the list with
// hipLaunchKernel / hipMemcpyAsync / hipEventRecord / hipStreamWaitEvent.
//
// Streams: every library launch made while the log is on notes (kernel, stream); the i-th node of a kernel takes the i-th noted
// stream of that kernel.  Nodes the library did not launch (PyTorch's own elementwise kernels, device-to-device copies)
// inherit the stream of a predecessor.  Any assignment is CORRECT (every graph edge becomes either stream order or an event);
// the noted streams make it the eager step's assignment.
//
// The kernel-argument arrays stay in the captured hipGraph, which the caller keeps alive together with the memory pool the
// capture allocated from (model.py holds the torch.cuda.CUDAGraph objects).
//
// Reference call site replaced: the body of `model.optimize_parameters()` in the train loop (train_whole_sync.py:76).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/viai_hip.h"
#include "viai_common.h"

int viai_plan_log_on = 0;

namespace {

struct Note {
    const void* func; hipStream_t stream; dim3 grid, block;
    std::vector<unsigned char> blob; std::vector<unsigned> sizes;
    bool used;
};
std::vector<Note> g_log;
std::mutex g_log_mu;               // launches are noted from whichever thread issues them: the forward from the caller's thread, the
                                   // backward from autograd's device thread

enum Kind { K_KERNEL, K_COPY, K_SET, K_EMPTY };

struct Node {
    Kind kind;
    hipKernelNodeParams kp;
    void* dst; const void* src; size_t bytes; hipMemcpyKind ckind;      // K_COPY
    hipMemsetParams ms;                                                  // K_SET
    int stream;                    // index into viai_plan::streams (0 = the stream handed to replay)
    int record;                    // event index recorded after the node, or -1
    std::vector<int> waits;        // event indices the node's stream waits for before it
};

}  // namespace

struct viai_plan {
    std::vector<Node> nodes;               // replay order (topological)
    std::vector<hipStream_t> streams;      // [0] unused (entry stream comes with the replay call)
    std::vector<hipEvent_t> events;
    hipEvent_t entry = nullptr;
    std::vector<int> first_on;             // per stream: order index of its first node, -1 if unused
    std::vector<hipEvent_t> exit_ev;       // per side stream
    int n_kernel = 0, n_copy = 0, n_set = 0, n_empty = 0, n_wait = 0, n_noted = 0;
};

void viai_plan_note(const void* func, void* stream, dim3 grid, dim3 block, const unsigned char* blob, const unsigned* sizes, int nargs) {
    Note n{func, (hipStream_t)stream, grid, block, {}, {}, false};
    size_t total = 0;
    for (int i = 0; i < nargs; ++i) total += sizes[i];
    n.blob.assign(blob, blob + total);
    n.sizes.assign(sizes, sizes + nargs);
    std::lock_guard<std::mutex> lk(g_log_mu);
    if (viai_plan_log_on) g_log.push_back(std::move(n));
}

extern "C" int viai_plan_log_begin(void) {
    std::lock_guard<std::mutex> lk(g_log_mu);
    if (viai_plan_log_on) return (int)hipErrorInvalidValue;       // one recorder at a time
    g_log.clear();
    viai_plan_log_on = 1;
    return 0;
}

extern "C" int viai_plan_log_end(void) {
    std::lock_guard<std::mutex> lk(g_log_mu);
    viai_plan_log_on = 0;
    return (int)g_log.size();
}

#define PLAN_TRY(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            if (getenv("VIAI_PLAN_DEBUG")) fprintf(stderr, "viai_plan: %s -> %s\n", #expr, hipGetErrorString(e_)); \
            viai_plan_destroy(pl);                                                                       \
            return (int)e_;                                                                              \
        }                                                                                                \
    } while (0)

extern "C" int viai_plan_build(void* graph_, void* capture_stream, viai_plan** out) {
    if (graph_ == nullptr || out == nullptr) return (int)hipErrorInvalidValue;
    *out = nullptr;
    hipGraph_t graph = (hipGraph_t)graph_;
    const bool dbg = getenv("VIAI_PLAN_DEBUG") != nullptr;
    viai_plan* pl = new viai_plan();
    size_t nn = 0, ne = 0;
    PLAN_TRY(hipGraphGetNodes(graph, nullptr, &nn));
    std::vector<hipGraphNode_t> gn(nn);
    if (nn) PLAN_TRY(hipGraphGetNodes(graph, gn.data(), &nn));
    PLAN_TRY(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> ef(ne), et(ne);
    if (ne) PLAN_TRY(hipGraphGetEdges(graph, ef.data(), et.data(), &ne));
    std::unordered_map<hipGraphNode_t, int> idx;
    for (size_t i = 0; i < nn; ++i) idx[gn[i]] = (int)i;
    std::vector<std::vector<int>> preds(nn), succs(nn);
    for (size_t e = 0; e < ne; ++e) {
        auto a = idx.find(ef[e]), b = idx.find(et[e]);
        if (a == idx.end() || b == idx.end()) { viai_plan_destroy(pl); return (int)hipErrorInvalidValue; }
        preds[b->second].push_back(a->second);
        succs[a->second].push_back(b->second);
    }
    // node payloads + the noted streams (creation order: the order hipGraphGetNodes returns is the capture order)
    std::vector<Node> raw(nn);
    std::vector<hipStream_t> st_of(nn, nullptr);
    std::vector<char> known(nn, 0);
    std::map<const void*, std::vector<Note*>> noted;
    for (Note& n : g_log) { n.used = false; noted[n.func].push_back(&n); }
    auto same_launch = [](const Note& t, const hipKernelNodeParams& kp) {
        if (t.grid.x != kp.gridDim.x || t.grid.y != kp.gridDim.y || t.grid.z != kp.gridDim.z || t.block.x != kp.blockDim.x ||
            t.block.y != kp.blockDim.y || t.block.z != kp.blockDim.z || kp.kernelParams == nullptr) return false;
        size_t off = 0;
        for (size_t a = 0; a < t.sizes.size(); ++a) {
            if (memcmp(kp.kernelParams[a], t.blob.data() + off, t.sizes[a]) != 0) return false;
            off += t.sizes[a];
        }
        return true;
    };
    for (size_t i = 0; i < nn; ++i) {
        Node& n = raw[i];
        n.record = -1; n.stream = -1; n.dst = nullptr; n.src = nullptr; n.bytes = 0; n.ckind = hipMemcpyDefault;
        hipGraphNodeType ty;
        PLAN_TRY(hipGraphNodeGetType(gn[i], &ty));
        if (ty == hipGraphNodeTypeKernel) {
            n.kind = K_KERNEL;
            PLAN_TRY(hipGraphKernelNodeGetParams(gn[i], &n.kp));
            auto q = noted.find(n.kp.func);
            if (q != noted.end())
                for (Note* t : q->second)
                    if (!t->used && same_launch(*t, n.kp)) { t->used = true; st_of[i] = t->stream; known[i] = 1; ++pl->n_noted; break; }
            ++pl->n_kernel;
        } else if (ty == hipGraphNodeTypeMemcpy) {
            n.kind = K_COPY;
            hipMemcpy3DParms p;
            PLAN_TRY(hipGraphMemcpyNodeGetParams(gn[i], &p));
            if (p.extent.height > 1 || p.extent.depth > 1 || p.srcArray != nullptr || p.dstArray != nullptr ||
                p.srcPos.x || p.srcPos.y || p.srcPos.z || p.dstPos.x || p.dstPos.y || p.dstPos.z) {
                if (dbg) fprintf(stderr, "viai_plan: node %zu is a strided copy: extent %zu x %zu x %zu, src %p pitch %zu (%zu x %zu) pos %zu %zu %zu, "
                                 "dst %p pitch %zu pos %zu %zu %zu, arrays %p %p, kind %d\n", i, p.extent.width, p.extent.height, p.extent.depth,
                                 p.srcPtr.ptr, p.srcPtr.pitch, p.srcPtr.xsize, p.srcPtr.ysize, p.srcPos.x, p.srcPos.y, p.srcPos.z,
                                 p.dstPtr.ptr, p.dstPtr.pitch, p.dstPos.x, p.dstPos.y, p.dstPos.z, (void*)p.srcArray, (void*)p.dstArray, (int)p.kind);
                viai_plan_destroy(pl);
                return (int)hipErrorNotSupported;
            }
            n.dst = p.dstPtr.ptr; n.src = p.srcPtr.ptr; n.bytes = p.extent.width; n.ckind = p.kind;
            ++pl->n_copy;
        } else if (ty == hipGraphNodeTypeMemset) {
            n.kind = K_SET;
            PLAN_TRY(hipGraphMemsetNodeGetParams(gn[i], &n.ms));
            if (n.ms.height > 1 || (n.ms.elementSize != 1 && n.ms.elementSize != 4)) { viai_plan_destroy(pl); return (int)hipErrorNotSupported; }
            ++pl->n_set;
        } else if (ty == hipGraphNodeTypeEmpty) {
            n.kind = K_EMPTY;
            ++pl->n_empty;
        } else {
            if (dbg) fprintf(stderr, "viai_plan: node %zu has type %d\n", i, (int)ty);
            viai_plan_destroy(pl);
            return (int)hipErrorNotSupported;
        }
    }
    // topological order, ties by creation index (the identity when the capture order is already topological)
    std::vector<int> indeg(nn), order;
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (size_t i = 0; i < nn; ++i) { indeg[i] = (int)preds[i].size(); if (!indeg[i]) ready.push((int)i); }
    while (!ready.empty()) {
        int i = ready.top(); ready.pop();
        order.push_back(i);
        for (int s : succs[i]) if (--indeg[s] == 0) ready.push(s);
    }
    if (order.size() != nn) { viai_plan_destroy(pl); return (int)hipErrorInvalidValue; }
    // streams: index 0 = the capture's origin stream (replaced by the replay argument)
    std::map<hipStream_t, int> sidx;
    pl->streams.push_back(nullptr);
    sidx[(hipStream_t)capture_stream] = 0;
    std::vector<int> pos(nn);
    for (size_t k = 0; k < nn; ++k) pos[order[k]] = (int)k;
    std::vector<int> tail;                         // per stream: last node placed on it
    tail.push_back(-1);
    for (size_t k = 0; k < nn; ++k) {
        const int i = order[k];
        int s = -1;
        if (known[i]) {
            auto f = sidx.find(st_of[i]);
            if (f == sidx.end()) { s = (int)pl->streams.size(); sidx[st_of[i]] = s; pl->streams.push_back(st_of[i]); tail.push_back(-1); }
            else s = f->second;
        } else {
            int best = -1;
            for (int p : preds[i]) {                // a predecessor that is still the end of its stream; the latest one on ties
                const int ps = raw[p].stream;
                const bool is_tail = tail[ps] == p;
                if (best < 0) best = p;
                else {
                    const bool best_tail = tail[raw[best].stream] == best;
                    if ((is_tail && !best_tail) || (is_tail == best_tail && pos[p] > pos[best])) best = p;
                }
            }
            s = best < 0 ? 0 : raw[best].stream;
        }
        raw[i].stream = s;
        tail[s] = i;
    }
    // dependencies: stream order covers same-stream edges, one event per producer covers the others
    const int S = (int)pl->streams.size();
    std::vector<std::vector<int>> seen(S, std::vector<int>(S, -1));      // [consumer stream][producer stream] = latest pos waited for
    pl->first_on.assign(S, -1);
    std::vector<int> ev_of(nn, -1);
    for (size_t k = 0; k < nn; ++k) {
        const int i = order[k];
        Node& n = raw[i];
        if (pl->first_on[n.stream] < 0) pl->first_on[n.stream] = (int)k;
        std::vector<int> ps(preds[i]);
        std::sort(ps.begin(), ps.end(), [&](int a, int b) { return pos[a] > pos[b]; });   // latest first: it subsumes earlier ones
        for (int p : ps) {
            const int sp = raw[p].stream;
            if (sp == n.stream) continue;
            if (seen[n.stream][sp] >= pos[p]) continue;
            seen[n.stream][sp] = pos[p];
            if (ev_of[p] < 0) {
                hipEvent_t ev;
                PLAN_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                ev_of[p] = (int)pl->events.size();
                pl->events.push_back(ev);
                raw[p].record = ev_of[p];
            }
            n.waits.push_back(ev_of[p]);
            ++pl->n_wait;
        }
    }
    pl->nodes.reserve(nn);
    for (size_t k = 0; k < nn; ++k) pl->nodes.push_back(raw[order[k]]);
    PLAN_TRY(hipEventCreateWithFlags(&pl->entry, hipEventDisableTiming));
    pl->exit_ev.assign(S, nullptr);
    for (int s = 1; s < S; ++s)
        if (pl->first_on[s] >= 0) PLAN_TRY(hipEventCreateWithFlags(&pl->exit_ev[s], hipEventDisableTiming));
    if (dbg)
        fprintf(stderr, "viai_plan: %zu nodes (%d kernels of which %d noted, %d copies, %d fills, %d empty), %zu edges, %d streams, "
                "%zu events, %d waits\n", nn, pl->n_kernel, pl->n_noted, pl->n_copy, pl->n_set, pl->n_empty, ne, S, pl->events.size(), pl->n_wait);
    g_log.clear();
    *out = pl;
    return 0;
}

extern "C" int viai_plan_replay(viai_plan* pl, void* stream) {
    if (pl == nullptr) return (int)hipErrorInvalidValue;
    hipStream_t s0 = (hipStream_t)stream;
    const int S = (int)pl->streams.size();
    auto st = [&](int s) { return s == 0 ? s0 : pl->streams[s]; };
    hipError_t err = hipSuccess;
    auto keep = [&](hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; };
    bool side = false;
    for (int s = 1; s < S; ++s) side |= pl->first_on[s] >= 0;
    if (side) keep(hipEventRecord(pl->entry, s0));
    const int N = (int)pl->nodes.size();
    for (int k = 0; k < N; ++k) {
        Node& n = pl->nodes[k];
        hipStream_t q = st(n.stream);
        if (n.stream != 0 && pl->first_on[n.stream] == k) keep(hipStreamWaitEvent(q, pl->entry, 0));   // behind everything queued before the plan
        for (int e : n.waits) keep(hipStreamWaitEvent(q, pl->events[e], 0));
        switch (n.kind) {
        case K_KERNEL:
            if (n.kp.kernelParams != nullptr)
                keep(hipLaunchKernel(n.kp.func, n.kp.gridDim, n.kp.blockDim, n.kp.kernelParams, n.kp.sharedMemBytes, q));
            else
                keep(hipModuleLaunchKernel((hipFunction_t)n.kp.func, n.kp.gridDim.x, n.kp.gridDim.y, n.kp.gridDim.z, n.kp.blockDim.x,
                                           n.kp.blockDim.y, n.kp.blockDim.z, n.kp.sharedMemBytes, q, nullptr, n.kp.extra));
            break;
        case K_COPY: keep(hipMemcpyAsync(n.dst, n.src, n.bytes, n.ckind, q)); break;
        case K_SET:
            if (n.ms.elementSize == 1) keep(hipMemsetAsync(n.ms.dst, (int)n.ms.value, n.ms.width, q));
            else keep(hipMemsetD32Async((hipDeviceptr_t)n.ms.dst, (int)n.ms.value, n.ms.width, q));
            break;
        case K_EMPTY: break;
        }
        if (n.record >= 0) keep(hipEventRecord(pl->events[n.record], q));
    }
    for (int s = 1; s < S; ++s)
        if (pl->first_on[s] >= 0) {
            keep(hipEventRecord(pl->exit_ev[s], pl->streams[s]));
            keep(hipStreamWaitEvent(s0, pl->exit_ev[s], 0));
        }
    return (int)err;
}

extern "C" int viai_plan_info(const viai_plan* pl, int* out, int n) {
    if (pl == nullptr || out == nullptr) return (int)hipErrorInvalidValue;
    int used = 0;
    for (size_t s = 0; s < pl->streams.size(); ++s) used += pl->first_on[s] >= 0;
    const int v[8] = {(int)pl->nodes.size(), pl->n_kernel, pl->n_noted, pl->n_copy, pl->n_set, used, (int)pl->events.size(), pl->n_wait};
    for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
    return 0;
}

extern "C" void viai_plan_destroy(viai_plan* pl) {
    if (pl == nullptr) return;
    for (hipEvent_t e : pl->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->exit_ev) if (e) (void)hipEventDestroy(e);
    if (pl->entry) (void)hipEventDestroy(pl->entry);
    delete pl;
}
