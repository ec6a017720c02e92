// program.hip -- command-list executor: one C-ABI call enqueues a whole recorded sequence of library calls (a network's
// forward or backward) on a main stream and the side stream(s).
//
// Why: at batch 8 the DCDP step is ~115 launches of 5-50 us.  Driven call by call from the Python host the enqueue path
// (~18 us per launch: tensor allocation, ctypes marshalling, event objects for the weight-gradient side stream) was
// LONGER than the GPU work, i.e. the step was host-bound.  A hipGraph removes the host cost but replays the two-stream
// backward serially; this executor keeps the main / side stream concurrency and costs one hipLaunchKernel per command.
//
// The host side (coivo_amd/program.py) records the commands once per (network, shape) with persistent activation
// buffers, patches the few per-call pointers (input images, output, incoming gradients) and replays.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"
#include "tuning.h"

using namespace colvo;

namespace {
// FORK / JOIN events: a ring of library-owned host objects.  Forward passes run on the caller's thread and backward
// passes on autograd's, so creation is once-only (std::call_once) and the cursor is atomic.  An event is re-recorded
// NEV uses later; by then the wait it fed has long been enqueued (hipStreamWaitEvent captures the record it sees).
constexpr unsigned NEV = 256;
hipEvent_t g_ev[NEV];
std::once_flag g_ev_once;
std::atomic<unsigned> g_ev_next{0};

hipEvent_t next_event() {
    std::call_once(g_ev_once, [] {
        // (hipEventDisableSystemFence on top -- an agent-scope release at the end of a kernel that carries the event -- measured no
        // difference: 1.335 ms per step either way)
        for (unsigned i = 0; i < NEV; ++i) (void)hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming);
    });
    return g_ev[g_ev_next.fetch_add(1, std::memory_order_relaxed) % NEV];
}

// Further side streams (one by default), owned by the library.  A weight-gradient kernel at its atomics-bound grid size puts ONE 4-wave
// workgroup on a CU; with a single side stream they run one after the other, and the backward pass ends ~90 us after the
// input-gradient chain with only such kernels left (profiles/r2_bench_kernel_stats.csv, kernel trace).  Consecutive FORKs
// alternate between the caller's side stream and this one, so two weight-gradient kernels overlap each other as well as
// the main stream.  Every call ends with the caller's side stream waiting for this one: joining the caller's stream -- all
// the host ever does -- covers both.  COLVO_SIDE_STREAMS=1 turns it off.
constexpr int MAX_AUX = 3;
hipStream_t g_aux[MAX_AUX] = {nullptr, nullptr, nullptr};
int g_naux = 0;
std::atomic<int> g_aux_limit{MAX_AUX};      // colvo_set_aux_side_streams(): how many of them colvo_run_commands may use

// number of library-owned side streams in use: min(COLVO_SIDE_STREAMS - 1 (default 1), colvo_set_aux_side_streams()); they are
// created on first use, so a process that limits them to 0 before its first backward pass never creates one (an extra stream,
// even idle, changes how the runtime deals its streams onto hardware queues).  Measured (ms per step): 1 side stream in all
// 1.582, 2 (default) 1.545, 3 -> 5.2, 4 -> 4.3: with four or more hardware queues active and cross-queue dependencies between
// them the whole backward pass serialises -- do not raise it.
std::mutex g_aux_mu;
int aux_streams() {
    const int want = std::max(0, std::min(MAX_AUX, (int)TUNE(side_streams) - 1));
    const int n = std::min(want, g_aux_limit.load(std::memory_order_relaxed));
    if (n <= g_naux) return n;
    std::lock_guard<std::mutex> lock(g_aux_mu);
    while (g_naux < n) {
        if (hipStreamCreateWithFlags(&g_aux[g_naux], hipStreamNonBlocking) != hipSuccess) break;
        ++g_naux;
    }
    return std::min(n, g_naux);
}

}  // namespace
namespace colvo { thread_local LaunchTap g_launch_tap; }
namespace {
int order_after(hipStream_t later, hipStream_t earlier, const char* what) {
    hipEvent_t e = next_event();
    hipError_t he = hipEventRecord(e, earlier);
    if (he == hipSuccess) he = hipStreamWaitEvent(later, e, 0);
    if (he != hipSuccess) set_error("colvo_run_commands: %s failed: %s", what, hipGetErrorString(he));
    return (int)he;
}
}  // namespace

extern "C" int colvo_set_aux_side_streams(int n) {
    COLVO_CHECK_ARG(n >= 0, "colvo_set_aux_side_streams: n must be >= 0");
    g_aux_limit.store(n > MAX_AUX ? MAX_AUX : n, std::memory_order_relaxed);
    return 0;
}

// one recorded command on stream `s` (FORK / JOIN are handled by the callers)
static int run_one(const ColvoCmd& c, int k, colvo_stream_t s) {
    switch (c.op) {
        case COLVO_CMD_CONV_FWD:
            return colvo_conv_fwd(&c.desc, c.p[0], c.p[1], c.p[2], (const float*)c.p[3], (void*)c.p[4], s);
        case COLVO_CMD_CONV_DGRAD:
            return colvo_conv_dgrad(&c.desc, c.i[0], c.p[0], c.p[1], c.p[2], (void*)c.p[3], c.i[1], s);
        case COLVO_CMD_CONV_DGRAD_BOTH:
            return colvo_conv_dgrad_both(&c.desc, c.p[0], c.p[1], c.p[2], c.p[3], (void*)c.p[4], (void*)c.p[5], s);
        case COLVO_CMD_CONV_WGRAD:
            if (c.p[5] && c.i[1]) return colvo_conv_wgrad_slabs(&c.desc, c.p[0], c.p[1], c.p[2], (void*)c.p[5], (size_t)(uint32_t)c.i[0], s);
            if (c.p[5]) return colvo_conv_wgrad_det(&c.desc, c.p[0], c.p[1], c.p[2], (float*)c.p[3], (float*)c.p[4], (void*)c.p[5],
                                                    (size_t)(uint32_t)c.i[0], s);
            if (c.i[2]) return colvo_conv_wgrad_clean(&c.desc, c.p[0], c.p[1], c.p[2], (float*)c.p[3], (float*)c.p[4], 1, s);
            return colvo_conv_wgrad(&c.desc, c.p[0], c.p[1], c.p[2], (float*)c.p[3], (float*)c.p[4], s);
        case COLVO_CMD_PACK_NCHW: {
            const float* src[4] = {(const float*)c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3]};
            return colvo_pack_nchw(c.i[0], src, &c.i[1], c.i[5], c.i[6], c.i[7], c.i[8], c.i[9], (void*)c.p[4], s);
        }
        case COLVO_CMD_UNPACK_NHWC_GRAD:
            return colvo_unpack_nhwc_grad(c.i[0], c.p[0], c.i[1], c.i[2], c.i[3], c.i[4], c.i[5], c.i[6], (float*)c.p[1], c.i[7], s);
        case COLVO_CMD_DEPTH_HEAD_FWD:
            return colvo_depth_head_fwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], c.i[1], c.i[2], c.i[3], c.i[4],
                                        c.f[0], c.f[1], (float*)c.p[3], s);
        case COLVO_CMD_DEPTH_HEAD_BWD:
            return colvo_depth_head_bwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3], c.i[1],
                                        c.i[2], c.i[3], c.i[4], c.f[0], c.f[1], (float*)c.p[4], (void*)c.p[5], (float*)c.p[6],
                                        (float*)c.p[7], s);
        case COLVO_CMD_DEPTH_HEAD_BWD_PARTS:
            return colvo_depth_head_bwd_parts(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3],
                                              (const float*)c.p[4], (const float*)c.p[5], (const float*)c.p[10],
                                              (const float*)c.p[6], (const float*)c.p[7],
                                              c.i[1], c.i[2], c.i[3], c.i[4], c.f[0], c.f[1], (float*)c.p[8], (void*)c.p[9], nullptr,
                                              nullptr, s);
        case COLVO_CMD_DEPTH_HEAD_WGRAD:
            if (c.p[4]) return colvo_depth_head_wgrad_det(c.i[0], c.p[0], (const float*)c.p[1], c.i[1], c.i[2], c.i[3], c.i[4],
                                                          (float*)c.p[2], (float*)c.p[3], (void*)c.p[4], (size_t)(uint32_t)c.i[5], s);
            return colvo_depth_head_wgrad(c.i[0], c.p[0], (const float*)c.p[1], c.i[1], c.i[2], c.i[3], c.i[4], (float*)c.p[2],
                                          (float*)c.p[3], s);
        case COLVO_CMD_POSE_HEAD_FWD:
            return colvo_pose_head_fwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], c.i[1], c.i[2], c.i[3], c.f[0],
                                       c.f[1], (float*)c.p[3], s);
        case COLVO_CMD_POSE_HEAD_BWD:
            if (c.i[4]) return colvo_pose_head_bwd_det(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3],
                                                       (const float*)c.p[4], (const float*)c.p[8], (const float*)c.p[9], c.i[1], c.i[2],
                                                       c.i[3], c.f[0], c.f[1], (void*)c.p[5], (float*)c.p[6], (float*)c.p[7], s);
            return colvo_pose_head_bwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3],
                                       (const float*)c.p[4], (const float*)c.p[8], (const float*)c.p[9], c.i[1], c.i[2], c.i[3],
                                       c.f[0], c.f[1], (void*)c.p[5], (float*)c.p[6], (float*)c.p[7], s);
        case COLVO_CMD_CONV_DGRAD_PLANES:
            return colvo_conv_dgrad_planes(&c.desc, c.p[0], (const float*)c.p[1], c.i[0], c.i[1], (float*)c.p[2], c.i[2], s);
        case COLVO_CMD_CONV_BWD_FUSED:
            return colvo_conv_bwd_fused(&c.desc, c.p[0], c.p[1], c.p[2], c.i[0], (void*)c.p[3], (float*)c.p[4], (float*)c.p[5],
                                        (const float*)c.p[6], (const float*)c.p[7], (float*)c.p[8], s);
        case COLVO_CMD_CONV_HEAD_FUSED:
            return colvo_conv_head_fused(&c.desc, c.p[0], c.p[1], (const float*)c.p[2], (const float*)c.p[3], (const float*)c.p[4], c.f[0],
                                         c.f[1], (void*)c.p[5], (float*)c.p[6], (void*)c.p[7], s);
        case COLVO_CMD_PACK_STEM_POSE:
            return colvo_pack_stem_pose((const float*)c.p[0], c.i[0], c.i[1], c.i[2], (void*)c.p[1], (void*)c.p[2], s);
        case COLVO_CMD_HEAD_WGRAD_MFMA:
            return colvo_depth_head_wgrad_mfma(c.p[0], (const float*)c.p[1], c.i[0], c.i[1], c.i[2], (float*)c.p[2], s);
        case COLVO_CMD_HEAD_WGRAD_REDUCE:
            return colvo_depth_head_wgrad_reduce((const float*)c.p[0], c.i[0], (float*)c.p[1], (float*)c.p[2], s);
        case COLVO_CMD_WGRAD_REDUCE_GROUP:
            return colvo_wgrad_reduce_group((const ColvoWgradSlabs*)c.p[0], c.i[0], s);
        case COLVO_CMD_SIDE_SYNC:
            return 0;       // (the eager executor handles it before it gets here; in a captured side chain the order is the list's)
        default:
            set_error("colvo_run_commands: unknown op %d in command %d", c.op, k);
            return (int)hipErrorInvalidValue;
    }
}

// ---- hipGraph form ------------------------------------------------------------------------------------------------------------ //
// While `main_stream` is being captured (hipStreamBeginCapture -- torch.cuda.graph) the command list is turned into graph nodes with
// EXPLICIT dependencies on that ONE capturing stream: no side stream, no events.  hipStreamGetCaptureInfo_v2 reads the dependency
// set the next captured node would get, hipStreamUpdateCaptureDependencies replaces it.  The main-stream commands form one chain;
// the side-stream commands (weight gradients) form a second chain whose segments hang off the main chain:
//   policy 0 (serial):  one branch -- side commands are captured in list order on the main chain;
//   policy 1 (fine):    every FORK is an edge main -> side (the eager schedule, node for node);
//   policy 2 (grouped): side commands are collected and flushed `group` at a time; a flushed segment depends on the main chain
//                       AS FAR AS IT HAS BEEN CAPTURED at the flush and on the previous side segment -- few cross-branch edges,
//                       the weight gradients of one segment run beside the input-gradient chain that follows it;
//   policy 3:           policy 2 with the segments alternating between TWO side chains (the eager schedule's auxiliary stream).
// (ROCm 7.2 replays a graph's branches on streams of its own; every cross-branch edge costs a marker with a signal -- the
// two-stream capture of round 2, one edge per layer, replayed 2.4 x slower than eager.)  Each call ends joined (the next captured
// node depends on both chains) unless carry mode is on (colvo_set_capture_carry: the side chain stays open from call to call, as
// the eager schedule's deferred join leaves the side stream running, until colvo_capture_join).  Commands are re-ordered only within what the eager schedule already allows: a side command never
// runs before the main-chain node it was forked from, nothing but the join depends on it, and every buffer it reads is private to
// its recorded pass (coivo_amd/program.py keeps them alive for the life of the program).
static std::atomic<int> g_capture_policy{2};
static std::atomic<int> g_capture_group{2};
static std::atomic<int> g_capture_carry{0};

struct CaptureTail {
    std::vector<hipGraphNode_t> nodes;
    int get(hipStream_t s, unsigned long long* id = nullptr, hipGraph_t* graph = nullptr) {
        hipStreamCaptureStatus st;
        const hipGraphNode_t* deps = nullptr;
        size_t n = 0;
        hipError_t e = hipStreamGetCaptureInfo_v2(s, &st, id, graph, &deps, &n);
        if (e != hipSuccess || st != hipStreamCaptureStatusActive) {
            set_error("colvo_run_commands: hipStreamGetCaptureInfo_v2 failed under capture: %s", hipGetErrorString(e));
            return e != hipSuccess ? (int)e : (int)hipErrorIllegalState;
        }
        nodes.assign(deps, deps + n);
        return 0;
    }
    int set(hipStream_t s) {
        hipError_t e = hipStreamUpdateCaptureDependencies(s, nodes.data(), nodes.size(), hipStreamSetCaptureDependencies);
        if (e != hipSuccess) set_error("colvo_run_commands: hipStreamUpdateCaptureDependencies failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    void merge(const CaptureTail& o) {
        for (hipGraphNode_t n : o.nodes)
            if (std::find(nodes.begin(), nodes.end(), n) == nodes.end()) nodes.push_back(n);
    }
};

// What one capture keeps between colvo_run_commands calls (carry mode, colvo_set_capture_carry): the open side chain(s), the side
// commands not yet turned into nodes -- COPIES: the caller may patch or free its list between calls -- and the fork point of
// policy 1.  Keyed by the capture's id: a new capture starts empty whatever an aborted one left behind.  One capture at a time per
// process (the step is captured by one thread); the mutex only keeps a stray concurrent call from corrupting the vectors.
struct CaptureState {
    unsigned long long id = ~0ull;          // (the runtime numbers its captures from 0: no capture has this id)
    CaptureTail sides[2], fork_at;
    int seg = 0;
    std::vector<ColvoCmd> pending;
    // statistics of what was built since colvo_graph_stats_reset (colvo_graph_stats)
    long long calls = 0, main_nodes = 0, side_nodes = 0, segments = 0, joins = 0, carried = 0, max_entry_deps = 0;
};
static CaptureState g_cap;
static std::mutex g_cap_mu;

// Capture the pending side commands as one segment of the side chain: it depends on `at` (a point of the main chain) and on the
// previous segment.  ORDER MATTERS: ROCm's graph executor keeps a node's FIRST child on the node's stream and opens a new
// stream for every further child, so a segment is captured AFTER the main command that follows its fork point -- the main
// chain then stays on one stream and the side chain (each segment the first child of the previous one) on a second.  Captured
// the other way round every fork moved the main chain to a fresh stream: four or more hardware queues, the serialised
// mode of DESIGN.md section 3.4 (profiles/r3_graph_replay.md).
static int capture_flush(CaptureState& st, const CaptureTail& at, int nchains, hipStream_t ms) {
    if (st.pending.empty()) return 0;
    CaptureTail cur;
    if (int rc = cur.get(ms)) return rc;                    // the main chain as far as it has been captured
    CaptureTail& side = st.sides[st.seg++ % nchains];
    CaptureTail deps = at;
    deps.merge(side);
    if (int rc = deps.set(ms)) return rc;
    for (size_t k = 0; k < st.pending.size(); ++k)
        if (int rc = run_one(st.pending[k], (int)k, (colvo_stream_t)ms)) return rc;
    st.side_nodes += (long long)st.pending.size();
    st.segments += 1;
    st.pending.clear();
    if (int rc = side.get(ms)) return rc;
    return cur.set(ms);                                     // back on the main chain
}

// the next captured node depends on the main chain AND on every open side chain
static int capture_join_all(CaptureState& st, int policy, int nchains, hipStream_t ms) {
    if (policy == 0) return 0;
    if (!st.pending.empty()) {
        if (policy >= 2 || st.fork_at.nodes.empty()) { if (int rc = st.fork_at.get(ms)) return rc; }
        if (int rc = capture_flush(st, st.fork_at, nchains, ms)) return rc;
    }
    if (!st.sides[0].nodes.empty() || !st.sides[1].nodes.empty()) {
        CaptureTail cur;
        if (int rc = cur.get(ms)) return rc;
        for (CaptureTail& sd : st.sides) { cur.merge(sd); sd.nodes.clear(); }
        if (int rc = cur.set(ms)) return rc;
        st.joins += 1;
    }
    return 0;
}

static int run_commands_captured(const ColvoCmd* cmds, int n, hipStream_t ms) {
    const int policy = g_capture_policy.load(std::memory_order_relaxed);
    const int group = std::max(1, g_capture_group.load(std::memory_order_relaxed));
    const bool carry = g_capture_carry.load(std::memory_order_relaxed) != 0 && policy != 0;
    const int nchains = policy == 3 ? 2 : 1;   // policy 3 = policy 2 with the segments alternating between TWO side chains
    std::lock_guard<std::mutex> lock(g_cap_mu);
    CaptureState& st = g_cap;
    {
        unsigned long long id = 0;
        CaptureTail entry;
        if (int rc = entry.get(ms, &id)) return rc;
        if (id != st.id) {                      // a new capture: nothing is carried over from an earlier (ended or aborted) one
            st.id = id; st.seg = 0; st.pending.clear(); st.fork_at.nodes.clear();
            for (CaptureTail& sd : st.sides) sd.nodes.clear();
        }
        st.calls += 1;
        st.max_entry_deps = std::max(st.max_entry_deps, (long long)entry.nodes.size());
        if (!st.pending.empty()) st.carried += 1;
    }
    // commands carried over from the previous call are due at this call's first main command whatever the group size: they have
    // waited for a main node to exist behind their fork point, nothing else
    bool carried_due = !st.pending.empty();
    for (int k = 0; k < n; ++k) {
        const ColvoCmd& c = cmds[k];
        if (c.op == COLVO_CMD_FORK) {
            // policy 1: the side commands that follow depend on the main chain as it stands NOW (consecutive forks without a
            // main command in between share the point)
            if (policy == 1 && st.pending.empty()) { if (int rc = st.fork_at.get(ms)) return rc; }
            continue;
        }
        if (c.op == COLVO_CMD_JOIN) {
            if (int rc = capture_join_all(st, policy, nchains, ms)) return rc;
            continue;
        }
        if (c.stream == 1 && policy != 0) {
            if (c.op == COLVO_CMD_SIDE_SYNC && nchains == 2) {
                // two side chains: what follows must see both -- flush, then let the chain the next segment lands on depend on the other
                if (!st.pending.empty()) {
                    if (policy >= 2 || st.fork_at.nodes.empty()) { if (int rc = st.fork_at.get(ms)) return rc; }
                    if (int rc = capture_flush(st, st.fork_at, nchains, ms)) return rc;
                }
                CaptureTail all = st.sides[0];
                all.merge(st.sides[1]);
                st.sides[0] = all; st.sides[1] = all;
                continue;
            }
            st.pending.push_back(c);
            continue;
        }
        const bool due = !st.pending.empty() && (policy == 1 || (int)st.pending.size() >= group || carried_due);
        carried_due = false;
        if (due && policy >= 2) { if (int rc = st.fork_at.get(ms)) return rc; }   // every pending command's inputs exist by now
        if (int rc = run_one(c, k, (colvo_stream_t)ms)) return rc;                // first child of the fork point: stays on its stream
        st.main_nodes += 1;
        if (due) { if (int rc = capture_flush(st, st.fork_at, nchains, ms)) return rc; }
    }
    // Carry mode: the side chain stays open and the commands still pending are held back until the NEXT call has captured its first
    // main command (or colvo_capture_join is called) -- created now, while the main chain has no successor yet, the segment would
    // become the first child of the main chain's last node and the next call's main nodes would open a new stream
    // (profiles/r3_graph_replay.md section 3: 1.52 -> 1.93 ms).  Otherwise the call ends joined.
    if (!carry) return capture_join_all(st, policy, nchains, ms);
    return 0;
}

extern "C" int colvo_set_capture_policy(int policy, int group) {
    COLVO_CHECK_ARG(policy >= 0 && policy <= 3 && group >= 1, "colvo_set_capture_policy: policy 0..3, group >= 1");
    g_capture_policy.store(policy, std::memory_order_relaxed);
    g_capture_group.store(group, std::memory_order_relaxed);
    return 0;
}

extern "C" int colvo_set_capture_carry(int on) {
    g_capture_carry.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

extern "C" int colvo_capture_join(colvo_stream_t stream) {
    hipStream_t ms = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ms, &cap) != hipSuccess || cap == hipStreamCaptureStatusNone) return 0;   // eager: the host joins streams
    std::lock_guard<std::mutex> lock(g_cap_mu);
    CaptureState& st = g_cap;
    unsigned long long id = 0;
    CaptureTail entry;
    if (int rc = entry.get(ms, &id)) return rc;
    if (id != st.id) return 0;                  // nothing of this capture is open
    const int policy = g_capture_policy.load(std::memory_order_relaxed);
    return capture_join_all(st, policy, policy == 3 ? 2 : 1, ms);
}

extern "C" int colvo_capture_reset(colvo_stream_t stream) {
    hipStream_t ms = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ms && hipStreamIsCapturing(ms, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        set_error("colvo_capture_reset: the stream is being captured (end or abandon the capture first)");
        return (int)hipErrorStreamCaptureUnsupported;
    }
    {
        std::lock_guard<std::mutex> lock(g_cap_mu);
        CaptureState& st = g_cap;
        st.id = ~0ull;                       // no capture ever has this id: the next one starts empty whatever its id is
        st.seg = 0;
        st.pending.clear(); st.pending.shrink_to_fit();
        st.fork_at.nodes.clear();
        for (CaptureTail& sd : st.sides) sd.nodes.clear();
    }
    g_capture_policy.store(2, std::memory_order_relaxed);
    g_capture_group.store(2, std::memory_order_relaxed);
    g_capture_carry.store(0, std::memory_order_relaxed);
    LaunchTap& tap = g_launch_tap;           // (a command that failed between arming and disarming cannot leave it armed, but
    tap.stop = nullptr; tap.stream = nullptr; tap.used = 0;   //  "pre-capture state" should not depend on that argument)
    return 0;
}

extern "C" int colvo_graph_stats_reset(void) {
    std::lock_guard<std::mutex> lock(g_cap_mu);
    g_cap.calls = g_cap.main_nodes = g_cap.side_nodes = g_cap.segments = g_cap.joins = g_cap.carried = g_cap.max_entry_deps = 0;
    return 0;
}

extern "C" int colvo_graph_stats(colvo_stream_t stream, long long* out, int n) {
    COLVO_CHECK_ARG(out && n >= COLVO_GRAPH_STATS_N, "colvo_graph_stats: out must hold COLVO_GRAPH_STATS_N values");
    for (int i = 0; i < n; ++i) out[i] = -1;
    {
        std::lock_guard<std::mutex> lock(g_cap_mu);
        out[8] = g_cap.calls; out[9] = g_cap.main_nodes; out[10] = g_cap.side_nodes; out[11] = g_cap.segments;
        out[12] = g_cap.joins; out[13] = g_cap.carried; out[14] = g_cap.max_entry_deps;
        out[15] = (long long)g_cap.pending.size();
    }
    hipStream_t ms = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (!ms || hipStreamIsCapturing(ms, &cap) != hipSuccess || cap == hipStreamCaptureStatusNone) return 0;
    // the graph under construction on `stream`: node / edge / root counts and its fork / join structure
    hipGraph_t graph = nullptr;
    CaptureTail tail;
    if (int rc = tail.get(ms, nullptr, &graph)) return rc;
    size_t nn = 0, ne = 0, nr = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &nn);
    if (e == hipSuccess) e = hipGraphGetRootNodes(graph, nullptr, &nr);
    if (e == hipSuccess) e = hipGraphGetEdges(graph, nullptr, nullptr, &ne);
    if (e != hipSuccess) { set_error("colvo_graph_stats: %s", hipGetErrorString(e)); return (int)e; }
    std::vector<hipGraphNode_t> from(ne), to(ne), nodes(nn);
    if (ne) e = hipGraphGetEdges(graph, from.data(), to.data(), &ne);
    if (e == hipSuccess && nn) e = hipGraphGetNodes(graph, nodes.data(), &nn);
    if (e != hipSuccess) { set_error("colvo_graph_stats: %s", hipGetErrorString(e)); return (int)e; }
    std::sort(nodes.begin(), nodes.end());
    std::vector<int> outdeg(nn, 0), indeg(nn, 0);
    for (size_t i = 0; i < ne; ++i) {
        const size_t a = std::lower_bound(nodes.begin(), nodes.end(), from[i]) - nodes.begin();
        const size_t b = std::lower_bound(nodes.begin(), nodes.end(), to[i]) - nodes.begin();
        if (a < nn) ++outdeg[a];
        if (b < nn) ++indeg[b];
    }
    long long forks = 0, joins = 0, max_out = 0, max_in = 0, leaves = 0;
    for (size_t i = 0; i < nn; ++i) {
        forks += outdeg[i] >= 2; joins += indeg[i] >= 2; leaves += outdeg[i] == 0;
        max_out = std::max(max_out, (long long)outdeg[i]); max_in = std::max(max_in, (long long)indeg[i]);
    }
    out[0] = (long long)nn; out[1] = (long long)ne; out[2] = (long long)nr; out[3] = leaves;
    out[4] = forks; out[5] = joins; out[6] = max_out; out[7] = max_in;
    return 0;
}

extern "C" int colvo_run_commands(const ColvoCmd* cmds, int n, colvo_stream_t main_stream, colvo_stream_t side_stream) {
    COLVO_CHECK_ARG(cmds && n >= 0, "colvo_run_commands: bad arguments");
    hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ms, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return run_commands_captured(cmds, n, ms);
    }
    const int naux = ss ? aux_streams() : 0;
    hipStream_t side_cur = ss;            // where stream-1 commands go until the next FORK
    int side_idx = 0;                     // 0: the caller's side stream, i > 0: g_aux[i - 1]
    bool aux_dirty[MAX_AUX] = {false, false, false};   // aux i holds work the caller's side stream has not been ordered after
    hipEvent_t main_tail = nullptr;       // an event that stands for everything enqueued on the main stream so far (or none)
    const bool stop_forks = ss && TUNE(fork_stop_event) != 0;
    for (int k = 0; k < n; ++k) {
        const ColvoCmd& c = cmds[k];
        COLVO_CHECK_ARG(c.stream == 0 || (c.stream == 1 && ss), "colvo_run_commands: command %d needs a side stream", k);
        colvo_stream_t s = c.stream ? (colvo_stream_t)side_cur : main_stream;
        if (c.stream && side_idx > 0) aux_dirty[side_idx - 1] = true;
        int rc = 0;
        switch (c.op) {
            case COLVO_CMD_FORK: {       // the side stream continues after everything enqueued so far on the main stream
                COLVO_CHECK_ARG(ss, "colvo_run_commands: FORK without a side stream");
                if (naux) { side_idx = (side_idx + 1) % (naux + 1); side_cur = side_idx ? g_aux[side_idx - 1] : ss; }
                if (main_tail) {         // the main stream's last kernel carried this event: no marker on the main queue
                    const hipError_t he = hipStreamWaitEvent(side_cur, main_tail, 0);
                    if (he != hipSuccess) { set_error("colvo_run_commands: fork failed: %s", hipGetErrorString(he)); rc = (int)he; }
                } else {
                    rc = order_after(side_cur, ms, "fork");
                }
                break;
            }
            case COLVO_CMD_SIDE_SYNC: {  // the side stream in use continues after everything enqueued so far on the OTHER side streams
                COLVO_CHECK_ARG(ss, "colvo_run_commands: SIDE_SYNC without a side stream");
                if (side_cur != ss) rc = order_after(side_cur, ss, "side sync");       // (earlier calls hand everything back to ss)
                for (int i = 0; rc == 0 && i < naux; ++i)
                    if (aux_dirty[i] && g_aux[i] != side_cur) rc = order_after(side_cur, g_aux[i], "side sync");
                break;
            }
            case COLVO_CMD_JOIN: {       // the main stream continues after everything enqueued so far on the side stream
                COLVO_CHECK_ARG(ss, "colvo_run_commands: JOIN without a side stream");
                rc = order_after(ms, ss, "join");
                for (int i = 0; rc == 0 && i < naux; ++i)
                    if (aux_dirty[i]) rc = order_after(ms, g_aux[i], "join");
                main_tail = nullptr;     // the main stream now also stands for the joined side work
                break;
            }
            default:
                if (c.stream) { rc = run_one(c, k, s); break; }
                main_tail = nullptr;
                if (stop_forks && k + 1 < n && cmds[k + 1].op == COLVO_CMD_FORK) {
                    // a FORK follows: the command's kernels carry the event the side stream will wait on (common.h LaunchTap)
                    LaunchTap& tap = g_launch_tap;
                    tap.stop = next_event(); tap.stream = ms; tap.used = 0;
                    rc = run_one(c, k, s);
                    if (rc == 0 && tap.used > 0) main_tail = tap.stop;
                    tap.stop = nullptr;
                } else {
                    rc = run_one(c, k, s);
                }
        }
        if (rc != 0) return rc;   // the failing entry point has set the message
    }
    for (int i = 0; i < naux; ++i)
        if (aux_dirty[i]) { if (int rc = order_after(ss, g_aux[i], "side-stream hand-back")) return rc; }
    return 0;
}
