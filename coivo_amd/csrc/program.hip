// program.hip -- command-list executor: one C-ABI call enqueues a whole recorded sequence of library calls (a network's
// forward or backward) on two streams.
//
// Why: at batch 8 the DCDP step is ~115 launches of 5-50 us.  Driven call by call from the Python host the enqueue path
// (~18 us per launch: tensor allocation, ctypes marshalling, event objects for the weight-gradient side stream) was
// LONGER than the GPU work, i.e. the step was host-bound.  A hipGraph removes the host cost but replays the two-stream
// backward serially; this executor keeps the main / side stream concurrency and costs one hipLaunchKernel per command.
//
// The host side (coivo_amd/program.py) records the commands once per (network, shape) with persistent activation
// buffers, patches the few per-call pointers (input images, output, incoming gradients) and replays.
#include <atomic>
#include <mutex>

#include "common.h"

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
        for (unsigned i = 0; i < NEV; ++i) (void)hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming);
    });
    return g_ev[g_ev_next.fetch_add(1, std::memory_order_relaxed) % NEV];
}
}  // namespace

extern "C" int colvo_run_commands(const ColvoCmd* cmds, int n, colvo_stream_t main_stream, colvo_stream_t side_stream) {
    COLVO_CHECK_ARG(cmds && n >= 0, "colvo_run_commands: bad arguments");
    hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
    for (int k = 0; k < n; ++k) {
        const ColvoCmd& c = cmds[k];
        COLVO_CHECK_ARG(c.stream == 0 || (c.stream == 1 && ss), "colvo_run_commands: command %d needs a side stream", k);
        colvo_stream_t s = c.stream ? side_stream : main_stream;
        int rc = 0;
        switch (c.op) {
            case COLVO_CMD_CONV_FWD:
                rc = colvo_conv_fwd(&c.desc, c.p[0], c.p[1], c.p[2], (const float*)c.p[3], (void*)c.p[4], s);
                break;
            case COLVO_CMD_CONV_DGRAD:
                rc = colvo_conv_dgrad(&c.desc, c.i[0], c.p[0], c.p[1], c.p[2], (void*)c.p[3], c.i[1], s);
                break;
            case COLVO_CMD_CONV_WGRAD:
                rc = colvo_conv_wgrad(&c.desc, c.p[0], c.p[1], c.p[2], (float*)c.p[3], (float*)c.p[4], s);
                break;
            case COLVO_CMD_PACK_NCHW: {
                const float* src[4] = {(const float*)c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3]};
                rc = colvo_pack_nchw(c.i[0], src, &c.i[1], c.i[5], c.i[6], c.i[7], c.i[8], c.i[9], (void*)c.p[4], s);
                break;
            }
            case COLVO_CMD_UNPACK_NHWC_GRAD:
                rc = colvo_unpack_nhwc_grad(c.i[0], c.p[0], c.i[1], c.i[2], c.i[3], c.i[4], c.i[5], c.i[6], (float*)c.p[1],
                                            c.i[7], s);
                break;
            case COLVO_CMD_DEPTH_HEAD_FWD:
                rc = colvo_depth_head_fwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], c.i[1], c.i[2], c.i[3],
                                          c.i[4], c.f[0], c.f[1], (float*)c.p[3], s);
                break;
            case COLVO_CMD_DEPTH_HEAD_BWD:
                rc = colvo_depth_head_bwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3],
                                          c.i[1], c.i[2], c.i[3], c.i[4], c.f[0], c.f[1], (float*)c.p[4], (void*)c.p[5],
                                          (float*)c.p[6], (float*)c.p[7], s);
                break;
            case COLVO_CMD_DEPTH_HEAD_BWD_PARTS:
                rc = colvo_depth_head_bwd_parts(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3],
                                                (const float*)c.p[4], (const float*)c.p[5], (const float*)c.p[6],
                                                (const float*)c.p[7], c.i[1], c.i[2], c.i[3], c.i[4], c.f[0], c.f[1],
                                                (float*)c.p[8], (void*)c.p[9], nullptr, nullptr, s);
                break;
            case COLVO_CMD_DEPTH_HEAD_WGRAD:
                rc = colvo_depth_head_wgrad(c.i[0], c.p[0], (const float*)c.p[1], c.i[1], c.i[2], c.i[3], c.i[4],
                                            (float*)c.p[2], (float*)c.p[3], s);
                break;
            case COLVO_CMD_POSE_HEAD_FWD:
                rc = colvo_pose_head_fwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], c.i[1], c.i[2], c.i[3],
                                         c.f[0], c.f[1], (float*)c.p[3], s);
                break;
            case COLVO_CMD_POSE_HEAD_BWD:
                rc = colvo_pose_head_bwd(c.i[0], c.p[0], (const float*)c.p[1], (const float*)c.p[2], (const float*)c.p[3],
                                         (const float*)c.p[4], (const float*)c.p[8], (const float*)c.p[9], c.i[1], c.i[2],
                                         c.i[3], c.f[0], c.f[1], (void*)c.p[5], (float*)c.p[6], (float*)c.p[7], s);
                break;
            case COLVO_CMD_FORK: {       // the side stream continues after everything enqueued so far on the main stream
                COLVO_CHECK_ARG(ss, "colvo_run_commands: FORK without a side stream");
                hipEvent_t e = next_event();
                hipError_t he = hipEventRecord(e, ms);
                if (he == hipSuccess) he = hipStreamWaitEvent(ss, e, 0);
                rc = (int)he;
                if (he != hipSuccess) set_error("colvo_run_commands: fork failed: %s", hipGetErrorString(he));
                break;
            }
            case COLVO_CMD_JOIN: {       // the main stream continues after everything enqueued so far on the side stream
                COLVO_CHECK_ARG(ss, "colvo_run_commands: JOIN without a side stream");
                hipEvent_t e = next_event();
                hipError_t he = hipEventRecord(e, ss);
                if (he == hipSuccess) he = hipStreamWaitEvent(ms, e, 0);
                rc = (int)he;
                if (he != hipSuccess) set_error("colvo_run_commands: join failed: %s", hipGetErrorString(he));
                break;
            }
            default:
                set_error("colvo_run_commands: unknown op %d in command %d", c.op, k);
                return (int)hipErrorInvalidValue;
        }
        if (rc != 0) return rc;   // the failing entry point has set the message
    }
    return 0;
}
