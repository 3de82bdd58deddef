// Launch plans: the native scheduler behind the hand-scheduled backbone.
//
// A forward (or one backward segment) of the E2-TTS backbone is a fixed sequence of ~1500 C-ABI calls whose pointer and
// size arguments do not change from step to step once the buffers are static.  Driving that sequence from Python costs
// ~35 us of interpreter / ctypes time per call -- as long as the kernels themselves take.  While a plan is being
// recorded every e2k_* compute entry point appends a closure of its own arguments (all plain pointers and integers) and
// still executes; e2k_plan_run then re-issues the recorded calls from C++ (one hipLaunchKernel each, a few
// microseconds), on whatever stream the caller passes.  It is NOT a HIP graph: the launches go to the stream eagerly, so
// the queue stays as deep as with hand-written host code and other streams (the RCCL gradient all-reduce) interleave
// between plan segments.
//
// Every compute entry point is written as  `static int foo_impl(args..., void* stream)`  +
//     extern "C" int e2k_foo(args..., void* stream) { return e2k::dispatch("foo", foo_impl, args..., stream); }
#pragma once
#include <functional>
#include <tuple>
#include <utility>
#include <vector>

namespace e2k {

// Launch lanes.  The backbone's schedule is not a chain: the text branches of layer i + 1 only need what the cross
// projection of layer i left behind, and no weight gradient is read before the optimizer (or the gradient all-reduce)
// runs.  A recorded call therefore carries a LANE -- lane 0 is the caller's stream, lanes 1.. are side streams handed to
// e2k_plan_run_lanes -- and the recording holds explicit ordering points between lanes: "record event e on lane a",
// "lane b waits for event e" (hipEventRecord / hipStreamWaitEvent at replay).  Replayed on one stream (e2k_plan_run,
// e2k_plan_profile) the lanes collapse into program order and the ordering points are skipped.
constexpr int PLAN_MAX_LANES = 4;
enum PlanOpKind { PLAN_CALL = 0, PLAN_EVENT_RECORD = 1, PLAN_EVENT_WAIT = 2 };

struct PlanOp {
    const char* name;
    std::function<int(void*)> run;      // argument: the stream to enqueue on
    int lane = 0;
    int kind = PLAN_CALL;
    int ev = -1;                        // PLAN_EVENT_*: index into Plan::events
};

struct Plan {
    std::vector<PlanOp> ops;
    std::vector<void*> events;          // hipEvent_t, created at the first multi-lane replay
    ~Plan();
};

struct PlanTls {
    Plan* recording = nullptr;
    int depth = 0;                      // entry points that call other entry points record only the outermost call
    int lane = 0;                       // lane of the calls recorded from now on (e2k_plan_lane)
};
PlanTls& plan_tls();

template <class T> struct same_ { typedef T type; };

template <class... A>
int dispatch(const char* name, int (*impl)(A...), typename same_<A>::type... a) {
    static_assert(sizeof...(A) >= 1, "the last argument of a compute entry point is its stream");
    PlanTls& t = plan_tls();
    if (t.recording && t.depth == 0) {
        std::tuple<A...> args(a...);
        PlanOp op;
        op.name = name;
        op.run = [impl, args](void* stream) mutable -> int {
            std::get<sizeof...(A) - 1>(args) = stream;
            return std::apply(impl, args);
        };
        op.lane = t.lane;
        t.recording->ops.push_back(std::move(op));
    }
    ++t.depth;
    const int rc = impl(a...);
    --t.depth;
    return rc;
}

}  // namespace e2k
