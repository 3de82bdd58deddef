// Launch-plan registry and replay (see plan.h): the C++ side of the backbone scheduler.
#include "e2k_device.h"
#include "plan.h"
#include "../../include/e2k.h"

#include <cstring>
#include <memory>
#include <mutex>

namespace e2k {

PlanTls& plan_tls() {
    static thread_local PlanTls t;
    return t;
}

Plan::~Plan() {
    for (void* e : events)
        if (e) hipEventDestroy((hipEvent_t)e);
}

namespace {
std::mutex g_mu;
std::vector<std::unique_ptr<Plan>> g_plans;        // handle = index + 1; freed slots stay as nullptr

Plan* lookup(int h) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (h <= 0 || h > (int)g_plans.size()) return nullptr;
    return g_plans[h - 1].get();
}
}  // namespace
}  // namespace e2k

using namespace e2k;

extern "C" int e2k_plan_begin(void) {
    PlanTls& t = plan_tls();
    if (t.recording) return E2K_ERR_ARG;
    t.recording = new Plan();
    t.depth = 0;
    t.lane = 0;
    return 0;
}

extern "C" int e2k_plan_lane(int lane) {
    if (lane < 0 || lane >= PLAN_MAX_LANES) return E2K_ERR_ARG;
    plan_tls().lane = lane;
    return 0;
}

static int plan_event_op(int lane, int ev, int kind) {
    if (lane < 0 || lane >= PLAN_MAX_LANES || ev < 0 || ev >= 65536) return E2K_ERR_ARG;
    PlanTls& t = plan_tls();
    if (!t.recording) return 0;                     // outside a recording the caller orders its streams itself
    if (kind == PLAN_EVENT_WAIT && ev >= (int)t.recording->events.size()) return E2K_ERR_ARG;     // wait before any record
    if (ev >= (int)t.recording->events.size()) t.recording->events.resize(ev + 1, nullptr);
    PlanOp op;
    op.name = kind == PLAN_EVENT_RECORD ? "lane_event_record" : "lane_event_wait";
    op.lane = lane;
    op.kind = kind;
    op.ev = ev;
    t.recording->ops.push_back(std::move(op));
    return 0;
}

extern "C" int e2k_plan_event_record(int lane, int ev) { return plan_event_op(lane, ev, PLAN_EVENT_RECORD); }
extern "C" int e2k_plan_event_wait(int lane, int ev) { return plan_event_op(lane, ev, PLAN_EVENT_WAIT); }

extern "C" int e2k_query_plan_recorded(void) {
    PlanTls& t = plan_tls();
    return t.recording ? (int)t.recording->ops.size() : -1;
}

extern "C" int e2k_query_plan_end(void) {
    PlanTls& t = plan_tls();
    if (!t.recording) return -E2K_ERR_ARG;
    std::unique_ptr<Plan> p(t.recording);
    t.recording = nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_plans.size(); ++i)
        if (!g_plans[i]) { g_plans[i] = std::move(p); return (int)i + 1; }
    g_plans.push_back(std::move(p));
    return (int)g_plans.size();
}

extern "C" int e2k_plan_abort(void) {
    PlanTls& t = plan_tls();
    delete t.recording;
    t.recording = nullptr;
    t.depth = 0;
    return 0;
}

extern "C" int e2k_plan_free(int plan) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (plan <= 0 || plan > (int)g_plans.size() || !g_plans[plan - 1]) return E2K_ERR_ARG;
    g_plans[plan - 1].reset();
    return 0;
}

extern "C" int e2k_query_plan_size(int plan) {
    Plan* p = lookup(plan);
    return p ? (int)p->ops.size() : -1;
}

extern "C" int e2k_plan_run_lanes(int plan, int first, int count, void** streams_host, int nstreams) {
    Plan* p = lookup(plan);
    if (!p || !streams_host || nstreams < 1 || nstreams > PLAN_MAX_LANES) return E2K_ERR_ARG;
    const int n = (int)p->ops.size();
    if (first < 0 || first > n) return E2K_ERR_ARG;
    const int last = count < 0 ? n : first + count;
    if (last > n) return E2K_ERR_ARG;
    PlanTls& t = plan_tls();
    if (t.recording) return E2K_ERR_ARG;            // a replay inside a recording would record the replayed calls again
    for (int i = first; i < last; ++i) {
        PlanOp& op = p->ops[i];
        const int lane = op.lane < nstreams ? op.lane : 0;      // lanes the caller has no stream for run on lane 0
        if (op.kind == PLAN_CALL) {
            const int rc = op.run(streams_host[lane]);
            if (rc) return rc;
            continue;
        }
        if (nstreams == 1) continue;                // one stream: program order is the order
        void*& ev = p->events[op.ev];
        if (!ev) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 1000;
            ev = (void*)e;
        }
        const hipError_t rc = op.kind == PLAN_EVENT_RECORD ? hipEventRecord((hipEvent_t)ev, (hipStream_t)streams_host[lane])
                                                           : hipStreamWaitEvent((hipStream_t)streams_host[lane], (hipEvent_t)ev, 0);
        if (rc != hipSuccess) return 1000 + (int)rc;
    }
    return 0;
}

// ---- HIP-graph form of a replay (round 6) -------------------------------------------------------------------------------------------
// e2k_plan_run_lanes issues ~1600 launches + ~1200 event operations per cfg3 step from the host: 34 ms of host time (21 us per recorded
// launch), which is the step of the dim-512 configuration (cfg2: 11.3 of 14.1 ms) and the floor under every future kernel gain.  A range
// of a plan that no other stream has to interleave with (a forward; a backward when no gradient exchange is installed) is captured ONCE
// -- the very same replay loop under hipStreamBeginCapture: lane 0 is the capturing stream, the side lanes are forked from it before the
// first call and joined back after the last, the recorded ordering points become graph edges -- and re-issued with one hipGraphLaunch.
// The eager replay stays the path of the backward segments between which the RCCL exchange stream interleaves.
namespace {
struct PlanGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
std::vector<std::unique_ptr<PlanGraph>> g_graphs;      // handle = index + 1 (g_mu)
}  // namespace

extern "C" int e2k_query_plan_graph_capture(int plan, int first, int count, void** streams_host, int nstreams) {
    Plan* p = lookup(plan);
    if (!p || !streams_host || nstreams < 1 || nstreams > PLAN_MAX_LANES) return -E2K_ERR_ARG;
    if (plan_tls().recording) return -E2K_ERR_ARG;
    // every event the replay loop may create lazily is created BEFORE the capture starts (object creation is not a stream operation)
    for (void*& ev : p->events)
        if (!ev) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1000;
            ev = (void*)e;
        }
    hipEvent_t fork = nullptr, join[PLAN_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return -1000;
    for (int k = 1; k < nstreams; ++k)
        if (hipEventCreateWithFlags(&join[k], hipEventDisableTiming) != hipSuccess) return -1000;
    // lane 0 is captured on a stream of the library's own: the caller's stream is usually the legacy default stream, which cannot capture
    // (hipErrorStreamCaptureUnsupported: the first hardware run of this entry point); the graph is launched on the caller's stream all the same
    static hipStream_t cap = nullptr;
    if (!cap && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) return -1000;
    void* lanes[PLAN_MAX_LANES] = {(void*)cap, nullptr, nullptr, nullptr};
    for (int k = 1; k < nstreams; ++k) lanes[k] = streams_host[k];
    streams_host = lanes;
    hipStream_t main = cap;
    int rc = 0;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(main, hipStreamCaptureModeRelaxed) != hipSuccess) rc = 1001;
    if (!rc) {
        // fork: a side lane whose first recorded call has no ordering point in front of it would otherwise run OUTSIDE the capture
        if (nstreams > 1) {
            if (hipEventRecord(fork, main) != hipSuccess) rc = 1002;
            for (int k = 1; k < nstreams && !rc; ++k)
                if (hipStreamWaitEvent((hipStream_t)streams_host[k], fork, 0) != hipSuccess) rc = 1002;
        }
        if (!rc) rc = e2k_plan_run_lanes(plan, first, count, streams_host, nstreams);
        // join: every lane's tail is an ancestor of the capture's end
        for (int k = 1; k < nstreams && !rc; ++k) {
            if (hipEventRecord(join[k], (hipStream_t)streams_host[k]) != hipSuccess) rc = 1003;
            else if (hipStreamWaitEvent(main, join[k], 0) != hipSuccess) rc = 1003;
        }
        if (hipStreamEndCapture(main, &graph) != hipSuccess && !rc) rc = 1004;
    }
    (void)hipEventDestroy(fork);
    for (int k = 1; k < nstreams; ++k) (void)hipEventDestroy(join[k]);
    if (rc || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        return -(rc ? rc : 1004);
    }
    std::unique_ptr<PlanGraph> g(new PlanGraph());
    g->graph = graph;
    if (hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return -1005;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_graphs.size(); ++i)
        if (!g_graphs[i]) { g_graphs[i] = std::move(g); return (int)i + 1; }
    g_graphs.push_back(std::move(g));
    return (int)g_graphs.size();
}

extern "C" int e2k_plan_graph_launch(int graph, void* stream) {
    hipGraphExec_t ex = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (graph <= 0 || graph > (int)g_graphs.size() || !g_graphs[graph - 1]) return E2K_ERR_ARG;
        ex = g_graphs[graph - 1]->exec;
    }
    return hipGraphLaunch(ex, (hipStream_t)stream) == hipSuccess ? 0 : 1006;
}

extern "C" int e2k_plan_graph_free(int graph) {
    std::unique_ptr<PlanGraph> g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (graph <= 0 || graph > (int)g_graphs.size() || !g_graphs[graph - 1]) return E2K_ERR_ARG;
        g = std::move(g_graphs[graph - 1]);
    }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    return 0;
}

extern "C" int e2k_plan_run(int plan, int first, int count, void* stream) {
    return e2k_plan_run_lanes(plan, first, count, &stream, 1);
}

extern "C" int e2k_plan_profile(int plan, int first, int count, float* ms_host, void* stream) {
    Plan* p = lookup(plan);
    if (!p || !ms_host) return E2K_ERR_ARG;
    const int n = (int)p->ops.size();
    if (first < 0 || first > n) return E2K_ERR_ARG;
    const int last = count < 0 ? n : first + count;
    if (last > n) return E2K_ERR_ARG;
    if (plan_tls().recording) return E2K_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int m = last - first;
    std::vector<hipEvent_t> ev(m + 1);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return 1000;
    int rc = 0;
    hipEventRecord(ev[0], st);
    for (int i = 0; i < m && !rc; ++i) {          // one stream: the lane ordering points are skipped (0 ms)
        if (p->ops[first + i].kind == PLAN_CALL) rc = p->ops[first + i].run(stream);
        hipEventRecord(ev[i + 1], st);
    }
    hipStreamSynchronize(st);
    if (!rc)
        for (int i = 0; i < m; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            ms_host[i] = ms;
        }
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

extern "C" int e2k_query_plan_op_lane(int plan, int index) {
    Plan* p = lookup(plan);
    if (!p || index < 0 || index >= (int)p->ops.size()) return -1;
    return p->ops[index].lane;
}

extern "C" int e2k_plan_op_name(int plan, int index, char* buf_host, int nbuf) {
    Plan* p = lookup(plan);
    if (!p || index < 0 || index >= (int)p->ops.size() || !buf_host || nbuf <= 0) return E2K_ERR_ARG;
    std::strncpy(buf_host, p->ops[index].name, (size_t)nbuf - 1);
    buf_host[nbuf - 1] = 0;
    return 0;
}
