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
    return 0;
}

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

extern "C" int e2k_plan_run(int plan, int first, int count, void* stream) {
    Plan* p = lookup(plan);
    if (!p) return E2K_ERR_ARG;
    const int n = (int)p->ops.size();
    if (first < 0 || first > n) return E2K_ERR_ARG;
    const int last = count < 0 ? n : first + count;
    if (last > n) return E2K_ERR_ARG;
    PlanTls& t = plan_tls();
    if (t.recording) return E2K_ERR_ARG;            // a replay inside a recording would record the replayed calls again
    for (int i = first; i < last; ++i) {
        const int rc = p->ops[i].run(stream);
        if (rc) return rc;
    }
    return 0;
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
    for (int i = 0; i < m && !rc; ++i) {
        rc = p->ops[first + i].run(stream);
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

extern "C" int e2k_plan_op_name(int plan, int index, char* buf_host, int nbuf) {
    Plan* p = lookup(plan);
    if (!p || index < 0 || index >= (int)p->ops.size() || !buf_host || nbuf <= 0) return E2K_ERR_ARG;
    std::strncpy(buf_host, p->ops[index].name, (size_t)nbuf - 1);
    buf_host[nbuf - 1] = 0;
    return 0;
}
