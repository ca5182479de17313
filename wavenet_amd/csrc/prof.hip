// Optional in-library profiler: when enabled, every per-op entry point brackets the kernels it
// enqueues with two hipEvents recorded on the caller's stream.  bench.py uses it to time the
// dominant kernel inside the timed region (the stack-level entry points call the per-op ones, so
// the breakdown survives the C++ layer loop).  Off by default: zero cost beyond one branch.
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "wn_common.hpp"

namespace wn {
namespace {
struct Rec { int id; hipEvent_t e0, e1; };
std::mutex g_mu;
bool g_on = false;
std::vector<std::string> g_names;
std::vector<Rec> g_recs;
thread_local int tl_group = -1;      // id of the open ProfGroup of this thread, if any
int name_id(const char* n) {
    for (size_t i = 0; i < g_names.size(); ++i)
        if (g_names[i] == n) return (int)i;
    g_names.emplace_back(n);
    return (int)g_names.size() - 1;
}
}  // namespace

ProfScope::ProfScope(const char* name, void* stream) : on(false) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return;
    id = name_id(name);
    if (id == tl_group) return;          // inside a ProfGroup of the same name: the group's bracket covers this call
    s = as_stream(stream);
    if (hipEventCreate(&e0) != hipSuccess) return;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; }
    (void)hipEventRecord(e0, s);
    on = true;
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(Rec{id, e0, e1});
}
ProfGroup::ProfGroup(const char* name, void* stream) : scope(name, stream) {
    if (scope.on) tl_group = scope.id;
}
ProfGroup::~ProfGroup() {
    if (scope.on) tl_group = -1;
}
}  // namespace wn

using namespace wn;

extern "C" {

int wn_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_recs.clear();
    g_on = on != 0;
    return WN_OK;
}

// Writes "name calls total_ms min_ms max_ms\n" lines; returns the number of bytes needed.
int wn_prof_report(char* buf, int buflen) {
    std::lock_guard<std::mutex> lk(g_mu);
    struct Agg { long calls = 0; double ms = 0, mn = 1e30, mx = 0; };
    std::map<int, Agg> agg;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Agg& a = agg[r.id];
        a.calls++; a.ms += ms; if (ms < a.mn) a.mn = ms; if (ms > a.mx) a.mx = ms;
    }
    std::string out;
    char line[256];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s %ld %.6f %.6f %.6f\n", g_names[kv.first].c_str(), kv.second.calls, kv.second.ms,
                 kv.second.mn, kv.second.mx);
        out += line;
    }
    if (buf && buflen > 0) {
        int n = (int)out.size() < buflen - 1 ? (int)out.size() : buflen - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int)out.size() + 1;
}

}  // extern "C"
