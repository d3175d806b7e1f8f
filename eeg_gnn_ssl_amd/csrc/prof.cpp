// Per-kernel timing with HIP events on the launch stream (prof.h; C ABI: include/eeg_dcrnn_prof.h).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "prof.h"

namespace eeg {
namespace {
struct ProfRec { const char* name; const char* sym; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open = nullptr;
const char* g_open_name = nullptr;
const char* g_open_sym = nullptr;
const char* g_prefix = nullptr;
std::vector<std::string*> g_names;
hipEvent_t prof_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
void prof_set_prefix(const char* prefix) { g_prefix = prefix; }
bool prof_is_on() { return g_prof_on; }
void prof_begin(const char* name, hipStream_t st, const char* sym) {
    if (!g_prof_on) return;
    g_open_sym = sym;
    g_open = prof_event();
    if (g_prefix != nullptr) {                       // interned so that records can keep a plain pointer
        std::string full = std::string(g_prefix) + name;
        const std::string* hit = nullptr;
        for (auto& n : g_names)
            if (*n == full) { hit = n; break; }
        if (hit == nullptr) { g_names.push_back(new std::string(full)); hit = g_names.back(); }
        name = hit->c_str();
    }
    g_open_name = name;
    (void)hipEventRecord(g_open, st);
}
void prof_end(hipStream_t st) {
    if (!g_prof_on || g_open == nullptr) return;
    hipEvent_t b = prof_event();
    (void)hipEventRecord(b, st);
    g_recs.push_back({g_open_name, g_open_sym, g_open, b});
    g_open = nullptr;
}
void prof_enable(bool on) { g_prof_on = on; }
size_t prof_report(char* buf, size_t cap) {
    struct Agg { const char* name; const char* sym; int count; double ms; };
    std::vector<Agg> agg;
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        bool found = false;
        for (auto& a : agg)
            if (strcmp(a.name, r.name) == 0 && a.sym == r.sym) { a.count++; a.ms += ms; found = true; break; }
        if (!found) agg.push_back({r.name, r.sym, 1, (double)ms});
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    std::string out;
    char line[512];
    for (auto& a : agg) {
        // kern_sym(): "const char *eeg::kern_sym() [K = &eeg::seq_fwd2_kernel<64, 3, 5, false>]" -> "seq_fwd2_kernel<64, 3, 5, false>"
        std::string sym = a.sym != nullptr ? a.sym : "?";
        const size_t k = sym.find("K = ");
        if (k != std::string::npos) {
            sym = sym.substr(k + 4);
            if (!sym.empty() && sym[0] == '&') sym = sym.substr(1);
            if (!sym.empty() && sym.back() == ']') sym.pop_back();
            if (sym.rfind("eeg::", 0) == 0) sym = sym.substr(5);
        }
        snprintf(line, sizeof(line), "%s %d %.6f %s\n", a.name, a.count, a.ms, sym.c_str());
        out += line;
    }
    if (out.size() + 1 > cap) return out.size() + 1;
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}
}  // namespace eeg
