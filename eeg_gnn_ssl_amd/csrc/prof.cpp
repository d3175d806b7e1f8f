// Per-kernel timing with HIP events on the launch stream (prof.h; C ABI: include/eeg_dcrnn_prof.h).
#include <cstdio>
#include <cstdlib>
#include <cxxabi.h>
#include <cstring>
#include <string>
#include <vector>

#include "prof.h"

namespace eeg {
namespace {
struct ProfRec { const char* name; const void* kern; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open = nullptr;
const char* g_open_name = nullptr;
const void* g_open_kern = nullptr;
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
void prof_begin(const char* name, hipStream_t st, const void* kern) {
    if (!g_prof_on) return;
    g_open_kern = kern;
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
    g_recs.push_back({g_open_name, g_open_kern, g_open, b});
    g_open = nullptr;
}
void prof_enable(bool on) { g_prof_on = on; }
size_t prof_report(char* buf, size_t cap) {
    struct Agg { const char* name; const void* kern; int count; double ms; };
    std::vector<Agg> agg;
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        bool found = false;
        for (auto& a : agg)
            if (strcmp(a.name, r.name) == 0 && a.kern == r.kern) { a.count++; a.ms += ms; found = true; break; }
        if (!found) agg.push_back({r.name, r.kern, 1, (double)ms});
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    std::string out;
    char line[512];
    for (auto& a : agg) {
        // the kernel's symbol as a rocprofv3 kernel trace spells it ("void eeg::seq_fwd2_kernel<64, 3, 5, false>(eeg::SeqFwdArgs)")
        // without the return type, the namespace and the argument list
        std::string sym = "?";
        const char* mangled = a.kern != nullptr ? hipKernelNameRefByPtr(a.kern, nullptr) : nullptr;
        if (mangled != nullptr) {
            int st = 0;
            char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
            sym = (st == 0 && dem != nullptr) ? dem : mangled;
            free(dem);
            if (sym.rfind("void ", 0) == 0) sym = sym.substr(5);
            int depth = 0;
            for (size_t i = 0; i < sym.size(); ++i) {
                depth += sym[i] == '<';
                depth -= sym[i] == '>';
                if (sym[i] == '(' && depth == 0) { sym = sym.substr(0, i); break; }
            }
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
