// Optional per-kernel timing with HIP events on the launch stream (used by bench.py to report the
// roofline fraction of each kernel "live").  Disabled by default: zero overhead unless
// eeg_dcrnn_prof_enable(1) was called.
#pragma once
#include "common.h"

namespace eeg {
// kern: the host-side function of the launched kernel instantiation (resolved to its symbol in the report), or nullptr
void prof_begin(const char* name, hipStream_t st, const void* kern = nullptr);
bool prof_is_on();
void prof_end(hipStream_t st);
void prof_set_prefix(const char* prefix);      // records made while set are named prefix+name
void prof_enable(bool on);
// "name count total_ms symbol\n" per (kernel role, kernel symbol) since the last report, into buf; returns the bytes needed (incl. NUL) if cap is too small, else 0
size_t prof_report(char* buf, size_t cap);
struct ProfPrefix {                             // RAII: tag the launches of one API call (e.g. "dec_")
    explicit ProfPrefix(const char* p) { prof_set_prefix(p); }
    ~ProfPrefix() { prof_set_prefix(nullptr); }
};

}  // namespace eeg

#define EEG_LAUNCH_P(name, kern, grid, block, smem, stream, ...) \
    do {                                                           \
        eeg::prof_begin(name, stream, reinterpret_cast<const void*>(&kern));      \
        EEG_LAUNCH(kern, grid, block, smem, stream, __VA_ARGS__);  \
        eeg::prof_end(stream);                                     \
    } while (0)
