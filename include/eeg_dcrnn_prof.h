/*
 * eeg_dcrnn_prof.h — measurement hook of libeeg_dcrnn_hip.so: per-kernel HIP-event timing on the
 * launch stream, which bench.py needs for its live `roofline` figures (a C-ABI call launches several
 * kernels, so the caller cannot bracket them itself).  Off by default and free when off; the recorder
 * is the only process-global state of the library and is never touched by the compute entry points
 * unless enabled.
 */
#ifndef EEG_DCRNN_PROF_H
#define EEG_DCRNN_PROF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enable(1) starts recording an event pair around every kernel launch; enable(0) stops. */
int eeg_dcrnn_prof_enable(int on);
/* Synchronises on the recorded events and writes one line per (kernel role, kernel symbol) into buf and clears the records:
 *     "role launches total_ms symbol\n"     -- FOUR fields; `symbol` is the demangled kernel name of the launches (it contains
 * blanks: split with a field limit of 3) and a role whose launches ran different instantiations yields several lines (sum them
 * for a per-role figure).  Returns the bytes needed (incl. NUL) when cap is too small, else 0. */
int eeg_dcrnn_prof_report(char* buf, size_t cap);
/* Enqueues a kernel on `stream` that keeps every SIMD of the chip streaming fp32 MFMAs for 200 us of the chip-wide 100 MHz counter
 * (s_memrealtime) and ADDS, per workgroup, {shader-clock cycles (s_memtime), 100 MHz ticks} of the second 100 us to out2 (device,
 * 3 x int64, zeroed by the caller): out2[0] / out2[1] * 100 = the shader clock in MHz the part holds under sustained fp32 matrix
 * load at that point of the stream.  bench.py launches it right behind the timed steps (outside the timed region): the MFMA peak
 * the roofline fractions are priced against is a 2.4 GHz figure (`frac_at_held_clock`). */
int eeg_dcrnn_prof_clock_probe(int64_t* out2, void* stream);
/* buf4 (device, 4 x int64, zeroed by the caller; NULL switches it off again): while set, every launch of the two-wave recurrent
 * kernels adds {shader-clock cycles, 100 MHz ticks} of ONE lane's time loop to buf4[0..1] (forward) / buf4[2..3] (backward):
 * cycles / ticks * 100 = the clock in MHz the part holds INSIDE those kernels (steady state on the boxes of round 4: 2.37-2.39 GHz
 * against the 2.4 GHz the MFMA peak is quoted at; a process that has just started runs its first tens of milliseconds at
 * 1.9-2.2 GHz -- the ramp behind the slower first steps of a fresh bench run).  Process-global like the event recorder. */
int eeg_dcrnn_prof_clock_samples(int64_t* buf4);

#ifdef __cplusplus
}
#endif
#endif /* EEG_DCRNN_PROF_H */
