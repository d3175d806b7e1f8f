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
/* Synchronises on the recorded events and writes "name launches total_ms" lines into buf (clears the records). */
int eeg_dcrnn_prof_report(char* buf, size_t cap);
/* Enqueues a kernel on `stream` that keeps every SIMD of the chip streaming fp32 MFMAs for 200 us of the chip-wide 100 MHz counter
 * (s_memrealtime) and ADDS, per workgroup, {shader-clock cycles (s_memtime), 100 MHz ticks} of the second 100 us to out2 (device,
 * 3 x int64, zeroed by the caller): out2[0] / out2[1] * 100 = the shader clock in MHz the part holds under sustained fp32 matrix
 * load at that point of the stream.  bench.py launches it right behind the timed steps (outside the timed region): the MFMA peak
 * the roofline fractions are priced against is a 2.4 GHz figure (`frac_at_held_clock`). */
int eeg_dcrnn_prof_clock_probe(int64_t* out2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EEG_DCRNN_PROF_H */
