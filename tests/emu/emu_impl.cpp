// TEST INFRASTRUCTURE: the one translation unit that holds the emulator's scheduler.
#define EEG_SIMT_EMU_IMPL
#include "simt_emu.h"
