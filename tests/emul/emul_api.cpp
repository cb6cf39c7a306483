// TEST INFRASTRUCTURE ONLY — exports the C-ABI over the CPU emulation of the device layer as `emu_*`.
#ifndef S4B_PREFIX
#define S4B_PREFIX emu_
#endif
#include "dev_cpu.hpp"
#define S4B_DEV s4b::DevCpu
#include "../../stan4bart_amd/csrc/c_api.inc"
