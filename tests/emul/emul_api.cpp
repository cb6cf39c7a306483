// TEST INFRASTRUCTURE ONLY — exports the C-ABI over the CPU emulation of the device layer as `emu_*`.
#define S4B_PREFIX emu_
#include "dev_cpu.hpp"
#define S4B_DEV s4b::DevCpu
#include "../../stan4bart_amd/csrc/c_api.inc"
