// Translation unit of the persistent tree sweep (k_sweep, dev_sweep.inc): the device helpers of dev_hip.hip + dev_step.inc and the
// one kernel, compiled with -mllvm -disable-machine-licm (see dev_hip.hip; Makefile).
#define S4B_SWEEP_TU 1
#include "dev_hip.hip"
