// kernel_ica_lw_clear.hip -- the cloudless and homogeneous instantiations of lw_ica_kernel (kernel_ica_lw.hip), as their own
// translation unit because they are compiled with another instruction-scheduling strategy (Makefile: SCHED_kernel_ica_lw_clear).
#define ECRAD_LW_TU_CLEAR 1
#include "kernel_ica_lw.hip"
