// kernel_spartacus_lw.hip -- the instantiations of spartacus_lw_kernel (kernel_spartacus.hip), as their own translation unit: the
// longwave flux-sweep kernel keeps the compiler's SLP vectoriser, the other SPARTACUS kernels are compiled without it (Makefile).
#define ECRAD_SP_TU_LW_SWEEP 1
#include "kernel_spartacus.hip"
