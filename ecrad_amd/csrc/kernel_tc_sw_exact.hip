// kernel_tc_sw_exact.hip -- the Tripleclouds shortwave kernel (kernel_tc.hip) with UNPACKED sweep records: five whole doubles per
// record instead of 39-bit mantissas in 32 bytes (kernels_common.h: pack5).  A translation unit of its own under other names, so that
// both forms are in the library; a handle created with ECRAD_HIP_EXACT_SCRATCH=1 in the environment launches these (pipeline.hip).
#define ECRAD_PACK_SW 0
#define ECRAD_TC_TU_EXACT 1
#define TcSwScratch TcSwScratchExact
#define sw_tc_kernel sw_tc_kernel_exact
#define sw_tc_scratch_doubles sw_tc_scratch_doubles_exact
#define launch_sw_tc launch_sw_tc_exact
#include "kernel_tc.hip"
