// kernel_ica_sw_exact.hip -- the cloudless / homogeneous / McICA shortwave kernel (kernel_ica_sw.hip) with UNPACKED sweep records:
// five whole doubles per record instead of 39-bit mantissas in 32 bytes (kernels_common.h: pack5).  A translation unit of its own
// under other names, so that both forms are in the library; a handle created with ECRAD_HIP_EXACT_SCRATCH=1 in the environment
// launches these (pipeline.hip).
#define ECRAD_PACK_SW 0
#define SwScratch SwScratchExact
#define SwRec SwRecExact
#define sw_ica_kernel sw_ica_kernel_exact
#define sw_ica_scratch_doubles sw_ica_scratch_doubles_exact
#define launch_sw_ica launch_sw_ica_exact
#include "kernel_ica_sw.hip"
