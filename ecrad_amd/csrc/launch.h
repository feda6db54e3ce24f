// launch.h -- host-callable launchers defined in the kernel_*.hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"

namespace ecrad {

// number of (nlev+1)*256-double scratch arrays a block of each kernel needs
int sw_ica_num_scratch_arrays(int mode);
int lw_ica_num_scratch_arrays(int mode);
int sw_tc_num_scratch_arrays();
int lw_tc_num_scratch_arrays();
size_t mcica_work_doubles(int nlev, int ng, int nloc);

hipError_t launch_sw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig* cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block);
hipError_t launch_lw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig* cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block);
hipError_t launch_sw_tc(int ngp, bool table_f32, int grid, size_t lds, hipStream_t st, const DevConfig* cfg,
                        const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep, double* scratch, size_t per_block);
hipError_t launch_lw_tc(int ngp, bool table_f32, int grid, size_t lds, hipStream_t st, const DevConfig* cfg,
                        const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep, double* scratch, size_t per_block);
hipError_t launch_crop(hipStream_t st, const DevConfig* cfg, const DevInputs& in);
hipError_t launch_tripleclouds_prep(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevCloudPrep& prep,
                                    double* cc_sw, double* cc_lw);
hipError_t launch_mcica_generator(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int ng, int seed_offset,
                                  double* od_scaling, double* tcc, int32_t* rng_state, double* work);
hipError_t launch_spectral_post(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevFlux& fx);
hipError_t launch_optics_dump(bool is_sw, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                              const DevConfig* cfg, const DevInputs& in, const DevOptics& out);

}  // namespace ecrad
