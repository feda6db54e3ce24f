// launch.h -- host-callable launchers defined in the kernel_*.hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"

namespace ecrad {

// Dynamic LDS above 64 KB per block needs an explicit opt-in (gfx950 has 160 KB per CU)
#define ECRAD_ALLOW_LDS(kernel, bytes)                                                                       \
  do {                                                                                                       \
    if ((bytes) > 60 * 1024) {                                                                               \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));         \
      if (e_ != hipSuccess) return e_;                                                                       \
    }                                                                                                        \
  } while (0)

// doubles of block-private sweep scratch each kernel needs per block
size_t sw_ica_scratch_doubles(int mode, int nlev);
size_t lw_ica_scratch_doubles(int mode, int nlev);
size_t lw_scat_scratch_doubles(int nlev);
size_t sw_tc_scratch_doubles(int nlev);
size_t lw_tc_scratch_doubles(int nlev, bool aerosol_scattering);

// g0: first g-point of the launch (spectra wider than 64 g-points run in chunks of `ngp`)
hipError_t launch_sw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide);
hipError_t launch_lw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide);
// the same solvers with longwave aerosol scattering (kernel_lw_scat.hip)
hipError_t launch_lw_scat(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide);
hipError_t launch_sw_tc(int ngp, bool table_f32, int grid, size_t lds, hipStream_t st, const DevConfig& cfg,
                        const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep, double* scratch, size_t per_block, int* counter,
                        const DevCkdModel& m, int g0);
// the same two shortwave launchers with UNPACKED sweep records (kernel_ica_sw_exact.hip, kernel_tc_sw_exact.hip: ECRAD_PACK_SW = 0),
// chosen per handle by ECRAD_HIP_EXACT_SCRATCH (host_internal.h: exact_scratch)
size_t sw_ica_scratch_doubles_exact(int mode, int nlev);
size_t sw_tc_scratch_doubles_exact(int nlev);
hipError_t launch_sw_ica_exact(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                               const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                               double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide);
hipError_t launch_sw_tc_exact(int ngp, bool table_f32, int grid, size_t lds, hipStream_t st, const DevConfig& cfg,
                              const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep, double* scratch, size_t per_block, int* counter,
                              const DevCkdModel& m, int g0);
// dst(col, l) = sum over chunks of partial profiles (chunk order), columns istartcol..iendcol
hipError_t launch_spectral_profile_sum(hipStream_t st, const DevInputs& in, const double* per_g, double* dst, int ng, int nspec,
                                       const int32_t* ispec);
hipError_t launch_combine_derivatives(hipStream_t st, const DevInputs& in, double* dst, const double* a, const double* b,
                                      size_t chunk_stride, int nchunk, const double* cloud_cover, double threshold);
hipError_t launch_combine_partials(hipStream_t st, const DevInputs& in, double* dst, const double* partial, size_t chunk_stride, int nchunk);
hipError_t launch_lw_tc(int ngp, bool table_f32, int grid, size_t lds, hipStream_t st, const DevConfig& cfg,
                        const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep, double* scratch, size_t per_block, int* counter,
                        const DevCkdModel& m, int g0, bool wide);
hipError_t launch_order(hipStream_t st, const DevInputs& in, int32_t* flag);
hipError_t launch_crop(hipStream_t st, const DevConfig* cfg, const DevInputs& in);
hipError_t launch_column_order(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int32_t* order, int window);
hipError_t launch_tripleclouds_prep(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevCloudPrep& prep,
                                    double* cc_sw, double* cc_lw, bool two_regions);
hipError_t launch_mcica_generator(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int ng, int seed_offset,
                                  double* od_scaling, double* tcc, int* counter);
hipError_t launch_mcica_generator_vec(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int ng, int seed_offset,
                                      double* od_scaling, double* tcc);
hipError_t launch_spectral_post(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevFlux& fx, bool wide);
hipError_t launch_optics_dump(bool is_sw, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                              const DevConfig& cfg, const DevInputs& in, const DevOptics& out, int g0, int* counter,
                              bool out_f32 = false, bool cloudy_only = false);

// SPARTACUS solvers (kernel_spartacus.hip): words of working precision of block-private slab per block and of layer
// matrices per (column, layer); the launch of one spectrum (work list, layer matrices, the two sweeps)
size_t spartacus_scratch_words(bool is_sw, int nlev);
size_t spartacus_layer_words(bool is_sw, int ngp);
int spartacus_sweep_blocks_per_cu(bool single, bool is_sw);
hipError_t launch_spartacus(bool is_sw, bool single, int ngp, int grid, int grid_layers, hipStream_t st, const ecrad_config_t& c,
                            const DevInputs& in, const DevOptics& op, const DevCloudPrep& prep, const DevFlux& fx, void* scratch,
                            size_t per_block_words, int* counter, const int32_t* d_i_band_from_reordered_g, void* lay, const uint32_t* list,
                            const int* item_of, const int* n_items, int g0, bool wide);
hipError_t launch_spartacus_list(hipStream_t st, const ecrad_config_t& c, const DevInputs& in, uint32_t* list, int* item_of, int* n_items);

// RRTMG gas optics (kernel_rrtmg.hip)
namespace rrtmg { struct DevRrtmg; }
size_t rrtmg_work_bytes(int nlev, int nloc);
RrtmgWork rrtmg_carve_work(void* base, int nlev, int nloc);
hipError_t launch_rrtmg_gas_optics(hipStream_t st, const rrtmg::DevRrtmg* tables, const DevConfig* cfg, const DevInputs& in,
                                   const RrtmgWork& w, const DevGasStage& out, bool do_lw, bool do_sw, const double* solar_scaling_host,
                                   hipStream_t st_sw, hipEvent_t ev_records, hipEvent_t ev_sw_done);

}  // namespace ecrad
