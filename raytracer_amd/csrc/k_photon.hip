// raytracer_amd/csrc/k_photon.hip -- GPU photon pass (placeholder until the grid kernels land).
#include "rt_ctx.h"

int rt_photon_emit(rt_hip_ctx*, int, int64_t, int64_t, float*, int64_t, int64_t*)
{
	rt_set_error("photon emission not built yet");
	return RT_HIP_ERR_STATE;
}
int rt_photon_build_grid(rt_hip_ctx*, const float*, int64_t)
{
	rt_set_error("photon grid not built yet");
	return RT_HIP_ERR_STATE;
}
int rt_photon_gather(rt_hip_ctx*, const float*, const float*, int, int, float, float*)
{
	rt_set_error("photon gather not built yet");
	return RT_HIP_ERR_STATE;
}
