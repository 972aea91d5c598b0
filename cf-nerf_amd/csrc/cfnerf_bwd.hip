// cfnerf_bwd.hip - loss, backward and optimiser kernels (placeholder until the backward lands)
#include "cfnerf_kernels.h"
#include "cfnerf_model.h"

extern "C" {
int cfnerf_loss_fwd_bwd(const float*, const float*, const float*, int64_t, int, float, int64_t, float*, float*, cfnerf_stream) { return CFNERF_E_UNSUPPORTED; }
int cfnerf_render_bwd(cfnerf_model*, const float*, const float*, const float*, float*, cfnerf_stream) { return CFNERF_E_UNSUPPORTED; }
int cfnerf_adam_step(cfnerf_model*, float*, const float*, float*, float*, int64_t, float, float, cfnerf_stream) { return CFNERF_E_UNSUPPORTED; }
}
