// pde.hip - velocity PDE regulariser (placeholder until the kernels land)
#include "common.h"
extern "C" int nvfi_pde_workspace_bytes(const nvfi_field_desc* f, int64_t P, int64_t* bytes) { (void)f; (void)P; *bytes = 256; return 0; }
extern "C" int nvfi_pde_loss(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, float loss_scale, float* out,
                             const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream) {
    return nvfi_fail(9, "nvfi_pde_loss: not built yet");
}
