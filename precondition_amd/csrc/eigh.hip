// eigh.hip — batched inverse p-th root by symmetric eigendecomposition
// (reference: matrix_inverse_pth_root_eigh, DS:943-1030).  Not built yet in this
// round: the entry points exist so that the ABI is complete and fail loudly.
#include "common.h"

extern "C" size_t ps_eigh_root_workspace_bytes(int batch, const int32_t* n) {
  (void)batch; (void)n;
  return 0;
}

extern "C" int ps_eigh_root_batched_f32(void* stream, const float* const* a,
                                        const int32_t* n, const int32_t* lda,
                                        const int32_t* p, const int32_t* padding_start,
                                        int batch, float ridge_epsilon,
                                        float error_tolerance, int relative_matrix_epsilon,
                                        float* const* out, const int32_t* ldo,
                                        float* metrics, void* workspace,
                                        size_t workspace_bytes) {
  (void)stream; (void)a; (void)n; (void)lda; (void)p; (void)padding_start; (void)batch;
  (void)ridge_epsilon; (void)error_tolerance; (void)relative_matrix_epsilon; (void)out;
  (void)ldo; (void)metrics; (void)workspace; (void)workspace_bytes;
  return PS_EUNSUPPORTED;
}
