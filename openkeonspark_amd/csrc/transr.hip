// TransR (per-relation projection matrices, TransR.py:16-87): relation-bucketed MFMA path.
#include "engine.hpp"

namespace kge {

int launch_forward_backward_transr(const kge_model_desc &, const float *const[4], const int32_t *, const int32_t *,
                                   const int32_t *, int64_t, int64_t, int64_t, int64_t, float *const[4], float *,
                                   hipStream_t) {
    return fail(KGE_ERR_UNSUPPORTED, "TransR forward/backward is not built yet");
}

int launch_predict_transr(const kge_model_desc &, const float *const[4], const int32_t *, const int32_t *, const int32_t *,
                          int64_t, float *, hipStream_t) {
    return fail(KGE_ERR_UNSUPPORTED, "TransR predict is not built yet");
}

}  // namespace kge
