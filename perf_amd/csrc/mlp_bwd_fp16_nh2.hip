// Instantiation unit of mlp_device.hpp: the backward kernels of the two-hidden-layer network for FP16 storage.
#include "mlp_device.hpp"

namespace perf {
void mlp_bwd_fp16_nh2(PERF_MLP_BWD_ARGS) {
    dispatch_bwd_nh<FP16, 2>(ks, blocks, st, mp, w, feat, feat_index, feat_stride, sel, dout, dfeat, partials, level_absmax, n, n_dev);
}
}  // namespace perf
