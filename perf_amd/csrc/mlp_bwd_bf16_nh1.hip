// Instantiation unit of mlp_device.hpp: the backward kernels of the one-hidden-layer network for BF16 storage.
#include "mlp_device.hpp"

namespace perf {
void mlp_bwd_bf16_nh1(PERF_MLP_BWD_ARGS) {
    dispatch_bwd_nh<BF16, 1>(ks, blocks, st, mp, w, feat, feat_index, feat_stride, sel, dout, dfeat, partials, level_absmax, n, n_dev);
}
}  // namespace perf
