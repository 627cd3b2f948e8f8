// Instantiation unit of mlp_device.hpp: the forward kernels (plain and fused encode + MLP) for BF16 storage.
#include "mlp_device.hpp"

namespace perf {
void mlp_fwd_bf16(PERF_MLP_FWD_ARGS) { dispatch_fwd<BF16>(nh, ks, blocks, st, mp, w, feat, sel, out, n, n_dev); }
void mlp_fused_bf16(PERF_MLP_FUSED_ARGS) { dispatch_fused<BF16>(nh, ks, blocks, st, mp, w, sel, out, n, n_dev, fz); }
}  // namespace perf
