// Instantiation unit of mlp_device.hpp: the forward kernels (plain and fused encode + MLP) for FP16 storage.
#include "mlp_device.hpp"

namespace perf {
void mlp_fwd_fp16(PERF_MLP_FWD_ARGS) { dispatch_fwd<FP16>(nh, ks, blocks, st, mp, w, feat, sel, out, n, n_dev); }
void mlp_fused_fp16(PERF_MLP_FUSED_ARGS) { dispatch_fused<FP16>(nh, ks, blocks, st, mp, w, sel, out, n, n_dev, fz); }
}  // namespace perf
