"""tiny-cuda-nn's torch module surface, re-implemented on the gfx950 kernels.

PeRF constructs `tcnn.NetworkWithInputEncoding(n_input_dims, n_output_dims, encoding_config,
network_config)` (modules/fields/ngp_nerf.py:96-134,179-197,230-245) and
`tcnn.Encoding(n_input_dims, encoding_config)` (modules/geo_predictors/pano_joint_predictor.py:30-41,
pano_geo_refiner.py:19).  The contract kept here:
  * one nn.Parameter named `params`: flat fp32, [network weights | grid table] (tcnn layout: row-major
    [out,in] matrices, first-layer input width and output rows padded to 16) -- this is the tensor that
    lands in ckpt.pth through NGPNeRF.state_dict() and is handed to torch.optim.Adam;
  * forward(x [N, 3] float32 on the GPU, values in [0,1]) -> [N, n_output_dims] in the 16-bit compute
    dtype (tcnn returns half), differentiable w.r.t. params (and x for Encoding);
  * the fp32 -> 16-bit weight cast happens inside forward, cached on the parameter's version counter.
There is no CPU path: a CPU tensor raises.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .grid import GridConfig, MlpConfig

DEFAULT_DTYPE = 'bf16'     # BASELINE.json config 2 names bf16; 'fp16' reproduces tcnn's own precision
DEFAULT_SEED = 1337        # tcnn's torch binding seeds its init with 1337
# accumulation of the grid gradient: 'fp32' (LDS float atomics) or 'fixed' (packed 2x int32 fixed point, integer LDS
# atomics, per-level power-of-two unit from max|dfeat| under a closed-loop headroom; see hashgrid_bwd.hip).  This is the DEFAULT
# new modules start with (NetworkWithInputEncoding.grid_grad_accum); nothing in the process mutates it behind a module's back.
import os as _os
GRID_GRAD_ACCUM = _os.environ.get('PERF_GRID_GRAD_ACCUM', 'fixed')


def _init_params(mlp: MlpConfig, grid: GridConfig, seed: int, device=None) -> torch.Tensor:
    """tcnn's initialisation: Xavier-uniform network weights, U(-1e-4, 1e-4) table entries, [network | grid].  On a GPU the
    table (millions of entries; the network is ~10 k) is drawn by the DEVICE generator seeded with `seed` -- tcnn initialises on
    the device too, and PeRF re-instantiates the density field every episode (reset_geo, nerf.py:136-141): the host draw +
    copy cost 65 ms of a 1.1 s episode.  Same seed -> same values on every run and every rank."""
    # (every host-side draw names its device: PeRF's runner makes CUDA the default tensor type, core_exp_runner.py:266, and a
    #  device-less torch.rand would then be handed this CPU generator)
    g = torch.Generator(device='cpu').manual_seed(seed)
    parts = []
    if mlp is not None:
        for (o, i) in mlp.shapes:
            s = math.sqrt(6.0 / (i + o))                 # Xavier uniform
            parts.append((torch.rand(o * i, generator=g, device='cpu') * 2 - 1) * s)
    device = torch.device(device) if device is not None else torch.device('cpu')
    if device.type != 'cuda':
        parts.append((torch.rand(grid.n_params, generator=g, device='cpu') * 2 - 1) * 1e-4)
        return torch.cat(parts).to(device)
    n_net = sum(p.numel() for p in parts)
    out = torch.empty(n_net + grid.n_params, dtype=torch.float32, device=device)
    if n_net:
        out[:n_net] = torch.cat(parts).to(device)
    out[n_net:].uniform_(-1e-4, 1e-4, generator=torch.Generator(device=device).manual_seed(seed))
    return out


class _FieldFn(torch.autograd.Function):
    """encode + MLP with the whole backward in two kernels (MLP backward recomputes the forward)."""

    @staticmethod
    def forward(ctx, x01, params, sel, module, n_dev=None):
        """n_dev (device int64 [1], optional): only the first min(len(x01), n_dev) rows are live (capacity-sized batch)."""
        w16 = module.working_copy(params)
        n_net = module.mlp.n_params
        # (inference -- no_grad, or frozen parameters -- never gets here: field_apply() below routes it to perf_field_infer;
        #  ctx.needs_input_grad cannot tell, it reports params.requires_grad even under torch.no_grad())
        feat = ops.hashgrid_fwd(module.grid, x01, w16[n_net:], n_dev=n_dev)
        out = ops.mlp_fwd(module.mlp, w16[:n_net], feat, sel, n_dev=n_dev)
        ctx.module = module
        ctx.n_dev = n_dev
        ctx.save_for_backward(x01, w16, feat, sel if sel is not None else torch.empty(0, device=x01.device))
        ctx.has_sel = sel is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x01, w16, feat, sel = ctx.saved_tensors
        module = ctx.module
        sel = sel if ctx.has_sel else None
        return None, _field_backward(module, x01, w16, feat, sel, dout, n_dev=ctx.n_dev, clear_flag=True), None, None, None


def field_apply(module, x01, params, sel=None, n_dev=None):
    """act(MLP(encode(x01))) * sel of one network: ONE boundary call that keeps nothing when no gradient can be asked for
    (torch.no_grad(), or frozen parameters), the autograd Function otherwise."""
    if not (torch.is_grad_enabled() and params.requires_grad):
        return ops.field_infer(module.grid, module.mlp, x01, sel, module.working_copy(params), n_dev=n_dev)
    return _FieldFn.apply(x01, params, sel, module, n_dev)


def _field_backward(module, x01, w16, feat, sel, dout, n_dev=None, clear_flag=False):
    """Flat gradient [network | grid] of one field.  Fixed-point accumulation never costs a step: should a field of the
    grid gradient near the int32 range (device flag), the predicated repair launch right behind the backward rewrites the
    table gradient with fp32 LDS accumulation -- a no-op dispatch otherwise; the host is not asked (no read-back per
    backward, capturable).  clear_flag: the autograd (shim) path has no perf_step_bookkeeping behind it to consume the flag."""
    n_net = module.mlp.n_params
    fixed = module.grid_grad_accum == 'fixed'
    if not fixed or module.redo_supported:
        # ONE boundary call: MLP backward -> grid backward -> predicated repair launch (perf_field_bwd), cached workspace
        grad = ops.field_bwd(module.grid, module.mlp, x01, w16[:n_net], feat, dout.contiguous().float(), sel, fixed=fixed, redo=True,
                             hr_state=module.headroom_state() if fixed else None, n_dev=n_dev)
        if fixed and clear_flag:
            ops.overflow_flag(x01.device).zero_()
        return grad
    grad = torch.empty(n_net + module.grid.n_params, dtype=torch.float32, device=x01.device)
    res = ops.mlp_bwd(module.mlp, w16[:n_net], feat, dout.contiguous().float(), sel, want_absmax=fixed, n_dev=n_dev, dw_out=grad[:n_net])
    ops.hashgrid_bwd_into(module.grid, x01, res[0], grad[n_net:], level_absmax=res[2] if fixed else None, n_dev=n_dev,
                          hr_state=module.headroom_state() if fixed else None)
    if fixed:
        if module.redo_supported:
            ops.hashgrid_bwd_redo(module.grid, x01, res[0], grad[n_net:], n_dev=n_dev, hr_state=module.headroom_state())
            if clear_flag:
                ops.overflow_flag(x01.device).zero_()
        elif clear_flag and not torch.cuda.is_current_stream_capturing():
            # (grids with levels beyond LDS owners -- BASELINE config 5 -- have no repair launch: ask the host, redo in fp32)
            flag = ops.overflow_flag(x01.device)
            if bool(int(flag.item())):
                flag.zero_()
                ops.hashgrid_bwd_into(module.grid, x01, res[0], grad[n_net:], level_absmax=None, n_dev=n_dev)
    return grad


class _DualFieldFn(torch.autograd.Function):
    """Two fields with the same grid geometry (PeRF's geo and app networks) at the same points: one encode pass
    shares the corner indices, then each network's MLP.  Returns (out_a, out_b)."""

    @staticmethod
    def forward(ctx, x01, params_a, params_b, sel, mod_a, mod_b):
        ctx.set_materialize_grads(False)
        wa, wb = mod_a.working_copy(params_a), mod_b.working_copy(params_b)
        na, nb = mod_a.mlp.n_params, mod_b.mlp.n_params
        fa, fb = ops.hashgrid_fwd2(mod_a.grid, x01, wa[na:], wb[nb:])
        out_a = ops.mlp_fwd(mod_a.mlp, wa[:na], fa, sel)
        out_b = ops.mlp_fwd(mod_b.mlp, wb[:nb], fb, sel)
        ctx.mods = (mod_a, mod_b)
        ctx.has_sel = sel is not None
        ctx.save_for_backward(x01, wa, wb, fa, fb, sel if sel is not None else torch.empty(0, device=x01.device))
        return out_a, out_b

    @staticmethod
    def backward(ctx, da, db):
        x01, wa, wb, fa, fb, sel = ctx.saved_tensors
        sel = sel if ctx.has_sel else None
        ga = gb = None
        if ctx.needs_input_grad[1] and da is not None:
            ga = _field_backward(ctx.mods[0], x01, wa, fa, sel, da, clear_flag=True)
        if ctx.needs_input_grad[2] and db is not None:
            gb = _field_backward(ctx.mods[1], x01, wb, fb, sel, db, clear_flag=True)
        return None, ga, gb, None, None, None


class NetworkWithInputEncoding(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=DEFAULT_SEED, dtype=None):
        super().__init__()
        if n_input_dims != 3:
            raise ValueError('the gfx950 hash grid is built for 3 input dims')
        otype = network_config.get('otype', 'FullyFusedMLP')
        if otype not in ('FullyFusedMLP', 'CutlassMLP'):
            raise ValueError(f'unsupported network otype {otype!r}')
        if network_config.get('activation', 'ReLU') != 'ReLU':
            raise ValueError('only ReLU hidden activation is supported')
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.encoding_config = dict(encoding_config)
        self.network_config = dict(network_config)
        self.seed = seed
        self.dtype_name = dtype or DEFAULT_DTYPE
        self.grid = GridConfig.from_tcnn(encoding_config)
        self.mlp = MlpConfig(n_levels=self.grid.n_levels,
                             n_hidden_layers=int(network_config.get('n_hidden_layers', 1)),
                             n_output_dims=n_output_dims,
                             output_activation=network_config.get('output_activation', 'None'),
                             n_neurons=int(network_config.get('n_neurons', 64)))
        self.params = nn.Parameter(_init_params(self.mlp, self.grid, seed, _default_device()))
        self._w16 = None
        self._w16_key = None
        self._hr_state = ops.headroom_state(self.params.device)     # (plain attribute; NeRFScene.state_dict() carries it)
        self.grid_grad_accum = GRID_GRAD_ACCUM                      # this module's accumulation mode ('fixed' | 'fp32')
        self.redo_supported = ops.hashgrid_bwd_redo_supported(self.grid)

    # -- 16-bit working copy, refreshed when the fp32 master changes -------------------------------
    def working_copy(self, params=None):
        p = self.params if params is None else params
        key = (p.data_ptr(), p._version, self.dtype_name)
        if self._w16 is None or self._w16_key != key:
            self._w16 = ops.cast_params(p.detach(), self.dtype_name, self._w16 if (self._w16 is not None and self._w16.dtype == ops.torch_dtype(self.dtype_name) and self._w16.numel() == p.numel()) else None)
            self._w16_key = key
        return self._w16

    def headroom_state(self):
        """Device state of the fixed-point headroom feedback of this network's grid gradient (perf_hashgrid_bwd)."""
        st = getattr(self, '_hr_state', None)
        if st is None or st.device != self.params.device:
            st = self._hr_state = ops.headroom_state(self.params.device)
        return st

    def fp32_redo_count(self):
        """How often the repair launch of the fixed-point grid backward really ran (one host read-back; diagnostics)."""
        return int(self.headroom_state()[2 * ops._lib.MAX_LEVELS + 1].item())

    def set_working_copy(self, w16):
        """Adopt a working copy written by the fused Adam kernel (perf_adam_step)."""
        self._w16 = w16
        self._w16_key = (self.params.data_ptr(), self.params._version, self.dtype_name)

    def forward(self, x, selector=None, out_fp32=False):
        x = x.reshape(-1, self.n_input_dims).contiguous().float()
        out = field_apply(self, x, self.params, selector)
        return out if out_fp32 else out.to(ops.torch_dtype(self.dtype_name))

    def extra_repr(self):
        return f'n_params={self.params.numel()}, grid={self.grid.n_levels}x2 T={self.grid.log2_hashmap_size}, ' \
               f'mlp={self.mlp.n_hidden_layers}x64->{self.n_output_dims}, dtype={self.dtype_name}'


class _EncodingFn(torch.autograd.Function):
    """y = encode(x; table).  Its backward calls _EncodingInputGradFn, itself an autograd Function, so that
    autograd.grad(y, x, create_graph=True) can be differentiated again (SphereDistanceField, pano_joint_predictor.py:64-67)."""

    @staticmethod
    def forward(ctx, x01, params, module):
        feat = ops.hashgrid_fwd_f32(module.grid, x01, params.detach())
        ctx.module = module
        ctx.save_for_backward(x01, params)
        return feat.permute(1, 0, 2).reshape(x01.shape[0], -1)

    @staticmethod
    def backward(ctx, dout):
        x01, params = ctx.saved_tensors
        module = ctx.module
        gx = gp = None
        if ctx.needs_input_grad[0]:
            gx = _EncodingInputGradFn.apply(x01, params, dout, module)
        if ctx.needs_input_grad[1]:
            gp = _EncodingParamGradFn.apply(x01, dout, module)
        return gx, gp, None


def _level_major(dout, module):
    n = dout.shape[0]
    return dout.float().reshape(n, module.grid.n_levels, 2).permute(1, 0, 2).contiguous()


class _EncodingParamGradFn(torch.autograd.Function):
    """d y / d table contracted with dout: linear in dout, so its own backward (w.r.t. dout and x) exists in principle; no
    PeRF consumer differentiates the TABLE gradient again, and asking for it raises instead of returning zeros."""

    @staticmethod
    def forward(ctx, x01, dout, module):
        return ops.hashgrid_bwd(module.grid, x01.detach(), _level_major(dout.detach(), module))

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError('second derivatives THROUGH the table gradient of tcnn.Encoding are not implemented '
                                  '(no consumer in PeRF; the input-gradient path is: _EncodingInputGradFn)')


class _EncodingInputGradFn(torch.autograd.Function):
    """gx = d(y . dout)/dx (kernel perf_hashgrid_bwd_input) with a kernel backward: given ggx = dL/d gx,
    d_x (Hessian-vector product of the interpolation weights), d_table (scatter) and d_dout (gather) come from
    perf_hashgrid_bwd_bwd_input / perf_hashgrid_bwd_bwd_param."""

    @staticmethod
    def forward(ctx, x01, params, dout, module):
        dfeat = _level_major(dout.detach(), module)
        ctx.module = module
        ctx.save_for_backward(x01, params, dfeat)
        return ops.hashgrid_bwd_input(module.grid, x01.detach(), dfeat, params.detach())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ggx):
        x01, params, dfeat = ctx.saved_tensors
        module = ctx.module
        n = x01.shape[0]
        ggx = ggx.contiguous().float()
        want_x, want_p, want_d = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        d_x = d_p = d_dout = None
        if want_x or want_d:
            dd, d_x = ops.hashgrid_bwd_bwd_input(module.grid, x01.detach(), dfeat, params.detach(), ggx, want_ddfeat=want_d, want_dx=want_x)
            if want_d:
                d_dout = dd.permute(1, 0, 2).reshape(n, -1)
        if want_p:
            d_p = ops.hashgrid_bwd_bwd_param(module.grid, x01.detach(), dfeat, ggx)
        return d_x, d_p, d_dout, None


class Encoding(nn.Module):
    """tcnn.Encoding (HashGrid, Linear or Smoothstep interpolation), fp32 table, first-order autograd
    w.r.t. the table and the input."""

    def __init__(self, n_input_dims, encoding_config, seed=DEFAULT_SEED, dtype=None):
        super().__init__()
        if n_input_dims != 3:
            raise ValueError('the gfx950 hash grid is built for 3 input dims')
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        self.grid = GridConfig.from_tcnn(encoding_config)
        self.n_output_dims = self.grid.n_output_dims
        # tcnn hands back half by default; dtype=torch.float32 / 'fp32' keeps the fp32 result
        self.out_dtype = torch.float32 if dtype in ('fp32', torch.float32) else ops.torch_dtype(dtype or 'fp16')
        self.params = nn.Parameter(_init_params(None, self.grid, seed, _default_device()))

    def forward(self, x):
        x = x.reshape(-1, self.n_input_dims).contiguous().float()
        # First order: kernels.  Second order (the caller differentiates the input gradient again: SphereDistanceField,
        # autograd.grad(..., create_graph=True), pano_joint_predictor.py:64-67): kernels too -- see _EncodingInputGradFn.
        return _EncodingFn.apply(x, self.params, self).to(self.out_dtype)

    def _forward_composed(self, x):
        """TEST REFERENCE ONLY: the same encoding composed from differentiable torch ops on top of the corner-index kernel
        (every order of derivative exists through autograd); tests compare the kernel double backward with it."""
        g = self.grid
        n = x.shape[0]
        idx = ops.hashgrid_corners(g, x.detach()).long()                       # [L, n, 8]
        scale = torch.as_tensor(g.scale, device=x.device)[:, None, None]       # [L, 1, 1]
        pos = x[None] * scale + 0.5                                            # [L, n, 3]
        # value of the kernels' single-rounding fmaf(x, scale, 0.5) (through float64), gradient of the plain expression
        pos = pos + ((x.detach()[None].double() * scale.double() + 0.5).float() - pos).detach()
        f = pos - torch.floor(pos).detach()
        if g.interpolation == 'Smoothstep':
            f = f * f * (3.0 - 2.0 * f)
        ws = []
        for c in range(8):
            wx = f[..., 0] if (c & 1) else 1.0 - f[..., 0]
            wy = f[..., 1] if (c & 2) else 1.0 - f[..., 1]
            wz = f[..., 2] if (c & 4) else 1.0 - f[..., 2]
            ws.append((wx * wy) * wz)
        w = torch.stack(ws, -1)                                                # [L, n, 8]
        vals = self.params.view(-1, 2)[idx.reshape(-1)].view(g.n_levels, n, 8, 2)
        feat = (w[..., None] * vals).sum(2)                                    # [L, n, 2]
        return feat.permute(1, 0, 2).reshape(n, -1)


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('perf_amd.tcnn needs a HIP device (there is no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())
