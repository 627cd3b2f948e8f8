"""Host-side mirror of PeRF's training/rendering scene (modules/scene/nerf.py:28-396) and of the two
supervision-pool functions on the hot path (modules/dataset/sup_info.py:236-259, 304-330).

Same names and semantics as the reference so that parity tests read like its code:
NeRFScene.render / render_once / fit / train_one_episode / train_one_step_geo / train_one_step_app / update_lr /
state_dict / load_state_dict / set_train / set_eval, SupInfoPool.rand_ray_color_data / gen_occ_grid.

Reference quirks that are reproduced on purpose:
  * the GradScaler(2**7) is only used for .scale(): Adam sees gradients multiplied by 128 (nerf.py:139,252-253);
  * geo `progress` is iter_i / app_res_iters (nerf.py:178), distortion ratio min(2*progress, 1) (:235);
  * the occupancy grid is static during an episode; the 256 warm-up calls of :160-168 leave binaries == pre_grid
    (up to a handful of cells from fp32 jitter), so by default binaries := pre_grid; warmup='reference' replays
    the 256 calls instead.
Data parallelism (not in the reference): every rank draws the SAME global index stream and takes
its slice; the local loss is pre-scaled so the summed gradient equals the 1-GPU gradient; one RCCL all-reduce of
the active network's flat gradient per step.
"""
import math
import os
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .distloss import flatten_eff_distloss
from .fields import NGPNeRF
from .nerfacc_impl import OccGridEstimator
from .renderer import NeRFOCCRenderer
from . import tcnn as _tcnn

_DP_GRAPH_VERDICT = None

def _capture(graph):
    """torch.cuda.graph(graph) -- thread-local capture mode once a process group exists: ProcessGroupNCCL's watchdog thread
    polls the events of EARLIER collectives (hipEventQuery) while this thread captures; under the default global mode that
    poll fails with hipErrorStreamCaptureUnsupported and takes the process down (seen once in five runs of
    tests/test_gpu_dist.py::test_rccl_exchange_on_a_world_of_one).  Thread-local mode checks only the capturing thread."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        torch.cuda.synchronize()                 # nothing of an earlier collective is left for the watchdog to wait for
        if dist.get_backend() == 'nccl':
            # ... but the watchdog only FORGETS a finished collective at its next poll (every 100 ms).  Until then it keeps
            # querying that collective's end event -- recorded on RCCL's stream, which joins the capture with the first captured
            # collective; HIP then answers hipErrorCapturedEvent ("event last recorded in a capturing stream") and the watchdog
            # takes the process down (1 in ~10 runs of tests/test_gpu_dist.py; never when the capture came > 100 ms after the
            # last eager collective).  Captures are rare (one per phase): give the watchdog three polls.
            import time
            time.sleep(float(os.environ.get('PERF_DP_CAPTURE_DRAIN_S', '0.3')))
        return torch.cuda.graph(graph, capture_error_mode='thread_local')
    return torch.cuda.graph(graph)


def _graph_node_count(graph):
    """Nodes (kernel launches, copies, fills) of a captured hipGraph -- what one replay issues.  Needs a graph created with
    keep_graph=True; asks the HIP runtime this process already loaded (hipGraphGetNodes).  None when unavailable."""
    try:
        import ctypes
        raw = graph.raw_cuda_graph()
        path = None
        for line in open('/proc/self/maps'):
            if 'libamdhip64' in line:
                path = line.split()[-1]
                break
        if path is None:
            return None
        hip = ctypes.CDLL(path)
        n = ctypes.c_size_t(0)
        rc = hip.hipGraphGetNodes(ctypes.c_void_p(int(raw)), None, ctypes.byref(n))
        return int(n.value) if rc == 0 else None
    except Exception:      # noqa: BLE001 -- a statistic, never a failure
        return None


OVERFLOW_CHECK_EVERY = 64      # training steps between reads of the fixed-point overflow flag (one host sync each)


class _Losses(dict):
    """last_losses of a scene: an entry may be a thunk (the fused steps leave per-ray loss terms on the device and sum them
    only when somebody reads the value); reading evaluates it."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return v() if callable(v) else v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def resolved(self):
        """A plain dict of evaluated values (each thunk costs a device reduction; float() of a value a host read-back).
        dict(losses) / {**losses} / json.dumps(losses) take dict's C fast path and would see the raw thunks: use this, .copy()
        or pickling instead."""
        return {k: self[k] for k in self.keys()}

    def copy(self):
        return self.resolved()

    def __reduce__(self):
        return (dict, (self.resolved(),))


@dataclass
class Rays:
    """utils/camera_utils.py:9-21"""
    o: torch.Tensor
    d: torch.Tensor

    def __len__(self):
        return len(self.o)

    def __getitem__(self, idx):
        return Rays(self.o[idx], self.d[idx])

    def collapse(self):
        return self.o, self.d


def gen_pano_rays(pose, height=512, width=1024, device='cuda'):
    """utils/camera_utils.py:229-234, generated in one kernel."""
    o, d = ops.pano_raygen(pose, height, width, device=device)
    return Rays(o, d)


def default_train_conf():
    """configs/nerf.yaml:24-74"""
    opt = lambda i, p, a, l: SimpleNamespace(init_lr=i, peak_lr=p, peak_at=a, lr_alpha=l)
    return SimpleNamespace(
        raw_phase_iter_geo=3000, raw_phase_iter_app=1500,
        geo_optimizer=opt(0.0, 1e-2, 0.2, 1e-2), app_optimizer=opt(0.0, 1e-2, 0.2, 1e-2),
        color_loss_weight=1., depth_loss_weight=1., density_loss_weight=0., distortion_loss_weight=0.1,
        pixel_loss_batch_size=8192)


def _edge_free(distance_map):
    """PanoSupInfo's depth-edge test (sup_info.py:77-82): kornia.filters.laplacian(kernel 3, reflect border, normalised:
    [[1,1,1],[1,-8,1],[1,1,1]] / 16), |.| < 0.01, erosion then dilation with a 3x3 box.  [H,W,1] -> bool [H,W,1]."""
    from . import visibility as V
    x = distance_map.permute(2, 0, 1)[None]
    k = torch.ones(3, 3, device=x.device); k[1, 1] = -8.0
    lap = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), (k / 16.0)[None, None])
    e = (lap.abs() < 0.01).float()
    box = torch.ones(3, 3, device=x.device)
    e = V.dilate(V.erode(e, box), box)
    return (e[0] > .5).permute(1, 2, 0)


class SupInfoPool:
    """Supervision rays of all registered panoramas (sup_info.py:141-259, 304-330): flat device tensors."""

    def __init__(self):
        self.all_sup_rays = None
        self.all_sup_colors = None
        self.all_sup_distances = None
        self.all_sup_normals = None
        self.n_panos = 0
        self.sup_infos = []            # per panorama: pose, distance_map [H,W,1], mask [H,W,1] (visibility tests)
        self._ranges = []              # per registration: [start, end) in the flat arrays (rand_mode 'only_first' / 'only_last')

    def register_rays(self, rays_o, rays_d, colors, distances, normals=None):
        o = rays_o.reshape(-1, 3).contiguous().float(); d = rays_d.reshape(-1, 3).contiguous().float()
        c = colors.reshape(-1, 3).contiguous().float(); t = distances.reshape(-1, 1).contiguous().float()
        n = torch.zeros_like(c) if normals is None else normals.reshape(-1, 3).contiguous().float()
        if self.all_sup_rays is None:
            self.all_sup_rays, self.all_sup_colors, self.all_sup_distances, self.all_sup_normals = Rays(o, d), c, t, n
        else:
            self.all_sup_rays = Rays(torch.cat([self.all_sup_rays.o, o]), torch.cat([self.all_sup_rays.d, d]))
            self.all_sup_colors = torch.cat([self.all_sup_colors, c])
            self.all_sup_distances = torch.cat([self.all_sup_distances, t])
            self.all_sup_normals = torch.cat([self.all_sup_normals, n])
        self._ranges.append((len(self) - len(c), len(self)))
        self.n_panos += 1

    def register_sup_info(self, pose, mask, rgb, distance, normal=None):
        """Panorama [H,W,*] maps -> supervision rays of its valid pixels, with PanoSupInfo's validity rules
        (sup_info.py:27-120): mask > 0.5 and distance > 1e-5; no depth edge (|normalised 3x3 Laplacian of the distance map|
        < 0.01, eroded then dilated by a 3x3 box); with a normal map, surfaces seen at cos > 0.15."""
        h, w, _ = rgb.shape
        dev = rgb.device
        pose = torch.as_tensor(pose, dtype=torch.float32, device=dev)
        distance = (torch.ones(h, w, 1, device=dev) if distance is None else distance.reshape(h, w, 1)).float()
        has_normal = normal is not None
        normal = normal.reshape(h, w, 3).float() if has_normal else torch.zeros(h, w, 3, device=dev)
        mask_raw = (mask.reshape(h, w, 1) > .5) & (distance > 1e-5)
        valid = mask_raw & _edge_free(distance)
        if has_normal:
            local = gen_pano_rays(torch.eye(4, device='cpu'), h, w, device=dev)
            valid = valid & (((-local.d) * normal).sum(-1, True).clip(0., 1.) > 0.15)
        rays = gen_pano_rays(pose, h, w, device=dev)
        idx = torch.where(valid[..., 0])
        self.sup_infos.append({'pose': pose, 'height': h, 'width': w, 'mask_raw': mask_raw, 'color_map': rgb.float(),
                               'distance_map': distance, 'normal_map': normal, 'mask': valid,
                               'sup_colors': rgb[idx].float(), 'sup_distances': distance[idx], 'sup_normals': normal[idx],
                               'sup_dirs': rays.d[idx], 'sup_positions': rays.o[idx]})
        self.register_rays(rays.o[idx], rays.d[idx], rgb[idx], distance[idx], normal[idx] if has_normal else None)

    def geo_check(self, rays, distances):
        """sup_info.py:261-302: 1 = consistent with every registered panorama, 0 = conflict (perf_amd/visibility.py)."""
        from .visibility import geo_check
        return geo_check(rays.o, rays.d, distances, self.sup_infos)

    _INFO_KEYS = ('pose', 'mask_raw', 'color_map', 'distance_map', 'normal_map', 'mask', 'sup_colors', 'sup_distances',
                  'sup_normals', 'sup_dirs', 'sup_positions')         # PanoSupInfo's buffers, in registration order

    def state_dict(self):
        """sup_info.py:332-340, key for key (including its unformatted '{}' height / width keys)."""
        ret = {'n_sup_infos': len(self.sup_infos)}
        for i, info in enumerate(self.sup_infos):
            ret['sup_info_{}_height'] = info['height']
            ret['sup_info_{}_width'] = info['width']
            ret['sup_info_{}'.format(i)] = {k: info[k] for k in self._INFO_KEYS}
        return ret

    def load_state_dict(self, state_dict):
        """Restores what the reference's loader intends (sup_info.py:342-359 rebuilds placeholder PanoSupInfo objects and
        never copies the saved buffers into them): every panorama's maps and supervision rays."""
        self.__init__()
        for i in range(state_dict['n_sup_infos']):
            info = dict(state_dict['sup_info_{}'.format(i)])
            info['height'], info['width'] = info['color_map'].shape[:2]
            self.sup_infos.append(info)
            self.register_rays(info['sup_positions'], info['sup_dirs'], info['sup_colors'], info['sup_distances'], info['sup_normals'])

    def __len__(self):
        return 0 if self.all_sup_colors is None else len(self.all_sup_colors)

    def rand_ray_color_data(self, batch_size, rand_mode='by_all_pixels', generator=None, rank=0, world_size=1):
        """sup_info.py:236-259.  With world_size > 1 every rank draws the same `batch_size` indices and keeps
        the contiguous slice [rank*b/W, (rank+1)*b/W) -- the union is the 1-GPU batch."""
        assert rand_mode in ['by_all_pixels', 'only_first', 'only_last']
        start, end = (0, len(self)) if rand_mode == 'by_all_pixels' else self._ranges[0 if rand_mode == 'only_first' else -1]
        indices = torch.randint(0, end - start, (batch_size,), device=self.all_sup_colors.device, generator=generator)
        if start:
            indices = indices + start
        if world_size > 1:
            per = batch_size // world_size
            indices = indices[rank * per:(rank + 1) * per]
        if not indices.is_cuda:
            return (self.all_sup_rays[indices], self.all_sup_colors[indices], self.all_sup_distances[indices],
                    self.all_sup_normals[indices])
        g = ops.gather_supervision(indices, self.all_sup_rays.o, self.all_sup_rays.d, self.all_sup_colors,
                                   self.all_sup_distances, self.all_sup_normals)            # one launch instead of five
        return Rays(g['o'], g['d']), g['color'], g['dist'], g['normal']

    def draw_batch(self, batch_size, seed, counter, rand_mode='by_all_pixels', rank=0, world_size=1, want_bg=False):
        """rand_ray_color_data (sup_info.py:236-259) AND the per-ray uniform draws of a training step (stratified jitter, distance
        noise, background colour: nerf_renderer.py:152,185,193) in one launch from the counter-based device generator
        (perf_draw_train_batch).  `counter` (device int64 [1]) numbers the draws and is advanced by the launch; with
        world_size > 1 every rank draws its slice [rank*b/W, (rank+1)*b/W) of the same global batch (same seed, same counter).
        -> (Rays, colors, distances, normals, {'jitter', 'noise', 'bg'})."""
        assert rand_mode in ['by_all_pixels', 'only_first', 'only_last']
        start, end = (0, len(self)) if rand_mode == 'by_all_pixels' else self._ranges[0 if rand_mode == 'only_first' else -1]
        per = batch_size // world_size
        g = ops.draw_train_batch(seed, counter, start, end, per, rank * per, self.all_sup_rays.o, self.all_sup_rays.d, self.all_sup_colors,
                                 self.all_sup_distances, self.all_sup_normals, want_bg=want_bg)
        return Rays(g['o'], g['d']), g['color'], g['dist'], g['normal'], {'jitter': g['jitter'], 'noise': g['noise'], 'bg': g['bg']}

    def gen_occ_grid(self, res):
        """sup_info.py:304-330 as one splat kernel.  Returns (occ uint8 [res^3], points of occupied cells)."""
        rays_o, rays_d = self.all_sup_rays.collapse()
        occ = ops.occ_splat(rays_o, rays_d, self.all_sup_distances, res)
        valid_idx = torch.where(occ > 0)[0]
        pts = torch.stack([valid_idx // (res * res), (valid_idx // res) % res, valid_idx % res], -1)
        return occ, (pts / float(res) - .5) * 2.


class FusedAdam:
    """torch.optim.Adam(params, lr) for ONE flat fp32 parameter, as a single kernel that also refreshes the
    network's 16-bit working copy (perf_adam_step_dev).  Step count and learning rate live in device scalars so
    that a captured hipGraph of the whole training step can be replayed while the schedule advances."""

    def __init__(self, net, lr, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        self.param_groups = [{'lr': lr, 'betas': betas, 'eps': eps}]
        p = net.params
        self.exp_avg = torch.zeros_like(p.data)
        self.exp_avg_sq = torch.zeros_like(p.data)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=p.device)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=p.device)
        self.eff_gate = torch.zeros(1, dtype=torch.int64, device=p.device)
        self.capturing = False
        self.w16 = torch.empty(p.numel(), dtype=ops.torch_dtype(net.dtype_name), device=p.device)
        self.w16.copy_(net.working_copy())         # (a gated-off first step must leave a valid working copy behind)
        self.sched_table = None                    # device-side schedule of graph-replayed phases (perf_step_bookkeeping)
        self.sched_iter = None
        self.sched_ratio_out = None

    @property
    def p(self):
        return self.net.params.data

    @property
    def step_count(self):
        return int(self.step_dev.item())

    def zero_grad(self):
        pass                                       # the step consumes the gradient and drops it

    def refresh_lr(self):
        if not self.capturing and self.sched_table is None:
            self.lr_dev.fill_(self.param_groups[0]['lr'])      # under capture the replay wrapper refreshes lr_dev instead

    SCHEDULE_ROWS = 32768

    def load_schedule(self, lrs, ratios=None, first=0, ratio_out=None):
        """Install the device-side schedule: row i = (learning rate, distortion-loss ramp) of iteration i; the next step is
        iteration `first`.  From then on perf_step_bookkeeping sets lr_dev (and ratio_out, the loss head's ramp scalar) itself:
        replaying a captured step needs no host-side scalar update.  The table has a fixed size (the captured launch keeps its
        pointers): a new phase just loads new rows.  Rows beyond the given ones repeat the last one."""
        if len(lrs) < 1:
            raise ValueError('load_schedule: empty schedule')
        dev = self.lr_dev.device
        if self.sched_table is None:                  # sized for the first phase it serves (a captured launch keeps its pointer)
            self.sched_table = torch.empty(max(self.SCHEDULE_ROWS, len(lrs)), 2, dtype=torch.float32, device=dev)
            self.sched_iter = torch.zeros(1, dtype=torch.int32, device=dev)
        total = self.sched_table.shape[0]
        if len(lrs) > total:
            import warnings
            warnings.warn(f'perf_amd: a phase of {len(lrs)} iterations exceeds the device-side schedule of {total} rows; '
                          'iterations beyond it keep the last row (learning rate / distortion ramp frozen)')
        n = min(len(lrs), total)
        rows = torch.empty(total, 2, dtype=torch.float32, device='cpu')
        rows[:n, 0] = torch.as_tensor(lrs[:n], dtype=torch.float32, device='cpu')
        rows[:n, 1] = torch.as_tensor(ratios[:n], dtype=torch.float32, device='cpu') if ratios is not None else 1.0
        rows[n:] = rows[n - 1]
        self.sched_table.copy_(rows)
        self.sched_iter.fill_(int(first))
        self.sched_ratio_out = ratio_out
        self.lr_dev.fill_(float(rows[min(first, n - 1), 0]))
        if ratio_out is not None:
            ratio_out.fill_(float(rows[min(first, n - 1), 1]))

    def clear_schedule(self):
        self.sched_table = self.sched_iter = self.sched_ratio_out = None

    def schedule_args(self):
        if self.sched_table is None:
            return None
        return (self.sched_table, self.sched_iter, self.lr_dev, self.sched_ratio_out)

    def make_book(self, gate=None, counters=None, n_marched=None, n_kept=None, capacity=0):
        """The bookkeeping step() would launch with these arguments, as an ops.StepBook a backward can carry in its repair launch
        (ops.field_bwd(book=...): one launch per step fewer); None when this optimizer's step has no repair launch to ride in."""
        fixed = getattr(self.net, 'grid_grad_accum', 'fp32') == 'fixed'
        if not (fixed and getattr(self.net, 'redo_supported', False) and self.book_in_repair_launch):
            return None
        return ops.StepBook(self.step_dev, gate, counters, n_marched, n_kept if n_kept is not None else gate, capacity=capacity,
                            overflow=ops.overflow_flag(self.net.params.device), remote_flags=None, eff_gate=self.eff_gate,
                            schedule=self.schedule_args(), overflow_redone=True)

    book_in_repair_launch = os.environ.get('PERF_BOOK_IN_REPAIR_LAUNCH', '1') != '0'

    def step(self, gate=None, counters=None, n_marched=None, n_kept=None, capacity=0, remote_flags=None, book=None):
        """gate (device int64 [1], optional): the number of samples behind this gradient.  The step is TAKEN unless the gate
        is 0 (the reference skips batches without samples, nerf.py:204-206), the fixed-point grid backward raised its
        overflow flag, or the batch was truncated at `capacity` samples -- decided on the device by perf_step_bookkeeping,
        which also advances the step count and accumulates `counters` (THIS rank's statistics; n_kept defaults to the gate)."""
        p = self.net.params
        if p.grad is None:
            return
        g = self.param_groups[0]
        self.refresh_lr()
        fixed = getattr(self.net, 'grid_grad_accum', 'fp32') == 'fixed'
        flag = ops.overflow_flag(p.device) if fixed else None
        # (a flagged fixed-point gradient was repaired in place by the redo launch behind the backward, NeRFScene._field_grad:
        #  the event is counted, the step is taken)
        rode = book is not None and book.done        # (the backward's repair launch did the bookkeeping: Adam consumes the flag)
        if not rode:
            ops.step_bookkeeping(self.step_dev, gate, counters, n_marched, n_kept if n_kept is not None else gate, capacity=capacity,
                                 overflow=flag, remote_flags=remote_flags, eff_gate=self.eff_gate, schedule=self.schedule_args(),
                                 overflow_redone=fixed and getattr(self.net, 'redo_supported', False))
        ops.adam_step_dev(p.data, self.exp_avg, self.exp_avg_sq, p.grad[:p.numel()], self.step_dev, self.lr_dev, g['betas'][0],
                          g['betas'][1], g['eps'], w16=self.w16, zero_grad=False, gate=self.eff_gate, clear_flag=flag if rode else None)
        p.grad = None                              # the next backward installs a fresh gradient (no accumulate pass)
        self.net.set_working_copy(self.w16)        # the kernel wrote the refreshed 16-bit copy


class _HipStepKernels:
    """The compute steps perf_amd.dp.ShardedExchange injects, on the gfx950 kernels, for one network."""

    def __init__(self, net, optimizer):
        self.net, self.opt = net, optimizer

    def stats_pack(self, level_absmax, field_max_prev, n_dev, n, out):
        ops.dp_stats_pack(level_absmax, field_max_prev, n_dev, n, out=out)

    def units(self, stats_all, world, shifts, n_total, margin_bits=0):
        ops.dp_units(self.net.grid, stats_all, world, self.net.headroom_state(), shifts, n_total, margin_bits=margin_bits,
                     want_total=n_total is not None)

    def slot_pack(self, level_absmax, field_max, n_dev, n, flag, n_marched, capacity, rank, world, out):
        ops.dp_slot_pack(level_absmax, field_max, n_dev, n, flag, n_marched, capacity, rank, world, out)

    def slot_unpack(self, slots, world, stats_all, job_flags, n_total):
        ops.dp_slot_unpack(slots, world, stats_all, job_flags, n_total)

    def unfix(self, shard, lo, hi, shifts, field_max, flag):
        ops.fixed_unfix(self.net.grid, shard, lo, hi, shifts, field_max, flag)

    def bookkeeping(self, step_dev, gate, counters, n_marched, n_kept, capacity, overflow, remote_flags, eff_gate):
        ops.step_bookkeeping(step_dev, gate, counters, n_marched, n_kept, capacity=capacity, overflow=overflow,
                             remote_flags=remote_flags, eff_gate=eff_gate, schedule=self.opt.schedule_args())

    def adam(self, p, m, v, g, w16, step_dev, lr_dev, gate):
        b = self.opt.param_groups[0]
        ops.adam_step_dev(p, m, v, g, step_dev, lr_dev, b['betas'][0], b['betas'][1], b['eps'], w16=w16, zero_grad=False, gate=gate)

    def overflow_flag(self):
        return ops.overflow_flag(self.net.params.device)


class NeRFScene:
    def __init__(self, base_exp_dir=None, train_conf=None, estimator_type='occ', renderer_conf=None, dtype=None,
                 fused_adam=True, writer=None):
        if estimator_type != 'occ':
            raise NotImplementedError("estimator_type 'prop' is dead code in the reference (nerf_renderer.py:73)")
        self.aabb = torch.tensor([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], device='cpu')
        self.base_exp_dir = base_exp_dir
        # The reference builds a tensorboard SummaryWriter under base_exp_dir/ts_log itself (nerf.py:37) and logs five scalars
        # per step (:213,238,255,286,295).  Here: any object with add_scalar(tag, value, step) -- the caller's, or, when the
        # scene is constructed the runner's way (a base_exp_dir, no writer) and tensorboard is importable, the same
        # SummaryWriter; None otherwise.  Scalars are written every `writer_every`-th step (a logged loss is a device
        # reduction + a host read-back; the reference pays that on every step).
        if writer is None and base_exp_dir is not None:
            try:
                from torch.utils.tensorboard import SummaryWriter
                writer = SummaryWriter(log_dir=os.path.join(base_exp_dir, 'ts_log'))
            except Exception:      # noqa: BLE001 -- tensorboard absent (or unusable): train without scalars, as before
                writer = None
        self.writer = writer
        self.writer_every = 16
        self.train_conf = train_conf or default_train_conf()
        self.nerf = NGPNeRF(aabb=self.aabb, dtype=dtype)
        self.estimator = OccGridEstimator(roi_aabb=self.aabb, resolution=256, levels=1).cuda()
        self.renderer = NeRFOCCRenderer(**(renderer_conf or {'max_radius': 2, 'bg_color': 'rand_noise'}))
        self.fused_adam = fused_adam
        self.global_iter_step_geo = 0
        self.global_iter_step_app = 0
        self.loss_scale = 2.0 ** 7
        self.last_losses = _Losses()
        self._capturing = False
        self.fused_steps = True        # explicit kernel chains for the two training steps (False: autograd formulation)
        # The reference's geometry step renders colours (query key 'rgb', nerf.py:197-201) that no loss term of that step
        # reads (:208-252).  True drops that colour-field forward: identical parameters, ~20 % less work.  Off by default
        # so that bench.py times the reference's step (one ray-sample = BOTH fields evaluated).
        self.skip_unused_color = False
        self.overlap_comm = True       # DP: overlap the gradient all-reduce with the next step's prefetch
        self.pixel_sup_rand_mode = 'by_all_pixels'     # what the reference passes everywhere (nerf.py:130,135)
        self.graph_steps = True        # train_one_episode replays hipGraph-captured steps when it can
        # DP payload of the gradient all-reduce: 'fp32' (exact sum, 26.6 MB) or 'bf16' (13.3 MB: every rank's gradient is
        # rounded to bf16 and summed in bf16 by RCCL -- ~2^-9 relative noise on a quantity Adam normalises anyway)
        self.comm_dtype = 'fp32'
        # The reference's geometry step evaluates the density field twice on the kept samples: without gradient inside
        # OccGridEstimator.sampling and with gradient in the renderer (nerf_renderer.py:145-148, :166-168) -- same parameters,
        # same positions.  True (sync-free mode): the encoded features of the sampler's pass are compacted along with the
        # samples -- together with the densities computed from them -- and the gradient pass starts from those instead of
        # encoding and evaluating again: bit-identical parameters (tests/test_gpu_counts.py), one encode and one MLP forward fewer.  On by default; False is the strict two-encode order of the reference
        # (bench.py reports both).
        self.reuse_sampling_features = True
        self._geo_pre = None
        self._ratio_dev = torch.zeros((), dtype=torch.float32, device='cuda')   # distortion-loss ramp min(2*progress, 1)
        # device-side statistics of perf_step_bookkeeping (int64 [8]): {marched, kept, steps, largest batch, steps skipped for
        # fixed-point overflow, steps skipped for truncation}.  bench.py reads them once after the timed region: throughput is
        # counted in samples that were really evaluated and composited; _poll_health reads them every 64 steps.
        self.sample_counters = ops.step_counters('cuda')
        self._health_seen = [0, 0]     # overflow / truncation skips already reported
        # Data parallelism: 'sharded' = int32 reduce-scatter of the fixed-point gradient fields -> Adam on this rank's slice
        # -> all-gather of the 16-bit working copy (perf_amd/dp.py; needs the fused Adam, the explicit step chains and the
        # fixed-point grid backward); 'allreduce' = one all-reduce of the flat fp32 (or bf16) gradient, Adam everywhere.
        self.dp_mode = 'sharded'
        # fixed-point units of the sharded exchange: 'lagged' = from the previous step's statistics (three collectives per
        # step, none between the MLP backward and the grid backward), 'exact' = statistics all-gather first (the units --
        # and, bit for bit, the summed table -- of the single process; four collectives).  perf_amd/dp.py.
        self.dp_units = os.environ.get('PERF_DP_UNITS', 'lagged')
        self._dp_timing = None         # bench.py: {collective: [(event, event)]} of eager data-parallel steps
        # Random draws of the explicit training steps (batch indices, stratified jitter, distance noise, background colour):
        # True = ONE launch of the counter-based device generator per step (SupInfoPool.draw_batch; seeded from
        # torch.initial_seed() at first use, so torch.manual_seed controls it; the ranks of a data-parallel job must seed
        # alike) -- a captured step then holds no torch random op, whose graph replays cost two extra launches each.
        # False, or an explicit `generator` / `rand` argument: torch.randint / torch.rand as in rounds 1-2.
        self.device_rng = True
        # sync-free training: let the health poll lower a sample capacity that proved far too large (see _poll_health).  The
        # lowered value lives as long as the density field it was measured on: a fresh geometry network or a new occupancy
        # (make_optimizer / prepare_occupancy) brings back the capacity the shrink started from (_capacity_unshrunk).
        self.auto_shrink_capacity = True
        self._capacity_unshrunk = None
        self._rng_seed = None
        self._rng_counter = None

    def _fixed_accum(self):
        return self.nerf.geo_mlp.grid_grad_accum == 'fixed' and self.nerf.app_mlp.grid_grad_accum == 'fixed'

    # ---- distributed helpers ---------------------------------------------------------------------
    _DP_OFF = False        # (class-wide switch, see bench.py: the plain step timed inside a multi-rank job)

    @staticmethod
    def _dist():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and not NeRFScene._DP_OFF:
            # (PERF_DP_SINGLE_RANK=1: a world of ONE rank takes the data-parallel path too -- how a single-GPU box exercises
            #  the RCCL exchange and its hipGraph capture, tests/test_gpu_dist.py)
            if dist.get_world_size() > 1 or os.environ.get('PERF_DP_SINGLE_RANK') == '1':
                return dist, dist.get_rank(), dist.get_world_size()
        return None, 0, 1

    # ---- rendering (nerf.py:74-123) --------------------------------------------------------------
    EVAL_SAMPLES_PER_RAY = 64      # initial per-ray sample capacity of the sync-free eval batches (grown on demand)

    @torch.no_grad()
    def render(self, rays: Rays, query_keys=('rgb',), batch_size=262144, sync_free=True):
        """NeRFScene.render (nerf.py:74-99).  The reference hard-codes 32,768-ray batches (:86); rays are independent and
        eval has no randomness, so the batch size does not change the result (tests/test_gpu_fullsize.py), only the launch
        overhead.  sync_free: every batch runs with capacity-sized sample arrays and device-side counts -- the host never
        waits for a sample count; ONE read-back at the end checks that no batch marched more samples than its capacity
        (otherwise the capacity is raised and the panorama rendered again, so results never depend on it)."""
        last_train = self.nerf.training
        self.set_eval()
        rays_o, rays_d = rays.collapse()
        pre_shape = list(rays_o.shape[:-1])
        rays_o = rays_o.reshape(-1, 3); rays_d = rays_d.reshape(-1, 3)
        r = self.renderer
        saved_capacity = r.sample_capacity
        try:
            while True:
                per_ray = getattr(self, '_eval_spp_cap', self.EVAL_SAMPLES_PER_RAY)
                ret = {k: [] for k in query_keys}
                marched = []
                for ro, rd in zip(rays_o.split(batch_size), rays_d.split(batch_size)):
                    if sync_free:
                        r.sample_capacity = int(ro.shape[0]) * per_ray
                    cur = self.render_once(Rays(ro, rd), list(query_keys) + (['n_marched_dev'] if sync_free else []))
                    for k in query_keys:
                        ret[k].append(cur[k])
                    if sync_free:
                        marched.append(cur['n_marched_dev'] / float(ro.shape[0]))
                if not sync_free:
                    break
                worst = float(torch.stack(marched).max().item())           # the render's single host read-back
                if worst <= per_ray:
                    break
                self._eval_spp_cap = int(math.ceil(worst * 1.25))          # truncated batch(es): render again with room
        finally:
            r.sample_capacity = saved_capacity
        for k in query_keys:
            ret[k] = torch.cat(ret[k], dim=0).reshape(pre_shape + [-1])
        if last_train:
            self.set_train()
        return ret

    def render_once(self, rays: Rays, query_keys=('rgb',), geo_inference=False, app_inference=False, rand=None):
        rays_o, rays_d = rays.collapse()
        assert len(rays_o.shape) == 2
        n = len(rays_o)
        # to_bounded_rays (nerf.py:313-319): constants the occupancy renderer never reads -- kept per (n, device) instead of being
        # filled on every call (three launches of a captured frame otherwise)
        # (a small dict: every batch size of a frame is met in the eager warm-up pass that precedes a capture, so a captured pass
        #  never allocates them from a graph's pool)
        cache = self.__dict__.setdefault('_bounds', {})
        key = (n, rays_o.device)
        if key not in cache:
            if len(cache) >= 8:
                cache.clear()
            cache[key] = (1e-2 * torch.ones([n, 1], device=rays_o.device), torch.ones([n, 1], device=rays_o.device))
        near, far = cache[key]
        res = self.renderer.render(self.nerf, self.estimator, rays_o, rays_d, near, far,
                                   geo_inference=geo_inference, app_inference=app_inference, rand=rand)
        if (res is None) or (not res['is_valid']):
            return res
        return {k: res.get(k) for k in list(query_keys) + ['is_valid']}

    # ---- training (nerf.py:125-311) ----------------------------------------------------------------
    def fit(self, sup_pool: SupInfoPool, **kw):
        self.train_one_episode(sup_pool, self.train_conf.raw_phase_iter_geo, self.train_conf.raw_phase_iter_app, **kw)

    def _restore_capacity(self):
        """Undo _poll_health's capacity shrink: the marched counts it was sized on belong to a density field / occupancy that is
        being replaced (a fresh field is transparent: PeRF's batches march ~300 k samples against ~35 k late in a phase)."""
        if self._capacity_unshrunk is not None:
            if self.renderer.sample_capacity is not None:
                self.renderer.sample_capacity = max(self.renderer.sample_capacity, self._capacity_unshrunk)
            self._capacity_unshrunk = None
            self.sample_counters[3] = 0               # the shrink window starts over

    def prepare_occupancy(self, sup_pool, warmup='direct'):
        self._geo_pre = None           # a batch prefetched under the previous episode's pool / occupancy must not be consumed
        self._restore_capacity()
        self.estimator = OccGridEstimator(roi_aabb=self.aabb, resolution=256, levels=1).cuda()
        self.estimator.train()
        pre_grid, _ = sup_pool.gen_occ_grid(res=256)
        if warmup == 'direct':
            self.estimator.set_binaries(pre_grid)
        else:
            occ_res = 256

            def occ_eval_fn(x):
                x = (x.clip(-0.999, 0.999) * .5 + .5) * occ_res
                x = x.to(torch.int64)
                return pre_grid[x[..., 0] * occ_res * occ_res + x[..., 1] * occ_res + x[..., 2]].float()

            for i in range(256):
                self.estimator.update_every_n_steps(step=i, occ_eval_fn=occ_eval_fn, occ_thre=1e-2, ema_decay=0.1,
                                                    warmup_steps=256, n=1)
        return pre_grid

    def make_optimizer(self, net, lr):
        # A new optimizer = a new phase on possibly new supervision: the closed loop on the fixed-point headroom starts over
        # from its safe guess.  At the end of a phase the gradient is noise (an entry's contributions cancel: sums ~ sqrt(N)),
        # on new data it is coherent (sums ~ N) -- a state tuned to the former lets the first steps of the latter run into the
        # overflow flag, and the gate drops them (tools/mini_perf_loop.py: one step per newly registered panorama).
        hr = getattr(net, 'headroom_state', None)
        if hr is not None:
            hr().zero_()
        if net is self.nerf.geo_mlp:
            self._restore_capacity()      # (callers of make_graphed_step / train_one_step_geo outside train_one_episode too)
        return FusedAdam(net, lr) if self.fused_adam else torch.optim.Adam(net.parameters(), lr=lr)

    def train_one_episode(self, sup_pool, geo_res_iters, app_res_iters, warmup='direct', callback=None, use_graphs=None):
        """nerf.py:137-184.  use_graphs (default: whenever possible = fused Adam, explicit step chains, single process): the
        first EAGER_HEAD iterations of a phase run eagerly, then the step is captured once and replayed -- same kernels,
        same order, no host work per step besides the learning-rate value."""
        self.set_train()
        self.prepare_occupancy(sup_pool, warmup)
        self.nerf.reset_geo()
        if use_graphs is None:
            use_graphs = self.graph_steps
        sync_free = self.fused_adam and self.fused_steps          # capacity-sized arrays + device-side counts: no host read-back
        use_graphs = bool(use_graphs) and sync_free and self.dp_graph_ok()
        saved_capacity = self.renderer.sample_capacity
        if sync_free and saved_capacity is None:
            per_rank = self.train_conf.pixel_loss_batch_size // max(self._dist()[2], 1)
            self.renderer.sample_capacity = per_rank * self.TRAIN_SAMPLES_PER_RAY
        try:
            geo_optimizer = self.make_optimizer(self.nerf.geo_mlp, self.train_conf.geo_optimizer.init_lr)
            self._run_phase('geo', geo_optimizer, self.train_conf.geo_optimizer, geo_res_iters, sup_pool, callback,
                            use_graphs and self._can_fuse(), lambda i: i / app_res_iters)
            app_optimizer = self.make_optimizer(self.nerf.app_mlp, self.train_conf.app_optimizer.init_lr)
            self._run_phase('app', app_optimizer, self.train_conf.app_optimizer, app_res_iters, sup_pool, callback,
                            use_graphs, lambda i: i / app_res_iters)
            self.sync_params()            # (sharded data parallelism: every rank leaves the episode with the full fp32 master)
        finally:
            self.renderer.sample_capacity = saved_capacity
            self._capacity_unshrunk = None

    EAGER_HEAD = 3

    def dp_graph_ok(self):
        """Can a training step be captured in a hipGraph in this process?  Single process: yes.  Data parallel: only the
        sharded exchange over RCCL, and only after a probe -- a tiny all-reduce on a group of its own, captured and replayed
        -- came back right on EVERY rank (the verdicts are agreed on with one eager all-reduce).  PERF_DP_GRAPH=0 opts out.
        Collective: every rank must call it at the same point."""
        dist, rank, world = self._dist()
        if dist is None:
            return True
        global _DP_GRAPH_VERDICT
        if _DP_GRAPH_VERDICT is not None:
            return _DP_GRAPH_VERDICT and self.dp_mode == 'sharded' and self.fused_adam and self._fixed_accum()
        import os
        ok = dist.get_backend() == 'nccl' and os.environ.get('PERF_DP_GRAPH', '1') != '0'
        if ok:
            try:
                group = dist.new_group()
                t = torch.ones(1024, device='cuda')
                dist.all_reduce(t, group=group)                  # (communicator set-up happens outside the capture)
                torch.cuda.synchronize()
                t.fill_(1.0)
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    dist.all_reduce(t, group=group)
                t.fill_(1.0)
                g.replay()
                torch.cuda.synchronize()
                ok = bool(float(t[0].item()) == float(world) and float(t[-1].item()) == float(world))
            except Exception as e:       # noqa: BLE001 -- any failure means: stay eager
                import warnings
                warnings.warn(f'perf_amd: capturing an RCCL collective in a hipGraph failed ({type(e).__name__}: {e}); data-parallel steps stay eager')
                ok = False
            v = torch.tensor([1.0 if ok else 0.0], device='cuda')
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            ok = bool(v.item() > 0.5)
        _DP_GRAPH_VERDICT = ok            # (one probe per process: the verdict is a property of the node and the backend)
        return ok and self.dp_mode == 'sharded' and self.fused_adam and self._fixed_accum()

    def _run_phase(self, kind, optimizer, conf, n_iters, sup_pool, callback, use_graphs, progress_of):
        step_fn = self.train_one_step_geo if kind == 'geo' else self.train_one_step_app
        graphed = None
        for iter_i in range(n_iters):
            if use_graphs and graphed is None and iter_i >= self.EAGER_HEAD:
                lrs = [self.lr_at(conf, i / n_iters) for i in range(n_iters)]
                ratios = [float(min(progress_of(i) * 2., 1.)) for i in range(n_iters)]      # (np.min([..]) of nerf.py:235: 4,500 of them cost 87 ms)
                graphed = self.make_graphed_step(kind, optimizer, sup_pool, warmup=0, schedule=(lrs, ratios, iter_i))
            if graphed is not None:
                graphed()
                if self._dist()[0] is not None and self._poll_health()['recapture']:
                    # (single-process replays poll and re-capture by themselves; data-parallel ones here, on every rank at
                    #  the same iteration -- the poll's verdict is job-wide.)  The capacity is baked into the captured step:
                    #  capture again, the device-side schedule carries on where it is.
                    graphed = self.make_graphed_step(kind, optimizer, sup_pool, warmup=0)
            else:
                self.update_lr(optimizer, conf, iter_i / n_iters)
                if kind == 'geo':
                    # (no prefetch from the step before the capture: the captured step has to draw its own batch)
                    last_eager = use_graphs and iter_i + 1 >= self.EAGER_HEAD
                    step_fn(optimizer, sup_pool, progress=progress_of(iter_i), prefetch_next=iter_i + 1 < n_iters and not last_eager)
                else:
                    step_fn(optimizer, sup_pool, progress=progress_of(iter_i))
            self._log_scalars(kind, self.lr_at(conf, iter_i / n_iters))
            if callback:
                callback(kind, iter_i)

    def _log_scalars(self, kind, lr):
        """nerf.py:213,238,255 (geometry step) / :286,295 (colour step): the step's loss terms and learning rate, tagged with the
        index of the step just taken (the reference logs before it increments its counter)."""
        w = self.writer
        if w is None:
            return
        step = (self.global_iter_step_geo if kind == 'geo' else self.global_iter_step_app) - 1
        if step < 0 or step % max(int(self.writer_every), 1) != 0:
            return
        tags = (('depth_loss', 'nerf_loss/depth_loss'), ('dist_loss', 'nerf_loss/dist_loss')) if kind == 'geo' \
            else (('color_loss', 'nerf_loss/color_loss'),)
        for key, tag in tags:
            if key in self.last_losses:
                w.add_scalar(tag, float(self.last_losses[key]), step)
        w.add_scalar('others/lr_' + kind, float(lr), step)

    def _batch(self, sup_pool, generator=None):
        dist, rank, world = self._dist()
        bs = self.train_conf.pixel_loss_batch_size
        rays, col, dep, nrm = sup_pool.rand_ray_color_data(bs, rand_mode=self.pixel_sup_rand_mode, generator=generator,
                                                            rank=rank, world_size=world)
        return rays, col, dep, bs, (dist, rank, world)

    def _use_device_rng(self, rand, generator):
        return self.device_rng and self.fused_steps and generator is None and not rand

    def _draw(self, sup_pool, want_bg):
        """The step's batch and uniform draws from the device generator -> (rays, colors, depths, global batch, dist_info, rand)."""
        dist, rank, world = self._dist()
        if self._rng_counter is None:
            self._rng_seed = int(torch.initial_seed())
            self._rng_counter = torch.zeros(1, dtype=torch.int64, device=sup_pool.all_sup_colors.device)
        bs = self.train_conf.pixel_loss_batch_size
        rays, col, dep, nrm, rand = sup_pool.draw_batch(bs, self._rng_seed, self._rng_counter, rand_mode=self.pixel_sup_rand_mode,
                                                        rank=rank, world_size=world, want_bg=want_bg)
        return rays, col, dep, bs, (dist, rank, world), rand

    def _finish_step(self, loss, net, optimizer, dist_info, overlap=None):
        """backward -> [one RCCL all-reduce of the flat gradient] -> Adam.  `overlap` (a callable) is run between the
        launch of the asynchronous all-reduce and the wait for it: the next step's parameter-independent work then
        executes on the compute stream while RCCL moves the gradient over xGMI on its own stream."""
        dist, rank, world = dist_info
        (loss * self.loss_scale).backward()                   # grad_scaler.scale(loss).backward(), never unscaled
        if dist is not None:
            g = net.params.grad
            if g is None:
                g = net.params.grad = torch.zeros_like(net.params)
            if overlap is not None:
                work = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
                overlap()
                work.wait()
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
        optimizer.step()
        self._poll_health(net)

    def _poll_health(self, net=None, n_marched=None, force=False):
        """Every OVERFLOW_CHECK_EVERY steps: ONE host read-back of the device-side counters (perf_step_bookkeeping).  A step
        whose fixed-point grid gradient overflowed was REPAIRED on the device (fp32 redo launch; skipped on every rank alike
        under data parallelism, where a repair would need a second exchange), a step whose batch was truncated at the sample
        capacity was skipped; here the host hears about them: it warns and doubles a capacity that proved too small.  Under
        data parallelism the ranks agree on what they saw (one tiny all-reduce), so they raise the capacity -- and re-capture
        -- in lockstep.  -> {'recapture': bool} for callers that replay a captured graph (the capacity is baked into it)."""
        self._steps_since_check = getattr(self, '_steps_since_check', 0) + 1
        if self._capturing or (self._steps_since_check < OVERFLOW_CHECK_EVERY and not force):
            return {'recapture': False}
        self._steps_since_check = 0
        import warnings
        c = self.sample_counters.tolist()
        seen = self._health_seen
        new_ovf = c[4] - seen[0] if c[4] >= seen[0] else c[4]          # (bench.py zeroes the counters between regions)
        new_trunc = c[5] - seen[1] if c[5] >= seen[1] else c[5]
        self._health_seen = [c[4], c[5]]
        recapture = False
        dist = self._dist()[0]
        multi = dist is not None
        if multi:
            # every rank must reach the same decisions (capacity, recapture) at the same step: the polls run in lockstep,
            # one tiny MAX all-reduce of what the ranks saw
            t = torch.tensor([new_ovf, new_trunc, c[3]], dtype=torch.int64, device=self.sample_counters.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            new_ovf, new_trunc, c[3] = [int(v) for v in t.tolist()]
        if new_ovf > 0:
            # (the headroom feedback has widened the fields by now; nothing to switch: single-GPU steps repaired the gradient
            #  in fp32 and were taken, data-parallel ones were skipped on every rank alike)
            warnings.warn(f'perf_amd: {new_ovf} fixed-point grid gradient(s) came within 4x of the int32 range; '
                          + ('the steps were skipped job-wide' if multi else 'they were redone with fp32 accumulation (no step was dropped)'))
        cap = self.renderer.sample_capacity
        if cap is not None and (new_trunc > 0 or c[3] > cap):
            warnings.warn(f'perf_amd: {new_trunc} training batch(es) marched up to {c[3]} samples, more than the capacity {cap}; '
                          'their steps were skipped; raising the capacity')
            self.renderer.sample_capacity = max(2 * cap, int(1.25 * c[3]))
            self.sample_counters[3] = 0
            recapture = True
        elif cap is not None and self.auto_shrink_capacity and c[3] > 0:
            # The capacity is the launch size of every per-sample kernel of the step (and the stride of the level-major
            # buffers): PeRF's 8,192-ray batches march 300 k samples while the density field is transparent and 35 k once it
            # has formed -- against a capacity of 8,192 x 128 rows.  Launches sized for a million rows that hold 35 k cost
            # their dispatch (the encode's 32,768 workgroups that read the count and leave: 10 us each, twice per step).
            # c[3] is the largest marched count SINCE THE LAST POLL (a window of 64 steps): when the capacity is more than
            # SHRINK_AT times that, it drops to SHRINK_TO times it -- far above anything the window saw; a batch that still
            # exceeded it would be truncated, its step skipped and the capacity raised again (above).  Results do not depend on
            # the capacity (tests/test_gpu_counts.py); the captured step is captured again.
            if cap > self.CAPACITY_SHRINK_AT * c[3]:
                new = max(int(self.CAPACITY_SHRINK_TO * c[3]), self.CAPACITY_MIN_ROWS)
                new = (new + 4095) // 4096 * 4096
                if new < cap:
                    if self._capacity_unshrunk is None:
                        self._capacity_unshrunk = cap
                    self.renderer.sample_capacity = new
                    recapture = True
            self.sample_counters[3] = 0               # the window starts over
        return {'recapture': recapture}

    CAPACITY_SHRINK_AT, CAPACITY_SHRINK_TO, CAPACITY_MIN_ROWS = 8, 4, 65536

    def _geo_prefetch(self, sup_pool, rand, generator):
        """Everything of a geometry step that does not depend on the geometry parameters: batch draw, and -- when the
        sampler needs no density pre-pass (early_stop_eps == 0) -- marching, positions and the frozen colour field."""
        rand = dict(rand or {})
        if self._use_device_rng(rand, generator):
            rays, gt_colors, gt_depths, bs, dist_info, rand = self._draw(sup_pool, want_bg=False)
        else:
            rays, gt_colors, gt_depths, bs, dist_info = self._batch(sup_pool, generator)
        if self.fused_steps and ('jitter' not in rand or 'noise' not in rand):
            # the step's two per-ray draws in one launch; under data parallelism every rank draws the GLOBAL batch's
            # values (same seed on every rank) and keeps its slice, like the index stream: the job then trains on the very
            # batch the single process would
            u = self._rand_rows(2, rays.o.shape[0], dist_info, rays.o.device)
            rand.setdefault('jitter', u[0].contiguous()); rand.setdefault('noise', u[1].contiguous().unsqueeze(1))
        st = None
        if self.renderer.early_stop_eps <= 0:
            with torch.no_grad():
                st = self.renderer.stage_sample(self.nerf, self.estimator, rays.o, rays.d, rand,
                                                with_rgb=not (self.fused_steps and self.skip_unused_color))
                st = st if st is not None else False
        return {'rays': rays, 'gt_depths': gt_depths, 'bs': bs, 'dist_info': dist_info, 'st': st, 'rand': rand}

    @staticmethod
    def _rand_rows(rows, n_local, dist_info, device):
        """torch.rand(rows, n_local) single-process; with W ranks the rank's column slice of torch.rand(rows, W * n_local)."""
        _, rank, world = dist_info
        if world == 1:
            return torch.rand(rows, n_local, device=device)
        return torch.rand(rows, n_local * world, device=device)[:, rank * n_local:(rank + 1) * n_local]

    @staticmethod
    def _rand_cols(n_local, cols, dist_info, device):
        """torch.rand(n_local, cols) single-process; with W ranks the rank's row slice of torch.rand(W * n_local, cols)."""
        _, rank, world = dist_info
        if world == 1:
            return torch.rand(n_local, cols, device=device)
        return torch.rand(n_local * world, cols, device=device)[rank * n_local:(rank + 1) * n_local].contiguous()

    # ---- fused steps: explicit kernel chain instead of autograd + ~25 tiny torch ops (same arithmetic) ----------
    def _field_grad(self, net, x01, w16, feat, sel, dout, n_dev=None, extra=0, book=None):
        """Flat gradient [network | grid] (+ `extra` trailing slots: the data-parallel path appends the sample count)."""
        n_net = net.mlp.n_params
        fixed = net.grid_grad_accum == 'fixed'
        n_all = n_net + net.grid.n_params
        if not fixed or net.redo_supported:
            # ONE boundary call (perf_field_bwd): MLP backward -> grid backward -> the predicated fp32 repair launch that keeps a
            # flagged fixed-point step from being dropped (a no-op dispatch otherwise)
            grad = ops.field_bwd(net.grid, net.mlp, x01, w16[:n_net], feat, dout, sel, fixed=fixed, redo=True,
                                 hr_state=net.headroom_state() if fixed else None, n_dev=n_dev, extra=extra, book=book)
            if fixed and not self.fused_adam:
                # torch.optim.Adam has no perf_step_bookkeeping behind it to consume the flag: left set, every later backward
                # would run its (slow) fp32 repair as well
                ops.overflow_flag(x01.device).zero_()
            return grad
        grad = torch.empty(n_all + extra, dtype=torch.float32, device=x01.device)
        res = ops.mlp_bwd(net.mlp, w16[:n_net], feat, dout, sel, want_absmax=fixed, n_dev=n_dev, dw_out=grad[:n_net])
        ops.hashgrid_bwd_into(net.grid, x01, res[0], grad[n_net:n_all], level_absmax=res[2] if fixed else None, n_dev=n_dev,
                              hr_state=net.headroom_state() if fixed else None)
        if fixed and net.redo_supported:
            # never drop a step: should a fixed-point field have neared the int32 range (device flag), this predicated launch
            # rewrites the table gradient with fp32 LDS accumulation; a no-op dispatch otherwise (perf_hashgrid_bwd, redo_flag)
            ops.hashgrid_bwd_redo(net.grid, x01, res[0], grad[n_net:n_all], n_dev=n_dev, hr_state=net.headroom_state())
            if not self.fused_adam:
                # torch.optim.Adam has no perf_step_bookkeeping behind it to consume the flag: left set, every later backward
                # would run its (slow) fp32 repair as well
                ops.overflow_flag(x01.device).zero_()
        return grad

    def _step_book(self, optimizer, dist_info, n_dev, n_marched):
        """Single process, sync-free mode, fused Adam: the step's bookkeeping as a block the backward's repair launch carries (the
        arguments _apply_grad hands optimizer.step); None otherwise."""
        if dist_info[0] is not None or not isinstance(optimizer, FusedAdam) or not torch.is_tensor(n_dev):
            return None
        return optimizer.make_book(gate=n_dev, counters=self.sample_counters, n_marched=n_marched, n_kept=n_dev,
                                   capacity=self.renderer.sample_capacity or 0)

    DP_EXTRA = 3           # trailing slots of the data-parallel gradient buffer: [sample count, overflow flag, truncated]

    def _apply_grad(self, net, grad, optimizer, dist_info, overlap, n_kept=None, n_marched=None, book=None):
        """[one RCCL all-reduce of the flat gradient] -> Adam.  n_kept: number of samples behind `grad` (device int64 [1], or a
        host int in the eager variable-count path).  Under data parallelism three trailing slots of the gradient buffer travel
        with it: the sample count -- the optimizer step is skipped when NO rank had samples (the reference's
        `if not is_valid: return`, nerf.py:204-206) --, this rank's fixed-point overflow flag and whether its batch was
        truncated at the sample capacity: after the one collective every rank holds the same sums and takes or skips the
        step alike (perf_step_bookkeeping's remote_flags)."""
        dist = dist_info[0]
        n = net.params.numel()
        gate = n_kept if torch.is_tensor(n_kept) else None
        remote = None
        if dist is not None:
            extra = grad.numel() - n
            if extra >= 1:
                if torch.is_tensor(n_kept):
                    grad[n:n + 1].copy_(n_kept)
                else:
                    grad[n:n + 1].fill_(float(n_kept if n_kept is not None else 1))
            if extra >= 3:
                if net.grid_grad_accum == 'fixed' and isinstance(optimizer, FusedAdam) and not net.redo_supported:
                    grad[n + 1:n + 2].copy_(ops.overflow_flag(grad.device))
                else:
                    grad[n + 1:n + 2].zero_()          # (a flagged local gradient was repaired in fp32 before it got here)
                cap = self.renderer.sample_capacity or 0
                if torch.is_tensor(n_marched) and cap > 0:
                    grad[n + 2:n + 3].copy_(n_marched > cap)
                else:
                    grad[n + 2:n + 3].zero_()
            payload = grad
            if self.comm_dtype == 'bf16':
                if extra >= 1:
                    grad[n:n + 1].clamp_(max=1.0)          # the count slot only has to say "some / none": exact in bf16
                payload = grad.to(torch.bfloat16)
            if self._dp_timing is not None:          # (bench.py: the collective alone, synchronous between two events)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                dist.all_reduce(payload, op=dist.ReduceOp.SUM)
                ev[1].record()
                self._dp_timing.setdefault('all_reduce_flat_gradient', []).append(ev)
                if overlap is not None:
                    overlap()
            elif overlap is not None:
                work = dist.all_reduce(payload, op=dist.ReduceOp.SUM, async_op=True)
                overlap()
                work.wait()
            else:
                dist.all_reduce(payload, op=dist.ReduceOp.SUM)
            if payload is not grad:
                grad.copy_(payload)
            if extra >= 1:
                gate = grad[n:n + 1].to(torch.int64)
            if extra >= 3:
                remote = grad[n + 1:n + 3]
        net.params.grad = grad[:n]                 # (without the trailing slots of the data-parallel buffer)
        if isinstance(optimizer, FusedAdam):
            optimizer.step(gate=gate, counters=self.sample_counters, n_marched=n_marched,
                           n_kept=n_kept if torch.is_tensor(n_kept) else None, capacity=self.renderer.sample_capacity or 0,
                           remote_flags=remote, book=book)
        else:
            if gate is None or int(gate.item()) > 0:
                optimizer.step()
            net.params.grad = None
        self._poll_health(net)

    # ---- data parallelism, sharded mode (perf_amd/dp.py) ---------------------------------------------------------------
    def _sharded(self, dist_info, optimizer):
        return (dist_info[0] is not None and self.dp_mode == 'sharded' and isinstance(optimizer, FusedAdam)
                and optimizer.net.grid_grad_accum == 'fixed')

    def _exchange_for(self, net, optimizer, dist_info):
        ex = getattr(net, '_dp_exchange', None)
        if ex is None or ex.k.opt is not optimizer:
            from .dp import Collectives, ShardedExchange
            dist, rank, world = dist_info
            ex = ShardedExchange(net.mlp.n_params, net.grid.n_params, world, rank, Collectives(dist), net.params.device,
                                 ops.torch_dtype(net.dtype_name), _HipStepKernels(net, optimizer), units=self.dp_units)
            ex.timing = self._dp_timing
            ex.seed_working_copy(net.working_copy())
            object.__setattr__(net, '_dp_exchange', ex)
        return ex

    def _dp_sharded_step(self, net, optimizer, dist_info, x01, w16, feat, sel, dout, n_dev, n_marched, early=None, late=None):
        """Backward + exchange + optimizer of one network under sharded data parallelism: MLP backward -> job-wide units ->
        grid backward into int32 fields -> reduce-scatter -> Adam on this rank's slice -> all-gather of the 16-bit copy.
        early / late: callables run while the (tiny) statistics all-gather / the gradient exchange are in flight.
        x01 None: this rank has no sample at all (it still takes part in every collective)."""
        ex = self._exchange_for(net, optimizer, dist_info)
        n_net = net.mlp.n_params
        if x01 is None:
            amax = torch.zeros(_tcnn.ops._lib.MAX_LEVELS, dtype=torch.float32, device=net.params.device)
            dw = torch.zeros(n_net, dtype=torch.float32, device=net.params.device)
            shifts, _ = ex.exchange_units(amax, None, 0, overlap=early)
            ex.payload.zero_()
        else:
            dfeat, dw, amax = ops.mlp_bwd(net.mlp, w16[:n_net], feat, dout, sel, want_absmax=True, n_dev=n_dev)
            shifts, _ = ex.exchange_units(amax, n_dev, x01.shape[0], overlap=early)
            ops.hashgrid_bwd_into(net.grid, x01, dfeat, ex.grid_payload_f32(), n_dev=n_dev, shifts=shifts, raw_fields=True)
        optimizer.refresh_lr()
        w16_new = ex.reduce_and_step(dw, optimizer, counters=self.sample_counters, n_marched=n_marched, n_kept=n_dev,
                                     capacity=self.renderer.sample_capacity or 0, overlap=late)
        net.set_working_copy(w16_new)
        net.params.grad = None
        self._poll_health(net)

    def sync_params(self):
        """Sharded data parallelism keeps the fp32 master of a table slice only on its owner: refresh the replicas (one fp32
        all-gather per network; before checkpoints, after an episode).  No-op otherwise."""
        for net in (self.nerf.geo_mlp, self.nerf.app_mlp):
            ex = getattr(net, '_dp_exchange', None)
            if ex is not None and self._dist()[0] is not None:
                ex.gather_master(net.params.data)

    @torch.no_grad()
    def _geo_step_fused(self, optimizer, sup_pool, progress, rand, generator, prefetch_next=True):
        """train_one_step_geo (nerf.py:186-257) as an explicit chain: sampling (marching + no-grad density pass + visibility
        compaction) -> density field (kept features) -> colour field -> compositing -> fused loss head -> distortion /
        compositing / MLP / grid backward -> [all-reduce] -> Adam.  With renderer.sample_capacity set nothing in the chain
        reads the device: sample counts stay in device memory (st['n_dev'])."""
        tc = self.train_conf
        rand = rand or {}
        pre = self._geo_pre or self._geo_prefetch(sup_pool, rand, generator)
        self._geo_pre = None
        rays, gt_depths, bs, dist_info = pre['rays'], pre['gt_depths'], pre['bs'], pre['dist_info']
        st = pre['st']
        rand_in, rand = rand, pre['rand']
        if st is None:
            # (under data parallelism the colour render is deferred until the gradient all-reduce is in flight, see below)
            st = self.renderer.stage_sample(self.nerf, self.estimator, rays.o, rays.d, rand,
                                            with_rgb=not (self.skip_unused_color or (dist_info[0] is not None and self.overlap_comm)),
                                            keep_features=self.reuse_sampling_features and self.renderer.sample_capacity is not None)
        geo = self.nerf.geo_mlp
        extra = self.DP_EXTRA if dist_info[0] is not None else 0
        sharded = self._sharded(dist_info, optimizer)
        if st is None or st is False:
            if sharded:                        # keep the collectives matched across ranks: this rank contributes nothing
                self._dp_sharded_step(geo, optimizer, dist_info, None, None, None, None, None, None, None)
            elif dist_info[0] is not None:     # the count slot says "no samples here"
                self._apply_grad(geo, torch.zeros(geo.params.numel() + self.DP_EXTRA, device=geo.params.device), optimizer, dist_info, None, n_kept=0)
            self.global_iter_step_geo += 1
            return
        x01, sel, packed, ts, te, n_dev = st['x01'], st['sel'], st['packed'], st['t_starts'], st['t_ends'], st['n_dev']
        self._last_counts = (st['n_marched_dev'], n_dev)
        n_net = geo.mlp.n_params
        w16 = geo.working_copy()
        feat = st.get('feat0')
        if feat is not None and st.get('sig0') is not None:
            # reuse_sampling_features: the sampler's density pass left the kept samples' features AND the densities it computed
            # from them (compacted together).  The gradient pass would compute the very same numbers -- same kernel, same
            # weights, same features, sample by sample -- and the MLP backward recomputes its forward in registers anyway:
            # the second forward launch is dropped (bit-identical parameters, tests/test_gpu_counts.py).
            sig = st['sig0'].reshape(-1, 1)
        else:
            if feat is None:
                feat = ops.hashgrid_fwd(geo.grid, x01, w16[n_net:], n_dev=n_dev)
            elif isinstance(feat, ops.IndexedFeat):       # (features without densities: no caller does that today)
                feat = feat.materialize()
            sig = ops.mlp_fwd(geo.mlp, w16[:n_net], feat, sel, n_dev=n_dev)
        # The colour render of this step (query key 'rgb', nerf.py:197-201) feeds no loss term (:208-252).  Under data
        # parallelism it is therefore issued AFTER the gradient all-reduce has been launched: the colour field's encode +
        # MLP + accumulation run on the compute stream while RCCL moves the gradient over xGMI on its own stream.
        defer_color = (dist_info[0] is not None and self.overlap_comm and not self.skip_unused_color and st['rgbs'] is None)
        rgbs = None if (self.skip_unused_color or defer_color) else (st['rgbs'] if st['rgbs'] is not None else self.nerf.rgb_at(x01, sel, n_dev))
        noise = rand['noise']            # (the background colour the reference also draws, :185, is not used by this step)
        if not self._capturing and getattr(optimizer, 'sched_table', None) is None:
            self._ratio_dev.fill_(float(np.min([progress * 2., 1])))
        # compositing forward -> loss head -> compositing backward: ONE launch, a wavefront per ray (perf_train_head_geo)
        hd = ops.train_head_geo(sig.view(-1), rgbs, ts, te, packed, gt_depths, noise, bs, tc.depth_loss_weight, tc.distortion_loss_weight,
                                self._ratio_dev, self.loss_scale)
        w, dsig = hd['weights'], hd['d_sigma']
        if rgbs is not None:
            self.last_colors = hd['color']   # the step's (unused) colour render, [n_rays, 3]
        # (the loss VALUES are reports: summed from the per-ray terms when somebody reads them)
        self.last_losses['depth_loss'] = lambda t=hd['depth_terms'], s=1.0 / bs: t.sum() * s
        self.last_losses['dist_loss'] = lambda t=hd['distloss_per_ray'], s=hd['inv_n']: t.sum() * s[0]

        def color_now():
            self.last_colors = ops.accumulate_fwd(w, self.nerf.rgb_at(x01, sel, n_dev), packed)

        def prefetch_now():
            self._geo_pre = self._geo_prefetch(sup_pool, rand_in, generator)

        if sharded:
            # What runs beside the 26.6 MB reduce-scatter of the gradient fields: the deferred colour render (an encode + MLP +
            # accumulation that feed no loss term of this step: ~0.2 ms of compute at 1 M samples) and the next step's batch
            # draw.  Exact units: the colour render hides the statistics all-gather instead (it sits on the critical path there).
            want_prefetch = self.overlap_comm and prefetch_next and not self._capturing
            exact = self.dp_units == 'exact'

            def beside_reduce_scatter():
                if defer_color and not exact:
                    color_now()
                if want_prefetch:
                    prefetch_now()
            self._dp_sharded_step(geo, optimizer, dist_info, x01, w16, feat, sel, dsig.view(-1, 1), n_dev, st['n_marched_dev'],
                                  early=color_now if (defer_color and exact) else None,
                                  late=beside_reduce_scatter if ((defer_color and not exact) or want_prefetch) else None)
            self.global_iter_step_geo += 1
            return
        book = self._step_book(optimizer, dist_info, n_dev, st['n_marched_dev'])
        grad = self._field_grad(geo, x01, w16, feat, sel, dsig.view(-1, 1), n_dev=n_dev, extra=extra, book=book)
        overlap = None
        if self.overlap_comm and dist_info[0] is not None:
            def overlap():
                if defer_color:
                    color_now()
                if prefetch_next:
                    prefetch_now()
        self._apply_grad(geo, grad, optimizer, dist_info, overlap, n_kept=n_dev if n_dev is not None else x01.shape[0],
                         n_marched=st['n_marched_dev'], book=book)
        self.global_iter_step_geo += 1

    @torch.no_grad()
    def _app_step_fused(self, optimizer, sup_pool, progress, rand, generator):
        """train_one_step_app (nerf.py:259-297): density without gradient (reused from the sampler's visibility pass when
        there is one), colour field with gradient, colour smooth-L1."""
        tc = self.train_conf
        rand = rand or {}
        if self._use_device_rng(rand, generator):
            rays, gt_colors, gt_depths, bs, dist_info, rand = self._draw(sup_pool, want_bg=self.renderer.bg_color == 'rand_noise')
        else:
            rays, gt_colors, gt_depths, bs, dist_info = self._batch(sup_pool, generator)
        st = self.renderer.stage_sample(self.nerf, self.estimator, rays.o, rays.d, rand)
        app = self.nerf.app_mlp
        extra = self.DP_EXTRA if dist_info[0] is not None else 0
        sharded = self._sharded(dist_info, optimizer)
        if st is None:
            if sharded:
                self._dp_sharded_step(app, optimizer, dist_info, None, None, None, None, None, None, None)
            elif dist_info[0] is not None:
                self._apply_grad(app, torch.zeros(app.params.numel() + self.DP_EXTRA, device=app.params.device), optimizer, dist_info, None, n_kept=0)
            self.global_iter_step_app += 1
            return
        x01, sel, packed, ts, te, n_dev = st['x01'], st['sel'], st['packed'], st['t_starts'], st['t_ends'], st['n_dev']
        self._last_counts = (st['n_marched_dev'], n_dev)
        sig = st['sig0'] if st['sig0'] is not None else self.nerf.density_at(x01, sel, n_dev)
        n_net = app.mlp.n_params
        w16 = app.working_copy()
        feat = ops.hashgrid_fwd(app.grid, x01, w16[n_net:], n_dev=n_dev)
        rgbs = ops.mlp_fwd(app.mlp, w16[:n_net], feat, sel, n_dev=n_dev)
        n_rays = packed.shape[0]
        bg = None
        if self.renderer.bg_color == 'rand_noise':
            bg = rand['bg'] if 'bg' in rand else self._rand_cols(n_rays, 3, dist_info, x01.device)
        elif self.renderer.bg_color == 'white':
            bg = torch.ones(n_rays, 3, device=x01.device)
        if 'noise' not in rand:
            self._rand_cols(n_rays, 1, dist_info, x01.device)      # the distance noise draw of :193 (unused by this loss)
        # compositing forward -> colour loss -> compositing backward: ONE launch (perf_train_head_app)
        hd = ops.train_head_app(sig.reshape(-1).contiguous(), rgbs, ts, te, packed, bg, gt_colors, bs, tc.color_loss_weight, self.loss_scale)
        drgb = hd['d_rgb']
        self.last_losses['color_loss'] = lambda t=hd['color_terms'], s=1.0 / (3 * bs): t.sum() * s
        if sharded:
            self._dp_sharded_step(app, optimizer, dist_info, x01, w16, feat, sel, drgb, n_dev, st['n_marched_dev'])
            self.global_iter_step_app += 1
            return
        book = self._step_book(optimizer, dist_info, n_dev, st['n_marched_dev'])
        grad = self._field_grad(app, x01, w16, feat, sel, drgb, n_dev=n_dev, extra=extra, book=book)
        self._apply_grad(app, grad, optimizer, dist_info, None, n_kept=n_dev if n_dev is not None else x01.shape[0],
                         n_marched=st['n_marched_dev'], book=book)
        self.global_iter_step_app += 1

    def _can_fuse(self):
        tc = self.train_conf
        return self.fused_steps and tc.density_loss_weight <= 1e-7 and tc.depth_loss_weight > 1e-7 and tc.distortion_loss_weight > 1e-7

    def train_one_step_geo(self, optimizer, sup_pool, progress, rand=None, generator=None, prefetch_next=True):
        if self._can_fuse():
            return self._geo_step_fused(optimizer, sup_pool, progress, rand, generator, prefetch_next)
        tc = self.train_conf
        optimizer.zero_grad()
        pre = getattr(self, '_geo_pre', None) or self._geo_prefetch(sup_pool, rand, generator)
        self._geo_pre = None
        rays, gt_depths, bs, dist_info = pre['rays'], pre['gt_depths'], pre['bs'], pre['dist_info']
        st = pre['st']
        if st is None:
            st = self.renderer.stage_sample(self.nerf, self.estimator, rays.o, rays.d, rand)
        res = None if (st is None or st is False) else self.renderer.stage_composite(self.nerf, st, app_inference=True, rand=rand)
        if res is None:
            if dist_info[0] is not None:                      # keep the collective matched across ranks
                self._finish_step(self.nerf.geo_mlp.params.sum() * 0.0, self.nerf.geo_mlp, optimizer, dist_info)
            self.global_iter_step_geo += 1
            return
        loss = 0.
        if tc.depth_loss_weight > 1e-7:
            # mean over the GLOBAL batch: local sum / global count
            depth_loss = F.smooth_l1_loss(res['distance'], gt_depths, beta=1e-2, reduction='sum') / bs
            loss = loss + depth_loss * tc.depth_loss_weight
            self.last_losses['depth_loss'] = depth_loss.detach()
        if tc.distortion_loss_weight > 1e-7:
            mid = (res['t_ends'] + res['t_starts']) * .5
            sec = res['t_ends'] - res['t_starts']
            dist_loss = flatten_eff_distloss(res['weights'], mid, sec, res['ray_indices'], packed_info=res['packed_info'])
            if dist_info[2] > 1:                               # local /n_local -> global /bs
                dist_loss = dist_loss * ((res['ray_indices'][-1:].float() + 1.0) / bs).squeeze()
            if not self._capturing:
                self._ratio_dev.fill_(float(np.min([progress * 2., 1])))
            loss = loss + dist_loss * (tc.distortion_loss_weight * self._ratio_dev)
            self.last_losses['dist_loss'] = dist_loss.detach()
        if tc.density_loss_weight > 1e-7:
            rand_pts = (torch.rand(8192, 3, device=gt_depths.device) * 2. - 1.) * 0.99
            density_loss = self.nerf.query_density(rand_pts).mean()
            loss = loss + density_loss * tc.density_loss_weight
        overlap = (lambda: setattr(self, '_geo_pre', self._geo_prefetch(sup_pool, rand, generator))) \
            if (self.overlap_comm and prefetch_next and dist_info[0] is not None) else None
        self._finish_step(loss, self.nerf.geo_mlp, optimizer, dist_info, overlap)
        self.global_iter_step_geo += 1

    def train_one_step_app(self, optimizer, sup_pool, progress, rand=None, generator=None):
        if self.fused_steps and self.train_conf.color_loss_weight > 1e-7:
            return self._app_step_fused(optimizer, sup_pool, progress, rand, generator)
        tc = self.train_conf
        optimizer.zero_grad()
        rays, gt_colors, gt_depths, bs, dist_info = self._batch(sup_pool, generator)
        res = self.render_once(rays, ['rgb', 'distance', 'weights', 't_starts', 't_ends', 'trans', 'ray_indices'],
                               geo_inference=True, rand=rand)
        if (res is None) or (not res['is_valid']):
            if dist_info[0] is not None:
                self._finish_step(self.nerf.app_mlp.params.sum() * 0.0, self.nerf.app_mlp, optimizer, dist_info)
            self.global_iter_step_app += 1
            return
        loss = 0.
        if tc.color_loss_weight > 1e-7:
            color_loss = F.smooth_l1_loss(res['rgb'], gt_colors, beta=5e-2, reduction='sum') / (bs * 3)
            loss = loss + color_loss * tc.color_loss_weight
            self.last_losses['color_loss'] = color_loss.detach()
        self._finish_step(loss, self.nerf.app_mlp, optimizer, dist_info)
        self.global_iter_step_app += 1

    # ---- hipGraph capture of a whole training step (launch-bound inner loop) ---------------------------
    TRAIN_SAMPLES_PER_RAY = 128    # default per-ray sample capacity of sync-free / graph-captured training batches

    def make_graphed_step(self, kind, optimizer, sup_pool, warmup=3, schedule=None):
        """Capture train_one_step_{geo,app} (batch draw, sampling incl. the no-grad density pass and visibility compaction,
        both fields, compositing, losses, backward, Adam) into one hipGraph.  Sample arrays are capacity-sized
        (renderer.sample_capacity; default pixel_loss_batch_size * TRAIN_SAMPLES_PER_RAY) and every count stays on the
        device, so the reference's variable-count step is a fixed launch sequence.  Needs the fused Adam (device-side
        step / lr).  Under data parallelism (sharded mode, RCCL) the step's collectives are captured with it -- every rank
        must capture and replay in lockstep, and draw from identically seeded default generators.  `warmup` real steps run
        first (they ARE training steps; pass 0 when the caller has already run the step eagerly).
        schedule = (lrs, ratios, first): learning rate and distortion ramp min(2 progress, 1) of every iteration of the phase,
        the first replay being iteration `first` -- installed on the device (FusedAdam.load_schedule), so that replay() needs no
        argument and issues nothing but the graph.  Without it: replay(lr, progress) refreshes the two scalars per call."""
        assert isinstance(optimizer, FusedAdam), 'graph capture needs the fused Adam (device-side step/lr)'
        dist_info = self._dist()
        assert dist_info[0] is None or self._sharded(dist_info, optimizer), \
            'graphed data-parallel steps need the sharded exchange (dp_mode = "sharded", fixed-point grid backward)'
        assert self._can_fuse() if kind == 'geo' else self.fused_steps, 'graph capture covers the explicit step chains'
        if self.renderer.sample_capacity is None:
            self.renderer.sample_capacity = self.train_conf.pixel_loss_batch_size // dist_info[2] * self.TRAIN_SAMPLES_PER_RAY
        step_fn = self.train_one_step_geo if kind == 'geo' else self.train_one_step_app
        if warmup > 0:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    if kind == 'geo':
                        step_fn(optimizer, sup_pool, progress=0.0, prefetch_next=False)
                    else:
                        step_fn(optimizer, sup_pool, progress=0.0)
            torch.cuda.current_stream().wait_stream(side)
        if dist_info[0] is not None and self.dp_units == 'lagged':
            # The exchange's FIRST step has no previous statistics and takes the exact path (statistics all-gather + units
            # before the grid backward): captured, that prologue would be baked into every replay -- four collectives, and the
            # lagged units derived at the tail overwritten by the next replay's own.  The choice is made at capture time: make
            # sure it is the steady-state one.
            ex = getattr(optimizer.net, '_dp_exchange', None)
            if ex is None or ex.k.opt is not optimizer or not ex.have_units:
                raise RuntimeError('make_graphed_step: under data parallelism with lagged units the step must have run eagerly at least once '
                                   'with this optimizer before it is captured (warmup >= 1, or EAGER_HEAD eager iterations as _run_phase does)')
        if schedule is not None:
            lrs, ratios, first = schedule
            optimizer.load_schedule(lrs, ratios if kind == 'geo' else None, first, ratio_out=self._ratio_dev if kind == 'geo' else None)
        if kind == 'geo' and getattr(self, '_geo_pre', None) is not None:
            # A batch the previous eager step prefetched (data parallelism) would be baked into the graph as constants -- every
            # replay would train on it again.  The device generator is a pure function of (seed, counter): take the draw back,
            # the captured step repeats it.
            if not (self.device_rng and self._rng_counter is not None):
                raise RuntimeError('make_graphed_step: a prefetched batch is pending; run the last eager geometry step with prefetch_next=False')
            self._rng_counter -= 1
            self._geo_pre = None
        torch.cuda.synchronize()
        count = getattr(self, 'count_graph_nodes', False)
        graph = torch.cuda.CUDAGraph(keep_graph=True) if count else torch.cuda.CUDAGraph()
        self._capturing = optimizer.capturing = True
        iters = (self.global_iter_step_geo, self.global_iter_step_app)
        try:
            with _capture(graph):
                step_fn(optimizer, sup_pool, progress=0.0)
        finally:
            self._capturing = optimizer.capturing = False
            self.global_iter_step_geo, self.global_iter_step_app = iters       # (the capture pass executes nothing: no step was taken)
        if count:          # (bench.py: launches per replayed step)
            if not hasattr(self, 'graph_nodes'):
                self.graph_nodes = {}
            self.graph_nodes[kind] = _graph_node_count(graph)
            graph.instantiate()

        state = {'graph': graph, 'n': 0, 'counts': self._last_counts, 'capacity': self.renderer.sample_capacity,
                 'mode': optimizer.net.grid_grad_accum}

        def replay(lr=None, progress=None):
            if optimizer.sched_table is None:                    # no device-side schedule: refresh the two scalars
                optimizer.lr_dev.fill_(lr)
                if kind == 'geo':
                    self._ratio_dev.fill_(float(np.min([progress * 2., 1])))
            state['graph'].replay()
            state['n'] += 1
            if kind == 'geo':
                self.global_iter_step_geo += 1
            else:
                self.global_iter_step_app += 1
            if state['n'] % OVERFLOW_CHECK_EVERY == 0 and dist_info[0] is None:
                # one read-back of the device-side health counters (flagged / truncated steps were skipped on the device);
                # accumulation mode and capacity are baked into the graph: capture again when either changed.  (Data-parallel
                # replays stay in lockstep and leave the counters to the caller: NeRFScene._poll_health(force=True).)
                if self._poll_health(force=True)['recapture']:
                    new = self.make_graphed_step(kind, optimizer, sup_pool, warmup=0)
                    state.update(new.state)

        replay.graph = graph
        replay.state = state
        return replay

    # ---- hipGraph capture of a whole eval frame (BASELINE config 4: render_dense, core_exp_runner.py:223-246) ----------
    @torch.no_grad()
    def make_graphed_render(self, height, width, query_keys=('rgb', 'distance'), batch_size=32768, samples_per_ray=None):
        """One panorama frame -- rays generated from a device-resident pose, then ceil(H*W / batch_size) batches of
        NeRFScene.render_once (marching, no-grad density pass, visibility compaction, colour field, compositing, eval
        background) -- captured as ONE hipGraph with capacity-sized sample arrays and device-side counts.  Returns
        frame(pose [4,4]) -> {key: [H, W, C]} (tensors owned by the graph: valid until the next call).  After a replay the
        per-batch marched counts are checked against the capacity (one read-back per frame, after the frame is complete);
        a frame that did not fit is rendered again through a re-captured, larger graph, so results never depend on the
        capacity."""
        self.set_eval()
        dev = self.nerf.aabb.device
        n = height * width
        n_batches = (n + batch_size - 1) // batch_size
        state = {'per_ray': int(samples_per_ray or getattr(self, '_eval_spp_cap', self.EVAL_SAMPLES_PER_RAY))}
        pose_dev = torch.eye(4, dtype=torch.float32, device=dev)
        o_buf = torch.empty(height, width, 3, dtype=torch.float32, device=dev)
        d_buf = torch.empty(height, width, 3, dtype=torch.float32, device=dev)
        width_of = {'rgb': 3, 'distance': 1, 'opacities': 1}
        # one batch per frame (config 4): the frame's results ARE the renderer's tensors (owned by the graph's pool: same addresses on
        # every replay) and the marched count is read where the renderer left it -- no copy nodes; several batches: gathered by copies
        single = n_batches == 1
        outs = {} if single else {k: torch.empty(n, width_of[k], dtype=torch.float32, device=dev) for k in query_keys}
        marched = torch.zeros(n_batches, dtype=torch.int64, device=dev)
        sizes = torch.tensor([min((b + 1) * batch_size, n) - b * batch_size for b in range(n_batches)], dtype=torch.float32, device=dev)
        r = self.renderer

        def body():
            ops.pano_raygen_dev(pose_dev, height, width, out=(o_buf, d_buf))
            fo, fd = o_buf.view(-1, 3), d_buf.view(-1, 3)
            for b in range(n_batches):
                lo, hi = b * batch_size, min((b + 1) * batch_size, n)
                r.sample_capacity = (hi - lo) * state['per_ray']
                cur = self.render_once(Rays(fo[lo:hi], fd[lo:hi]), list(query_keys) + ['n_marched_dev'])
                if single:
                    for k in query_keys:
                        outs[k] = cur[k]
                    state['marched_ref'] = cur['n_marched_dev']
                    continue
                for k in query_keys:
                    outs[k][lo:hi].copy_(cur[k])
                marched[b:b + 1].copy_(cur['n_marched_dev'])

        def capture():
            saved = r.sample_capacity
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    body()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    body()
            finally:
                r.sample_capacity = saved
            return g

        state['graph'] = capture()

        def frame(pose):
            pose_dev.copy_(torch.as_tensor(pose, dtype=torch.float32, device=pose.device if torch.is_tensor(pose) else 'cpu').reshape(4, 4), non_blocking=True)
            while True:
                state['graph'].replay()
                worst = float(state['marched_ref'].item()) / n if single else float((marched.float() / sizes).max().item())
                if worst <= state['per_ray']:
                    break
                state['per_ray'] = self._eval_spp_cap = int(math.ceil(worst * 1.25))
                state['graph'] = capture()
            return {k: outs[k].view(height, width, -1) for k in query_keys}

        frame.state = state
        return frame

    @staticmethod
    def lr_at(optim_conf, progress):
        """nerf.py:300-311"""
        if progress < optim_conf.peak_at:
            lp = progress / optim_conf.peak_at
            return optim_conf.peak_lr * lp + optim_conf.init_lr * (1. - lp)
        lp = (progress - optim_conf.peak_at) / (1. - optim_conf.peak_at)
        return optim_conf.peak_lr * ((np.cos(lp * np.pi) + 1.) * .5 * (1. - optim_conf.lr_alpha) + optim_conf.lr_alpha)

    def update_lr(self, optimizer, optim_conf, progress):
        lr = self.lr_at(optim_conf, progress)
        for p in optimizer.param_groups:
            p['lr'] = lr

    # ---- visibility (nerf.py:321-358) ---------------------------------------------------------------------
    def get_pano_visibility_mask(self, sup_pool, rays: Rays):
        from .visibility import pano_visibility_mask
        distance = self.render(rays, query_keys=['distance'])['distance'].squeeze(-1)
        return pano_visibility_mask(rays.o, rays.d, distance, sup_pool.sup_infos)

    # ---- state (nerf.py:368-395) --------------------------------------------------------------------
    _STATE_KEY = '_perf_amd'

    def state_dict(self, sync=True):
        """nerf.py:374-380.  Under sharded data parallelism the fp32 master of a table slice is current only on its owner:
        the replicas are refreshed first (sync_params: one all-gather per network -- a COLLECTIVE, every rank must call
        state_dict() at the same point, as with any checkpoint of a data-parallel job).  The reference's own
        `if rank == 0: save(...)`-style checkpointing on ONE rank would therefore hang: call sync_params() on every rank first
        (train_one_episode does so at the end of each episode) and then state_dict(sync=False) on the saving rank
        (INTEGRATION.md 3)."""
        if sync:
            self.sync_params()
        # The fixed-point headroom feedback of the two grid gradients makes results depend on the call history: it travels
        # with the checkpoint, under a private TOP-LEVEL key -- the reference's loader reads only 'render' / 'nerf' /
        # 'estimator' (nerf.py:368-380) and its strict nerf.load_state_dict would reject a key inside 'nerf'.
        return {'render': self.renderer.state_dict(), 'nerf': self.nerf.state_dict(),
                'estimator': self.estimator.state_dict(),
                self._STATE_KEY: {'geo_headroom': self.nerf.geo_mlp.headroom_state().detach().clone(),
                                  'app_headroom': self.nerf.app_mlp.headroom_state().detach().clone()}}

    def load_state_dict(self, state_dict):
        self.renderer.load_state_dict(state_dict['render'])
        self.nerf.load_state_dict(state_dict['nerf'])
        self.estimator.load_state_dict(state_dict['estimator'])
        self.estimator._bits = None
        extra = state_dict.get(self._STATE_KEY)            # (absent in checkpoints the reference or earlier rounds wrote)
        if extra:
            for name, net in (('geo', self.nerf.geo_mlp), ('app', self.nerf.app_mlp)):
                if name + '_headroom' in extra:
                    net.headroom_state().copy_(extra[name + '_headroom'].to(net.params.device))

    def set_train(self):
        self.nerf.train(); self.estimator.train(); self.renderer.train()

    def set_eval(self):
        self.nerf.eval(); self.estimator.eval(); self.renderer.eval()


def psnr(pred, gt):
    """-10 log10(mean((pred-gt)^2))  (the reference never computes it)."""
    return float(-10.0 * torch.log10(torch.mean((pred.float() - gt.float()) ** 2)))
