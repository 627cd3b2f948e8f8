"""Host-side mirror of PeRF's pose samplers (modules/pose_sampler/circle_pose_sampler.py:44-118,
dense_travel_pose_sampler.py:26-116).  They run once per scene on the host (scipy filters, a
10,000-step annealed tour) and only produce the 4x4 poses the ray-generation kernel consumes, so they stay on the
CPU; the reference's hard-coded .cuda() hops are dropped.  Pinned on golden poses made by the reference
(tests/golden/poses.npz).

The dense trajectory is a pure function of (anchor positions, n_dense_poses, dir_bias_ratio, numpy's global RNG state): its
10,000 sequential accept/reject decisions consume that stream step by step and compare float32 tour lengths summed by
torch.sum, so it cannot be vectorised or moved to the GPU without changing the trajectory the golden poses pin.  It is
taken OFF THE CRITICAL PATH instead: `DenseTravelPoseSampler.start(...)` runs it in a forked worker
process (CPU only) as soon as the anchors exist -- i.e. while the scene still trains -- and `.result()` hands back the
sampler plus the RNG state the sequential call would have left behind; results are memoised per input key.  The frame
loop of render_dense then starts without waiting (perf_amd/traverse.py, tools/render_dense.py reports both wall times)."""
import hashlib
import os
import time
import multiprocessing as mp

import numpy as np
import torch
import torch.nn.functional as F
from scipy.ndimage import gaussian_filter1d, minimum_filter1d


def _odd(n):
    return n // 2 * 2 + 1


def _equator_dirs(width):
    """unit directions of the panorama's middle row, pixel centres (camera_utils.py:113-147 at y = 0.5)"""
    x = torch.linspace(.5 / width, 1. - .5 / width, width, device='cpu')
    alpha = -(x - .5) * 2. * np.pi
    beta = torch.zeros(width, device='cpu')
    return torch.stack([torch.cos(alpha) * torch.cos(beta), torch.sin(alpha) * torch.cos(beta), torch.sin(beta)], -1)


def resample_closed_curve(pts: torch.Tensor) -> torch.Tensor:
    """Re-parametrise a closed polyline by arc length: 128x linear upsampling, cumulative chord length, n equally
    spaced parameters (both samplers use this helper)."""
    n = len(pts)
    fine = F.interpolate(pts.t()[None], size=n * 128, mode='linear')[0].t()
    seg = torch.linalg.norm(torch.roll(fine, -1, 0) - fine, 2, -1)
    cum = torch.cumsum(seg, 0)
    cum = cum / cum[-1]
    return fine[torch.searchsorted(cum, torch.linspace(0., 1. - 1. / n, n, device=pts.device))]


def look_at(to_vec: torch.Tensor) -> torch.Tensor:
    """camera_utils.py:79-95 with the default up vector (0,0,1): columns (right, down, forward)."""
    up = torch.zeros_like(to_vec); up[:, 2] = 1.
    fwd = to_vec / torch.linalg.norm(to_vec, 2, -1, True)
    right = torch.linalg.cross(-up, fwd)
    right = right / torch.linalg.norm(right, 2, -1, True)
    down = torch.linalg.cross(fwd, right)
    return torch.stack([right, down, fwd], 2)


class CirclePoseSampler:
    """Anchor camera positions on rings inside the free space around the panorama centre."""

    def __init__(self, distance_map, traverse_ratios, n_anchors_per_ratio, test_z_min_max=(0., 0.)):
        dm = distance_map.detach().cpu().numpy() if torch.is_tensor(distance_map) else np.asarray(distance_map)
        dm = dm.squeeze()
        h, w = dm.shape
        rows = torch.linspace(.5 / h, 1. - .5 / h, h, device='cpu')
        beta = (-(rows - .5) * np.pi).numpy()
        horiz = dm * np.cos(beta)[:, None]                       # distance projected on the horizontal plane
        band = horiz[h // 2 - 10: h // 2 + 10].copy()
        band[band < 1e-5] = 1e9
        free = band.min(axis=0)
        for i in range(1, w):                                     # fill invalid columns from the left, then from the right
            if free[i] > 1e8:
                free[i] = free[i - 1]
        for i in range(w - 2, -1, -1):
            if free[i] > 1e8:
                free[i] = free[i + 1]
        shrunk = minimum_filter1d(free, size=_odd(w // 16), mode='wrap')
        smooth = gaussian_filter1d(shrunk, sigma=_odd(w // 8), mode='wrap')
        ring = gaussian_filter1d(shrunk, sigma=_odd(w // 64), mode='wrap')
        dirs = _equator_dirs(w)
        ring_t = torch.from_numpy(ring)
        z_lo, z_hi = test_z_min_max
        anchors = []
        for i, ratio in enumerate(traverse_ratios):
            loop = resample_closed_curve(dirs * ring_t[:, None] * ratio)
            n = n_anchors_per_ratio[i]
            pos = torch.linspace(.5 / n, 1. - .5 / n, n, device='cpu') + (0. if i % 2 == 0 else .5 / n)
            pts = loop[(pos * w).to(torch.long).clip(0, w - 1)].clone()
            for j in range(len(pts)):
                pts[j, 2] = z_lo if (i + j) % 2 == 0 else z_hi
            anchors.append(pts)
        self.anchor_pts = torch.cat(anchors, 0).float()
        self.traverse_pts = resample_closed_curve(dirs * torch.from_numpy(smooth)[:, None] * .3)
        self.n_anchors = self.n_poses = len(self.anchor_pts)

    def sample_pose(self, idx):
        pose = torch.eye(4, device='cpu')
        pose[:3, 3] = self.anchor_pts[idx]
        return pose


def _annealed_tour(positions: torch.Tensor, n_steps=10000):
    """Random pair swaps with an annealed acceptance ratio (1 - step/n)^5; consumes numpy's global RNG in the
    reference's order (two randint per step, one rand only when the swap does not shorten the tour).
    (dense_travel_pose_sampler.py:26-48 recomputes 23 difference vectors and norms with five tiny torch ops per step;
    here the pairwise distances ||p_i - p_j|| are formed once with the same operations, a step gathers 23 of them and adds
    them with the same torch.sum -- the same float32 tour lengths, hence the same accept / reject decisions, in a quarter
    of the time.)"""
    n = len(positions)
    pos = positions.detach().cpu().float()
    dmat = torch.linalg.norm(pos[:, None, :] - pos[None, :, :], 2, -1).numpy()        # [n, n] float32
    order = np.arange(n)
    best = 1e8
    for it in range(n_steps):
        a = np.random.randint(n); b = np.random.randint(n)
        cand = order.copy()
        cand[a], cand[b] = order[b], order[a]
        length = float(torch.from_numpy(dmat[cand[:-1], cand[1:]]).sum())
        if length < best or np.random.rand() < (1. - it / n_steps) ** 5:
            order, best = cand, length
    return torch.from_numpy(order)


_DENSE_CACHE = {}


def _rng_digest():
    st = np.random.get_state()
    return hashlib.sha256(st[1].tobytes() + str((st[0], st[2], st[3], st[4])).encode()).hexdigest()


def _dense_worker(conn, sparse_poses, n_dense_poses, dir_bias_ratio, rng_state):
    """Forked worker (never touches the GPU): the sequential construction, from the parent's RNG state."""
    try:
        torch.set_num_threads(1)
        np.random.set_state(rng_state)
        d = DenseTravelPoseSampler(_Poses(sparse_poses), n_dense_poses, dir_bias_ratio, _cache=False)
        conn.send((d.sample_poses.numpy(), np.random.get_state(), None))
    except Exception as e:       # noqa: BLE001
        conn.send((None, None, repr(e)))
    conn.close()


class _Poses:
    """Anything with n_poses / sample_pose(i), from a stack of 4x4 poses (what crosses the process boundary)."""

    def __init__(self, poses):
        self.poses = poses
        self.n_poses = len(poses)

    def sample_pose(self, idx):
        return self.poses[idx]


class DensePoseFuture:
    """Handle of a dense trajectory being computed in a worker process (DenseTravelPoseSampler.start)."""

    def __init__(self, key, proc=None, conn=None, ready=None, fallback=None):
        self.key, self.proc, self.conn, self._ready = key, proc, conn, ready
        self._fallback = fallback          # (sparse poses, n_dense_poses, dir_bias_ratio, numpy RNG state at start())

    def done(self):
        return self._ready is not None or self.conn.poll()

    # Ceiling of the wait for a LIVING worker (a dead or failed one is noticed at once): generous and configurable -- the in-line
    # fallback of a slow but healthy worker would pay the whole construction a second time.
    TIMEOUT_S = float(os.environ.get('PERF_DENSE_POSES_TIMEOUT_S', '1800'))

    def result(self, timeout=None):
        """The sampler; numpy's global RNG is left in the state the sequential construction would have left it in.  The
        worker was forked from a process that holds HIP / RCCL / OpenMP state (a fork-after-threads hazard the single-threaded
        child sidesteps, but cannot rule out): should it die or fail -- or still be running after `timeout` seconds (default:
        PERF_DENSE_POSES_TIMEOUT_S, 30 min) -- the trajectory is built HERE, sequentially, from the RNG state saved at
        start(): same poses, same RNG afterwards.  A slow but living worker is waited for (polled every 0.25 s)."""
        if self._ready is None:
            poses = rng_state = None
            why = None
            limit = self.TIMEOUT_S if timeout is None else timeout
            t_start = time.monotonic()
            try:
                while True:
                    if self.conn.poll(0.25):
                        poses, rng_state, err = self.conn.recv()
                        if err is not None:
                            why, poses = f'worker failed: {err}', None
                        break
                    if self.proc is not None and not self.proc.is_alive() and not self.conn.poll(0):
                        why = 'worker died'
                        break
                    if time.monotonic() - t_start >= limit:
                        why = f'worker still running after {limit:.0f} s'
                        break
            except (EOFError, OSError) as e:
                why = f'worker died ({type(e).__name__})'
            if self.proc is not None:
                if why is not None and self.proc.is_alive():
                    self.proc.terminate()
                self.proc.join(5.0)
            if poses is None:
                if self._fallback is None:
                    raise RuntimeError(f'dense pose sampler: {why}')
                import warnings
                warnings.warn(f'perf_amd: dense pose sampler {why}; building the trajectory in line')
                sparse, n_dense, bias, state0 = self._fallback
                np.random.set_state(state0)
                seq = DenseTravelPoseSampler.__new__(DenseTravelPoseSampler)
                seq._build(sparse, n_dense, bias)
                poses, rng_state = seq.sample_poses.numpy(), np.random.get_state()
            _DENSE_CACHE[self.key] = (torch.from_numpy(np.asarray(poses)).clone(), rng_state)
            self._ready = _DENSE_CACHE[self.key]
        poses, rng_state = self._ready
        np.random.set_state(rng_state)
        return DenseTravelPoseSampler._from_poses(poses)


class DenseTravelPoseSampler:
    """Smooth dense trajectory through the sparse anchors with look-ahead orientations."""

    @staticmethod
    def _key(sparse, n_dense_poses, dir_bias_ratio):
        return (hashlib.sha256(sparse.numpy().tobytes()).hexdigest(), int(n_dense_poses), float(dir_bias_ratio), _rng_digest())

    @classmethod
    def _from_poses(cls, poses):
        self = cls.__new__(cls)
        self.sample_poses = poses.clone()
        self.n_poses = len(poses)
        return self

    @classmethod
    def start(cls, sparse_pose_sampler, n_dense_poses, dir_bias_ratio=-1):
        """Begin the construction in a forked worker process and return at once -> DensePoseFuture.  Call it as soon as the
        anchors exist (before / while the scene trains); the worker starts from the CURRENT global numpy RNG state, and
        .result() restores the state the sequential call would have left, so nothing downstream changes -- provided nobody
        draws from numpy's global RNG in between."""
        sparse = torch.stack([sparse_pose_sampler.sample_pose(i) for i in range(sparse_pose_sampler.n_poses)], 0).float().cpu()
        key = cls._key(sparse, n_dense_poses, dir_bias_ratio)
        if key in _DENSE_CACHE:
            return DensePoseFuture(key, ready=_DENSE_CACHE[key])
        ctx = mp.get_context('fork')
        parent, child = ctx.Pipe(duplex=False)
        state0 = np.random.get_state()
        proc = ctx.Process(target=_dense_worker, args=(child, sparse, n_dense_poses, dir_bias_ratio, state0), daemon=True)
        proc.start()
        child.close()
        return DensePoseFuture(key, proc, parent, fallback=(sparse, n_dense_poses, dir_bias_ratio, state0))

    def __init__(self, sparse_pose_sampler, n_dense_poses, dir_bias_ratio=-1, _cache=True):
        sparse = torch.stack([sparse_pose_sampler.sample_pose(i) for i in range(sparse_pose_sampler.n_poses)], 0).float().cpu()
        key = self._key(sparse, n_dense_poses, dir_bias_ratio) if _cache else None
        if key is not None and key in _DENSE_CACHE:             # same anchors, same RNG state: same trajectory, same RNG afterwards
            poses, rng_state = _DENSE_CACHE[key]
            np.random.set_state(rng_state)
            self.sample_poses, self.n_poses = poses.clone(), len(poses)
            return
        self._build(sparse, n_dense_poses, dir_bias_ratio)
        if key is not None:
            _DENSE_CACHE[key] = (self.sample_poses.clone(), np.random.get_state())

    def _build(self, sparse, n_dense_poses, dir_bias_ratio):
        tour = sparse[_annealed_tour(sparse[:, :3, 3])][:, :3, 3]
        total = n_dense_poses * 50
        seg_len = torch.linalg.norm(tour[1:] - tour[:-1], 2, -1, True)
        counts = torch.round(total * seg_len / seg_len.sum()).to(torch.int64)
        chunks = []
        for i in range(len(counts)):
            k = counts[i].item()
            t = torch.linspace(.5 / k, 1. - .5 / k, k, device='cpu')
            chunks.append(tour[i][None] * (1. - t)[:, None] + tour[i + 1][None] * t[:, None])
        pts = resample_closed_curve(torch.cat(chunks, 0))[::50].numpy()
        for a in range(3):
            pts[:, a] = gaussian_filter1d(pts[:, a], sigma=20)
        pts = torch.from_numpy(pts)
        self.sample_poses = torch.eye(4, device='cpu')[None].repeat(len(pts), 1, 1)
        self.sample_poses[:, :3, 3] = pts
        self.n_poses = len(pts)
        fwd = pts.clone()
        fwd[:-1] = pts[1:] - pts[:-1]
        fwd[-1] = fwd[-2]
        for a in range(3):
            fwd[:, a] = torch.from_numpy(gaussian_filter1d(fwd[:, a].numpy(), sigma=30))
        fwd = fwd / torch.linalg.norm(fwd, 2, -1, True)
        up = torch.zeros_like(fwd); up[:, 2] = 1.
        left = torch.linalg.cross(up, fwd)
        left = left / torch.linalg.norm(left, 2, -1, True)
        fwd = fwd + dir_bias_ratio * left
        fwd = fwd / torch.linalg.norm(fwd, 2, -1, True)
        self.sample_poses[:, :3, :3] = look_at(fwd)

    def sample_pose(self, idx):
        return self.sample_poses[idx]
