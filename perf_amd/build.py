"""Build libperf_hip.so (gfx950) in-tree with hipcc.  `python -m perf_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libperf_hip.so')
# (the MLP kernels are instantiated in six units of their own: one unit took 65 s, the longest of them now takes ~15 s)
SOURCES = ['mlp_bwd_bf16_nh2.hip', 'mlp_bwd_fp16_nh2.hip', 'mlp_bwd_bf16_nh1.hip', 'mlp_bwd_fp16_nh1.hip', 'mlp_fwd_bf16.hip', 'mlp_fwd_fp16.hip',
           'hashgrid_bwd.hip', 'hashgrid_fwd.hip', 'hashgrid_aux.hip', 'march.hip', 'misc.hip', 'composite.hip', 'mlp.hip', 'visibility.hip']
HEADERS = ['common.hpp', 'grid_device.hpp', 'grid_fixed_point.hpp', 'mlp_reduce_device.hpp', 'step_book_device.hpp', 'mlp_device.hpp']
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (no v_accvgpr_read per accumulator register before the epilogues)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
         '-mllvm', '-amdgpu-mfma-vgpr-form=1']


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, '..', 'include', 'perf_hip.h')]
    deps = [os.path.join(CSRC, s) for s in SOURCES] + hdrs
    if not force and _newer(LIB, deps):
        return LIB
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        own = [h for h in hdrs if src.startswith('mlp') or not h.endswith('mlp_device.hpp')]       # (only the mlp units include mlp_device.hpp)
        if not force and _newer(obj, [os.path.join(CSRC, src)] + own):
            return obj
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr:
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(os.cpu_count() or 4, 4))) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
