"""ctypes binding of libperf_hip.so (the C ABI declared in include/perf_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, this
module raises.  Build the library with `python -m perf_amd.build` (or __graft_entry__.build()).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libperf_hip.so')

ABI_VERSION = 13         # PERF_ABI_VERSION of include/perf_hip.h this binding was written against
MAX_LEVELS = 24
DTYPE_BF16, DTYPE_FP16 = 0, 1
ACT_NONE, ACT_SIGMOID, ACT_EXP = 0, 1, 2
INTERP_LINEAR, INTERP_SMOOTHSTEP = 0, 1
LAYOUT_TCNN, LAYOUT_LINE_LOCAL, LAYOUT_LINE_OVERLAP = 0, 1, 2
LATTICE = {'single': 0, 'repeated': 1, None: 1}          # PERF_LATTICE_*; None = the default
DEFAULT_LATTICE = 'repeated'                 # t_{k+1} = fl(t_k + step): how nerfacc's traverse_grids is understood to march (include/perf_hip.h)


class GridDesc(Structure):
    _fields_ = [('n_levels', c_int32), ('interpolation', c_int32),
                ('scale', c_float * MAX_LEVELS), ('res', c_uint32 * MAX_LEVELS), ('size', c_uint32 * MAX_LEVELS),
                ('offset', c_uint64 * MAX_LEVELS), ('hashed', c_uint32 * MAX_LEVELS),
                ('layout', c_int32), ('sb_shift', c_uint32 * 3), ('local', c_uint32 * MAX_LEVELS), ('nsx', c_uint32 * MAX_LEVELS),
                ('nsxy', c_uint32 * MAX_LEVELS)]


class MlpDesc(Structure):
    _fields_ = [('n_levels', c_int32), ('n_hidden_layers', c_int32), ('n_out', c_int32), ('out_act', c_int32),
                ('exp_shift', c_float)]


class StepBook(Structure):       # perf_step_book: the arguments of perf_step_bookkeeping as a POD block (perf_field_bwd_book)
    _fields_ = [('step_dev', c_void_p), ('gate_dev', c_void_p), ('counters', c_void_p), ('n_marched_dev', c_void_p), ('n_kept_dev', c_void_p),
                ('capacity', c_int64), ('overflow_flag', c_void_p), ('remote_flags', c_void_p), ('eff_gate_out', c_void_p),
                ('schedule', c_void_p), ('iter_dev', c_void_p), ('lr_out', c_void_p), ('ratio_out', c_void_p), ('n_schedule', c_int32),
                ('overflow_redone', c_int32)]


class PerfError(RuntimeError):
    pass


P = c_void_p
LOSS_SCALARS = 256       # PERF_LOSS_SCALARS
STEP_COUNTERS = 8        # PERF_STEP_COUNTERS
HEADROOM_STATE_WORDS = 2 * MAX_LEVELS + 8    # PERF_HEADROOM_STATE_WORDS
DP_STATS = 64            # PERF_DP_STATS
DP_SLOT = 80             # PERF_DP_SLOT
_SIGS = {
    'perf_version': (c_int, []),
    'perf_last_error': (c_char_p, []),
    'perf_sizeof_grid_desc': (c_int64, []),
    'perf_sizeof_mlp_desc': (c_int64, []),
    'perf_sizeof_step_book': (c_int64, []),
    'perf_cast_params': (c_int, [P, P, c_int64, c_int, P]),
    'perf_adam_step': (c_int, [P, P, P, P, P, c_int64, c_int, c_int32, c_float, c_float, c_float, c_float, c_int, P]),
    'perf_adam_step_dev': (c_int, [P, P, P, P, P, c_int64, c_int, P, P, P, c_float, c_float, c_float, c_int, P, P]),
    'perf_step_bookkeeping': (c_int, [P, P, P, P, P, c_int64, P, P, c_int32, P, P, c_int32, P, P, P, P]),
    'perf_points_from_rays': (c_int, [P, P, P, P, P, POINTER(c_float), P, P, c_int64, P, P]),
    'perf_points_normalize': (c_int, [P, POINTER(c_float), P, P, c_int64, P]),
    'perf_hashgrid_fwd': (c_int, [POINTER(GridDesc), P, P, P, c_int64, P, c_int, P]),
    'perf_hashgrid_fwd2': (c_int, [POINTER(GridDesc), P, P, P, P, P, c_int64, c_int, P]),
    'perf_hashgrid_fwd_f32': (c_int, [POINTER(GridDesc), P, P, P, c_int64, P]),
    'perf_hashgrid_bwd_workspace_bytes': (c_int64, [POINTER(GridDesc), c_int64]),
    'perf_hashgrid_bwd': (c_int, [POINTER(GridDesc), P, P, P, c_int64, P, c_int, P, P, P, P, c_int, P, P, c_int64, P]),
    'perf_dp_stats_pack': (c_int, [P, P, P, c_int64, P, P]),
    'perf_dp_units': (c_int, [POINTER(GridDesc), P, c_int32, P, P, P, c_int32, P]),
    'perf_dp_slot_pack': (c_int, [P, P, P, c_int64, P, P, c_int64, c_int32, c_int32, P, P]),
    'perf_dp_slot_unpack': (c_int, [P, c_int32, P, P, P, P]),
    'perf_fixed_unfix': (c_int, [POINTER(GridDesc), P, c_int64, c_int64, P, P, P, P]),
    'perf_hashgrid_corners': (c_int, [POINTER(GridDesc), P, P, c_int64, P]),
    'perf_hashgrid_bwd_input': (c_int, [POINTER(GridDesc), P, P, P, P, c_int64, P]),
    'perf_hashgrid_bwd_bwd_input': (c_int, [POINTER(GridDesc), P, P, P, P, P, P, c_int64, P]),
    'perf_hashgrid_bwd_bwd_param': (c_int, [POINTER(GridDesc), P, P, P, P, c_int64, P]),
    'perf_mlp_fwd': (c_int, [POINTER(MlpDesc), P, P, P, P, c_int64, P, c_int, P]),
    'perf_field_infer_scratch_bytes': (c_int64, [POINTER(GridDesc), c_int64]),
    'perf_field_infer': (c_int, [POINTER(GridDesc), POINTER(MlpDesc), P, P, P, P, P, c_int64, P, P, c_int64, P, c_int, P]),
    'perf_field_bwd_workspace_bytes': (c_int64, [POINTER(GridDesc), POINTER(MlpDesc), c_int64, P, P, P]),
    'perf_field_bwd': (c_int, [POINTER(GridDesc), POINTER(MlpDesc), P, P, P, P, c_int64, P, P, P, c_int32, c_int32, P, P, P, c_int64, c_int64, P, c_int, P]),
    'perf_field_bwd_book': (c_int, [POINTER(GridDesc), POINTER(MlpDesc), P, P, P, P, c_int64, P, P, P, P, P, P, c_int64, c_int64, P, c_int, POINTER(StepBook), P]),
    'perf_mlp_bwd_workspace_bytes': (c_int64, [POINTER(MlpDesc), c_int64]),
    'perf_mlp_bwd': (c_int, [POINTER(MlpDesc), P, P, P, c_int64, P, P, P, P, P, P, c_int64, c_int64, P, c_int, P]),
    'perf_pano_raygen': (c_int, [POINTER(c_float), c_int32, c_int32, c_int32, c_int32, P, P, P]),
    'perf_pano_raygen_dev': (c_int, [P, c_int32, c_int32, c_int32, c_int32, P, P, P]),
    'perf_occ_pack_bits': (c_int, [P, P, c_int64, P]),
    'perf_occ_jitter_points': (c_int, [c_uint64, c_uint64, c_int64, c_int64, c_int32, POINTER(c_float), P, P]),
    'perf_occ_ema_update': (c_int, [P, P, c_int64, c_float, P, P]),
    'perf_occ_threshold': (c_int, [P, c_int64, P, c_float, P, P]),
    'perf_occ_mask_words': (c_int64, [c_int32]),
    'perf_occ_lattice_table_len': (c_int64, [c_int32]),
    'perf_occ_lattice_table': (c_int, [c_float, c_float, c_int32, c_int32, P, P]),
    'perf_occ_lattice_runs_len': (c_int64, [c_int64]),
    'perf_occ_lattice_runs': (c_int, [P, c_float, c_float, c_int64, c_float, c_int32, P, P]),
    'perf_occ_march_count': (c_int, [P, P, P, c_float, c_float, c_int64, P, P, c_int32, POINTER(c_float), c_float, c_float, c_int32, c_int32, P, P, P, P]),
    'perf_occ_march_count_head': (c_int, [P, P, P, c_float, c_float, c_int64, P, P, c_int32, POINTER(c_float), c_float, c_float, c_int32, c_int32, P, P, P,
                                  c_int32, P, P, P, P, POINTER(c_float), P, P, P]),
    'perf_occ_coarse_words': (c_int64, [c_int32]),
    'perf_occ_build_coarse': (c_int, [P, c_int32, P, P]),
    'perf_scan_workspace_bytes': (c_int64, [c_int64]),
    'perf_exclusive_scan_i32': (c_int, [P, P, P, c_int64, c_int64, P, P, c_int64, P]),
    'perf_occ_march_write': (c_int, [P, c_float, c_float, c_int64, c_float, c_int32, c_int32, P, P, P, P, c_int64, P, P, P, P, P]),
    'perf_occ_march_write_points': (c_int, [P, c_float, c_float, c_int64, c_float, c_int32, c_int32, P, P, P, P, c_int64, P, P, P, P, P, P, POINTER(c_float), P, P, c_int32, P]),
    'perf_head_tail_counts': (c_int, [P, c_int64, c_int32, P, P, P]),
    'perf_visibility_count2': (c_int, [P, P, P, P, P, P, P, P, c_int64, c_float, P, P]),
    'perf_compact_prefix2': (c_int, [P] * 14 + [c_int64, c_int64] + [P] * 7 + [P, c_int64, P, c_int64, P, c_int64, c_int32, P]),
    'perf_visibility_count': (c_int, [P, P, P, P, c_int64, c_float, P, P, P, c_int32, P, P]),
    'perf_compact_prefix': (c_int, [P, P, P, c_int64, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int64, P, c_int64, c_int32, P, P]),
    'perf_composite_fwd': (c_int, [P, P, P, P, P, c_int64, P, P, P, P, P, P, P]),
    'perf_composite_bwd': (c_int, [P, P, P, P, c_int64, P, P, P, P, P, P, P, P, P, P, P]),
    'perf_render_finish_eval': (c_int, [P, P, P, c_int64, P, P]),
    'perf_accumulate_fwd': (c_int, [P, P, P, c_int64, c_int32, P, P]),
    'perf_pack_info': (c_int, [P, c_int64, c_int64, P, P]),
    'perf_distloss_fwd': (c_int, [P, P, P, P, c_int64, P, P]),
    'perf_distloss_bwd': (c_int, [P, P, P, P, c_int64, c_float, P, P, P]),
    'perf_geo_loss': (c_int, [P, P, P, P, P, P, c_int64, c_int64, c_float, c_float, P, c_float, P, P, P, P, P]),
    'perf_app_loss': (c_int, [P, P, P, P, c_int64, c_int64, c_float, c_float, P, P, P]),
    'perf_train_head_geo': (c_int, [P, P, P, P, P, c_int64, c_int64, P, P, c_int64, c_float, c_float, P, c_float, P, P, P, P, P, P, P, P, P, P]),
    'perf_train_head_app': (c_int, [P, P, P, P, P, c_int64, c_int64, P, P, c_int64, c_float, c_float, P, P, P, P, P, P, P, P]),
    'perf_composite_distloss_fwd': (c_int, [P, P, P, P, P, c_int64, P, P, P, P, P, P, P]),
    'perf_composite_distloss_bwd': (c_int, [P, P, P, P, c_int64, P, P, P, P, P, P, c_float, P, P, P]),
    'perf_gather_supervision': (c_int, [P, c_int64, P, P, P, P, P, P, P, P, P, P, P]),
    'perf_draw_train_batch': (c_int, [c_uint64, P, P, c_int64, c_int64, c_int64, c_int64] + [P] * 14 + [P]),
    'perf_pdf_resample': (c_int, [P, P, P, c_int64, c_int32, c_int32, P, P]),
    'perf_pano_reproject': (c_int, [P, c_int64, POINTER(c_float), P, c_int32, c_int32, c_int32, c_float, P, P]),
    'perf_morph_binary': (c_int, [P, P, c_int32, c_int32, POINTER(c_uint32), c_int32, c_int32, c_int32, P]),
    'perf_occ_splat': (c_int, [P, P, P, c_int64, c_int32, P, P]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises PerfError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: it carries its own libamdhip64, and the HIP runtime that owns the device
    # context has to be the one this library's kernels register with (one runtime per process).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PerfError(f'{LIB_PATH} not found: build it with `python -m perf_amd.build` '
                        '(there is no CPU fallback for the HIP path)')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # the binding and the library must agree on the ABI version and on the layout of the POD descriptors
    if lib.perf_version() != ABI_VERSION:
        raise PerfError(f'{LIB_PATH} has ABI version {lib.perf_version()}, this binding expects {ABI_VERSION}: '
                        'rebuild with `python -m perf_amd.build --force`')
    if (lib.perf_sizeof_grid_desc() != ctypes.sizeof(GridDesc) or lib.perf_sizeof_mlp_desc() != ctypes.sizeof(MlpDesc)
            or lib.perf_sizeof_step_book() != ctypes.sizeof(StepBook)):
        raise PerfError('descriptor layout mismatch between perf_amd/_lib.py and include/perf_hip.h')
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what=''):
    if rc != 0:
        msg = load().perf_last_error()
        raise PerfError(f'{what} failed (code {rc}): {msg.decode() if msg else ""}')


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)
