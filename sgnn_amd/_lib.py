"""ctypes binding of libsgnn_hip.so (C ABI declared in include/sgnn_hip.h).

The product path has no CPU fallback: if the shared library is missing, or no MI355X is
visible, every operator raises.  PyTorch is used only for device memory, streams and autograd
bookkeeping; no torch type crosses the ABI (raw device pointers + sizes + the HIP stream).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SGNN_LIB') or os.path.join(_HERE, 'lib', 'libsgnn_hip.so')   # SGNN_LIB: kernel-variant builds (measurements)

c_i32, c_i64, c_f32, c_vp, c_cp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_char_p

# name -> (restype, argtypes); mirrors include/sgnn_hip.h one to one
PROTOTYPES = {
    'sgnn_last_error': (c_cp, []),
    'sgnn_version': (c_i32, []),
    'sgnn_arch': (c_cp, []),
    'sgnn_tune_set': (c_i64, [c_cp, c_i64]),
    'sgnn_tune_get': (c_i64, [c_cp]),
    'sgnn_tune_names': (c_cp, []),
    'sgnn_tune_current': (c_vp, []),
    'sgnn_hash_capacity': (c_i64, [c_i64]),
    'sgnn_coords_from_i64': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'sgnn_coords_to_i64': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_hash_build': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_hash_lookup': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_rulebook_subm3': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_rulebook_subm3_multi': (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'sgnn_rulebook_subm3_dense': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_rulebook_subm3_volume': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_down2_ws_bytes': (c_i64, [c_i64]),
    'sgnn_rulebook_down2': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_down2_chain_ws_bytes': (c_i64, [c_i64]),
    'sgnn_down2_chain': (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_down2_chain_tables_ws_bytes': (c_i64, [c_i64, c_i32]),
    'sgnn_down2_chain_tables': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_i64, c_vp]),
    'sgnn_down2_tables': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_conv_fwd': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp]),
    'sgnn_conv_fwd_ex': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    'sgnn_conv_bwd_weight_ex': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    'sgnn_conv_stats_blocks': (c_i64, [c_i64]),
    'sgnn_conv_fwd_epi': (c_i32, [c_vp, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp]),
    'sgnn_conv_bwd_weight_ws_bytes': (c_i64, [c_i64, c_i32, c_i32, c_i32]),
    'sgnn_conv_bwd_weight': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp]),
    'sgnn_conv_bwd_fused_ws_bytes': (c_i64, [c_i64, c_i32, c_i32]),
    'sgnn_conv_bwd_fused_supported': (c_i32, [c_i64, c_i32, c_i32, c_i32]),
    'sgnn_conv_bwd_fused': (c_i32, [c_vp, c_i64, c_i32, c_i64, c_vp, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_bn_ws_bytes': (c_i64, [c_i64, c_i32]),
    'sgnn_bn_fwd': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_bn_bwd': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_bn_bwd_add': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_bn_fwd_ex': (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'sgnn_bn_bwd_ex': (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'sgnn_copy_multi': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp]),
    'sgnn_gather_rows': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_gather_rows_dn': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_scatter_rows': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_gather_sum': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp]),
    'sgnn_repeat_rows': (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    'sgnn_sum_groups': (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    'sgnn_concat_rows': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_concat_rows_bwd': (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'sgnn_add': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_expand8_coords': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_expand8_coords_i64': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'sgnn_dense_coords': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'sgnn_compact_ws_bytes': (c_i64, [c_i64]),
    'sgnn_compact_sigmoid': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_compact_dense': (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_compact_mask': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_compact_sigmoid_cap': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_compact_dense_cap': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_compact_sigmoid_cap_locs': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_compact_dense_cap_locs': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_adam_flat': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp]),
    'sgnn_seg_flags': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    'sgnn_status_merge': (c_i32, [c_vp, c_vp, c_vp]),
    'sgnn_sparse_to_dense': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'sgnn_dense_to_sparse': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'sgnn_linear_ws_bytes': (c_i64, [c_i64, c_i32, c_i32]),
    'sgnn_linear_fwd': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    'sgnn_linear_bwd': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_loss_ws_bytes': (c_i64, []),
    'sgnn_loss_level_fwd': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64,
                                    c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_loss_level_bwd': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64,
                                    c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'sgnn_loss_targets': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_f32, c_i32, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'sgnn_loss_multi_ws_bytes': (c_i64, []),
    'sgnn_loss_levels_fwd': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_loss_levels_bwd': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'sgnn_loss_combine': (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    'sgnn_loss_combine_bwd': (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp]),
    'sgnn_prog_arena_floats': (c_i64, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32]),
    'sgnn_prog_ws_bytes': (c_i64, [c_vp, c_i32, c_vp, c_i32]),
    'sgnn_prog_buffer_offset': (c_i64, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32]),
    'sgnn_prog_set_side_stream': (c_i32, [c_vp, c_vp, c_i64]),
    'sgnn_prog_defer_join': (c_i32, [c_i32]),
    'sgnn_prog_forward': (c_i32, [c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                  c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_prog_backward': (c_i32, [c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                   c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]),
    'sgnn_concat3_rows': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_concat3_rows_bwd': (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp,
                                      c_i64, c_vp]),
    'sgnn_expand_weights': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    'sgnn_expand_weights_bwd': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    'sgnn_io_layout': (c_i32, [c_vp, c_i64, c_i32, c_vp]),
    'sgnn_io_flag_entries': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_f32, c_i64, c_vp, c_vp]),
    'sgnn_io_emit_entries': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_io_scatter_dense': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp]),
    'sgnn_iou_counts': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'sgnn_l1_tgtsurf_ws_bytes': (c_i64, []),
    'sgnn_l1_tgtsurf': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_mc_ws_bytes': (c_i64, [c_i32, c_i32, c_i32]),
    'sgnn_mc_count': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_mc_emit': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_weld_slots': (c_i64, [c_i64]),
    'sgnn_weld_build': (c_i32, [c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'sgnn_weld_sweep': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_weld_lookup': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'sgnn_weld_number': (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    'sgnn_mesh_faces': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_take_rows3': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'sgnn_launch_count': (c_i64, []),
    'sgnn_prof_enable': (c_i32, [c_i32]),
    'sgnn_prof_disable': (c_i32, []),
    'sgnn_prof_resume': (c_i32, []),
    'sgnn_prof_count': (c_i32, []),
    'sgnn_prof_dropped': (c_i32, []),
    'sgnn_prof_get': (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'sgnn_stamp_enable': (c_i32, [c_vp, c_i32]),
    'sgnn_stamp_reset': (c_i32, []),
    'sgnn_stamp': (c_i32, [c_cp, c_vp]),
    'sgnn_stamp_count': (c_i32, []),
    'sgnn_stamp_label': (c_cp, [c_i32]),
}

_lib = None


class SgnnError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise SgnnError('libsgnn_hip.so not found at %s — run `python -c "import __graft_entry__ as g; g.build()"` '
                            '(there is no CPU fallback for the sparse operators)' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            if os.environ.get('SGNN_LIB') and not hasattr(lib, name):
                continue        # a kernel-variant / older build loaded for a measurement: entry points added since are absent
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        # measurements only: SGNN_TUNE="conv_dw_blocks=341,prog_lin_bn=0" sets the listed fields of the library's switch table
        # (include/sgnn_hip.h, struct sgnn_tune) once at load (scripts/ab_env2.sh SGNN_TUNE a=1 a=2 compares two settings)
        for item in filter(None, os.environ.get('SGNN_TUNE', '').split(',')):
            name, _, val = item.partition('=')
            try:
                ival = int(val)
            except ValueError:
                raise SgnnError('SGNN_TUNE: %s=%r is not an integer' % (name, val))
            if not hasattr(lib, 'sgnn_tune_set'):
                raise SgnnError('SGNN_TUNE: the library at %s has no switch table' % LIB_PATH)
            if lib.sgnn_tune_set(name.encode(), ival) == TUNE_UNKNOWN:
                raise SgnnError('SGNN_TUNE: %s' % lib.sgnn_last_error().decode())
        _lib = lib
    return _lib


TUNE_UNKNOWN = -(1 << 63)
RULEBOOK_MULTI_MAX = 4      # SGNN_RULEBOOK_MULTI_MAX of include/sgnn_hip.h


def tune(name, value=None):
    """One switch of the library's measurement table (struct sgnn_tune, include/sgnn_hip.h): set it (returns the previous
    value) or, with value None, read it.  Raises on an unknown name / a value out of range."""
    lib = load()
    r = lib.sgnn_tune_get(name.encode()) if value is None else lib.sgnn_tune_set(name.encode(), int(value))
    if r == TUNE_UNKNOWN:
        raise SgnnError(lib.sgnn_last_error().decode())
    return r


STAMPS = False          # scripts/lane_stamps.py: device time stamps at the lane forks / joins of a captured step


def stamp(label):
    """A device time stamp on the current stream (sgnn_stamp) — nothing at all unless STAMPS was switched on."""
    if STAMPS:
        load().sgnn_stamp(label.encode(), stream())


def require_gpu():
    if not torch.cuda.is_available():
        raise SgnnError('sgnn_amd needs a visible MI355X (torch.cuda.is_available() is False); '
                        'the HIP operators have no CPU fallback')


_fn_cache = {}
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


_HOST_DELAY = float(os.environ.get('SGNN_HOST_DELAY_US', '0')) * 1e-6   # diagnostics: is a step host- or GPU-bound?


def call(name, *args):
    """Invoke an int-returning entry point on the current torch stream; raise on error."""
    if _HOST_DELAY:
        import time
        t = time.perf_counter() + _HOST_DELAY
        while time.perf_counter() < t:
            pass
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args, stream())
    if rc != 0:
        raise SgnnError('%s failed (%d): %s' % (name, rc, load().sgnn_last_error().decode()))


def query(name, *args):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    return fn(*args)


def stream():
    """Raw hipStream_t of torch's current stream on the current device (as an integer handle)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device address of a (contiguous) tensor as an integer, or None (NULL)."""
    return None if t is None else t.data_ptr()
