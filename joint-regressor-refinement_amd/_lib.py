"""ctypes binding of libjrr_hip.so (the C ABI declared in include/jrr.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the
compute modules raises.  torch is imported first so that the library's libamdhip64.so.7
dependency resolves to the HIP runtime torch already loaded (one runtime per process).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

import torch  # noqa: F401  (must precede CDLL, see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
# JRR_LIB (experiments only, tools/exp/ab_libs.sh): load a variant library built by tools/exp/build_variant.py instead of the
# in-tree one -- the in-tree file is never overwritten by an experiment
LIB_PATH = os.environ.get('JRR_LIB') or os.path.join(HERE, 'libjrr_hip.so')

# every exported symbol of include/jrr.h with (restype, argtypes)
_P = c_void_p
SIGNATURES = {
    'jrr_last_error': (c_char_p, []),
    'jrr_version': (c_int, []),
    'jrr_model_create': (c_int, [_P, _P, _P, _P, _P, _P, POINTER(_P)]),
    'jrr_model_destroy': (None, [_P]),
    'jrr_model_bytes': (c_size_t, []),
    'jrr_model_create_in': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, POINTER(_P)]),
    'jrr_model_create_hinted': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, c_size_t, POINTER(_P)]),
    'jrr_model_info': (c_int, [_P, POINTER(c_int32), c_int]),
    'jrr_model_set_faces': (c_int, [_P, _P, c_int]),
    'jrr_engine_workspace_bytes': (c_size_t, [c_int, c_int]),
    'jrr_engine_create': (c_int, [_P, c_int, c_int, _P, c_size_t, c_int, POINTER(_P)]),
    'jrr_engine_destroy': (None, [_P]),
    'jrr_engine_set_batch_norm': (c_int, [_P, c_int]),
    'jrr_engine_set_folded': (c_int, [_P, c_int, _P]),
    'jrr_engine_set_j_regressor': (c_int, [_P, _P, _P, _P]),
    'jrr_engine_set_pose_disc': (c_int, [_P, _P, _P]),
    'jrr_engine_set_shape_disc': (c_int, [_P, _P, _P]),
    'jrr_rot6d_forward': (c_int, [_P, _P, c_int, _P]),
    'jrr_rot6d_backward': (c_int, [_P, _P, _P, c_int, _P]),
    'jrr_rodrigues_forward': (c_int, [_P, _P, c_int, _P]),
    'jrr_rodrigues_backward': (c_int, [_P, _P, _P, c_int, _P]),
    'jrr_find_joints_forward': (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    'jrr_find_joints_backward': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'jrr_smpl_vertices_backward': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'jrr_smpl_posed_joints': (c_int, [_P, _P, _P, _P]),
    'jrr_smpl_posed_joints_backward': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'jrr_find_joints_after_j_step': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_pose_disc_vjp_input': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_joint_loss': (c_int, [_P, _P, c_float, c_int, c_int, _P, _P, _P]),
    'jrr_pose_disc_forward': (c_int, [_P, _P, _P, _P]),
    'jrr_pose_disc_backward_input': (c_int, [_P, _P, c_float, c_float, _P, _P]),
    'jrr_pose_disc_backward_params': (c_int, [_P, _P, c_float, _P, _P, _P]),
    'jrr_shape_disc_backward_params': (c_int, [_P, _P, c_float, _P, _P, _P]),
    'jrr_pose_disc_vjp_params': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_shape_disc_vjp_params': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_shape_disc_forward': (c_int, [_P, _P, _P, _P]),
    'jrr_shape_disc_vjp_input': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_refine_aux_losses': (c_int, [_P, _P, _P, _P]),
    'jrr_adam_step': (c_int, [_P, _P, _P, _P, c_size_t, _P, c_float, c_float, c_float, c_float, _P]),
    'jrr_evaluate': (c_int, [_P, _P, _P, _P, c_int, _P]),
    'jrr_project_joints': (c_int, [_P, _P, _P, c_int, _P]),
    'jrr_engine_set_reprojection': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_camera_prefit': (c_int, [_P, _P, _P, _P, _P, c_int, c_float, _P, _P]),
    'jrr_silhouette_forward': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_silhouette_backward': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_silhouette_loss_grad': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'jrr_silhouette_pix_to_face': (c_int, [_P, _P, _P]),
    'jrr_engine_set_silhouette': (c_int, [_P, _P, _P, _P, _P]),
    'jrr_refine_run': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_float, c_int, _P, _P]),
    'jrr_j_regressor_grad': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    'jrr_j_step_apply': (c_int, [_P, _P, _P, _P, _P, _P, c_float, _P, _P]),
    'jrr_j_support_info': (c_int, [_P, POINTER(c_int32), POINTER(c_int32), _P]),
    'jrr_engine_support_tiles': (c_int, [_P, POINTER(c_int32)]),
    'jrr_engine_support_vertices': (c_int, [_P, POINTER(c_int32)]),
    'jrr_j_regressor_grad_support': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    'jrr_j_step_apply_support': (c_int, [_P, _P, _P, _P, _P, _P, c_float, _P, _P]),
    'jrr_refine_run_after_j_step': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_float, c_int, _P, _P]),
    'jrr_refine_run_j_steps': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_float, c_int, _P, c_int, _P, _P, _P, _P, c_float, _P, _P, c_int, _P]),
    'jrr_engine_set_loss_history': (c_int, [_P, _P, c_int, c_int]),
    'jrr_engine_loss_history_count': (c_int, [_P]),
    'jrr_engine_info': (c_int, [_P, POINTER(c_int32), c_int]),
    'jrr_engine_set_profiling': (c_int, [_P, c_int]),
    'jrr_engine_profile_read': (c_int, [_P, POINTER(c_float), POINTER(c_int32)]),
    'jrr_engine_probe_read': (c_int, [_P, POINTER(ctypes.c_int64)]),
}

_lib = None


class JrrError(RuntimeError):
    pass


def load(build_if_missing: bool = False) -> ctypes.CDLL:
    """Load the HIP library; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import build as _build
            _build.build(verbose=False)
        else:
            raise JrrError(f'{LIB_PATH} not found: build it with `python {os.path.join(HERE, "build.py")}` '
                           f'(the HIP path has no CPU fallback)')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = load().jrr_last_error()
        raise JrrError(f'{what} failed (status {rc}): {msg.decode() if msg else ""}')


def ptr(t):
    """Device/host pointer of a contiguous float32/int32 tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'jrr expects contiguous tensors'
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
