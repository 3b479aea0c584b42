"""ctypes binding of libmcq_hip.so (C ABI: include/mcq.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There
is no CPU fallback: a missing library or a non-HIP tensor is an error.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Another build of the same library for same-box A/B timing of kernel variants (tools/ab_lib.sh): honoured only when the process
# opts in with MCQ_ALLOW_LIB_PATH=1 beside MCQ_LIB_PATH, so that a deployed process never loads a library an inherited
# environment variable names.
_ALT = os.environ.get("MCQ_LIB_PATH") if os.environ.get("MCQ_ALLOW_LIB_PATH") == "1" else None
LIB_PATH = _ALT or os.path.join(_HERE, "lib", "libmcq_hip.so")

# every symbol include/mcq.h declares
SYMBOLS = (
    "mcq_abi_version", "mcq_padded_dim", "mcq_prepared_bytes", "mcq_prepared_decode_bytes", "mcq_prepared_mean_offset", "mcq_prepare", "mcq_prepare_dev", "mcq_prepare_params", "mcq_encode_workspace_bytes",
    "mcq_encode", "mcq_encode_ex", "mcq_refine_indexes", "mcq_decode", "mcq_decode_backward", "mcq_logits", "mcq_logits_workspace_bytes", "mcq_last_encode_launches", "mcq_test_select", "mcq_profile_encode", "mcq_profile_category_name",
    "mcq_logits_argmax", "mcq_logits_refine", "mcq_logits_refine_codes", "mcq_loss_workspace_bytes", "mcq_loss_fwd", "mcq_loss_bwd", "mcq_recon_fwd", "mcq_loss_tail",
    "mcq_jcl_prefix_fwd", "mcq_jcl_prefix_bwd", "mcq_scatter_rows", "mcq_decode_backward_u8",
    "mcq_weight_grad", "mcq_weight_grad_workspace_bytes", "mcq_adam_step", "mcq_loss_head", "mcq_loss_head_tail", "mcq_scales_exp",
    "mcq_decode_backward_waves", "mcq_decode_backward_u8_ex", "mcq_loss_bwd_waves", "mcq_loss_bwd_ex", "mcq_grad_tail",
)

MCQ_EINVAL, MCQ_EUNSUPPORTED, MCQ_EWORKSPACE = -1, -2, -3
_lib = None


class McqError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McqError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); quantization_amd has no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, f32, i32, i64, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
    L.mcq_abi_version.restype = i32
    L.mcq_padded_dim.restype = i32
    L.mcq_padded_dim.argtypes = [i32]
    L.mcq_prepared_bytes.restype = sz
    L.mcq_prepared_bytes.argtypes = [i32, i32, i32]
    L.mcq_prepared_decode_bytes.restype = sz
    L.mcq_prepared_decode_bytes.argtypes = [i32, i32, i32]
    L.mcq_prepare.restype = i32
    L.mcq_prepare.argtypes = [vp, f32, vp, vp, i32, i32, i32, vp, vp]
    L.mcq_prepare_dev.restype = i32
    L.mcq_prepare_dev.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.mcq_prepare_params.restype = i32
    L.mcq_prepare_params.argtypes = [vp, vp, vp, f32, vp, vp, i32, i32, i32, vp, vp, vp]
    L.mcq_prepared_mean_offset.restype = sz
    L.mcq_prepared_mean_offset.argtypes = [i32, i32, i32]
    L.mcq_encode_workspace_bytes.restype = sz
    L.mcq_encode_workspace_bytes.argtypes = [i64, i32, i32, i32]
    L.mcq_encode.restype = i32
    L.mcq_encode.argtypes = [vp, i64, vp, f32, i32, i32, i32, i32, vp, vp, vp, sz, vp]
    L.mcq_encode_ex.restype = i32
    L.mcq_encode_ex.argtypes = [vp, i64, vp, f32, i32, i32, i32, i32, vp, vp, vp, sz, vp, ctypes.c_uint]
    L.mcq_refine_indexes.restype = i32
    L.mcq_refine_indexes.argtypes = [vp, i64, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]
    L.mcq_decode.restype = i32
    L.mcq_decode.argtypes = [vp, i32, i32, i64, vp, i32, i32, i32, vp, vp]
    L.mcq_decode_backward.restype = i32
    L.mcq_decode_backward.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    L.mcq_logits.restype = i32
    L.mcq_logits.argtypes = [vp, i64, vp, f32, i32, i32, i32, vp, vp, sz, vp]
    L.mcq_test_select.restype = i32
    L.mcq_test_select.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.mcq_logits_workspace_bytes.restype = sz
    L.mcq_logits_workspace_bytes.argtypes = [i64, i32, i32]
    L.mcq_logits_argmax.restype = i32
    L.mcq_logits_argmax.argtypes = [vp, i64, vp, f32, i32, i32, i32, vp, vp, vp, sz, vp, ctypes.c_uint]
    L.mcq_logits_refine.restype = i32
    L.mcq_logits_refine.argtypes = [vp, i64, vp, f32, i32, i32, i32, i32, vp, vp, vp, sz, vp, ctypes.c_uint]
    L.mcq_logits_refine_codes.restype = i32
    L.mcq_logits_refine_codes.argtypes = [vp, i64, vp, f32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp, ctypes.c_uint]
    L.mcq_loss_workspace_bytes.restype = sz
    L.mcq_loss_workspace_bytes.argtypes = [i64, i32, i32]
    L.mcq_loss_fwd.restype = i32
    L.mcq_loss_fwd.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, sz, vp]
    L.mcq_loss_bwd.restype = i32
    L.mcq_loss_bwd.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, vp]
    L.mcq_loss_tail.restype = i32
    L.mcq_loss_tail.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]
    L.mcq_recon_fwd.restype = i32
    L.mcq_recon_fwd.argtypes = [vp, vp, i64, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.mcq_jcl_prefix_fwd.restype = i32
    L.mcq_jcl_prefix_fwd.argtypes = [vp, vp, vp, i64, i32, i32, i32, f32, vp, vp]
    L.mcq_jcl_prefix_bwd.restype = i32
    L.mcq_jcl_prefix_bwd.argtypes = [vp, vp, i64, i32, i32, f32, vp, vp, vp]
    L.mcq_decode_backward_u8.restype = i32
    L.mcq_decode_backward_u8.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    L.mcq_scatter_rows.restype = i32
    L.mcq_scatter_rows.argtypes = [vp, i64, i64, vp, i32, i64, i32, i32, i32, vp, vp]
    L.mcq_weight_grad.restype = i32
    L.mcq_weight_grad.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp, sz, vp]
    L.mcq_weight_grad_workspace_bytes.restype = sz
    L.mcq_weight_grad_workspace_bytes.argtypes = [i64, i32, i32]
    f64 = ctypes.c_double
    L.mcq_adam_step.restype = i32
    L.mcq_adam_step.argtypes = [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, f64, f64, vp]
    L.mcq_loss_head.restype = i32
    L.mcq_loss_head.argtypes = [vp, vp, i64, vp, i32, f32, vp, vp]
    L.mcq_loss_head_tail.restype = i32
    L.mcq_loss_head_tail.argtypes = [vp, vp, i64, vp, i32, f32, vp, vp, vp, i32, f32, vp, vp, vp, vp]
    L.mcq_scales_exp.restype = i32
    L.mcq_scales_exp.argtypes = [vp, vp, f32, vp, vp]
    L.mcq_decode_backward_waves.restype = i64
    L.mcq_decode_backward_waves.argtypes = [i32, i32, i32]
    L.mcq_decode_backward_u8_ex.restype = i32
    L.mcq_decode_backward_u8_ex.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp]
    L.mcq_loss_bwd_waves.restype = i64
    L.mcq_loss_bwd_waves.argtypes = [i64, i32, i32]
    L.mcq_loss_bwd_ex.restype = i32
    L.mcq_loss_bwd_ex.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp]
    L.mcq_grad_tail.restype = i32
    L.mcq_grad_tail.argtypes = [vp, i64, vp, vp, f32, vp, i64, f32, vp, vp, vp]
    L.mcq_last_encode_launches.restype = i32
    L.mcq_profile_encode.restype = i32
    L.mcq_profile_encode.argtypes = [vp, i64, vp, f32, i32, i32, i32, i32, vp, sz, vp, ctypes.POINTER(f32), ctypes.POINTER(i32), i32]
    L.mcq_profile_category_name.restype = ctypes.c_char_p
    L.mcq_profile_category_name.argtypes = [i32]
    assert L.mcq_abi_version() == 7
    _lib = L
    return L


def check(rc: int, what: str):
    if rc == 0:
        return
    names = {MCQ_EINVAL: "invalid argument", MCQ_EUNSUPPORTED: "unsupported (codebook_size, num_codebooks)",
             MCQ_EWORKSPACE: "workspace too small"}
    raise McqError(f"{what}: {names.get(rc, 'HIP error %d' % rc)}")
