"""ctypes binding of libsconf_hip.so — the C-ABI drop-in boundary (include/sconf.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', 'libsconf_hip.so')

_lib = None

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> argtypes (all return int status; 0 = ok).  Must match include/sconf.h.
PROTOTYPES = {
    'sconf_gemm_bf16': [i32, vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i64, vp, i64, vp, i64, f32, i32, i32, i32, vp],
    'sconf_gemm_qkv_rotary': [vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, vp, i64, vp],
    'sconf_gemm_softmax_bwd': [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, vp],
    'sconf_rowdot': [vp, vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_gemm_num_splits': [i64, i32],
    'sconf_splitk_reduce': [vp, vp, i64, i64, i32, vp],
    'sconf_norm_fwd': [i32, vp, i32, vp, vp, vp, i32, vp, vp, i64, i64, f32, vp],
    'sconf_norm_bwd': [i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i64, vp, vp, i64, i64, f32, vp],
    'sconf_norm2_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i64, f32, f32, vp],
    'sconf_norm2_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i64, i64, vp],
    'sconf_cast': [vp, i32, vp, i32, i64, vp],
    'sconf_cast_transpose': [vp, vp, i64, i64, vp],
    'sconf_cast_shadows': [vp, i64, i64, vp],
    'sconf_rotary_qkv': [i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp],
    'sconf_softmax_fwd': [i32, vp, i32, vp, i32, i64, i64, vp],
    'sconf_softmax_bwd': [i32, vp, i32, vp, i32, vp, i32, vp, vp, i64, i64, vp],
    'sconf_colsum': [vp, i32, vp, i64, i64, i64, f32, vp],
    'sconf_mask_rows': [vp, i32, vp, i64, i64, i64, vp],
    'sconf_attn_fwd': [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, i32, i32, f32, vp],
    'sconf_attn_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp],
    'sconf_rotary_inplace': [vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_glu_dwconv_fwd': [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_brn_finalize': [vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, vp],
    'sconf_affine_silu_fwd': [vp, vp, vp, i64, i64, vp],
    'sconf_convmod_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, f32, vp],
    'sconf_sub_conv0_fwd': [vp, i32, vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_sub_dwconv_fwd': [vp, vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_sub_dwconv_bwd': [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_sub_conv0_bwd': [vp, vp, i32, vp, vp, i64, i64, i64, i64, vp],
    'sconf_sub_stage01_fwd': [vp, i32, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_sub_stage01_bwd': [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp],
    'sconf_sub_silu_transpose': [i32, vp, vp, vp, i64, i64, i64, vp],
    'sconf_ctc_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp],
    'sconf_ctc_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp],
    'sconf_ctc_fwd_logits': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp],
    'sconf_ctc_bwd_logits': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp],
    'sconf_overlap_add_exp': [vp, i64, i64, i64, i64, i64, vp, vp, i64, vp],
    'sconf_overlap_finalize': [vp, vp, vp, i64, i64, vp],
    'sconf_argmax_rows': [vp, i64, i64, vp, vp],
    'sconf_sumsq': [vp, i64, vp, vp],
    'sconf_madgrad_step': [vp, vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, f32, i64, vp, vp],
    'sconf_madgrad_advance': [vp, vp, f32, vp],
}
PLAIN = {'sconf_softmax_bwd_workspace': ([i64, i64], C.c_int64), 'sconf_gemm_variant': ([i32, i64, i64, i64, i64, i64, i32, i32, i32, i32], C.c_int), 'sconf_norm_bwd_workspace': ([i64, i64], C.c_int64), 'sconf_norm2_bwd_workspace': ([i64, i64], C.c_int64), 'sconf_ctc_bwd_logits_workspace': ([i64, i64], C.c_int64), 'sconf_version': ([], C.c_int), 'sconf_num_cus': ([], C.c_int), 'sconf_last_error': ([], C.c_char_p)}


def load():
    """Load the library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'libsconf_hip.so not found at {LIB_PATH}. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C long-context-asr_amd/csrc`. There is no CPU/PyTorch fallback for the HIP path.')
    lib = C.CDLL(LIB_PATH)
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (args, res) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed: {lib.sconf_last_error().decode()}')
