"""ctypes binding of libhoisdf_hip.so (the C ABI declared in include/hoisdf.h).

The HIP library is the product path: there is NO CPU / PyTorch fallback.  ``lib()`` raises
``HoisdfLibraryError`` if the shared object is missing (run ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C hoisdf_amd/csrc``), and every call raises
``HoisdfError`` with the library's message on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HOISDF_LIB", os.path.join(_HERE, "libhoisdf_hip.so"))   # override: A/B experiments
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "hoisdf.h")
MAX_LEVELS = 8


class HoisdfLibraryError(RuntimeError):
    pass


class HoisdfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libhoisdf_hip status {code}: {msg}")
        self.code = code


class Pyramid(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("B", C.c_int), ("data", C.c_void_p * MAX_LEVELS),
                ("C", C.c_int * MAX_LEVELS), ("H", C.c_int * MAX_LEVELS), ("W", C.c_int * MAX_LEVELS)]


class SdfWeights(C.Structure):
    """include/hoisdf.h hoisdf_sdf_weights"""
    _fields_ = [("C", C.c_int), ("sdfin_w0", C.c_void_p), ("sdfin_b0", C.c_void_p), ("sdfin_w1", C.c_void_p),
                ("sdfin_b1", C.c_void_p), ("dec_w0", C.c_void_p), ("dec_b0", C.c_void_p), ("dec_ld0", C.c_int),
                ("dec_w1", C.c_void_p), ("dec_b1", C.c_void_p), ("dec_w2", C.c_void_p), ("dec_b2", C.c_void_p),
                ("dec_w3", C.c_void_p), ("dec_b3", C.c_void_p), ("dec_w4", C.c_void_p), ("dec_b4", C.c_void_p),
                ("emu_img", C.c_void_p * 6), ("emu_img_t", C.c_void_p * 6)]


class EncoderLayerDesc(C.Structure):
    """include/hoisdf.h hoisdf_encoder_layer_desc"""
    _fields_ = [("B", C.c_int), ("S", C.c_int), ("E", C.c_int), ("F", C.c_int), ("H", C.c_int), ("n_query", C.c_int),
                ("n_inter", C.c_int), ("eps", C.c_float), ("drop_p", C.c_float), ("seed", C.c_uint64 * 4), ("attention", C.c_int),
                ("attention_bwd_emulated", C.c_int), ("training", C.c_int), ("x_mag", C.c_void_p)]


_ENC_W = ("w_in", "b_in", "w_out", "b_out", "g1", "be1", "w1", "b1", "w2", "b2", "g2", "be2", "g3", "be3")
_ENC_IMG = ("img_in", "img_in_q", "img_in_kv", "img_out", "img_1", "img_2")


class EncoderLayerWeights(C.Structure):
    """include/hoisdf.h hoisdf_encoder_layer_weights"""
    _fields_ = [(n, C.c_void_p) for n in _ENC_W + _ENC_IMG + tuple(n.replace("img_", "img_t_") for n in _ENC_IMG)]


class DecoderLayerDesc(C.Structure):
    """include/hoisdf.h hoisdf_decoder_layer_desc"""
    _fields_ = [("B", C.c_int), ("Q", C.c_int), ("S", C.c_int), ("E", C.c_int), ("F", C.c_int), ("H", C.c_int), ("kv_len", C.c_int),
                ("eps", C.c_float), ("drop_p", C.c_float), ("seed", C.c_uint64 * 6), ("training", C.c_int)]


_DEC_W = ("sa_w_in", "sa_b_in", "sa_w_out", "sa_b_out", "ca_w_in", "ca_b_in", "ca_w_out", "ca_b_out", "w1", "b1", "w2", "b2",
          "g1", "be1", "g2", "be2", "g3", "be3", "g4", "be4")


class DecoderLayerWeights(C.Structure):
    """include/hoisdf.h hoisdf_decoder_layer_weights"""
    _fields_ = [(n, C.c_void_p) for n in _DEC_W + ("img_ca_kv", "img_t_ca_kv")]


class DecoderLayerGrads(C.Structure):
    """include/hoisdf.h hoisdf_decoder_layer_grads"""
    _fields_ = [("d" + n, C.c_void_p) for n in _DEC_W]


class EncoderLayerGrads(C.Structure):
    """include/hoisdf.h hoisdf_encoder_layer_grads"""
    _fields_ = [("d" + n, C.c_void_p) for n in _ENC_W]


_SDF_G = ("d_sdfin_w0", "d_sdfin_b0", "d_sdfin_w1", "d_sdfin_b1", "d_dec_w0", "d_dec_b0", "d_dec_w1", "d_dec_b1", "d_dec_w2", "d_dec_b2",
          "d_dec_w3", "d_dec_b3", "d_dec_w4", "d_dec_b4")


class SdfWeightGrads(C.Structure):
    """include/hoisdf.h hoisdf_sdf_weight_grads"""
    _fields_ = [(n, C.c_void_p) for n in _SDF_G]


MLP_MAX_LAYERS = 4


class Mlp(C.Structure):
    """include/hoisdf.h hoisdf_mlp"""
    _fields_ = [("n_layers", C.c_int), ("act_last", C.c_int), ("dims", C.c_int * (MLP_MAX_LAYERS + 1)),
                ("w", C.c_void_p * MLP_MAX_LAYERS), ("b", C.c_void_p * MLP_MAX_LAYERS),
                ("img", C.c_void_p * MLP_MAX_LAYERS), ("img_t", C.c_void_p * MLP_MAX_LAYERS)]


class EmuPrepItem(C.Structure):
    """include/hoisdf.h hoisdf_emu_prep_item"""
    _fields_ = [("W", C.c_void_p), ("image", C.c_void_p), ("first_block", C.c_long), ("ldw", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("transpose", C.c_int)]


class MlpGrads(C.Structure):
    """include/hoisdf.h hoisdf_mlp_grads"""
    _fields_ = [("dw", C.c_void_p * MLP_MAX_LAYERS), ("db", C.c_void_p * MLP_MAX_LAYERS)]


_P, _I, _L, _F, _U64, _D = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_double
_PYR = C.POINTER(Pyramid)
_SDFW = C.POINTER(SdfWeights)

# name -> argument ctypes (return type is int unless listed in _RET)
SIGNATURES: Dict[str, List] = {
    "hoisdf_project_gather_fwd": [_PYR, _P, _P, _L, _I, _P, _P, _F, _I, _I, _P, _I, _P, _P, _P],
    "hoisdf_project_gather_bwd": [_PYR, _P, _P, _L, _I, _P, _P, _F, _I, _I, _P, _I, _P],
    "hoisdf_linear_fwd": [_P, _I, _P, _I, _P, _P, _I, _L, _I, _I, _I, _F, _U64, _P, _P],
    "hoisdf_linear_bwd_input": [_P, _I, _P, _F, _P, _I, _P, _I, _L, _I, _I, _I, _P],
    "hoisdf_linear_bwd_weight": [_P, _I, _P, _F, _P, _I, _P, _I, _P, _L, _I, _I, _P, _L, _P],
    "hoisdf_linear_emu_prepare": [_P, _I, _I, _I, _I, _P, _P],
    "hoisdf_linear_emu_prepare_batch": [_P, _I, _L, _P],
    "hoisdf_linear_fwd_emu": [_P, _I, _P, _P, _P, _I, _L, _I, _I, _I, _F, _U64, _P, _P],
    "hoisdf_mag_measure": [_P, _L, _L, _I, _P, _P],
    "hoisdf_head_mag_measure": [_P, _L, _L, _I, _I, _P, _P],
    "hoisdf_linear_fwd_emu_heads": [_P, _I, _P, _P, _P, _I, _L, _I, _I, _P, _P, _P, _I, _P],
    "hoisdf_linear_bwd_input_emu_heads": [_P, _I, _P, _P, _I, _L, _I, _I, _P, _P, _P, _I, _P],
    "hoisdf_linear_fwd_emu_mag": [_P, _I, _P, _P, _P, _I, _L, _I, _I, _I, _F, _U64, _P, _P, _P, _P],
    "hoisdf_linear_bwd_input_emu": [_P, _I, _P, _F, _P, _P, _I, _L, _I, _I, _I, _P],
    "hoisdf_linear_bwd_input_emu_mag": [_P, _I, _P, _F, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P],
    "hoisdf_linear_fwd_emu_small": [_P, _I, _P, _I, _P, _P, _I, _L, _I, _I, _I, _F, _U64, _P, _P],
    "hoisdf_linear_bwd_input_emu_small": [_P, _I, _P, _F, _P, _I, _P, _I, _L, _I, _I, _I, _P],
    "hoisdf_linear_bwd_weight_emu_small": [_P, _I, _P, _F, _P, _I, _P, _I, _P, _L, _I, _I, _P],
    "hoisdf_linear_bwd_weight_emu": [_P, _I, _P, _F, _P, _I, _P, _I, _P, _L, _I, _I, _P, _L, _P],
    "hoisdf_linear_bwd_weight_emu_mag": [_P, _I, _P, _F, _P, _I, _P, _I, _P, _L, _I, _I, _P, _L, _P, _P, _P],
    "hoisdf_relu_dropout_bwd": [_P, _I, _P, _I, _P, _I, _L, _I, _F, _P],
    "hoisdf_posenc_fwd": [_P, _L, _P, _I, _I, _P, _P],
    "hoisdf_sdf_query_fwd": [_PYR, _P, _P, _L, _I, _P, _P, _F, _I, _I, _P, _P, _SDFW, _F, _F, _U64, _P, _P, _P, _P, _P, _L, _P],
    "hoisdf_weightnorm_fwd": [_P, _P, _P, _I, _P, _I, _I, _P],
    "hoisdf_weightnorm_bwd": [_P, _P, _P, _I, _P, _P, _I, _I, _P],
    "hoisdf_sdf_head_fwd": [_P, _I, _P, _P, _P, _P, _L, _I, _F, _P],
    "hoisdf_sdf_head_bwd": [_P, _P, _P, _I, _P, _P, _I, _P, _P, _L, _I, _F, _P],
    "hoisdf_lattice_count": [_P, _P, _P, _F, _I, _I, _P, _P],
    "hoisdf_lattice_fill": [_P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P],
    "hoisdf_select_smallest_abs": [_P, _P, _P, _I, _I, _P, _P],
    "hoisdf_gather_rows": [_P, _I, _P, _L, _I, _P, _I, _P],
    "hoisdf_sdf_sample_keys": [_P, _I, _P, _P, _P, _P, _I, _I, _F, _U64, _P, _P, _P],
    "hoisdf_adamw_step": [_P, _I, _D, _D, _D, _D, _D, _L, _F, _P],
    "hoisdf_aux_image_losses_fwd": [_P, _L, _L, _L, _L, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "hoisdf_aux_image_losses_bwd": [_P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "hoisdf_bn_stats": [_P, _L, _L, _I, _P, _P, _P, _P, _F, _F, _P, _L, _P],
    "hoisdf_bn_apply_fwd": [_P, _L, _P, _L, _P, _P, _I, _F, _P, _P, _I, _P, _P, _L, _I, _P],
    "hoisdf_bn_bwd": [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _L, _P],
    "hoisdf_token_build_fwd": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "hoisdf_token_build_bwd": [_P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "hoisdf_token_build_bwd_ordered": [_P, _P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "hoisdf_attention_fwd": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _U64, _P],
    "hoisdf_attention_bwd": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                             _I, _F, _U64, _P],
    "hoisdf_attention_fwd_f16": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "hoisdf_attention_fwd_bf16x2": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "hoisdf_attention_fwd_emu": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _U64, _P, _L, _I, _P],
    "hoisdf_attention_fwd_emu_mag": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _U64, _P, _L, _I, _P, _P, _P, _P, _P],
    "hoisdf_attention_bwd_emu": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _U64, _P, _P,
                                 _L, _P],
    "hoisdf_attention_bwd_emu_mag": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _U64, _P, _P,
                                 _L, _P, _P, _P, _P, _P, _P],
    "hoisdf_attention_small_fwd": [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _F, _U64, _P],
    "hoisdf_attention_small_bwd": [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _F,
                                   _U64, _P],
    "hoisdf_add_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _U64, _P],
    "hoisdf_add_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _U64, _P],
    "hoisdf_residual_dropout": [_P, _P, _P, _L, _I, _F, _U64, _P],
    "hoisdf_layernorm_rows_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _P],
    "hoisdf_layernorm_rows_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "hoisdf_sdf_query_train_fwd": [_PYR, _P, _P, _L, _I, _P, _P, _F, _I, _I, _SDFW, _F, _F, _U64, _P, _P, _P, _P, _L, _P, _L, _P],
    "hoisdf_sdf_query_bwd": [_PYR, _P, _P, _L, _I, _P, _P, _F, _I, _I, _SDFW, _F, _F, _P, _L, _P, _P, _P, _L, _P],
    "hoisdf_sdf_infer_count": [_P, _P, _P, _F, _I, _I, _P, _P, _P, _P],
    "hoisdf_sdf_infer_count_begin": [_P, _P, _P, _F, _I, _I, _P, _P, _P],
    "hoisdf_sdf_infer": [_PYR, _P, _P, _P, _F, _I, _I, _P, _P, _I, _I, _I, _SDFW, _F, _F, _U64, _P, _P, _P, _P, _L, _P],
    "hoisdf_decoder_layer_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P],
    "hoisdf_decoder_layer_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _P, _P, _P, _L, _P],
    "hoisdf_encoder_layer_fwd": [_P, _P, _P, _P, _P, _P, _L, _P, _L, _P],
    "hoisdf_encoder_layer_bwd": [_P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P],
    "hoisdf_mano_prepare": [_P, _P, _P, _P, _P],
    "hoisdf_mano_head_fwd": [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "hoisdf_mano_head_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "hoisdf_vote_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hoisdf_vote_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hoisdf_vote_loss_fwd": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hoisdf_vote_loss_bwd": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hoisdf_tokens_fwd": [_P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "hoisdf_tokens_bwd": [_P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _L, _P],
    "hoisdf_heads_vote_fwd": [_P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "hoisdf_heads_vote_bwd": [_P, _P, _P, _P, _P, _F, _P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _L, _P],
    "hoisdf_point_loss_fwd": [_P, _P, _L, _L, _I, _L, _I, _F, _F, _F, _P, _P, _P],
    "hoisdf_point_loss_bwd": [_P, _P, _L, _L, _I, _L, _I, _F, _F, _F, _P, _P, _P],
}
_RET = {"hoisdf_version": C.c_char_p, "hoisdf_last_error": C.c_char_p}
_OTHER = {"hoisdf_set_deterministic": ([_I], None), "hoisdf_set_gemm_emu": ([_I], None), "hoisdf_get_gemm_emu": ([], C.c_int), "hoisdf_get_deterministic": ([], C.c_int),
          "hoisdf_mano_dirs_image_floats": ([], C.c_long),
          "hoisdf_point_loss_blocks": ([_L], C.c_int),
          "hoisdf_bn_workspace_floats": ([_L, _I], C.c_long),
          "hoisdf_token_build_bwd_partials": ([], C.c_int),
          "hoisdf_tokens_saved_bytes": ([_P, _L, _I], C.c_long),
          "hoisdf_tokens_workspace_bytes": ([_P, _L, _I], C.c_long),
          "hoisdf_heads_vote_saved_bytes": ([_P, _P, _I, _I, _I, _I], C.c_long),
          "hoisdf_heads_vote_workspace_bytes": ([_P, _P, _I, _I, _I, _I, _I], C.c_long),
          "hoisdf_sdf_infer_workspace": ([_L, _I, _I], C.c_long),
          "hoisdf_sdf_query_train_saved_bytes": ([_L, _I], C.c_long),
          "hoisdf_sdf_query_train_workspace_bytes": ([_L, _I, _I], C.c_long),
          "hoisdf_decoder_layer_saved_bytes": ([_P], C.c_long),
          "hoisdf_decoder_layer_workspace_bytes": ([_P, _I], C.c_long),
          "hoisdf_encoder_layer_saved_bytes": ([_P], C.c_long),
          "hoisdf_encoder_layer_out_mag": ([_P, _P], C.c_void_p),
          "hoisdf_encoder_layer_workspace_bytes": ([_P, _I], C.c_long),
          "hoisdf_sdf_query_workspace": ([_L, _I, _I], C.c_long),
          "hoisdf_linear_bwd_weight_workspace": ([_L, _I, _I], C.c_long),
          "hoisdf_linear_emu_image_bytes": ([_I, _I], C.c_long),
          "hoisdf_linear_emu_prepare_blocks": ([_I, _I, _I], C.c_long),
          "hoisdf_linear_bwd_weight_emu_workspace": ([_L, _I, _I], C.c_long),
          "hoisdf_linear_emu_supported": ([_P, _L, _I], C.c_int),
          "hoisdf_linear_emu_small_max_rows": ([], C.c_int),
          "hoisdf_linear_emu_pieces": ([], C.c_int),
          "hoisdf_mag_words": ([_L], C.c_long),
          "hoisdf_head_mag_words": ([_L, _I, _I], C.c_long),
          "hoisdf_linear_emu_small_supported": ([_P, _L, _P, _L, _L, _I, _I], C.c_int),
          "hoisdf_attention_f16_workspace": ([_I, _I, _I], C.c_long),
          "hoisdf_attention_bf16x2_workspace": ([_I, _I, _I, _I], C.c_long),
          "hoisdf_attention_emu_workspace": ([_I, _I, _I, _I, _I], C.c_long),
          "hoisdf_attention_bwd_emu_workspace": ([_I, _I, _I, _I, _I], C.c_long)}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HoisdfLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built (make -C hoisdf_amd/csrc). "
                "There is no CPU fallback for the hot path.")
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for name, ret in _RET.items():
            fn = getattr(L, name)
            fn.argtypes = []
            fn.restype = ret
        for name, (args, ret) in _OTHER.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ret
        _lib = L
    return _lib


_timer = None


def set_timer(timer) -> None:
    """bench.py hook: ``timer.begin(name, args)`` / ``timer.end(name)`` bracket every C-ABI call whose
    name is in ``timer.names`` (HIP events on the launch stream).  None disables."""
    global _timer
    _timer = timer


def call(name: str, *args) -> None:
    t = _timer
    timed = t is not None and name in t.names
    if timed:
        t.begin(name, args)
    rc = getattr(lib(), name)(*args)
    if timed:
        t.end(name)
    if rc != 0:
        raise HoisdfError(rc, lib().hoisdf_last_error().decode())


def exported_symbols() -> List[str]:
    return list(SIGNATURES) + list(_RET) + list(_OTHER)
