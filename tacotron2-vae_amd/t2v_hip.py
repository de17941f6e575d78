"""ctypes binding of libt2vae_hip.so (C ABI in include/t2vae.h) + the autograd wrappers
the boundary modules in model.py call.

There is no CPU fallback: every op raises `T2VHipError` when the shared library or a GPU
is missing, so a silent eager path can never stand in for the HIP kernels.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (T2V_LIB: another build of the same library — A/B measurements of two kernel variants on ONE GPU box, tools/dbg/ab_lib.sh)
_LIB_PATH = os.environ.get('T2V_LIB') or os.path.join(_HERE, 'libt2vae_hip.so')
_lib = None

H, E, PRE, A, F_LOC, KS, XW, KATT, KATT_INF, G4 = 1024, 512, 256, 128, 32, 31, 2560, 1536, 1792, 4096


class T2VHipError(RuntimeError):
    pass


class _DecWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'packF_att', 'packF_dec', 'packB_att', 'packB_dec', 'bias_att', 'bias_dec', 'wqT', 'wcomb', 'v')] + [
        ('packs_bf16', C.c_int32)]


class _DecTrainBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'gpre', 'memory', 'pm', 'lengths', 'XS', 'CA', 'CD', 'GA', 'GD', 'QP', 'AL', 'ACUM', 'S')]


class _DecBwdBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'dHC', 'DGA', 'DGD', 'DQ', 'DCTX', 'YD', 'YA', 'DCA', 'DCD', 'GPREV', 'GCUM', 'DV')]


class _DecPersistWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'w_ih_att', 'w_hh_att', 'w_ih_dec', 'w_hh_dec', 'bias_att', 'bias_dec', 'wq', 'wcomb', 'v', 'proj_w', 'proj_b',
        'prenet_w1')]


class _DecPersistBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'memory', 'pm', 'lengths', 'pre_first', 'MEL', 'GATE', 'AL', 'stop_flag', 'exchange', 'err_word')]


class _DecTrainPersistWeights(C.Structure):
    C_NAME = 't2v_dec_train_persist_weights'
    _fields_ = [(n, C.c_void_p) for n in (
        'w_ih_att', 'w_hh_att', 'w_ih_dec', 'w_hh_dec', 'bias_dec', 'wq', 'wcomb', 'v')]


class _GemmGroup(C.Structure):
    C_NAME = 't2v_gemm_group'
    _fields_ = [('A', C.c_void_p), ('sAi', C.c_long), ('sAk', C.c_long), ('nb', C.c_int), ('B', C.c_void_p * 3), ('sBj', C.c_long * 3),
                ('sBk', C.c_long * 3), ('N', C.c_int * 3), ('C', C.c_void_p * 3), ('ldc', C.c_int * 3)]


class _DecInferBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'memory', 'pm', 'lengths', 'XS', 'CA', 'CD', 'QP', 'AL', 'ACUM', 'PRE', 'MEL', 'GATE', 'stop_flag',
        'prenet_w1', 'proj_w', 'proj_b')]


EXPORTS = ('t2v_version', 't2v_last_error', 't2v_stamp', 't2v_debug_spin', 't2v_pack_lstm_weights', 't2v_pack_lstm_weights_bf16', 't2v_decoder_train_fwd',
           't2v_decoder_train_bwd', 't2v_clip_adam_step', 't2v_clip_adam_step_guarded', 't2v_decoder_replay_fwd_kernels', 't2v_mel_frontend', 't2v_set_phase_profile', 't2v_decoder_infer_steps', 't2v_conv1d_stat_blocks', 't2v_conv1d_stat_blocks_bf16', 't2v_conv1d_fwd', 't2v_conv1d_bwd', 't2v_conv1d_fwd_bf16', 't2v_conv1d_bwd_bf16', 't2v_conv1d_dw_scratch_floats', 't2v_conv1d_flip_weights', 't2v_gemm_bf16', 't2v_gemm_bf16_splitk', 't2v_gemm_bf16_splitk_scratch_floats', 't2v_attn_wgrad', 't2v_attn_wgrad_scratch_floats',
           't2v_bn_act_fwd', 't2v_bn_act_bwd', 't2v_bn_act_bwd_eval', 't2v_bilstm_fwd', 't2v_bilstm_bwd', 't2v_gemm_f32', 't2v_conv2d_s2_fwd', 't2v_conv2d_s2_bwd', 't2v_conv2d_s2_dw_scratch_floats', 't2v_conv2d_s2_fwd_gemm', 't2v_conv2d_s2_bwd_gemm',
           't2v_conv2d_s2_gemm_scratch_floats',
           't2v_gru_fwd', 't2v_gru_bwd', 't2v_loss_fwd_bwd', 't2v_fuse_location_weights', 't2v_decoder_qp_floats',
           't2v_set_step_params', 't2v_set_step_params_stream', 't2v_decoder_replay_bwd_kernels', 't2v_embedding_fwd', 't2v_embedding_bwd', 't2v_gemm_f32_splitk', 't2v_gemm_splitk_scratch_floats', 't2v_gemm_f32_batched',
           't2v_decoder_infer_persistent', 't2v_decoder_persist_supported', 't2v_decoder_persist_scratch_floats',
           't2v_attn_bwd_slices', 't2v_colsum', 't2v_colsum_scratch_floats', 't2v_gemm_epilogue_bwd',
           't2v_decoder_train_fwd_persistent', 't2v_decoder_train_persist_supported',
           't2v_decoder_train_persist_scratch_floats', 't2v_decoder_bwd_persist_supported',
           't2v_decoder_bwd_dchain_scratch_floats', 't2v_decoder_bwd_dchain', 't2v_decoder_bwd_achain_scratch_floats',
           't2v_decoder_bwd_achain', 't2v_decoder_bwd_achain2', 't2v_decoder_bwd_achain_prepare',
           't2v_decoder_bwd_achain_prepared', 't2v_decoder_bwd_persist_slices', 't2v_decoder_bwd_achain_dq_offset', 't2v_mask_outputs', 't2v_reparam_fwd',
           't2v_reparam_bwd', 't2v_gather_words', 't2v_concat2_rows',
           't2v_decoder_train_fwd_persistent16', 't2v_decoder_train_persist16_supported',
           't2v_decoder_train_persist16_scratch_floats', 't2v_decoder_bwd_persistent16', 't2v_decoder_bwd_persist16_supported',
           't2v_decoder_bwd_persist16_scratch_floats', 't2v_decoder_bwd_persist16_dq_offset', 't2v_decoder_bwd_persist16_slices',
           't2v_decoder_bwd_persist16_fits', 't2v_gemm_f32_set_mode', 't2v_conv1d_x3_set_mode', 't2v_gemm_f32_grouped',
           't2v_gemm_f32_grouped_scratch_floats', 't2v_gemm_bf16_grouped', 't2v_gemm_bf16_grouped_scratch_floats',
           't2v_decoder_bwd_persistent16_prepare', 't2v_decoder_bwd_persistent16_prepared')


def lib_path():
    return _LIB_PATH


def load_library():
    """dlopen the in-tree library (no compute); raises T2VHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(_LIB_PATH):
        raise T2VHipError("libt2vae_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C tacotron2-vae_amd/csrc)")
    lib = C.CDLL(_LIB_PATH)
    lib.t2v_version.restype = C.c_char_p
    lib.t2v_last_error.restype = C.c_char_p
    lib.t2v_pack_lstm_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t2v_pack_lstm_weights_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t2v_fuse_location_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t2v_decoder_qp_floats.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_qp_floats.restype = C.c_long
    lib.t2v_attn_bwd_slices.argtypes = [C.c_int]
    lib.t2v_decoder_train_fwd.argtypes = [C.POINTER(_DecWeights), C.POINTER(_DecTrainBufs), C.c_int, C.c_int,
                                          C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_train_bwd.argtypes = [C.POINTER(_DecWeights), C.POINTER(_DecTrainBufs),
                                          C.POINTER(_DecBwdBufs), C.c_int, C.c_int, C.c_int, C.c_float,
                                          C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_train_fwd_persistent.argtypes = [C.POINTER(_DecTrainPersistWeights), C.POINTER(_DecTrainBufs), C.c_void_p,
                                                     C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_train_persist_supported.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_train_fwd_persistent16.argtypes = lib.t2v_decoder_train_fwd_persistent.argtypes
    lib.t2v_decoder_train_persist16_supported.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_train_persist16_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_decoder_train_persist16_scratch_floats.restype = C.c_long
    lib.t2v_decoder_bwd_persistent16.argtypes = [C.POINTER(_DecTrainPersistWeights), C.POINTER(_DecTrainBufs)] + [C.c_void_p] * 8 + [
        C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_bwd_persistent16_prepared.argtypes = lib.t2v_decoder_bwd_persistent16.argtypes
    lib.t2v_decoder_bwd_persistent16_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.t2v_decoder_bwd_persist16_supported.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_bwd_persist16_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_decoder_bwd_persist16_scratch_floats.restype = C.c_long
    lib.t2v_decoder_bwd_persist16_dq_offset.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_decoder_bwd_persist16_dq_offset.restype = C.c_long
    lib.t2v_decoder_bwd_persist16_slices.argtypes = [C.c_int]
    lib.t2v_decoder_bwd_persist16_fits.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_gemm_f32_set_mode.argtypes = [C.c_int]
    lib.t2v_conv1d_x3_set_mode.argtypes = [C.c_int]
    lib.t2v_gemm_f32_grouped_scratch_floats.argtypes = [C.POINTER(_GemmGroup), C.c_int, C.c_int, C.c_int]
    lib.t2v_gemm_f32_grouped_scratch_floats.restype = C.c_long
    lib.t2v_gemm_f32_grouped.argtypes = [C.POINTER(_GemmGroup), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.t2v_gemm_bf16_grouped_scratch_floats.argtypes = lib.t2v_gemm_f32_grouped_scratch_floats.argtypes
    lib.t2v_gemm_bf16_grouped_scratch_floats.restype = C.c_long
    lib.t2v_gemm_bf16_grouped.argtypes = lib.t2v_gemm_f32_grouped.argtypes
    lib.t2v_decoder_train_persist_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_decoder_train_persist_scratch_floats.restype = C.c_long
    lib.t2v_decoder_bwd_persist_supported.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_bwd_dchain_scratch_floats.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_bwd_dchain_scratch_floats.restype = C.c_long
    lib.t2v_decoder_bwd_dchain.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_bwd_achain_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_decoder_bwd_achain_scratch_floats.restype = C.c_long
    lib.t2v_decoder_bwd_achain.argtypes = [C.POINTER(_DecTrainPersistWeights), C.c_void_p, C.POINTER(_DecTrainBufs)] + [C.c_void_p] * 8 + [
        C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_bwd_achain2.argtypes = lib.t2v_decoder_bwd_achain.argtypes + [C.c_void_p]
    lib.t2v_decoder_bwd_persist_slices.argtypes = [C.c_int]
    lib.t2v_decoder_bwd_achain_dq_offset.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_decoder_bwd_achain_dq_offset.restype = C.c_long
    lib.t2v_decoder_bwd_achain_prepared.argtypes = lib.t2v_decoder_bwd_achain2.argtypes
    lib.t2v_decoder_bwd_achain_prepare.argtypes = [C.POINTER(_DecTrainBufs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                   C.c_float, C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_replay_fwd_kernels.argtypes = [C.POINTER(_DecWeights), C.POINTER(_DecTrainBufs), C.c_int,
                                                   C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_int,
                                                   C.c_void_p]
    lib.t2v_decoder_replay_bwd_kernels.argtypes = [C.POINTER(_DecWeights), C.POINTER(_DecTrainBufs),
                                                   C.POINTER(_DecBwdBufs), C.c_int, C.c_int, C.c_int, C.c_float,
                                                   C.c_float, C.c_uint64, C.c_int, C.c_void_p]
    lib.t2v_clip_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float,
                                       C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t2v_clip_adam_step_guarded.argtypes = lib.t2v_clip_adam_step.argtypes[:-1] + [C.c_void_p, C.c_int, C.c_void_p]
    lib.t2v_mask_outputs.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    lib.t2v_reparam_fwd.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_void_p]
    lib.t2v_reparam_bwd.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_void_p]
    lib.t2v_concat2_rows.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
    lib.t2v_gather_words.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.t2v_embedding_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.t2v_embedding_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.t2v_gemm_epilogue_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
    lib.t2v_colsum.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t2v_colsum_scratch_floats.argtypes = [C.c_long, C.c_long]
    lib.t2v_colsum_scratch_floats.restype = C.c_long
    lib.t2v_set_step_params.argtypes = [C.c_void_p]
    lib.t2v_set_step_params.restype = None
    lib.t2v_set_step_params_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.t2v_mel_frontend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.t2v_set_phase_profile.argtypes = [C.c_void_p]
    lib.t2v_stamp.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.t2v_debug_spin.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.t2v_decoder_infer_persistent.argtypes = [C.POINTER(_DecPersistWeights), C.POINTER(_DecPersistBufs), C.c_int, C.c_int,
                                                 C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p]
    lib.t2v_decoder_persist_supported.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_persist_scratch_floats.argtypes = [C.c_int, C.c_int]
    lib.t2v_decoder_persist_scratch_floats.restype = C.c_long
    lib.t2v_decoder_infer_steps.argtypes = [C.POINTER(_DecWeights), C.POINTER(_DecInferBufs), C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_float, C.c_float, C.c_int, C.c_uint64, C.c_void_p]
    vp = C.c_void_p
    lib.t2v_conv1d_stat_blocks.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.t2v_conv1d_stat_blocks_bf16.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.t2v_conv1d_fwd.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_conv1d_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_conv1d_dw_scratch_floats.argtypes = [C.c_int] * 5
    lib.t2v_conv1d_flip_weights.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.t2v_conv1d_fwd_bf16.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_conv1d_bwd_bf16.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_bn_act_fwd.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    lib.t2v_bn_act_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_uint64, C.c_uint32, C.c_uint32, vp]
    lib.t2v_bn_act_bwd_eval.argtypes = lib.t2v_bn_act_bwd.argtypes
    lib.t2v_bilstm_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.t2v_bilstm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.t2v_gemm_f32.argtypes = [vp, C.c_long, C.c_long, vp, C.c_long, C.c_long, vp, vp, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    lib.t2v_gemm_bf16.argtypes = lib.t2v_gemm_f32.argtypes
    lib.t2v_gemm_bf16_splitk.argtypes = lib.t2v_gemm_f32.argtypes[:-1] + [vp, vp]
    lib.t2v_gemm_bf16_splitk_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_gemm_bf16_splitk_scratch_floats.restype = C.c_long
    lib.t2v_gemm_f32_splitk.argtypes = lib.t2v_gemm_f32.argtypes[:-1] + [vp, vp]
    lib.t2v_gemm_f32_batched.argtypes = [vp, C.c_long, C.c_long, C.c_long, vp, C.c_long, C.c_long, C.c_long, vp, C.c_long,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_gemm_splitk_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.t2v_gemm_splitk_scratch_floats.restype = C.c_long
    lib.t2v_attn_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]     # 8 pointers
    lib.t2v_attn_wgrad_scratch_floats.argtypes = []
    lib.t2v_conv2d_s2_fwd.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_conv2d_s2_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_conv2d_s2_fwd_gemm.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_conv2d_s2_bwd_gemm.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.t2v_conv2d_s2_gemm_scratch_floats.argtypes = [C.c_int] * 6
    lib.t2v_conv2d_s2_gemm_scratch_floats.restype = C.c_long
    lib.t2v_conv2d_s2_dw_scratch_floats.argtypes = [C.c_int] * 6
    lib.t2v_gru_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.t2v_gru_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.t2v_loss_fwd_bwd.argtypes = [vp] * 15 + [C.c_uint64, C.c_int, C.c_int, C.c_float, vp]
    for name in EXPORTS:
        getattr(lib, name)
    _lib = lib
    return lib


# ---- precision switch (hparams.bf16_run, BASELINE configs[4]).  Off: everything is fp32.  On: the k=5 Conv1d
# forward / data-gradient kernels, the time-batched linear layers (own GEMM) and the deferred LSTM weight-gradient
# GEMMs (library) take bf16 operands with fp32 accumulation; master weights, BatchNorm, the recurrent state, the
# per-step LSTM / attention kernels, loss and Adam stay fp32.
_BF16 = False

# BatchNorm `num_batches_tracked` counters touched by the current forward pass: bumped together by ONE
# multi-tensor launch (flush_bn_counters) instead of one tiny kernel per layer (14 per step)
_BN_PENDING = []


def note_bn_counter(t):
    _BN_PENDING.append(t)


_BN_DEFER = [0]


def flush_bn_counters():
    if _BN_PENDING and not _BN_DEFER[0]:
        torch._foreach_add_(_BN_PENDING, 1)
        del _BN_PENDING[:]


class defer_bn_counters(object):
    """inside this context the per-module flushes are postponed: the whole model's counters (14 BatchNorm layers) are
    bumped by one launch when the outermost context exits"""

    def __enter__(self):
        _BN_DEFER[0] += 1

    def __exit__(self, *exc):
        _BN_DEFER[0] -= 1
        if exc[0] is None:
            flush_bn_counters()
        elif not _BN_DEFER[0]:
            del _BN_PENDING[:]
        return False


def set_bf16(on):
    global _BF16
    _BF16 = bool(on)


def bf16_enabled():
    return _BF16


# ---- per-step parameters in device memory (include/t2vae.h: t2v_step_params).  Kernel arguments are frozen when the
# training step is captured into a HIP graph; the kernels read this 32-byte record at run time instead (dropout epoch,
# Adam lr / bias corrections, KL weight).  Host values go through a ring of pinned slots: the H2D copy is asynchronous
# and the host runs ahead of the GPU, so a slot must not be rewritten before its copy has executed.
_DEFAULT_REC = [None]      # the StepParams currently installed as the process default (stream=None records)


class StepParams(object):
    """One engine's record.  stream=None: installed as the process default (read by launches on any stream that has no
    binding of its own); stream=<torch.cuda.Stream>: bound to that stream only (t2v_set_step_params_stream), so two engines
    in one process never share a record."""
    RING = 256
    FIELDS = {'lr': 2, 'bc1': 3, 'bc2s': 4, 'kl_weight': 5}

    def __init__(self, device, stream=None):
        lib = load_library()
        self.host = torch.zeros(self.RING, 8, dtype=torch.float32).pin_memory()
        self.dev = torch.zeros(8, dtype=torch.float32, device=device)
        self.cur = dict(epoch=0, lr=0.0, bc1=1.0, bc2s=1.0, kl_weight=0.0)
        self.dirty = True
        self._i = 0
        self.stream = stream
        self.extra_streams = []      # the engine's side streams (Overlap): kernels launched there read the same record
        if stream is None:
            lib.t2v_set_step_params(C.c_void_p(self.dev.data_ptr()))
            _DEFAULT_REC[0] = self
        else:
            self.bind()

    def bind(self, extra=None):
        """(re)bind this record to its stream(s).  torch hands out pooled native streams round-robin, so a later engine
        may have taken over a handle: the engine whose step is being issued re-binds before it launches"""
        if extra is not None:
            self.extra_streams = list(extra)
        if self.stream is None:
            return
        lib = load_library()
        for st in [self.stream] + self.extra_streams:
            _check(lib.t2v_set_step_params_stream(C.c_void_p(st.cuda_stream), C.c_void_p(self.dev.data_ptr())),
                   't2v_set_step_params_stream')

    def set(self, **kw):
        for k, v in kw.items():
            v = int(v) if k == 'epoch' else float(v)
            if self.cur[k] != v:
                self.cur[k] = v
                self.dirty = True

    def upload(self, force=False):
        """enqueue the record on the current stream if it changed (never inside a graph capture: the training engine
        sets every field before it captures or replays)"""
        if not (self.dirty or force):
            return
        if torch.cuda.is_current_stream_capturing():
            raise T2VHipError("step parameters changed inside a graph capture")
        slot = self.host[self._i % self.RING]
        self._i += 1
        slot.view(torch.int64)[0] = self.cur['epoch']
        for k, i in self.FIELDS.items():
            slot[i] = self.cur[k]
        self.dev.copy_(slot, non_blocking=True)
        self.dirty = False

    def release(self):
        lib = load_library()
        if self.stream is None:
            if _DEFAULT_REC[0] is self:         # a later engine may have installed ITS record as the process default
                lib.t2v_set_step_params(None)
                _DEFAULT_REC[0] = None
        else:
            for st in [self.stream] + self.extra_streams:
                lib.t2v_set_step_params_stream(C.c_void_p(st.cuda_stream), None)


_STEP = None          # the ACTIVE record: the one of the engine whose step is being issued (or the process default)
_STEP_ALL = []


# ---- explicit multi-stream choreography of a training step (round 4).  A step is a handful of long dependent chains
# (encoder conv bank -> BiLSTM, reference encoder, Prenet -> gpre, Postnet data gradients, BiLSTM BPTT ...) made of
# latency-bound kernels that fill a fraction of the 256 CUs, plus bulk work nothing downstream waits for (every weight
# gradient).  `Overlap` owns the side streams of ONE training engine: `side(name)` forks a named stream off the current
# one (the side stream waits for everything issued so far on the current stream, later launches on the two run
# concurrently); `join()` makes the current stream wait for every side stream used since the last join and is called by
# the engine before gradients are consumed (all-reduce / clip + Adam).  Under graph capture the forks and joins become
# graph edges.  Tensors that were allocated on the forking stream and are read on a side stream are kept alive until the
# join (`keep=`), so the caching allocator cannot hand their memory to a later main-stream launch while the side stream
# still reads it.  With no engine active (`overlap() is None`, e.g. a bare `model(x)` + `loss.backward()`), `side()`
# runs its body inline on the current stream: nothing changes for code outside the training engine.
class Overlap(object):
    # reference-encoder branch; deferred ("nobody waits for it") work: weight gradients, Prenet; the four chip-filling LSTM
    # weight-gradient GEMMs of the decoder get a stream of their own so the long tail of small deferred kernels runs next to them
    # 'd': the free-running decoder_rnn chain of the persistent reverse pass and, behind it, its two weight-gradient GEMMs
    NAMES = ('vae', 'w', 'g', 'd')

    def __init__(self, device=None):
        self.on = True
        # (T2V_MAIN_PRIO / T2V_SIDE_PRIO, measurement: stream priorities of the engine's own stream and of the side streams —
        # three same-box pairs showed no difference)
        lo = int(os.environ.get('T2V_SIDE_PRIO', '0'))
        self._streams = {n: torch.cuda.Stream(device=device, priority=lo) for n in self.NAMES}
        self._used = []
        self._keep = []

    def streams(self):
        return list(self._streams.values())

    def stream(self, name):
        return self._streams[name]

    class _Inline(object):
        def __enter__(self):
            return False

        def __exit__(self, *exc):
            return False

    class _Fork(object):
        def __init__(self, st):
            self.ctx = torch.cuda.stream(st)

        def __enter__(self):
            self.ctx.__enter__()
            return True

        def __exit__(self, *exc):
            return self.ctx.__exit__(*exc)

    def mark(self):
        """an event on the current stream: `side(..., after=mark)` forks from THAT point even when the fork is issued later
        (the host issues the critical chain first, the side chains still start where the mark was taken)"""
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def side(self, name, keep=(), after=None):
        """`with ov.side('w', keep=(dy, x)) as forked:` — body runs on the named side stream (forked=True) or inline"""
        if not self.on:
            return Overlap._Inline()
        st = self._streams[name]
        cur = torch.cuda.current_stream()
        if cur == st:
            return Overlap._Inline()
        if after is not None:
            st.wait_event(after)
        else:
            st.wait_stream(cur)
        self._keep.extend(t for t in keep if t is not None)
        if st not in self._used:
            self._used.append(st)
        return Overlap._Fork(st)

    def wait(self, name):
        """the current stream waits for what has been issued on one side stream so far (a mid-step hand-over; the
        stream stays registered for the final join)"""
        st = self._streams[name]
        if self.on and st in self._used and torch.cuda.current_stream() != st:
            torch.cuda.current_stream().wait_stream(st)

    def join(self):
        cur = torch.cuda.current_stream()
        if _STAMPS['on']:
            for n, st in self._streams.items():
                if st in self._used and st != cur:
                    with torch.cuda.stream(st):
                        stamp('side_%s_end' % n)
        for st in self._used:
            if st != cur:
                cur.wait_stream(st)
        self._used = []
        self._keep = []
        _PREFLIP.clear()        # flipped conv weights belong to the step that made them
        BiLSTM._prep.clear()
        flush_err_notes()       # every stream that wrote an error word has been joined: one gather launch for the step


# ---- phase stamps (T2V_STAMPS=1): one-thread launches that write the 100 MHz wall clock at named points of the step
_STAMPS = {'on': os.environ.get('T2V_STAMPS', '0') == '1', 'buf': None, 'names': []}


def stamp(name):
    """drop a time stamp into the current stream (no-op unless T2V_STAMPS=1); tools/stamps.py reads them back"""
    if not _STAMPS['on']:
        return
    only = os.environ.get('T2V_STAMP_ONLY')
    if only and name not in only.split(','):
        return
    if _STAMPS['buf'] is None:
        _STAMPS['buf'] = torch.zeros(256, dtype=torch.int64, device='cuda')
    if name not in _STAMPS['names']:
        _STAMPS['names'].append(name)
    slot = _STAMPS['names'].index(name)
    _check(load_library().t2v_stamp(C.c_void_p(_STAMPS['buf'].data_ptr()), slot, _stream()), 't2v_stamp')


def read_stamps():
    """{name: microseconds since the earliest stamp} of the last executed step (synchronises)"""
    if _STAMPS['buf'] is None:
        return {}
    torch.cuda.synchronize()
    v = _STAMPS['buf'][:len(_STAMPS['names'])].cpu().tolist()
    t0 = min(v)
    return {n: (x - t0) / 100.0 for n, x in zip(_STAMPS['names'], v)}


_OVERLAP = [None]


def overlap():
    """the Overlap manager of the engine whose step is being issued, else None"""
    return _OVERLAP[0]


def set_overlap(ov):
    prev = _OVERLAP[0]
    _OVERLAP[0] = ov
    return prev


def side(name, keep=(), after=None):
    """fork onto a side stream of the active engine, or run inline when no engine is active / overlap is off"""
    ov = _OVERLAP[0]
    return ov.side(name, keep, after) if ov is not None else Overlap._Inline()


def mark():
    ov = _OVERLAP[0]
    return ov.mark() if ov is not None else None


def step_params(create=True, stream=None, fresh=False):
    """the active StepParams (None until a training engine asks for one with create=True).  fresh=True: a new record for
    a new engine (bound to `stream` when given); it becomes the active one."""
    global _STEP
    if (_STEP is None and create) or fresh:
        _STEP = StepParams(torch.device('cuda', torch.cuda.current_device()), stream=stream)
        _STEP_ALL.append(_STEP)
    return _STEP


def activate_step_params(sp):
    """an engine calls this at the top of every step: the Python-side helpers (FlatAdam.step, VAELoss, the model's
    per-forward dropout counters) then talk to THIS engine's record.  A record that is not bound to a stream is also
    (re)installed as the process default."""
    global _STEP
    if sp is not None and sp.stream is not None:
        sp.bind()       # every step: another engine (dead or alive) may have taken or dropped the pooled stream handle
    if _STEP is not sp:
        _STEP = sp
    if sp is not None and sp.stream is None and _DEFAULT_REC[0] is not sp:
        load_library().t2v_set_step_params(C.c_void_p(sp.dev.data_ptr()))
        _DEFAULT_REC[0] = sp


def drop_step_params(sp):
    """an engine that is closed / garbage-collected unbinds its record (captured graphs keep the memory alive)"""
    global _STEP
    if sp in _STEP_ALL:
        _STEP_ALL.remove(sp)
        try:
            sp.release()
        except Exception:
            pass
    if _STEP is sp:
        _STEP = None


def release_step_params():
    """uninstall every device record (the kernels fall back to their by-value arguments); engines created later install
    new ones.  Captured graphs of an engine keep reading the old record's memory, which stays allocated with them."""
    global _STEP
    for sp in _STEP_ALL:
        sp.release()
    del _STEP_ALL[:]
    if _STEP is not None:
        load_library().t2v_set_step_params(None)
        _STEP = None
    _DEFAULT_REC[0] = None


# ---- asynchronous error ledger.  The cooperative kernels (BiLSTM, GRU, attention exchange, reverse-step hand-off)
# use bounded spins: on a timeout they set an error word and leave instead of hanging.  Reading those words right
# away would force a device sync per call, so the wrappers copy each word (device-to-device, 4 bytes) into a small
# per-device ledger and `check_async_errors()` — called by the training loop where it syncs anyway (loss.item()),
# by validate(), bench.py and the tests — reads the ledger once and raises if any call reported a timeout.
#
# Layout of the ledger (round 5): two regions.
#   [0, _ERR_GRAPH_SLOTS)            blocks of _ERR_BLOCK slots, one block per CAPTURED graph.  The gather launches of a
#                                    captured step write their words into the graph's own block on every replay, so the block
#                                    stays reserved while the graph lives and goes back to the free list when the graph is
#                                    evicted (err_capture_begin / err_capture_end / err_release).  A training engine reads
#                                    `err_words(span)` of the graph it just replayed: that is the guard of the optimiser step.
#   [_ERR_GRAPH_SLOTS, _ERR_SLOTS)   eager notes.  A step reserves a contiguous run (err_mark: when fewer than
#                                    _ERR_STEP_RESERVE slots are left the ledger is checked — a sync, and a raise if a word
#                                    is set — and the cursor goes back to the start of the region), so err_range(mark) never
#                                    spans a wrap and no slot index ever reaches _ERR_SLOTS.
_ERR_POOL = {}
_ERR_SLOTS = 4096
_ERR_BLOCK = 32             # ledger words of one captured graph (a step has six or seven cooperative launches)
_ERR_GRAPH_SLOTS = 2048     # 64 captured graphs alive at once, over all engines of the process
_ERR_STEP_RESERVE = 64


# test hook: a substring — the next ledger entry whose label contains it is recorded as a time-out — or (substring, label): the
# entry is recorded as a time-out under that label (two ranks sharing one GPU cannot run the persistent kernels, but the
# recovery path keys on their label)
_ERR_INJECT = [None]


class _ErrPool(object):
    def __init__(self, device):
        self.words = torch.zeros(_ERR_SLOTS, device=device, dtype=torch.int32)
        self.cursor = _ERR_GRAPH_SLOTS          # next eager slot
        self.labels = {}
        self.free_blocks = list(range(0, _ERR_GRAPH_SLOTS, _ERR_BLOCK))
        self.capture = None                     # [lo, next, hi] of the block the running capture writes
        self.used_hi = _ERR_GRAPH_SLOTS         # eager slots [GRAPH_SLOTS, used_hi) may hold unchecked words


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _err_pool(device=None):
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    key = str(dev)
    pool = _ERR_POOL.get(key)
    if pool is None:
        pool = _ERR_POOL[key] = _ErrPool(dev)
    return key, pool


def _err_take_slot(pool):
    if pool.capture is not None:
        lo, nxt, hi = pool.capture
        if nxt >= hi:
            raise T2VHipError("more than %d cooperative launches inside one captured graph" % _ERR_BLOCK)
        pool.capture[1] = nxt + 1
        return nxt
    if pool.cursor >= _ERR_SLOTS:
        # a loop that never checks its errors: check now (syncs; raises if a kernel timed out), then start over
        if _capturing():
            raise T2VHipError("error ledger full inside a graph capture that did not reserve a block")
        check_async_errors()
    slot = pool.cursor
    pool.cursor += 1
    pool.used_hi = max(pool.used_hi, pool.cursor)
    return slot


def _err_note(label, word):
    """word: a 1-element int32 device view holding a kernel's error word (valid once the stream reaches it)"""
    key, pool = _err_pool(word.device)
    slot = _err_take_slot(pool)
    assert 0 <= slot < _ERR_SLOTS
    # (round 4) the 4-byte copy into the ledger is deferred: flush_err_notes() moves up to 16 words per launch (a step has
    # six or seven cooperative launches; each copy was a launch of its own on the critical path)
    want = _ERR_INJECT[0]
    inject = want is not None and (want if isinstance(want, str) else want[0]) in label
    if inject:
        if not isinstance(want, str):
            label = want[1]
        _ERR_INJECT[0] = None
    _ERR_PENDING.setdefault(key, []).append((slot, word, torch.cuda.current_stream(word.device), inject))
    pool.labels[slot] = label
    if len(_ERR_PENDING[key]) >= 16:
        flush_err_notes()


_ERR_PENDING = {}


def flush_err_notes():
    """copy the pending error words into the ledger with one launch per 16 words, on the current stream (which first waits
    for any other stream a pending word was written on).  The training engine calls it where its side streams have joined;
    check_async_errors() calls it before it reads the ledger."""
    for key, pend in list(_ERR_PENDING.items()):
        if not pend:
            continue
        pool = _ERR_POOL[key]
        dev = pool.words.device
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            for st in {id(p[2]): p[2] for p in pend if p[2] != cur}.values():
                cur.wait_stream(st)
            n = len(pend)
            src = (C.c_void_p * n)(*[p[1].data_ptr() for p in pend])
            dst = (C.c_void_p * n)(*[pool.words.data_ptr() + 4 * p[0] for p in pend])
            _check(load_library().t2v_gather_words(src, dst, n, _stream()), 't2v_gather_words')
            for p in pend:
                p[1].record_stream(cur)
                if p[3]:
                    pool.words[p[0]:p[0] + 1].fill_(1)
        del pend[:]


def err_mark(device=None):
    """position of the error ledger now: an engine takes it at the top of a step; err_range(mark) then names the ledger words
    the step's kernels wrote (the guard of the fused optimiser step).  Reserves a contiguous run for the step: with fewer
    than _ERR_STEP_RESERVE eager slots left the ledger is checked (sync; raises on a recorded time-out) and restarted."""
    _, pool = _err_pool(device)
    if pool.capture is not None:
        return pool.capture[1]
    if pool.cursor + _ERR_STEP_RESERVE > _ERR_SLOTS and not _capturing():
        check_async_errors()
    return pool.cursor


def err_range(mark, device=None):
    """int32 device view of the ledger words written since `mark` (None when there are none)"""
    _, pool = _err_pool(device)
    cur = pool.capture[1] if pool.capture is not None else pool.cursor
    if cur <= mark:
        return None
    return pool.words[mark:cur]


def err_words(span, device=None):
    """int32 device view of the ledger block (lo, hi) of a captured graph (None for an empty block)"""
    if span is None or span[1] <= span[0]:
        return None
    _, pool = _err_pool(device)
    return pool.words[span[0]:span[1]]


def err_capture_begin(device=None):
    """right before a graph capture: reserve a ledger block for the graph.  The notes of the capture land in it (and every
    replay rewrites them there).  Returns False when no block is free — the caller then does not capture (the step runs
    eagerly), instead of pinning ledger slots without bound."""
    _, pool = _err_pool(device)
    if pool.capture is not None:
        raise T2VHipError("nested graph captures share no error-ledger block")
    flush_err_notes()               # eager notes still pending must not be swept into the captured gather
    if not pool.free_blocks:
        return False
    lo = pool.free_blocks.pop(0)
    pool.capture = [lo, lo, lo + _ERR_BLOCK]
    return True


def err_capture_end(device=None, keep=True):
    """after the capture: (lo, hi) of the words the graph writes on every replay; keep=False (capture failed) frees the block"""
    key, pool = _err_pool(device)
    if pool.capture is None:
        return None
    lo, nxt, _hi = pool.capture
    pool.capture = None
    if _ERR_PENDING.get(key):       # notes of a capture that died before its flush: their words never get written
        _ERR_PENDING[key] = [p for p in _ERR_PENDING[key] if not (lo <= p[0] < lo + _ERR_BLOCK)]
    if not keep:
        err_release((lo, nxt), device)
        return None
    return (lo, nxt)


def err_release(span, device=None):
    """the graph that owned ledger block `span` is gone: labels dropped, block back on the free list"""
    if span is None:
        return
    _, pool = _err_pool(device)
    lo = span[0] - span[0] % _ERR_BLOCK
    for i in range(lo, lo + _ERR_BLOCK):
        pool.labels.pop(i, None)
    if lo not in pool.free_blocks and 0 <= lo < _ERR_GRAPH_SLOTS:
        pool.words[lo:lo + _ERR_BLOCK].zero_()
        pool.free_blocks.append(lo)
        pool.free_blocks.sort()


def check_async_errors():
    """Raise T2VHipError if any cooperative kernel since the last check reported a barrier timeout (syncs)."""
    bad = []
    flush_err_notes()
    for key, pool in _ERR_POOL.items():
        n = max(pool.used_hi, pool.cursor)
        vals = pool.words[:n].cpu()
        for i in torch.nonzero(vals).flatten().tolist():
            bad.append(pool.labels.get(i, '?'))
        pool.words[:n].zero_()
        pool.cursor = pool.used_hi = _ERR_GRAPH_SLOTS
        for i in [i for i in pool.labels if i >= _ERR_GRAPH_SLOTS]:
            del pool.labels[i]
    if bad:
        e = T2VHipError("cooperative kernel barrier timed out (results of that step are invalid): %s"
                        % ", ".join(sorted(set(bad))))
        e.labels = sorted(set(bad))
        raise e


def _require_gpu(*tensors):
    lib = load_library()
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise T2VHipError("HIP op called with a CPU tensor; this path has no CPU fallback")
    return lib


def _check(rc, what):
    if rc != 0:
        raise T2VHipError("%s failed: rc=%d (%s)" % (what, rc, load_library().t2v_last_error().decode()))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    assert t.dtype == torch.float32
    return t.contiguous()


MAX_DEC_B = 16      # batch columns of one MFMA tile in the per-step kernels; larger batches run in chunks


def replay_fwd_kernels(kernel_mask):
    """Re-issue the k_lstm_fwd256 (mask 1) / k_attn_fwd (mask 2) launches of the most recent DecoderCore.forward
    (first batch chunk) on its saved arena (bench.py roofline leg; needs DecoderCore.keep_last = True)."""
    if DecoderCore.last_call is None:
        raise T2VHipError("replay_fwd_kernels: set DecoderCore.keep_last = True before the forward pass")
    W, Sb, (B, T_in, T, p_att, p_dec, seed), _keep = DecoderCore.last_call
    _check(load_library().t2v_decoder_replay_fwd_kernels(C.byref(W), C.byref(Sb), B, T_in, T, p_att, p_dec,
                                                         seed, int(kernel_mask), _stream()),
           't2v_decoder_replay_fwd_kernels')
    return T + 1 if kernel_mask == 1 else T


def replay_persistent_forward():
    """Re-issue the persistent forward launch (csrc/decoder_train_persist.hip) of the most recent DecoderCore.forward on
    its arena — timing only (bench.py roofline leg; needs DecoderCore.keep_last = True).  Returns the launches issued (1)."""
    if DecoderCore.last_persist is None:
        raise T2VHipError("replay_persistent_forward: the last forward pass did not run on the persistent kernel (or "
                          "DecoderCore.keep_last was off)")
    PW, Sb, scratch, (B, T_in, T, p_att, p_dec, seed), _keep = DecoderCore.last_persist
    lib = load_library()
    fn = lib.t2v_decoder_train_fwd_persistent16 if DecoderCore.last_kernel == 'k_dec_train_persist16' else lib.t2v_decoder_train_fwd_persistent
    _check(fn(C.byref(PW), C.byref(Sb), _p(scratch), B, T_in, T, p_att, p_dec, seed, _stream()), 't2v_decoder_train_fwd_persistent')
    return 1


def replay_persistent_backward():
    """Re-issue the one-launch persistent reverse pass (csrc/decoder_train_bwd_persist.hip) of the most recent
    DecoderCore.backward on its buffers — timing only (S already holds dpre: the numbers it produces are not used)."""
    if DecoderCore.last_bwd_persist is None:
        raise T2VHipError("replay_persistent_backward: the last backward pass did not run on the persistent kernel (or "
                          "DecoderCore.keep_last was off)")
    PW, Sb, (dhc, DGA, DGD, DCTX, DV, DQP, scratch, errw), (B, T_in, T, p_att, p_dec, seed), _keep = DecoderCore.last_bwd_persist
    if DecoderCore.last_bwd_kernel == 'k_bwd_persist16':
        _check(load_library().t2v_decoder_bwd_persistent16(C.byref(PW), C.byref(Sb), _p(dhc), _p(DGA), _p(DGD), _p(DCTX), _p(DV), _p(DQP),
                                                           _p(scratch), _p(errw), B, T_in, T, p_att, p_dec, seed, _stream()),
               't2v_decoder_bwd_persistent16')
        return 1
    _check(load_library().t2v_decoder_bwd_achain(C.byref(PW), None, C.byref(Sb), _p(dhc), _p(DGA), _p(DGD), _p(DCTX), _p(DV), _p(DQP),
                                                 _p(scratch), _p(errw), B, T_in, T, p_att, p_dec, seed, _stream()),
           't2v_decoder_bwd_achain')
    return 1


def replay_bwd_kernels(kernel_mask):
    """Re-issue the k_lstm_bwd256 (mask 1) / k_attn_cell_bwd (mask 2) launches of the most recent DecoderCore.backward
    (first batch chunk) on its buffers — timing only (bench.py roofline leg; needs DecoderCore.keep_last = True)."""
    if DecoderCore.last_bwd is None:
        raise T2VHipError("replay_bwd_kernels: no backward pass was kept (DecoderCore.keep_last = True before the step)")
    W, Sb, Gb, (B, T_in, T, p_att, p_dec, seed), _keep = DecoderCore.last_bwd
    _check(load_library().t2v_decoder_replay_bwd_kernels(C.byref(W), C.byref(Sb), C.byref(Gb), B, T_in, T, p_att, p_dec,
                                                         seed, int(kernel_mask), _stream()),
           't2v_decoder_replay_bwd_kernels')
    return T if kernel_mask == 1 else T + 1


def pack_decoder_weights(w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, k_att, need_bwd, bf16=False, need_fwd=True):
    """MFMA-fragment tiles of the two decoder LSTM cells, read straight from the nn.LSTMCell tensors.  bf16=True
    (bf16_run, training pass): the same tiles rounded to bf16 — the per-step kernels stream half the bytes."""
    lib = load_library()
    dev = w_ih_att.device
    f32 = dict(device=dev, dtype=torch.float32)
    w_ih_att, w_hh_att, w_ih_dec, w_hh_dec = (_f32c(t.detach()) for t in (w_ih_att, w_hh_att, w_ih_dec, w_hh_dec))
    need_fwd = need_fwd or not need_bwd
    packF_att = torch.empty(G4 * k_att, **f32) if need_fwd else None
    packF_dec = torch.empty(G4 * XW, **f32) if need_fwd else None
    packB_att = torch.empty(G4 * KATT, **f32) if need_bwd else None
    packB_dec = torch.empty(G4 * XW, **f32) if need_bwd else None
    fn = lib.t2v_pack_lstm_weights_bf16 if bf16 else lib.t2v_pack_lstm_weights
    _check(fn(_p(w_ih_att), _p(w_hh_att), _p(w_ih_dec), _p(w_hh_dec), int(k_att),
              _p(packF_att), _p(packF_dec), _p(packB_att), _p(packB_dec), _stream()), 't2v_pack_lstm_weights')
    return packF_att, packF_dec, packB_att, packB_dec


def fuse_location_weights(loc_conv, loc_dense):
    """Fused filter bank of LocationLayer (conv then dense, both bias-free: one linear map), W_comb (128,64) stored
    twice in the orders the forward / backward kernels read it: [0] = F[d][g][st] = W_comb[d][4st+g],
    [1] viewed (64,4,32) = R[kk][g][st] = W_comb[4st+g][kk]."""
    lib = load_library()
    wcomb = torch.empty(2, A, 64, device=loc_conv.device, dtype=torch.float32)
    _check(lib.t2v_fuse_location_weights(_p(loc_conv), _p(loc_dense), _p(wcomb), _stream()), 't2v_fuse_location_weights')
    return wcomb


class DecoderCore(torch.autograd.Function):
    """Teacher-forced decoder recurrence (reference Decoder.forward loop, model.py:415-421).

    inputs : gpre (T,B,4096), memory (B,T_in,512), pm (B,T_in,128), lengths int32 (B) | None,
             attention_rnn weight_ih/weight_hh, decoder_rnn weight_ih/weight_hh, bias_dec (4096),
             query weight (128,1024), location conv (32,2,31), location dense (128,32), v (1,128)
    outputs: HC (T,B,1536) = [h_dec_t | ctx_t],  alignments (B,T,T_in)
    Batches larger than 16 run as independent chunks of <= 16 items (the loop has no cross-item coupling), so the
    reference's default batch_size = 64 works; weight gradients are summed over the chunks.
    """
    last_call = None
    last_bwd = None
    keep_last = False       # bench / tests: keep the (first chunk's) arena of the last forward for replays

    # T2V_TRAIN_PERSISTENT=0 forces the launch-per-step forward; the default takes the one-launch persistent kernel
    # (csrc/decoder_train_persist.hip) whenever t2v_decoder_train_persist_supported(B, T_in): B <= 6, T_in <= 560
    persistent = None
    persistent_bwd = None   # same switch for the reverse pass (env T2V_BWD_PERSISTENT, default: as T2V_TRAIN_PERSISTENT)
    last_bwd_mode = None
    last_mode = None        # 'persistent' | 'launch-per-step' of the most recent forward chunk (bench / tests)
    last_bwd_persist = None
    last_persist = None     # keep_last: (weights, bufs, scratch, dims, tensors) of the last persistent forward, for replays
    # round 5: under bf16_run, batches of 7..16 items take the MFMA-batched persistent forward (csrc/decoder_train_persist16.hip:
    # bf16 weight tiles in registers, the batch is the N dimension of the MFMA).  None: on (env T2V_PERSIST16=0 switches it
    # off); False: off; 'force': also for B <= 6, where the fp32-weight kernel is the default (tests)
    persistent16 = None
    last_kernel = None      # name of the forward kernel of the most recent chunk
    last_bwd_kernel = None

    @staticmethod
    def use_persistent16(lib, B, T_in, T):
        flag = DecoderCore.persistent16
        if flag is None:
            flag = os.environ.get('T2V_PERSIST16', '1') != '0'
        if not flag or not _BF16:
            return False
        gate = DecoderCore.persistent
        if gate is None:
            gate = os.environ.get('T2V_TRAIN_PERSISTENT', '1') != '0'
        if not gate or not lib.t2v_decoder_train_persist16_supported(int(B), int(T_in)):
            return False
        if flag != 'force' and DecoderCore.use_persistent(lib, B, T_in, T):
            return False            # B <= 6: the fp32-weight persistent kernel
        return 4 * lib.t2v_decoder_train_persist16_scratch_floats(int(B), int(T_in), int(T)) < 2 ** 31 - 1

    @staticmethod
    def use_persistent(lib, B, T_in, T):
        flag = DecoderCore.persistent
        if flag is None:
            flag = os.environ.get('T2V_TRAIN_PERSISTENT', '1') != '0'
        if not (bool(flag) and bool(lib.t2v_decoder_train_persist_supported(int(B), int(T_in)))):
            return False
        return 4 * lib.t2v_decoder_train_persist_scratch_floats(int(B), int(T_in), int(T)) < 2 ** 31 - 1     # 31-bit buffer offsets

    @staticmethod
    def use_persistent_bwd(lib, B, T_in, T):
        flag = DecoderCore.persistent_bwd
        if flag is None:
            # measured (B = 6, T_in = 84, T = 400, round 3): 12.9 us per reverse step against 18.2 for the launch-per-step
            # pass.  Follows T2V_TRAIN_PERSISTENT unless set itself (both kernels want the GPU to themselves)
            flag = os.environ.get('T2V_BWD_PERSISTENT', os.environ.get('T2V_TRAIN_PERSISTENT', '1')) != '0'
        if not (bool(flag) and bool(lib.t2v_decoder_bwd_persist_supported(int(B), int(T_in)))):
            return False
        return 4 * lib.t2v_decoder_bwd_achain_scratch_floats(int(B), int(T_in), int(T)) < 2 ** 31 - 1

    @staticmethod
    def use_persistent16_bwd(lib, B, T_in, T):
        """bf16_run, B <= 16: the MFMA-batched one-launch reverse pass (csrc/decoder_train_bwd_persist16.hip).  Follows
        persistent16 / T2V_PERSIST16 and the reverse-pass switches; B <= 6 keeps the fp32-weight reverse pass unless forced."""
        flag = DecoderCore.persistent16
        if flag is None:
            flag = os.environ.get('T2V_PERSIST16', '1') != '0'
        if not flag or not _BF16 or os.environ.get('T2V_PERSIST16_BWD', '1') == '0':
            return False
        gate = DecoderCore.persistent_bwd
        if gate is None:
            gate = os.environ.get('T2V_BWD_PERSISTENT', os.environ.get('T2V_TRAIN_PERSISTENT', '1')) != '0'
        # ..._fits = supported(B, T_in) AND every exchange array below the kernel's 31-bit buffer offsets at this T_out (the
        # same test q16_run makes: a long utterance falls back to the launch-per-step pass HERE, not after the forward)
        if not gate or not lib.t2v_decoder_bwd_persist16_fits(int(B), int(T_in), int(T)):
            return False
        if flag != 'force' and DecoderCore.use_persistent_bwd(lib, B, T_in, T):
            return False
        return True

    @staticmethod
    def _fwd_chunk(lib, gpre, memory, pm, lengths, packs, bias_dec, wqT, wcomb, vv, need_grad, p_att, p_dec, seed, wbf=False,
                   raw=None, bwd_prepare=False):
        T, B, _ = gpre.shape
        T_in = memory.shape[1]
        f32 = dict(device=gpre.device, dtype=torch.float32)
        # row 0 of the state arenas (zero initial states) is cleared by the library's one reset launch
        XS = torch.empty(T + 2, B, XW, **f32)
        CA = torch.empty(T + 1, B, H, **f32)
        CD = torch.empty(T + 1, B, H, **f32)
        GA = torch.empty(T, B, G4, **f32) if need_grad else None
        GD = torch.empty(T, B, G4, **f32) if need_grad else None
        QP = torch.empty(lib.t2v_decoder_qp_floats(B, T_in), **f32)
        AL = torch.empty(T + 1, B, T_in, **f32)
        ACUM = torch.empty(T + 1, B, T_in, **f32)
        S = torch.empty(T, B, T_in, A, **f32) if need_grad else None
        packF_att, packF_dec, packB_att, packB_dec = packs
        W = _DecWeights(_p(packF_att), _p(packF_dec), _p(packB_att), _p(packB_dec), None, _p(bias_dec),
                        _p(wqT), _p(wcomb), _p(vv), int(bool(wbf)))
        Sb = _DecTrainBufs(_p(gpre), _p(memory), _p(pm), _p(lengths), _p(XS), _p(CA), _p(CD), _p(GA), _p(GD),
                           _p(QP), _p(AL), _p(ACUM), _p(S))
        p16 = raw is not None and DecoderCore.use_persistent16(lib, B, T_in, T)
        if p16 or (raw is not None and DecoderCore.use_persistent(lib, B, T_in, T)):
            w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, wq = raw
            nscr = (lib.t2v_decoder_train_persist16_scratch_floats if p16 else lib.t2v_decoder_train_persist_scratch_floats)(B, T_in, T)
            scratch = torch.empty(nscr, **f32)
            PW = _DecTrainPersistWeights(_p(w_ih_att), _p(w_hh_att), _p(w_ih_dec), _p(w_hh_dec), _p(bias_dec), _p(wq),
                                         _p(wcomb), _p(vv))
            run = lib.t2v_decoder_train_fwd_persistent16 if p16 else lib.t2v_decoder_train_fwd_persistent
            _check(run(C.byref(PW), C.byref(Sb), _p(scratch), B, T_in, T, float(p_att), float(p_dec), int(seed), _stream()),
                   't2v_decoder_train_fwd_persistent')
            _err_note('decoder forward (persistent kernel hand-off)', QP.view(torch.int32)[B * 256 * A + 31:][:1])
            DecoderCore.last_mode = 'persistent'
            DecoderCore.last_kernel = 'k_dec_train_persist16' if p16 else 'k_dec_train_persist'
            if DecoderCore.keep_last:
                DecoderCore.last_persist = (PW, Sb, scratch, (B, T_in, T, float(p_att), float(p_dec), int(seed)), raw)
            if need_grad and bwd_prepare and DecoderCore.use_persistent16_bwd(lib, B, T_in, T):
                # bf16 reverse pass: its sentinel fills (1.4 MB per time step) go out NOW, on the deferred-work stream
                # (round 5, measured: issued at the TOP of the step instead, next to the encoder, the step time does not change —
                # 12.90 vs 12.91 ms — so the buffers are not held for the whole forward pass)
                NS = lib.t2v_decoder_bwd_persist16_slices(T_in)
                DQP = torch.empty(T, B, NS, A, **f32)
                bscr = torch.empty(lib.t2v_decoder_bwd_persist16_scratch_floats(B, T_in, T), **f32)
                errw = torch.zeros(1, device=gpre.device, dtype=torch.int32)
                with side('w', keep=(XS,)):
                    _check(lib.t2v_decoder_bwd_persistent16_prepare(_p(DQP), _p(bscr), _p(errw), B, T_in, T, _stream()),
                           't2v_decoder_bwd_persistent16_prepare')
                    ev = torch.cuda.Event()
                    ev.record()
                prep = (DQP, bscr, errw, ev, torch.cuda.current_stream())
                return W, Sb, (gpre, memory, pm, lengths, XS, CA, CD, GA, GD, QP, AL, ACUM, S), prep
            if need_grad and bwd_prepare and DecoderCore.use_persistent_bwd(lib, B, T_in, T):
                # the preparation of the reverse pass (sentinel fills, factor arrays of both cells: ~150 us of launches that
                # need nothing but this forward pass) goes out NOW, on the deferred-work stream, next to the Postnet
                NS = lib.t2v_decoder_bwd_persist_slices(T_in)
                DQP = torch.empty(T, B, NS, A, **f32)
                bscr = torch.empty(lib.t2v_decoder_bwd_achain_scratch_floats(B, T_in, T), **f32)
                errw = torch.zeros(1, device=gpre.device, dtype=torch.int32)
                with side('w', keep=(XS, CA, CD, GA, GD)):
                    _check(lib.t2v_decoder_bwd_achain_prepare(C.byref(Sb), _p(DQP), _p(bscr), _p(errw), B, T_in, T, float(p_att),
                                                              float(p_dec), int(seed), _stream()), 't2v_decoder_bwd_achain_prepare')
                    ev = torch.cuda.Event()
                    ev.record()
                # (the prepared buffers travel with the chunk on the autograd ctx — a process-wide table keyed by id(XS) could hand
                # a later chunk the scratch of a forward pass whose backward never ran)
                prep = (DQP, bscr, errw, ev, torch.cuda.current_stream())
                return W, Sb, (gpre, memory, pm, lengths, XS, CA, CD, GA, GD, QP, AL, ACUM, S), prep
            return W, Sb, (gpre, memory, pm, lengths, XS, CA, CD, GA, GD, QP, AL, ACUM, S), None
        _check(lib.t2v_decoder_train_fwd(C.byref(W), C.byref(Sb), B, T_in, T, float(p_att), float(p_dec),
                                         int(seed), _stream()), 't2v_decoder_train_fwd')
        _err_note('decoder forward (attention exchange)', QP.view(torch.int32)[B * 256 * A + 31:][:1])
        DecoderCore.last_mode = 'launch-per-step'
        DecoderCore.last_kernel = 'k_lstm_fwd256 + k_attn_fwd'
        return W, Sb, (gpre, memory, pm, lengths, XS, CA, CD, GA, GD, QP, AL, ACUM, S), None

    @staticmethod
    def forward(ctx, gpre, memory, pm, lengths, w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, bias_dec,
                wq, loc_conv, loc_dense, v, p_att, p_dec, seed, grad_mode=True, pre=None, b_ih_att=None, b_hh_att=None,
                gpre_ready=None):
        lib = _require_gpu(memory, pm, w_ih_att)
        pre2 = None
        if pre is not None:
            # gpre = pre · weight_ih[:, :256]^T + b_ih + b_hh computed here (gpre argument None), so attention_rnn.weight_ih
            # has ONE gradient producer and its prenet columns are written in place by the backward
            assert gpre is None
            pre2 = _f32c(pre).view(-1, PRE)
            if gpre_ready is not None:      # the same product, issued earlier on a side stream (Decoder.prepare)
                gpre = gpre_ready.view(pre.shape[0], pre.shape[1], G4)
            else:
                gpre = gemm(pre2, w_ih_att.detach()[:, :PRE], (b_ih_att + b_hh_att).detach()).view(pre.shape[0], pre.shape[1], G4)
        T, B, _ = gpre.shape
        T_in = memory.shape[1]
        gpre, memory, pm = _f32c(gpre), _f32c(memory), _f32c(pm)
        # grad_mode = torch.is_grad_enabled() of the CALLER (inside Function.forward it is always False): under
        # no_grad (validate(), inference) nothing is saved and the backward-only buffers are not even allocated
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad)
        wbf = bool(_BF16)
        # the persistent forward reads the nn.LSTMCell tensors themselves; the forward packs are only built when some
        # chunk takes the launch-per-step path (the transposed packs of the backward are always needed)
        fwd_persist = all(DecoderCore.use_persistent(lib, min(B, b0 + MAX_DEC_B) - b0, T_in, T) or
                          DecoderCore.use_persistent16(lib, min(B, b0 + MAX_DEC_B) - b0, T_in, T) for b0 in range(0, B, MAX_DEC_B))
        bwd_persist = need_grad and all(DecoderCore.use_persistent_bwd(lib, min(B, b0 + MAX_DEC_B) - b0, T_in, T) or
                                        DecoderCore.use_persistent16_bwd(lib, min(B, b0 + MAX_DEC_B) - b0, T_in, T)
                                        for b0 in range(0, B, MAX_DEC_B))
        if fwd_persist and (bwd_persist or not need_grad):
            packs = (None, None, None, None)        # both passes read the nn.LSTMCell tensors themselves
        else:
            packs = pack_decoder_weights(w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, KATT, need_grad and not bwd_persist, bf16=wbf,
                                         need_fwd=not fwd_persist)
        raw = tuple(_f32c(t.detach()) for t in (w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, wq))
        wqT = wq.detach().t().contiguous()
        bias_dec = _f32c(bias_dec.detach())
        loc_conv, loc_dense, vv = _f32c(loc_conv.detach()), _f32c(loc_dense.detach()), _f32c(v.detach()).view(-1)
        wcomb = fuse_location_weights(loc_conv, loc_dense)
        chunks = []
        for b0 in range(0, B, MAX_DEC_B):
            b1 = min(B, b0 + MAX_DEC_B)
            if b0 == 0 and b1 == B:
                g_c, m_c, pm_c, l_c = gpre, memory, pm, lengths
            else:
                g_c, m_c, pm_c = gpre[:, b0:b1].contiguous(), memory[b0:b1].contiguous(), pm[b0:b1].contiguous()
                l_c = None if lengths is None else lengths[b0:b1].contiguous()
            chunks.append(DecoderCore._fwd_chunk(lib, g_c, m_c, pm_c, l_c, packs, bias_dec, wqT, wcomb, vv, need_grad,
                                                 p_att, p_dec, (int(seed) + 7919 * b0) & 0x7FFFFFFFFFFFFFFF, wbf, raw,
                                                 bwd_prepare=bwd_persist))
        hcs = []
        for _w, _s, k, _prep in chunks:          # (h_dec(t), context(t)) rows of the projection, gathered from the arena in one launch
            XSc = k[4]
            Bc, XWc = XSc.size(1), XSc.size(2)
            hc_c = torch.empty(T, Bc, XWc - H, device=XSc.device, dtype=torch.float32)
            _check(lib.t2v_concat2_rows(_p(XSc[2:, :, KATT:]), XWc, XWc - KATT, _p(XSc[1:, :, H:KATT]), XWc, KATT - H, _p(hc_c),
                                        T * Bc, _stream()), 't2v_concat2_rows')
            hcs.append(hc_c)
        als = [k[10][1:].permute(1, 0, 2) for _, _, k, _ in chunks]
        HC = hcs[0] if len(hcs) == 1 else torch.cat(hcs, 1)
        align = als[0] if len(als) == 1 else torch.cat(als, 0)
        ctx.dims = (B, T_in, T, float(p_att), float(p_dec), int(seed))
        ctx.consts = (packs, bias_dec, wqT, wcomb, vv, loc_conv, loc_dense)
        ctx.wbf = wbf
        ctx.wrefs = (w_ih_att, w_hh_att, w_ih_dec, w_hh_dec)
        ctx.raw = raw
        ctx.pre2 = pre2
        ctx.pre_on_side = gpre_ready is not None     # Decoder.prepare ran the Prenet on the engine's deferred-work stream
        ctx.chunks = [k for _, _, k, _ in chunks] if need_grad else None
        ctx.prepared = [pr for _, _, _, pr in chunks] if need_grad else None
        ctx.mark_non_differentiable(align)
        if DecoderCore.keep_last:
            W0, S0, k0, _ = chunks[0]
            DecoderCore.last_call = (W0, S0, (k0[0].shape[1], T_in, T, float(p_att), float(p_dec), int(seed)),
                                     k0 + (packs, bias_dec, wqT, wcomb, vv))
        return HC, align

    @staticmethod
    def backward(ctx, dHC, _dalign):
        lib = load_library()
        Bt, T_in, T, p_att, p_dec, seed = ctx.dims
        packs, bias_dec, wqT, wcomb, vv, loc_conv, loc_dense = ctx.consts
        packF_att, packF_dec, packB_att, packB_dec = packs
        if ctx.chunks is None:
            raise T2VHipError("DecoderCore.backward without a saved arena (forward ran under no_grad)")
        dev = dHC.device
        f32 = dict(device=dev, dtype=torch.float32)
        dHC = _f32c(dHC)
        NS = lib.t2v_attn_bwd_slices(T_in)
        tcap = (T_in + 15) // 16 * 16
        acc = bacc = None
        wg = None
        dga_l, dmem_l, dpm_l, dpre_l = [], [], [], []
        b0 = 0
        for ci, keep in enumerate(ctx.chunks):
            gpre, memory, pm, lengths, XS, CA, CD, GA, GD, QP, AL, ACUM, S = keep
            B = gpre.shape[1]
            dhc_c = dHC if B == Bt else dHC[:, b0:b0 + B].contiguous()
            DGA = torch.empty(T, B, G4, **f32)
            DGD = torch.empty(T, B, G4, **f32)
            DCTX = torch.empty(T, B, E, **f32)
            p16b = packB_att is None and DecoderCore.use_persistent16_bwd(lib, B, T_in, T)
            if p16b:
                NS = lib.t2v_decoder_bwd_persist16_slices(T_in)
            elif packB_att is None:     # the one-launch reverse pass has its own slice geometry (one workgroup per item up to 96 symbols)
                NS = lib.t2v_decoder_bwd_persist_slices(T_in)
            DV = torch.empty(B, NS, A, **f32)
            W = _DecWeights(_p(packF_att), _p(packF_dec), _p(packB_att), _p(packB_dec), None, _p(bias_dec),
                            _p(wqT), _p(wcomb), _p(vv), int(bool(ctx.wbf)))
            Sb = _DecTrainBufs(_p(gpre), _p(memory), _p(pm), _p(lengths), _p(XS), _p(CA), _p(CD), _p(GA), _p(GD),
                               _p(QP), _p(AL), _p(ACUM), _p(S))
            if p16b:
                # bf16_run, B <= 16: the whole reverse pass as ONE persistent launch on bf16 MFMA tiles (csrc/decoder_train_bwd_persist16.hip)
                w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, wq_raw = ctx.raw
                PW = _DecTrainPersistWeights(_p(w_ih_att), _p(w_hh_att), _p(w_ih_dec), _p(w_hh_dec), _p(bias_dec), _p(wq_raw),
                                             _p(wcomb), _p(vv))
                prep, ctx.prepared[ci] = ctx.prepared[ci], None
                if prep is not None:
                    DQP, scratch, errw, pev, pst = prep
                    torch.cuda.current_stream().wait_event(pev)     # (always: the preparation ran on a side stream)
                    run16 = lib.t2v_decoder_bwd_persistent16_prepared
                else:
                    DQP = torch.empty(T, B, NS, A, **f32)
                    scratch = torch.empty(lib.t2v_decoder_bwd_persist16_scratch_floats(B, T_in, T), **f32)
                    errw = torch.zeros(1, device=dev, dtype=torch.int32)
                    run16 = lib.t2v_decoder_bwd_persistent16
                stamp('dec_bwd_begin')
                _check(run16(C.byref(PW), C.byref(Sb), _p(dhc_c), _p(DGA), _p(DGD), _p(DCTX), _p(DV), _p(DQP),
                                                        _p(scratch), _p(errw), B, T_in, T, p_att, p_dec,
                                                        (seed + 7919 * b0) & 0x7FFFFFFFFFFFFFFF, _stream()), 't2v_decoder_bwd_persistent16')
                split_d = False
                _err_note('decoder backward (persistent kernel hand-off)', errw)
                DecoderCore.last_bwd_mode = 'persistent'
                DecoderCore.last_bwd_kernel = 'k_bwd_persist16'
                dq_off = lib.t2v_decoder_bwd_persist16_dq_offset(B, T_in, T)      # (T, 16, 128): slice 0 of every item has summed the slices
                dq_sum = scratch[dq_off:dq_off + T * 16 * A].view(T, 16, A)[:, :B].reshape(T * B, A)
                if DecoderCore.keep_last and b0 == 0:
                    DecoderCore.last_bwd_persist = (PW, Sb, (dhc_c, DGA, DGD, DCTX, DV, DQP, scratch, errw), (B, T_in, T, p_att, p_dec, seed),
                                                    keep + (ctx.raw, wcomb, vv, bias_dec))
            elif packB_att is None:
                # the whole reverse pass as ONE persistent launch (csrc/decoder_train_bwd_persist.hip)
                DecoderCore.last_bwd_kernel = 'k_achain_bwd'
                w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, wq_raw = ctx.raw
                PW = _DecTrainPersistWeights(_p(w_ih_att), _p(w_hh_att), _p(w_ih_dec), _p(w_hh_dec), _p(bias_dec), _p(wq_raw),
                                             _p(wcomb), _p(vv))
                prep, ctx.prepared[ci] = ctx.prepared[ci], None
                if prep is not None:
                    DQP, scratch, errw, pev, pst = prep
                    torch.cuda.current_stream().wait_event(pev)     # (always: the preparation ran on a side stream)
                    run_fn = lib.t2v_decoder_bwd_achain_prepared
                else:
                    DQP = torch.empty(T, B, NS, A, **f32)
                    scratch = torch.empty(lib.t2v_decoder_bwd_achain_scratch_floats(B, T_in, T), **f32)
                    errw = torch.zeros(1, device=dev, dtype=torch.int32)
                    run_fn = lib.t2v_decoder_bwd_achain2
                stamp('dec_bwd_begin')
                # T2V_BWD_SPLIT=1 (opt-in): the decoder_rnn chain as a launch of its own on stream 'd' — its END is then something
                # the two decoder_rnn weight-gradient GEMMs can wait for, and they run on the CUs it frees while the attention
                # chain goes on.  Measured: graph replay 11.43 -> 11.24 ms per step, eager 11.44 -> 11.77.  Not the default: the
                # two launches need each other's progress, and a replayed HIP graph does not promise that two branches run
                # CONCURRENTLY — with a few more nodes in the graph the executor put the attention chain BEHIND the other branch
                # (correct, the decoder_rnn chain needs nothing from it, but 8 ms instead of 4.4); one launch has no such cliff.
                ov = overlap()
                st_d = ov.stream('d') if (ov is not None and ov.on and os.environ.get('T2V_BWD_SPLIT', '0') == '1') else None
                if st_d is not None:
                    with ov.side('d', keep=(dhc_c, DGD, scratch) + tuple(keep)):
                        pass        # registers the stream for the engine's join (and orders it behind this point)
                _check(run_fn(C.byref(PW), None, C.byref(Sb), _p(dhc_c), _p(DGA), _p(DGD), _p(DCTX), _p(DV),
                                                   _p(DQP), _p(scratch), _p(errw), B, T_in, T, p_att, p_dec,
                                                   (seed + 7919 * b0) & 0x7FFFFFFFFFFFFFFF, _stream(),
                                                   None if st_d is None else C.c_void_p(st_d.cuda_stream)), 't2v_decoder_bwd_achain2')
                split_d = st_d is not None
                _err_note('decoder backward (persistent kernel hand-off)', errw)
                DecoderCore.last_bwd_mode = 'persistent'
                dq_off = lib.t2v_decoder_bwd_achain_dq_offset(B, T_in, T)      # slice 0 of every item has summed the slices already
                dq_sum = scratch[dq_off:dq_off + T * B * A].view(T * B, A)
                if DecoderCore.keep_last and b0 == 0:
                    DecoderCore.last_bwd_persist = (PW, Sb, (dhc_c, DGA, DGD, DCTX, DV, DQP, scratch, errw), (B, T_in, T, p_att, p_dec, seed),
                                                    keep + (ctx.raw, wcomb, vv, bias_dec))
            else:
                split_d = False
                DQ = torch.empty(T, B, NS, A, 2, **f32)
                YD = torch.empty(B, XW, **f32); YA = torch.empty(B, KATT, **f32)
                DCA = torch.empty(B, H, **f32); DCD = torch.empty(B, H, **f32)
                GPREV = torch.empty(2, B, NS, 2, 64, **f32); GCUM = torch.empty(B * NS * tcap + 64, **f32)
                Gb = _DecBwdBufs(_p(dhc_c), _p(DGA), _p(DGD), _p(DQ), _p(DCTX), _p(YD), _p(YA), _p(DCA),
                                 _p(DCD), _p(GPREV), _p(GCUM), _p(DV))
                _check(lib.t2v_decoder_train_bwd(C.byref(W), C.byref(Sb), C.byref(Gb), B, T_in, T, p_att, p_dec,
                                                 (seed + 7919 * b0) & 0x7FFFFFFFFFFFFFFF, _stream()), 't2v_decoder_train_bwd')
                _err_note('decoder backward (dq hand-off)', GCUM.view(torch.int32)[B * NS * tcap + 1:][:1])
                DecoderCore.last_bwd_mode = 'launch-per-step'
                DecoderCore.last_bwd_kernel = 'k_lstm_bwd256 + k_attn_cell_bwd'
                dq_sum = DQ[..., 0].sum(2).view(T * B, A)
                if DecoderCore.keep_last and b0 == 0:
                    DecoderCore.last_bwd = (W, Sb, Gb, (B, T_in, T, p_att, p_dec, seed),
                                            keep + (dhc_c, DGA, DGD, DQ, DCTX, YD, YA, DCA, DCD, GPREV, GCUM, DV, packs, bias_dec,
                                                    wqT, wcomb, vv))
            TB = T * B
            stamp('dec_bwd_end')
            fork = mark()
            # ---- what the encoder's backward waits for stays on the current stream ...
            d_memory = torch.empty(B, T_in, E, **f32)
            # per item: alpha_b^T (T_in x T) · dctx_b (T x 512), all items in one launch
            _check(lib.t2v_gemm_f32_batched(_p(AL[1:]), T_in, 1, B * T_in, _p(DCTX), E, 1, B * E, _p(d_memory), T_in * E, E, B,
                                            T_in, E, T, _stream()), 't2v_gemm_f32_batched')
            dpre = S                                   # overwritten in place by the backward kernels
            d_pm = colsum(dpre.view(T, B * T_in * A)).view(B, T_in, A)
            dmem_l.append(d_memory); dpm_l.append(d_pm)
            if ctx.pre2 is not None and ctx.needs_input_grad[17] and not ctx.pre_on_side:
                # the Prenet ran on the caller's stream: its backward is ordered behind THIS stream only
                dpre_l.append(gemm(DGA.view(TB, G4), ctx.wrefs[0].detach()[:, :PRE].t()).view(T, B, PRE))
            # ---- ... every weight gradient goes to the engine's deferred-work stream (inline without an engine): the
            # ≈ 80 GFLOP of time-batched LSTM weight-gradient GEMMs then run next to the BiLSTM / reference-encoder
            # backward chains instead of in front of them
            with side('w', keep=(DGA, DGD, DCTX, DV, dq_sum, dhc_c) + tuple(keep), after=fork):
                dga2, dgd2 = DGA.view(TB, G4), DGD.view(TB, G4)
                x_prev = XS[0:T].reshape(TB, XW)          # [h_att_{t-1} | ctx_{t-1} | .]
                x_cur = XS[1:T + 1].reshape(TB, XW)       # [h_att_t | ctx_t | h_dec_{t-1}]
                first = wg is None
                if first:       # gradient tensors of the four nn.LSTMCell weights (arena slots when FlatAdam registered them)
                    wg = [grad_slot(w) for w in ctx.wrefs]
                    wg = [torch.empty(w.shape, **f32) if g is None else g for g, w in zip(wg, ctx.wrefs)]
                    if ctx.pre2 is None:        # the prenet columns of attention_rnn.weight_ih get their gradient via gpre
                        wg[0][:, :PRE].zero_()
                d_w_ih_att, d_w_hh_att, d_w_ih_dec, d_w_hh_dec = wg
                # round 6: the five LSTM weight-gradient products as ONE launch of the plane kernel — both gate-gradient operands split
                # (fp32: x3 planes; bf16_run: rounded) once, 1 024 tiles = two full rounds of the chip (T2V_DW_GROUPED=0: one by one)
                grouped = (not split_d and os.environ.get('T2V_DW_GROUPED', '1') != '0'
                           and os.environ.get('T2V_DW_TWO_STREAMS', '0') != '1')
                if ctx.pre2 is not None:    # the input projection of the prenet output, folded into this node: the Prenet's
                    # own backward (issued on this same stream when an engine is active) waits for d_pre, so it goes first
                    pre_c = ctx.pre2 if B == Bt else ctx.pre2.view(T, Bt, PRE)[:, b0:b0 + B].reshape(TB, PRE)
                    if ctx.needs_input_grad[17] and ctx.pre_on_side:
                        dpre_l.append(gemm(dga2, ctx.wrefs[0].detach()[:, :PRE].t()).view(T, B, PRE))
                    if not grouped:
                        gemm(dga2.t(), pre_c.t(), out=d_w_ih_att[:, :PRE], accumulate=not first)
                # each product lands in its own tensor (no split / copy afterwards).  fp32: the own large-tile fp32 MFMA GEMM;
                # bf16_run: the own large-tile bf16 GEMM (k_gemm_bf16_big_rr: operands rounded to bf16 while staged, fp32
                # accumulation) — no library GEMM is left in either step
                # (measured round 4: without these four products the step is 0.71 ms shorter, alone they take 0.82 ms — they run
                # NEXT to the other chains but the chip is shared, so almost all of their time is still on the step's clock)
                two = os.environ.get('T2V_DW_TWO_STREAMS', '0') == '1' and not split_d      # measurement: the two pairs side by side
                with side('g', after=fork):
                    if grouped:
                        att = [(x_prev[:, :H].t(), d_w_hh_att), (x_prev[:, H:KATT].t(), d_w_ih_att[:, PRE:])]
                        if ctx.pre2 is not None:
                            att.insert(0, (pre_c.t(), d_w_ih_att[:, :PRE]))
                        gemm_grouped([(dga2.t(), att),
                                      (dgd2.t(), [(x_cur[:, :KATT].t(), d_w_ih_dec), (x_cur[:, KATT:].t(), d_w_hh_dec)])], accumulate=not first)
                    else:
                        gemm(dga2.t(), x_prev[:, :H].t(), out=d_w_hh_att, accumulate=not first)
                        gemm(dga2.t(), x_prev[:, H:KATT].t(), out=d_w_ih_att[:, PRE:], accumulate=not first)
                    if not grouped and not split_d and not two:
                        gemm(dgd2.t(), x_cur[:, :KATT].t(), out=d_w_ih_dec, accumulate=not first)
                        gemm(dgd2.t(), x_cur[:, KATT:].t(), out=d_w_hh_dec, accumulate=not first)
                if two:
                    with side('d', keep=(DGD, dgd2), after=fork):
                        gemm(dgd2.t(), x_cur[:, :KATT].t(), out=d_w_ih_dec, accumulate=not first)
                        gemm(dgd2.t(), x_cur[:, KATT:].t(), out=d_w_hh_dec, accumulate=not first)
                if split_d:         # in order behind the decoder_rnn chain on ITS stream: they start the moment it ends
                    with torch.cuda.stream(overlap().stream('d')):
                        gemm(dgd2.t(), x_cur[:, :KATT].t(), out=d_w_ih_dec, accumulate=not first)
                        gemm(dgd2.t(), x_cur[:, KATT:].t(), out=d_w_hh_dec, accumulate=not first)
                # the attention weight gradients stay on the deferred-work stream.  (Round 5: the captured DAG's longest path behind the
                # reverse pass is this stream's serial order — Prenet data gradient, these ~450 us, the Prenet backward, the BiLSTM
                # and encoder-conv weight gradients: 1.3 .. 1.7 ms — so they were tried on a stream of their own, T2V_ATTN_WGRAD_STREAM=d:
                # fp32 step 11.24 -> 11.36 ms, bf16 12.86 -> 12.91, two alternating pairs of 60 steps.  A fifth concurrent branch on
                # the graph executor's four queues costs more than the shorter path buys; tools/graph_critical_path.py)
                with side(os.environ.get('T2V_ATTN_WGRAD_STREAM', 'w'), keep=(dq_sum, DV, dpre, AL, ACUM), after=fork):
                    d_wq = gemm(dq_sum.t(), x_cur[:, :H].t())            # (128,1024)
                    d_v = DV.sum((0, 1)).view(1, A)
                    d_loc_dense, d_loc_conv = attn_wgrad(dpre, AL, ACUM, loc_conv, loc_dense, B, T_in, T)
                    parts = [d_wq, d_loc_conv, d_loc_dense, d_v]
                    acc = parts if acc is None else [x + y for x, y in zip(acc, parts)]
            # bias gradients flow on through an Add node (bias_ih + bias_hh) / are handed to two inputs: node's stream
            if split_d:
                torch.cuda.current_stream().wait_stream(overlap().stream('d'))       # (DGD: that chain ended long before this one)
            bparts = [colsum(DGD.view(TB, G4))] + ([colsum(DGA.view(TB, G4))] if ctx.pre2 is not None else [])
            bacc = bparts if bacc is None else [x + y for x, y in zip(bacc, bparts)]
            dga_l.append(DGA)
            b0 += B
        ctx.chunks = ctx.prepared = None   # the arena is released as soon as the backward has consumed it (side-stream readers: keep=)
        d_wq, d_loc_conv, d_loc_dense, d_v = acc
        d_bias_dec = bacc[0]
        d_w_ih_att, d_w_hh_att, d_w_ih_dec, d_w_hh_dec = wg
        d_memory = dmem_l[0] if len(dmem_l) == 1 else torch.cat(dmem_l, 0)
        d_pm = dpm_l[0] if len(dpm_l) == 1 else torch.cat(dpm_l, 0)
        d_pre = d_b_att = None
        if ctx.pre2 is not None:
            DGA = None
            d_b_att = bacc[1]
            if dpre_l and ctx.pre_on_side:
                with side('w'):
                    d_pre = dpre_l[0] if len(dpre_l) == 1 else torch.cat(dpre_l, 1)
            elif dpre_l:
                d_pre = dpre_l[0] if len(dpre_l) == 1 else torch.cat(dpre_l, 1)
        else:
            DGA = dga_l[0] if len(dga_l) == 1 else torch.cat(dga_l, 1)
        return (DGA, d_memory, d_pm, None, d_w_ih_att, d_w_hh_att, d_w_ih_dec, d_w_hh_dec, d_bias_dec,
                d_wq, d_loc_conv, d_loc_dense, d_v, None, None, None, None, d_pre, d_b_att, d_b_att, None)


def attn_wgrad(dpre, AL, ACUM, loc_conv, loc_dense, B, T_in, T):
    """location_dense / location_conv weight gradients summed over the whole decoder pass (csrc/attn_wgrad.hip)."""
    lib = _require_gpu(dpre, AL, ACUM, loc_conv, loc_dense)
    f32 = dict(device=dpre.device, dtype=torch.float32)
    part = torch.empty(lib.t2v_attn_wgrad_scratch_floats(), **f32)
    d_dense, d_conv = torch.empty(A, F_LOC, **f32), torch.empty(F_LOC, 2, KS, **f32)
    _check(lib.t2v_attn_wgrad(_p(dpre), _p(AL), _p(ACUM), _p(_f32c(loc_conv)), _p(_f32c(loc_dense)), _p(part),
                              _p(d_dense), _p(d_conv), B, T_in, T, _stream()), 't2v_attn_wgrad')
    return d_dense, d_conv


def mel_frontend(wav, n_samples, tables, scale=1.0, t_stride=None):
    """Batched STFT→mel on device.  wav: (B,N) float32 or int16 CUDA tensor; n_samples: (B) int64.
    tables: dict from layers.TacotronSTFT (device tensors).  Returns (B,80,T_max) float32."""
    lib = _require_gpu(wav)
    B, N = wav.shape
    wav = wav.contiguous()
    if int(n_samples.min()) <= 512 or int(n_samples.max()) > N:
        # reflect padding of n_fft/2 needs more than n_fft/2 samples (torch raises for the reference's F.pad too)
        raise ValueError("mel_frontend: every utterance needs 512 < n_samples <= %d, got [%d, %d]"
                         % (N, int(n_samples.min()), int(n_samples.max())))
    n_samples = n_samples.to(device=wav.device, dtype=torch.int64).contiguous()
    if t_stride is None:
        t_stride = int(n_samples.max().item()) // 256 + 1
    out = torch.empty(B, 80, t_stride, device=wav.device, dtype=torch.float32)
    is16 = wav.dtype == torch.int16
    if not is16 and wav.dtype != torch.float32:
        raise T2VHipError("mel_frontend takes float32 or int16 samples")
    _check(lib.t2v_mel_frontend(None if is16 else _p(wav), _p(wav) if is16 else None, _p(n_samples), B, N,
                                float(scale), 1024, 256, 80, _p(tables['window']), _p(tables['tw512']),
                                _p(tables['tw1024']), _p(tables['mel_start']), _p(tables['mel_len']),
                                _p(tables['mel_w']), int(tables['maxw']), _p(out), t_stride, _stream()),
           't2v_mel_frontend')
    return out


class InferenceSession(object):
    """Arena + packed weights of one free-running decode (Decoder.inference / the per-step
    initialize_decoder_states → prenet → decode call sequence of synthesizer.py:135-154)."""
    INT_MAX = 2 ** 31 - 1

    def __init__(self, memory, pm, lengths, w_ih_att, w_hh_att, b_att, w_ih_dec, w_hh_dec, b_dec, wq, loc_conv,
                 loc_dense, v, prenet_w0, prenet_w1, proj_w, proj_b, gate_w, gate_b, max_steps):
        if 'T2V_HOST_THREADS' in os.environ:        # (an inference session no longer throttles the host application)
            limit_host_threads()
        lib = _require_gpu(memory, pm, w_ih_att)
        B, T_in, _ = memory.shape
        if B > 8:
            raise T2VHipError("one decode session takes B <= 8 utterances (Decoder.inference runs larger batches 8 at a time)")
        dev = memory.device
        f32 = dict(device=dev, dtype=torch.float32)
        self.B, self.T_in, self.max_steps = B, T_in, int(max_steps)
        T = self.max_steps
        self.memory, self.pm, self.lengths = _f32c(memory.detach()), _f32c(pm.detach()), lengths
        # the persistent one-launch loop reads the nn.LSTMCell tensors themselves (weights stay in registers); the 71 MB pack
        # of the launch-per-stage loop is built by the first run() that needs it (round 4: it cost every utterance ~0.1 ms)
        self.packF_att = self.packF_dec = self.W = None
        self.raw = tuple(_f32c(t.detach()) for t in (w_ih_att, w_hh_att, w_ih_dec, w_hh_dec))
        self.wq = _f32c(wq.detach())
        self.b_att, self.b_dec = _f32c(b_att.detach()), _f32c(b_dec.detach())
        self.wqT = wq.detach().t().contiguous()
        self.wcomb = fuse_location_weights(_f32c(loc_conv.detach()), _f32c(loc_dense.detach()))
        self.v = _f32c(v.detach()).view(-1)
        self.w1 = _f32c(prenet_w1.detach())
        # Prenet layer 0 (bias-free, linear in the mel frame) folded into the projection: rows 81.. = W0·P, W0·b
        w0 = _f32c(prenet_w0.detach())
        pw, pb = _f32c(proj_w.detach()), _f32c(proj_b.detach())
        w0p = gemm(w0, pw.t())                                                   # (256,1536) = W0 (256,80) · P (80,1536)
        b0p = gemm(w0, pb.view(1, -1)).view(-1)                                  # (256) = W0 · b
        self.proj_w = torch.cat((pw, gate_w.detach(), w0p), 0).contiguous()      # (337,1536)
        self.proj_b = torch.cat((pb, gate_b.detach(), b0p), 0).contiguous()
        # the persistent launch keeps the recurrent state on chip: it needs PRE[0] and writes MEL / GATE / AL / stop.  The state
        # arenas of the launch-per-stage loop (XS, CA, CD, QP, ACUM: 20 MB, five clears) are made by the first run() (round 4)
        self.XS = self.CA = self.CD = self.QP = self.ACUM = self.S = None
        self.AL = torch.empty(T + 1, B, T_in, **f32); self.AL[0].zero_()
        self.PRE = torch.empty(T + 1, B, PRE, **f32)
        self.MEL = torch.empty(T, B, 80, **f32)
        self.GATE = torch.empty(T, B, **f32)
        self.stop = torch.full((1,), self.INT_MAX, device=dev, dtype=torch.int32)
        self.t = 0

    def _stage_arena(self):
        if self.S is not None:
            return
        lib = load_library()
        B, T_in, T = self.B, self.T_in, self.max_steps
        f32 = dict(device=self.memory.device, dtype=torch.float32)
        self.XS = torch.empty(T + 2, B, XW, **f32); self.XS[0:2].zero_()
        self.CA = torch.empty(T + 1, B, H, **f32); self.CA[0].zero_()
        self.CD = torch.empty(T + 1, B, H, **f32); self.CD[0].zero_()
        self.QP = torch.empty(lib.t2v_decoder_qp_floats(B, T_in), **f32)
        self.ACUM = torch.empty(T + 1, B, T_in, **f32); self.ACUM[0].zero_()
        self.S = _DecInferBufs(_p(self.memory), _p(self.pm), _p(self.lengths), _p(self.XS), _p(self.CA), _p(self.CD),
                               _p(self.QP), _p(self.AL), _p(self.ACUM), _p(self.PRE), _p(self.MEL), _p(self.GATE),
                               _p(self.stop), _p(self.w1), _p(self.proj_w), _p(self.proj_b))

    def persistent_supported(self):
        return bool(load_library().t2v_decoder_persist_supported(self.B, self.T_in))

    def run_persistent(self, gate_threshold, p_prenet, seed):
        """frames 0..max_steps-1 (or up to the frame the gate fires) as ONE persistent launch (csrc/decoder_persist.hip);
        PRE[0] must hold Prenet(go frame).  Outputs: MEL, GATE, AL, stop."""
        lib = load_library()
        dev = self.memory.device
        if not hasattr(self, '_gran'):
            self._gran = torch.empty(lib.t2v_decoder_persist_scratch_floats(self.B, self.max_steps), device=dev, dtype=torch.float32)
            self._perr = torch.zeros(1, device=dev, dtype=torch.int32)
        W = _DecPersistWeights(_p(self.raw[0]), _p(self.raw[1]), _p(self.raw[2]), _p(self.raw[3]), _p(self.b_att), _p(self.b_dec),
                               _p(self.wq), _p(self.wcomb), _p(self.v), _p(self.proj_w), _p(self.proj_b), _p(self.w1))
        Bf = _DecPersistBufs(_p(self.memory), _p(self.pm), _p(self.lengths), _p(self.PRE[0]), _p(self.MEL), _p(self.GATE),
                             _p(self.AL), _p(self.stop), _p(self._gran), _p(self._perr))
        _check(lib.t2v_decoder_infer_persistent(C.byref(W), C.byref(Bf), self.B, self.T_in, self.max_steps,
                                                float(gate_threshold), float(p_prenet), int(seed), _stream()),
               't2v_decoder_infer_persistent')

    def persistent_timed_out(self):
        """True when a bounded spin of the last run_persistent() gave up (the 256 workgroups were not co-resident);
        synchronises — Decoder.inference reads the stop frame at the same point anyway"""
        return bool(self._perr.item())

    def reset_for_rerun(self):
        """after a failed persistent run: the launch-per-stage loop starts again from the zero state (rows 0 of the state
        arenas are still zero — the persistent kernel keeps its state on chip — and PRE[0] still holds Prenet(go frame))"""
        self.stop.fill_(self.INT_MAX)
        self._perr.zero_()

    def run(self, t0, t1, gate_threshold, p_prenet, external_prenet, seed):
        self._stage_arena()
        if self.W is None:
            self.packF_att, self.packF_dec, _, _ = pack_decoder_weights(self.raw[0], self.raw[1], self.raw[2], self.raw[3], KATT_INF,
                                                                        False)
            self.W = _DecWeights(_p(self.packF_att), _p(self.packF_dec), None, None, _p(self.b_att), _p(self.b_dec),
                                 _p(self.wqT), _p(self.wcomb), _p(self.v))
        _check(load_library().t2v_decoder_infer_steps(C.byref(self.W), C.byref(self.S), self.B, self.T_in, int(t0),
                                                      int(t1), float(gate_threshold), float(p_prenet),
                                                      int(bool(external_prenet)), int(seed), _stream()),
               't2v_decoder_infer_steps')


def limit_host_threads(n=None):
    """Cap torch's intra-op (OpenMP) host threads for the training process (default 4, env T2V_HOST_THREADS; 0 = leave).
    Every step touches a few host tensors (the 768 KB mel of the collated batch is copied into the pinned staging buffer
    with a parallel copy); on a 256-thread host torch starts 128 workers for that, and they spin after the region.  The GPU
    boxes run containers with a CPU quota (cgroup cpu.max = 16 CPUs per 100 ms): the spinning workers exhaust it and the
    WHOLE process — the thread that feeds the GPU included — is frozen for the rest of each period.  Measured: eager steps
    13.7, 13.7, 71 ms, ... (every ~100 ms an ~60 ms freeze; 33–39 ms per step on average) against a steady 13.7 ms with 4
    threads; the replayed-graph step 14.07 -> 13.43 ms.  Data-loader workers are separate processes and unaffected."""
    prev = torch.get_num_threads()
    if n is None:
        n = int(os.environ.get('T2V_HOST_THREADS', '2' if int(os.environ.get('WORLD_SIZE', '1')) > 1 else '4'))
    if n > 0 and prev > n:
        torch.set_num_threads(n)
    return prev         # (the training engine restores it in close(): the cap belongs to the training loop, not to the host app)


ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2


# ---- data-gradient operands of the k=5 convolutions, flipped once per step (Tacotron2._forward issues ONE launch for the
# eight layers on the deferred-work stream during the forward pass) instead of one flip launch in front of every dX conv
_PREFLIP = {}


def preflip_conv_weights(weights):
    """weights: fp32 (Cout, Cin, KS) conv weights with one common KS.  Fills _PREFLIP[data_ptr] = flipped copy; the
    ConvBNAct1d backward of this step picks it up (and falls back to its own flip launch when there is none)."""
    lib = load_library()
    _PREFLIP.clear()
    ws = [w.detach() for w in weights if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()]
    if not ws or _BF16:
        return
    n = len(ws)
    KS = ws[0].shape[2]
    if n > 16 or any(w.shape[2] != KS for w in ws):
        return
    total = sum(w.numel() for w in ws)
    flat = torch.empty(total, device=ws[0].device, dtype=torch.float32)
    outs, o = [], 0
    for w in ws:
        outs.append(flat[o:o + w.numel()].view(w.shape[1], w.shape[0], KS))
        o += w.numel()
    PA, IA = C.c_void_p * n, C.c_int * n
    _check(lib.t2v_conv1d_flip_weights(PA(*[w.data_ptr() for w in ws]), PA(*[t.data_ptr() for t in outs]),
                                       IA(*[w.shape[0] for w in ws]), IA(*[w.shape[1] for w in ws]), int(KS), n, _stream()),
           't2v_conv1d_flip_weights')
    ev = torch.cuda.Event()
    ev.record()             # consumers on another stream wait for THIS point, not for whatever that stream is doing by then
    st = torch.cuda.current_stream()
    for w, t in zip(ws, outs):
        _PREFLIP[w.data_ptr()] = (t, st, ev)


class ConvBNAct1d(torch.autograd.Function):
    """dropout(act(BatchNorm1d(Conv1d(x)))) — one block of the encoder conv bank / Postnet
    (reference model.py:143-148, 175-177) on the HIP implicit-GEMM conv + per-channel BN kernels.
    `x` is saved by reference: the in-place output masking of reference model.py:515 therefore reaches
    the first Postnet block's weight gradient exactly like it does in the reference (Appendix B-5)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, act, p_drop, seed,
                rng_stream, rng_t):
        lib = _require_gpu(x, weight)
        x = x if x.is_contiguous() else x.contiguous()
        B, Cin, T = x.shape
        Cout, _, KS = weight.shape
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        y = torch.empty(B, Cout, T, **f32)
        w = weight.contiguous()
        # bf16 only for the wide layers (the three 512->512 Postnet convs, the encoder bank); the 80-channel first
        # and last Postnet layers are cheap and stay fp32 (the output layer in particular)
        use_bf16 = _BF16 and KS == 5 and Cin % 16 == 0 and Cin >= 128 and Cout >= 128
        # (the number of BatchNorm partials follows the kernel that will run: the bf16 kernels keep their own tiles, the fp32 entry
        #  may take the x3 convolution with its 128-position tiles)
        nblk = (lib.t2v_conv1d_stat_blocks_bf16 if use_bf16 else lib.t2v_conv1d_stat_blocks)(B, T, Cin, Cout, KS)
        part = torch.empty(nblk, Cout, 2, **f32) if training else None
        if use_bf16:
            wp = torch.empty(w.numel(), device=dev, dtype=torch.bfloat16)
            _check(lib.t2v_conv1d_fwd_bf16(_p(w), _p(x), _p(bias), _p(y), _p(part), _p(wp), B, Cin, T, Cout, KS,
                                           _stream()), 't2v_conv1d_fwd_bf16')
        else:
            _check(lib.t2v_conv1d_fwd(_p(w), _p(x), _p(bias), _p(y), _p(part), B, Cin, T, Cout, KS, _stream()),
                   't2v_conv1d_fwd')
        mean = torch.empty(Cout, **f32) if training else None
        rstd = torch.empty(Cout, **f32) if training else None
        out = torch.empty(B, Cout, T, **f32)
        _check(lib.t2v_bn_act_fwd(_p(y), _p(part), nblk, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                  _p(mean), _p(rstd), _p(out), B, Cout, T, int(act), int(bool(training)),
                                  float(p_drop if training else 0.0), 0.1, 1e-5, int(seed), int(rng_stream),
                                  int(rng_t), _stream()), 't2v_bn_act_fwd')
        ctx.cfg = (B, Cin, T, Cout, KS, int(act), float(p_drop if training else 0.0), int(seed), int(rng_stream),
                   int(rng_t), bool(training))
        ctx.keep = (x, w, y, mean, rstd, gamma, beta, running_mean, running_var)
        ctx.small = (gamma, beta, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = load_library()
        B, Cin, T, Cout, KS, act, p, seed, rs, rt, training = ctx.cfg
        x, w, y, mean, rstd, gamma, beta, running_mean, running_var = ctx.keep
        f32 = dict(device=x.device, dtype=torch.float32)
        bn_bwd = lib.t2v_bn_act_bwd
        if not training:   # eval-mode BN backward (model.eval() with gradients, e.g. fine-tuning with frozen statistics)
            mean = running_mean
            rstd = torch.rsqrt(running_var + 1e-5)
            bn_bwd = lib.t2v_bn_act_bwd_eval
        dout = dout.contiguous()
        dy = torch.empty(B, Cout, T, **f32)
        # d(bias) of a conv feeding a training-mode BatchNorm is identically zero (dy has zero channel mean): the BN
        # kernel writes those zeros along with dgamma / dbeta (eval mode: the real bias gradient)
        dgamma, dbeta, dbias = _small_grads(ctx.small, Cout, f32)
        _check(bn_bwd(_p(y), _p(dout), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dy), _p(dgamma),
                      _p(dbeta), _p(dbias), B, Cout, T, act, p, seed, rs, rt, _stream()), 't2v_bn_act_bwd')
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty(B, Cin, T, **f32) if need_dx else None
        fork = mark()       # the weight gradient below depends on dy only: it forks from HERE, not from behind the dX conv
        # the data gradient (what the next layer down waits for) on the current stream ...
        if need_dx and _BF16 and KS == 5 and Cin % 16 == 0 and Cout % 16 == 0 and Cin >= 128 and Cout >= 128:
            wp = torch.empty(w.numel(), device=x.device, dtype=torch.bfloat16)
            _check(lib.t2v_conv1d_bwd_bf16(_p(w), _p(x), _p(dy), _p(dx), None, _p(wp), None, B, Cin, T, Cout, KS,
                                           _stream()), 't2v_conv1d_bwd_bf16')
        elif need_dx:
            pre = _PREFLIP.get(w.data_ptr())
            if pre is not None and pre[0].numel() == w.numel():
                wt, st, ev = pre
                if st != torch.cuda.current_stream():
                    torch.cuda.current_stream().wait_event(ev)          # flipped on the deferred-work stream, long ago
                _check(lib.t2v_conv1d_bwd(None, _p(x), _p(dy), _p(dx), None, _p(wt), None, B, Cin, T, Cout, KS,
                                          _stream()), 't2v_conv1d_bwd')
            else:
                wt = torch.empty_like(w)
                _check(lib.t2v_conv1d_bwd(_p(w), _p(x), _p(dy), _p(dx), None, _p(wt), None, B, Cin, T, Cout, KS,
                                          _stream()), 't2v_conv1d_bwd')
        # ... the weight gradient (nobody waits for it before the optimiser) on the engine's deferred-work stream
        with side('w', keep=(x, dy), after=fork):
            dw = grad_slot(w)
            dw = torch.empty_like(w) if dw is None else dw
            nscr = lib.t2v_conv1d_dw_scratch_floats(B, Cin, T, Cout, KS)
            scr = torch.empty(nscr, **f32) if nscr else None
            if _BF16 and KS == 5 and Cin % 16 == 0 and os.environ.get('T2V_CONV_DW_BF16', '1') != '0':
                # (round 5) bf16_run: the weight gradient on bf16 MFMA too (dY, X rounded while staged; fp32 accumulation)
                _check(lib.t2v_conv1d_bwd_bf16(None, _p(x), _p(dy), None, _p(dw), None, _p(scr), B, Cin, T, Cout, KS,
                                               _stream()), 't2v_conv1d_bwd_bf16')
            else:
                _check(lib.t2v_conv1d_bwd(None, _p(x), _p(dy), None, _p(dw), None, _p(scr), B, Cin, T, Cout, KS,
                                          _stream()), 't2v_conv1d_bwd')
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None


class SymbolEmbedding(torch.autograd.Function):
    """nn.Embedding lookup of the text encoder (reference model.py:474-482,528).  Returns the (B,T,C) tensor of
    nn.Embedding as a transposed VIEW of a (B,C,T) buffer, so the reference's `.transpose(1, 2)` that follows hands the
    first encoder convolution a contiguous channel-major tensor without a copy."""

    @staticmethod
    def forward(ctx, ids, weight):
        lib = _require_gpu(ids, weight)
        ids = ids.contiguous().long()
        B, T = ids.shape
        n, Cc = weight.shape
        out = torch.empty(B, Cc, T, device=weight.device, dtype=torch.float32)
        _check(lib.t2v_embedding_fwd(_p(ids), _p(_f32c(weight)), _p(out), B, T, Cc, n, _stream()), 't2v_embedding_fwd')
        ctx.save_for_backward(ids)
        ctx.dims = (B, T, Cc, n)
        return out.transpose(1, 2)

    @staticmethod
    def backward(ctx, dy):
        lib = load_library()
        ids, = ctx.saved_tensors
        B, T, Cc, n = ctx.dims
        dy_bct = dy.transpose(1, 2)
        dy_bct = dy_bct if dy_bct.is_contiguous() else dy_bct.contiguous()
        dW = torch.empty(n, Cc, device=dy.device, dtype=torch.float32)
        _check(lib.t2v_embedding_bwd(_p(ids), _p(_f32c(dy_bct)), _p(dW), B, T, Cc, n, _stream()), 't2v_embedding_bwd')
        return None, dW


def mask_outputs(mel, mel_post, gate, lengths_i32, gate_fill=1e3):
    """Tacotron2.parse_output's three in-place fills (reference model.py:513-517) as one launch"""
    lib = _require_gpu(mel, mel_post, gate, lengths_i32)
    assert mel.is_contiguous() and mel_post.is_contiguous() and gate.is_contiguous() and lengths_i32.dtype == torch.int32
    assert mel.dtype == torch.float32 and mel_post.shape == mel.shape and gate.shape == (mel.size(0), mel.size(2))
    B, Cn, T = mel.shape
    _check(lib.t2v_mask_outputs(_p(mel), _p(mel_post), _p(gate), _p(lengths_i32), B, Cn, T, float(gate_fill), _stream()),
           't2v_mask_outputs')


class Reparam(torch.autograd.Function):
    """z = eps * exp(0.5 logvar) + mu (reference modules.py:74-81) and its gradient, one launch each"""

    @staticmethod
    def forward(ctx, eps, mu, logvar):
        lib = _require_gpu(eps, mu, logvar)
        eps, mu, logvar = _f32c(eps), _f32c(mu), _f32c(logvar)
        z = torch.empty_like(mu)
        _check(lib.t2v_reparam_fwd(_p(eps), _p(mu), _p(logvar), _p(z), mu.numel(), _stream()), 't2v_reparam_fwd')
        ctx.save_for_backward(eps, logvar)
        return z

    @staticmethod
    def backward(ctx, dz):
        eps, logvar = ctx.saved_tensors
        dz = _f32c(dz)
        dlv = torch.empty_like(logvar)
        _check(load_library().t2v_reparam_bwd(_p(dz), _p(eps), _p(logvar), _p(dlv), dz.numel(), _stream()), 't2v_reparam_bwd')
        return None, dz, dlv


class BiLSTM(torch.autograd.Function):
    """Encoder BiLSTM over per-sequence lengths (== pack_padded_sequence → nn.LSTM → pad_packed_sequence,
    reference model.py:183-190).  Input projections / weight gradients are time-batched GEMMs; the
    recurrence (forward and BPTT) runs in the persistent cooperative kernels of csrc/bilstm.hip, 16 sequences per
    call (larger batches run as independent chunks)."""

    _prep = {}      # id(w_hh) -> (whh, bias, bias_r, event): see prepare()

    @staticmethod
    def prepare(w_hh, b_ih, b_hh, w_hh_r, b_ih_r, b_hh_r):
        """the parameter-only operands of forward() (both W_hh stacked, b_ih + b_hh per direction), computed ahead — the
        model issues this on the deferred-work stream at the top of a step, the encoder reaches its BiLSTM 250 us later"""
        with torch.no_grad():
            whh = torch.stack((w_hh, w_hh_r)).contiguous()
            bias, bias_r = b_ih + b_hh, b_ih_r + b_hh_r
        ev = torch.cuda.Event()
        ev.record()
        BiLSTM._prep[id(w_hh)] = (whh, bias, bias_r, ev)

    @staticmethod
    def forward(ctx, x, lengths, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, save):
        lib = _require_gpu(x, w_ih)
        B, T, _ = x.shape
        f32 = dict(device=x.device, dtype=torch.float32)
        x = _f32c(x)
        prep = BiLSTM._prep.pop(id(w_hh), None)
        if prep is not None:        # stacked recurrent weights / summed biases of this step, made ahead on a side stream
            whh, bias, bias_r, ev = prep
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in (whh, bias, bias_r):
                t.record_stream(cur)
        else:
            whh = torch.stack((w_hh, w_hh_r)).contiguous()
            bias, bias_r = b_ih + b_hh, b_ih_r + b_hh_r
        y = torch.empty(B, T, 512, **f32)       # cleared by t2v_bilstm_fwd's reset launch
        chunks = []
        for b0 in range(0, B, MAX_DEC_B):
            b1 = min(B, b0 + MAX_DEC_B)
            Bc = b1 - b0
            gx = torch.empty(2, Bc, T, 1024, **f32)
            gemm(x[b0:b1].view(Bc * T, -1), w_ih, bias, out=gx[0].view(Bc * T, 1024))
            gemm(x[b0:b1].view(Bc * T, -1), w_ih_r, bias_r, out=gx[1].view(Bc * T, 1024))
            gates = torch.empty(2, Bc, T, 1024, **f32) if save else None
            cells = torch.empty(2, Bc, T, 256, **f32) if save else None
            hx = torch.empty(2 * 2 * 2 * 16 * 256, **f32)              # 8-byte granules
            sync = torch.empty(3, device=x.device, dtype=torch.int32)
            _check(lib.t2v_bilstm_fwd(_p(gx), _p(whh), _p(lengths[b0:b1]), _p(y[b0:b1]), _p(gates), _p(cells), _p(hx),
                                      _p(sync), Bc, T, _stream()), 't2v_bilstm_fwd')
            _err_note('BiLSTM forward', sync[2:3])
            chunks.append((b0, b1, gates, cells, sync))
        # (an alias of y, not y itself: the returned tensor gets this node as its grad_fn — keeping it here would tie the
        # node and its arenas into a reference cycle that only the cyclic GC frees, AccumulateGrad nodes included)
        ctx.keep = (x, lengths, w_ih, w_ih_r, whh, y.detach(), chunks)
        ctx.dims = (B, T)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load_library()
        x, lengths, w_ih, w_ih_r, whh, y, chunks = ctx.keep
        B, T = ctx.dims
        if chunks[0][2] is None:
            raise T2VHipError("BiLSTM forward ran without saving activations")
        f32 = dict(device=x.device, dtype=torch.float32)
        dy = _f32c(dy)
        dg = torch.empty(2, B, T, 1024, **f32) if len(chunks) == 1 else None     # cleared by t2v_bilstm_bwd
        dgs = []
        stamp('bilstm_bwd_begin')
        for b0, b1, gates, cells, sync in chunks:
            Bc = b1 - b0
            dg_c = dg if dg is not None else torch.empty(2, Bc, T, 1024, **f32)
            dgx = torch.empty(2 * 2 * 2 * 16 * 16 * 256, **f32)        # 8-byte granules: (dir, parity, producer, item, unit)
            _check(lib.t2v_bilstm_bwd(_p(whh), _p(lengths[b0:b1]), _p(dy[b0:b1]), _p(gates), _p(cells), _p(dg_c), _p(dgx),
                                      _p(sync), Bc, T, _stream()), 't2v_bilstm_bwd')
            _err_note('BiLSTM backward', sync[2:3])
            dgs.append(dg_c)
        stamp('bilstm_bwd_end')
        BT = B * T
        if dg is None:
            dg = torch.cat(dgs, 1)
        d0, d1, x2 = dg[0].reshape(BT, 1024), dg[1].reshape(BT, 1024), x.view(BT, -1)
        fork = mark()
        dx = gemm(d0, w_ih.t())
        gemm(d1, w_ih_r.t(), out=dx, accumulate=True)
        dx = dx.view(B, T, -1)
        db0, db1 = colsum(d0), colsum(d1)        # each handed to two inputs (b_ih, b_hh): autograd clones -> this stream
        with side('w', keep=(dg, d0, d1, dy), after=fork):          # weight gradients: deferred-work stream
            z = y.new_zeros(B, 1, 256)
            hp0 = torch.cat((z, y[:, :-1, :256]), 1).reshape(BT, 256)      # h_{t-1} of the forward direction
            hp1 = torch.cat((y[:, 1:, 256:], z), 1).reshape(BT, 256)       # h_{t+1} of the reverse direction
            g = (gemm(d0.t(), x2.t(), out=grad_slot(w_ih)), gemm(d0.t(), hp0.t()),
                 gemm(d1.t(), x2.t(), out=grad_slot(w_ih_r)), gemm(d1.t(), hp1.t()))
        return (dx, None, g[0], g[1], db0, db0, g[2], g[3], db1, db1, None)


def bilstm_check(sync):
    """Raises if a cooperative kernel reported a barrier timeout (forces a device sync; tests only)."""
    if int(sync[2].item()) != 0:
        raise T2VHipError("BiLSTM cooperative kernel timed out on its inter-workgroup barrier")


def colsum(x):
    """sum over the rows of a 2-D fp32 matrix (unit column stride): the `grad.sum(0)` of a bias, on t2v_colsum"""
    lib = _require_gpu(x)
    assert x.dim() == 2 and x.dtype == torch.float32
    if x.stride(1) != 1 and x.shape[1] > 1:
        x = x.contiguous()
    M, N = x.shape
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    nscr = lib.t2v_colsum_scratch_floats(M, N)
    scr = torch.empty(nscr, device=x.device, dtype=torch.float32) if nscr else None
    _check(lib.t2v_colsum(_p(x), x.stride(0) if M > 1 else N, M, N, _p(scr), _p(out), _stream()), 't2v_colsum')
    return out


# ---- gradient slots: FlatAdam keeps every parameter's gradient in one flat arena; a backward that knows the slot of
# its weight writes the gradient GEMM straight into it (autograd then adopts the returned view as `.grad` and
# FlatAdam.gather_grads() finds it already in place) instead of a fresh tensor that is copied into the arena later.
_GRAD_SLOTS = {}


def register_grad_slots(slots):
    """slots: {param.data_ptr(): (flat grads tensor, offset, shape)}; replaces the previous registration"""
    _GRAD_SLOTS.clear()
    _GRAD_SLOTS.update(slots)


def grad_slot(weight):
    """a fresh view of the arena slot that holds d(weight), or None (= let the GEMM allocate) when no slot is
    registered or the parameter already carries a gradient (a second backward before zero_grad must ADD to it)"""
    if not weight.is_leaf or weight.grad is not None:
        return None
    ent = _GRAD_SLOTS.get(weight.data_ptr())
    if ent is None:
        return None
    flat, off, shape = ent
    if tuple(shape) != tuple(weight.shape) or flat.device != weight.device:
        return None
    return flat[off:off + weight.numel()].view(shape)


def _small_grads(params, n, f32):
    """gradient tensors for (gamma, beta, conv bias): their arena slots when registered, else one fresh tensor each"""
    out = []
    for p in params:
        g = grad_slot(p) if p is not None else None
        out.append(torch.empty(n, **f32) if g is None else g)
    return out


def gemm_grouped(groups, accumulate=False):
    """groups: [(A (M,K), [(B_p (N_p,K), out_p (M,N_p)), ...]), ...] (at most two groups of at most three parts, one common M and K):
    out_p (+)= A · B_p^T for every part.  Every operand is split (fp32: into its three bf16 planes; bf16_run: rounded to bf16) once, all
    tiles run as ONE launch (t2v_gemm_f32_grouped / t2v_gemm_bf16_grouped); shapes the grouped kernel does not take: the products one by
    one through gemm()."""
    lib = _require_gpu(groups[0][0])
    M, K = groups[0][0].shape
    ok = 1 <= len(groups) <= 2
    for A, parts in groups:
        ok = ok and A.shape == (M, K) and A.dtype == torch.float32 and 1 <= len(parts) <= 3
        for Bp, out in parts:
            ok = ok and Bp.shape[1] == K and Bp.dtype == torch.float32 and out.shape == (M, Bp.shape[0]) and out.stride(1) == 1 and Bp.shape[0] % 128 == 0
    if not ok:
        for A, parts in groups:
            for Bp, out in parts:
                gemm(A, Bp, out=out, accumulate=accumulate)
        return
    arr = (_GemmGroup * len(groups))()
    for g, (A, parts) in zip(arr, groups):
        g.A, g.sAi, g.sAk, g.nb = A.data_ptr(), A.stride(0), A.stride(1), len(parts)
        for p, (Bp, out) in enumerate(parts):
            g.B[p], g.sBj[p], g.sBk[p], g.N[p] = Bp.data_ptr(), Bp.stride(0), Bp.stride(1), Bp.shape[0]
            g.C[p], g.ldc[p] = out.data_ptr(), out.stride(0)
    size_fn, run_fn = ((lib.t2v_gemm_bf16_grouped_scratch_floats, lib.t2v_gemm_bf16_grouped) if _BF16 else
                       (lib.t2v_gemm_f32_grouped_scratch_floats, lib.t2v_gemm_f32_grouped))
    nscr = size_fn(arr, len(groups), M, K)
    scr = torch.empty(max(nscr, 4), device=groups[0][0].device, dtype=torch.float32)
    _check(run_fn(arr, len(groups), M, K, int(bool(accumulate)), _p(scr), _stream()), 't2v_gemm_grouped')


def set_f32_gemm_mode(x3):
    """True (default): the large fp32 products run as six bf16 MFMAs on exactly 3-way-split operands (fp32-class accuracy, see
    include/t2vae.h: t2v_gemm_f32_set_mode); False: fp32 MFMA only; None: query.  Returns the previous setting."""
    return bool(load_library().t2v_gemm_f32_set_mode(-1 if x3 is None else int(bool(x3))))


def gemm(A, B, bias=None, out=None, relu=False, accumulate=False, p_drop=0.0, seed=0, rng_stream=0, rng_t=0):
    """out[i][j] (+)= sum_k A[i][k] * B[j][k] (+ bias[j]) on the HIP MFMA GEMM; A (M,K), B (N,K) may be any
    2-D strided views (pass W for x·W^T, W.t() for dy·W, dy.t()/x.t() for dy^T·x)."""
    lib = _require_gpu(A, B)
    M, K = A.shape
    N, K2 = B.shape
    assert K == K2 and A.dtype == torch.float32 and B.dtype == torch.float32
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    # `out` may be a column block of a wider row-major matrix (row stride ldc >= N): the weight-gradient GEMMs write
    # the [weight_ih | weight_hh] halves of one product straight into their own tensors
    assert out.shape == (M, N) and out.dtype == torch.float32 and (N == 1 or out.stride(1) == 1)
    ldc = out.stride(0) if M > 1 else max(N, out.stride(0))
    assert ldc >= N and (ldc == N or p_drop == 0.0)
    # skinny deep-K product (a handful of tiles, K = T*B): split K over workgroups, fixed-order partial sum.  Under bf16_run
    # too: the bf16 64x64 kernel has no split-K, and on 8..48 workgroups with K = 6400 these products took 190-200 us each
    # (B = 16 step trace) against 20-30 us on the fp32 split-K kernel — fp32 operands are no loss of accuracy.
    if True:
        nscr = lib.t2v_gemm_splitk_scratch_floats(M, N, K)
        if nscr and _BF16 and ((M + 63) // 64) * ((N + 63) // 64) >= 64:
            nscr = 0        # enough tiles for the bf16 kernels (the LSTM weight gradients go to the 128x128 bf16 GEMM)
        if nscr:
            scr = torch.empty(nscr, device=A.device, dtype=torch.float32)
            _check(lib.t2v_gemm_f32_splitk(_p(A), A.stride(0), A.stride(1), _p(B), B.stride(0), B.stride(1), _p(bias), _p(out),
                                           ldc, M, N, K, int(relu), int(accumulate), float(p_drop), int(seed), int(rng_stream),
                                           int(rng_t), _p(scr), _stream()), 't2v_gemm_f32_splitk')
            return out
    if _BF16:
        # a bf16 kernel on a grid that leaves CUs idle with a deep K (the deferred LSTM weight gradients on the 128x128 tile; the
        # projection, the BiLSTM data / weight gradients on the 64x64 tile): split over k, fixed-order sum of the partials
        nscr = lib.t2v_gemm_bf16_splitk_scratch_floats(M, N, K)
        if nscr:
            scr = torch.empty(nscr, device=A.device, dtype=torch.float32)
            _check(lib.t2v_gemm_bf16_splitk(_p(A), A.stride(0), A.stride(1), _p(B), B.stride(0), B.stride(1), _p(bias), _p(out),
                                            ldc, M, N, K, int(relu), int(accumulate), float(p_drop), int(seed), int(rng_stream),
                                            int(rng_t), _p(scr), _stream()), 't2v_gemm_bf16_splitk')
            return out
    fn = lib.t2v_gemm_bf16 if _BF16 else lib.t2v_gemm_f32
    _check(fn(_p(A), A.stride(0), A.stride(1), _p(B), B.stride(0), B.stride(1), _p(bias), _p(out), ldc,
              M, N, K, int(relu), int(accumulate), float(p_drop), int(seed), int(rng_stream),
              int(rng_t), _stream()), 't2v_gemm_bf16' if _BF16 else 't2v_gemm_f32')
    return out


class LinearHIP(torch.autograd.Function):
    """y = dropout(relu?(x·W^T + b)) over the last dimension of x, on t2v_gemm_f32 (forward and both
    gradients).  The dropout mask is regenerated in the backward from (seed, stream, t)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, p_drop, seed, rng_stream, rng_t):
        _require_gpu(x, weight)
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.dtype == torch.float32 else x2.float()
        y = gemm(x2, weight, bias, relu=relu, p_drop=p_drop, seed=seed, rng_stream=rng_stream, rng_t=rng_t)
        ctx.save_for_backward(x2, weight, y if (relu or p_drop > 0) else None)
        ctx.cfg = (x.shape, bool(relu), float(p_drop), bias is not None)
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        xshape, relu, p, has_bias = ctx.cfg
        dy2 = _f32c(dy.reshape(-1, weight.shape[0]))
        if relu or p > 0:
            # y already carries relu and the 1/(1-p) scaling: d/dpre = dy * [y != 0] * scale
            scale = 1.0 / (1.0 - p) if p > 0 else 1.0
            masked = torch.empty_like(dy2)
            _check(load_library().t2v_gemm_epilogue_bwd(_p(dy2), _p(y), _p(masked), dy2.numel(), scale, _stream()),
                   't2v_gemm_epilogue_bwd')
            dy2 = masked
        fork = mark()
        dx = gemm(dy2, weight.t()) if ctx.needs_input_grad[0] else None          # (N,K) = dy · W
        # Deferral rule (also in the other nodes): only a gradient that travels STRAIGHT into a leaf's AccumulateGrad as
        # the sole reference (autograd adopts the tensor without touching its data) may be produced on the deferred-work
        # stream.  Anything autograd may read on the node's own stream before the engine's join — a tensor handed to two
        # inputs (cloned), a gradient that flows on through Cat/Add/Slice nodes — is produced on the node's stream.
        if weight.is_leaf:
            with side('w', keep=(dy2, x2), after=fork):     # weight gradient: deferred-work stream of the engine
                dw = gemm(dy2.t(), x2.t(), out=grad_slot(weight))                  # (M,K) = dy^T · x
        else:
            dw = gemm(dy2.t(), x2.t())
        db = colsum(dy2) if has_bias else None
        return (dx.view(xshape) if dx is not None else None), dw, db, None, None, None, None, None


# reference-encoder convolutions as im2col + batched GEMM (T2V_CONV2D_GEMM=0: the direct-form kernels)
CONV2D_GEMM = os.environ.get('T2V_CONV2D_GEMM', '1') != '0'


class Conv2dBNReLU(torch.autograd.Function):
    """relu(BatchNorm2d(Conv2d 3x3 s2 p1 (x [+ CoordConv channels]))) — one layer of the reference encoder
    (reference modules.py:68-71) on the direct-form HIP conv + the per-channel BN kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, coord):
        lib = _require_gpu(x, weight)
        x = _f32c(x)
        B, Cx, Hh, Ww = x.shape
        Cout = weight.shape[0]
        Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
        f32 = dict(device=x.device, dtype=torch.float32)
        w = weight.contiguous()
        y = torch.empty(B, Cout, Ho, Wo, **f32)
        gscr = None
        if CONV2D_GEMM:      # im2col + batched MFMA GEMM (the im2col matrix stays in gscr for the backward pass)
            gscr = torch.empty(lib.t2v_conv2d_s2_gemm_scratch_floats(B, Cx, Hh, Ww, Cout, int(coord)), **f32)
            _check(lib.t2v_conv2d_s2_fwd_gemm(_p(x), _p(w), _p(bias), _p(y), _p(gscr), B, Cx, Hh, Ww, Cout, int(coord), _stream()),
                   't2v_conv2d_s2_fwd_gemm')
        else:
            _check(lib.t2v_conv2d_s2_fwd(_p(x), _p(w), _p(bias), _p(y), B, Cx, Hh, Ww, Cout, int(coord), _stream()),
                   't2v_conv2d_s2_fwd')
        mean = torch.empty(Cout, **f32) if training else None
        rstd = torch.empty(Cout, **f32) if training else None
        out = torch.empty_like(y)
        _check(lib.t2v_bn_act_fwd(_p(y), None, 0, _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(mean),
                                  _p(rstd), _p(out), B, Cout, Ho * Wo, ACT_RELU, int(bool(training)), 0.0, 0.1, 1e-5,
                                  0, 0, 0, _stream()), 't2v_bn_act_fwd')
        ctx.keep = (x, w, y, mean, rstd, gamma, beta)
        ctx.gscr = gscr
        ctx.running = (running_mean, running_var)
        ctx.small = (gamma, beta, bias)
        ctx.cfg = (B, Cx, Hh, Ww, Cout, Ho, Wo, int(coord), bool(training))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = load_library()
        x, w, y, mean, rstd, gamma, beta = ctx.keep
        B, Cx, Hh, Ww, Cout, Ho, Wo, coord, training = ctx.cfg
        bn_bwd = lib.t2v_bn_act_bwd
        if not training:        # eval-mode BatchNorm2d: running statistics are constants
            mean, rstd = ctx.running[0], torch.rsqrt(ctx.running[1] + 1e-5)
            bn_bwd = lib.t2v_bn_act_bwd_eval
        f32 = dict(device=x.device, dtype=torch.float32)
        dout = dout.contiguous()
        dy = torch.empty_like(y)
        dgamma, dbeta, dbias = _small_grads(ctx.small, Cout, f32)
        _check(bn_bwd(_p(y), _p(dout), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dy), _p(dgamma),
                      _p(dbeta), _p(dbias), B, Cout, Ho * Wo, ACT_RELU, 0.0, 0, 0, 0, _stream()), 't2v_bn_act_bwd')
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        if ctx.gscr is not None:
            _check(lib.t2v_conv2d_s2_bwd_gemm(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(ctx.gscr), B, Cx, Hh, Ww, Cout, coord,
                                              _stream()), 't2v_conv2d_s2_bwd_gemm')
            ctx.gscr = None
        else:
            nscr = lib.t2v_conv2d_s2_dw_scratch_floats(B, Cx, Hh, Ww, Cout, coord)
            scr = torch.empty(nscr, **f32) if nscr else None
            _check(lib.t2v_conv2d_s2_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(scr), B, Cx, Hh, Ww, Cout, coord,
                                         _stream()), 't2v_conv2d_s2_bwd')
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None


class GRULast(torch.autograd.Function):
    """Last hidden state of nn.GRU(batch_first) (reference modules.py:78-80)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        lib = _require_gpu(x, w_ih)
        B, T, I = x.shape
        f32 = dict(device=x.device, dtype=torch.float32)
        x2 = _f32c(x).view(B * T, I)
        gi = gemm(x2, w_ih, b_ih)                                   # (B*T,768)
        hs = torch.empty(B, T + 1, 256, **f32)
        gsave = torch.empty(B, T, 4, 256, **f32)
        whh = w_hh.contiguous()
        xchg = torch.empty(2 * 16 * 768, **f32)                       # exchange buffer (forward uses 2*16*256 of it)
        syncs = []
        gi3 = gi.view(B, T, 768)
        for b0 in range(0, B, MAX_DEC_B):          # 16 sequences per cooperative launch; rows of a chunk are contiguous
            b1 = min(B, b0 + MAX_DEC_B)
            sync = torch.empty(2, device=x.device, dtype=torch.int32)
            _check(lib.t2v_gru_fwd(_p(gi3[b0:b1]), _p(whh), _p(b_hh), _p(hs[b0:b1]), _p(gsave[b0:b1]), _p(xchg), _p(sync),
                                   b1 - b0, T, _stream()), 't2v_gru_fwd')
            _err_note('GRU forward', sync[1:2])
            syncs.append(sync)
        ctx.keep = (x2, w_ih, whh, hs, gsave, xchg, syncs)
        ctx.dims = (B, T, I)
        return hs[:, T].clone()

    @staticmethod
    def backward(ctx, dh):
        lib = load_library()
        x2, w_ih, whh, hs, gsave, xchg, syncs = ctx.keep
        B, T, I = ctx.dims
        f32 = dict(device=x2.device, dtype=torch.float32)
        dgi, dgh = torch.empty(B, T, 768, **f32), torch.empty(B, T, 768, **f32)
        dh = dh.contiguous()
        for i, b0 in enumerate(range(0, B, MAX_DEC_B)):
            b1 = min(B, b0 + MAX_DEC_B)
            _check(lib.t2v_gru_bwd(_p(whh), _p(hs[b0:b1]), _p(gsave[b0:b1]), _p(dh[b0:b1]), _p(dgi[b0:b1]), _p(dgh[b0:b1]),
                                   _p(xchg), _p(syncs[i]), b1 - b0, T, _stream()), 't2v_gru_bwd')
            _err_note('GRU backward', syncs[i][1:2])
        dgi2, dgh2 = dgi.view(B * T, 768), dgh.view(B * T, 768)
        fork = mark()
        dx = gemm(dgi2, w_ih.t()).view(B, T, I)
        with side('w', keep=(dgi, dgh, hs, x2), after=fork):
            hprev = hs[:, :T].reshape(B * T, 256)
            g = (gemm(dgi2.t(), x2.t()), gemm(dgh2.t(), hprev.t()))
        return (dx,) + g + (colsum(dgi2), colsum(dgh2))


class VAELoss(torch.autograd.Function):
    """Tacotron2Loss_VAE value + gradient in one HIP launch (reference loss_function.py:27-44)."""
    _scratch = {}

    @staticmethod
    def forward(ctx, mel, post, gate, mu, logvar, mel_t, gate_t, kl_weight):
        lib = _require_gpu(mel, post, gate, mu, logvar)
        dev = mel.device
        key = str(dev)
        if key not in VAELoss._scratch:
            VAELoss._scratch[key] = (torch.empty(192, device=dev), torch.zeros(1, device=dev, dtype=torch.int32))
        part, ticket = VAELoss._scratch[key]
        mel, post, gate, mu, logvar = (_f32c(t) for t in (mel, post, gate, mu, logvar))
        mel_t, gate_t = _f32c(mel_t), _f32c(gate_t)
        out = torch.empty(4, device=dev, dtype=torch.float32)
        sp = step_params(create=False)
        if sp is not None:       # the kernel reads the KL weight from the device record once one is installed
            sp.set(kl_weight=kl_weight)
            sp.upload()
        grads = [torch.empty_like(t) for t in (mel, post, gate, mu, logvar)]
        _check(lib.t2v_loss_fwd_bwd(_p(mel), _p(post), _p(mel_t), _p(gate), _p(gate_t), _p(mu), _p(logvar),
                                    *[_p(g) for g in grads], _p(part), _p(out), _p(ticket), mel.numel(),
                                    gate.numel(), mu.numel(), float(kl_weight), _stream()), 't2v_loss_fwd_bwd')
        ctx.grads = grads
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, dtotal, _dout):
        g = torch._foreach_mul(ctx.grads, dtotal)       # one multi-tensor launch instead of five
        return (g[0], g[1], g[2], g[3], g[4], None, None, None)
