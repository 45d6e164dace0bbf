"""ctypes binding of libfootprints_hip.so (the C ABI declared in include/footprints_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel
reports an error, a RuntimeError is raised -- nothing silently routes to PyTorch
ops or to the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfootprints_hip.so")

# enum fp_gather
GATHER_FWD_ZERO, GATHER_FWD_REFLECT, GATHER_FWD_REFLECT_UP2, GATHER_DGRAD_ZERO, GATHER_DGRAD_REFLECT, GATHER_STEM = range(6)
ACT_NONE, ACT_ELU, ACT_RELU = range(3)
EPI_BIAS, EPI_ADDEND, EPI_ADDEND_MASK, EPI_ACTGRAD_ELU, EPI_ACTGRAD_RELU, EPI_ACCUM, EPI_BF16X2 = 1, 2, 4, 8, 16, 32, 64


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "OH", "OW", "IH", "IW", "C0", "C1", "Nout", "KH", "KW", "stride", "pad",
                                         "gather", "act")] + [("epi", C.c_uint32)]


PACK_FWD, PACK_DGRAD, PACK_STEM, PACK_UP2_FWD, PACK_UP2_DGRAD, PACK_FWD_BF3, PACK_DGRAD_BF3, PACK_UP2_FWD_BF3, PACK_UP2_DGRAD_BF3 = range(9)
PACK_FWD_HP, PACK_DGRAD_HP, PACK_UP2_FWD_HP, PACK_UP2_DGRAD_HP, PACK_STEM_HP = range(9, 14)


class PackJob(C.Structure):
    """fp_pack_job (include/footprints_hip.h): one entry of the device-resident table of fp_pack_weights_batched"""
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p)] + [(n, C.c_int32) for n in (
        "Cout", "Cin", "KH", "KW", "kind", "c_begin", "c_count", "block_begin", "block_count")] + [("amax", C.c_void_p)]


_P, _I32, _I64, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_DESC = C.POINTER(ConvDesc)


class Aux(C.Structure):
    """fp_aux (include/footprints_hip.h): optional side outputs of ONE launch, passed explicitly (round 6; rounds 3-5 armed per-thread
    `*_out_next` sinks instead) -- an amax slot receiving max |output|, BatchNorm partials out of a convolution's epilogue"""
    _fields_ = [("amax_out", C.c_void_p), ("bn_part", C.c_void_p), ("bn_capacity_floats", C.c_int64), ("bn_nblk_out", C.POINTER(C.c_int32)),
                ("bnb_z", C.c_void_p), ("bnb_mean", C.c_void_p), ("bnb_invstd", C.c_void_p)]


_AUX = C.POINTER(Aux)

# name -> (restype, argtypes); mirrors include/footprints_hip.h one to one
SIGNATURES = {
    "fp_seg_loss_workspace": (_I64, [_I32, _I32, _I32]),
    "fp_seg_loss_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _I64, _P]),
    "fp_comm_unique_id_bytes": (_I32, []),
    "fp_comm_unique_id": (C.c_int, [_P, _I32]),
    "fp_comm_init": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_void_p)]),
    "fp_comm_version": (_I32, []),
    "fp_comm_count": (_I32, [_P]),
    "fp_comm_allreduce_async": (C.c_int, [_P, _P, _I64, _P]),
    "fp_comm_broadcast": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "fp_comm_wait": (C.c_int, [_P, _P, _P]),
    "fp_comm_destroy": (C.c_int, [_P]),
    "fp_ktime_begin": (C.c_int, []),
    "fp_ktime_end": (_I32, []),
    "fp_ktime_row": (C.c_int, [_I32, C.c_char_p, _I32, C.POINTER(_I64), C.POINTER(_D)]),
    "fp_plan_begin": (_P, []),
    "fp_plan_mark": (_I32, [_P]),
    "fp_plan_end": (_I32, [_P]),
    "fp_plan_replay": (C.c_int, [_P, _I32, _I32]),
    "fp_plan_destroy": (None, [_P]),
    "fp_event_record": (_I64, [_P]),
    "fp_event_wait": (C.c_int, [_P, _I64]),
    "fp_aug_params_bytes": (_I32, []),
    "fp_assemble_images": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "fp_assemble_labels": (C.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _D, _D, _D, _P]),
    "fp_adaptive_avgpool_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_adaptive_avgpool_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, C.c_int, _P]),
    "fp_bilinear_ac_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_bilinear_ac_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_copy_channels": (C.c_int, [_P, _P, _I64, _I32, _I32, _I32, _I32, _I32, C.c_int, _P]),
    "fp_conv_igemm_workspace": (_I64, [_DESC]),
    "fp_conv_igemm": (C.c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _AUX, _P]),
    "fp_conv_wgrad_workspace": (_I64, [_DESC]),
    "fp_conv_wgrad": (C.c_int, [_DESC, _P, _P, _P, _P, C.c_int, _P, _I64, _P]),
    "fp_packed_weight_elems": (_I64, [_I32, _I32, _I32, _I32, _I32, _I32]),
    "fp_pack_conv_weight": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_pack_conv_weight_dgrad": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_up2_packed_weight_elems": (_I64, [_I32, _I32]),
    "fp_pack_up2_weight": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_pack_conv_weight_slice": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_conv_up2_phase_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_pack_up2_weight_dgrad": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_pack_conv_weight_dgrad_slice": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_up2_fold_bwd": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _AUX, _P]),
    "fp_conv_wgrad_slice": (C.c_int, [_DESC, _P, _P, _P, _P, _I32, _I32, C.c_int, _P, _I64, _P]),
    "fp_conv_up2_phase_wgrad_workspace": (_I64, [_I32, _I32, _I32, _I32, _I32]),
    "fp_conv_up2_phase_wgrad": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, C.c_int, _P, _I64, _P]),
    "fp_conv_up2_phase_wgrad_bf3": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, C.c_int, _P, _I64, _P]),
    "fp_pack_job_blocks": (_I32, [_I32, _I32, _I32, _I32, _I32]),
    "fp_pack_weights_batched": (C.c_int, [_P, _P, _I32, _P]),
    "fp_pack_weights_batched_capped": (C.c_int, [_P, _P, _I32, _I32, _P]),
    "fp_pack_up2_weight_bf3": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_pack_up2_weight_dgrad_bf3": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_conv_up2_phase_dgrad_bf3": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_conv_up2_phase_fwd_bf3": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _AUX, _P]),
    "fp_conv_wgrad_bf3_workspace": (_I64, [_DESC]),
    "fp_conv_wgrad_bf3": (C.c_int, [_DESC, _P, _P, _P, _P, _I32, _I32, C.c_int, _P, _I64, _P]),
    "fp_conv3x3_bf3_supported": (C.c_int, [_DESC]),
    "fp_conv3x3_bf3_workspace": (_I64, [_DESC]),
    "fp_conv3x3_bf3": (C.c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _AUX, _P]),
    "fp_packed_weight_elems_bf3": (_I64, [_I32, _I32, _I32, _I32, _I32]),
    "fp_amax_slot_elems": (_I32, []),
    "fp_conv_wgrad_hp": (C.c_int, [_DESC, _P, _P, _P, _P, _I32, _I32, C.c_int, _P, _I64, _P, _P, _P]),
    "fp_conv_up2_phase_wgrad_hp": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, C.c_int, _P, _I64, _P, _P, _P]),
    "fp_conv_up2_phase_fwd_hp": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "fp_conv_up2_phase_dgrad_hp": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "fp_zero_u32": (C.c_int, [_P, _I64, _P]),
    "fp_amax_f32": (C.c_int, [_P, _I64, _P, _P]),
    "fp_weight_amax": (C.c_int, [_P, _I64, _P, _P]),
    "fp_packed_weight_elems_hp": (_I64, [_I32, _I32, _I32, _I32, _I32]),
    "fp_pack_conv_weight_hp": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "fp_pack_weights_amax": (C.c_int, [_P, _P, _I32, _P]),
    "fp_conv3x3_hp": (C.c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _AUX, _P]),
    "fp_pack_conv_weight_bf3": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_adam_hyper": (C.c_int, [_D, _D, _D, _D, _I32, _D, _P]),
    "fp_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P]),
    "fp_scale_rows": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "fp_eval_mask_counts": (C.c_int, [_P, _I32, _P, _P, _I32, _I32, _I64, _I64, _P, _P]),
    "fp_eval_depth_sums": (C.c_int, [_P, _I32, _P, _I32, _I64, _I64, C.c_double, C.c_double, C.c_double, C.c_double, _P, _P]),
    "fp_pack_pred_fp16": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "fp_colsum_workspace": (_I64, [_I64, _I32]),
    "fp_colsum": (C.c_int, [_P, _I64, _I32, _P, C.c_int, _P, _I64, _P]),
    "fp_up2cat_bwd": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, C.c_int, _P]),
    "fp_head_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_head_upsample": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_head_upsample_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "fp_head_dgrad": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _AUX, _P]),
    "fp_conv_stem_hp_supported": (C.c_int, [_P]),
    "fp_conv_stem_hp": (C.c_int, [_P, _P, _P, _P, _P, _P, _AUX, _P]),
    "fp_conv_stem_wgrad_hp": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _I64, _P, _P]),
    "fp_head_wgrad_workspace": (_I64, [_I32, _I32, _I32, _I32]),
    "fp_head_wgrad": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, C.c_int, _P, _I64, _P]),
    "fp_bn_workspace": (_I64, [_I64, _I32]),
    "fp_bn_train_stats": (C.c_int, [_P, _I64, _I32, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "fp_bn_train_stats_partials": (C.c_int, [_P, _I32, _I32, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fp_bn_bwd_partials": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _I64, _I32, _P, _I32, _P, _AUX, _P]),
    "fp_bn_eval_coeffs": (C.c_int, [_P, _P, _P, _P, _F, _I32, _P, _P, _P]),
    "fp_conv_igemm_hp_supported": (C.c_int, [_DESC]),
    "fp_conv_igemm_hp": (C.c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _AUX, _P]),
    "fp_conv_igemm_bf3": (C.c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _AUX, _P]),
    "fp_bn_apply": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _AUX, _P]),
    "fp_bn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _I64, _I32, _P, _I64, _AUX, _P]),
    "fp_maxpool_fwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _AUX, _P]),
    "fp_maxpool_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, C.c_int, _P]),
    "fp_loss_workspace": (_I64, [_I32, _I32, _I32]),
    "fp_loss_fwd_bwd": (C.c_int, [C.POINTER(_P), _P, _P, _P, _P, _P, _P, _F, _F, _F, C.POINTER(_P), _P, _I32, _I32, _I32,
                                  _P, _I64, _P]),
    "fp_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _D, _D, _D, _D, _I32, _D, _P]),
    "fp_nchw_to_nhwc": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_nhwc_to_nchw": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "fp_fill": (C.c_int, [_P, _I64, _F, _P]),
    "fp_clock_probe": (C.c_int, [_P, _P]),
    "fp_wall_clock_khz": (C.c_int, []),
    "fp_mfma_probe": (C.c_int, [_P, _P, _I32, _I32, _P]),
    "fp_mfma_probe_flop": (C.c_double, [_I32]),
    "fp_version": (C.c_int, []),
    "fp_last_error_string": (C.c_char_p, []),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("FP_LIB", LIB_PATH)           # FP_LIB: another build of the same ABI (A/B measurements)
    if not os.path.exists(path):
        raise RuntimeError(
            "footprints_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C footprints_amd/csrc`). There is no CPU / PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_SYNC_ONLY = os.environ.get("FP_SYNC_ONLY", "")      # debugging aid: synchronise only after calls whose name contains this
_SYNC = bool(int(os.environ.get("FP_SYNC", "0")))     # debugging aid: device-synchronise after every C-ABI call


def check(rc, what=""):
    if _SYNC or (_SYNC_ONLY and _SYNC_ONLY in what):
        import torch
        torch.cuda.synchronize()
    if rc != 0:
        msg = load().fp_last_error_string()
        raise RuntimeError("footprints_hip %s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))
