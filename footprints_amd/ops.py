"""Tensor-level wrappers over the C ABI (include/footprints_hip.h).

Every function takes contiguous float32 CUDA tensors, launches on torch's
current stream and returns immediately (async).  PyTorch only owns memory and
streams here; there is no torch compute op and no fallback in this module.
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import ConvDesc

_workspaces = {}
# allocation generation: bumped whenever a buffer that kernels address by raw pointer is (re)allocated -- the activation arena
# (Engine.buf), a workspace below, the flat parameter / gradient buffers, the pack tables.  A recorded launch plan freezes raw
# pointers; TrainStep compares the generation it recorded at with the current one and drops its plans on a mismatch.
_alloc_gen = [0]


def alloc_generation():
    return _alloc_gen[0]


def bump_alloc_generation():
    _alloc_gen[0] += 1
_NO_SPLITK = bool(int(__import__("os").environ.get("FP_NO_SPLITK", "0")))   # debugging aid: never split small grids along K


def _chk(t, name="tensor"):
    if t is None:
        return 0
    if not (t.is_cuda and t.is_contiguous()):
        raise RuntimeError("footprints_amd.ops: %s must be a contiguous CUDA tensor (no CPU path in the product)" % name)
    return t.data_ptr()


def _f32(t, name="tensor"):
    if t is not None and t.dtype != torch.float32:
        raise RuntimeError("footprints_amd.ops: %s must be float32" % name)
    return _chk(t, name)


_cur_stream = None      # (torch stream, raw handle) while inside on_stream(): saves ~900 torch.cuda.current_stream() calls per step


def stream():
    return _cur_stream[1] if _cur_stream is not None else torch.cuda.current_stream().cuda_stream


class on_stream:
    """`with torch.cuda.stream(s)` that also caches the raw handle for the launch wrappers of this module"""

    def __init__(self, s):
        self.s = s
        self.ctx = torch.cuda.stream(s)

    def __enter__(self):
        global _cur_stream
        self.prev = _cur_stream
        self.ctx.__enter__()
        _cur_stream = (self.s, self.s.cuda_stream)
        return self.s

    def __exit__(self, *exc):
        global _cur_stream
        _cur_stream = self.prev
        return self.ctx.__exit__(*exc)


def current_stream():
    return _cur_stream[0] if _cur_stream is not None else torch.cuda.current_stream()


def event_record(stream_obj=None):
    """record a library event on `stream_obj` (default: the current launch stream) -> event id for event_wait"""
    h = stream_obj.cuda_stream if stream_obj is not None else stream()
    ev = _lib.load().fp_event_record(h)
    if ev < 0:
        raise RuntimeError("fp_event_record: %s" % _lib.load().fp_last_error_string().decode())
    return ev


def event_wait(stream_obj, ev):
    _lib.check(_lib.load().fp_event_wait(stream_obj.cuda_stream, ev), "fp_event_wait")


def stream_wait_stream(consumer, producer):
    """everything queued on `producer` so far happens before whatever is queued on `consumer` from now on"""
    event_wait(consumer, event_record(producer))


_query_cache = {}


def _cached_query(fn, desc):
    """shape predicates / workspace sizes are pure functions of the descriptor: ask the library once per distinct shape"""
    key = (fn, bytes(desc))
    v = _query_cache.get(key)
    if v is None:
        v = _query_cache[key] = getattr(_lib.load(), fn)(C.byref(desc))
    return v


_RELEASE_SYNC = bool(int(os.environ.get("FP_RELEASE_SYNC", "1")))
_release_hooks = []      # test instrumentation (tests/test_gpu_lifetime.py): callables (old tensor, what) run whenever a buffer is let go


def release(old, what):
    """Every place that REPLACES a device buffer which earlier launches were handed (a workspace that grows, an arena buffer of the engine
    that is re-allocated) lets go of the old one through here.  Kernels that follow on the stream are ordered behind the launches that used
    it; what the HOST does next is not -- the caching allocator hands the block out again at once, and a small pageable host-to-device copy
    into it does not wait for kernels still writing there (round 5, profiles/round5_notes.md section 7: a split-K convolution's late partial
    sums landed on a pack-job table).  So: wait for the device first (start-up and shape changes only; never under stream capture, where the
    pool keeps captured blocks alive), then tell the hooks -- the lifetime stress test fills the old block with a sentinel on an idle stream
    and checks afterwards that nothing wrote into it (round 6, VERDICT r5 "Next" 9)."""
    if old is None:
        return
    if _RELEASE_SYNC and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize(old.device)
    for hook in _release_hooks:
        hook(old, what)


def workspace(nbytes, device, tag="main"):
    """Grow-only scratch buffer per (device, stream, tag); a growth lets go of the old buffer through `release` (which says why it waits)."""
    # one scratch buffer per (device, stream, tag): kernels on concurrent streams must not share partials
    key = (device.index, stream(), tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        release(ws, "workspace:%s" % tag)
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
        bump_alloc_generation()
    return ws


def make_desc(N, OH, OW, IH, IW, C0, C1, Nout, K, stride, pad, gather, act=0, epi=0):
    return ConvDesc(N, OH, OW, IH, IW, C0, C1, Nout, K, K, stride, pad, gather, act, epi)


def conv_igemm(desc, src0, src1, wpacked, y, bias=None, addend=None, addend_mask=None, actsrc=None, bn_out=None):
    lib = _lib.load()
    epi = desc.epi
    if bias is not None:
        epi |= _lib.EPI_BIAS
    if addend is not None:
        epi |= _lib.EPI_ADDEND
    if addend_mask is not None:
        epi |= _lib.EPI_ADDEND_MASK
    d = ConvDesc.from_buffer_copy(desc)
    d.epi = epi
    need = 0 if _NO_SPLITK else _cached_query("fp_conv_igemm_workspace", d)
    ws_ptr, ws_n = 0, 0
    if need > 0:
        ws = workspace(need, y.device, "igemm")
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    _lib.check(lib.fp_conv_igemm(C.byref(d), _f32(src0, "src0"), _f32(src1, "src1"), _f32(wpacked, "wpacked"), _f32(bias),
                                 _f32(addend), _f32(addend_mask), _f32(actsrc), _f32(y, "y"), ws_ptr, ws_n, _aux(bn_out=bn_out), stream()), "fp_conv_igemm")
    return y


def conv_stem_hp_supported(desc):
    return bool(_cached_query("fp_conv_stem_hp_supported", desc))


def conv_stem_hp(desc, img, wpacked_hp, y, amax_w, bias=None, amax_out=None, bn_out=None):
    """the 7x7 / 2 stem on the NCHW image with fp16-pair operands (weights from a PACK_STEM_HP job); optional side outputs: amax_out, bn_out"""
    d = ConvDesc.from_buffer_copy(desc)
    if bias is not None:
        d.epi |= _lib.EPI_BIAS
    _lib.check(_lib.load().fp_conv_stem_hp(C.byref(d), _f32(img), _f32(wpacked_hp), _f32(bias), _f32(y), _u32(amax_w, "amax_w"),
                                           _aux(amax_out, bn_out), stream()),
               "fp_conv_stem_hp")
    return y


def conv_stem_wgrad_hp(desc, img, dz, dw, amax_dz, accumulate=False):
    lib = _lib.load()
    ws = workspace(_cached_query("fp_conv_wgrad_workspace", desc), dz.device)
    _lib.check(lib.fp_conv_stem_wgrad_hp(C.byref(desc), _f32(img), _f32(dz), _f32(dw), int(bool(accumulate)), ws.data_ptr(), ws.numel(),
                                         _u32(amax_dz, "amax_dz"), stream()), "fp_conv_stem_wgrad_hp")
    return dw


def conv_igemm_hp_supported(desc):
    return bool(_cached_query("fp_conv_igemm_hp_supported", desc))


def conv_igemm_hp(desc, src, wpacked_hp, y, amax_src, amax_w, bias=None, addend=None, addend_mask=None, actsrc=None, bn_out=None):
    """flattened implicit GEMM with fp16-pair operands (3x3 stride 2, 1x1, their data gradients): weights from FP_PACK_{FWD,DGRAD}_HP"""
    lib = _lib.load()
    epi = desc.epi
    if bias is not None:
        epi |= _lib.EPI_BIAS
    if addend is not None:
        epi |= _lib.EPI_ADDEND
    if addend_mask is not None:
        epi |= _lib.EPI_ADDEND_MASK
    d = ConvDesc.from_buffer_copy(desc)
    d.epi = epi
    need = 0 if _NO_SPLITK else _cached_query("fp_conv_igemm_workspace", d)
    ws_ptr, ws_n = 0, 0
    if need > 0:
        ws = workspace(need, y.device, "igemm")
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    _lib.check(lib.fp_conv_igemm_hp(C.byref(d), _f32(src, "src"), _chk(wpacked_hp, "wpacked_hp"), _f32(bias), _f32(addend), _f32(addend_mask),
                                    _f32(actsrc), _f32(y, "y"), ws_ptr, ws_n, _u32(amax_src), _u32(amax_w), _aux(bn_out=bn_out), stream()),
               "fp_conv_igemm_hp")
    return y


def conv3x3_bf3_supported(desc):
    return bool(_cached_query("fp_conv3x3_bf3_supported", desc))


_bf16x2 = False      # opt-in inference mode of the bf16 tile kernel (Engine.forward sets it around an eval forward): two bf16 terms per operand


def conv3x3_bf3(desc, src, wpacked_bf3, y, bias=None, addend=None, addend_mask=None, actsrc=None, src1=None, bn_out=None):
    """3x3 stride-1 conv / data-gradient with exactly split bf16x3 operands (same semantics as conv_igemm; src1 = the skip tensor
    of the GATHER_FWD_REFLECT_UP2 concat)"""
    epi = desc.epi | (_lib.EPI_BIAS if bias is not None else 0) | (_lib.EPI_ADDEND if addend is not None else 0) | \
        (_lib.EPI_ADDEND_MASK if addend_mask is not None else 0)
    if _bf16x2 and desc.gather in (_lib.GATHER_FWD_ZERO, _lib.GATHER_FWD_REFLECT, _lib.GATHER_FWD_REFLECT_UP2):
        epi |= _lib.EPI_BF16X2
    lib = _lib.load()
    d = ConvDesc.from_buffer_copy(desc)
    d.epi = epi
    need = _cached_query("fp_conv3x3_bf3_workspace", d)
    ws_ptr, ws_n = 0, 0
    if need > 0:
        ws = workspace(need, y.device, "igemm")
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    _lib.check(lib.fp_conv3x3_bf3(C.byref(d), _f32(src, "src"), _f32(src1, "src1"), _f32(wpacked_bf3, "wpacked"), _f32(bias), _f32(addend),
                                  _f32(addend_mask), _f32(actsrc), _f32(y, "y"), ws_ptr, ws_n, _aux(bn_out=bn_out), stream()), "fp_conv3x3_bf3")
    return y


# ---- fp16-pair operands (include/footprints_hip.h "hp"): amax slots are int32 tensors of amax_elems() elements ---------------------
_amax_elems = None


def amax_elems():
    """int32 elements of storage per amax slot (FP_AMAX_ELEMS of the loaded library)"""
    global _amax_elems
    if _amax_elems is None:
        _amax_elems = int(_lib.load().fp_amax_slot_elems())
    return _amax_elems


def _u32(t, what="slot"):
    if t is None:
        return 0
    if t.dtype != torch.int32 or not t.is_cuda or t.numel() < amax_elems():
        raise RuntimeError("footprints_amd: %s must be a CUDA int32 tensor of >= %d elements" % (what, amax_elems()))
    return t.data_ptr()


class BnOut:
    """BatchNorm partials out of a convolution's epilogue (fp_aux.bn_* of include/footprints_hip.h): pass as `bn_out=` to the convolution
    call; afterwards `.nblk` = number of partial blocks the launch wrote (0 = it could not emit: run the BatchNorm's own reduction pass).
    Forward statistics (count, mean, M2) with z2d None; the backward sums (sum g, sum g * xhat) of the BatchNorm whose input z2d and saved
    statistics are given otherwise.  Round 6: an explicit argument of the launch -- rounds 3-5 armed a per-thread sink in the library."""

    def __init__(self, part, z2d=None, save_mean=None, save_invstd=None):
        self.part, self.z, self.mean, self.invstd = part, z2d, save_mean, save_invstd
        self._n = C.c_int32(0)

    @property
    def nblk(self):
        return int(self._n.value)


def _aux(amax_out=None, bn_out=None):
    """the `const fp_aux*` argument of one launch: None when there is nothing to ask for (the struct is read during the call only)"""
    if amax_out is None and bn_out is None:
        return None
    a = _lib.Aux()
    if amax_out is not None:
        a.amax_out = _u32(amax_out, "amax_out")
    if bn_out is not None:
        bn_out._n.value = 0
        a.bn_part, a.bn_capacity_floats, a.bn_nblk_out = _f32(bn_out.part, "part"), bn_out.part.numel(), C.pointer(bn_out._n)
        if bn_out.z is not None:
            a.bnb_z, a.bnb_mean, a.bnb_invstd = _f32(bn_out.z, "z"), _f32(bn_out.mean, "save_mean"), _f32(bn_out.invstd, "save_invstd")
    return C.byref(a)


def zero_u32(t):
    _lib.check(_lib.load().fp_zero_u32(t.data_ptr(), t.numel(), stream()), "fp_zero_u32")
    return t


def amax_f32(x, slot):
    """slot (zeroed by the caller) <- max |x| (bit pattern, spread over the sub-slots)"""
    _lib.check(_lib.load().fp_amax_f32(_f32(x, "x"), x.numel(), _u32(slot), stream()), "fp_amax_f32")
    return slot


def amax_value(slot):
    """host-side read of a slot (tests / debugging): the float it encodes"""
    return float(slot[:amax_elems()].max().view(torch.int32).cpu().view(torch.float32))


def weight_amax(w, slot):
    """slot <- max |w| (zeroes the slot first): the scale of every fp16-pair packing of this weight tensor"""
    _lib.check(_lib.load().fp_weight_amax(_f32(w), w.numel(), _u32(slot), stream()), "fp_weight_amax")
    return slot


def new_slot(device="cuda"):
    return torch.zeros(amax_elems(), dtype=torch.int32, device=device)


def packed_weight_elems_hp(Cout, Cin, K, for_dgrad=False):
    return int(_lib.load().fp_packed_weight_elems_hp(Cout, Cin, K, K, int(for_dgrad)))


def pack_conv_weight_hp(w, wp, slot, for_dgrad=False, amax_ready=False):
    Cout, Cin, KH, KW = w.shape
    _lib.check(_lib.load().fp_pack_conv_weight_hp(_f32(w), _f32(wp), Cout, Cin, KH, KW, int(for_dgrad), _u32(slot), int(amax_ready), stream()),
               "fp_pack_conv_weight_hp")
    return wp


def conv3x3_hp(desc, src, wpacked_hp, y, amax_src, amax_w, amax_out=None, bias=None, addend=None, addend_mask=None, actsrc=None, src1=None,
               amax_src1=None, bn_out=None):
    """conv3x3_bf3 with fp16-pair operands; amax_* are slots (see amax_f32); amax_out (zeroed by the caller) receives max |y|"""
    epi = desc.epi | (_lib.EPI_BIAS if bias is not None else 0) | (_lib.EPI_ADDEND if addend is not None else 0) | \
        (_lib.EPI_ADDEND_MASK if addend_mask is not None else 0)
    lib = _lib.load()
    d = ConvDesc.from_buffer_copy(desc)
    d.epi = epi
    need = _cached_query("fp_conv3x3_bf3_workspace", d)
    ws_ptr, ws_n = 0, 0
    if need > 0:
        ws = workspace(need, y.device, "igemm")
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    _lib.check(lib.fp_conv3x3_hp(C.byref(d), _f32(src, "src"), _f32(src1, "src1"), _f32(wpacked_hp, "wpacked"), _f32(bias), _f32(addend),
                                 _f32(addend_mask), _f32(actsrc), _f32(y, "y"), ws_ptr, ws_n, _u32(amax_src, "amax_src"),
                                 _u32(amax_src1, "amax_src1"), _u32(amax_w, "amax_w"), _u32(amax_out, "amax_out"), _aux(bn_out=bn_out), stream()),
               "fp_conv3x3_hp")
    return y


def conv_igemm_bf3(desc, src, wpacked_bf3, y, bias=None, addend=None, addend_mask=None, actsrc=None, bn_out=None):
    """flattened implicit GEMM with EXACTLY split bf16x3 operands (3x3 stride 2, 1x1, their data gradients; the default operand format):
    weights from FP_PACK_{FWD,DGRAD}_BF3; shapes as conv_igemm_hp_supported"""
    lib = _lib.load()
    epi = desc.epi
    if bias is not None:
        epi |= _lib.EPI_BIAS
    if addend is not None:
        epi |= _lib.EPI_ADDEND
    if addend_mask is not None:
        epi |= _lib.EPI_ADDEND_MASK
    d = ConvDesc.from_buffer_copy(desc)
    d.epi = epi
    need = 0 if _NO_SPLITK else _cached_query("fp_conv_igemm_workspace", d)
    ws_ptr, ws_n = 0, 0
    if need > 0:
        ws = workspace(need, y.device, "igemm")
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    _lib.check(lib.fp_conv_igemm_bf3(C.byref(d), _f32(src, "src"), _chk(wpacked_bf3, "wpacked_bf3"), _f32(bias), _f32(addend), _f32(addend_mask),
                                     _f32(actsrc), _f32(y, "y"), ws_ptr, ws_n, _aux(bn_out=bn_out), stream()), "fp_conv_igemm_bf3")
    return y


def packed_weight_elems_bf3(Cout, Cin, K, for_dgrad=False):
    return int(_lib.load().fp_packed_weight_elems_bf3(Cout, Cin, K, K, int(for_dgrad)))


def pack_conv_weight_bf3(w, wp, for_dgrad=False):
    Cout, Cin, KH, KW = w.shape
    _lib.check(_lib.load().fp_pack_conv_weight_bf3(_f32(w), _f32(wp), Cout, Cin, KH, KW, int(for_dgrad), stream()), "fp_pack_conv_weight_bf3")
    return wp


def conv_wgrad(desc, src0, src1, dz, dw, accumulate=False):
    lib = _lib.load()
    need = _cached_query("fp_conv_wgrad_workspace", desc)
    ws = workspace(need, dz.device)
    _lib.check(lib.fp_conv_wgrad(C.byref(desc), _f32(src0), _f32(src1), _f32(dz), _f32(dw), int(bool(accumulate)),
                                 ws.data_ptr(), ws.numel(), stream()), "fp_conv_wgrad")
    return dw


def conv_wgrad_bf3_supported(desc):
    return _cached_query("fp_conv_wgrad_bf3_workspace", desc) >= 0


def conv_wgrad_bf3(desc, x, dz, dw, k_begin=0, accumulate=False, db=None, amax=None):
    """3x3 stride-1 weight gradient with split operands, into dw[:, k_begin:k_begin + C0]; db (optional) receives the bias gradient
    (column sums of dz) from the same pass.  amax = (slot of x, slot of dz): scaled fp16 pairs (three products); None: the exact
    bf16x3 split (six)"""
    lib = _lib.load()
    need = _cached_query("fp_conv_wgrad_bf3_workspace", desc)
    if need < 0:
        raise RuntimeError("fp_conv_wgrad_bf3: shape not supported")
    ws = workspace(need, dz.device)
    if amax is not None:
        return conv_wgrad_hp(desc, x, dz, dw, amax[0], amax[1], k_begin, accumulate, db, ws)
    _lib.check(lib.fp_conv_wgrad_bf3(C.byref(desc), _f32(x), _f32(dz), _f32(dw), _f32(db), dw.shape[1], k_begin, int(bool(accumulate)),
                                     ws.data_ptr(), ws.numel(), stream()), "fp_conv_wgrad_bf3")
    return dw


def conv_wgrad_slice(desc, src0, src1, dz, dw, k_begin, accumulate=False):
    """weight gradient of the input-channel slice [k_begin, k_begin + C0 + C1) of the wider gradient dw [Nout][Cin][K][K]"""
    lib = _lib.load()
    ws = workspace(lib.fp_conv_wgrad_workspace(C.byref(desc)), dz.device)
    _lib.check(lib.fp_conv_wgrad_slice(C.byref(desc), _f32(src0), _f32(src1), _f32(dz), _f32(dw), dw.shape[1], k_begin,
                                       int(bool(accumulate)), ws.data_ptr(), ws.numel(), stream()), "fp_conv_wgrad_slice")
    return dw


def up2_phase_wgrad_supported(N, h, w, C0, Nout):
    return _lib.load().fp_conv_up2_phase_wgrad_workspace(N, h, w, C0, Nout) >= 0


def conv_wgrad_hp(desc, x, dz, dw, amax_x, amax_dz, k_begin=0, accumulate=False, db=None, ws=None):
    lib = _lib.load()
    if ws is None:
        ws = workspace(_cached_query("fp_conv_wgrad_bf3_workspace", desc), dz.device)
    _lib.check(lib.fp_conv_wgrad_hp(C.byref(desc), _f32(x), _f32(dz), _f32(dw), _f32(db), dw.shape[1], k_begin, int(bool(accumulate)),
                                    ws.data_ptr(), ws.numel(), _u32(amax_x, "amax_x"), _u32(amax_dz, "amax_dz"), stream()), "fp_conv_wgrad_hp")
    return dw


def conv_up2_phase_wgrad_hp(low, dz, dw, amax_low, amax_dz, k_begin=0, accumulate=False, db=None):
    lib = _lib.load()
    N, h, w, C0 = low.shape
    Nout = dz.shape[3]
    need = lib.fp_conv_up2_phase_wgrad_workspace(N, h, w, C0, Nout)
    if need < 0:
        raise RuntimeError("fp_conv_up2_phase_wgrad_hp: shape not supported")
    ws = workspace(need, dz.device)
    _lib.check(lib.fp_conv_up2_phase_wgrad_hp(_f32(low), _f32(dz), _f32(dw), _f32(db), N, h, w, C0, Nout, dw.shape[1], k_begin,
                                              int(bool(accumulate)), ws.data_ptr(), ws.numel(), _u32(amax_low, "amax_low"),
                                              _u32(amax_dz, "amax_dz"), stream()), "fp_conv_up2_phase_wgrad_hp")
    return dw


def conv_up2_phase_wgrad(low, dz, dw, k_begin=0, accumulate=False, bf3=False, db=None):
    """db (bf3 only): also produce the bias gradient (column sums of dz) from the same pass"""
    lib = _lib.load()
    if db is not None and not bf3:
        raise RuntimeError("conv_up2_phase_wgrad: the fused bias gradient needs the bf16x3 kernel")
    N, h, w, C0 = low.shape
    Nout = dz.shape[3]
    need = lib.fp_conv_up2_phase_wgrad_workspace(N, h, w, C0, Nout)
    if need < 0:
        raise RuntimeError("fp_conv_up2_phase_wgrad: shape not supported")
    ws = workspace(need, dz.device)
    tail = (N, h, w, C0, Nout, dw.shape[1], k_begin, int(bool(accumulate)), ws.data_ptr(), ws.numel(), stream())
    if bf3:
        _lib.check(lib.fp_conv_up2_phase_wgrad_bf3(_f32(low), _f32(dz), _f32(dw), _f32(db), *tail), "fp_conv_up2_phase_wgrad_bf3")
    else:
        _lib.check(lib.fp_conv_up2_phase_wgrad(_f32(low), _f32(dz), _f32(dw), *tail), "fp_conv_up2_phase_wgrad")
    return dw


def packed_weight_elems(Cout, Cin, K, for_dgrad=False, stem=False):
    return int(_lib.load().fp_packed_weight_elems(Cout, Cin, K, K, int(for_dgrad), int(stem)))


def pack_conv_weight(w, wp, stem=False):
    Cout, Cin, KH, KW = w.shape
    _lib.check(_lib.load().fp_pack_conv_weight(_f32(w), _f32(wp), Cout, Cin, KH, KW, int(stem), stream()), "fp_pack_conv_weight")
    return wp


def pack_conv_weight_dgrad(w, wp):
    Cout, Cin, KH, KW = w.shape
    _lib.check(_lib.load().fp_pack_conv_weight_dgrad(_f32(w), _f32(wp), Cout, Cin, KH, KW, stream()), "fp_pack_conv_weight_dgrad")
    return wp


def up2_packed_weight_elems(Cout, c_count):
    return _lib.load().fp_up2_packed_weight_elems(Cout, c_count)


def pack_up2_weight(w, wp, c_begin, c_count):
    Cout, Cin = w.shape[:2]
    _lib.check(_lib.load().fp_pack_up2_weight(_f32(w), _f32(wp), Cout, Cin, c_begin, c_count, stream()), "fp_pack_up2_weight")
    return wp


def pack_conv_weight_slice(w, wp, c_begin, c_count):
    Cout, Cin = w.shape[:2]
    _lib.check(_lib.load().fp_pack_conv_weight_slice(_f32(w), _f32(wp), Cout, Cin, c_begin, c_count, stream()), "fp_pack_conv_weight_slice")
    return wp


def pack_up2_weight_bf3(w, wp, c_begin, c_count):
    Cout, Cin = w.shape[:2]
    _lib.check(_lib.load().fp_pack_up2_weight_bf3(_f32(w), _f32(wp), Cout, Cin, c_begin, c_count, stream()), "fp_pack_up2_weight_bf3")
    return wp


def conv_up2_phase_fwd_hp(low, wphase_hp, bias, y, amax_low, amax_w, amax_out=None, act=0, addend=None):
    N, h, w, C0 = low.shape
    _lib.check(_lib.load().fp_conv_up2_phase_fwd_hp(_f32(low), _f32(wphase_hp), _f32(bias), _f32(addend), _f32(y), N, h, w, C0, y.shape[3],
                                                    int(act), _u32(amax_low, "amax_low"), _u32(amax_w, "amax_w"), _u32(amax_out, "amax_out"),
                                                    stream()), "fp_conv_up2_phase_fwd_hp")
    return y


def conv_up2_phase_dgrad_hp(dz, wpacked_hp, ext, amax_dz, amax_w):
    N, H2, W2, Cout = dz.shape
    _lib.check(_lib.load().fp_conv_up2_phase_dgrad_hp(_f32(dz), _f32(wpacked_hp), _f32(ext), N, H2 // 2, W2 // 2, Cout, ext.shape[3],
                                                      _u32(amax_dz, "amax_dz"), _u32(amax_w, "amax_w"), stream()), "fp_conv_up2_phase_dgrad_hp")
    return ext


def conv_up2_phase_fwd_bf3(low, wphase_bf3, bias, y, act=0, addend=None, amax_out=None):
    N, h, w, C0 = low.shape
    _lib.check(_lib.load().fp_conv_up2_phase_fwd_bf3(_f32(low), _f32(wphase_bf3), _f32(bias), _f32(addend), _f32(y), N, h, w, C0, y.shape[3],
                                                     int(act), _aux(amax_out), stream()), "fp_conv_up2_phase_fwd_bf3")
    return y


def pack_up2_weight_dgrad_bf3(w, wp, c_begin, c_count):
    Cout, Cin = w.shape[:2]
    _lib.check(_lib.load().fp_pack_up2_weight_dgrad_bf3(_f32(w), _f32(wp), Cout, Cin, c_begin, c_count, stream()),
               "fp_pack_up2_weight_dgrad_bf3")
    return wp


def conv_up2_phase_dgrad_bf3(dz, wpacked_bf3, ext):
    """gradient wrt the low-res input of an upsample conv on the (h+2) x (w+2) extended grid (up2_fold_bwd folds it)"""
    N, H2, W2, Cout = dz.shape
    _lib.check(_lib.load().fp_conv_up2_phase_dgrad_bf3(_f32(dz), _f32(wpacked_bf3), _f32(ext), N, H2 // 2, W2 // 2, Cout, ext.shape[3],
                                                       stream()), "fp_conv_up2_phase_dgrad_bf3")
    return ext


def conv_up2_phase_fwd(low, wphase, bias, y, act=0, addend=None):
    N, h, w, C0 = low.shape
    _lib.check(_lib.load().fp_conv_up2_phase_fwd(_f32(low), _f32(wphase), _f32(bias), _f32(addend), _f32(y), N, h, w, C0, y.shape[3],
                                                 int(act), stream()), "fp_conv_up2_phase_fwd")
    return y


def pack_up2_weight_dgrad(w, wp, c_begin, c_count):
    Cout, Cin = w.shape[:2]
    _lib.check(_lib.load().fp_pack_up2_weight_dgrad(_f32(w), _f32(wp), Cout, Cin, c_begin, c_count, stream()), "fp_pack_up2_weight_dgrad")
    return wp


def pack_conv_weight_dgrad_slice(w, wp, c_begin, c_count):
    Cout, Cin = w.shape[:2]
    _lib.check(_lib.load().fp_pack_conv_weight_dgrad_slice(_f32(w), _f32(wp), Cout, Cin, c_begin, c_count, stream()),
               "fp_pack_conv_weight_dgrad_slice")
    return wp


def up2_fold_bwd(ext, dlow, addend=None, ylow=None, amax_out=None):
    N, h, w, Cn = dlow.shape
    _lib.check(_lib.load().fp_up2_fold_bwd(_f32(ext), N, h, w, Cn, _f32(addend), _f32(ylow), _f32(dlow), _aux(amax_out), stream()), "fp_up2_fold_bwd")
    return dlow


def build_pack_table(jobs, device):
    """jobs: list of (kind, w, wp, c_begin, c_count) over torch tensors that never move.  Returns the device-resident table
    (jobs, blk2job, nblocks) for pack_weights_batched."""
    lib = _lib.load()
    if not jobs:
        return None, None, 0
    arr = (_lib.PackJob * len(jobs))()
    blk2job = []
    for i, job in enumerate(jobs):
        kind, w, wp, c_begin, c_count = job[:5]
        Cout, Cin, KH, KW = w.shape
        nb = lib.fp_pack_job_blocks(kind, Cout, KH, KW, c_count)
        j = arr[i]
        j.w, j.wp = _f32(w), _f32(wp)
        j.Cout, j.Cin, j.KH, j.KW, j.kind, j.c_begin, j.c_count = Cout, Cin, KH, KW, kind, c_begin, c_count
        j.block_begin, j.block_count = len(blk2job), nb
        j.amax = job[5].data_ptr() if len(job) > 5 else None        # fp16-pair kinds: the weight tensor's amax slot
        blk2job += [i] * nb
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    b2j = torch.tensor(blk2job, dtype=torch.int32, device=device)
    return raw, b2j, len(blk2job)


def pack_weights_amax(table):
    """max |w| of every fp16-pair job's weight tensor into its slot (zeroed by the caller); before pack_weights_batched"""
    raw, b2j, nblocks = table
    if nblocks == 0:
        return
    _lib.check(_lib.load().fp_pack_weights_amax(raw.data_ptr(), b2j.data_ptr(), nblocks, stream()), "fp_pack_weights_amax")


def pack_weights_batched(table, max_wgs=0):
    """max_wgs > 0: a persistent launch of at most that many workgroups (a repack on a side stream beside a latency-bound chain)"""
    raw, b2j, nblocks = table
    if nblocks == 0:
        return
    _lib.check(_lib.load().fp_pack_weights_batched_capped(raw.data_ptr(), b2j.data_ptr(), nblocks, int(max_wgs), stream()), "fp_pack_weights_batched")


def colsum(x2d, out, accumulate=False):
    lib = _lib.load()
    M, Cn = x2d.shape
    ws = workspace(lib.fp_colsum_workspace(M, Cn), x2d.device)
    _lib.check(lib.fp_colsum(_f32(x2d), M, Cn, _f32(out), int(bool(accumulate)), ws.data_ptr(), ws.numel(), stream()), "fp_colsum")
    return out


def up2cat_bwd(dxv, N, h, w, C0, C1, dlow, addend=None, ylow=None, dskip=None, accumulate_skip=False):
    _lib.check(_lib.load().fp_up2cat_bwd(_f32(dxv), N, h, w, C0, C1, _f32(addend), _f32(ylow), _f32(dlow), _f32(dskip),
                                         int(bool(accumulate_skip)), stream()), "fp_up2cat_bwd")
    return dlow


def head_fwd(x, w, b, low, sigmoid):
    N, h, wd, Cin = x.shape
    _lib.check(_lib.load().fp_head_fwd(_f32(x), _f32(w), _f32(b), _f32(low), N, h, wd, Cin, int(bool(sigmoid)), stream()), "fp_head_fwd")
    return low


def head_upsample(low, out_nchw, scale, c0):
    N, h, w, _ = low.shape
    _lib.check(_lib.load().fp_head_upsample(_f32(low), _f32(out_nchw), N, h, w, scale, out_nchw.shape[1], c0, stream()), "fp_head_upsample")
    return out_nchw


def head_upsample_bwd(dout_nchw, low, dzlow, scale, c0, sigmoid):
    N, h, w, _ = dzlow.shape
    _lib.check(_lib.load().fp_head_upsample_bwd(_f32(dout_nchw), _f32(low), _f32(dzlow), N, h, w, scale, dout_nchw.shape[1], c0,
                                                int(bool(sigmoid)), stream()), "fp_head_upsample_bwd")
    return dzlow


def head_dgrad(dzlow, w, dx, elu_src=None, amax_out=None):
    N, h, wd, Cin = dx.shape
    _lib.check(_lib.load().fp_head_dgrad(_f32(dzlow), _f32(w), _f32(elu_src), _f32(dx), N, h, wd, Cin, _aux(amax_out), stream()), "fp_head_dgrad")
    return dx


def head_wgrad(x, dzlow, dw, db, accumulate=False):
    lib = _lib.load()
    N, h, wd, Cin = x.shape
    ws = workspace(lib.fp_head_wgrad_workspace(N, h, wd, Cin), x.device)
    _lib.check(lib.fp_head_wgrad(_f32(x), _f32(dzlow), _f32(dw), _f32(db), N, h, wd, Cin, int(bool(accumulate)), ws.data_ptr(),
                                 ws.numel(), stream()), "fp_head_wgrad")


def bn_train_stats(z2d, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, scale, shift, eps=1e-5, momentum=0.1):
    lib = _lib.load()
    M, Cn = z2d.shape
    ws = workspace(lib.fp_bn_workspace(M, Cn), z2d.device)
    if nbt is not None and nbt.dtype != torch.int64:
        raise RuntimeError("num_batches_tracked must be int64")
    _lib.check(lib.fp_bn_train_stats(_f32(z2d), M, Cn, _f32(gamma), _f32(beta), eps, momentum, _f32(running_mean), _f32(running_var),
                                     _chk(nbt), _f32(save_mean), _f32(save_invstd), _f32(scale), _f32(shift), ws.data_ptr(),
                                     ws.numel(), stream()), "fp_bn_train_stats")


# statistics out of the producing tile convolution's epilogue (csrc/conv3x3_tile_bf3.hip, fp_aux.bn_part): FP_BN_EPI=0 switches it off
_BN_EPI = bool(int(os.environ.get("FP_BN_EPI", "1")))


def bn_stats_out(part):
    """forward statistics out of the next convolution's epilogue: conv(..., bn_out=bn_stats_out(part)), then `.nblk`"""
    return BnOut(part)


def bn_train_stats_partials(part, nblk, Cn, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, scale, shift, eps=1e-5,
                            momentum=0.1):
    if nbt is not None and nbt.dtype != torch.int64:
        raise RuntimeError("num_batches_tracked must be int64")
    _lib.check(_lib.load().fp_bn_train_stats_partials(_f32(part), int(nblk), int(Cn), _f32(gamma), _f32(beta), eps, momentum,
                                                      _f32(running_mean), _f32(running_var), _chk(nbt), _f32(save_mean), _f32(save_invstd),
                                                      _f32(scale), _f32(shift), stream()), "fp_bn_train_stats_partials")


# the BatchNorm-backward reduction out of the producing data gradient's epilogue (csrc/conv3x3_tile_bf3.hip, fp_aux.bnb_*): FP_BN_BWD_EPI=0
# keeps fp_bn_bwd's own reduction pass
_BN_BWD_EPI = bool(int(os.environ.get("FP_BN_BWD_EPI", "1")))


def bn_bwd_out(part, z2d, save_mean, save_invstd):
    """backward sums (sum g, sum g * xhat) out of a tile data gradient's epilogue: conv(..., bn_out=bn_bwd_out(...)), then `.nblk`"""
    return BnOut(part, z2d, save_mean, save_invstd)


def bn_bwd_partials(g2d, z2d, save_mean, save_invstd, gamma, dz2d, dgamma, dbeta, part, nblk, accumulate=False, amax_out=None):
    """fp_bn_bwd without its reduction pass: g2d is already masked, `part` holds (sum g, sum g * xhat) per pixel tile and channel"""
    lib = _lib.load()
    M, Cn = z2d.shape
    coef = workspace(lib.fp_bn_workspace(M, Cn), z2d.device)
    _lib.check(lib.fp_bn_bwd_partials(_f32(g2d), _f32(z2d), _f32(save_mean), _f32(save_invstd), _f32(gamma), _f32(dz2d), _f32(dgamma),
                                      _f32(dbeta), int(bool(accumulate)), M, Cn, _f32(part), int(nblk), coef.data_ptr(), _aux(amax_out), stream()),
               "fp_bn_bwd_partials")
    return dz2d


def bn_eval_coeffs(gamma, beta, rm, rv, scale, shift, eps=1e-5):
    _lib.check(_lib.load().fp_bn_eval_coeffs(_f32(gamma), _f32(beta), _f32(rm), _f32(rv), eps, gamma.numel(), _f32(scale), _f32(shift),
                                             stream()), "fp_bn_eval_coeffs")


def scale_rows(w, scale, out):
    rows = w.shape[0]
    _lib.check(_lib.load().fp_scale_rows(_f32(w), _f32(scale), _f32(out), rows, w.numel() // rows, stream()), "fp_scale_rows")
    return out


def bn_apply(z2d, scale, shift, y2d, residual=None, relu=True, amax_out=None):
    M, Cn = z2d.shape
    _lib.check(_lib.load().fp_bn_apply(_f32(z2d), _f32(scale), _f32(shift), _f32(residual), _f32(y2d), M, Cn, int(bool(relu)), _aux(amax_out), stream()),
               "fp_bn_apply")
    return y2d


def bn_bwd(dy2d, relu_out, z2d, save_mean, save_invstd, gamma, dz2d, dgamma, dbeta, g_out=None, accumulate=False, amax_out=None):
    """amax_out: slot receiving max |dz| (the apply stage publishes it)"""
    lib = _lib.load()
    M, Cn = z2d.shape
    ws = workspace(lib.fp_bn_workspace(M, Cn), z2d.device)
    _lib.check(lib.fp_bn_bwd(_f32(dy2d), _f32(relu_out), _f32(z2d), _f32(save_mean), _f32(save_invstd), _f32(gamma), _f32(dz2d),
                             _f32(g_out), _f32(dgamma), _f32(dbeta), int(bool(accumulate)), M, Cn, ws.data_ptr(), ws.numel(), _aux(amax_out), stream()),
               "fp_bn_bwd")
    return dz2d


def maxpool_fwd(x, y, argmax, amax_out=None):
    N, H, W, Cn = x.shape
    _lib.check(_lib.load().fp_maxpool_fwd(_f32(x), _f32(y), _chk(argmax), N, H, W, Cn, _aux(amax_out), stream()), "fp_maxpool_fwd")
    return y


def maxpool_bwd(dy, argmax, dx, accumulate=False):
    N, H, W, Cn = dx.shape
    _lib.check(_lib.load().fp_maxpool_bwd(_f32(dy), _chk(argmax), _f32(dx), N, H, W, Cn, int(bool(accumulate)), stream()), "fp_maxpool_bwd")
    return dx


def loss_fwd_bwd(preds, targets, losses_out, dpreds=None, depth_range=(0.1, 100.0), prior=0.25):
    """preds: 4 tensors [B,4,H,W] in order '1/8','1/4','1/2','1/1'; targets: dict with the reference batch keys."""
    lib = _lib.load()
    B, _, H, W = preds[0].shape
    ws = workspace(lib.fp_loss_workspace(B, H, W), preds[0].device)
    P = (C.c_void_p * 4)(*[_f32(p, "pred") for p in preds])
    D = (C.c_void_p * 4)(*[_f32(p, "dpred") for p in dpreds]) if dpreds is not None else None
    _lib.check(lib.fp_loss_fwd_bwd(P, _f32(targets["visible_ground"]), _f32(targets["all_ground"]), _f32(targets["depth"]),
                                   _f32(targets["ground_depth"]), _f32(targets["moving_object_mask"]), _f32(targets["depth_mask"]),
                                   float(depth_range[0]), float(depth_range[1]), float(prior), D, _f32(losses_out), B, H, W,
                                   ws.data_ptr(), ws.numel(), stream()), "fp_loss_fwd_bwd")
    return losses_out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _lib.check(_lib.load().fp_adam_step(_f32(param), _f32(grad), _f32(exp_avg), _f32(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
                                        int(step), grad_scale, stream()), "fp_adam_step")


def adam_hyper(lr, beta1, beta2, eps, step, grad_scale=1.0):
    """the seven float scalars of one Adam step (host tensor), derived exactly like fp_adam_step does"""
    h = torch.empty(7, dtype=torch.float32)
    _lib.check(_lib.load().fp_adam_hyper(lr, beta1, beta2, eps, int(step), grad_scale, h.data_ptr()), "fp_adam_hyper")
    return h


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, hyper_dev):
    _lib.check(_lib.load().fp_adam_step_dev(_f32(param), _f32(grad), _f32(exp_avg), _f32(exp_avg_sq), param.numel(), _f32(hyper_dev),
                                            stream()), "fp_adam_step_dev")


def pack_pred_fp16(pred, out=None):
    """[B,4,H,W] fp32 network output -> float16 with sigmoid on the mask channels (the test-set inference file format)"""
    B, Cn, H, W = pred.shape
    if Cn != 4:
        raise RuntimeError("pack_pred_fp16 expects the 4-channel network output")
    if out is None:
        out = torch.empty((B, 4, H, W), dtype=torch.float16, device=pred.device)
    _lib.check(_lib.load().fp_pack_pred_fp16(_f32(pred), _chk(out), B, H, W, stream()), "fp_pack_pred_fp16")
    return out


def eval_mask_counts(pred, gt, region=None, invert=False):
    """pred [B,H,W] float16/float32 (may be a channel slice of [B,4,H,W]), gt [B,H,W] float32, region [B,H,W] uint8 or None
    -> int64 [B,4] = n_true, tp, fp, fn (evaluate_model.py:72-99 thresholds)"""
    B = pred.shape[0]
    pixels = pred.shape[1] * pred.shape[2]
    if pred.dtype not in (torch.float16, torch.float32) or pred.stride(2) != 1 or pred.stride(1) != pred.shape[2]:
        raise RuntimeError("eval_mask_counts: pred must be float16/float32 with contiguous images")
    if region is not None and (region.dtype != torch.uint8 or tuple(region.shape) != tuple(gt.shape)):
        raise RuntimeError("eval_mask_counts: region must be uint8 with the ground truth's shape")
    if not pred.is_cuda:
        raise RuntimeError("eval_mask_counts: pred must be a CUDA tensor (no CPU path in the product)")
    counts = torch.empty((B, 4), dtype=torch.int64, device=pred.device)
    _lib.check(_lib.load().fp_eval_mask_counts(pred.data_ptr(), int(pred.dtype == torch.float16), _f32(gt), _chk(region) if region is not None else None,
                                               int(bool(invert)), B, pixels, pred.stride(0), _chk(counts), stream()), "fp_eval_mask_counts")
    return counts


def eval_depth_sums(disp, gt, min_depth=0.1, max_depth=100.0, clip=(0.5, 20.0)):
    """disp [B,H,W] float16/float32 sigmoid output, gt [B,H,W] float32 -> float64 [B,5] = n, n(a1), sum sq, sum abs_rel, sum sq_rel"""
    B = disp.shape[0]
    pixels = disp.shape[1] * disp.shape[2]
    if disp.dtype not in (torch.float16, torch.float32) or disp.stride(2) != 1 or disp.stride(1) != disp.shape[2]:
        raise RuntimeError("eval_depth_sums: disp must be float16/float32 with contiguous images")
    if not disp.is_cuda:
        raise RuntimeError("eval_depth_sums: disp must be a CUDA tensor (no CPU path in the product)")
    sums = torch.empty((B, 5), dtype=torch.float64, device=disp.device)
    _lib.check(_lib.load().fp_eval_depth_sums(disp.data_ptr(), int(disp.dtype == torch.float16), _f32(gt), B, pixels, disp.stride(0), float(min_depth),
                                              float(max_depth), float(clip[0]), float(clip[1]), _chk(sums), stream()), "fp_eval_depth_sums")
    return sums


def nchw_to_nhwc(x, y=None):
    N, Cn, H, W = x.shape
    if y is None:
        y = torch.empty((N, H, W, Cn), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().fp_nchw_to_nhwc(_f32(x), _f32(y), N, Cn, H, W, stream()), "fp_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, y=None):
    N, H, W, Cn = x.shape
    if y is None:
        y = torch.empty((N, Cn, H, W), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().fp_nhwc_to_nchw(_f32(x), _f32(y), N, Cn, H, W, stream()), "fp_nhwc_to_nchw")
    return y


def fill(x, value):
    _lib.check(_lib.load().fp_fill(_f32(x), x.numel(), float(value), stream()), "fp_fill")
    return x


# ---- pyramid pooling (segmentation network, footprints/preprocessing/segmentation/network.py:174-207) -------------------------
def adaptive_avgpool_fwd(x, y):
    N, H, W, Cn = x.shape
    P = y.shape[1]
    _lib.check(_lib.load().fp_adaptive_avgpool_fwd(_f32(x), _f32(y), N, H, W, Cn, P, stream()), "fp_adaptive_avgpool_fwd")
    return y


def adaptive_avgpool_bwd(dy, dx, accumulate=False):
    N, H, W, Cn = dx.shape
    P = dy.shape[1]
    _lib.check(_lib.load().fp_adaptive_avgpool_bwd(_f32(dy), _f32(dx), N, H, W, Cn, P, int(bool(accumulate)), stream()), "fp_adaptive_avgpool_bwd")
    return dx


def bilinear_ac_fwd(src, dst, c_off):
    """src [N,P,P,C] -> dst[N,H,W,dstC][..., c_off:c_off+C] (bilinear, align_corners=True)"""
    N, P, _, Cn = src.shape
    _, H, W, dstC = dst.shape
    _lib.check(_lib.load().fp_bilinear_ac_fwd(_f32(src), _f32(dst), N, P, Cn, H, W, dstC, c_off, stream()), "fp_bilinear_ac_fwd")
    return dst


def bilinear_ac_bwd(ddst, dsrc, c_off):
    N, P, _, Cn = dsrc.shape
    _, H, W, dstC = ddst.shape
    _lib.check(_lib.load().fp_bilinear_ac_bwd(_f32(ddst), _f32(dsrc), N, P, Cn, H, W, dstC, c_off, stream()), "fp_bilinear_ac_bwd")
    return dsrc


def copy_channels(src, dst, C_, src_off=0, dst_off=0, accumulate=False):
    """dst[..., dst_off:dst_off+C] (+)= src[..., src_off:src_off+C] for NHWC tensors with equal pixel counts"""
    srcC, dstC = src.shape[-1], dst.shape[-1]
    M = src.numel() // srcC
    if dst.numel() // dstC != M:
        raise RuntimeError("copy_channels: pixel counts differ")
    _lib.check(_lib.load().fp_copy_channels(_f32(src), _f32(dst), M, C_, srcC, src_off, dstC, dst_off, int(bool(accumulate)), stream()),
               "fp_copy_channels")
    return dst


def seg_loss_fwd_bwd(preds, ground_mask, loss_mask, losses_out, dpreds=None):
    """segmentation trainer's loss (+ gradient): preds = four logit maps [B,1,h,w] (any batch stride: channel slices of wider buffers
    are fine as long as a map's rows are dense), masks [B,H,W]; losses_out: 5 B + 1 floats (per-scale / per-image means, their average,
    batch mean); dpreds: four tensors addressed like preds, or None"""
    lib = _lib.load()
    B, H, W = ground_mask.shape
    for t in list(preds) + (list(dpreds) if dpreds is not None else []):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] == 1 and t.stride(3) == 1 and t.stride(2) == t.shape[3]):
            raise RuntimeError("footprints_amd.ops.seg_loss_fwd_bwd: maps must be float32 CUDA [B,1,h,w] with dense rows")
    ws = workspace(lib.fp_seg_loss_workspace(B, H, W), ground_mask.device)
    P = (C.c_void_p * 4)(*[p.data_ptr() for p in preds])
    D = (C.c_void_p * 4)(*[p.data_ptr() for p in dpreds]) if dpreds is not None else None
    hs = (C.c_int32 * 4)(*[p.shape[2] for p in preds])
    wss = (C.c_int32 * 4)(*[p.shape[3] for p in preds])
    bs = (C.c_int64 * 4)(*[p.stride(0) if B > 1 else p.shape[2] * p.shape[3] for p in preds])      # a batch of one has no batch stride
    if dpreds is not None and B > 1 and any(d.stride(0) != p.stride(0) for d, p in zip(dpreds, preds)):
        raise RuntimeError("footprints_amd.ops.seg_loss_fwd_bwd: gradient maps must be addressed like the predictions")
    _lib.check(lib.fp_seg_loss_fwd_bwd(P, D, hs, wss, bs, _f32(ground_mask), _f32(loss_mask), B, H, W, _f32(losses_out), ws.data_ptr(), ws.numel(),
                                       stream()), "fp_seg_loss_fwd_bwd")
    return losses_out
