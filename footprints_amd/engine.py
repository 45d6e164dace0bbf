"""HIP execution engine for FootprintNetwork: explicit forward / backward schedules over the C ABI.

Design (MI355X-first, not an autograd graph):
  * activations NHWC fp32, resident in a named arena that is allocated once and reused every step (static
    addresses => the whole step is capturable in a hipGraph: TrainStep(graph=True));
  * 3x3 stride-1 convolutions (forward, data- and weight-gradient) multiply split operands: by default the exact three-term bf16 split
    (x = h + m + l, six MFMA products, fp32 accumulation: every operand bit of fp32 is kept) with the rest on the fp32 MFMA kernels; with
    FP_OPERANDS=fp16_pair (opt-in, _format.py) scaled fp16 pairs (22 significant bits after a per-tensor power-of-two scaling, three fp16
    MFMA products; the scale comes from amax slots that the producing kernels publish, AmaxBook below) -- one kernel template for both
    (conv3x3_tile_bf3.hip);
  * nearest-x2 upsample, skip concat, reflection / zero padding, ELU / ReLU, residual adds and their
    gradients never exist as tensors -- they are loader / epilogue modes of the implicit-GEMM kernels;
  * all live parameters are views of ONE flat fp32 buffer (same for gradients) in forward order, so Adam is
    one kernel launch and the data-parallel all-reduce works on contiguous buckets;
  * backward is a hand-written schedule (decoders, then encoder, reverse order); gradients of tensors with
    several consumers are accumulated in a fixed order by epilogue flags (deterministic; the only atomics are the integer-max
    publications of the amax slots, which are order-independent).

Reference call stack being replaced: FootprintNetwork.forward (network.py:21-30) and autograd's backward of
it (training/train.py:155).
"""
import os

import torch

from . import _lib as L
from . import ops
from ._format import operand_format

# Concurrency: the two decoders are independent, and every weight gradient is off the critical dgrad chain.
# Running them on side streams lets workgroups of 2-3 kernels share the CUs, which fills the occupancy ramp /
# tail of each launch (a conv launch is only ~2-3 "rounds" of workgroups per CU).  FP_SERIAL=1 disables it.
_CONCURRENT = not bool(int(os.environ.get("FP_SERIAL", "0")))
# inference: fold the encoder's eval-mode BatchNorm into the conv weights; FP_NO_FOLD=1 keeps conv + BN launches
_FOLD = not bool(int(os.environ.get("FP_NO_FOLD", "0")))
# exactly split bf16x3 operands for the 3x3 stride-1 tile kernel (conv3x3_tile_bf3.hip); FP_NO_BF3=1 keeps the fp32 MFMA
_BF3 = not bool(int(os.environ.get("FP_NO_BF3", "0")))
_WBF3 = _BF3 and not bool(int(os.environ.get("FP_NO_WBF3", "0")))      # ... and for the weight-gradient kernel
_PWBF3 = not bool(int(os.environ.get("FP_NO_PHASE_WBF3", "0")))           # ... and for the phase weight-gradient kernel (A/B switch)
# fp16-pair operands (two fp16 terms after a per-tensor power-of-two scaling, three MFMA products, 22 significant bits) for the same
# kernels, with the scaling taken from amax slots that producers publish / a reduction fills (csrc/fp_common.h).  OPT-IN since round 5
# (FP_OPERANDS=fp16_pair; footprints_amd/_format.py): the default is the exact bf16x3 split, which keeps every operand bit of fp32
_HP = _BF3 and operand_format() == "fp16_pair"
# per kernel family (A/B and precision studies, profiles/round3_notes.md): FP_HP_WGRAD=0 keeps the weight-gradient kernels on the exact
# bf16 split while forward / data-gradient use fp16 pairs; FP_HP_TILE=0 the other way round
_HP_WGRAD = _HP and bool(int(os.environ.get("FP_HP_WGRAD", "1")))
_HP_TILE = _HP and bool(int(os.environ.get("FP_HP_TILE", "1")))
# fp16 pairs also for the stride-2 3x3 / 1x1 convolutions and their data gradients (fp_conv_igemm_hp, round 3); FP_HP_IGEMM=0: fp32 MFMA
_HP_IGEMM = _HP and bool(int(os.environ.get("FP_HP_IGEMM", "1")))
# exact operand format (round 5): the same convolutions with exactly split bf16x3 operands (fp_conv_igemm_bf3); FP_BF3_IGEMM=0: fp32 MFMA
_BF3_IGEMM = _BF3 and not _HP and bool(int(os.environ.get("FP_BF3_IGEMM", "1")))
# the 1x1 downsample branch of a BasicBlock (conv -> BN, and its gradients) on the aux stream beside the block's main branch: the encoder
# is the serial spine of the step (one kernel on the GPU at a time), the aux stream idles until the decoders start.  FP_DS_AUX=0: in line
_DS_AUX = bool(int(os.environ.get("FP_DS_AUX", "1")))
# nearest-x2 phase decomposition of the upsample convs (conv_up2_phase.hip); FP_NO_PHASE=1 keeps the fused-gather path.
_PHASE = not bool(int(os.environ.get("FP_NO_PHASE", "0")))

# role -> side-stream pool index: aux, encoder weight gradients, mask decoder's / depth decoder's weight gradients (see Engine.__init__)
_STREAM_LAYOUT = "0,1,2,2"
_HEAD_WGRAD_SIDE = bool(int(os.environ.get("FP_HEAD_WGRAD_SIDE", "0")))   # 1: the heads' weight gradients on the decoder's weight-gradient stream.  Measured round 6 (3 alternating pairs, one box): 14.43 / 14.61 / 14.77 ms on the chain vs 14.54 / 14.74 / 14.68 on the side stream -- no gain (the step is bound by the chip's throughput, not by the chain), so the default stays on the chain
_WGRAD_PAIR_FORK = bool(int(os.environ.get("FP_WGRAD_PAIR_FORK", "1")))   # one stream fork per residual block for its two weight gradients (0: one each)
_HP_STEM = bool(int(os.environ.get("FP_HP_STEM", "1")))        # 0: the stem convolution on fp32 MFMA (rounds 1-3)
_NEED32_SYNC = bool(int(os.environ.get("FP_NEED32_SYNC", "1")))   # device-wide synchronizes around a first-touch pack of an fp32 layout, outside stream capture (Engine._need32)
_LAZY32 = bool(int(os.environ.get("FP_PACK_LAZY32", "1")))     # 0: every fp32 packed layout is refreshed every step (rounds 1-3)
_PACK_SIDE_WGS = int(os.environ.get("FP_PACK_SIDE_WGS", "0"))        # workgroups of the side-stream weight repack (0 = one per tile)
_PACK_DGRAD_LATE = bool(int(os.environ.get("FP_PACK_DGRAD_LATE", "0")))   # 1: the data-gradient layouts are repacked under the decoders' forward instead of the encoder's

SCALE_KEYS = ("1/8", "1/4", "1/2", "1/1")
_ALL_SCALES = frozenset(range(4))


class ConvRec:
    """One convolution: parameters + packed copies for the implicit-GEMM kernels."""

    def __init__(self, name, conv, stem=False, head=False):
        self.name = name
        self.w = conv.weight
        self.b = conv.bias
        self.Cout, self.Cin, self.K, _ = conv.weight.shape
        self.stride = conv.stride[0]
        self.pad = self.K // 2
        self.stem, self.head = stem, head
        self.wp = None    # forward packing
        self.wpd = None   # dgrad packing
        self.up2 = None   # (C0, C1) for convs fed by cat[nearest_x2(low C0), skip C1]: phase-decomposed packings below
        self.wph = self.wsk = self.wdu = self.wds = None   # fwd phase / fwd skip slice / dgrad 4x4 s2 / dgrad skip slice
        # bf16x3-split copies (3x3 stride-1 convs): forward, dgrad, and the skip-slice pair of the upsample convs
        self.bf3 = _BF3 and self.K == 3 and self.stride == 1 and not stem and not head
        self.wp3 = self.wpd3 = self.wsk3 = self.wds3 = self.wph3 = self.wdu3 = None
        # fp16-pair copies of the same four tile packings, and the slot holding max |w| they were scaled by
        self.hp = self.bf3 and _HP
        # ... and of the flattened-kernel packings for the convolutions the tile kernel does not take (3x3 stride 2, 1x1): fp_conv_igemm_hp
        self.hp_ig = _HP_IGEMM and not self.bf3 and not stem and not head and self.K in (1, 3)
        # ... or, with the exact operand format, bf16x3 copies of the same packings in the wp3 / wpd3 slots: fp_conv_igemm_bf3
        self.bf3_ig = _BF3_IGEMM and not self.bf3 and not stem and not head and self.K in (1, 3) and self.Cin % 4 == 0
        self.hp_f = self.hp_d = self.hp_sk = self.hp_ds = self.hp_ph = self.hp_du = None      # + phase forward / phase data-gradient
        self.wslot = None
        self.gw = None    # gradient views (flat grad buffer)
        self.gb = None


class BNRec:
    def __init__(self, name, bn, device):
        self.name = name
        self.bn = bn
        self.C = bn.num_features
        self.scale = torch.empty(self.C, device=device)
        self.shift = torch.empty(self.C, device=device)
        self.mean = torch.empty(self.C, device=device)
        self.invstd = torch.empty(self.C, device=device)
        self.gg = None
        self.gb = None


class BlockRec:
    def __init__(self, prefix, blk, device):
        self.prefix = prefix
        self.stride = blk.stride
        self.c1 = ConvRec(prefix + ".conv1", blk.conv1)
        self.bn1 = BNRec(prefix + ".bn1", blk.bn1, device)
        self.c2 = ConvRec(prefix + ".conv2", blk.conv2)
        self.bn2 = BNRec(prefix + ".bn2", blk.bn2, device)
        self.ds = self.bnd = None
        if blk.downsample is not None:
            self.ds = ConvRec(prefix + ".downsample.0", blk.downsample[0])
            self.bnd = BNRec(prefix + ".downsample.1", blk.downsample[1], device)
        self.Cin, self.Cout = self.c1.Cin, self.c1.Cout


class PSPRec:
    """pyramid pooling module of the segmentation decoder (preprocessing/segmentation/network.py:193-207): four
    AdaptiveAvgPool2d(P) -> 1x1 conv 512 -> 128 (no bias) -> bilinear(align_corners=True) branches, concatenated as
    [x, x6, x4, x2, x1] (1024 channels)"""

    def __init__(self, name, psp):
        self.blocks = []          # (pool size, reduce conv, channel offset in the concatenation)
        for bname, off in (("block1", 896), ("block2", 768), ("block3", 640), ("block4", 512)):
            blk = getattr(psp, bname)
            self.blocks.append((blk.pool_size, ConvRec("%s.%s.reduce" % (name, bname), blk.reduce), off))


class DecoderRec:
    """one SkipDecoder.  seg=False: a Footprints decoder (network.py:62-101: 2-channel heads, bilinear x8/x4/x2/x1 into the
    [B,4,H,W] outputs); seg=True: the ground-segmentation decoder (preprocessing/segmentation/network.py:54-99: 1-channel heads at
    their own resolution, optional pyramid pooling in front of block1)"""

    def __init__(self, name, dec, seg=False):
        self.name = name
        self.seg = seg
        self.sig = False if seg else dec.apply_sigmoid
        self.c0 = 2 if name == "depth_decoder" else 0
        self.head_scales = (1, 1, 1, 1) if seg else (8, 4, 2, 1)
        self.psp = PSPRec(name + ".PSP", dec.PSP) if (seg and dec.use_PSP) else None
        self.chans = [(1024 if self.psp is not None else 512, 256), (256, 128), (128, 64), (64, 64)]
        self.blocks = []
        for i in (1, 2, 3, 4):
            b = getattr(dec, "block%d" % i)
            p = "%s.block%d" % (name, i)
            self.blocks.append(dict(
                pre1=ConvRec(p + ".pre_concat_conv.conv1", b.pre_concat_conv.conv1),
                pre2=ConvRec(p + ".pre_concat_conv.conv2", b.pre_concat_conv.conv2),
                post1=ConvRec(p + ".post_concat_conv.conv1", b.post_concat_conv.conv1),
                post2=ConvRec(p + ".post_concat_conv.conv2", b.post_concat_conv.conv2)))
        self.heads = [ConvRec("%s.outconv%d.conv1" % (name, i), getattr(dec, "outconv%d" % i).conv1, head=True) for i in (1, 2, 3)]
        self.heads.append(ConvRec("%s.outconv4.1.conv1" % name, dec.outconv4[1].conv1, head=True))
        self.o41 = ConvRec("%s.outconv4.0.conv1" % name, dec.outconv4[0].conv1)
        self.o42 = ConvRec("%s.outconv4.0.conv2" % name, dec.outconv4[0].conv2)
        if _PHASE:
            for b in self.blocks:
                b["post1"].up2 = (b["post1"].Cin // 2, b["post1"].Cin // 2)
            self.o41.up2 = (self.o41.Cin, 0)

    def convs(self):
        out = [c for _, c, _ in self.psp.blocks] if self.psp is not None else []
        for b in self.blocks:
            out += [b["pre1"], b["pre2"], b["post1"], b["post2"]]
        return out + [self.o41, self.o42]


class AmaxBook:
    """amax slots of the fp16-pair kernels (include/footprints_hip.h): which slot holds max |x| of which activation / gradient tensor.

    Slots are addressed by a TAG (arena buffer name + role), so a recorded launch plan replays the same addresses every step; the
    whole pool is zeroed at the start of a forward.  An association tensor -> slot is made when a kernel publishes the amax of its
    output or when `get` reduces the tensor on the stream that is about to consume it; it is dropped when the arena hands the
    buffer out again (Engine.buf: every producer obtains its output there).  A slot filled on one stream is only reused by a
    consumer on the same stream -- or on any stream once `globalize` declared the streams joined (end of the forward pass)."""

    def __init__(self, device, nslots=2048):
        self.pool = torch.zeros(nslots * ops.amax_elems(), dtype=torch.int32, device=device)
        self.nslots = nslots
        self.by_tag = {}
        self.assoc = {}        # data_ptr -> (slot, raw stream handle or None = any stream)
        self.name_of = {}      # data_ptr -> arena name (Engine.buf)
        # an arena buffer that is handed out several times per step (g.g, g.da1, ...dzl, ...) gets a FRESH slot per hand-out (tag =
        # name + use count): a slot is only ever max-combined, so sharing one between uses would scale a later tensor by an
        # earlier, larger one's amax and give up mantissa bits of the 22-bit operand format.  The counts restart with every
        # forward (begin) and, for a backward pass repeated on one saved forward, at the point the forward left them (globalize).
        self.uses = {}
        self._uses_fwd = None

    def slot(self, tag):
        i = self.by_tag.get(tag)
        if i is None:
            i = self.by_tag[tag] = len(self.by_tag)
            if i >= self.nslots:
                raise RuntimeError("footprints_amd: amax slot pool exhausted")
        return self.pool[i * ops.amax_elems():(i + 1) * ops.amax_elems()]

    def begin(self):
        ops.zero_u32(self.pool)
        self.assoc.clear()
        self.uses = {}
        self._uses_fwd = None

    def handed_out(self, name):
        self.uses[name] = self.uses.get(name, 0) + 1

    def _tag(self, t):
        name = self.name_of.get(t.data_ptr(), "ptr%x" % t.data_ptr())
        return "%s#%d" % (name, self.uses.get(name, 0))

    def globalize(self):
        for k, (sl, _) in list(self.assoc.items()):
            self.assoc[k] = (sl, None)
        if self._uses_fwd is None:
            self._uses_fwd = dict(self.uses)
        else:                                  # a backward pass repeated on the same saved forward: the same slots again
            self.uses = dict(self._uses_fwd)

    def drop(self, t):
        self.assoc.pop(t.data_ptr(), None)

    def out_slot(self, t):
        """slot a producer of `t` publishes into (associate with `published` after the launch)"""
        return self.slot("out:" + self._tag(t))

    def published(self, t, sl, any_stream=False):
        self.assoc[t.data_ptr()] = (sl, None if any_stream else ops.stream())

    def get(self, t, any_stream=False):
        """slot holding max |t|, valid for a consumer launched on the current stream from now on"""
        e = self.assoc.get(t.data_ptr())
        cur = ops.stream()
        if e is not None and (e[1] is None or e[1] == cur):
            if any_stream and e[1] is not None:
                self.assoc[t.data_ptr()] = (e[0], None)
            return e[0]
        sl = self.slot("in:%s@%x" % (self._tag(t), cur))
        ops.amax_f32(t, sl)
        self.assoc[t.data_ptr()] = (sl, None if any_stream else cur)
        return sl


class Engine:
    def __init__(self, model):
        self.model = model
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("footprints_amd.Engine needs the model on a CUDA (HIP) device")
        L.load()
        dev = self.device
        enc = model.encoder
        self.stem = ConvRec("encoder.layer0.0", enc.layer0[0], stem=True)
        self.bn0 = BNRec("encoder.layer0.1", enc.layer0[1], dev)
        self.blocks = [BlockRec(p, b, dev) for p, b in enc.blocks()]
        if hasattr(model, "decoder"):        # Segmentor (preprocessing/segmentation/network.py:13-25): one decoder, 1-channel heads
            self.decoders = [DecoderRec("decoder", model.decoder, seg=True)]
        else:
            self.decoders = [DecoderRec("mask_decoder", model.mask_decoder), DecoderRec("depth_decoder", model.depth_decoder)]
        self._bufs = {}
        self.amax = AmaxBook(dev)
        self._flatten()
        self._alloc_packed()
        # 1-channel heads (segmentation) run through the Cin -> 2 head kernels with a zero second filter: padded copies of the
        # weight / bias (refreshed with the packed weights) and of their gradients (row 0 is handed to the parameter's gradient)
        for d in self.decoders:
            for hd in d.heads:
                hd.pad = hd.Cout == 1
                if hd.pad:
                    hd.w2 = torch.zeros((2, hd.Cin, 3, 3), device=dev)
                    hd.b2 = torch.zeros(2, device=dev)
                    hd.gw2 = torch.zeros((2, hd.Cin, 3, 3), device=dev)
                    hd.gb2 = torch.zeros(2, device=dev)
        self._pack_table = None
        self._pack_table_key = None
        self._pack_ev = None
        self._pack_ev_dgrad = None
        self._pack_dgrad_pending = False
        # fp32 packed layouts are repacked LAZILY (round 4): with fp16-pair operands nearly every convolution reads only its *_hp layouts, and
        # refreshing the fp32 FWD / DGRAD / phase layouts of all 31 M weights as well was half of the repack's 1 GB of traffic per step.  A
        # layout joins the batched repack the first time a launch reads it (_need32: packed on the spot, on every later refresh by the
        # table); the packed buffer starts as NaN, so a reader this bookkeeping does not know about fails loudly instead of reading stale weights.
        self._w32_lazy = {}         # data_ptr of an fp32 layout -> its pack job
        self._w32_used = set()      # ... that some launch has read: part of the batched table
        self._w32_fresh = set()     # ... packed since the weights last changed
        self._w32_tables = {}
        self.fold_eval = _FOLD
        # opt-in inference precision: "bf16x2" runs the 3x3 stride-1 tile convolutions of an eval / no-grad forward with two bf16 terms per
        # operand (three MFMA products instead of six).  NOT exact (measured output error in profiles/round2_notes.md); never used when
        # a backward pass may follow.  model.inference_precision = "bf16x2" sets it through FootprintNetwork.forward.
        self.inference_bf16x2 = False
        self._fold_ready = False        # folded (conv * BN scale, BN shift) copies of the encoder are current
        self._fold_buf = None
        self._fold_vers = None
        self.weights_dirty = True
        self._versions = None
        self.saved = None
        self.debug_hook = None      # tests/debugging: called after every encoder block of the backward schedule
        self.concurrent = _CONCURRENT
        # Side streams.  The HIP runtime maps a process's streams onto at most FOUR hardware queues per priority (GPU_MAX_HW_QUEUES;
        # five or six measured 23 ms per step instead of 13.9); a fifth stream silently shares the queue of an earlier one, every
        # packet carries the barrier bit, so two streams on one hardware queue execute strictly in order -- and which pair shares
        # used to be decided by the order in which the streams first ran a kernel (with four roles + the main stream the two
        # decoders' weight-gradient streams ended up on one queue; once an RCCL communicator had created its own streams first, other
        # pairs did: that was round 2's unexplained 2.7 ms "data-parallel tax", profiles/round3_notes.md).  So the engine owns the
        # decision: the four roles are mapped onto THREE side streams (main + 3 = the four queues), and every stream runs one tiny
        # kernel right here, so its hardware queue is created and owned before any other library of the process creates streams.
        # FP_STREAM_LAYOUT = "aux,wg,dwg0,dwg1" as indices into the side-stream pool.
        lay = [int(v) for v in os.environ.get("FP_STREAM_LAYOUT", _STREAM_LAYOUT).split(",")]
        if len(lay) != 4 or min(lay) < 0:
            raise ValueError("FP_STREAM_LAYOUT: four pool indices expected (aux, wg, dwg0, dwg1)")
        pool = [torch.cuda.Stream(device=self.device) for _ in range(max(lay) + 1)]
        self.side_streams = pool
        self.aux = pool[lay[0]]     # second decoder, downsample branch of the stride-2 blocks
        self.wg = pool[lay[1]]      # encoder weight gradients, side part of the weight repack
        self.dwg = [pool[lay[2]], pool[lay[3]]]   # decoder weight gradients
        tiny = torch.zeros(64, device=self.device)
        torch.cuda.synchronize(self.device)          # `tiny` may sit in a recycled block: its previous owner's queued work first (see _flatten)
        for st in pool:
            with ops.on_stream(st):
                ops.fill(tiny, 0.0)
        torch.cuda.synchronize(self.device)
        self._ev_dF = [None] * 5

    # ------------------------------------------------------------------------------------------------
    # parameters: one flat buffer, one flat gradient buffer (forward order, 16-byte aligned slots)
    # ------------------------------------------------------------------------------------------------
    def _flatten(self):
        named = self.model.live_named_parameters()
        self.live_names = [n for n, _ in named]
        self.live_params = [p for _, p in named]
        offs, total = [], 0
        for p in self.live_params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat_param = torch.zeros(total, device=self.device)
        self.flat_grad = torch.zeros(total, device=self.device)
        ops.bump_alloc_generation()
        self.offsets = offs
        self.grad_views = []
        with torch.no_grad():
            for p, o in zip(self.live_params, offs):
                v = self.flat_param[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.grad_views.append(self.flat_grad[o:o + p.numel()].view(p.shape))
        # `p.data = v` dropped the last reference to every original parameter storage while the copies OUT of them are only queued: the caching
        # allocator hands those blocks to the next allocations of this stream at once -- fine for launches on this stream, not for a block that
        # is then first written from ANOTHER stream (the start-up fills of the side streams below, arena buffers of the decoder streams).  With
        # work pending on the stream when the engine is built (a long kernel, a broadcast), 64 floats of a head's weights were zeroed before
        # they were copied (round 6: found by tests/test_gpu_lifetime.py behind a spin kernel; first step of a fresh engine only).  Start-up
        # and re-flattening only: wait for the copies.
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize(self.device)
        gv = dict(zip(self.live_names, self.grad_views))

        def bind_conv(c):
            c.gw = gv[c.name + ".weight"]
            c.gb = gv[c.name + ".bias"] if c.b is not None else None

        def bind_bn(b):
            b.gg, b.gb = gv[b.name + ".weight"], gv[b.name + ".bias"]

        bind_conv(self.stem)
        bind_bn(self.bn0)
        for blk in self.blocks:
            for c in (blk.c1, blk.c2, blk.ds):
                if c is not None:
                    bind_conv(c)
            for b in (blk.bn1, blk.bn2, blk.bnd):
                if b is not None:
                    bind_bn(b)
        for d in self.decoders:
            for c in d.convs() + d.heads:
                bind_conv(c)

    def invalidate(self):
        """the parameters were written behind autograd's back (`p.data.copy_`, a broadcast, ...): `Parameter._version` does not
        move for `.data` writes, so the packed / BN-folded weight copies must be told"""
        self.weights_dirty = True
        self._fold_ready = False

    def params_alias_flat(self):
        p0 = self.live_params[0]
        return p0.data_ptr() == self.flat_param.data_ptr()

    def all_convs(self):
        out = [self.stem]
        for blk in self.blocks:
            out += [c for c in (blk.c1, blk.c2, blk.ds) if c is not None]
        for d in self.decoders:
            out += d.convs()
        return out

    def _alloc_packed(self):
        total = 0
        plan = []
        for c in self.all_convs():
            nf = ops.packed_weight_elems(c.Cout, c.Cin, c.K, False, c.stem)
            nd = 0 if c.stem else ops.packed_weight_elems(c.Cout, c.Cin, c.K, True, False)
            ex = [0, 0, 0, 0]
            if c.up2 is not None:
                C0, C1 = c.up2
                ex = [ops.up2_packed_weight_elems(c.Cout, C0), ops.packed_weight_elems(c.Cout, C1, 3) if C1 else 0,
                      ops.up2_packed_weight_elems(C0, c.Cout), ops.packed_weight_elems(c.Cout, C1, 3, True) if C1 else 0]
            if c.bf3:
                if c.up2 is None:
                    ex += [ops.packed_weight_elems_bf3(c.Cout, c.Cin, 3, False), ops.packed_weight_elems_bf3(c.Cout, c.Cin, 3, True), 0, 0, 0, 0]
                else:
                    C0, C1 = c.up2
                    ex += [ops.packed_weight_elems_bf3(c.Cout, c.Cin, 3, False) if C1 else 0, 0,      # full concat pack: small images
                           ops.packed_weight_elems_bf3(c.Cout, C1, 3, False) if C1 else 0,
                           ops.packed_weight_elems_bf3(c.Cout, C1, 3, True) if C1 else 0, ops.up2_packed_weight_elems(c.Cout, C0) * 3 // 2,
                           ops.up2_packed_weight_elems(C0, c.Cout) * 3 // 2]
            elif c.bf3_ig:           # stride-2 3x3 / 1x1: the flattened kernel's bf16x3 packings live in the wp3 / wpd3 slots
                ex += [ops.packed_weight_elems_bf3(c.Cout, c.Cin, c.K, False), ops.packed_weight_elems_bf3(c.Cout, c.Cin, c.K, True), 0, 0, 0, 0]
            else:
                ex += [0, 0, 0, 0, 0, 0]
            if c.hp or c.hp_ig:      # fp16-pair copies of the tile packings (same roles as wp3 / wpd3 / wsk3 / wds3)
                if c.up2 is None:
                    ex += [ops.packed_weight_elems_hp(c.Cout, c.Cin, c.K, False), ops.packed_weight_elems_hp(c.Cout, c.Cin, c.K, True), 0, 0, 0, 0]
                else:
                    C0, C1 = c.up2
                    ex += [ops.packed_weight_elems_hp(c.Cout, c.Cin, 3, False) if C1 else 0, 0,
                           ops.packed_weight_elems_hp(c.Cout, C1, 3, False) if C1 else 0, ops.packed_weight_elems_hp(c.Cout, C1, 3, True) if C1 else 0,
                           ops.up2_packed_weight_elems(c.Cout, C0), ops.up2_packed_weight_elems(C0, c.Cout)]      # two fp16 = one float per weight
            else:
                ex += [0, 0, 0, 0, 0, 0]
            plan.append((c, total, nf, nd, ex))
            total += nf + nd + sum(ex)
        self.packed = torch.full((total,), float("nan"), device=self.device)      # see _need32
        for c, o, nf, nd, ex in plan:
            c.wp = self.packed[o:o + nf]
            c.wpd = self.packed[o + nf:o + nf + nd] if nd else None
            o += nf + nd
            views = []
            for n in ex:
                views.append(self.packed[o:o + n] if n else None)
                o += n
            c.wph, c.wsk, c.wdu, c.wds, c.wp3, c.wpd3, c.wsk3, c.wds3, c.wph3, c.wdu3, c.hp_f, c.hp_d, c.hp_sk, c.hp_ds, c.hp_ph, c.hp_du = views
        convs = self.all_convs()
        self.wamax = torch.zeros(len(convs) * ops.amax_elems(), dtype=torch.int32, device=self.device)
        for i, c in enumerate(convs):
            c.wslot = self.wamax[i * ops.amax_elems():(i + 1) * ops.amax_elems()]
        if _HP and _HP_STEM:        # the stem's fp16-pair layout (FP_PACK_STEM_HP: 11 K-steps x 2 planes x 64 x 16 halves)
            self.stem.hp_f = torch.full((11 * 64 * 16,), float("nan"), device=self.device)

    def refresh_packed(self, force=False, overlap=False):
        """repack every convolution's weights if they changed.  overlap=True (Engine.forward only): all but the stem / layer1
        tables are packed on a side stream and the caller waits for self._pack_ev (self._wait_pack) before layer2."""
        vers = tuple(c.w._version for c in self.all_convs())
        key = (self.flat_param.data_ptr(), bool(self.inference_bf16x2))
        if not (force or self.weights_dirty or vers != self._versions or key != self._pack_table_key):
            return
        if self._pack_table is None or self._pack_table_key != key:
            # (re)building the device-resident job tables allocates and fills small tensors from pageable host memory: like a first touch in
            # _need32 it drains the device first (start-up and the step after a first touch only; never under stream capture)
            if _NEED32_SYNC and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize(self.device)

            def jobs_of(convs):
                jobs = []
                for c in convs:
                    jobs.append((L.PACK_STEM if c.stem else L.PACK_FWD, c.w.data, c.wp, 0, c.Cin))
                    if c.stem and c.hp_f is not None:
                        jobs.append((L.PACK_STEM_HP, c.w.data, c.hp_f, 0, c.Cin, c.wslot))
                    if c.wpd is not None:
                        jobs.append((L.PACK_DGRAD, c.w.data, c.wpd, 0, c.Cin))
                    if c.hp or c.hp_ig:      # fp16-pair packings, scaled by the weight tensor's amax slot
                        if c.hp_f is not None:
                            jobs.append((L.PACK_FWD_HP, c.w.data, c.hp_f, 0, c.Cin, c.wslot))
                        if c.hp_d is not None:
                            jobs.append((L.PACK_DGRAD_HP, c.w.data, c.hp_d, 0, c.Cin, c.wslot))
                        if c.hp_sk is not None:
                            jobs.append((L.PACK_FWD_HP, c.w.data, c.hp_sk, c.up2[0], c.up2[1], c.wslot))
                            jobs.append((L.PACK_DGRAD_HP, c.w.data, c.hp_ds, c.up2[0], c.up2[1], c.wslot))
                        if c.hp_ph is not None:
                            jobs.append((L.PACK_UP2_FWD_HP, c.w.data, c.hp_ph, 0, c.up2[0], c.wslot))
                            jobs.append((L.PACK_UP2_DGRAD_HP, c.w.data, c.hp_du, 0, c.up2[0], c.wslot))
                    keep3 = (not c.hp) or self.inference_bf16x2 or not _HP_TILE       # the bf16 tile packings: only where a kernel still reads them
                    if c.wp3 is not None and keep3:
                        jobs.append((L.PACK_FWD_BF3, c.w.data, c.wp3, 0, c.Cin))
                    if c.wpd3 is not None and keep3:
                        jobs.append((L.PACK_DGRAD_BF3, c.w.data, c.wpd3, 0, c.Cin))
                    if c.wsk3 is not None and keep3:
                        jobs.append((L.PACK_FWD_BF3, c.w.data, c.wsk3, c.up2[0], c.up2[1]))
                        jobs.append((L.PACK_DGRAD_BF3, c.w.data, c.wds3, c.up2[0], c.up2[1]))
                    if c.wph3 is not None and (not c.hp or not _HP_TILE):
                        jobs.append((L.PACK_UP2_FWD_BF3, c.w.data, c.wph3, 0, c.up2[0]))
                        jobs.append((L.PACK_UP2_DGRAD_BF3, c.w.data, c.wdu3, 0, c.up2[0]))
                    if c.up2 is not None:
                        C0, C1 = c.up2
                        jobs.append((L.PACK_UP2_FWD, c.w.data, c.wph, 0, C0))
                        jobs.append((L.PACK_UP2_DGRAD, c.w.data, c.wdu, 0, C0))
                        if C1:
                            jobs.append((L.PACK_FWD, c.w.data, c.wsk, C0, C1))
                            jobs.append((L.PACK_DGRAD, c.w.data, c.wds, C0, C1))
                keep = []
                for j in jobs:
                    if _LAZY32 and j[0] in (L.PACK_FWD, L.PACK_DGRAD, L.PACK_UP2_FWD, L.PACK_UP2_DGRAD):
                        self._w32_lazy[j[2].data_ptr()] = j
                        if j[2].data_ptr() not in self._w32_used:
                            continue
                    keep.append(j)
                return keep
            # the stem and layer1 (0.2 M parameters) on the calling stream; everything else (31 M) on a side stream under the stem / layer1
            # kernels, in two parts with an event each: the forward layouts (the forward waits for them where layer2 starts, self._pack_ev) and
            # the data-gradient layouts (first read by the decoder backward, self._pack_ev_dgrad).  The side-stream launches are persistent
            # with at most _PACK_SIDE_WGS workgroups: a repack that fills every wave slot of the chip stretched the first layer-1
            # convolutions from 30 to 142 us (ordered trace, profiles/round4_notes.md)
            first = [self.stem] + [c for blk in self.blocks if blk.Cout == 64 and blk.stride == 1 for c in (blk.c1, blk.c2)]
            rest = [c for c in self.all_convs() if not any(c is f for f in first)]
            rest_jobs = jobs_of(rest)
            is_dgrad = lambda j: j[0] in (L.PACK_DGRAD, L.PACK_DGRAD_BF3, L.PACK_DGRAD_HP, L.PACK_UP2_DGRAD, L.PACK_UP2_DGRAD_BF3, L.PACK_UP2_DGRAD_HP)
            self._pack_table = (ops.build_pack_table(jobs_of(first), self.device), ops.build_pack_table(rest_jobs, self.device),
                                ops.build_pack_table([j for j in rest_jobs if not is_dgrad(j)], self.device),
                                ops.build_pack_table([j for j in rest_jobs if is_dgrad(j)], self.device))
            self._pack_table_key = key      # parameters live in self.flat_param: pointers are stable
            ops.bump_alloc_generation()     # the previous tables (device-resident job lists) are gone
        first_t, rest_all, rest_fwd, rest_dgrad = self._pack_table
        if _HP:
            ops.zero_u32(self.wamax)
            ops.pack_weights_amax(first_t)
        ops.pack_weights_batched(first_t)
        cur = ops.current_stream()
        if self.concurrent and overlap:
            ops.event_wait(self.wg, self._record(cur))
            with ops.on_stream(self.wg):
                if _HP:
                    ops.pack_weights_amax(rest_all)
                ops.pack_weights_batched(rest_fwd, max_wgs=_PACK_SIDE_WGS)
                self._pack_ev = self._record(self.wg)
                if _PACK_DGRAD_LATE:
                    self._pack_dgrad_pending = True
                else:
                    ops.pack_weights_batched(rest_dgrad, max_wgs=_PACK_SIDE_WGS)
                    self._pack_ev_dgrad = self._record(self.wg)
        else:
            if _HP:
                ops.pack_weights_amax(rest_all)
            ops.pack_weights_batched(rest_all)
            self._pack_ev = self._pack_ev_dgrad = None
        for d in self.decoders:
            for hd in d.heads:
                if hd.pad:
                    hd.w2[0].copy_(hd.w.data[0])
                    hd.b2[0:1].copy_(hd.b.data)
        self._fold_ready = False
        self._versions = vers
        self.weights_dirty = False
        self._w32_fresh = set(self._w32_used)       # exactly the lazy layouts the tables above contain

    # ------------------------------------------------------------------------------------------------
    # activation arena
    # ------------------------------------------------------------------------------------------------
    def buf(self, name, shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= s
        t = self._bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            ops.release(t, "arena:" + name)           # replacing a buffer drops the old one: kernels of the previous shape may still be using it
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._bufs[name] = t
            self.amax.name_of[t.data_ptr()] = name
            ops.bump_alloc_generation()               # recorded launch plans hold the old address
        self.amax.assoc.pop(t.data_ptr(), None)       # the buffer is about to be rewritten: its amax slot no longer describes it
        self.amax.handed_out(name)                    # ... and its next producer publishes into a fresh slot
        return t[:n].view(shape)

    # ------------------------------------------------------------------------------------------------
    # small helpers over the ops
    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _record(stream):
        """event id on `stream` (library events, not framework events: a recording launch plan has to see every ordering edge)"""
        return ops.event_record(stream)

    def _bn_coeffs(self, rec, z, training):
        bn = rec.bn
        M = z.numel() // rec.C
        if training and getattr(rec, "stats_nblk", 0) > 0:     # the producing convolution wrote the partials (_conv_enc)
            nblk, rec.stats_nblk = rec.stats_nblk, 0
            ops.bn_train_stats_partials(rec.stats_part, nblk, rec.C, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var,
                                        bn.num_batches_tracked, rec.mean, rec.invstd, rec.scale, rec.shift, bn.eps,
                                        bn.momentum if bn.momentum is not None else 0.1)
        elif training:
            ops.bn_train_stats(z.view(M, rec.C), bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var,
                               bn.num_batches_tracked, rec.mean, rec.invstd, rec.scale, rec.shift, bn.eps,
                               bn.momentum if bn.momentum is not None else 0.1)
        else:
            ops.bn_eval_coeffs(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, rec.scale, rec.shift, bn.eps)

    def _bn_apply(self, rec, z, out, residual=None, relu=True):
        M = z.numel() // rec.C
        so = self.amax.out_slot(out) if _HP else None        # the next convolution's operand scale comes out of this pass
        ops.bn_apply(z.view(M, rec.C), rec.scale, rec.shift, out.view(M, rec.C),
                     residual=None if residual is None else residual.view(M, rec.C), relu=relu, amax_out=so)
        if so is not None:
            self.amax.published(out, so)
        return out

    def _bn(self, rec, z, out, training, residual=None, relu=True):
        """BatchNorm of one encoder conv output: train mode = batch statistics + normalisation (statistics -- out of the producing tile
        convolution's epilogue where that applies --, their combination, normalisation)"""
        self._bn_coeffs(rec, z, training)
        return self._bn_apply(rec, z, out, residual=residual, relu=relu)

    @staticmethod
    def _head_wb(hd):
        return (hd.w2, hd.b2) if hd.pad else (hd.w.data, hd.b.data)

    def _head_wgrad(self, hd, x, dzl, acc, side=None):
        """weight / bias gradient of a 2-channel head.  side = a stream (round 6): launched there behind a fork from the current stream, like
        the convolutions' weight gradients (_wgrad) -- nothing on the data-gradient chain reads it, and at full resolution it is 160 us of
        HBM-bound work that used to sit in front of the first big data gradient of the backward pass (ordered trace, profiles/round5_*).
        The caller keeps x and dzl untouched until the join at the end of Engine.backward: dzl lives in a buffer of its own per scale."""
        def launch():
            if not hd.pad:
                return ops.head_wgrad(x, dzl, hd.gw, hd.gb, accumulate=acc)
            ops.head_wgrad(x, dzl, hd.gw2, hd.gb2, accumulate=False)
            if acc:
                hd.gw.add_(hd.gw2[0:1])
                hd.gb.add_(hd.gb2[0:1])
            else:
                hd.gw.copy_(hd.gw2[0:1])
                hd.gb.copy_(hd.gb2[0:1])
        if side is None or not _HEAD_WGRAD_SIDE:
            return launch()
        ops.event_wait(side, self._record(ops.current_stream()))
        with ops.on_stream(side):
            launch()

    def _cv(self, d, src, w32, w3, out, hp=None, publish=True, **kw):
        """one 3x3 / 1x1 convolution or data-gradient launch: the split-operand tile kernel where it applies (hp = (fp16-pair packing,
        weight amax slot): three fp16 products; else w3: six bf16 products), else fp_conv_igemm"""
        use_hp = _HP_TILE and hp is not None and hp[0] is not None and not ops._bf16x2
        if (w3 is not None or use_hp) and ops.conv3x3_bf3_supported(d):
            if use_hp:
                return self._cv_hp(d, src, hp[0], hp[1], out, publish=publish, **kw)
            return ops.conv3x3_bf3(d, src, w3, out, **kw)
        if _BF3_IGEMM and w3 is not None and not ops._bf16x2 and ops.conv_igemm_hp_supported(d):
            return ops.conv_igemm_bf3(d, src, w3, out, **kw)
        if _HP_IGEMM and hp is not None and hp[0] is not None and not ops._bf16x2 and ops.conv_igemm_hp_supported(d):
            return ops.conv_igemm_hp(d, src, hp[0], out, self.amax.get(src), hp[1], **kw)
        return ops.conv_igemm(d, src, None, self._need32(w32), out, **kw)

    def _sink_slot(self, t):
        """amax slot the producer of `t` publishes into (None with the bf16 operand format); follow the launch with _sink_done(t)"""
        return self.amax.out_slot(t) if _HP else None

    def _sink_done(self, t):
        if _HP:
            self.amax.published(t, self.amax.out_slot(t))

    def _cv_hp(self, d, src, wh, wslot, out, src1=None, publish=True, **kw):
        book = self.amax
        sa = book.get(src)
        sa1 = book.get(src1) if (src1 is not None and d.C1) else None
        so = None if ((d.epi & L.EPI_ACCUM) or not publish) else book.out_slot(out)        # an accumulated tensor is never a tile-conv operand
        ops.conv3x3_hp(d, src, wh, out, sa, wslot, amax_out=so, src1=src1, amax_src1=sa1, **kw)
        if so is not None:
            book.published(out, so)
        return out

    # ------------------------------------------------------------------------------------------------
    # inference fast path: eval-mode BatchNorm folded into the encoder convs
    # ------------------------------------------------------------------------------------------------
    def _enc_pairs(self):
        pairs = [(self.stem, self.bn0)]
        for blk in self.blocks:
            pairs += [(blk.c1, blk.bn1), (blk.c2, blk.bn2)]
            if blk.ds is not None:
                pairs.append((blk.ds, blk.bnd))
        return pairs

    def _build_fold(self):
        """W' = W * gamma / sqrt(running_var + eps) per output channel, bias' = beta - running_mean * that: one conv launch with
        bias (+ residual) + ReLU replaces conv -> coefficients -> apply.  Rebuilt when weights or running statistics changed."""
        pairs = self._enc_pairs()
        if self._fold_buf is None:
            total = 0
            for c, _ in pairs:
                n3 = ops.packed_weight_elems_bf3(c.Cout, c.Cin, c.K, False) if (c.bf3 or c.bf3_ig) else 0
                n3 += ops.packed_weight_elems_hp(c.Cout, c.Cin, c.K, False) if (c.hp or c.hp_ig) else 0
                total += c.w.numel() + ops.packed_weight_elems(c.Cout, c.Cin, c.K, False, c.stem) + n3 + c.Cout
            self._fold_buf = torch.empty(total, device=self.device)
            o = 0
            for c, rec in pairs:
                n = c.w.numel()
                c.fw = self._fold_buf[o:o + n].view(c.w.shape)
                o += n
                n = ops.packed_weight_elems(c.Cout, c.Cin, c.K, False, c.stem)
                c.fwp = self._fold_buf[o:o + n]
                o += n
                n = ops.packed_weight_elems_bf3(c.Cout, c.Cin, c.K, False) if (c.bf3 or c.bf3_ig) else 0
                c.fwp3 = self._fold_buf[o:o + n] if n else None
                o += n
                n = ops.packed_weight_elems_hp(c.Cout, c.Cin, c.K, False) if (c.hp or c.hp_ig) else 0
                c.fhp = self._fold_buf[o:o + n] if n else None
                o += n
                c.fslot = torch.zeros(ops.amax_elems(), dtype=torch.int32, device=self.device) if n else None
                rec.fshift = self._fold_buf[o:o + c.Cout]
                o += c.Cout
        for c, rec in pairs:
            bn = rec.bn
            ops.bn_eval_coeffs(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, rec.scale, rec.fshift, bn.eps)
            ops.scale_rows(c.w.data, rec.scale, c.fw)
            ops.pack_conv_weight(c.fw, c.fwp, c.stem)
            if c.fwp3 is not None and (c.fhp is None or self.inference_bf16x2):
                ops.pack_conv_weight_bf3(c.fw, c.fwp3, False)
            if c.fhp is not None:
                ops.pack_conv_weight_hp(c.fw, c.fhp, c.fslot, False)
        self._fold_ready = True

    def _encoder_eval_folded(self, image, N, H, W, S):
        buf = self.buf
        vers = tuple(t._version for _, rec in self._enc_pairs() for t in (rec.bn.running_mean, rec.bn.running_var, rec.bn.weight, rec.bn.bias))
        if not self._fold_ready or vers != self._fold_vers:
            self._build_fold()
            self._fold_vers = vers
        h, w = H // 2, W // 2
        f0 = buf("f0", (N, h, w, 64))
        ops.conv_igemm(ops.make_desc(N, h, w, H, W, 3, 0, 64, 7, 2, 3, L.GATHER_STEM, act=L.ACT_RELU), image, None, self.stem.fwp, f0,
                       bias=self.bn0.fshift)
        hp, wp_ = (h + 1) // 2, (w + 1) // 2
        pool = buf("pool", (N, hp, wp_, 64))
        so = self.amax.out_slot(pool) if _HP else None
        ops.maxpool_fwd(f0, pool, buf("pool.argmax", (N, hp, wp_, 64), torch.uint8), amax_out=so)
        if so is not None:
            self.amax.published(pool, so)
        feats, dims = [f0], [(h, w)]
        x, h, w = pool, hp, wp_
        self._wait_pack()          # the decoders' packed weights (the folded encoder copies are packed by _build_fold)
        for i, blk in enumerate(self.blocks):
            s = blk.stride
            oh, ow = (h - 1) // s + 1, (w - 1) // s + 1
            d1 = ops.make_desc(N, oh, ow, h, w, blk.c1.Cin, 0, blk.Cout, 3, s, 1, L.GATHER_FWD_ZERO, act=L.ACT_RELU)
            ev_idt = None
            if blk.ds is not None:
                dd = ops.make_desc(N, oh, ow, h, w, blk.ds.Cin, 0, blk.Cout, 1, s, 0, L.GATHER_FWD_ZERO)
                if self.concurrent and _DS_AUX:           # 1x1 shortcut beside conv1 on the (still idle) aux stream
                    if _HP_IGEMM:
                        self.amax.get(x, any_stream=True)
                    ops.event_wait(self.aux, self._record(ops.current_stream()))
                    with ops.on_stream(self.aux):
                        idt = self._cv(dd, x, blk.ds.fwp, blk.ds.fwp3, buf("b%d.idt" % i, (N, oh, ow, blk.Cout)), hp=(blk.ds.fhp, blk.ds.fslot),
                                       bias=blk.bnd.fshift)
                        ev_idt = self._record(self.aux)
                else:
                    idt = self._cv(dd, x, blk.ds.fwp, blk.ds.fwp3, buf("b%d.idt" % i, (N, oh, ow, blk.Cout)), hp=(blk.ds.fhp, blk.ds.fslot),
                                   bias=blk.bnd.fshift)
            else:
                idt = x
            a1 = self._cv(d1, x, blk.c1.fwp, blk.c1.fwp3, buf("b%d.a1" % i, (N, oh, ow, blk.Cout)), hp=(blk.c1.fhp, blk.c1.fslot),
                          bias=blk.bn1.fshift)
            if ev_idt is not None:
                ops.event_wait(ops.current_stream(), ev_idt)
            d2 = ops.make_desc(N, oh, ow, oh, ow, blk.Cout, 0, blk.Cout, 3, 1, 1, L.GATHER_FWD_ZERO, act=L.ACT_RELU)
            out = self._cv(d2, a1, blk.c2.fwp, blk.c2.fwp3, buf("b%d.out" % i, (N, oh, ow, blk.Cout)), hp=(blk.c2.fhp, blk.c2.fslot),
                           bias=blk.bn2.fshift, addend=idt)
            x, h, w = out, oh, ow
            if (i + 1 == len(self.blocks)) or (self.blocks[i + 1].stride == 2):
                feats.append(out)
                dims.append((h, w))
        S["blocks"] = []
        S["feats"], S["dims"] = feats, dims

    def _wait_pack(self):
        """the side-stream part of the weight repack (everything after layer1) must have landed"""
        if self._pack_ev is not None:
            ops.event_wait(ops.current_stream(), self._pack_ev)
            self._pack_ev = None

    def _need32(self, t):
        """`t`: an fp32 packed layout about to be read by a launch on the current stream (see __init__)"""
        if t is None:
            return t
        key = t.data_ptr()
        if key in self._w32_fresh or key not in self._w32_lazy:
            return t
        # (round 5, notes section 7) a first touch allocates small device tensors (the one-job table) and copies into them from pageable host
        # memory between two convolution launches: drain the device FIRST, so that neither the allocator's recycled blocks nor the copies can
        # meet kernels that are still in flight -- the driver's whole-suite command reproducibly corrupted one split-K convolution of
        # tests/test_gpu_network.py::test_g2 (and only with lazy packing AND split-K on) until this was here
        if _NEED32_SYNC and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize(self.device)
        tab = self._w32_tables.get(key)
        if tab is None:
            tab = self._w32_tables[key] = ops.build_pack_table([self._w32_lazy[key]], self.device)
        ops.pack_weights_batched(tab)
        # first touch only.  Readers on the engine's other streams must see the layout too: an event edge from this stream to each of them
        # (round 5, ADVICE r4: the device-wide synchronize that stood here is illegal under stream capture and invisible to a recording
        # launch plan).  And a plan that is recording RIGHT NOW would bake this one-off launch into every replay: bumping the allocation
        # generation makes TrainStep discard it and record again on the next step, when the layout is part of the regular repack.
        ev = self._record(ops.current_stream())
        cur = ops.current_stream().cuda_stream
        for st in {s.cuda_stream: s for s in (self.aux, self.wg, self.dwg[0], self.dwg[1])}.values():
            if st.cuda_stream != cur:
                ops.event_wait(st, ev)
        ops.bump_alloc_generation()
        # ... and, wherever it is legal, still the device-wide synchronize of rounds 1-4 behind the pack (a first touch happens a handful of
        # times in a process's life, never in a steady-state step); the event edges above are what a stream capture / a recording plan needs
        if _NEED32_SYNC and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize(self.device)
        self._w32_fresh.add(key)
        if key not in self._w32_used:
            self._w32_used.add(key)
            self._pack_table = None                 # the next refresh rebuilds the tables with this layout in them
        return t

    def _launch_pack_dgrad(self):
        """FP_PACK_DGRAD_LATE: the deferred part of refresh_packed, on the repack stream behind everything queued on the calling stream"""
        if self._pack_dgrad_pending:
            self._pack_dgrad_pending = False
            ops.event_wait(self.wg, self._record(ops.current_stream()))
            with ops.on_stream(self.wg):
                ops.pack_weights_batched(self._pack_table[3], max_wgs=_PACK_SIDE_WGS)
                self._pack_ev_dgrad = self._record(self.wg)

    def _wait_pack_dgrad(self):
        """... and the data-gradient layouts, before the first backward kernel that reads one"""
        self._launch_pack_dgrad()
        if self._pack_ev_dgrad is not None:
            ops.event_wait(ops.current_stream(), self._pack_ev_dgrad)
            self._pack_ev_dgrad = None

    def _conv_enc(self, c, x, N, H, W, out, bn=None, part="bn.part"):
        """encoder convolution; `bn` = the BatchNorm record that follows in train mode: a tile-kernel launch then also writes the
        Welford partials of its output (ops.bn_stats_out -> the launch's fp_aux) and _bn_coeffs skips the statistics pass over the activation"""
        OH, OW = (H + 2 * c.pad - c.K) // c.stride + 1, (W + 2 * c.pad - c.K) // c.stride + 1
        d = ops.make_desc(N, OH, OW, H, W, c.Cin, 0, c.Cout, c.K, c.stride, c.pad, L.GATHER_FWD_ZERO)
        bo = None
        if bn is not None:
            bn.stats_nblk = 0
            tile = (c.wp3 is not None or (_HP_TILE and c.hp_f is not None and not ops._bf16x2)) and ops.conv3x3_bf3_supported(d)
            # ... or a strided / 1 x 1 convolution through the fp16-pair implicit GEMM: its split grids emit from their reduce launch
            tile = tile or (_HP_IGEMM and c.hp_f is not None and not ops._bf16x2 and ops.conv_igemm_hp_supported(d))
            tile = tile or (_BF3_IGEMM and c.bf3_ig and not ops._bf16x2 and ops.conv_igemm_hp_supported(d))      # ... or the exact one
            if tile and ops._BN_EPI:
                # one (count, mean, M2) triple per pixel tile and channel: tiles of 8 x 16 or 6 x 20 pixels, bounded by 6 x 16-pixel ones
                cap = N * ((OH + 5) // 6) * ((OW + 15) // 16) * c.Cout * 3
                if c.Cout % 4 == 0 and 256 % (c.Cout // 4) == 0:       # split-K grids: one triple per block of the reduce launch (fp_splitk_reduce_stats_launch)
                    rows = 256 // (c.Cout // 4) * 4
                    cap = max(cap, min(512, (N * OH * OW + rows - 1) // rows) * c.Cout * 3)
                bn.stats_part = self.buf(part, (max(cap, 1),))      # (`part`: the shortcut branch runs beside conv1 on another stream: its own buffer)
                bo = ops.bn_stats_out(bn.stats_part)
        y = self._cv(d, x, c.wp, c.wp3, out, hp=(c.hp_f, c.wslot), bn_out=bo)
        if bo is not None:
            bn.stats_nblk = bo.nblk
        return y

    @staticmethod
    def _phase_ok(h, w):
        """phase kernels tile the low-res grid 8 x 16: keep the fused-gather path when most of a tile would be padding"""
        return h * w * 10 >= ((h + 7) // 8 * 8) * ((w + 15) // 16 * 16) * 6

    def _conv_dec(self, c, x0, x1, N, H, W, C0, C1, up2, out):
        if up2 and c.up2 is not None and self._phase_ok(H // 2, W // 2):
            if C1:      # skip half at full resolution (raw partial sums), then the four phases of the upsampled half on top
                d = ops.make_desc(N, H, W, H, W, C1, 0, c.Cout, 3, 1, 1, L.GATHER_FWD_REFLECT)
                self._cv(d, x1, c.wsk, c.wsk3, out, hp=(c.hp_sk, c.wslot), publish=False)   # raw partial sums: not the tensor's amax
            if c.hp_ph is not None and _HP_TILE:
                so = self.amax.out_slot(out)
                ops.conv_up2_phase_fwd_hp(x0, c.hp_ph, c.b.data, out, self.amax.get(x0), c.wslot, amax_out=so, act=L.ACT_ELU,
                                          addend=out if C1 else None)
                self.amax.published(out, so)
                return out
            phase_fwd = ops.conv_up2_phase_fwd_bf3 if c.wph3 is not None else ops.conv_up2_phase_fwd
            return phase_fwd(x0, c.wph3 if c.wph3 is not None else self._need32(c.wph), c.b.data, out, act=L.ACT_ELU, addend=out if C1 else None)
        gather = L.GATHER_FWD_REFLECT_UP2 if up2 else L.GATHER_FWD_REFLECT
        d = ops.make_desc(N, H, W, H, W, C0, C1, c.Cout, 3, 1, 1, gather, act=L.ACT_ELU)
        if x1 is None and not up2:
            return self._cv(d, x0, c.wp, c.wp3, out, hp=(c.hp_f, c.wslot), bias=c.b.data)
        if up2 and c.wp3 is not None and ops.conv3x3_bf3_supported(d):      # concat gather inside the split-operand tile kernel
            if c.hp_f is not None and not ops._bf16x2 and _HP_TILE:
                return self._cv_hp(d, x0, c.hp_f, c.wslot, out, src1=x1, bias=c.b.data)
            return ops.conv3x3_bf3(d, x0, c.wp3, out, bias=c.b.data, src1=x1)
        return ops.conv_igemm(d, x0, x1, self._need32(c.wp), out, bias=c.b.data)

    # ------------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------------
    def forward(self, image, training=True, save_for_backward=True, outputs=None, scales=None):
        """scales: indices into ('1/8','1/4','1/2','1/1') whose heads are evaluated (inference only; None = all four)"""
        if scales is not None and save_for_backward:
            raise ValueError("a subset of output scales is an inference-only option")
        with ops.on_stream(torch.cuda.current_stream()):      # caches the raw stream handle for the launch wrappers
            ops._bf16x2 = bool(self.inference_bf16x2 and not training and not save_for_backward)
            try:
                return self._forward(image, training, save_for_backward, outputs, scales)
            finally:
                ops._bf16x2 = False

    def _forward(self, image, training, save_for_backward, outputs, scales=None):
        if image.dim() != 4 or image.shape[1] != 3:
            raise ValueError("expected image [B,3,H,W]")
        N, _, H, W = image.shape
        if H % 32 or W % 32 or H < 64 or W < 64:
            raise ValueError("H and W must be multiples of 32 and >= 64 (5 stride-2 stages + ReflectionPad2d(1))")
        if not self.params_alias_flat():
            self._flatten()
            self.weights_dirty = True
        self.refresh_packed(overlap=True)
        if _HP:
            self.amax.begin()
        image = image.contiguous().float()
        S = {"N": N, "H": H, "W": W, "image": image, "training": training,
             "scales": set(range(4)) if scales is None else set(scales)}
        buf = self.buf
        if training:
            self._fold_ready = False          # running statistics are about to change
        folded = (not training) and self.fold_eval and not save_for_backward
        if folded:
            self._encoder_eval_folded(image, N, H, W, S)
            return self._decoders_forward(S, outputs, save_for_backward)
        # ---- encoder --------------------------------------------------------------------------------
        h, w = H // 2, W // 2
        z0 = buf("z0", (N, h, w, 64))
        bo = None
        if training and ops._BN_EPI:            # the stem's tile kernel writes the Welford partials of its output (one per 8 x 16 pixel tile and channel)
            self.bn0.stats_nblk = 0
            self.bn0.stats_part = buf("bn.part0", (N * ((h + 7) // 8) * ((w + 15) // 16) * 64 * 3,))
            bo = ops.bn_stats_out(self.bn0.stats_part)
        d0 = ops.make_desc(N, h, w, H, W, 3, 0, 64, 7, 2, 3, L.GATHER_STEM)
        if self.stem.hp_f is not None and not ops._bf16x2 and ops.conv_stem_hp_supported(d0):
            ops.conv_stem_hp(d0, image, self.stem.hp_f, z0, self.stem.wslot, bn_out=bo)
        else:
            ops.conv_igemm(d0, image, None, self.stem.wp, z0, bn_out=bo)
        if bo is not None:
            self.bn0.stats_nblk = bo.nblk
        f0 = self._bn(self.bn0, z0, buf("f0", (N, h, w, 64)), training)
        hp, wp_ = (h + 1) // 2, (w + 1) // 2
        pool = buf("pool", (N, hp, wp_, 64))
        am = buf("pool.argmax", (N, hp, wp_, 64), torch.uint8)
        so = self.amax.out_slot(pool) if _HP else None
        ops.maxpool_fwd(f0, pool, am, amax_out=so)
        if so is not None:
            self.amax.published(pool, so)
        feats = [f0]
        dims = [(h, w)]
        x, h, w = pool, hp, wp_
        S["blocks"] = []
        for i, blk in enumerate(self.blocks):
            s = blk.stride
            if s == 2:
                self._wait_pack()
            oh, ow = (h - 1) // s + 1, (w - 1) // s + 1
            zd, idt, ev_idt = None, x, None

            def shortcut():
                zd_ = self._conv_enc(blk.ds, x, N, h, w, buf("b%d.zd" % i, (N, oh, ow, blk.Cout)), bn=blk.bnd if training else None, part="bn.part.ds")
                return zd_, self._bn(blk.bnd, zd_, buf("b%d.idt" % i, (N, oh, ow, blk.Cout)), training, relu=False)
            if blk.ds is not None and self.concurrent and _DS_AUX:       # downsample branch beside conv1 / conv2 (joined before the residual add)
                if _HP_IGEMM:
                    self.amax.get(x, any_stream=True)                    # published on this stream; the aux stream is ordered behind it below
                ops.event_wait(self.aux, self._record(ops.current_stream()))
                with ops.on_stream(self.aux):
                    zd, idt = shortcut()
                    ev_idt = self._record(self.aux)
            z1 = self._conv_enc(blk.c1, x, N, h, w, buf("b%d.z1" % i, (N, oh, ow, blk.Cout)), bn=blk.bn1 if training else None)
            a1 = self._bn(blk.bn1, z1, buf("b%d.a1" % i, (N, oh, ow, blk.Cout)), training)
            z2 = self._conv_enc(blk.c2, a1, N, oh, ow, buf("b%d.z2" % i, (N, oh, ow, blk.Cout)), bn=blk.bn2 if training else None)
            self._bn_coeffs(blk.bn2, z2, training)              # the statistics do not need the shortcut branch: before the join
            if blk.ds is not None and ev_idt is None:
                zd, idt = shortcut()
            if ev_idt is not None:
                ops.event_wait(ops.current_stream(), ev_idt)
            ob = buf("b%d.out" % i, (N, oh, ow, blk.Cout))
            out = self._bn_apply(blk.bn2, z2, ob, residual=idt)
            S["blocks"].append(dict(x=x, z1=z1, a1=a1, z2=z2, zd=zd, out=out, hin=h, win=w, h=oh, w=ow))
            x, h, w = out, oh, ow
            last_of_layer = (i + 1 == len(self.blocks)) or (self.blocks[i + 1].stride == 2)
            if last_of_layer:
                feats.append(out)
                dims.append((h, w))
        S["feats"], S["dims"] = feats, dims
        return self._decoders_forward(S, outputs, save_for_backward)

    def _decoders_forward(self, S, outputs, save_for_backward):
        N, H, W = S["N"], S["H"], S["W"]
        if outputs is None:
            if self.decoders[0].seg:      # 1-channel heads at their own resolution; channel 1 of each buffer is the padding filter's zero
                outputs = [torch.empty((N, 2, H // s, W // s), device=self.device) for s in (8, 4, 2, 1)]
            else:
                outputs = [torch.empty((N, 4, H, W), device=self.device) for _ in range(4)]
        S["dec"] = [{} for _ in self.decoders]
        main = ops.current_stream()
        self._launch_pack_dgrad()
        if _HP:
            for f in S["feats"]:                     # both decoders read the features: one reduction each, before the streams fork
                self.amax.get(f, any_stream=True)
        if self.concurrent and len(self.decoders) == 2:
            ops.event_wait(self.aux, self._record(main))                 # encoder features ready
            self._interleave([(self.aux, self._decoder_forward(self.decoders[1], S, outputs, S["dec"][1])),
                              (main, self._decoder_forward(self.decoders[0], S, outputs, S["dec"][0]))])
            ops.stream_wait_stream(main, self.aux)                              # join: both decoders wrote their output channels
        else:
            for di, dec in enumerate(self.decoders):
                for _ in self._decoder_forward(dec, S, outputs, S["dec"][di]):
                    pass
        # the repack of the data-gradient layouts (refresh_packed) is the one piece of side-stream work a forward pass can leave behind: join
        # it here, not only where the backward starts -- a caller that drops the model after a forward (inference, tests) frees the packed
        # buffer on THIS stream's timeline, and the allocator may hand the memory to the next model while the repack still writes into it
        # (seen once as a 1e-1 error in the next test's first decoder output).  The repack finished milliseconds ago: the wait costs a packet.
        self._wait_pack_dgrad()
        self.saved = S if save_for_backward else None
        return outputs

    def _psp_forward(self, dec, f4, N, h, w, D):
        """[x, x6, x4, x2, x1] (segmentation/network.py:198-207) into one [N,h,w,1024] buffer; keeps the pooled maps for the backward"""
        buf = self.buf
        cat = buf(dec.name + ".psp.cat", (N, h, w, 1024))
        ops.copy_channels(f4, cat, 512)
        D["psp"] = []
        for P, c, off in dec.psp.blocks:
            pooled = ops.adaptive_avgpool_fwd(f4, buf("%s.psp.pool%d" % (dec.name, P), (N, P, P, 512)))
            d = ops.make_desc(N, P, P, P, P, 512, 0, 128, 1, 1, 0, L.GATHER_FWD_ZERO)
            red = ops.conv_igemm(d, pooled, None, self._need32(c.wp), buf("%s.psp.red%d" % (dec.name, P), (N, P, P, 128)))
            ops.bilinear_ac_fwd(red, cat, off)
            D["psp"].append(pooled)
        return cat

    def _psp_backward(self, dec, D, dcat, dF4, N, h, w, acc, accum_feat):
        """dcat [N,h,w,1024] = gradient of the concatenation -> dF4 (+)= identity slice + the four pooled branches; reduce-conv gradients"""
        buf = self.buf
        ops.copy_channels(dcat, dF4, 512, accumulate=accum_feat)
        for (P, c, off), pooled in zip(dec.psp.blocks, D["psp"]):
            dred = ops.bilinear_ac_bwd(dcat, buf("g.%s.psp.dred%d" % (dec.name, P), (N, P, P, 128)), off)
            dw = ops.make_desc(N, P, P, P, P, 512, 0, 128, 1, 1, 0, L.GATHER_FWD_ZERO)
            ops.conv_wgrad(dw, pooled, None, dred, c.gw, accumulate=acc)
            dd = ops.make_desc(N, P, P, P, P, 128, 0, 512, 1, 1, 0, L.GATHER_DGRAD_ZERO)
            dpool = ops.conv_igemm(dd, dred, None, self._need32(c.wpd), buf("g.%s.psp.dpool%d" % (dec.name, P), (N, P, P, 512)))
            ops.adaptive_avgpool_bwd(dpool, dF4, accumulate=True)

    def _decoder_forward(self, dec, S, outputs, D):
        """generator (see _interleave): fills D with the activations the backward pass needs"""
        N, feats, dims = S["N"], S["feats"], S["dims"]
        buf = self.buf
        D.update({"y": [], "x": [], "low": []})
        x = feats[4]
        h, w = dims[4]
        if dec.psp is not None:
            x = self._psp_forward(dec, x, N, h, w, D)
        D["x0"] = x
        for bi, (blk, (cin, cout)) in enumerate(zip(dec.blocks, dec.chans)):
            if bi:
                yield
            tag = "%s.b%d." % (dec.name, bi + 1)
            skip = feats[3 - bi]
            y1 = self._conv_dec(blk["pre1"], x, None, N, h, w, cin, 0, False, buf(tag + "y1", (N, h, w, cout)))
            y2 = self._conv_dec(blk["pre2"], y1, None, N, h, w, cout, 0, False, buf(tag + "y2", (N, h, w, cout)))
            h, w = 2 * h, 2 * w
            assert (h, w) == dims[3 - bi] and skip.shape[3] == cout
            y3 = self._conv_dec(blk["post1"], y2, skip, N, h, w, cout, cout, True, buf(tag + "y3", (N, h, w, cout)))
            xo = self._conv_dec(blk["post2"], y3, None, N, h, w, cout, 0, False, buf(tag + "x", (N, h, w, cout)))
            D["y"].append((y1, y2, y3))
            D["x"].append(xo)
            x = xo
            if bi >= 1 and (bi - 1) in S.get("scales", _ALL_SCALES):     # heads on block2/3/4 outputs: scales 8, 4, 2
                scale = dec.head_scales[bi - 1]
                low = buf("%s.low%d" % (dec.name, bi), (N, h, w, 2))
                hw_, hb_ = self._head_wb(dec.heads[bi - 1])
                ops.head_fwd(x, hw_, hb_, low, dec.sig)
                ops.head_upsample(low, outputs[bi - 1], scale, dec.c0)
                D["low"].append(low)
        # outconv4: nearest x2 (virtual) -> ConvBlock(64->32) -> head, scale 1
        yield
        h, w = 2 * h, 2 * w
        y51 = self._conv_dec(dec.o41, x, None, N, h, w, 64, 0, True, buf(dec.name + ".y51", (N, h, w, 32)))
        x5 = self._conv_dec(dec.o42, y51, None, N, h, w, 32, 0, False, buf(dec.name + ".x5", (N, h, w, 32)))
        if 3 in S.get("scales", _ALL_SCALES):
            low = buf(dec.name + ".low4", (N, h, w, 2))
            hw_, hb_ = self._head_wb(dec.heads[3])
            ops.head_fwd(x5, hw_, hb_, low, dec.sig)
            ops.head_upsample(low, outputs[3], 1, dec.c0)
            D["low"].append(low)
        D["y51"], D["x5"] = y51, x5

    # ------------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------------
    def _wgrad(self, c, gather, src0, src1, dz, N, OH, OW, IH, IW, C0, C1, acc, side=None, fork=True):
        """Weight (+bias) gradient of one conv.  side = a stream: launch there, ordered after everything already
        queued on the current stream (dz is ready) -- the caller guarantees dz / src stay untouched until the join.
        fork=False: no new event -- the caller has just forked `side` from this stream (the previous _wgrad call) and launched nothing since.
        Every fork is a barrier packet on the launching stream: ~6 us in which the data-gradient chain does not advance (ordered trace,
        profiles/round4_notes.md), so the two weight gradients of a residual block share one."""
        d = ops.make_desc(N, OH, OW, IH, IW, C0, C1, c.Cout, c.K, c.stride, c.pad, gather)
        split = _WBF3 and src1 is None and ops.conv_wgrad_bf3_supported(d)
        # fp16-pair operands: both amax slots are settled on THIS stream (the producers', or a reduction here) before the side stream forks
        am = (self.amax.get(src0), self.amax.get(dz)) if (split and _HP_WGRAD) else None

        def launch():
            if split:
                ops.conv_wgrad_bf3(d, src0, dz, c.gw, 0, accumulate=acc, db=c.gb, amax=am)       # bias gradient from the same pass over dz
                return
            ops.conv_wgrad(d, src0, src1, dz, c.gw, accumulate=acc)
            if c.gb is not None:
                ops.colsum(dz.view(-1, c.Cout), c.gb, accumulate=acc)
        if side is None:
            return launch()
        if fork:
            ops.event_wait(side, self._record(ops.current_stream()))
        with ops.on_stream(side):
            launch()

    def _wgrad_up2(self, c, low, skip, dz, N, hl, wl, C0, C1, acc, side=None, fork=True):
        """weight (+bias) gradient of a conv over cat[nearest_x2(low), skip]: upsampled half by output phase, skip half as a
        channel slice of the same gradient tensor; shapes the phase kernel does not take keep the fused-gather kernel.  fork: see _wgrad."""
        H, W = 2 * hl, 2 * wl
        if c.up2 is None or not ops.up2_phase_wgrad_supported(N, hl, wl, C0, c.Cout):
            d_lo = ops.make_desc(N, H, W, H, W, C0, 0, c.Cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2)
            d_sk = ops.make_desc(N, H, W, H, W, C1, 0, c.Cout, 3, 1, 1, L.GATHER_FWD_REFLECT) if C1 else None
            if not (_WBF3 and ops.conv_wgrad_bf3_supported(d_lo) and (d_sk is None or ops.conv_wgrad_bf3_supported(d_sk))):
                return self._wgrad(c, L.GATHER_FWD_REFLECT_UP2, low, skip, dz, N, H, W, H, W, C0, C1, acc, side, fork=fork)
            s_dz = self.amax.get(dz) if _HP_WGRAD else None
            am_lo = (self.amax.get(low), s_dz) if _HP_WGRAD else None
            am_sk = (self.amax.get(skip), s_dz) if (_HP_WGRAD and d_sk is not None) else None

            def launch_small():      # small images (12x40): both halves through the split-operand kernel, the upsampling folded into its gather
                ops.conv_wgrad_bf3(d_lo, low, dz, c.gw, 0, accumulate=acc, db=c.gb, amax=am_lo)
                if d_sk is not None:
                    ops.conv_wgrad_bf3(d_sk, skip, dz, c.gw, C0, accumulate=acc, amax=am_sk)
            if side is None:
                return launch_small()
            if fork:
                ops.event_wait(side, self._record(ops.current_stream()))
            with ops.on_stream(side):
                return launch_small()

        pbf3 = _WBF3 and _PWBF3
        d_skip = ops.make_desc(N, H, W, H, W, C1, 0, c.Cout, 3, 1, 1, L.GATHER_FWD_REFLECT) if C1 else None
        skip_split = d_skip is not None and _WBF3 and ops.conv_wgrad_bf3_supported(d_skip)
        s_dz = self.amax.get(dz) if (_HP_WGRAD and (pbf3 or skip_split)) else None
        s_low = self.amax.get(low) if (_HP_WGRAD and pbf3) else None
        am_sk = (self.amax.get(skip), s_dz) if (_HP_WGRAD and skip_split) else None

        def launch():
            if s_low is not None:
                ops.conv_up2_phase_wgrad_hp(low, dz, c.gw, s_low, s_dz, 0, accumulate=acc, db=c.gb)               # + bias gradient
            else:
                ops.conv_up2_phase_wgrad(low, dz, c.gw, 0, accumulate=acc, bf3=pbf3, db=c.gb if pbf3 else None)   # + bias gradient
            bias_done = pbf3 or c.gb is None
            if C1:
                d = d_skip
                if skip_split:
                    ops.conv_wgrad_bf3(d, skip, dz, c.gw, C0, accumulate=acc, db=None if bias_done else c.gb, amax=am_sk)
                    bias_done = True
                else:
                    ops.conv_wgrad_slice(d, skip, None, dz, c.gw, C0, accumulate=acc)
            if not bias_done:
                ops.colsum(dz.view(-1, c.Cout), c.gb, accumulate=acc)
        if side is None:
            return launch()
        if fork:
            ops.event_wait(side, self._record(ops.current_stream()))
        with ops.on_stream(side):
            launch()

    def _dgrad_dec(self, c, dz, N, H, W, out, actsrc=None, addend=None, accum=False):
        epi = (L.EPI_ACTGRAD_ELU if actsrc is not None else 0) | (L.EPI_ACCUM if accum else 0)
        d = ops.make_desc(N, H, W, H, W, c.Cout, 0, c.Cin, 3, 1, 1, L.GATHER_DGRAD_REFLECT, epi=epi)
        return self._cv(d, dz, c.wpd, c.wpd3, out, hp=(c.hp_d, c.wslot), actsrc=actsrc, addend=addend)

    def _dgrad_up2_ext(self, c, dz, N, hl, wl, C0, pfx):
        """gradient wrt the low-res input of an upsample conv on the (hl+2) x (wl+2) extended grid (ops.up2_fold_bwd folds it)"""
        ext = self.buf(pfx + "XV", (N, hl + 2, wl + 2, C0))
        if c.wdu3 is not None and self._phase_ok(hl + 2, wl + 2):      # split-operand phase kernel (8x16 tiles of the extended grid)
            if c.hp_du is not None and _HP_TILE:
                return ops.conv_up2_phase_dgrad_hp(dz, c.hp_du, ext, self.amax.get(dz), c.wslot)
            return ops.conv_up2_phase_dgrad_bf3(dz, c.wdu3, ext)
        d = ops.make_desc(N, hl + 2, wl + 2, 2 * hl, 2 * wl, c.Cout, 0, C0, 4, 2, 3, L.GATHER_FWD_ZERO)
        return ops.conv_igemm(d, dz, None, self._need32(c.wdu), ext)

    def stage_streams(self):
        """every stream that may still be writing parameter gradients when `on_stage` fires (None: everything is on the caller's stream)"""
        if not self.concurrent:
            return None
        return [ops.current_stream(), self.aux, self.wg, self.dwg[0], self.dwg[1]]

    def backward(self, grad_outputs, accumulate=False, on_stage=None):
        with ops.on_stream(torch.cuda.current_stream()):
            return self._backward(grad_outputs, accumulate, on_stage)

    def _backward(self, grad_outputs, accumulate, on_stage):
        """grad_outputs: 4 tensors [B,4,H,W] (d loss / d outputs['1/8','1/4','1/2','1/1']).
        Writes every live parameter gradient into self.flat_grad (views: self.grad_views).
        on_stage(name) is called as soon as all gradients of a flat-buffer stage have been LAUNCHED -- on this stream and on the weight-gradient
        side streams (`stage_streams()`): the caller orders its collective after all of them; the main stream itself does not wait for the
        side streams here (that join cost 2.6 ms per step: the weight gradients overlap the rest of the backward pass).  Stages: ('mask_decoder', 'depth_decoder',
        'encoder.layer4' .. 'encoder.layer0') have been launched -- the data-parallel reducer hooks in here."""
        S = self.saved
        if S is None:
            raise RuntimeError("backward called without a saved forward")
        if not S["training"]:
            raise RuntimeError("backward through eval-mode BatchNorm is not supported by the HIP engine")
        if not all(p.requires_grad for p in self.live_params):
            frozen = [n for n, p in zip(self.live_names, self.live_params) if not p.requires_grad]
            raise RuntimeError("footprints_amd: frozen parameters (requires_grad=False) are not supported by the fused backward / "
                               "Adam -- they would be trained silently: %s ..." % frozen[:3])
        N = S["N"]
        feats, dims = S["feats"], S["dims"]
        buf = self.buf
        self.amax.globalize()            # every stream of the forward pass was joined: its slots are valid for any backward stream
        gouts = [g.contiguous() for g in grad_outputs]
        dF = [buf("dF%d" % i, tuple(f.shape)) for i, f in enumerate(feats)]
        main = ops.current_stream()
        self._decoders_backward(S, gouts, dF, accumulate, on_stage)
        side = self.wg if self.concurrent else None
        # ---- encoder ----------------------------------------------------------------------------------
        nblk = len(self.blocks)
        feat_of_block = {}
        fi = 1
        for i in range(nblk):
            if (i + 1 == nblk) or (self.blocks[i + 1].stride == 2):
                feat_of_block[i] = fi
                fi += 1
        dnext = None       # gradient wrt this block's output coming from the next block (same layer)
        dnext_part = None  # (partials, blocks) when `dnext` already IS g = dout * (out > 0) and its producer wrote bn2's backward sums
        # (round 5: also with the exact bf16x3 operands -- the sink is a property of the tile kernel's epilogue, not of the operand format)
        bwd_epi = ops._BN_BWD_EPI and (_HP_TILE or _BF3) and not ops._bf16x2

        def bnb_arm(name, h_, w_, C_, z, rec):
            """the BatchNorm-backward side output (ops.BnOut) for a tile data gradient at (h_, w_, C_): partial sums per pixel tile (8 x 16 or 6 x 20
            pixels, bounded by 6 x 16-pixel ones) and channel"""
            cap = N * ((h_ + 5) // 6) * ((w_ + 15) // 16) * C_ * 2
            if C_ % 4 == 0 and 256 % (C_ // 4) == 0:              # split-K grids: one pair per block of the reduce launch (fp_splitk_reduce_bnb_launch)
                rows = 256 // (C_ // 4) * 4
                cap = max(cap, min(512, (N * h_ * w_ + rows - 1) // rows) * C_ * 2)
            part = buf(name, (max(cap, 1),))
            return part, ops.bn_bwd_out(part, z.view(-1, C_), rec.mean, rec.invstd)
        for i in range(nblk - 1, -1, -1):
            blk, B = self.blocks[i], S["blocks"][i]
            h, w, hin, win = B["h"], B["w"], B["hin"], B["win"]
            C, Cin = blk.Cout, blk.Cin
            M = N * h * w
            dout = dF[feat_of_block[i]] if i in feat_of_block else dnext
            dz2 = buf("g.dz2.%d" % i, (N, h, w, C))      # per-block: read later by the side-stream wgrad
            if i not in feat_of_block and dnext_part is not None:
                # the next block's conv1 data gradient stored g = (its sum + the residual gradient) * (out > 0) and the two per-channel sums
                # of this BatchNorm's backward with it: no reduction pass, no separate g
                g = dout
                ops.bn_bwd_partials(g.view(M, C), B["z2"].view(M, C), blk.bn2.mean, blk.bn2.invstd, blk.bn2.bn.weight.data, dz2.view(M, C),
                                    blk.bn2.gg, blk.bn2.gb, dnext_part[0], dnext_part[1], accumulate=accumulate, amax_out=self._sink_slot(dz2))
            else:
                g = buf("g.g", (N, h, w, C))
                ops.bn_bwd(dout.view(M, C), B["out"].view(M, C), B["z2"].view(M, C), blk.bn2.mean, blk.bn2.invstd, blk.bn2.bn.weight.data,
                           dz2.view(M, C), blk.bn2.gg, blk.bn2.gb, g_out=g.view(M, C), accumulate=accumulate, amax_out=self._sink_slot(dz2))
            dnext_part = None
            self._sink_done(dz2)
            ev_ds = None
            if blk.ds is not None and self.concurrent and _DS_AUX:
                # downsample branch on the aux stream: BN backward, weight gradient, and its data gradient FIRST into the previous layer's
                # feature gradient; the main branch's conv1 data gradient accumulates on top after the join (fixed order)
                tgt = dF[feat_of_block[i - 1]]
                ops.event_wait(self.aux, self._record(ops.current_stream()))
                with ops.on_stream(self.aux):
                    dzd = buf("g.dzd.%d" % i, (N, h, w, C))
                    ops.bn_bwd(g.view(M, C), None, B["zd"].view(M, C), blk.bnd.mean, blk.bnd.invstd, blk.bnd.bn.weight.data,
                               dzd.view(M, C), blk.bnd.gg, blk.bnd.gb, accumulate=accumulate,
                               amax_out=self._sink_slot(dzd) if blk.ds.hp_ig else None)
                    if blk.ds.hp_ig:
                        self._sink_done(dzd)
                    self._wgrad(blk.ds, L.GATHER_FWD_ZERO, B["x"], None, dzd, N, h, w, hin, win, Cin, 0, accumulate, side)
                    d1 = ops.make_desc(N, hin, win, h, w, C, 0, Cin, 1, blk.stride, 0, L.GATHER_DGRAD_ZERO, epi=L.EPI_ACCUM)
                    self._cv(d1, dzd, blk.ds.wpd, blk.ds.wpd3, tgt, hp=(blk.ds.hp_d, blk.ds.wslot))
                    ev_ds = self._record(self.aux)
            if not _WGRAD_PAIR_FORK:
                self._wgrad(blk.c2, L.GATHER_FWD_ZERO, B["a1"], None, dz2, N, h, w, h, w, C, 0, accumulate, side)
            da1 = buf("g.da1", (N, h, w, C))
            dz1 = buf("g.dz1.%d" % i, (N, h, w, C))
            d2 = ops.make_desc(N, h, w, h, w, C, 0, C, 3, 1, 1, L.GATHER_DGRAD_ZERO)
            if bwd_epi and (blk.c2.hp_d is not None or blk.c2.wpd3 is not None) and ops.conv3x3_bf3_supported(d2):
                # conv2's data gradient applies bn1's ReLU mask itself (da1 = gradient * (a1 > 0)) and emits bn1's backward sums
                d2.epi = L.EPI_ACTGRAD_RELU
                part, bo = bnb_arm("bnb.part1", h, w, C, B["z1"], blk.bn1)
                self._cv(d2, dz2, blk.c2.wpd, blk.c2.wpd3, da1, hp=(blk.c2.hp_d, blk.c2.wslot), actsrc=B["a1"], bn_out=bo)
                if bo.nblk > 0:
                    ops.bn_bwd_partials(da1.view(M, C), B["z1"].view(M, C), blk.bn1.mean, blk.bn1.invstd, blk.bn1.bn.weight.data,
                                        dz1.view(M, C), blk.bn1.gg, blk.bn1.gb, part, bo.nblk, accumulate=accumulate,
                                        amax_out=self._sink_slot(dz1))
                else:                                   # split grid: the mask is applied, the sums are not there
                    ops.bn_bwd(da1.view(M, C), None, B["z1"].view(M, C), blk.bn1.mean, blk.bn1.invstd, blk.bn1.bn.weight.data,
                               dz1.view(M, C), blk.bn1.gg, blk.bn1.gb, accumulate=accumulate, amax_out=self._sink_slot(dz1))
            else:
                self._cv(d2, dz2, blk.c2.wpd, blk.c2.wpd3, da1, hp=(blk.c2.hp_d, blk.c2.wslot))
                ops.bn_bwd(da1.view(M, C), B["a1"].view(M, C), B["z1"].view(M, C), blk.bn1.mean, blk.bn1.invstd, blk.bn1.bn.weight.data,
                           dz1.view(M, C), blk.bn1.gg, blk.bn1.gb, accumulate=accumulate, amax_out=self._sink_slot(dz1))
            self._sink_done(dz1)
            if _WGRAD_PAIR_FORK:                         # both weight gradients of the block behind ONE fork (dz2 has its own buffer per block)
                if _HP_WGRAD and side is not None:       # a missing amax slot is reduced on THIS stream: before the fork, not between the two launches
                    self.amax.get(B["x"])
                    self.amax.get(dz1)
                self._wgrad(blk.c2, L.GATHER_FWD_ZERO, B["a1"], None, dz2, N, h, w, h, w, C, 0, accumulate, side)
            self._wgrad(blk.c1, L.GATHER_FWD_ZERO, B["x"], None, dz1, N, h, w, hin, win, Cin, 0, accumulate, side,
                        fork=not (_WGRAD_PAIR_FORK and side is not None))
            first_of_layer = (i == 0) or blk.stride == 2
            dgd = ops.make_desc(N, hin, win, h, w, C, 0, Cin, 3, blk.stride, 1, L.GATHER_DGRAD_ZERO)
            if blk.ds is not None and ev_ds is not None:
                ops.event_wait(ops.current_stream(), ev_ds)
                dgd.epi = L.EPI_ACCUM
                self._cv(dgd, dz1, blk.c1.wpd, blk.c1.wpd3, dF[feat_of_block[i - 1]], hp=(blk.c1.hp_d, blk.c1.wslot))
                dnext = None
            elif blk.ds is not None:
                dzd = buf("g.dzd.%d" % i, (N, h, w, C))
                ops.bn_bwd(g.view(M, C), None, B["zd"].view(M, C), blk.bnd.mean, blk.bnd.invstd, blk.bnd.bn.weight.data,
                           dzd.view(M, C), blk.bnd.gg, blk.bnd.gb, accumulate=accumulate,
                           amax_out=self._sink_slot(dzd) if blk.ds.hp_ig else None)
                if blk.ds.hp_ig:
                    self._sink_done(dzd)
                self._wgrad(blk.ds, L.GATHER_FWD_ZERO, B["x"], None, dzd, N, h, w, hin, win, Cin, 0, accumulate, side)
                tgt = dF[feat_of_block[i - 1]]          # block input is the previous layer's feature (already holds decoder grads)
                dgd.epi = L.EPI_ACCUM
                self._cv(dgd, dz1, blk.c1.wpd, blk.c1.wpd3, tgt, hp=(blk.c1.hp_d, blk.c1.wslot))
                d1 = ops.make_desc(N, hin, win, h, w, C, 0, Cin, 1, blk.stride, 0, L.GATHER_DGRAD_ZERO, epi=L.EPI_ACCUM)
                self._cv(d1, dzd, blk.ds.wpd, blk.ds.wpd3, tgt, hp=(blk.ds.hp_d, blk.ds.wslot))
                dnext = None
            elif first_of_layer:                        # layer1 block 0: input is the max-pool output
                dpool = buf("g.dpool", (N, hin, win, Cin))
                self._cv(dgd, dz1, blk.c1.wpd, blk.c1.wpd3, dpool, hp=(blk.c1.hp_d, blk.c1.wslot), addend=g)
                ops.maxpool_bwd(dpool, self._bufs["pool.argmax"][:dpool.numel()].view(dpool.shape), dF[0], accumulate=True)
                dnext = None
            else:
                dx = buf("g.dx%d" % (i & 1), (N, hin, win, Cin))
                if bwd_epi and (blk.c1.hp_d is not None or blk.c1.wpd3 is not None) and ops.conv3x3_bf3_supported(dgd):
                    # this block's input is the previous block's output (after its ReLU): conv1's data gradient + the residual gradient,
                    # masked by (input > 0), IS the g of the previous block's bn2 -- stored as such, with that BatchNorm's backward sums
                    Bp, blkp = S["blocks"][i - 1], self.blocks[i - 1]
                    dgd.epi = L.EPI_ACTGRAD_RELU
                    part, bo = bnb_arm("bnb.part2", hin, win, Cin, Bp["z2"], blkp.bn2)
                    self._cv(dgd, dz1, blk.c1.wpd, blk.c1.wpd3, dx, hp=(blk.c1.hp_d, blk.c1.wslot), addend=g, actsrc=B["x"], bn_out=bo)
                    if bo.nblk > 0:
                        dnext_part = (part, bo.nblk)
                    else:                               # masked, no sums: the regular backward below must not mask again -- it would not
                        dnext_part = None               # change anything (g * (out > 0) is idempotent), so it simply runs as before
                else:
                    self._cv(dgd, dz1, blk.c1.wpd, blk.c1.wpd3, dx, hp=(blk.c1.hp_d, blk.c1.wslot), addend=g)
                dnext = dx
            if self.debug_hook is not None:
                self.debug_hook(i, dict(dout=dout, g=g, dz2=dz2, da1=da1, dz1=dz1, dnext=dnext, B=B))
            if first_of_layer and on_stage is not None:       # this layer's weight gradients live on the side stream: see stage_streams()
                on_stage("encoder.layer%d" % (feat_of_block[min(k for k in feat_of_block if k >= i)]))
        # ---- stem ---------------------------------------------------------------------------------------
        h0, w0 = dims[0]
        M0 = N * h0 * w0
        z0 = self._bufs["z0"][:M0 * 64].view(M0, 64)
        dz0 = buf("g.dz0", (N, h0, w0, 64))
        d = ops.make_desc(N, h0, w0, S["H"], S["W"], 3, 0, 64, 7, 2, 3, L.GATHER_STEM)
        stem_hp = self.stem.hp_f is not None and ops.conv_stem_hp_supported(d)         # fp16-pair weight gradient: dz0's amax out of bn_bwd
        ops.bn_bwd(dF[0].view(M0, 64), feats[0].view(M0, 64), z0, self.bn0.mean, self.bn0.invstd, self.bn0.bn.weight.data,
                   dz0.view(M0, 64), self.bn0.gg, self.bn0.gb, accumulate=accumulate, amax_out=self._sink_slot(dz0) if stem_hp else None)
        if stem_hp:
            self._sink_done(dz0)
            ops.conv_stem_wgrad_hp(d, S["image"], dz0, self.stem.gw, self.amax.get(dz0), accumulate=accumulate)
        else:
            ops.conv_wgrad(d, S["image"], None, dz0, self.stem.gw, accumulate=accumulate)
        if side is not None:
            ops.stream_wait_stream(main, side)                # join: every weight gradient is complete
            ops.stream_wait_stream(main, self.dwg[0])
            ops.stream_wait_stream(main, self.dwg[1])
        if on_stage is not None:
            on_stage("encoder.layer0")

    def _decoders_backward(self, S, gouts, dF, accumulate, on_stage, join=False):
        """both decoders, from d loss / d outputs to the feature gradients dF[0..4] plus every decoder weight gradient"""
        main = ops.current_stream()
        self._wait_pack_dgrad()
        if self.concurrent and len(self.decoders) == 2:
            # mask decoder on the main stream, depth decoder on the aux stream; the depth decoder ACCUMULATES into the
            # feature gradients, so each of its accumulate launches waits for the mask decoder's write of that level
            self._ev_dF = [None] * 5
            ops.event_wait(self.aux, self._record(main))
            self._interleave([(main, self._decoder_backward(self.decoders[0], S["dec"][0], S, gouts, dF, first=True, acc=accumulate)),
                              (self.aux, self._decoder_backward(self.decoders[1], S["dec"][1], S, gouts, dF, first=False, acc=accumulate))])
            ops.stream_wait_stream(main, self.aux)
            if join:                                 # measurement hook: the caller times the decoders' backward including their weight gradients
                ops.stream_wait_stream(main, self.dwg[0])
                ops.stream_wait_stream(main, self.dwg[1])
            if on_stage is not None:
                on_stage(self.decoders[0].name)
                on_stage(self.decoders[1].name)
        else:
            for di, dec in enumerate(self.decoders):
                for _ in self._decoder_backward(dec, S["dec"][di], S, gouts, dF, first=(di == 0), acc=accumulate):
                    pass
                if self.concurrent and join:
                    ops.stream_wait_stream(main, self.dwg[0 if di == 0 else 1])
                if on_stage is not None:
                    on_stage(dec.name)

    def decoders_backward(self, grad_outputs):
        """Measurement hook: the decoder backward alone -- both decoders, from d loss / d outputs to
        d loss / d features plus all decoder weight gradients -- on the saved forward; every side stream is joined before it
        returns to the caller's stream.  Idempotent: can be repeated on one saved forward."""
        with ops.on_stream(torch.cuda.current_stream()):
            S = self.saved
            if S is None or not S["training"]:
                raise RuntimeError("decoders_backward needs a saved training-mode forward")
            self.amax.globalize()
            gouts = [g.contiguous() for g in grad_outputs]
            dF = [self.buf("dF%d" % i, tuple(f.shape)) for i, f in enumerate(S["feats"])]
            self._decoders_backward(S, gouts, dF, False, None, join=True)
            return dF

    @staticmethod
    def _interleave(jobs):
        """Round-robin the sections of several (stream, generator) launch sequences.  The host runs ahead of the GPU until the
        stream's queue back-pressures it, so a sequence issued only after another one has been issued completely does not start
        until that one has almost drained: issue order has to alternate for the streams to overlap on the GPU."""
        jobs = list(jobs)
        while jobs:
            for job in list(jobs):
                stream, gen = job
                with ops.on_stream(stream):
                    try:
                        next(gen)
                    except StopIteration:
                        jobs.remove(job)

    def _decoder_backward(self, dec, D, S, gouts, dF, first, acc):
        """generator: yields between sections (see _interleave); must be resumed under the same stream context"""
        N, feats, dims = S["N"], S["feats"], S["dims"]
        buf = self.buf
        pfx = "g.%s." % dec.name                    # per-decoder temporaries: the two decoders may run concurrently
        H, W = S["H"], S["W"]
        accum_feat = not first                      # the second decoder accumulates into the feature gradients
        cur = ops.current_stream()
        # weight gradients go to this decoder's side stream and are only joined at the end of Engine.backward (they overlap
        # the rest of the decoder AND the encoder backward), so every dZ they read lives in its own buffer (never reused).
        side = self.dwg[0 if first else 1] if self.concurrent else None

        def order_dF(k):
            """first decoder: publish its write of dF[k]; second: wait for it before accumulating (fixed order)."""
            if not self.concurrent:
                return
            if first:
                self._ev_dF[k] = self._record(cur)
            elif self._ev_dF[k] is not None:
                ops.event_wait(cur, self._ev_dF[k])
        # ---- full-resolution tail: head4 <- o42 <- o41 <- up2(x4) ------------------------------------
        x4 = D["x"][3]
        h0, w0 = dims[0]
        side_h = side if _HEAD_WGRAD_SIDE else None
        dzl = buf(pfx + "dzl3", (N, H, W, 2))         # (one buffer per scale: the side stream reads it until the end of the backward pass)
        ops.head_upsample_bwd(gouts[3], D["low"][3], dzl, dec.head_scales[3], dec.c0, dec.sig)
        hd = dec.heads[3]
        if side_h is None:
            self._head_wgrad(hd, D["x5"], dzl, acc)
        A = buf(pfx + "dz.o42", (N, H, W, 32))
        ops.head_dgrad(dzl, self._head_wb(hd)[0], A, elu_src=D["x5"], amax_out=self._sink_slot(A))
        self._sink_done(A)
        if side_h is not None:                        # ONE fork for the head's and o42's weight gradients (dzl, x5 and A are all queued by now)
            if _HP_WGRAD:
                self.amax.get(D["y51"]); self.amax.get(A)
            self._head_wgrad(hd, D["x5"], dzl, acc, side_h)
        self._wgrad(dec.o42, L.GATHER_FWD_REFLECT, D["y51"], None, A, N, H, W, H, W, 32, 0, acc, side, fork=side_h is None)
        Bz = self._dgrad_dec(dec.o42, A, N, H, W, buf(pfx + "dz.o41", (N, H, W, 32)), actsrc=D["y51"])
        yield
        self._wgrad_up2(dec.o41, x4, None, Bz, N, h0, w0, 64, 0, acc, side)
        phase41 = dec.o41.up2 is not None
        if phase41:
            XV = self._dgrad_up2_ext(dec.o41, Bz, N, h0, w0, 64, pfx)
        else:
            XV = self._dgrad_dec(dec.o41, Bz, N, H, W, buf(pfx + "XV", (N, H, W, 64)))
        # head3 on x4
        dzl = buf(pfx + "dzl2", (N, h0, w0, 2))
        ops.head_upsample_bwd(gouts[2], D["low"][2], dzl, dec.head_scales[2], dec.c0, dec.sig)
        hd = dec.heads[2]
        self._head_wgrad(hd, x4, dzl, acc, side_h)
        XH = buf(pfx + "XH", (N, h0, w0, 64))
        ops.head_dgrad(dzl, self._head_wb(hd)[0], XH)
        A = buf(pfx + "dz.post2.3", (N, h0, w0, 64))
        if phase41:
            ops.up2_fold_bwd(XV, A, addend=XH, ylow=x4, amax_out=self._sink_slot(A))
            self._sink_done(A)
        else:
            ops.up2cat_bwd(XV, N, h0, w0, 64, 0, A, addend=XH, ylow=x4)
        # ---- blocks 4..1 ----------------------------------------------------------------------------------
        for bi in (3, 2, 1, 0):
            yield
            blk = dec.blocks[bi]
            cin, cout = dec.chans[bi]
            y1, y2, y3 = D["y"][bi]
            hh, ww = dims[3 - bi]                   # resolution of the post-concat convs
            hl, wl = hh // 2, ww // 2               # resolution of the pre-concat convs
            skip = feats[3 - bi]
            xin = D["x"][bi - 1] if bi > 0 else D["x0"]          # block1's input: the 1/32 feature map (or its pyramid-pooling concatenation)
            # A = dZ of post2 at (hh, ww)
            pair = _WGRAD_PAIR_FORK and side is not None      # the two weight gradients of each half of the block behind one fork (see _wgrad)
            if not pair:
                self._wgrad(blk["post2"], L.GATHER_FWD_REFLECT, y3, None, A, N, hh, ww, hh, ww, cout, 0, acc, side)
            Bz = self._dgrad_dec(blk["post2"], A, N, hh, ww, buf(pfx + "dz.post1.%d" % bi, (N, hh, ww, cout)), actsrc=y3)
            if pair:
                if _HP_WGRAD:
                    for t in (Bz, y2, skip):
                        self.amax.get(t)
                self._wgrad(blk["post2"], L.GATHER_FWD_REFLECT, y3, None, A, N, hh, ww, hh, ww, cout, 0, acc, side)
            self._wgrad_up2(blk["post1"], y2, skip, Bz, N, hl, wl, cout, cout, acc, side, fork=not pair)
            A = buf(pfx + "dz.pre2.%d" % bi, (N, hl, wl, cout))
            if blk["post1"].up2 is not None:
                # d(low) = 4x4 stride-2 conv over dZ + border fold (* ELU'); d(skip) straight into the feature gradient
                ops.up2_fold_bwd(self._dgrad_up2_ext(blk["post1"], Bz, N, hl, wl, cout, pfx), A, ylow=y2, amax_out=self._sink_slot(A))
                self._sink_done(A)
                if not first:
                    order_dF(3 - bi)
                ds = ops.make_desc(N, hh, ww, hh, ww, cout, 0, cout, 3, 1, 1, L.GATHER_DGRAD_REFLECT,
                                   epi=L.EPI_ACCUM if accum_feat else 0)
                self._cv(ds, Bz, blk["post1"].wds, blk["post1"].wds3, dF[3 - bi], hp=(blk["post1"].hp_ds, blk["post1"].wslot))
            else:
                XV = self._dgrad_dec(blk["post1"], Bz, N, hh, ww, buf(pfx + "XV", (N, hh, ww, 2 * cout)))
                if not first:
                    order_dF(3 - bi)
                ops.up2cat_bwd(XV, N, hl, wl, cout, cout, A, ylow=y2, dskip=dF[3 - bi], accumulate_skip=accum_feat)
            if first:
                order_dF(3 - bi)
            yield
            if not pair:
                self._wgrad(blk["pre2"], L.GATHER_FWD_REFLECT, y1, None, A, N, hl, wl, hl, wl, cout, 0, acc, side)
            A2 = A
            Bz = self._dgrad_dec(blk["pre2"], A, N, hl, wl, buf(pfx + "dz.pre1.%d" % bi, (N, hl, wl, cout)), actsrc=y1)
            if pair:
                if _HP_WGRAD:
                    self.amax.get(Bz)
                    self.amax.get(xin)
                self._wgrad(blk["pre2"], L.GATHER_FWD_REFLECT, y1, None, A2, N, hl, wl, hl, wl, cout, 0, acc, side)
            self._wgrad(blk["pre1"], L.GATHER_FWD_REFLECT, xin, None, Bz, N, hl, wl, hl, wl, cin, 0, acc, side, fork=not pair)
            if bi == 0:
                if not first:
                    order_dF(4)
                if dec.psp is not None:
                    dcat = self._dgrad_dec(blk["pre1"], Bz, N, hl, wl, buf(pfx + "psp.dcat", (N, hl, wl, cin)))
                    self._psp_backward(dec, D, dcat, dF[4], N, hl, wl, acc, accum_feat)
                else:
                    self._dgrad_dec(blk["pre1"], Bz, N, hl, wl, dF[4], accum=accum_feat)
                if first:
                    order_dF(4)
            else:
                XH = None
                if bi >= 2:                          # heads on block2 / block3 outputs (x2: scale 8, x3: scale 4)
                    k = bi - 2
                    dzl = buf(pfx + "dzl%d" % k, (N, hl, wl, 2))
                    ops.head_upsample_bwd(gouts[k], D["low"][k], dzl, dec.head_scales[k], dec.c0, dec.sig)
                    hd = dec.heads[k]
                    self._head_wgrad(hd, xin, dzl, acc, side_h)
                    XH = buf(pfx + "XH", (N, hl, wl, cin))
                    ops.head_dgrad(dzl, self._head_wb(hd)[0], XH)
                A = self._dgrad_dec(blk["pre1"], Bz, N, hl, wl, buf(pfx + "dz.post2.%d" % (bi - 1), (N, hl, wl, cin)), actsrc=xin, addend=XH)

    # ------------------------------------------------------------------------------------------------
    def bind_grads(self, accumulate_existing=True):
        """Point every live parameter's .grad at its view of the flat gradient buffer."""
        for p, v in zip(self.live_params, self.grad_views):
            p.grad = v


class NetFunction(torch.autograd.Function):
    """One autograd node for the whole network (drop-in `loss.backward()` support)."""

    @staticmethod
    def forward(ctx, eng, image, *params):
        ctx.eng = eng
        outs = eng.forward(image, training=eng.model.training, save_for_backward=True)
        ctx.token = eng.saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        eng = ctx.eng
        if eng.saved is not ctx.token:
            raise RuntimeError("footprints_amd: activations of this forward were overwritten by a later forward; "
                               "call backward before the next forward (the arena is reused every step)")
        shape = next(g.shape for g in gouts if g is not None)
        grads = [torch.zeros(shape, device=eng.device) if g is None else g.contiguous() for g in gouts]
        # .grad semantics of autograd: ACCUMULATE into existing gradients.  Gradients live in the flat buffer, so decide per
        # parameter where its previous gradient is: nowhere (None), already in its flat slot (aliased), or in a foreign tensor.
        state = [0 if p.grad is None else (1 if p.grad.data_ptr() == v.data_ptr() else 2) for p, v in zip(eng.live_params, eng.grad_views)]
        if all(s == 0 for s in state):
            eng.backward(grads, accumulate=False)           # the usual step: zero_grad(set_to_none=True) -> overwrite
        else:
            if not all(s == 1 for s in state):               # mixed: bring every previous gradient into its flat slot first
                for p, v, s in zip(eng.live_params, eng.grad_views, state):
                    if s == 0:
                        v.zero_()
                    elif s == 2:
                        v.copy_(p.grad)
            eng.backward(grads, accumulate=True)
        for p, v in zip(eng.live_params, eng.grad_views):    # hand the flat views to the parameters without a copy
            p.grad = v
        return (None, None) + tuple(None for _ in eng.live_params)
