"""CPU: host-side logic of the product package -- the C-ABI library loads and exports every symbol the header
declares, the nn.Module tree reproduces the reference state_dict, CLI / preprocessing plumbing, and the refusal
of any CPU compute path.  No kernel is launched here (no GPU in this container)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from footprints_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "footprints_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fp_version() >= 1
    assert isinstance(lib.fp_last_error_string(), bytes)


def test_desc_struct_matches_header():
    from footprints_amd._lib import ConvDesc
    hdr = open(os.path.join(ROOT, "include", "footprints_hip.h")).read()
    body = re.search(r"typedef struct fp_conv_desc \{(.*?)\} fp_conv_desc;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields += [f.strip() for f in decl.split(None, 1)[1].split(",")]
    assert fields == [f[0] for f in ConvDesc._fields_]
    import ctypes
    assert ctypes.sizeof(ConvDesc) == 4 * len(fields)


def test_workspace_queries_and_argument_validation_without_gpu():
    """Host-side entry points that do not launch kernels work on a CPU-only box."""
    import ctypes as C
    from footprints_amd import _lib, ops
    lib = _lib.load()
    d = ops.make_desc(2, 8, 8, 8, 8, 16, 0, 8, 3, 1, 1, _lib.GATHER_FWD_REFLECT)
    assert lib.fp_conv_wgrad_workspace(C.byref(d)) > 0
    assert lib.fp_conv_igemm_workspace(C.byref(d)) == 24 * 2 * 8 * 8 * 8 * 4      # small grid: split-K partials
    big = ops.make_desc(12, 96, 320, 96, 320, 64, 0, 64, 3, 1, 1, _lib.GATHER_FWD_REFLECT)
    assert lib.fp_conv_igemm_workspace(C.byref(big)) == 0
    assert lib.fp_packed_weight_elems(64, 3, 7, 7, 0, 1) == 10 * 64 * 16
    assert lib.fp_packed_weight_elems(8, 20, 3, 3, 0, 0) == 9 * 2 * 8 * 16
    assert lib.fp_loss_workspace(12, 192, 640) > 0
    bad = ops.make_desc(2, 8, 8, 8, 8, 6, 0, 8, 3, 1, 1, _lib.GATHER_FWD_REFLECT)     # C0 not a multiple of 4
    rc = lib.fp_conv_igemm(C.byref(bad), 1, 0, 1, 0, 0, 0, 0, 1, None, 0, None, None)
    assert rc == -1 and b"multiples of 4" in lib.fp_last_error_string()
    # round 6: side outputs are an explicit `const fp_aux*` argument; a launch that fails its argument checks reports "nothing emitted"
    n = C.c_int32(7)
    aux = _lib.Aux()
    aux.bn_part, aux.bn_capacity_floats, aux.bn_nblk_out = 1, 1 << 20, C.pointer(n)
    rc = lib.fp_conv_igemm(C.byref(bad), 1, 0, 1, 0, 0, 0, 0, 1, None, 0, C.byref(aux), None)
    assert rc == -1 and n.value == 0


def test_no_hidden_state_in_the_c_abi():
    """VERDICT r5 "Next" 8: the `*_out_next` sinks are gone -- no such symbol in the header, the binding or the library, fp_aux's layout in the
    binding equals the header's field list, and the only thread-local object of api.cpp is the error string"""
    import ctypes as C
    from footprints_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "footprints_hip.h")).read()
    decls = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                       # declarations only: the comments may mention the old names
    assert "_out_next" not in decls and not [n for n in _lib.SIGNATURES if n.endswith("_next")]
    lib = _lib.load()
    for old in ("fp_amax_out_next", "fp_bn_stats_out_next", "fp_bn_bwd_out_next"):
        assert not hasattr(lib, old), old
    body = re.search(r"typedef struct fp_aux \{(.*?)\} fp_aux;", decls, flags=re.S).group(1)
    fields = [re.findall(r"(\w+)\s*;", line)[0] for line in body.splitlines() if ";" in line]
    assert fields == [f[0] for f in _lib.Aux._fields_], (fields, _lib.Aux._fields_)
    assert C.sizeof(_lib.Aux) == 7 * 8
    api = open(os.path.join(ROOT, "footprints_amd", "csrc", "api.cpp")).read()
    assert re.findall(r"thread_local\s+[^;=]*?(\w+)\s*(?:\[|=|;)", api) == ["g_err"]
    n_aux = sum(1 for sig in _lib.SIGNATURES.values() if _lib._AUX in sig[1])
    assert n_aux == 13 and decls.count("const fp_aux* aux") == n_aux


def test_hp_host_side_contract_without_gpu():
    """fp16-pair operand format: header constants = binding, pack-job struct layout, storage sizes, argument validation (no launches)"""
    import ctypes as C
    from footprints_amd import _lib, ops
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "footprints_hip.h")).read()
    slots = int(re.search(r"#define FP_AMAX_SLOTS (\d+)", hdr).group(1))
    stride = int(re.search(r"#define FP_AMAX_STRIDE (\d+)", hdr).group(1))
    assert ops.amax_elems() == lib.fp_amax_slot_elems() == slots * stride and stride * 4 == 128      # one 128-byte line per sub-slot
    # fp_pack_job: the amax pointer was appended behind the nine int32 fields (8-byte aligned)
    assert _lib.PackJob.amax.offset == 56 and C.sizeof(_lib.PackJob) == 64
    kinds = re.search(r"FP_PACK_FWD_HP = (\d+), FP_PACK_DGRAD_HP = (\d+), FP_PACK_UP2_FWD_HP = (\d+), FP_PACK_UP2_DGRAD_HP = (\d+)", hdr).groups()
    assert tuple(int(k) for k in kinds) == (_lib.PACK_FWD_HP, _lib.PACK_DGRAD_HP, _lib.PACK_UP2_FWD_HP, _lib.PACK_UP2_DGRAD_HP)
    # two fp16 per weight = one float of storage per padded weight; the exact bf16 split needs 1.5
    assert lib.fp_packed_weight_elems_hp(64, 64, 3, 3, 0) == 9 * 4 * 64 * 16
    assert lib.fp_packed_weight_elems_bf3(64, 64, 3, 3, 0) == 9 * 4 * 64 * 16 * 3 // 2
    assert lib.fp_packed_weight_elems_hp(8, 20, 3, 3, 1) == 9 * 1 * 20 * 16          # data-gradient layout: chunks over Cout, columns = Cin
    d = ops.make_desc(12, 48, 160, 48, 160, 64, 0, 64, 3, 1, 1, _lib.GATHER_FWD_REFLECT)
    assert lib.fp_conv3x3_bf3_supported(C.byref(d)) == 1 and lib.fp_conv3x3_bf3_workspace(C.byref(d)) == 0
    small = ops.make_desc(12, 6, 20, 6, 20, 512, 0, 512, 3, 1, 1, _lib.GATHER_FWD_ZERO)      # 96 tiles: split over the channel chunks
    assert lib.fp_conv3x3_bf3_workspace(C.byref(small)) > 0
    mid = ops.make_desc(12, 12, 40, 12, 40, 256, 0, 256, 3, 1, 1, _lib.GATHER_FWD_ZERO)      # 180 tiles: unsplit since round 2
    assert lib.fp_conv3x3_bf3_workspace(C.byref(mid)) == 0
    rc = lib.fp_conv3x3_hp(C.byref(d), 1, 0, 1, 0, 0, 0, 0, 1, None, 0, None, None, None, None, None, None)      # amax slots missing
    assert rc == -1 and b"amax slots missing" in lib.fp_last_error_string()
    rc = lib.fp_conv_wgrad_hp(C.byref(d), 1, 1, 1, 0, 64, 0, 0, 1, 1 << 30, None, None, None)
    assert rc == -1 and b"amax slots missing" in lib.fp_last_error_string()


def test_module_tree_reproduces_reference_state_dict():
    from footprints_amd import FootprintNetwork
    from footprints_amd.network import is_dead_param
    from oracle import restatement as R
    m = FootprintNetwork(pretrained=True)            # pretrained=True must degrade gracefully offline
    sd = m.state_dict()
    spec = R.state_spec()
    assert [k for k in sd] == [s[0] for s in spec]
    assert [tuple(v.shape) for v in sd.values()] == [tuple(s[1]) for s in spec]
    live = m.live_named_parameters()
    assert len(live) == 196 and sum(p.numel() for _, p in live) == 31012944
    assert sum(1 for n, _ in m.named_parameters() if is_dead_param(n)) == 72
    P, B = R.make_state(tag="cpu")
    m.load_state_dict({**P, **B})                    # reference-format checkpoint loads unchanged
    assert torch.equal(m.encoder.layer1[1][0].conv1.weight, P["encoder.layer1.1.0.conv1.weight"])


def test_no_cpu_compute_path():
    from footprints_amd import FootprintNetwork
    from footprints_amd.training.losses import LossManager
    m = FootprintNetwork(pretrained=False)
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        LossManager((0.1, 100), 0.25)({k: torch.zeros(1, 4, 8, 8) for k in ("1/8", "1/4", "1/2", "1/1")}, {})
    with pytest.raises(RuntimeError):
        m.encoder.layer0(torch.zeros(1, 3, 8, 8)) if False else m.mask_decoder(None)


def test_predict_simple_plumbing_cpu():
    from PIL import Image
    from footprints_amd.predict_simple import MODEL_HEIGHT_WIDTH, InferenceManager, parse_args, preprocess
    from oracle import filler
    from tests.golden.digest import compare, load
    assert MODEL_HEIGHT_WIDTH == {"kitti": (192, 640), "matterport": (512, 640), "handheld": (256, 448)}
    img = (filler.uniform("g6.image", (269, 477, 3)) * 255).astype(np.uint8)
    x = preprocess(Image.fromarray(img, "RGB"), (192, 640))
    assert x.shape == (1, 3, 192, 640) and x.dtype == torch.float32
    assert compare(load("g6_predict"), "predict.input", x) == 0.0
    a = parse_args(["--image", "x.jpg", "--model", "kitti", "--no_cuda", "--no_save_vis", "--save_dir", "d"])
    assert a.no_cuda and a.no_save_vis and a.save_dir == "d"
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        InferenceManager("kitti", "unused", use_cuda=False)
    # visualisation quirk: the mask thresholds the LOGIT at 0.5 (predict_simple.py:77)
    pred = np.zeros((4, 8, 8), np.float32)
    pred[1, :4] = 0.4      # sigmoid(0.4) > 0.5 but logit < 0.5 -> NOT ground in the visualisation
    pred[1, 4:] = 0.6
    pred[3] = 0.5
    vis = InferenceManager.visualise(pred, Image.fromarray(np.full((8, 8, 3), 128, np.uint8)))
    assert (vis[:3] == 128).all() and not (vis[5:] == 128).all()


def test_bucket_ranges_cover_flat_buffer_in_backward_order():
    from footprints_amd import FootprintNetwork
    from footprints_amd.parallel import bucket_ranges
    m = FootprintNetwork(pretrained=False)
    names, offs, total = [], [], 0
    for n, p in m.live_named_parameters():
        names.append(n)
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    b = bucket_ranges(names, offs, total, max_elems=4 << 20)
    assert sum(hi - lo for _, lo, hi in b) == total
    ivs = sorted((lo, hi) for _, lo, hi in b)
    assert ivs[0][0] == 0 and all(a[1] == c[0] for a, c in zip(ivs, ivs[1:])) and ivs[-1][1] == total
    order = [s for s, _, _ in b]
    assert order[0] == "mask_decoder" and order[-1] == "encoder.layer0"
    assert order.index("depth_decoder") < order.index("encoder.layer4") < order.index("encoder.layer1")


def test_staged_adam_ranges_partition_the_flat_buffer():
    """TrainStep's staged update: one range per stage (16-byte aligned), and whatever subset of stages reports, the closing launches cover
    exactly the rest"""
    from footprints_amd import FootprintNetwork
    from footprints_amd.parallel import bucket_ranges
    from footprints_amd.training.train import _ADAM_STAGE_MIN, _complement
    m = FootprintNetwork(pretrained=False)
    names, offs, total = [], [], 0
    for n, p in m.live_named_parameters():
        names.append(n)
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    ranges = {s: (lo, hi) for s, lo, hi in bucket_ranges(names, offs, total, max_elems=total)}
    assert len(ranges) == 7 and all(lo % 4 == 0 and hi % 4 == 0 for lo, hi in ranges.values())
    big = [s for s, (lo, hi) in ranges.items() if hi - lo >= _ADAM_STAGE_MIN]
    assert {"mask_decoder", "depth_decoder", "encoder.layer4", "encoder.layer3"} <= set(big) and "encoder.layer0" not in big
    for reported in ([], big[:1], big, list(ranges)):
        done = [ranges[s] for s in reported]
        rest = _complement(done, total)
        cover = sorted(done + rest)
        assert cover[0][0] == 0 and cover[-1][1] == total and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        assert all(hi > lo for lo, hi in rest)


def test_no_packed_fp32_valu_in_device_code(tmp_path):
    """Tripwire.  Round 1 saw head_wgrad_kernel, built with clang's SLP vectoriser, return different sums from run to run next to the
    bf16-MFMA convolution.  Round 4 narrowed it to ONE instruction form (profiles/round4_notes.md section 12): `v_pk_fma_f32 ... op_sel:[0,1,0]`
    (the low result takes its second operand from the high dword) intermittently loses the low result's product for one 16-lane pass
    while an MFMA-issuing wave shares the SIMD -- in any kernel, with or without the vectoriser; the other op_sel forms, v_pk_add_f32 and
    scalar FMAs tested clean.  The vectoriser emits that form for every acc[c][o] += x[c] * z[o] loop, so the library is built with
    -fno-slp-vectorize -fno-vectorize, and since nothing in it needs packed fp32 VALU at all this test keeps ALL of it out: it
    disassembles every gfx950 code object of the built .so, so a compiler flag change cannot bring the form back unnoticed."""
    import shutil
    import subprocess
    from footprints_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("llvm-objdump or the built library is not available")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    objs = [p for p in tmp_path.iterdir() if p.name.endswith("gfx950")]
    assert objs, "no gfx950 code objects found in the library"
    mfma = 0
    for p in objs:
        asm = subprocess.run([objdump, "-d", str(p)], check=True, capture_output=True, text=True).stdout
        bad = [l for l in asm.splitlines() if re.search(r"\bv_pk_(fma|add|mul)_f32\b", l)]
        assert not bad, "%s: packed fp32 VALU in device code: %s" % (p.name, bad[:3])
        mfma += asm.count("v_mfma_f32_32x32x16_bf16")
    assert mfma > 0            # the disassembly really covered the MFMA kernels


def test_no_register_spills_in_the_matrix_kernels(tmp_path):
    """Tripwire.  The MFMA kernels are written to their register budgets (128 / 256 VGPRs: occupancy and co-residency of the tile and
    weight-gradient kernels on a SIMD depend on them); a spill is a scratch round trip in an epilogue at best and a `s_waitcnt vmcnt(0)`
    that drains a prefetch ring at worst (round 3: the statistics block behind the tile kernel's stores spilled 11 - 45 registers in every
    forward launch until it moved in front of them).  Reads the spill counts out of the code objects' metadata."""
    import shutil
    import subprocess
    from footprints_amd import _lib
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm tools or the built library are not available")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    watched, seen = ("conv3x3_tile_bf3_kernel", "wgrad3x3_hp_pf_kernel", "wgrad3x3_bf3_v3_kernel", "igemm_hp_kernel", "up2_phase_",
                     "stem_tile_hp_kernel", "stem_wgrad_hp_kernel", "head_wgrad_kernel"), 0
    private = 0
    for pth in tmp_path.iterdir():
        if not pth.name.endswith("gfx950"):
            continue
        notes = subprocess.run([readelf, "--notes", str(pth)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split(":", 1)[1].strip()
            elif line.startswith(".vgpr_spill_count:") and name and any(w in name for w in watched):
                seen += 1
                assert int(line.split(":")[1]) == 0, "%s spills %s VGPRs" % (name, line.split(":")[1].strip())
            elif line.startswith(".private_segment_fixed_size:") and name and any(w in name for w in watched):
                # nothing of these kernels lives in scratch at all: rounds 1-3 shipped the tile kernel with a 16-byte private segment in all
                # 34 instantiations (pix[] behind a pointer phi: a scratch load + s_waitcnt vmcnt(0) in front of every chunk's halo loads)
                private += 1
                assert int(line.split(":")[1]) == 0, "%s keeps %s bytes per lane in scratch" % (name, line.split(":")[1].strip())
    assert seen >= 43 and private >= 43            # the metadata really covered the kernels


def test_options_surface_matches_reference_flags():
    """footprints_amd/options.py mirrors footprints/options.py:13-128: every reference flag with the same default (checked against
    the reference's own parser when /root/reference is present, else against the committed table)"""
    import importlib.util
    from footprints_amd.options import Options
    ours = vars(Options().parse([]))
    table = {"mode": "train", "height": 192, "width": 640, "depth_range": [0.1, 100], "training_dataset": "kitti", "epochs": 10,
             "log_freq": 250, "val_batches": 10, "batch_size": 12, "lr": 1e-4, "use_footprint_prior": False, "footprint_prior": 0.25,
             "no_depth_mask": False, "moving_objects_method": "ours", "project_down_baseline": False, "num_workers": 8,
             "config_path": "paths.yaml", "model_name": "model", "log_path": "./logs", "inference_data_type": "kitti",
             "load_path": None, "inference_save_path": None, "save_test_visualisations": False}
    ref_path = "/root/reference/footprints/options.py"
    if os.path.exists(ref_path):
        spec = importlib.util.spec_from_file_location("ref_options", ref_path)
        mod = importlib.util.module_from_spec(spec)
        sys.dont_write_bytecode = True
        spec.loader.exec_module(mod)
        argv, sys.argv = sys.argv, ["x"]
        try:
            ref = vars(mod.Options().parse())
        finally:
            sys.argv = argv
        assert ref == table
    for k, v in table.items():
        assert ours[k] == v, k
    assert set(ours) - set(table) == {"synthetic_steps", "device_augment"}
