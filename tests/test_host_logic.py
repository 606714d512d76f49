"""CPU-only tests: the C-ABI library loads and exports every declared symbol, the compat shims export
the reference's names, and the host-side model surgery (convert & friends) behaves like the
reference's (pycbinfer/__init__.py).  No kernel is launched here."""
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "cbinfer_hip.h")).read()
    declared = set(re.findall(r"\b(cbinfer_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(os.path.join(REPO, "cbinfer_amd", "libcbinfer_hip.so"))
    for name in declared:
        assert hasattr(lib, name), name
    from cbinfer_amd import _lib
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert lib.cbinfer_abi_version() == 11
    # split-state geometry helpers (host, pure): 64 ch 7x7 @80x120: (80 + 13) x (120 + 6) records of 256 B;
    # 16 ch 7x7: two x-adjacent taps per stage -> 7 x 4 stages, records of 64 B, one more column on the right
    lib.cbinfer_split_state_bytes.restype = ctypes.c_long
    lib.cbinfer_split_prepared_bytes.restype = ctypes.c_long
    assert lib.cbinfer_split_supported(64, 256, 7, 7) == 1 and lib.cbinfer_split_supported(3, 16, 7, 7) == 0
    assert lib.cbinfer_split_state_bytes(64, 80, 120, 7, 7) == 4096 + (80 + 13) * (120 + 6) * 256
    assert lib.cbinfer_split_state_bytes(16, 160, 240, 7, 7) == 4096 + (160 + 13) * (240 + 7) * 64
    # prepared weights: fragments, stage table (padded to 16 B), then the plain f32 filter bank (the exact path's)
    assert lib.cbinfer_split_prepared_bytes(64, 256, 7, 7) == 98 * 8 * 4096 + 400 + 256 * 64 * 49 * 4
    assert lib.cbinfer_split_prepared_bytes(16, 64, 7, 7) == 28 * 2 * 4096 + 112 + 64 * 16 * 49 * 4
    lib.cbinfer_mask_words.restype = ctypes.c_long
    assert lib.cbinfer_mask_words_per_row(480) == 8 and lib.cbinfer_mask_words(320, 480) == 2560
    assert lib.cbinfer_weights_kpad(8) == 32 and lib.cbinfer_weights_ckkpad(147, 0) == 160
    # fp16 prepared weights are padded to stage PAIRS (2 x 64): the helper must agree with the byte count
    assert lib.cbinfer_weights_ckkpad(147, 1) == 256
    lib.cbinfer_prepared_weights_bytes.restype = ctypes.c_long
    assert lib.cbinfer_prepared_weights_bytes(8, 3, 7, 7, 1) == 32 * 256 * 2 + 256 * 8


def test_compat_shims_export_reference_symbols():
    import platform
    d = os.path.join(REPO, "cbinfer_amd", "compat")
    cg = ["changeDetection", "changePropagation", "genXMatrix", "updateOutput", "maxPool2d"]
    fg = ["changeDetectionFG", "updateOutputFG", "conv2d_fg_cpu"]
    for fname, names in (("cbconv2d_cg_backend_%s.so", cg), ("cbconv2d_cg_half_backend_%s.so", cg),
                         ("cbconv2d_fg_backend_%s.so", fg)):
        lib = ctypes.CDLL(os.path.join(d, fname % platform.machine()))
        for n in names:
            assert hasattr(lib, n), (fname, n)


def test_host_fg_routine_matches_golden(golden_dir):
    import numpy as np
    from cbinfer_amd import conv2d_fg
    for name in ("fg_test1.npz", "fg_case1.npz"):
        d = dict(np.load(os.path.join(golden_dir, name)))
        out = conv2d_fg.cbconvFG(torch.from_numpy(d["input"]), torch.from_numpy(d["prevInput"]),
                                 torch.from_numpy(d["prevOutput"].copy()), torch.from_numpy(d["weight"]),
                                 float(d["threshold"]))
        np.testing.assert_allclose(out.numpy(), d["output"], rtol=0, atol=1e-5)


def _baseline():
    return nn.Sequential(
        nn.Conv2d(3, 16, 7, padding=3), nn.ReLU(), nn.MaxPool2d(2, 2),
        nn.Conv2d(16, 64, 7, padding=3), nn.ReLU(), nn.MaxPool2d(2, 2),
        nn.Conv2d(64, 256, 7, padding=3), nn.ReLU(),
        nn.Conv2d(256, 64, 1), nn.ReLU(),
        nn.Conv2d(64, 8, 1))


def test_convert_structure():
    import pycbinfer
    base = _baseline()
    cb = pycbinfer.convert(base, threshold=0.07)
    # names preserved, ReLUs merged: children 0,2,3,5,6,8,10
    assert [n for n, _ in cb.named_children()] == ['0', '2', '3', '5', '6', '8', '10']
    convs = [m for m in cb if type(m) is pycbinfer.CBConv2d]
    assert len(convs) == 5 and type(cb[1]) is nn.MaxPool2d
    assert [m.withReLU for m in convs] == [True, True, True, True, False]
    assert all(m.threshold == 0.07 for m in convs)
    # weights are SHARED with the source module (conv2d.py:105-106)
    assert convs[0].weight is base[0].weight and convs[0].bias is base[0].bias
    # flags' defaults (conv2d.py:108-118)
    m = convs[0]
    assert (m.saveChangeMap, m.propChangeIndexes, m.gatherComputationStats, m.finegrained, m.copyInput,
            m.feedbackLoop) == (False, False, False, False, True, False)
    assert pycbinfer.conv2d.CBConv2d is pycbinfer.CBConv2d
    r = repr(convs[0])
    assert r.startswith("CBConv2d (th=0.07, 3->16, k=(7, 7), s=(1, 1), copyInput=True, pad=(3, 3)")
    assert "withReLU=True" in r and r.endswith("propChgIdxs=False)")


def test_convert_nested_dropout_ignorelist():
    import pycbinfer

    class Lambda(nn.Module):
        def forward(self, x):
            return x

    net = nn.Sequential(nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.ReLU(), nn.Dropout()),
                        Lambda(), nn.Conv2d(4, 2, 1))
    cb = pycbinfer.convert(net, ignoreList=[Lambda])
    assert [n for n, _ in cb.named_children()] == ['0', '2']
    inner = cb[0]
    assert [n for n, _ in inner.named_children()] == ['0'] and inner[0].withReLU
    assert type(cb[1]) is pycbinfer.CBConv2d and not cb[1].withReLU


def test_constructor_asserts():
    import pycbinfer
    for bad in (nn.Conv2d(4, 4, 3, padding=1, groups=2), nn.Conv2d(3, 4, 3, padding=0),
                nn.Conv2d(3, 4, 3, padding=1, stride=2), nn.Conv2d(3, 4, 3, padding=2, dilation=2),
                nn.Conv2d(3, 4, 3, padding=1, bias=False)):
        with pytest.raises(AssertionError):
            pycbinfer.CBConv2d(bad, 0.1)
    with pytest.raises(AssertionError):
        pycbinfer.CBPoolMax2d(nn.MaxPool2d(3, 3))
    p = pycbinfer.CBPoolMax2d(nn.MaxPool2d(2, 2, ceil_mode=True))
    assert p.ceil_mode and p.kernel_size == (2, 2)


def test_prop_change_indexes_of_1x1():
    import pycbinfer
    cb = pycbinfer.propChangeIndexesOf1x1(pycbinfer.convert(_baseline()))
    convs = [m for m in cb if type(m) is pycbinfer.CBConv2d]
    assert [m.propChangeIndexes for m in convs] == [False, False, True, True, False]


def test_pickle_roundtrip_and_state_buffers():
    import pycbinfer
    cb = pycbinfer.convert(_baseline())
    pycbinfer.clearMemory(cb)
    buf = io.BytesIO()
    torch.save(cb, buf)
    buf.seek(0)
    cb2 = torch.load(buf, weights_only=False)
    conv = cb2[0]
    assert type(conv) is pycbinfer.CBConv2d
    assert set(dict(conv.named_buffers())) == {'prevInput', 'prevOutput'}
    assert all(t.numel() == 0 for t in pycbinfer.getStateTensors(cb2))
    # attributes missing in old pickles are back-filled (conv2d.py:292-304)
    del conv.__dict__['feedbackLoop'], conv.__dict__['copyInput']
    conv._setDefaultValues()
    assert conv.feedbackLoop is False and conv.copyInput is True


def test_experiment_presets_structure():
    import pycbinfer
    from cbinfer_amd import workloads
    base = workloads.sceneLabelingBaseline()
    assert len(base) == 11
    assert workloads.denseOps(workloads.SCENE_LABELING_SPEC, 320, 480) == 20314521600
    t6 = workloads.configureExperiment(base, pycbinfer.convert(base), 6)
    kinds = [type(m).__name__ for m in t6]
    assert kinds == ['CBConv2d', 'CBPoolMax2d', 'CBConv2d', 'CBPoolMax2d', 'CBConv2d', 'Conv2d', 'ReLU',
                     'Conv2d']
    assert t6[0].propChangeIndexes and t6[2].propChangeIndexes and not t6[4].propChangeIndexes
    assert all(m.feedbackLoop for m in t6 if type(m) is pycbinfer.CBConv2d)
    t7 = workloads.configureExperiment(base, pycbinfer.convert(base), 7)
    assert all(m.finegrained for m in t7 if type(m) is pycbinfer.CBConv2d)


def test_synthetic_video_change_ratio():
    from cbinfer_amd import workloads
    vid = workloads.SyntheticVideo(H=64, W=96, ratio=0.10, block=16, seed=1, device='cpu')
    assert vid.cells == 24 and vid.nblocks == 2
    f0 = vid.frame
    f1 = vid.next()
    changed = (f0 != f1).any(dim=1)
    assert changed.float().mean().item() == pytest.approx(vid.ratio, abs=1e-3)


def test_insert_cb_pooling_structure():
    """insertCBPooling reproduces by rule what modelLoader.py:62-78 builds by hand for experiment 5/6."""
    import pycbinfer
    from cbinfer_amd import workloads
    base = workloads.sceneLabelingBaseline()
    net = pycbinfer.insertCBPooling(pycbinfer.convert(base, threshold=0.05))
    kids = list(net.children())
    assert [type(k).__name__ for k in kids[:5]] == ['CBConv2d', 'CBPoolMax2d', 'CBConv2d', 'CBPoolMax2d', 'CBConv2d']
    assert kids[0].propChangeIndexes and kids[2].propChangeIndexes and not kids[4].propChangeIndexes
    assert kids[2].copyInput and kids[4].copyInput
    # a pool that does not follow a CBConv2d, or is not 2x2/2, stays
    other = pycbinfer.insertCBPooling(pycbinfer.convert(nn.Sequential(
        nn.MaxPool2d(2, 2), nn.Conv2d(3, 4, 3, padding=1), nn.MaxPool2d(3, 2, 1)), threshold=0.1))
    assert [type(k).__name__ for k in other.children()] == ['MaxPool2d', 'CBConv2d', 'MaxPool2d']


def test_oracle_pool_change_indexes():
    from oracle import cb_oracle as orc
    idx = np.array([0, 1, 7, 8, 9, 16, 63], dtype=np.int32)          # 8x8 input
    assert orc.poolChangeIndexes(idx, (8, 8), (4, 4)).tolist() == [0, 3, 4, 15]
    idx7 = np.array([6, 13, 48], dtype=np.int32)                     # 7x7 input, floor pool 3x3
    assert orc.poolChangeIndexes(idx7, (7, 7), (3, 3)).tolist() == []
    assert orc.poolChangeIndexes(idx7, (7, 7), (4, 4)).tolist() == [3, 15]        # ceil-mode pool


def test_split_kernel_fragment_reads_are_not_touched_in_flight():
    """cb_split.hip reads its MFMA fragments with inline-asm ds_read_b128 (so that the compiler does not drain the
    LDS-DMA ring in front of every read); the results only arrive behind the next s_waitcnt lgkmcnt(0).  The
    generated code must not move, spill or read such a register in between (tools/lint_split_isa.py)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "lint_split_isa.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
    assert b"19 cbs_conv_kernel instance(s), 0 finding(s)" in out.stdout


def test_bench_default_build_flags_match_the_makefile():
    """bench.kernel_source_hash() folds the build flags into the hash that ties a bench line to the PMC passes it may
    quote; where the build directory did not travel it uses bench.DEFAULT_BUILD_FLAGS, which must be what a plain
    `make` records in build/.flags."""
    import subprocess
    import bench
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cbinfer_amd", "csrc")
    # (the Makefile's CFLAGS starts with $(EXTRA), empty in a plain build: the leading blank stays)
    rec = subprocess.run(["make", "-s", "-C", csrc, "--eval", "show: ; @echo '$(CFLAGS)'", "show"],
                         capture_output=True, text=True).stdout
    assert rec.encode() == bench.DEFAULT_BUILD_FLAGS, (rec, bench.DEFAULT_BUILD_FLAGS)
    flags = os.path.join(csrc, "build", ".flags")
    if os.path.exists(flags):
        assert open(flags, "rb").read() == bench.DEFAULT_BUILD_FLAGS


def test_fused_tail_link_survives_pickle_and_deepcopy():
    """pycbinfer.fuseTail1x1 leaves the producing layer a plain reference to the CBTail1x1 (not a child module: the
    network's module names stay the reference's); copies of the network keep it pointing at THEIR tail."""
    import copy
    import pickle
    import pycbinfer
    from cbinfer_amd import workloads
    _, net = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.05, device="cpu")
    names = [n for n, _ in net.named_modules()]
    pycbinfer.fuseTail1x1(net)
    for clone in (net, pickle.loads(pickle.dumps(net)), copy.deepcopy(net)):
        kids = list(clone.children())
        head = [m for m in kids if type(m) is pycbinfer.CBConv2d][-1]
        assert type(kids[-1]) is pycbinfer.CBTail1x1 and head.__dict__["_fusedTail"] is kids[-1]
        assert not any(type(m) is pycbinfer.CBTail1x1 for m in head.modules())
    assert [n for n, _ in net.named_modules()] == names[:len(names) - 2]


def test_bench_final_line_stays_small():
    """The driver parses the FINAL stdout line of bench.py out of a tail of a few KB (round 4 lost its record to a
    20 KB line): bench.compact_line of a realistic full result -- the largest one a round has produced, with every
    optional section present -- stays below 4 KB and carries the record's required objects."""
    import json
    import bench
    full = json.load(open(os.path.join(REPO, "profiles", "r04_bench.json")))
    assert len(json.dumps(full)) > 15000          # (the realistic input: the line that was cut off)
    full["details_file"] = "gpurun_out/bench_details.json"
    # prose growing in the full result must not reach the line
    full["roofline"]["kernel"] = full["roofline"]["kernel"] * 8
    full["cpu_baseline"]["sample"] = full["cpu_baseline"]["sample"] * 8
    full["variants"]["exact_f32"]["arithmetic"] = "x" * 5000
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.COMPACT_LIMIT == 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert line["config"]["workload"] and "model" not in line["config"]
    assert json.loads(text)["value"] == pytest.approx(full["value"], rel=1e-4)
    assert all(isinstance(v, (int, float)) for v in line["variants"].values())


def test_no_store_between_a_load_and_its_hand_counted_wait():
    """over every kernel of the library, no store / atomic sits among the N youngest operations of a
    hand-counted `s_waitcnt vmcnt(N)` that covers a load or an LDS-DMA (tools/lint_vmcnt.py, on the generated ISA)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "lint_vmcnt.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:]
    m = re.search(r"(\d+) kernel\(s\), (\d+) hand-counted wait\(s\), 0 finding\(s\)", text)
    assert m and int(m.group(1)) >= 100 and int(m.group(2)) >= 100, text[-1000:]
