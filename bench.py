#!/usr/bin/env python3
"""Headline benchmark: frames/s (+ effective GFLOP/s vs dense) of the change-based scene-labeling CNN
on synthetic 480x320 video at a fixed change ratio (BASELINE.json configs[1]) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one video frame through the converted network (reference experiment 6: CBConv2d with
feedback loop + CBPoolMax2d for the three 7x7 layers, dense 1x1 tail; sceneLabeling/modelLoader.py:62-78).
Every rank owns one independent sequence (weak scaling); the only collective is the final
MAX(elapsed) reduction.  Frames are resident in HBM before the timed region starts.

Rank 0's FINAL stdout line is the record the driver parses: one JSON object below 4 KB (compact_line) with the
metric, `roofline` (measured live on the launch stream for the dominant kernel) and `cpu_baseline` (the oracle
port -- test infrastructure -- of the same path timed on the host cores over a bounded sample).  Everything else
the run measures goes to gpurun_out/bench_details.json and is printed as one {"bench_details": ...} line before it.
"""
import argparse
import contextlib
import glob
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

H, W = 320, 480
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0     # ... BF16/FP16 MFMA, dense
F16X2_CEILING_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3      # f32-equivalent work on f16 pairs: three products per multiply
BF16X3_CEILING_TFLOPS = F16_MFMA_PEAK_TFLOPS / 6     # ... on bf16 triples: six
HBM_PEAK_GBS = 8000.0


def split_arith():
    """Arithmetic of the fp32 split-state kernels in this process: 'x3' (bf16 triples, f32-equivalent; the default),
    'f16x2' (f16 pairs) or None (CBINFER_ARITH names rounds 1-2's kernels)."""
    a = os.environ.get("CBINFER_ARITH", "x3")
    return a if a in ("x3", "f16x2") else None


def split_ceiling():
    return BF16X3_CEILING_TFLOPS if split_arith() == "x3" else F16X2_CEILING_TFLOPS


ARITH_TEXT = {
    "x3": ("f32-equivalent: f32 tensors, every operand as three bf16 terms (24 bits, exact), six products on the bf16 MFMA, "
           "f32 accumulate"),
    "f16x2": "f32 tensors, f16x2 split multiply (22-23 significant bits per operand), f32 accumulate",
    None: "f32 tensors, bf16x3 split multiply on rounds 1-2's kernels (24-bit operands), f32 accumulate",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ratio", type=float, default=0.10)
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--pattern", choices=["blocks", "region"], default="blocks",
                    help="blocks: BASELINE.md config 2 (scattered re-drawn blocks); region: one moving rectangle")
    ap.add_argument("--experiment", type=int, default=6)
    ap.add_argument("--threshold", type=float, default=0.05)
    ap.add_argument("--mode", choices=["auto", "graph", "graph4", "graph8", "eager", "program"], default="auto",
                    help="program: the recorded library calls of a frame replayed as plain stream launches "
                         "(pycbinfer.FrameProgram); graph: one captured hipGraph per frame; graph4 / graph8: four / eight consecutive frames per "
                         "graph, two graphs alternating, the next graph's frames staged on a side stream "
                         "(MultiFrameRunner); eager: plain stream launches (the only form for configurations that "
                         "alias their input as state); auto: calibrate the forms on a short run and time the fastest")
    ap.add_argument("--sequences", type=int, default=1,
                    help="independent video sequences in flight per GPU, one HIP stream + graph each "
                         "(a step then feeds one frame to every sequence)")
    ap.add_argument("--no-fuse-pool", action="store_true",
                    help="keep the change-based pool launches (default: pooling is folded into the next "
                         "layer's change detection, pycbinfer.fusePoolingIntoDetection; same results)")
    ap.add_argument("--multi", type=int, default=4,
                    help="also report the throughput with this many concurrent sequences (0/1: skip)")
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the timed region repeats the K steps until it lasts at least this long (a 20-step "
                         "region is 2 ms: too short for a stable headline); `steps` in the JSON line is the "
                         "number of steps actually timed, `steps_requested` is K, `steps_policy` says which "
                         "rule applied.  --min-seconds 0: exactly K steps are timed")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra measurements that price the structural choices of the headline "
                         "configuration (reference-structured network, one-moving-rectangle video)")
    ap.add_argument("--no-last-frame", action="store_true",
                    help="skip the reference's own timing protocol (evalTools.py:7-35: last frame, min of 3)")
    ap.add_argument("--no-fuse-tail", action="store_true",
                    help="keep the dense 1x1 tail as torch modules (default: pycbinfer.fuseTail1x1, one "
                         "change-based launch for conv1x1->ReLU->conv1x1; results within 1e-4)")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra measurement with frame pipelining (pycbinfer.FramePipeline: the 64->256 "
                         "layer + tail of frame t on a side stream while the first two layers of frame t+1 run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the bounded samples of BASELINE configs 3 (change-ratio sweep, coarse- and fine-grained) "
                         "and 4 (OpenPose fp16)")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the per-layer isolated measurement at an exact post-dilation ratio (SURVEY 8d)")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="(internal) run only the CPU baseline leg and print its JSON object")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel table to stderr")
    ap.add_argument("--pool-clone", action="store_true",
                    help="CBPoolMax2d returns a fresh copy of its state every frame like the reference "
                         "(default: the state tensor itself; identical results, one copy kernel less)")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------
# What goes where.  The driver keeps the TAIL of the output and parses the final stdout line, so that line is
# small and fixed (compact_line: < 4 KB, enforced by tests/test_host_logic.py); everything else this run measures
# -- per-layer tables, kernel trace, isolated layers, sequences per GPU, the config-3/4 samples, all prose -- is the
# DETAILS object: written to a side file (gpurun_out/bench_details.json, or ./bench_details.json) and printed as
# one line {"bench_details": ...} BEFORE the final line.
# ---------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096


def _num(v, digits=10):
    """Floats at 10 significant digits (value, steps and timed_region_s must stay consistent with each other to the
    driver's tolerance); non-finite -> null."""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, v))
    return v


def compact_line(result):
    """The final stdout line: BASELINE.json's metric on its configuration + `roofline` + `cpu_baseline`, values
    only.  `result` is the full dict of main()."""
    cfg = result.get("config", {})
    line = {k: _num(result.get(k)) for k in (
        "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_region_s",
        "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    if result.get("literal_regions_s"):      # (`value` is the median of these three K-step regions)
        line["literal_regions_s"] = [round(float(x), 7) for x in result["literal_regions_s"]]
    line["config"] = {k: cfg.get(k) for k in ("workload", "sequences_per_gpu", "launch", "dist_backend")}
    for k in ("speedup_vs_dense", "dense_fps", "dense_find_mode", "dense_on_cb_kernels_fps", "effective_gflops"):
        if k in result:
            line[k] = _num(result[k])
    rf = result.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {k: (_num(rf.get(k)) if not isinstance(rf.get(k), str) else rf[k][:160])
                            for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                      "avg_duration_us", "units_per_launch") if k in rf}
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: (_num(cb.get(k)) if not isinstance(cb.get(k), str) else cb[k][:200])
                                for k in ("value", "unit", "cores", "kind", "dense_cpu_fps", "sample") if k in cb}
    var = result.get("variants")
    if isinstance(var, dict):
        line["variants"] = {k: _num(v["value"]) for k, v in var.items()
                            if isinstance(v, dict) and isinstance(v.get("value"), (int, float))}
    if isinstance(result.get("repeated_region"), dict):      # (the K steps repeated for >= 0.5 s, beside the literal region)
        line.setdefault("variants", {})["repeated_region"] = _num(result["repeated_region"]["value"], 6)
    # (one more figure of the same run, value only: several sequences on the GPU)
    ms = result.get("multi_sequence")
    if isinstance(ms, dict) and isinstance(ms.get("value"), (int, float)):
        line["multi_sequence"] = {"sequences_per_gpu": ms.get("sequences_per_gpu"), "value": _num(ms["value"], 6)}
    if result.get("details_file"):
        line["details_file"] = result["details_file"]
    return line


def emit(result):
    """Side file + details line, then the final (compact) line."""
    details = {k: v for k, v in result.items()}
    path = None
    for d in (os.path.join(REPO, "gpurun_out"), REPO, os.getcwd()):
        try:
            os.makedirs(d, exist_ok=True)
            cand = os.path.join(d, "bench_details.json")
            with open(cand, "w") as f:
                json.dump(details, f, indent=1)
            path = os.path.relpath(cand, REPO) if cand.startswith(REPO) else cand
            break
        except OSError:
            continue
    result["details_file"] = path
    line = compact_line(result)
    text = json.dumps(line)
    if len(text) >= COMPACT_LIMIT:      # (never let prose push the record out of the driver's window again)
        line.pop("variants", None)
        line["roofline"] = {k: v for k, v in line.get("roofline", {}).items() if k != "kernel"}
        text = json.dumps(line)
    print(json.dumps({"bench_details": details}), flush=True)
    print(text, flush=True)


class PipelinedRunner(object):
    """Feeds frames to a pycbinfer.FramePipeline (eager, one sequence, two streams)."""

    def __init__(self, model, cut, side_stream=None):
        import pycbinfer
        self.pipe = pycbinfer.FramePipeline(model, cut, side_stream)
        self.model, self.mode, self.graph = model, "eager", None
        self.out = None

    def prime(self, frames):
        for f in frames:
            self.out = self.pipe(f)

    def step(self, frame):
        self.out = self.pipe.submit(frame)
        return self.out


class FrameRunner(object):
    """Runs model(frame) either eagerly or as a replayed hipGraph with a static input buffer, on its own
    HIP stream when one is given (several sequences in flight on one GPU)."""

    def __init__(self, model, example, mode, stream=None):
        self.model, self.mode = model, mode
        self.stream = stream
        self.static_in = example.clone()
        if stream is not None:   # model weights, frames and this clone were produced on the default stream
            stream.wait_stream(torch.cuda.current_stream())
        self.graph = None
        self.out = None

    def _ctx(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def prime(self, frames):
        """Process `frames` eagerly (allocates state/workspaces); in graph mode capture afterwards."""
        with torch.no_grad(), self._ctx():
            for f in frames:
                if self.mode == "graph":
                    self.static_in.copy_(f)
                    self.out = self.model(self.static_in)
                else:
                    self.out = self.model(f)
            if self.mode == "graph":
                # capture on this runner's own side stream: the split-K workspace is keyed by stream, so
                # graphs replayed concurrently never share one
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self.out = self.model(self.static_in)      # same frame again: no change
                torch.cuda.current_stream().wait_stream(s)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=s):
                    self.out = self.model(self.static_in)
                self._capture_stream = s

    def step(self, frame):
        with self._ctx():
            if self.graph is not None:
                self.static_in.copy_(frame)
                self.graph.replay()
            else:
                # eager: hand the module the frame tensor itself.  Modules that alias their input as
                # state (copyInput=False without feedback loop, fine-grained; conv2d.py:175,237-238) need
                # a NEW tensor per frame -- such configurations cannot be graph-captured.
                with torch.no_grad():
                    self.out = self.model(frame)
        return self.out


class MultiFrameRunner(object):
    """K consecutive frames per replayed hipGraph (round 6).  A replayed graph costs the GPU ~9 us of idle time per
    hipGraphLaunch whatever it holds, and a one-frame graph needs its frame copied into the graph's input buffer in front
    of it (5 us on the critical path): with one frame per graph, replay lost to eager launches (9.7k vs 10.7k frames/s,
    round 5).  Here a graph holds K frames reading K input buffers, there are TWO such graphs with a buffer set each, and
    the frames of the next replay are copied into the other set on a side stream WHILE the current replay runs -- the
    copies leave the critical path, the launch bubble is paid once per K frames, and the host issues one replay and K
    small copies per K frames instead of 6 K launches.  Results are those of the eager network, bit for bit (same kernels,
    same order: __graft_entry__.smoke()).  Frames handed to step() are consumed in order; flush() runs what is left
    (< K frames) eagerly."""

    def __init__(self, model, example, K=4, stream=None):
        self.model, self.mode, self.K = model, "graph%d" % K, int(K)
        self.stream = stream
        self.graph = None
        self.sets, self.cur, self.pending, self.out = None, 0, [], None
        self.example = example
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())

    def _ctx(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def prime(self, frames):
        from cbinfer_amd.streams import side_stream
        with torch.no_grad(), self._ctx():
            for f in frames:
                self.out = self.model(f)
            self.copy_stream = side_stream(self.example.device)
            cap = torch.cuda.Stream()      # (the capture stream: the modules' call plans are keyed by stream)
            self.sets = []
            for _ in range(2):
                bufs = [frames[-1].clone() for _ in range(self.K)]
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap):
                    for b in bufs:
                        self.out = self.model(b)      # (the same frame again: no change; builds this stream's plans)
                torch.cuda.current_stream().wait_stream(cap)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    for b in bufs:
                        self.out = self.model(b)
                self.sets.append(dict(bufs=bufs, graph=g, ready=torch.cuda.Event(), done=None))
            self.graph = self.sets[0]['graph']
            torch.cuda.current_stream().synchronize()

    def _launch(self):
        s = self.sets[self.cur]
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.copy_stream):
            if s['done'] is not None:
                self.copy_stream.wait_event(s['done'])      # (the previous replay of this set has read its buffers)
            for b, f in zip(s['bufs'], self.pending):
                b.copy_(f, non_blocking=True)
            s['ready'].record(self.copy_stream)
        main.wait_event(s['ready'])
        s['graph'].replay()
        if s['done'] is None:
            s['done'] = torch.cuda.Event()
        s['done'].record(main)
        self.cur ^= 1
        self.pending = []

    def step(self, frame):
        with self._ctx():
            if self.sets is None:
                with torch.no_grad():
                    self.out = self.model(frame)
                return self.out
            self.pending.append(frame)
            if len(self.pending) == self.K:
                self._launch()
        return self.out

    def flush(self):
        with self._ctx(), torch.no_grad():
            for f in self.pending:
                self.out = self.model(f)
            self.pending = []


class ProgramRunner(object):
    """Feeds frames to a pycbinfer.FrameProgram: the library calls of one eager frame, recorded once and replayed as plain
    stream launches with the frame's address patched in (cbinfer_amd/program.py) -- eager kernels without the per-module
    host work, no graph."""

    def __init__(self, model, example, stream=None):
        import pycbinfer
        self.model, self.mode, self.graph, self.stream = model, "program", None, stream
        self.program = pycbinfer.FrameProgram(model)
        self.out = None
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())

    def _ctx(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def prime(self, frames):
        with torch.no_grad(), self._ctx():
            for f in frames:
                self.out = self.model(f)
            for _ in range(2):      # (the same frame again: no change; the per-frame decisions settle)
                self.out = self.model(frames[-1])
            self.out = self.program.record(frames[-1])

    def step(self, frame):
        with self._ctx():
            self.out = self.program(frame)
        return self.out


def build_bench_model(experiment=6, threshold=0.05, fuse_tail=True, fuse_pool=True, pool_clone=False,
                      device="cuda", fuse_detect=None, window_order="auto"):
    """(dense baseline, change-based test network) exactly as the headline measurement runs them: the
    reference's experiment preset (sceneLabeling/modelLoader.py:41-87) on the scene-labeling CNN, then the two
    execution-level fusions (pycbinfer.fuseTail1x1, fusePoolingIntoDetection) and CBPoolMax2d.cloneOutput.
    tests/test_gpu_modules.py::test_bench_configuration_fullsize_parity and __graft_entry__.smoke() build their
    networks through this function, so what is timed is what is checked."""
    import pycbinfer
    from cbinfer_amd import workloads
    base, test = workloads.sceneLabelingModels(experimentIdx=experiment, threshold=threshold, device=device)
    for m in test.modules():
        if type(m) is pycbinfer.CBPoolMax2d:
            m.cloneOutput = bool(pool_clone)
    pycbinfer.fuseTail1x1(test, enabled=fuse_tail)
    pycbinfer.fusePoolingIntoDetection(test, enabled=fuse_pool)
    # (a producer's launch doing the next layer's pooled detection: only where the pool is folded anyway)
    pycbinfer.fuseDetectionIntoProducer(test, enabled=fuse_pool if fuse_detect is None else (fuse_detect and fuse_pool),
                                        windowOrder=window_order)
    return base, test


def bench_video(seed, ratio=0.10, block=32, pattern="blocks", device="cuda"):
    """The synthetic static-camera sequence of BASELINE.md config 2 as the bench feeds it."""
    from cbinfer_amd import workloads
    return workloads.SyntheticVideo(H=H, W=W, ratio=ratio, block=block, seed=seed, pattern=pattern,
                                    device=device)


def pingpong(i, L):
    """Index into a list of L frames walked back and forth (0..L-1..1,0..): consecutive frames always
    differ by exactly one frame's change set, however long the walk (evalTools.py:62 extends a frame list
    the same way for the power measurement)."""
    if L < 2:
        return 0
    j = i % (2 * L - 2)
    return j if j < L else 2 * L - 2 - j


def timed_loop(runners, frames, steps, barrier, start=0):
    """`steps` steps; a step feeds one frame to every runner (one runner = one sequence)."""
    if not isinstance(runners, (list, tuple)):
        runners, frames = [runners], [frames]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(start, start + steps):
        for r, fr in zip(runners, frames):
            r.step(fr[pingpong(i, len(fr))])
    for r in runners:      # (a multi-frame graph runner: the frames that did not fill a graph)
        if hasattr(r, "flush"):
            r.flush()
    torch.cuda.synchronize()
    barrier()
    return time.perf_counter() - t0


def event_time_ms(fn, reps):
    """Average duration of fn() (which enqueues on torch's current stream) over reps launches."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def inframe_layer_times(test, frames, start, reps=40):
    """Duration of every launch group of the frame INSIDE the frame: the frame is enqueued eagerly module by
    module as the network would, except that each change-based layer is issued as its separate library calls
    (detection, then the contraction launch(es)) so that HIP events can be recorded around ONE of them on the
    launch stream.  The kernel then runs on what the preceding layers just left in the caches, next to the same
    neighbours as in the timed loop (stand-alone re-launches run warm and came out up to 35 % shorter).  One
    call is bracketed per pass -- every other launch of the frame runs undisturbed -- and one more pass records
    the two events back to back (nothing between them): that empty-pair time (the cost of the event packets
    themselves, ~5 us) is reported beside the measurements as their upper error bound, not subtracted.
    The walk over `frames` continues at step `start`.  Returns (rows, next step) or None if a layer is not in a
    sync-free feedback-mode form.  rows: one dict per module with the algorithmic work of SURVEY 8d."""
    import pycbinfer
    from cbinfer_amd.conv2d import LazyPool
    from cbinfer_amd._lib import C as lib, check, ptr, stream_ptr, dtype_code
    from cbinfer_amd.conv2d_cg import ChangeIndexes, MaskChangeIndexes
    mods = list(test.children())
    step = start
    # the calls of a frame, in order: (module index, 'detect' | 'conv' | 'tail')
    calls = []
    for i, m in enumerate(mods):
        if type(m) is pycbinfer.CBConv2d and not m.finegrained:
            calls += [(i, 'detect'), (i, 'conv')]
        elif type(m) is pycbinfer.CBTail1x1:
            calls += [(i, 'tail')]
    info = {}

    def bracket(which, key, empty, sink, fn):
        mine = which == key
        if mine:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if empty:
                e1.record()
        fn()
        if mine:
            if not empty:
                e1.record()
            sink.append((e0, e1))

    def run_frame(x, which, empty, sink):
        side = [None]      # (round 6: a row-pair layer's state refresh waiting for the next contraction's idle workgroups)
        for mi, m in enumerate(mods):
            if type(m) is pycbinfer.CBTail1x1:
                xin = x
                out = []
                folded = getattr(xin[2], 'tailDone', None) is m      # evaluated by the producing layer's second launch
                bracket(which, (mi, 'tail'), empty, sink, lambda: out.append(m(xin)))
                x = out[0]
                info[mi] = dict(layer="tail 1x1 %d->%d->%d @%dx%d" % (m.in_channels, m.hidden_channels,
                                                                       m.out_channels, xin[1].shape[-2],
                                                                       xin[1].shape[-1]),
                                count=xin[2].count, folded=folded,
                                flops_per_px=2.0 * (m.in_channels * m.hidden_channels +
                                                    m.hidden_channels * m.out_channels))
                continue
            if type(m) is not pycbinfer.CBConv2d or m.finegrained:
                x = m(x)
                continue
            work = m._work
            if (work is None or not work['selfc'] or not m.feedbackLoop or m.syncIndexes or isinstance(x, tuple)):
                return None
            K, C, kH, kW = m.weight.shape
            folded_detect = False
            lazy = x if isinstance(x, LazyPool) else None
            src = (lazy.source if lazy is not None else x).contiguous()
            Hh, Ww = (lazy.outSize[-2:] if lazy is not None else src.shape[-2:])
            dt, st = dtype_code(src), stream_ptr(src)
            s_el = 4 if src.dtype == torch.float32 else 2
            if m._split_ok(src.dtype, Hh, Ww):
                sp = work['split']
                if sp is None:
                    return None
                wp, scale = m._split_weights(Hh, Ww)
                q = sp['seq'][0]
                pm = lazy.producerMask() if lazy is not None else None
                q.input, q.producerMask = src.data_ptr(), ptr(pm)
                # mode: pooled bit + the state's arithmetic (bf16 triples: CBINFER_SPLIT_X3 -- the frame entry points
                # derive that bit from weightScale == 0; a direct call has to pass it)
                pooled = int(lazy is not None) | (8 if float(scale) == 0.0 else 0)
                pH, pW = (src.size(-2), src.size(-1)) if lazy is not None else (0, 0)
                tokm = m._detect_token()
                folded_detect = (lazy is not None and tokm is not None and
                                 getattr(lazy.indexes, 'nextDetect', None) == tokm)
                if folded_detect:       # (the producing layer's row-pair launch was this layer's detection)
                    bracket(which, (mi, 'detect'), empty, sink, lambda: None)
                else:
                    bracket(which, (mi, 'detect'), empty, sink, lambda: check(lib.cbinfer_split_detect(
                        sp['seq'], 1, pooled, pH, pW, C, Hh, Ww, kH, kW, float(m.threshold), st)))
                tail = m._folded_tail(sp, Hh, Ww, src.device)
                # (round 6: the contraction in window order is the NEXT layer's pooled detection as well, as in the product path)
                pl = m._plan if (m._plan and m._plan.get('split')) else None
                wnext, wtok = (pl.get('keep'), pl.get('nextToken')) if pl is not None else (None, None)
                if wnext is not None and m._next_detect(Hh, Ww)[1] != pl.get('nextRaw'):
                    wnext, wtok = None, None
                if tail is not None:
                    import ctypes
                    bracket(which, (mi, 'conv'), empty, sink, lambda: check(lib.cbinfer_split_conv_tail(
                        sp['seq'], 1, ptr(wp), ptr(m.bias.detach()), C, Hh, Ww, K, kH, kW, float(scale),
                        int(bool(m.withReLU)), ptr(sp['ws']), 0, ctypes.pointer(sp['tail']), st)))
                elif wnext is not None and side[0] is not None and folded_detect:
                    import ctypes
                    from cbinfer_amd import _lib as _l
                    sr = _l.SideRefresh()
                    sr.frame, sr.state, sr.C, sr.H, sr.W, sr.threshold = side[0]
                    side[0] = None
                    bracket(which, (mi, 'conv'), empty, sink, lambda: check(lib.cbinfer_split_conv_next_refresh(
                        sp['seq'], 1, ptr(wp), ptr(m.bias.detach()), C, Hh, Ww, K, kH, kW, float(scale),
                        int(bool(m.withReLU)), ptr(sp['ws']), ctypes.pointer(wnext), ctypes.pointer(sr), st)))
                elif wnext is not None:
                    import ctypes
                    bracket(which, (mi, 'conv'), empty, sink, lambda: check(lib.cbinfer_split_conv_next(
                        sp['seq'], 1, ptr(wp), ptr(m.bias.detach()), C, Hh, Ww, K, kH, kW, float(scale),
                        int(bool(m.withReLU)), ptr(sp['ws']), ctypes.pointer(wnext), st)))
                elif (side[0] is not None and folded_detect and sp['arith'] == 'x3' and
                      lib.cbinfer_split_refresh_supported(C, K, kH, kW, Hh, Ww)):
                    import ctypes
                    from cbinfer_amd import _lib as _l
                    sr = _l.SideRefresh()
                    sr.frame, sr.state, sr.C, sr.H, sr.W, sr.threshold = side[0]
                    side[0] = None
                    bracket(which, (mi, 'conv'), empty, sink, lambda: check(lib.cbinfer_split_conv_refresh(
                        sp['seq'], 1, ptr(wp), ptr(m.bias.detach()), C, Hh, Ww, K, kH, kW, float(scale),
                        int(bool(m.withReLU)), ptr(sp['ws']), ctypes.pointer(sr), st)))
                else:
                    bracket(which, (mi, 'conv'), empty, sink, lambda: check(lib.cbinfer_split_conv(
                        sp['seq'], 1, ptr(wp), ptr(m.bias.detach()), C, Hh, Ww, K, kH, kW, float(scale),
                        int(bool(m.withReLU)), ptr(sp['ws']), 0, st)))
                ci = MaskChangeIndexes(sp['copy'], (Hh, Ww), work['idx'], work['count'], made=True)
                ci.tailDone = tail
                ci.nextDetect = wtok
                kern = "cbs_conv_kernel (split-state, LDS-DMA, %s%s)" % (
                    "bf16-triple products: f32-equivalent" if sp['arith'] == 'x3' else "f16-pair products",
                    "; window order + the next layer's pooled change detection" if wnext is not None else "") + \
                    (" + cbs_reduce_tail_kernel (second launch: sums the k-slices' partial tiles AND evaluates the "
                     "1x1 tail)" if tail is not None else " + cbs_reduce_kernel" if sp['ws'] is not None else "")
            else:
                mpath = m._rows_path(src.dtype, Hh, Ww)
                rows = m._rows_workspace(work, Hh, Ww, src.device) if mpath else None
                bits = rows['bits'] if rows is not None else work['bits']
                if lazy is not None:
                    if rows is not None:
                        det = lambda: check(lib.cbinfer_change_detection_bits_pooled(
                            ptr(src), src.shape[-2], src.shape[-1], ptr(lazy.producerMask()), ptr(m.prevInput),
                            ptr(bits), Ww, Hh, C, (kH - 1) // 2, (kW - 1) // 2, float(m.threshold), dt, st))
                    else:
                        det = lambda: check(lib.cbinfer_change_detection_frame_pooled(
                            ptr(src), src.shape[-2], src.shape[-1], ptr(m.prevInput), ptr(bits), Ww, Hh, C,
                            (kH - 1) // 2, (kW - 1) // 2, float(m.threshold), dt, st))
                else:
                    dfn = lib.cbinfer_change_detection_bits if rows is not None else \
                        lib.cbinfer_change_detection_frame
                    det = lambda: check(dfn(ptr(src), ptr(m.prevInput), ptr(bits), Ww, Hh, C, (kH - 1) // 2,
                                            (kW - 1) // 2, float(m.threshold), 1, dt, st))
                pairs_tok = None
                own_det = False
                if rows is not None and mpath == 'rows' and lazy is None and m._pairs_ok(Hh, Ww):
                    import ctypes
                    nxt, pairs_tok = m._next_detect(Hh, Ww)
                    nptr = ctypes.pointer(nxt) if nxt is not None else None
                    own_det = m._pair_detect_ok(nxt, kH)      # (round 6: as the product path decides)
                if own_det:
                    folded_detect = True
                    bracket(which, (mi, 'detect'), empty, sink, lambda: None)
                    conv = lambda: check(lib.cbinfer_conv_rowpairs_detect(
                        ptr(src), ptr(m.prevInput), ptr(m.prevOutput), ptr(rows['copy']), ptr(m._masked_call(mpath)[1]),
                        ptr(m.bias.detach()), C, Hh, Ww, K, kH, kW, float(m.threshold), int(m.withReLU), nptr, st))
                    side[0] = (src.data_ptr(), m.prevInput.data_ptr(), C, Hh, Ww, float(m.threshold))
                    kern = "cbp_rowpair_kernel<DET> (row pairs + the layer's own change detection + the next layer's pooled one)"
                else:
                    bracket(which, (mi, 'detect'), empty, sink, det)
                if own_det:
                    pass
                elif rows is not None and mpath == 'rows' and lazy is None and m._pairs_ok(Hh, Ww):
                    conv = lambda: check(lib.cbinfer_conv_changed_rowpairs(
                        ptr(m.prevInput), ptr(rows['bits']), ptr(rows['arrive']), ptr(rows['copy']),
                        ptr(m._masked_call(mpath)[1]), ptr(m.bias.detach()), ptr(m.prevOutput), C, Hh, Ww, K, kH, kW,
                        int(m.withReLU), nptr, st))
                    kern = "cbp_rowpair_kernel (row pairs, persistent" + \
                        ("; + the next layer's pooled change detection)" if nxt is not None else ")")
                elif rows is not None:
                    kfn = lib.cbinfer_conv_changed_rows if mpath == 'rows' else lib.cbinfer_conv_changed_blocks
                    conv = lambda: check(kfn(ptr(m.prevInput), ptr(rows['bits']), ptr(rows['arrive']),
                                             ptr(rows['copy']), ptr(m._masked_call(mpath)[1]), ptr(m.bias.detach()),
                                             ptr(m.prevOutput), C, Hh, Ww, K, kH, kW, int(m.withReLU), st))
                    kern = "cb_rowconv_f32_kernel" if mpath == 'rows' else "cb_blockconv_kernel"
                else:
                    ar = m._arith(src)
                    conv = lambda: check(lib.cbinfer_conv_changed_from_mask(
                        ptr(m.prevInput), ptr(work['bits']), ptr(work['idx']), ptr(work['count']),
                        ptr(m._prepared_weights(Hh, Ww, ar)), ptr(m.bias.detach()), ptr(m.prevOutput), C, Hh, Ww, K,
                        kH, kW, int(m.withReLU), ptr(work['conv']), ar, st))
                    kern = "cb_mfma_f32_kernel<X3: bf16x3 split>" if ar == 2 else "cb_mfma_f32_kernel"
                bracket(which, (mi, 'conv'), empty, sink, conv)
                ci = (MaskChangeIndexes(rows['copy'], (Hh, Ww), work['idx'], work['count']) if rows is not None
                      else ChangeIndexes(work['idx'], work['count'], (Hh, Ww)))
                if rows is not None:
                    ci.nextDetect = pairs_tok
            info[mi] = dict(layer="conv %d->%d k%d @%dx%d" % (C, K, kH, Hh, Ww), conv_kernel=kern, count=ci,
                            HW=Hh * Ww, C=C, K=K, k=kH * kW, s=s_el, pooled=lazy is not None,
                            detect_folded=folded_detect)
            x = ('changeIndexes', m.prevOutput, ci) if m.propChangeIndexes else m.prevOutput
        if side[0] is not None:      # (nobody carried the refresh: a launch of its own, as CBConv2d.forward does)
            fr, stt, c_, h_, w_, th_ = side[0]
            check(lib.cbinfer_refresh_state(fr, stt, c_, h_, w_, th_, st))
        return x

    times, counts = {}, {}
    with torch.no_grad():
        for key in [None] + calls:          # None: the empty-pair pass (bracket at the first call's place)
            sink = []
            cnt = {}
            for it in range(reps + 3):
                got = []
                if run_frame(frames[pingpong(step, len(frames))], calls[0] if key is None else key, key is None,
                             got) is None:
                    return None
                step += 1
                if it >= 3:
                    sink += got
                    if key is not None and key[1] != 'detect':
                        c = info[key[0]]['count']
                        c = c.count if hasattr(c, 'count') and not isinstance(c, torch.Tensor) else c
                        cnt.setdefault(key[0], torch.zeros(1, dtype=torch.int64, device=frames[0].device))
                        cnt[key[0]] += c
            torch.cuda.synchronize()
            times[key] = 1e3 * sum(a.elapsed_time(b) for a, b in sink) / len(sink)      # microseconds
            for k, v in cnt.items():
                counts[k] = v.item() / float(len(sink))
    rows = []
    for mi, m in enumerate(mods):
        if mi not in info:
            if type(m) is pycbinfer.CBPoolMax2d:
                rows.append(dict(layer="pool (folded into the next detection)" if getattr(m, 'lazy', False)
                                 else "pool (own launch: timed with its consumer)"))
            continue
        d = info[mi]
        if 'flops_per_px' in d:
            n = counts.get(mi, 0.0)
            r = dict(layer=d['layer'], N=n, tail_ms=times[(mi, 'tail')] * 1e-3, tail_flops=d['flops_per_px'] * n)
            if d.get('folded'):
                r['tail_ms'] = 0.0
                r['folded'] = "evaluated in the second launch of the layer in front (its conv_ms covers it)"
            rows.append(r)
            continue
        n = counts.get(mi, 0.0)
        rows.append(dict(layer=d['layer'], N=n, ratio=n / float(d['HW']), conv_kernel=d['conv_kernel'],
                         detect_ms=times[(mi, 'detect')] * 1e-3, detect_pooled=d['pooled'],
                         detect_in_producer_launch=d['detect_folded'],
                         detect_bytes=(5 if d['pooled'] else 2) * d['C'] * d['HW'] * d['s'] + d['HW'] // 8,
                         conv_ms=times[(mi, 'conv')] * 1e-3, conv_flops=2.0 * n * d['C'] * d['k'] * d['K'],
                         conv_bytes=(n * d['C'] * d['k'] + d['K'] * d['C'] * d['k'] + n * d['K']) * d['s']))
    return rows, step, times[None]



class _ConvLike(object):
    """What CBConv2d / CBTail1x1 read from the module they are built from (shares the parameters)."""

    def __init__(self, weight, bias, k):
        self.weight, self.bias = weight, bias
        self.out_channels, self.in_channels = weight.shape[0], weight.shape[1]
        self.kernel_size, self.stride, self.dilation = (k, k), (1, 1), (1, 1)
        self.padding, self.output_padding, self.groups, self.transposed = (k // 2, k // 2), (0, 0), 1, False


def isolated_layers(test, ratio=0.10, reps=40):
    """every converted layer of the bench network IN ISOLATION at a post-dilation change ratio of exactly
    `ratio`: its own module (shared weights, fresh state) on N(0,1) feature maps of the layer's size, fed two
    alternating frames that differ in one rectangle whose dilation by the filter support covers exactly ratio*H*W
    pixels (pooled layers: the rectangle in front of the pool).  Timed: the layer's library call (detection launch +
    contraction launch(es)) back to back over `reps` frames between ONE pair of HIP events -- no bracket overhead
    to speak of -- and, in two more passes, the detection launch and the contraction launch(es) bracketed separately;
    their bracketed times net of the empty pair are scaled so that they sum to the back-to-back time.  Algorithmic
    work per SURVEY 8(d).  No producer mask in isolation: the pooled detections read their whole input (in the frame
    the producer's mask lets them skip ~80 % of it)."""
    import pycbinfer
    from cbinfer_amd.conv2d import LazyPool
    mods = list(test.children())
    rows = []
    size = (H, W)
    pooled_next = False
    gen = torch.Generator(device="cpu")
    gen.manual_seed(4242)
    for mi, m in enumerate(mods):
        if type(m) is pycbinfer.CBPoolMax2d:
            pooled_next = True
            continue
        if type(m) is not pycbinfer.CBConv2d or m.finegrained or not m.feedbackLoop:
            continue
        K, C, kH, kW = m.weight.shape
        srcH, srcW = size
        if pooled_next:
            size = (srcH // 2, srcW // 2)
        Hh, Ww = size
        N = int(round(ratio * Hh * Ww))
        # dilated rectangle rh x rw = N exactly, core = it minus the filter's reach on every side
        rh = next(h for h in range(int(round((ratio * Hh * Hh) ** 0.5)), 0, -1) if N % h == 0 and N // h <= Ww)
        rw = N // rh
        ch, cw = rh - (kH - 1), rw - (kW - 1)
        y0, x0 = (Hh - ch) // 2, (Ww - cw) // 2
        layer = pycbinfer.CBConv2d(_ConvLike(m.weight, m.bias, kH), m.threshold)
        layer.feedbackLoop, layer.withReLU, layer.copyInput = True, m.withReLU, m.copyInput
        tail = m.__dict__.get('_fusedTail')
        if tail is not None:
            t2 = pycbinfer.CBTail1x1(_ConvLike(tail.weight1, tail.bias1, 1), _ConvLike(tail.weight2, tail.bias2, 1),
                                     relu=tail.relu)
            layer.propChangeIndexes = True
            layer.__dict__['_fusedTail'] = t2
        sc = 2 if pooled_next else 1
        a = torch.randn(1, C, srcH, srcW, generator=gen).cuda()
        b = a.clone()
        patch = torch.randn(1, C, ch * sc, cw * sc, generator=gen).cuda()
        b[:, :, y0 * sc:(y0 + ch) * sc, x0 * sc:(x0 + cw) * sc] = patch + 4.0      # (every window's maximum moves)
        frames = [a, b]

        def feed(f):
            if pooled_next:
                return layer(LazyPool(f, (1, C, Hh, Ww), False, None))
            return layer(f)
        with torch.no_grad():
            for i in range(4):
                feed(frames[i & 1])
            torch.cuda.synchronize()
            n = int(layer.lastChangeIndexes().count.item())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                feed(frames[i & 1])
            e1.record()
            torch.cuda.synchronize()
            pair_us = 1e3 * e0.elapsed_time(e1) / reps
        split = bool(layer._plan and layer._plan.get('split'))
        s_el = 4
        det_bytes = (5 if pooled_next else 2) * C * Hh * Ww * s_el + Hh * Ww // 8 + 2 * ch * cw * C * s_el
        flops = 2.0 * n * C * kH * kW * K
        conv_bytes = (2 * n * C * kH * kW + K * C * kH * kW + 2 * n * K) * s_el + 4 * n
        row = dict(layer="conv %d->%d k%d @%dx%d" % (C, K, kH, Hh, Ww), N=n, target_N=N, ratio=n / float(Hh * Ww),
                   pooled=pooled_next, kernels=("cbs_detect_kernel + cbs_conv_kernel" +
                                                (" + cbs_reduce_tail_kernel" if tail is not None else "")) if split
                   else ("cb_detect_kernel + cbp_rowpair_kernel" if (layer._plan and layer._plan.get('pairs'))
                         else "cb_detect_kernel + cb_rowconv_f32_kernel"), call_us=pair_us,
                   detect_bytes=det_bytes, conv_flops=flops, conv_bytes=conv_bytes,
                   tail_flops=(2.0 * n * (tail.in_channels * tail.hidden_channels + tail.hidden_channels *
                                          tail.out_channels)) if tail is not None else 0.0)
        rows.append((row, layer, feed, frames))
        pooled_next = False
    # the split of each call into its detection and contraction parts: one bracketed pass each + the empty pair
    out = []
    for row, layer, feed, frames in rows:
        parts = {}
        with torch.no_grad():
            for which in ("empty", "detect", "conv"):
                sink = []
                for i in range(reps // 2 + 2):
                    _bracketed_call(layer, feed, frames[i & 1], which, sink if i >= 2 else [])
                torch.cuda.synchronize()
                parts[which] = 1e3 * sum(x.elapsed_time(y) for x, y in sink) / max(len(sink), 1)
        det = max(parts["detect"] - parts["empty"], 0.05)
        con = max(parts["conv"] - parts["empty"], 0.05)
        scale = row["call_us"] / (det + con)
        row["detect_us"], row["conv_us"] = det * scale, con * scale
        row["event_pair_us"] = parts["empty"]
        row["detect_GBps"] = row["detect_bytes"] / (row["detect_us"] * 1e-6) / 1e9
        row["detect_frac_hbm"] = row["detect_GBps"] / HBM_PEAK_GBS
        row["conv_TFLOPs"] = row["conv_flops"] / (row["conv_us"] * 1e-6) / 1e12
        row["conv_GBps_algorithmic"] = row["conv_bytes"] / (row["conv_us"] * 1e-6) / 1e9
        split = "cbs_conv" in row["kernels"]
        row["conv_ceiling_TFLOPs"] = split_ceiling() if split else FP32_MFMA_PEAK_TFLOPS
        row["conv_frac_of_ceiling"] = row["conv_TFLOPs"] / row["conv_ceiling_TFLOPs"]
        out.append({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()})
    return out


def _bracketed_call(layer, feed, frame, which, sink):
    """One isolated frame of `layer` with HIP events around its detection launch or around its contraction
    launch(es) ('empty': the two events back to back in front of the call)."""
    from cbinfer_amd._lib import C as lib
    if which == "empty":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        feed(frame)
        sink.append((e0, e1))
        return
    # the library's frame entry points enqueue detection + contraction in one call: split them here through the
    # same plan arguments the module uses
    plan = layer._plan
    if plan is None:
        feed(frame)
        return
    import ctypes
    st = plan['args'][-1]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    if plan.get('split'):
        a = plan['args']
        seqs, nS, pooled, pH, pW, wp, bias, C_, Hh, Ww, K, kH, kW, th, scale, relu, ws = a[:17]
        plan['seq'].input = frame.data_ptr()
        if which == "detect":
            ev[0].record()
        # (the frame entry point derives CBINFER_SPLIT_X3 from weightScale == 0: the direct call passes it)
        _chk(lib.cbinfer_split_detect(seqs, nS, (int(pooled) & ~8) | (8 if float(scale) == 0.0 else 0), pH, pW, C_, Hh, Ww,
                                      kH, kW, th, st))
        if which == "detect":
            ev[1].record()
        else:
            ev[0].record()
        if plan['tail'] is not None:
            _chk(lib.cbinfer_split_conv_tail(seqs, nS, wp, bias, C_, Hh, Ww, K, kH, kW, scale, relu, ws, 0, a[18], st))
        else:
            _chk(lib.cbinfer_split_conv(seqs, nS, wp, bias, C_, Hh, Ww, K, kH, kW, scale, relu, ws, 0, st))
        if which == "conv":
            ev[1].record()
    elif plan.get('pairs'):
        a = list(plan['args'])
        # cbinfer_cbconv2d_forward_rowpairs(input, state, out, bits, ctl, copy, wprep, bias, C, H, W, K, kH, kW, th, relu,
        #                                   next, stream)
        _, state, outp, bits, ctl, copy, wprep, bias, C_, Hh, Ww, K, kH, kW, th, relu, nxt = a[:17]
        if which == "detect":
            ev[0].record()
        _chk(lib.cbinfer_change_detection_bits(frame.data_ptr(), state, bits, Ww, Hh, C_, (kH - 1) // 2, (kW - 1) // 2,
                                               th, 1, 0, st))
        if which == "detect":
            ev[1].record()
        else:
            ev[0].record()
        _chk(lib.cbinfer_conv_changed_rowpairs(state, bits, ctl, copy, wprep, bias, outp, C_, Hh, Ww, K, kH, kW, relu,
                                               nxt, st))
        if which == "conv":
            ev[1].record()
    else:
        a = list(plan['args'])
        a[plan['srcSlot']] = frame.data_ptr()
        # cbinfer_cbconv2d_forward_rows(input, pooledSrc, pH, pW, pmask, state, out, bits, arrive, copy, wprep, bias,
        #                               C, H, W, K, kH, kW, th, feedback, copyInput, relu, stream)
        inp, psrc, pH, pW, pmask, state, outp, bits, arrive, copy, wprep, bias, C_, Hh, Ww, K, kH, kW, th = a[:19]
        relu = a[21]
        if which == "detect":
            ev[0].record()
        _chk(lib.cbinfer_change_detection_bits(inp, state, bits, Ww, Hh, C_, (kH - 1) // 2, (kW - 1) // 2, th, 1, 0, st))
        if which == "detect":
            ev[1].record()
        else:
            ev[0].record()
        _chk(lib.cbinfer_conv_changed_rows(state, bits, arrive, copy, wprep, bias, outp, C_, Hh, Ww, K, kH, kW, relu, st))
        if which == "conv":
            ev[1].record()
    sink.append((ev[0], ev[1]))


def _chk(status):
    from cbinfer_amd._lib import check
    check(status)



def openpose_config(args, measure):
    """BASELINE.json configs[3]: OpenPose T=2 (poseDetection/openPose/PoseModel.py:34-68, :122-137), 368x654, fp16
    (cg_half path), all 36 convs converted per sub-model as poseDetection/modelConverter.py:20-24 does, 10 % of the input
    re-drawn per frame in 16x16 blocks -- on a LIVE network (round 5): variance-preserving random weights
    (workloads.OpenPoseModel(init='kaiming')) and per-layer thresholds that give every layer a post-dilation change ratio
    of ~10 % in the running change-based network (workloads.calibrateChangeRatio), so that all 36 layers recompute and
    the frame carries the ~22 GFLOP BASELINE.md budgets.  Frames are FRESH and consecutive (no walk back and forth: a
    pixel refreshed one frame ago jumps less than one that has been stale for ten, so a ping-pong walk under-states the
    change behind thresholded layers)."""
    import pycbinfer
    from cbinfer_amd import workloads
    Hp, Wp = 368, 654
    vid = workloads.SyntheticVideo(H=Hp, W=672, ratio=0.10, block=16, seed=3)

    def prep(f):        # PoseDetector.py:72
        return (f[:, :, :, :Wp] * (255.0 / 256.0) - 0.5).half().contiguous()

    def live():
        return workloads.OpenPoseModel(T=2, init='kaiming').cuda().half()
    psteps, pwarm = 30, 30
    base = live()

    def converted(model=None, **kw):
        """The converted network in the execution form of round 6: every chained layer's change detection inside its
        producer's launch (workloads.fuseOpenPoseDetections) and the two branches of a stage walked in lockstep, one launch
        per layer pair (OpenPoseModel(groupedBranches=True): pycbinfer.BranchGroup) -- results bit-identical to the
        separate launches (tests/test_gpu_hgroup.py)."""
        if model is None:
            model = workloads.OpenPoseModel(T=2, init='kaiming', groupedBranches=True).cuda().half()
        return workloads.fuseOpenPoseDetections(workloads.convertOpenPose(model, threshold=0.02, **kw))
    test = converted()
    ths = workloads.calibrateChangeRatio(test, lambda: prep(vid.next()), target=0.10)
    convs = [m for m in test.modules() if type(m) is pycbinfer.CBConv2d]
    frames = [prep(vid.frame)] + [prep(vid.next()) for _ in range(2 + pwarm + psteps + 22)]
    fresh, frames = frames[-22:], frames[:-22]

    def with_thresholds(model):
        for m, th in zip([m for m in model.modules() if type(m) is pycbinfer.CBConv2d], ths):
            m.threshold = th
        return model
    dense = max(measure(base, frames, m, psteps, 3) for m in ("graph", "eager"))
    cb_modes = {m: measure(test, frames, m, psteps, pwarm) for m in ("eager", "graph")}
    cb = max(cb_modes.values())
    # the same network with every layer running its own detection launch (round 5's form: 88 launches per frame)
    unf = with_thresholds(workloads.convertOpenPose(live(), threshold=0.02))
    cb_unfolded = max(measure(unf, frames, m, psteps, pwarm) for m in ("graph", "eager"))
    del unf
    # per-layer change ratios and recomputed work: the mean over ten fresh frames behind the timed ones
    counts = [0.0] * len(convs)
    with torch.no_grad():
        test(frames[-1])
        for f in fresh[:10]:
            test(f)
            for i, m in enumerate(convs):
                counts[i] += m.lastChangeIndexes().numel() / 10.0
    rs, flops, byts = [], 0.0, 0.0
    folded_det = 0
    layer_rows = []
    for m, n in zip(convs, counts):
        K, Cc, kh, kw = m.weight.shape
        hw = float(m.prevInput.size(-1) * m.prevInput.size(-2))
        rs.append(n / hw)
        flops += 2.0 * n * Cc * kh * kw * K
        # input + state read (f16), the mask -- of the layers that still run a detection LAUNCH; a layer whose
        # detection rides in its producer's launch reads 2 N C s bytes there (the values just written and their state)
        hsw = (m._work or {}).get('hsplit')
        if hsw is not None and hsw['layer'][0].detect == 0:
            folded_det += 1
        else:
            byts += 2.0 * Cc * hw * 2 + hw / 8
        layer_rows.append({"layer": "%d->%d k%d @%dx%d" % (Cc, K, kh, m.prevInput.size(-2), m.prevInput.size(-1)),
                           "ratio": round(n / hw, 4), "threshold": round(float(m.threshold), 4),
                           "path": m._plan['fn'].__name__ if getattr(m, '_plan', None) and m._plan.get('fn') else None})
    # roofline of the frame's launches (kernel durations: in-process kernel trace of eager frames of the same walk)
    pose_roofline = None
    try:
        walk = fresh[10:]      # (consecutive frames, each used once: traced_kernel_durations runs 3 untraced + 8 traced steps)

        def pstep(i):
            with torch.no_grad():
                test(walk[min(i, len(walk) - 1)])
        got = traced_kernel_durations(pstep, 8)
        if got[0] is not None:
            k = got[0]
            conv_us = sum(v["launches_per_frame"] * v["avg_us"] for n, v in k.items()
                          if "conv_kernel" in n or "cb_mfma" in n or "reduce" in n)
            det_us = sum(v["launches_per_frame"] * v["avg_us"] for n, v in k.items() if "detect" in n)
            pose_roofline = {
                "contractions": {"bound": "mfma", "flops_per_frame": flops, "us_per_frame": conv_us,
                                 "achieved": flops / conv_us / 1e6, "peak": 2500.0, "unit": "TFLOP/s",
                                 "frac": flops / conv_us / 1e6 / 2500.0,
                                 "note": "f16 MFMA flops of the recomputed pixels (2 N C k K summed over the 36 layers, mean "
                                         "of ten frames) over the summed durations of the contraction launches of a frame "
                                         "(all 36 layers are busy on the live network); peak = dense f16 MFMA"},
                "detections": {"bound": "hbm", "bytes_per_frame": byts, "us_per_frame": det_us,
                               "achieved": byts / det_us / 1e3, "peak": 8000.0, "unit": "GB/s",
                               "frac": byts / det_us / 1e3 / 8000.0,
                               "detection_launches": len(rs) - folded_det, "detections_in_producer_launches": folded_det,
                               "note": "SURVEY 8(d) bytes (input + state read, mask) of the layers that run a detection "
                                       "LAUNCH over the summed durations of those launches; the other layers' detection "
                                       "rides in their producers' launches"},
                "kernels": {n[:60]: v for n, v in sorted(k.items(), key=lambda x: -x[1]["avg_us"] * x[1]["launches_per_frame"])[:8]},
                "launches_per_frame": sum(v["launches_per_frame"] for v in k.values()),
                "busy_us_per_frame": got[1]}
    except Exception as e:      # (an add-on: never at the expense of the line)
        pose_roofline = {"error": repr(e)}
    # the reference's "recursive mode" (modelConverter.py:84-86): feedback loop -- thresholds calibrated in that mode (a
    # state that is refreshed at the changed pixels only drifts differently: the copy mode's thresholds let the change die
    # out behind the fourth layer there), on the video's next frames
    testf = converted(feedbackLoop=True)
    workloads.calibrateChangeRatio(testf, lambda: prep(vid.next()), target=0.10)
    fframes = [prep(vid.frame)] + [prep(vid.next()) for _ in range(2 + 3 + psteps + 10)]
    ffresh, fframes = fframes[-10:], fframes[:-10]
    cbf = max(measure(testf, fframes, m, psteps, 3) for m in ("eager",))
    fconvs = [m for m in testf.modules() if type(m) is pycbinfer.CBConv2d]
    fcount = [0.0] * len(fconvs)
    with torch.no_grad():
        for f in ffresh:
            testf(f)
            for i, m in enumerate(fconvs):
                fcount[i] += m.lastChangeIndexes().numel() / 10.0
    fratio = [n / float(m.prevInput.size(-1) * m.prevInput.size(-2)) for m, n in zip(fconvs, fcount)]
    fflops = sum(2.0 * n * m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3] * m.weight.shape[0]
                 for m, n in zip(fconvs, fcount))
    del testf
    # every layer scanning its whole input as the reference's layers do (conv2d.py:228-233): what the chained entry
    # (cbinfer_cbconv2d_forward_after) contributes -- nothing to speak of on a network whose layers all change
    from cbinfer_amd import conv2d as _c2
    _c2._NO_CHAIN = True
    try:
        plain = with_thresholds(converted())
        cbu = max(measure(plain, frames, m, psteps, pwarm) for m in ("graph", "eager"))
        del plain
    finally:
        _c2._NO_CHAIN = os.environ.get("CBINFER_NO_CHAIN", "0") == "1"
    # the three VGG pools change-based as well and folded into their consumers' detections (pycbinfer.insertCBPooling +
    # fusePoolingIntoDetection: what sceneLabeling/modelLoader.py:62-78 does by hand; the pose converter of the
    # reference leaves the pools dense, so this is an option beside the configuration)
    cbp = with_thresholds(converted())
    pycbinfer.insertCBPooling(cbp, cloneOutput=False)
    pycbinfer.fusePoolingIntoDetection(cbp)
    cbpool_modes = {m: measure(cbp, frames, m, psteps, pwarm) for m in ("graph", "eager")}
    # ... and, with the stage inputs concatenated by the library (pycbinfer.ChannelConcat instead of torch.cat), the frame
    # consists of library calls only: replayed as a recorded launch program (pycbinfer.FrameProgram)
    library_calls = None
    try:
        cbp.libraryConcat = True
        with torch.no_grad():
            for f in frames[:4]:
                cbp(f)
        prog = pycbinfer.FrameProgram(cbp)
        prog.record(frames[4])
        library_calls = len(prog.calls)
        walk = frames[5:]
        with torch.no_grad():
            for f in walk[:pwarm]:
                prog(f)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in walk[pwarm:pwarm + psteps]:
                prog(f)
            torch.cuda.synchronize()
            cbpool_modes["program"] = len(walk[pwarm:pwarm + psteps]) / (time.perf_counter() - t0)
        del prog
    except Exception as e:      # (an add-on: never at the expense of the line)
        cbpool_modes["program_error"] = repr(e)
    cbpool = max(v for v in cbpool_modes.values() if isinstance(v, float))
    del cbp
    # the two branches of every stage (PoseModel.py:122-137: independent until the concat) on two HIP streams, fork / join
    # per stage -- an execution option of the MODEL, applied to the dense network as well
    cbc = with_thresholds(converted(workloads.OpenPoseModel(T=2, init='kaiming', concurrentBranches=True).cuda().half()))
    cbconc = measure(cbc, frames, "eager", psteps, pwarm)
    del cbc
    densec = measure(workloads.OpenPoseModel(T=2, init='kaiming', concurrentBranches=True).cuda().half(), frames, "eager",
                     psteps, 3)
    # several sequences per GPU (the serving case: a camera per sequence): S copies of the calibrated network -- own states,
    # own video --, one captured graph and one stream each (streams probed to overlap: cbinfer_amd/streams.py), frames
    # fed round-robin; the dense network takes the S frames as ONE batch
    multi = {}
    try:
        from cbinfer_amd.streams import overlapping_streams
        for S in (2, 4):
            sstreams = overlapping_streams(S)
            vids = [workloads.SyntheticVideo(H=Hp, W=672, ratio=0.10, block=16, seed=3 + 101 * (q + 1)) for q in range(S)]
            fl = [[prep(v.frame)] + [prep(v.next()) for _ in range(2 + pwarm + psteps)] for v in vids]
            nets = [with_thresholds(converted()) for _ in range(S)]
            runners = [FrameRunner(n, f[0], "graph", st) for n, f, st in zip(nets, fl, sstreams)]
            for r, f in zip(runners, fl):
                r.prime(f[:2])
            for i in range(2, 2 + pwarm):
                for r, f in zip(runners, fl):
                    r.step(f[i])
            dt = timed_loop(runners, [f[2 + pwarm:] for f in fl], psteps, lambda: None)
            xb = [torch.cat([f[i] for f in fl], 0) for i in range(len(fl[0]))]
            dS = S * max(measure(base, xb, m, psteps, 3) for m in ("graph", "eager"))
            multi[str(S)] = {"cb_fps": S * psteps / dt, "dense_fps_batched": dS, "speedup": S * psteps / dt / dS}
            del runners, nets, fl, xb
            torch.cuda.synchronize()
        multi["how"] = ("S sequences: S converted copies with the calibrated thresholds, own video each, one hipGraph and "
                        "one HIP stream per sequence; dense: the S frames as one batch (better of graph / eager)")
    except Exception as e:      # (an add-on: never at the expense of the line)
        multi = {"error": repr(e)}
    pose_ops = workloads.openPoseDenseOps(2, Hp, Wp)
    return {
        "sequences_per_gpu": multi,
        "dense_fps": dense, "cb_fps": cb, "cb_launch": max(cb_modes, key=cb_modes.get), "speedup": cb / dense,
        "cb_own_detection_launches_fps": cb_unfolded, "own_detection_launches_speedup": cb_unfolded / dense,
        "cb_feedback_mode_fps": cbf, "feedback_speedup": cbf / dense,
        "feedback_mode_mean_ratio": sum(fratio) / max(1, len(fratio)), "feedback_mode_recomputed_gflop": fflops / 1e9,
        "feedback_mode_ratios": [round(r, 3) for r in fratio], "cb_unchained_fps": cbu,
        "unchained_speedup": cbu / dense, "cb_with_change_based_pools_fps": cbpool,
        "change_based_pools_speedup": cbpool / dense, "cb_with_change_based_pools_by_launch": cbpool_modes,
        "library_calls_per_frame_in_the_program": library_calls,
        "concurrent_branches": {"cb_fps": cbconc, "dense_fps": densec, "speedup": cbconc / densec,
                                "what": "the two branches of every stage on two HIP streams (eager), both networks"},
        "effective_gflops": cb * pose_ops / 1e9, "dense_ops_per_frame": pose_ops,
        "recomputed_gflop_per_frame": flops / 1e9,
        "mean_post_dilation_ratio": sum(rs) / max(1, len(rs)), "min_ratio": min(rs), "max_ratio": max(rs),
        "layers": len(rs), "layers_without_change": sum(1 for r in rs if r == 0.0),
        "detections_in_producer_launches": folded_det,
        "per_layer": layer_rows, "roofline": pose_roofline,
        "note": "LIVE network: 36 converted convs, fp16 (cg_half path, f16 MFMA / f32 accumulation), variance-preserving "
                "random weights, per-layer thresholds calibrated to a post-dilation change ratio of ~10 % in the running "
                "network (per_layer[]), fresh consecutive frames.  Layers of >= 64 input channels on the fp16 split-state "
                "kernels (deep ones in 4-16 k-chunks + a reduce launch, 185 channels padded to 192), the 3-channel "
                "first layer and the 1x1 layers behind a propagating producer on rounds 1-2's list kernels.  Rounds "
                "1-4 measured this configuration on nn.Conv2d's default initialisation, where the change dies out "
                "behind the fifth conv (31 idle layers): 5.5-5.9x then was launches returning early, not work"}


def secondary_configs(args):
    """BASELINE.json configs[2] (fine-grained + CBPoolMax2d, sweep of the change ratio) and configs[3] (OpenPose T=2,
    368x654, coarse-grained fp16) in the driver line -- bounded samples of what tools/sweep.py measures in full:
    three ratios of the sweep (1 / 10 / 50 % of the pixels re-drawn per frame in 16x16 blocks) for the coarse-grained
    experiment 6 network of the headline and the fine-grained experiment 7 network (in-place form), and the OpenPose
    network (LIVE: openpose_config) in the reference's two modes, each beside the dense network on the same GPU, the
    better of eager / graph-replayed launches for every network."""
    import pycbinfer
    from cbinfer_amd import workloads
    steps, warm = 40, 5

    def measure(model, frames, mode, steps=steps, warm=warm):
        runner = FrameRunner(model, frames[0], mode)
        runner.prime(frames[:2])
        for f in frames[2:2 + warm]:
            runner.step(f)
        seq = frames[2 + warm:2 + warm + steps]
        return len(seq) / timed_loop(runner, seq, len(seq), lambda: None)

    def ratios(model):
        out = []
        for m in model.modules():
            if type(m) is pycbinfer.CBConv2d and m.lastChangeIndexes() is not None:
                ci = m.lastChangeIndexes()
                out.append(round(ci.numel() / float(ci.size[0] * ci.size[1]), 3))
        return out
    out = {"config3_sweep": [], "what": secondary_configs.__doc__.replace("\n    ", " ")}
    n = 2 + warm + steps
    for ratio in (0.01, 0.10, 0.50):
        vid = workloads.SyntheticVideo(H=H, W=W, ratio=ratio, block=16, seed=7)
        frames = vid.frames(n)
        base, cg = build_bench_model(6, args.threshold)
        dense = max(measure(base, frames, m) for m in ("graph", "eager"))
        fcg = max(measure(cg, frames, m) for m in ("graph", "eager"))
        # (the same network with the 16 -> 64 contraction forced into pixel order -- the 64 -> 256 layer's detection in a
        #  launch of its own: round 5's frame -- and forced into window order: `cg` decides by itself, per sequence)
        _, cgp = build_bench_model(6, args.threshold, window_order=False)
        fcgp = max(measure(cgp, frames, m) for m in ("graph", "eager"))
        _, cgw = build_bench_model(6, args.threshold, window_order=True)
        fcgw = max(measure(cgw, frames, m) for m in ("graph", "eager"))
        del cgp, cgw
        _, fg = workloads.sceneLabelingModels(experimentIdx=7, threshold=args.threshold)
        for m in fg.modules():
            if type(m) is pycbinfer.CBConv2d:
                m.fgInPlace = True
        pycbinfer.fuseTail1x1(fg)
        ffg = max(measure(fg, frames, m) for m in ("graph", "eager"))
        # BASELINE.json configs[2] word for word: fine-grained convs + CBPoolMax2d
        _, fp = workloads.sceneLabelingModels(experimentIdx=7, threshold=args.threshold)
        for m in fp.modules():
            if type(m) is pycbinfer.CBConv2d:
                m.fgInPlace = True
        pycbinfer.insertCBPooling(fp, cloneOutput=False)
        pycbinfer.fuseTail1x1(fp)
        pycbinfer.fusePoolingIntoDetection(fp)      # (the pools in the fine-grained detections: no launch of their own)
        ffp = max(measure(fp, frames, m) for m in ("graph", "eager"))
        del fp
        out["config3_sweep"].append({"input_change": vid.ratio, "dense_fps": dense, "cg_exp6_fps": fcg,
                                     "cg_exp6_pixel_order_fps": fcgp, "cg_exp6_window_order_fps": fcgw,
                                     "cg_speedup": fcg / dense, "cg_post_dilation_ratio_per_layer": ratios(cg),
                                     "fg_exp7_inplace_fps": ffg, "fg_speedup": ffg / dense,
                                     "fg_inplace_with_cbpoolmax2d_fps": ffp,
                                     "fg_touched_ratio_per_layer": ratios(fg)})
        del base, cg, fg, frames
        torch.cuda.synchronize()
    c4 = out["config4_openpose_fp16"] = openpose_config(args, measure)
    # (single figures for the compact line: the network in the reference's structure -- dense pools, torch.cat -- and with
    #  the pools change-based + folded and the concatenation done by the library: library calls only, replayed program)
    c4["value"] = c4["cb_fps"]
    out["config4_openpose_library_only"] = {"value": c4["cb_with_change_based_pools_fps"], "unit": "frames/s",
                                            "dense_fps": c4["dense_fps"],
                                            "speedup": c4["cb_with_change_based_pools_fps"] / c4["dense_fps"]}
    out["config4_dense"] = {"value": c4["dense_fps"], "unit": "frames/s"}
    try:
        mid = [r for r in out["config3_sweep"] if abs(r["input_change"] - 0.10) < 0.02][0]
        out["config3_fg_cbpool_at_10pct"] = {"value": mid["fg_inplace_with_cbpoolmax2d_fps"], "unit": "frames/s",
                                             "dense_fps": mid["dense_fps"]}
    except Exception:
        pass
    return out



def traced_kernel_durations(step, nframes):
    """Per-kernel launch durations of `nframes` frames from the kernel trace torch.profiler records in-process (roctracer:
    the device-side begin/end timestamps rocprofv3 --kernel-trace reports).  step(i) enqueues frame i.
    -> ({kernel name: {"launches_per_frame", "avg_us"}}, busy_us_per_frame, span_us_per_frame) or (None, why)."""
    if os.environ.get("CBINFER_BENCH_NO_TRACE", "0") == "1":
        return None, "CBINFER_BENCH_NO_TRACE=1"
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or \
            "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "an external profiler is attached (its own kernel trace is the evidence)"
    try:
        from torch.profiler import profile, ProfilerActivity
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(3, 3 + nframes):
                step(i)
            torch.cuda.synchronize()
        kern = {}
        t0, t1, busy = None, None, 0.0
        for e in prof.events():
            if "cuda" not in str(getattr(e, "device_type", "")).lower():
                continue
            name = e.name
            if name.startswith("Memcpy") or name.startswith("Memset"):
                continue
            dur = float(e.time_range.end - e.time_range.start)          # microseconds
            k = kern.setdefault(name, [0, 0.0])
            k[0] += 1
            k[1] += dur
            busy += dur
            t0 = e.time_range.start if t0 is None else min(t0, e.time_range.start)
            t1 = e.time_range.end if t1 is None else max(t1, e.time_range.end)
        if not kern:
            return None, "the profiler recorded no kernels"
        out = {n: {"launches_per_frame": c / float(nframes), "avg_us": tot / c} for n, (c, tot) in kern.items()}
        return out, busy / nframes, (t1 - t0) / nframes
    except Exception as e:      # the headline must not depend on the tracer
        return None, "torch.profiler kernel trace failed: %r" % (e,)


DEFAULT_BUILD_FLAGS = (b" -O3 --offload-arch=gfx950 -fPIC -fopenmp -std=c++17 -Wall -Wno-unused-function "
                       b"-Wno-bitwise-instead-of-logical\n")


def kernel_source_hash():
    """sha256 over the kernel sources: profiles/rNN_pmc_traffic.json records it at collection time, so a
    bench line only carries `traffic` figures that were measured on the kernels it ran."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "cbinfer_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    # the compile flags the library was built with (a diagnostic build -- make EXTRA=-DCB_STAMP ... -- is another
    # library from the same sources): cbinfer_amd/csrc/Makefile keeps them in build/.flags
    # (where the build directory did not travel -- the GPU box gets the built library only -- the library is the
    #  plain build: the Makefile's default flags, tests/test_host_logic.py keeps the two in step)
    flags = os.path.join(REPO, "cbinfer_amd", "csrc", "build", ".flags")
    h.update(open(flags, "rb").read() if os.path.exists(flags) else DEFAULT_BUILD_FLAGS)
    return h.hexdigest()[:16]


def measured_traffic(kname, layer):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes
    (tools/collect_profiles.sh), or (None, why)."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no profiles/r*_pmc_traffic.json"
    pmc = json.load(open(files[-1]))
    name = os.path.basename(files[-1])
    if pmc.get("kernel_source_sha256") != kernel_source_hash():
        return None, "%s was collected on other kernel sources (%s, now %s)" % (
            name, pmc.get("kernel_source_sha256"), kernel_source_hash())
    key = layer
    if key not in pmc:
        return None, "%s has no entry %r" % (name, key)
    return pmc[key].get("bytes_per_launch"), "%s (commit %s)" % (name, pmc.get("commit", "?"))


def cpu_baseline_leg(args):
    """The oracle's port of the reference CPU path (C/OpenMP loops for detect/compact/gather/scatter/pool,
    torch CPU matmul for the contraction as conv2d_cg.py:346 does), timed on the host cores.  Sample: frame 0
    (100 % change) untimed, then steady-state frames of the SAME workload for ~10 s of CPU work (at most 400
    frames).  Runs in a process of its own without the GPU (see cpu_baseline)."""
    import numpy as np
    import pycbinfer
    from cbinfer_amd import workloads
    from oracle import cb_oracle as orc          # checker / baseline only
    budget_frames, budget_seconds = 400, 12.0

    def matmul(X, weight, bias, accMode=0):
        K = weight.shape[0]
        Y = torch.from_numpy(X).matmul(torch.from_numpy(np.ascontiguousarray(weight.reshape(K, -1))).t())
        return (Y + torch.from_numpy(bias)).numpy()

    orc.matrixMult = matmul
    base, test = workloads.sceneLabelingModels(experimentIdx=args.experiment, threshold=args.threshold,
                                               device="cpu")
    layers = []
    for m in test.children():
        if type(m) is pycbinfer.CBConv2d:
            layers.append(orc.OracleCBConv2d(m.weight.detach().numpy(), m.bias.detach().numpy(), m.threshold,
                                             withReLU=m.withReLU, feedbackLoop=m.feedbackLoop,
                                             propChangeIndexes=m.propChangeIndexes, copyInput=m.copyInput,
                                             finegrained=m.finegrained))
        elif type(m) is pycbinfer.CBPoolMax2d:
            layers.append(orc.OracleCBPoolMax2d(m.ceil_mode, m.propChangeIndexes))
        elif type(m) is torch.nn.ReLU:
            layers.append(orc.OracleReLU())
        elif type(m) is torch.nn.MaxPool2d:
            layers.append(orc.OracleMaxPool2d(m.ceil_mode))
        elif type(m) is torch.nn.Conv2d:
            layers.append(type("DenseTorch", (), {
                "forward": lambda self, x, mc=m: mc(torch.from_numpy(x)).detach().numpy(),
                "clearMemory": lambda self: None})())
        else:
            raise AssertionError(type(m))
    net = orc.OracleSequential(layers)
    vid = workloads.SyntheticVideo(device="cpu", H=H, W=W, ratio=args.ratio, block=args.block, seed=1234,
                                   pattern=args.pattern)
    # the dense baseline CNN on the same host cores (torch CPU conv2d), min of 3 (evalTools.py:33)
    xin = vid.frame.clone()
    with torch.no_grad():
        base(xin)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            base(xin)
            ts.append(time.perf_counter() - t0)
    frames = [f.numpy() for f in vid.frames(budget_frames + 1)]
    done = 0
    with torch.no_grad():
        net.forward(frames[0])
        t0 = time.perf_counter()
        for f in frames[1:]:
            net.forward(f)
            done += 1
            if time.perf_counter() - t0 > budget_seconds:
                break
        dt = time.perf_counter() - t0
    # BASELINE.md config 1 as it defines it (SURVEY 8c trap 4: experiment 6 has no CPU path in the reference):
    # pycbinfer.convert() of the baseline CNN as sceneLabeling/modelConverter.py:12-13 does -- every conv
    # change-based, ReLUs merged, feedbackLoop=False, plain nn.MaxPool2d -- on ONE frame pair: x0 untimed (100 %
    # change), x1 = x0 with 10 % of the pixels re-drawn timed, min of 3 (evalTools.py:33)
    conv1 = pycbinfer.convert(base, threshold=args.threshold)
    l1 = []
    for m in conv1.children():
        if type(m) is pycbinfer.CBConv2d:
            l1.append(orc.OracleCBConv2d(m.weight.detach().numpy(), m.bias.detach().numpy(), m.threshold,
                                         withReLU=m.withReLU, feedbackLoop=False, copyInput=True))
        elif type(m) is torch.nn.MaxPool2d:
            l1.append(orc.OracleMaxPool2d(m.ceil_mode))
        elif type(m) is torch.nn.ReLU:
            l1.append(orc.OracleReLU())
        else:
            raise AssertionError(type(m))
    net1 = orc.OracleSequential(l1)
    t1 = []
    for _ in range(3):
        net1.clearMemory()
        net1.forward(frames[0])
        t0 = time.perf_counter()
        net1.forward(frames[1])
        t1.append(time.perf_counter() - t0)
    config1 = dict(cb_pair_ms=1e3 * min(t1), dense_frame_ms=1e3 * min(ts), cb_fps=1.0 / min(t1),
                   what="BASELINE.md config 1: convert()-ed net (5 CBConv2d, feedbackLoop=False, nn.MaxPool2d), "
                        "single 480x320 frame pair @10 %, second frame timed, min of 3; oracle port")
    return dict(value=done / dt, unit="frames/s", cores=max(torch.get_num_threads(), orc.num_threads()),
                kind="port", dense_cpu_fps=1.0 / min(ts), config1=config1,
                sample="%d steady-state frames (%.1f s) of the same 480x320 sequence after an untimed "
                       "100%%-change first frame; oracle C/OpenMP ops + torch CPU matmul, own process, "
                       "OMP_WAIT_POLICY=passive" % (done, dt))


def cpu_baseline(args):
    """Run cpu_baseline_leg in a child process that never sees the GPU: the two OpenMP runtimes involved
    (the oracle's libgomp and torch's own) otherwise spin against each other on the host cores -- round 1
    measured 4 frames/s that way, 50x below the dense CPU network -- and passive waiting has to be set in the
    environment before either is loaded."""
    env = dict(os.environ)
    env.update(OMP_WAIT_POLICY="passive", GOMP_SPINCOUNT="0", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="",
               ROCR_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LD_PRELOAD"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--ratio", str(args.ratio),
           "--block", str(args.block), "--pattern", args.pattern, "--experiment", str(args.experiment),
           "--threshold", str(args.threshold)]
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
        line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:      # the headline must not depend on the baseline leg
        return {"value": None, "unit": "frames/s", "cores": None, "kind": "port",
                "sample": "cpu baseline leg failed: %r" % (e,)}


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process and
    before this process has made any GPU call, and finish with the child's exit code."""
    from cbinfer_amd.shard import visible_gpu_count
    n = visible_gpu_count()                # sysfs / environment only: no HIP call is made in this process
    shared = os.environ.get("CBINFER_ALLOW_SHARED_DEVICE", "0") == "1"
    if args.gpus > n and not (shared and n >= 1):
        # (CBINFER_ALLOW_SHARED_DEVICE=1 with CBINFER_DIST_BACKEND=gloo: rehearsal of the N-rank control flow
        #  with several ranks on one GPU -- RCCL refuses duplicate devices; never a scaling measurement)
        sys.exit("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (args.gpus, n))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_leg(args)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    from cbinfer_amd.shard import SequenceShard
    shard = SequenceShard()
    world, rank = shard.world, shard.rank
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run "
                 "--nproc-per-node %d, or without a launcher)" % (args.gpus, world, args.gpus))
    shard.device()
    barrier = shard.barrier

    import pycbinfer
    from cbinfer_amd import workloads

    # Streams for concurrent sequences are created before anything else touches the device: HIP maps
    # streams onto its few hardware queues in creation order, and sequences whose streams share a queue
    # do not overlap (measured: 7.4k instead of 8.9k frames/s with 4 sequences).
    stream_pool = [torch.cuda.Stream() for _ in range(max(args.sequences, args.multi, 1))]

    def build_sequences(S, nframes, seq0, mode, pipelined=False, variant=None):
        """S independent sequences (model + state + synthetic video + runner), primed on two frames."""
        seqs = []
        for q in range(S):
            v = dict(fuse_tail=not args.no_fuse_tail, fuse_pool=not args.no_fuse_pool,
                     pool_clone=args.pool_clone, pattern=args.pattern)
            v.update(variant or {})
            base, test = build_bench_model(v.get('experiment', args.experiment), args.threshold, v['fuse_tail'],
                                           v['fuse_pool'], v['pool_clone'])
            vid = bench_video(shard.sequence_seed(1234) + 7919 * (seq0 + q), args.ratio, args.block,
                              v['pattern'])
            # 2 priming frames + a walk of nframes frames, all resident in HBM (1.8 MB each); the timed
            # loop goes back and forth over the walk, so any number of steps sees the same change per step
            allframes = vid.frames(2 + nframes)
            if pipelined:     # cut behind the second pool: [conv, pool, conv, pool | conv, tail ...]
                cut = [i for i, m in enumerate(test.children()) if type(m) is pycbinfer.CBPoolMax2d][-1] + 1
                runner = PipelinedRunner(test, cut, None)      # (FramePipeline probes for a stream that overlaps)
            elif mode == "program":
                runner = ProgramRunner(test, allframes[0], stream_pool[q] if S > 1 else None)
            elif mode.startswith("graph") and mode != "graph":      # "graph4": four frames per replayed graph
                runner = MultiFrameRunner(test, allframes[0], int(mode[5:]), stream_pool[q] if S > 1 else None)
            else:
                runner = FrameRunner(test, allframes[0], mode, stream_pool[q] if S > 1 else None)
            runner.prime(allframes[:2])
            seqs.append(dict(base=base, test=test, vid=vid, runner=runner, frames=allframes[2:]))
        torch.cuda.synchronize()
        return seqs

    def run_sequences(S, steps, warmup, seq0, bar, mode, min_seconds=0.0, agree=None, pipelined=False,
                      variant=None):
        """Warm up, then time `steps` steps (x an integer repeat count that makes the region last
        min_seconds; all ranks agree on it through `agree`).  Returns (elapsed, steps timed, sequences)."""
        nframes = max(8, min(max(steps, warmup), 256))
        seqs = build_sequences(S, nframes, seq0, mode, pipelined, variant)
        runners, frs = [q['runner'] for q in seqs], [q['frames'] for q in seqs]
        t_warm = timed_loop(runners, frs, max(warmup, 1), lambda: None)
        reps = 1
        if min_seconds > 0:
            per_step = t_warm / max(warmup, 1)
            # (the warm-up steps run slower than the steady state: 25 % on top, so the region does not fall short)
            reps = max(1, int(math.ceil(1.25 * min_seconds / max(per_step * steps, 1e-9))))
            if agree is not None:
                reps = agree(reps)
        total = steps * reps
        for q in seqs:
            q['pos'] = max(warmup, 1) + total        # where the walk over the frames stands afterwards
        return timed_loop(runners, frs, total, bar, start=max(warmup, 1)), total, seqs

    S = max(1, args.sequences)
    # a module that keeps a reference to its input as state (copyInput=False without feedback loop, or the
    # fine-grained default form; conv2d.py:175,237-238) needs a new input tensor per frame: not capturable
    _, probe = workloads.sceneLabelingModels(experimentIdx=args.experiment, threshold=args.threshold)
    capturable = all(((m.feedbackLoop or m.copyInput) and not m.finegrained)
                     for m in probe.modules() if type(m) is pycbinfer.CBConv2d)
    # effective GFLOP/s numerator: the dense op count as the reference's own statistics define it
    # (compStats.totalInputValues per converted layer, conv2d.py:216) + the layers left dense
    for m in probe.modules():
        if type(m) is pycbinfer.CBConv2d:
            m.gatherComputationStats = True
    with torch.no_grad():
        probe(torch.zeros(1, 3, H, W, device="cuda"))
    cb_ops = sum(int(m.compStats["totalInputValues"]) for m in probe.modules() if type(m) is pycbinfer.CBConv2d)
    dense_ops = workloads.denseOps(workloads.SCENE_LABELING_SPEC, H, W)
    assert cb_ops <= dense_ops
    del probe
    # throughput mode: several sequences in flight on the one GPU, one stream + graph each;
    # reported beside the headline, which stays the reference's one-sequence-at-a-time protocol.  Measured
    # FIRST: with a single-sequence graph captured earlier in the process the same measurement comes out
    # ~17 % lower (7.4k vs 8.9k frames/s; cause not found).
    multi_result = None
    if S == 1 and world == 1 and args.multi > 1 and capturable:
        melapsed, msteps, mseqs = run_sequences(args.multi, args.steps, args.warmup, 1, lambda: None, "graph",
                                                min_seconds=args.min_seconds)
        multi_result = {"sequences_per_gpu": args.multi, "steps": msteps,
                        "value": args.multi * msteps / melapsed, "unit": "frames/s"}
        if not args.no_dense:
            dfr = [torch.cat([q['frames'][i] for q in mseqs]) for i in range(min(len(mseqs[0]['frames']), 32))]
            drunner = FrameRunner(mseqs[0]['base'], dfr[0], "graph")
            drunner.prime(dfr[:2])
            timed_loop(drunner, dfr, 3, lambda: None)
            dsteps = max(5, msteps // 8)
            multi_result["dense_fps_batched"] = args.multi * dsteps / timed_loop(
                drunner, dfr, dsteps, lambda: None)
            del drunner, dfr
        del mseqs
        torch.cuda.synchronize()
        multi_result["streams_value"] = multi_result["value"]
        multi_result["how"] = ("streams: one model copy, HIP stream and captured graph per sequence; batched: "
                               "pycbinfer.SequenceBatch -- ONE launch per step of the frame for all sequences "
                               "(own state each, shared weights; per sequence bit-identical to a run alone, "
                               "tests/test_gpu_batch.py).  value = the best batched form")
        # the same sequences through pycbinfer.SequenceBatch (experiment 5/6 networks with both fusions): ONE batch of
        # S sequences for S in {1, 2, 4, 8} (eager and replayed from a hipGraph, the better of the two), and two batches
        # of S/2 on two streams
        if args.experiment in (5, 6) and not args.no_fuse_tail and not args.no_fuse_pool:
            import pycbinfer as _pk
            nfr = max(8, min(max(args.steps, args.warmup), 256))
            Smax = 8
            bvids = [bench_video(shard.sequence_seed(1234) + 7919 * (1 + q), args.ratio, args.block, args.pattern)
                     for q in range(Smax)]
            bfr = [v.frames(2 + nfr) for v in bvids]
            per_seq_fps = multi_result["streams_value"] / args.multi

            def time_batches(groups):
                """groups: [(SequenceBatch, stream or None, its sequences' frame lists)]; eager, one host thread."""
                tot = sum(len(g[2]) for g in groups)

                def gstep(i):
                    for gsb, gstream, gfr in groups:
                        fr = [f[i] for f in gfr] if i < 2 else [f[2 + pingpong(i - 2, len(f) - 2)] for f in gfr]
                        if gstream is None:
                            gsb(fr)
                        else:
                            with torch.cuda.stream(gstream):
                                gsb(fr)
                with torch.no_grad():
                    for i in range(2 + max(args.warmup, 1)):
                        gstep(i)
                    torch.cuda.synchronize()
                    n = max(args.steps, int(math.ceil(args.min_seconds * 0.6 * per_seq_fps / max(1.0, tot ** 0.5))))
                    t0 = time.perf_counter()
                    for i in range(n):
                        gstep(2 + max(args.warmup, 1) + i)
                    torch.cuda.synchronize()
                    return tot * n / (time.perf_counter() - t0), n

            def time_graph(sb, frs):
                """One SequenceBatch replayed from a hipGraph (static input buffers)."""
                walk = [f[2:] for f in frs]
                static = [w[0].clone() for w in walk]
                with torch.no_grad():
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        sb(static)
                    torch.cuda.current_stream().wait_stream(side)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        sb(static)

                    def bstep(i):
                        for q, w in enumerate(walk):
                            static[q].copy_(w[pingpong(i, len(w))])
                        g.replay()
                    for i in range(max(args.warmup, 1)):
                        bstep(i)
                    torch.cuda.synchronize()
                    n = max(args.steps, int(math.ceil(args.min_seconds * 0.6 * per_seq_fps / max(1.0, len(frs) ** 0.5))))
                    t0 = time.perf_counter()
                    for i in range(n):
                        bstep(max(args.warmup, 1) + i)
                    torch.cuda.synchronize()
                    return len(frs) * n / (time.perf_counter() - t0), n

            batched = {}
            for Sb in (1, 2, 4, 8):
                _, bnet = build_bench_model(args.experiment, args.threshold, True, True, args.pool_clone)
                sb = _pk.SequenceBatch(bnet, Sb)
                fe, ne = time_batches([(sb, None, bfr[:Sb])])
                fg, ng = time_graph(sb, bfr[:Sb])
                batched[str(Sb)] = {"value": max(fe, fg), "eager": fe, "graph": fg, "steps": ne if fe >= fg else ng}
                if Sb == 8:
                    # where a step of eight sequences goes (in-process kernel trace) and what the 64->256 contraction
                    # reaches with all CUs filled: f32-equivalent flops over the f16-pair ceiling
                    def bstep8(i, sb=sb):
                        with torch.no_grad():
                            sb([f[2 + pingpong(i, len(f) - 2)] for f in bfr[:8]])
                    got8 = traced_kernel_durations(bstep8, 24)
                    if got8[0] is not None:
                        ks = sorted(got8[0].items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_frame"])
                        multi_result["step_of_8_kernels_us"] = {n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:72]: round(v["avg_us"] * v["launches_per_frame"], 2)
                                                                for n, v in ks[:6]}
                        multi_result["step_of_8_busy_us"] = got8[1]
                        c3 = [v["avg_us"] for n, v in got8[0].items() if "cbs_conv_kernel<128" in n.replace(" ", "")]
                        if c3:
                            # (flops: 2 N C k K of the 64->256 layer with N = the mean change-list length of one sequence
                            #  of this workload, x 8 -- the sequences differ by a few percent)
                            multi_result["contraction_64_256_us_at_8"] = c3[0]
                            multi_result["contraction_64_256_flops_note"] = (
                                "frac_of_ceiling_at_8 = 8 x the single sequence's 2 N C k K (layers[]) / this "
                                "launch / the arithmetic's ceiling; filled in below once the single-sequence layers are measured")
                del sb, bnet
                torch.cuda.synchronize()
            grouped = {}
            for Sb in (4, 8):
                half = Sb // 2
                groups = []
                from cbinfer_amd.streams import overlapping_streams
                gstreams = overlapping_streams(2)      # (two streams that share a hardware queue would take turns)
                for gi in range(2):
                    _, gnet = build_bench_model(args.experiment, args.threshold, True, True, args.pool_clone)
                    groups.append((_pk.SequenceBatch(gnet, half), gstreams[gi], bfr[gi * half:(gi + 1) * half]))
                torch.cuda.synchronize()
                fg2, ng2 = time_batches(groups)
                grouped[str(Sb)] = {"value": fg2, "steps": ng2,
                                    "how": "2 SequenceBatch groups of %d sequences, one stream each, eager" % half}
                del groups
                torch.cuda.synchronize()
            key = str(args.multi) if str(args.multi) in batched else "4"
            multi_result["batched"] = batched                      # one SequenceBatch of S sequences, S = 1, 2, 4, 8
            multi_result["grouped"] = grouped                      # two SequenceBatch groups of S/2 on two streams
            multi_result["batched_value"] = batched[key]["value"]
            multi_result["batched_launch"] = "eager" if batched[key]["eager"] >= batched[key]["graph"] else "graph"
            multi_result["grouped_value"] = grouped.get(key, {}).get("value")
            multi_result["value"] = max(batched[key]["value"], grouped.get(key, {}).get("value") or 0.0)
            multi_result["steps"] = batched[key]["steps"]
            multi_result["best_any_S"] = max([v["value"] for v in batched.values()] +
                                             [v["value"] for v in grouped.values()])
            multi_result["how"] += " or, if faster, two such batches of S/2 on two streams (grouped)"
            del bfr, bvids
            torch.cuda.synchronize()

    mode, calibration = args.mode, None
    if (mode.startswith("graph") or mode == "program") and not capturable:
        log("bench: this configuration aliases its input as state and cannot be graph-captured -> eager")
        mode = "eager"
    if mode == "auto":
        if not capturable:
            mode = "eager"
        else:
            # (each form twice, alternating, best of each: a single short run right after the other form's has come
            #  out 20 % low for no reason found)
            calibration = {}
            for cand in ("eager", "program", "graph", "graph4", "eager", "program", "graph", "graph4"):
                cel, csteps, cseqs = run_sequences(S, 100, 10, 100, lambda: None, cand, min_seconds=0.15)
                log("bench: calibration %s %.0f frames/s" % (cand, S * csteps / cel))
                calibration[cand] = max(calibration.get(cand, 0.0), S * csteps / cel)
                del cseqs
            mode = max(calibration, key=calibration.get)
            # (within 1 % of the best, the recorded launch program is preferred: the steady state is GPU-bound for both, and
            #  the program leaves the host three ctypes calls per frame instead of six modules' worth of Python -- what counts
            #  when a short timed region starts on an empty launch queue, and for eight ranks sharing one host)
            if calibration.get("program", 0.0) >= 0.99 * calibration[mode]:
                mode = "program"
            if shard.dist is not None:       # every rank must run the same launch form (rank 0's choice)
                forms = ["eager", "graph", "graph4", "program"]
                pick = forms.index(mode)
                mode = forms[(3 if shard.broadcast_flag(pick == 3, 0) else
                              (2 if shard.broadcast_flag(pick == 2, 0) else (1 if shard.broadcast_flag(pick == 1, 0) else 0)))]
    args.mode = mode

    agree = shard.agree_max

    elapsed, steps_timed, seqs = run_sequences(S, args.steps, args.warmup, 0, barrier, mode,
                                               min_seconds=args.min_seconds, agree=agree)
    base, test, vid = seqs[0]['base'], seqs[0]['test'], seqs[0]['vid']
    frames = seqs[0]['frames']

    total_frames, elapsed = shard.aggregate(steps_timed * S, elapsed, device="cuda")
    # The contract's letter: EXACTLY the K requested steps between the barriers are `value` / `steps` / `ms_per_step`.  The
    # repeated region above (the K steps repeated until the region lasts --min-seconds: a 20-step region is 2 ms) has
    # settled the clocks and is reported beside it as variants.repeated_region -- never as `value` (round 6).
    repeated = None
    literal_regions = None
    if steps_timed != args.steps and args.steps > 0:
        repeated = {"steps": steps_timed, "value": total_frames / elapsed, "ms_per_step": 1e3 * elapsed / steps_timed,
                    "timed_region_s": elapsed}
        # (the W warm-up steps once more, right in front: between the two regions the ranks exchange their results and the
        #  GPU idles for milliseconds)
        # Three such regions, the MEDIAN is `value` (each one exactly K steps between two barrier + synchronize pairs, each
        # behind its own W warm-up steps; all three in details.literal_regions_s): a 2 ms region carries 50-180 us of
        # start-up and wake-up latency that varies from shot to shot (9.9k / 10.5k / 10.5k frames/s on one box).
        runners_, frames_ = [q['runner'] for q in seqs], [q['frames'] for q in seqs]
        shots = []
        for _ in range(3):
            timed_loop(runners_, frames_, max(args.warmup, 1), lambda: None, start=seqs[0]['pos'])
            for q in seqs:
                q['pos'] += max(args.warmup, 1)
            lt = timed_loop(runners_, frames_, args.steps, barrier, start=seqs[0]['pos'])
            for q in seqs:
                q['pos'] += args.steps
            shots.append(shard.aggregate(args.steps * S, lt, device="cuda"))
        literal_regions = [e for _, e in shots]
        total_frames, elapsed = sorted(shots, key=lambda te: te[1])[1]
        steps_timed = args.steps
    fps = total_frames / elapsed

    dist_backend = shard.backend if shard.dist is not None else None
    if rank != 0:
        shard.finish()
        return

    result = {
        "metric": "frames/sec + effective GFLOP/s vs dense, scene-labeling CNN 480x320 @10% change",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": steps_timed,
        "steps_requested": args.steps, "warmup": args.warmup,
        "steps_policy": "`value` = exactly the requested steps between barrier + synchronize: the MEDIAN of three such "
                        "regions (literal_regions_s), each behind its own warm-up steps; "
                        "variants.repeated_region = the same steps repeated until the region lasts >= %g s "
                        "(--min-seconds), run just before" % args.min_seconds,
        "ms_per_step": 1e3 * elapsed / steps_timed, "timed_region_s": elapsed, "literal_regions_s": literal_regions,
        "repeated_region": repeated,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ARITH_TEXT[split_arith()],
        "data": "synthetic",
        "arithmetic": "`value` is measured on the DEFAULT arithmetic (CBINFER_ARITH=x3): f32 tensors, f32 accumulation, "
                      "and f32-EQUIVALENT products -- the 7x7 contractions of the 16- and 64-channel layers keep every "
                      "f32 operand as THREE bf16 terms (x = b0 + b1 + b2 exactly: 24 bits, f32's exponent range, no scale "
                      "and no range limit) and multiply them as the six term products above 2^-24 on the bf16 MFMA (b0 w0 "
                      "in one accumulator, the five small ones in a second); the 3->16 layer and the 1x1 tail run on the "
                      "exact f32 MFMA.  Operands as wide as the reference's sgemm (conv2d_cg.py:342-349) with fewer "
                      "accumulation roundings than its fma chain: tests/test_gpu_split.py::"
                      "test_split_hostile_data_accuracy_fullsize asserts that its error is no larger than the exact f32 "
                      "chain's on both layers.  variants.f16x2 is rounds 3-4's narrower form (f16 pairs, 22-23 bits per "
                      "operand, three products; CBINFER_ARITH=f16x2), variants.exact_f32 the f32 fma chain itself "
                      "(CBINFER_EXACT_F32=1) and variants.bf16x3 rounds 1-2's bf16-triple kernels, all measured by this run",
        "config": {"workload": "sceneLabeling CBConv2d coarse-grained fp32, synthetic 480x320 seq @%g%% "
                               "change (%s), experiment %d, %s per GPU"
                               % (100 * vid.ratio, ("%dx%d re-drawn blocks" % (args.block, args.block))
                                  if args.pattern == "blocks" else "one moving re-drawn rectangle",
                                  args.experiment,
                                  "one sequence" if S == 1 else "%d concurrent sequences (one stream each)" % S),
                   "sequences_per_gpu": S,
                   "launch": args.mode, "launch_calibration_fps": calibration, "threshold": args.threshold,
                   "pool_clone": bool(args.pool_clone),
                   "pool_fused_into_detection": not args.no_fuse_pool,
                   "tail_1x1_fused": not args.no_fuse_tail,
                   "dist_backend": dist_backend},
        # dense op count per frame: compStats.totalInputValues of the converted layers (conv2d.py:216)
        # + the 1x1 tail the experiment leaves unconverted
        "effective_gflops": fps * dense_ops / 1e9,
        "dense_ops_per_frame": dense_ops, "dense_ops_converted_layers_compstats": cb_ops,
    }

    if multi_result is not None:
        result["multi_sequence"] = multi_result
    if world == 1 and S == 1 and not args.no_pipelined and args.experiment in (5, 6):
        pel, psteps, pseqs = run_sequences(1, args.steps, args.warmup, 50, lambda: None, "eager",
                                           min_seconds=args.min_seconds, pipelined=True)
        result["pipelined"] = {"value": psteps / pel, "unit": "frames/s", "steps": psteps,
                               "what": "the same single sequence with frame pipelining (pycbinfer.FramePipeline): "
                                       "stage 2 (64->256 layer + tail) of frame t on a side stream while stage 1 of "
                                       "frame t+1 runs; outputs identical to the serial execution"}
        del pseqs
        torch.cuda.synchronize()

    # the reference's own timing protocol (poseDetection/evalTools.py:7-35; dense: sceneLabeling/eval01.py:68):
    # state cleared, frames 0..T-2 untimed, the LAST frame + device sync timed on the host, min of 3 repeats
    if world == 1 and S == 1 and not args.no_last_frame:
        from cbinfer_amd import evalTools
        fs = frames[:8]
        t_cb = evalTools.inferFramesetBenchmark(test, fs, cuda=True, numIter=3)
        t_dn = evalTools.inferFramesetBenchmark(base, fs[-1:], cuda=True, numIter=3)
        result["last_frame_ms_min3"] = {
            "cb": 1e3 * t_cb, "dense": 1e3 * t_dn, "speedup": t_dn / t_cb, "frames": len(fs),
            "protocol": "evalTools.inferFramesetBenchmark: clearMemory, frames 0..T-2 untimed, last frame + "
                        "synchronize timed with the host clock, min of 3; one eager frame incl. its launch "
                        "and sync latency (not a throughput)"}
        # (the state now is "after frames[len(fs)-1]" of a fresh run: the per-kernel section below continues
        #  the walk from there; the timed loop's captured graph refers to the state tensors clearMemory freed
        #  and is dropped)
        seqs[0]['runner'].graph = None
        seqs[0]['pos'] = len(fs)
        torch.cuda.synchronize()

    # what the structural choices of the headline configuration are worth: the same measurement on (a) the
    # reference-structured network -- change-based pool launches with a cloned output, dense torch 1x1 tail
    # (sceneLabeling/modelLoader.py:62-78 as is) -- and (b) the video as ONE moving rectangle of the same area
    if world == 1 and S == 1 and not args.no_variants:
        result["variants"] = {}
        for name, var in (("f16x2", dict(env=dict(CBINFER_ARITH="f16x2"))),
                          ("exact_f32", dict(env=dict(CBINFER_EXACT_F32="1"))),
                          ("bf16x3", dict(env=dict(CBINFER_ARITH="bf16x3"))),
                          ("reference_structured", dict(fuse_tail=False, fuse_pool=False, pool_clone=True)),
                          ("no_feedback_experiment2", dict(experiment=2, fuse_tail=False, fuse_pool=False)),
                          ("pattern_region", dict(pattern="region"))):
            if name == "pattern_region" and args.pattern == "region":
                continue
            var = dict(var)
            env = var.pop("env", {})
            saved_env = {k: os.environ.get(k) for k in env}
            os.environ.update(env)          # (the modules read these switches when a frame is dispatched)
            best = None
            for vmode in (("graph", "eager") if capturable else ("eager",)):
                vel, vsteps, vseqs = run_sequences(1, args.steps, args.warmup, 200, lambda: None, vmode,
                                                   min_seconds=min(args.min_seconds, 0.25), variant=var)
                if best is None or vsteps / vel > best[0]:
                    best = (vsteps / vel, vmode, vsteps)
                del vseqs
                torch.cuda.synchronize()
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            result["variants"][name] = {"value": best[0], "unit": "frames/s", "launch": best[1],
                                        "steps": best[2], "differs_by": dict(var, **env)}
            if name == "exact_f32":
                result["variants"][name]["arithmetic"] = (
                    "every contraction on the exact f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32: an f32 fma chain, "
                    "bitwise), list kernels of rounds 1-2: the f32 number as the reference's sgemm defines it")
            if name == "no_feedback_experiment2":
                result["variants"][name]["what"] = (
                    "sceneLabeling/modelLoader.py:45-47: what convert() makes -- feedbackLoop=False, copyInput=True, "
                    "nn.MaxPool2d, dense 1x1 layers -- the 16->64 and 64->256 layers on the split-state kernels with "
                    "every value of the frame copied into both states by the detection (CBINFER_SPLIT_COPY_ALL, round 4)")
            if name == "bf16x3":
                result["variants"][name]["arithmetic"] = (
                    "f32 operands as three bf16 terms (24 bits), the six products above 2^-24, f32 accumulation: "
                    "f32-equivalent; rounds 1-2's patch-staged and list kernels")
            if name == "f16x2":
                result["variants"][name]["arithmetic"] = (
                    "f16 PAIRS on the split-state kernels (rounds 3-4's default): 22-23 significant bits per operand, "
                    "three products -- NARROWER than f32, reported for comparison only")

    # dense network on the same GPU, timed the same way (eval01.py:68) -- in MIOpen's default find mode AND with its
    # exhaustive find (torch.backends.cudnn.benchmark = True; the find runs in the untimed priming frames): `dense_fps`
    # is the better of the two, so the speed-up is quoted against the fastest dense network this GPU offers (round 6)
    if not args.no_dense and world == 1:
        # (with S sequences per GPU the dense network gets them as one batch of S frames)
        dframes = [torch.cat([q['frames'][i] for q in seqs]) for i in range(min(len(frames), 64))]
        dense = {}
        saved_bench = torch.backends.cudnn.benchmark
        for find in ("default", "exhaustive"):
            torch.backends.cudnn.benchmark = find == "exhaustive"
            try:
                for dmode in ("graph", "eager"):        # the dense network gets the better of the two as well
                    drunner = FrameRunner(base, dframes[0], dmode)
                    drunner.prime(dframes[:2])
                    t5 = timed_loop(drunner, dframes, 5, lambda: None)
                    dsteps = max(10, int(math.ceil(args.min_seconds / (t5 / 5))))
                    dense[(find, dmode)] = S * dsteps / timed_loop(drunner, dframes, dsteps, lambda: None)
                    del drunner
            except Exception as e:      # (the find is an add-on: never at the expense of the line)
                log("bench: dense network, find mode %s: %r" % (find, e))
        torch.backends.cudnn.benchmark = saved_bench
        best_dense = max(dense, key=dense.get)
        result["dense_fps"] = dense[best_dense]
        result["dense_find_mode"], result["dense_launch"] = best_dense
        result["dense_fps_by_mode"] = {"%s/%s" % k: v for k, v in dense.items()}
        result["speedup_vs_dense"] = fps / result["dense_fps"]
        result["dense_tflops"] = result["dense_fps"] * dense_ops / 1e12
        # the same machinery without the sparsity: the converted network with every pixel changed every frame (threshold
        # -1: |d| > -1 holds for every value) -- what the change-based kernels cost when there is nothing to skip
        try:
            _, allnet = build_bench_model(args.experiment, -1.0, not args.no_fuse_tail, not args.no_fuse_pool,
                                          args.pool_clone)
            best_all = 0.0
            for amode in (("graph", "eager") if capturable else ("eager",)):
                arunner = FrameRunner(allnet, frames[0], amode)
                arunner.prime(frames[:2])
                t3 = timed_loop(arunner, frames, 3, lambda: None)
                asteps = max(10, int(math.ceil(min(args.min_seconds, 0.25) / (t3 / 3))))
                best_all = max(best_all, asteps / timed_loop(arunner, frames, asteps, lambda: None))
                del arunner
            result["dense_on_cb_kernels_fps"] = best_all
            del allnet
            torch.cuda.synchronize()
        except Exception as e:
            result["dense_on_cb_kernels_fps"] = None
            log("bench: dense_on_cb_kernels: %r" % (e,))

    # per-kernel measurement (HIP events on the launch stream) -> roofline of the dominant kernel
    if world == 1:
        got = inframe_layer_times(test, frames, seqs[0]['pos'])
        if got is not None:
            test_rows, _, pair_us = got
            timing = ("in-frame: HIP events around the library call (one or two launches) inside the eager frame, "
                      "one call bracketed per pass; event_pair_ms = an empty event pair at the same place -- the "
                      "upper bound of what the bracket itself adds (not subtracted); mean N")
            result["layers"] = [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()}
                                for r in test_rows]
            result["layers_timing"] = timing
            result["event_pair_ms"] = round(pair_us * 1e-3, 5)
            # the launches are parts of the frame: their in-frame durations -- each of which carries up to one
            # event pair of bracket overhead -- must fit into one step plus those overheads
            n_meas = sum(("conv_ms" in r) * (1 if r.get("detect_in_producer_launch") else 2) +
                         ("tail_ms" in r and "folded" not in r) for r in test_rows)
            for r in test_rows:
                if r.get("detect_in_producer_launch"):
                    r["detect_ms"] = 0.0          # (no launch: the bracket held nothing)
            tot = sum(r.get("conv_ms", 0.0) + r.get("detect_ms", 0.0) + r.get("tail_ms", 0.0) for r in test_rows)
            result["layers_check"] = {"sum_ms_in_frame": round(tot, 5), "measurements": n_meas,
                                      "ms_per_step": round(result["ms_per_step"], 5),
                                      "consistent": bool(tot - n_meas * pair_us * 1e-3 <= result["ms_per_step"])}
            # The launches of a frame run back to back (the kernel trace shows no gaps: profiles/r0N_frame_timeline.txt),
            # so their durations sum to the frame period.  Every bracketed figure carries the bracket's own cost (the
            # empty pair) and what the two event packets cost the neighbouring launches; net of the empty pair and scaled
            # so that the sum over the frame's launches equals the measured ms_per_step they are the launch durations a
            # kernel trace reports (profiles/r0N_bench_kernel_clusters.txt, within a few percent).
            meas = []
            for r in test_rows:
                for k in ("detect_ms", "conv_ms", "tail_ms"):
                    if k in r and not (k == "tail_ms" and "folded" in r) and \
                            not (k == "detect_ms" and r.get("detect_in_producer_launch")):
                        meas.append((r, k))
            net = [max(r[k] * 1e3 - pair_us, 0.2) for r, k in meas]
            scale = result["ms_per_step"] * 1e3 / max(sum(net), 1e-9)
            for (r, k), v in zip(meas, net):
                r[k.replace("_ms", "_us_in_frame")] = v * scale
            for r in test_rows:
                if "detect_us_in_frame" in r:
                    r["detect_GBps"] = r["detect_bytes"] / (r["detect_us_in_frame"] * 1e-6) / 1e9
                    r["detect_frac_hbm"] = r["detect_GBps"] / HBM_PEAK_GBS
                if "conv_us_in_frame" in r:
                    r["conv_TFLOPs"] = r["conv_flops"] / (r["conv_us_in_frame"] * 1e-6) / 1e12
            result["layers"] = [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()}
                                for r in test_rows]
            result["layers_in_frame_note"] = ("*_us_in_frame: the bracketed time net of the empty event pair, scaled by "
                                              "%.3f so that the frame's launches sum to ms_per_step (they run back to "
                                              "back); *_ms: as bracketed" % scale)
            # Kernel durations as a kernel trace reports them (device timestamps; what rocprofv3 --kernel-trace --stats
            # of this command shows: profiles/r0N_bench_kernel_stats.csv), recorded in-process over 60 frames of the same
            # walk: the roofline of the dominant launch is computed from THESE; the event-bracketed figures above stay
            # as the cross-check they are.
            runner0 = seqs[0]['runner']
            runner0.graph = None
            pos0 = [seqs[0]['pos'] + 200]
            with torch.no_grad():
                for i in range(4):          # (the in-frame pass above left the walk somewhere else: settle)
                    test(frames[pingpong(pos0[0] + i, len(frames))])

            def traced_step(i):
                with torch.no_grad():
                    test(frames[pingpong(pos0[0] + 4 + i, len(frames))])
            traced = traced_kernel_durations(traced_step, 60)
            ktrace = traced[0]
            if ktrace is not None:
                # Tracing itself stretches the launches: the traced frames' span per frame is a few percent longer than
                # the un-traced frame period, all of it in the kernel durations (the traced stream has no gaps to speak
                # of).  The durations are therefore scaled by ms_per_step / span: then the traced frame fits the period
                # the timed loop measured, and the figures agree with `rocprofv3 --kernel-trace --stats` of the same
                # command (profiles/r0N_bench_kernel_clusters.txt), whose host-bound run leaves the kernels unstretched.
                tscale = min(1.0, result["ms_per_step"] * 1e3 / max(traced[2], 1e-9))
                result["kernel_trace"] = {
                    "kernels_as_traced": {n: {k: round(v, 3) for k, v in d.items()} for n, d in sorted(ktrace.items())},
                    "busy_us_per_frame_as_traced": round(traced[1], 3), "span_us_per_frame_as_traced": round(traced[2], 3),
                    "scale_to_untraced_period": round(tscale, 4),
                    "how": "torch.profiler (roctracer) kernel trace of 60 eager frames of the timed walk, in-process; "
                           "the *_us_kernel_trace figures of layers[] and roofline.avg_duration_us are the traced "
                           "durations x scale_to_untraced_period (= ms_per_step / traced span per frame: tracing "
                           "stretches the launches by a few percent)"}
                ktrace = {n: dict(d, avg_us=d["avg_us"] * tscale) for n, d in ktrace.items()}
            else:
                result["kernel_trace"] = {"kernels": None, "why": traced[1]}
            if ktrace is not None:
                # per layer: the launch durations the trace reports (what rocprofv3 --kernel-trace --stats shows)
                def tr(*pats):
                    for n, d in ktrace.items():
                        nn = n.replace(" ", "")
                        if all(p_ in nn for p_ in pats):
                            return d
                    return None
                for r in test_rows:
                    if "conv_kernel" not in r:
                        continue
                    ck = r["conv_kernel"]
                    if "cbp_rowpair" in ck:
                        cd, dd = tr("cbp_rowpair_kernel"), tr("cb_detect_kernel<float,true,false>")
                    elif "cb_rowconv" in ck:
                        cd, dd = tr("cb_rowconv_f32_kernel"), tr("cb_detect_kernel<float,true,false>")
                    elif "split-state" in ck:
                        big = "cbs_reduce" in ck
                        cd = tr("cbs_conv_kernel<128,") if big else tr("cbs_conv_kernel<64,")
                        dd = None if r.get("detect_in_producer_launch") else tr("cbs_detect_kernel<true>")
                        if big and cd is not None:
                            r2 = tr("cbs_reduce")
                            cd = dict(cd, avg_us=cd["avg_us"] + (r2["avg_us"] if r2 else 0.0))
                    else:
                        cd = dd = None
                    if cd is not None:
                        r["conv_us_kernel_trace"] = cd["avg_us"]
                        r["conv_TFLOPs"] = r["conv_flops"] / (cd["avg_us"] * 1e-6) / 1e12
                    if dd is not None:
                        r["detect_us_kernel_trace"] = dd["avg_us"]
                        r["detect_GBps"] = r["detect_bytes"] / (dd["avg_us"] * 1e-6) / 1e9
                        r["detect_frac_hbm"] = r["detect_GBps"] / HBM_PEAK_GBS
                result["layers"] = [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()}
                                    for r in test_rows]
                result["layers_in_frame_note"] += ("; *_us_kernel_trace: the launch durations of the in-process kernel "
                                                   "trace -- detect_GBps / conv_TFLOPs are computed from these where "
                                                   "present")
            best = max((r for r in test_rows if "conv_ms" in r), key=lambda r: r["conv_ms"], default=None)
            ms = result.get("multi_sequence")
            if best is not None and ms and ms.get("contraction_64_256_us_at_8") and "64->256" in best["layer"]:
                ms["contraction_64_256_frac_of_ceiling_at_8"] = (
                    8.0 * best["conv_flops"] / (ms["contraction_64_256_us_at_8"] * 1e-6) / 1e12 / split_ceiling())
            if best is not None:
                r = best
                split = "split-state" in r["conv_kernel"]
                exact = "cb_mfma_f32_kernel" == r["conv_kernel"]
                dur_us, dur_src = r["conv_us_in_frame"], "HIP events (layers[].conv_us_in_frame)"
                first_us = None
                if ktrace is not None and split:
                    # the launches of this contraction: the 128-row-tile kernel and the second launch behind it
                    mine = [d for n, d in ktrace.items() if ("cbs_conv_kernel<128" in n.replace(" ", "") or
                                                             "cbs_reduce" in n)]
                    if mine:
                        dur_us = sum(d["avg_us"] * min(d["launches_per_frame"], 1.0) for d in mine)
                        dur_src = "kernel trace: " + " + ".join("%.2f us" % d["avg_us"] for d in mine)
                        firsts = [d["avg_us"] for n, d in ktrace.items() if "cbs_conv_kernel<128" in n.replace(" ", "")]
                        first_us = firsts[0] if firsts else None
                ach = r["conv_flops"] / (dur_us * 1e-6) / 1e12
                ceiling = split_ceiling() if split else (FP32_MFMA_PEAK_TFLOPS if exact else BF16X3_CEILING_TFLOPS)
                traffic, traffic_src = measured_traffic("conv", r["layer"])
                result["roofline"] = {
                    "kernel": r["conv_kernel"] + " (fused gather->MFMA->scatter), " + r["layer"],
                    "bound": "mfma", "achieved": ach, "peak": ceiling, "unit": "TFLOP/s",
                    "frac": ach / ceiling,
                    "peak_note": "`achieved` = f32-equivalent flops (2 N C k K) of the contraction per launch pair / its "
                                 "in-frame duration; `peak` = the ceiling of the unit that executes them: %s.  The ratio "
                                 "to the f32 MFMA peak (the tensors' dtype) is kept as frac_of_f32_mfma_peak -- it can "
                                 "exceed what that unit could do, because that unit does none of the work"
                                 % (("%.1f TFLOP/s = 2.5 PFLOP/s of bf16 MFMA / 6 products per multiply (bf16 triples)"
                                     % BF16X3_CEILING_TFLOPS) if (split and split_arith() == "x3") else
                                    "%.1f TFLOP/s = 2.5 PFLOP/s of f16 MFMA / 3 products per multiply (f16 pairs)"
                                    % F16X2_CEILING_TFLOPS if split else
                                    "157.3 TFLOP/s exact f32 MFMA" if exact else
                                    "%.1f TFLOP/s = 2.5 PFLOP/s of bf16 MFMA / 6 products per multiply (bf16x3)"
                                    % BF16X3_CEILING_TFLOPS),
                    "f32_mfma_peak": FP32_MFMA_PEAK_TFLOPS, "frac_of_f32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                    # (the contraction launch alone -- the second launch also scatters the layer's outputs and
                    #  evaluates the 1x1 tail, work `achieved` does not count)
                    "frac_first_launch_alone": (r["conv_flops"] / (first_us * 1e-6) / 1e12 / ceiling) if first_us else None,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_duration_us": dur_us, "avg_duration_source": dur_src,
                    "avg_duration_us_hip_events_bracketed": r["conv_ms"] * 1e3,
                    "avg_duration_us_hip_events_in_frame": r["conv_us_in_frame"], "event_pair_us": pair_us,
                    "duration_timing": "kernel trace recorded in-process (device timestamps) where available, else: "
                                       + timing,
                    "units_per_launch": r["N"],
                    "launches": "a deep contraction is two launches: the partial tiles of a k-range split over "
                                "workgroups (short change lists) are summed by the second one, which -- when the fused "
                                "1x1 tail follows the layer -- also evaluates that tail on the finished columns "
                                "(0.12 GFLOP here, not counted in `achieved`); avg_duration_us and traffic cover both "
                                "launches"}
            if args.breakdown:
                for r in test_rows:
                    log(json.dumps(r))

    # every layer in isolation at a post-dilation change ratio of exactly 10 %
    if world == 1 and S == 1 and not args.no_isolated and args.experiment in (4, 5, 6):
        try:
            result["isolated_layers"] = isolated_layers(test, ratio=args.ratio)
            result["isolated_layers_note"] = isolated_layers.__doc__.split("\n\n")[0].replace("\n    ", " ")
        except Exception as e:      # the headline must not depend on it
            result["isolated_layers"] = "failed: %r" % (e,)

    # BASELINE.json configs[2] and [3] (bounded samples; tools/sweep.py has the full tables)
    if world == 1 and S == 1 and not args.no_secondary:
        try:
            sec = secondary_configs(args)
            result.setdefault("variants", {}).update(sec)
        except Exception as e:      # the headline must not depend on it
            result.setdefault("variants", {})["secondary_failed"] = repr(e)

    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(args)

    emit(result)
    shard.finish()


if __name__ == "__main__":
    main()
