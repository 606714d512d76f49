"""cbinfer_amd/streams.py: the side stream of FramePipeline / OpenPoseModel(concurrentBranches) is one whose kernels
really run beside the caller's (HIP's stream -> hardware queue mapping depends on the process's stream count)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_side_stream_overlaps_with_the_callers_stream_whatever_was_created_before():
    from cbinfer_amd import streams
    junk = [torch.cuda.Stream() for _ in range(5)]      # (shift the creation-order mapping)
    s = streams.side_stream()
    assert s.cuda_stream != torch.cuda.current_stream().cuda_stream
    assert streams.overlaps(torch.cuda.current_stream(), s)
    assert streams.side_stream() is s                    # cached per (device, caller stream)
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):
        s2 = streams.side_stream()
        assert streams.overlaps(other, s2)
    del junk
    a, b = streams.overlapping_streams(2)
    assert streams.overlaps(a, b)


def test_pipeline_and_forked_branches_use_it():
    import pycbinfer
    from cbinfer_amd import streams, workloads
    import bench
    _, net = bench.build_bench_model()
    cut = [i for i, m in enumerate(net.children()) if type(m) is pycbinfer.CBPoolMax2d][-1] + 1
    pipe = pycbinfer.FramePipeline(net, cut)
    frames = bench.bench_video(5).frames(4)
    outs = [pipe.submit(f) for f in frames]
    pipe.wait()
    assert pipe.side is streams.side_stream(frames[0].device)
    assert all(torch.isfinite(o).all() for o in outs)
    pipe.close()
    assert workloads._side_stream(frames[0].device) is pipe.side
