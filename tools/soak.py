#!/usr/bin/env python3
"""Soak test: a long synthetic sequence through the change-based network at threshold 0 (every changed
input value propagates, so the output must track the dense network all the time), in the bench's execution
options, several sequences concurrently.  Any lost update (a race in the split-K hand-off, the mask parity
protocol, the call plan ...) shows up as a growing difference to the dense network.
usage: soak.py [frames] [sequences]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pycbinfer  # noqa: E402
from cbinfer_amd import workloads  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    seqs = []
    for q in range(S):
        base, test = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.0, seed=q)
        for m in test.modules():
            if type(m) is pycbinfer.CBPoolMax2d:
                m.cloneOutput = False
        pycbinfer.fuseTail1x1(test)              # (the tail in the contraction's second launch, as the bench runs it)
        pycbinfer.fusePoolingIntoDetection(test)
        pycbinfer.fuseDetectionIntoProducer(test)    # (the 16->64 layer's pooled detection in the 3->16 layer's launch)
        seqs.append((base, test, workloads.SyntheticVideo(H=320, W=480, ratio=0.05 + 0.05 * q, block=32,
                                                          seed=100 + q), torch.cuda.Stream()))
    torch.cuda.synchronize()
    worst = 0.0
    with torch.no_grad():
        for t in range(T):
            outs = []
            for base, test, vid, st in seqs:
                f = vid.frame if t == 0 else vid.next()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append((test(f), base, f))
            if t % 100 == 0 or t == T - 1:
                torch.cuda.synchronize()
                err = max((o - b(f)).abs().max().item() for o, b, f in outs)
                worst = max(worst, err)
                print("frame %5d  max |cb - dense| = %.3e" % (t, err), flush=True)
    torch.cuda.synchronize()
    print("worst %.3e over %d frames x %d sequences" % (worst, T, S))
    assert worst <= 1e-4
    # the same through ONE SequenceBatch (one launch per step for all sequences) at the bench's threshold against
    # independent networks, bit for bit
    import bench
    nets = [bench.build_bench_model()[1] for _ in range(S)]
    batch = pycbinfer.SequenceBatch(bench.build_bench_model()[1], S)
    vids = [bench.bench_video(500 + q, ratio=0.05 + 0.05 * q) for q in range(S)]
    bad = 0
    with torch.no_grad():
        for t in range(T // 4):
            frames = [v.frame if t == 0 else v.next() for v in vids]
            outs = batch(frames)
            for q in range(S):
                y = nets[q](frames[q])
                if t % 25 == 0 or t == T // 4 - 1:
                    bad += int(not torch.equal(outs[q], y))
    torch.cuda.synchronize()
    print("SequenceBatch vs independent networks over %d steps: %d mismatching outputs" % (T // 4, bad))
    assert bad == 0


if __name__ == "__main__":
    main()
