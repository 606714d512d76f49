import sys, os
sys.path.insert(0, '/root/repo')
import torch, pycbinfer
from cbinfer_amd import workloads
mode = sys.argv[1]   # graph-conc, graph-serial, eager-conc
S, T = 3, 6
H, W = 160, 240
vids = [workloads.SyntheticVideo(H=H, W=W, ratio=0.1, block=16, seed=11 + q).frames(T) for q in range(S)]
ref_out = []
with torch.no_grad():
    for q in range(S):
        _, eager = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.05, seed=1)
        outs = []
        for f in vids[q]:
            y = eager(f).clone()
            cs = [mm for mm in eager.modules() if type(mm) is pycbinfer.CBConv2d]
            outs.append((y, [(c.prevInput.clone(), c.prevOutput.clone(), c.lastChangeIndexes().tensor().clone()) for c in cs]))
        ref_out.append(outs)
    torch.cuda.synchronize()
    runners = []
    for q in range(S):
        _, m = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.05, seed=1)
        st = torch.cuda.Stream()
        static_in = vids[q][0].clone()
        st.wait_stream(torch.cuda.current_stream())
        g = out = None
        with torch.cuda.stream(st):
            out = m(static_in)
            if mode.startswith('graph'):
                cap = torch.cuda.Stream()
                cap.wait_stream(st)
                with torch.cuda.stream(cap):
                    m(static_in)
                st.wait_stream(cap)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    out = m(static_in)
        runners.append((st, static_in, g, out, m))
    torch.cuda.synchronize()
    worst = 0
    for t in range(1, T):
        for q, (st, static_in, g, out, m) in enumerate(runners):
            with torch.cuda.stream(st):
                static_in.copy_(vids[q][t])
                if g is not None:
                    g.replay()
                else:
                    runners[q] = (st, static_in, g, m(static_in), m)
            if mode.endswith('serial'):
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for q, (st, static_in, g, out, m) in enumerate(runners):
            e = (out - ref_out[q][t][0]).abs().max().item()
            worst = max(worst, e)
            if e > 1e-4:
                convs = [mm for mm in m.modules() if type(mm) is pycbinfer.CBConv2d]
                print(mode, 'q', q, 't', t, 'err', e)
                for li, (c, (pi, po, ix)) in enumerate(zip(convs, ref_out[q][t][1])):
                    idx = c.lastChangeIndexes().tensor()
                    dpo = (c.prevOutput - po).abs()
                    bad = (dpo.amax(dim=(0, 1)) > 1e-4).nonzero()
                    print('   layer', li, 'prevInput equal', torch.equal(c.prevInput, pi), 'list equal',
                          idx.numel() == ix.numel() and torch.equal(idx, ix), idx.numel(), ix.numel(),
                          'prevOutput maxdiff %.3g' % dpo.max().item(), 'bad pixels', bad.shape[0],
                          bad[:6].tolist(), 'bad channels', (dpo.amax(dim=(0, 2, 3)) > 1e-4).nonzero().flatten()[:10].tolist())
                sys.exit(1)
print(mode, 'worst', worst)
