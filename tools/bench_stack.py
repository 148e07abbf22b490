#!/usr/bin/env python3
"""A generator's inner layers as ONE persistent stack launch against the same layers as separate launches, each captured into a
hipGraph and replayed back to back (one stream):  python tools/bench_stack.py [batch] [model: pix2pix|cyclegan]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, GeneratorNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
norm = 'instancenorm' if (len(sys.argv) > 2 and sys.argv[2] == 'cyclegan') else 'batchnorm'
res = {}
for stacks in (1, 0):
    L.set_option('conv.stack', stacks)
    ctx = Ctx('cuda:0', 'bf16')
    G = GeneratorNet(ctx, 1, norm, seed=1)
    g = G.new_call(B, 256, dropout=True)
    g.xin.t.copy_(torch.randn_like(g.xin.t.float()).to(ctx.tdtype))
    g.forward(); g.dgen.t.normal_(); g.backward(use_dgen2=False)
    torch.cuda.synchronize()
    fwd = g.fwd_ops
    names = [o[2] + ((' ' + o[3]['kernel']) if len(o) > 3 and isinstance(o[3], dict) else '') for o in fwd]
    # the inner segment: from the first stack op (or the first op that a stack would have replaced) to the last
    if stacks:
        idx = [i for i, o in enumerate(fwd) if o[2] == 'conv_stack']
        seg = fwd[idx[0]:idx[-1] + 1]
        nlay = sum(len(o[3]['keep'][1]) for o in seg if o[2] == 'conv_stack')
        first_shape = seg[0][3]['keep'][1][0][3]['shape']
        res['first'], res['n'] = first_shape, nlay
    else:
        i0 = [i for i, o in enumerate(fwd) if len(o) > 3 and isinstance(o[3], dict) and o[3].get('shape') == res['first']][0]
        seg, cnt = [], 0
        for o in fwd[i0:]:
            seg.append(o)
            if len(o) > 3 and isinstance(o[3], dict) and 'stack' in o[3]:
                cnt += 1
                if cnt == res['n']:
                    break
    print(f"stacks={stacks}: segment of {len(seg)} launches: {[o[2] for o in seg]}")
    gr = ctx.capture_graph(lambda: ctx.run(seg))
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    ctx.assert_no_stack_timeout()
    print(f"   {e0.elapsed_time(e1) / 300 * 1e3:.1f} us per pass over the segment ({res['n']} layers)")
