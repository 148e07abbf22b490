#!/usr/bin/env python3
"""Model-free probe of hipGraph fork/join patterns on this runtime (ROCm 7.2, torch 2.10), each scenario in its own child
process so that a crash inside hipStreamEndCapture is a result, not the end of the probe.  DESIGN.md section 5 quotes it.

  nested   origin -> lane A -> lane B (B forked from the forked lane A), joined B -> A -> origin
  nested2  the same, but B joined into A AND into the origin stream
  flat     origin -> A, origin -> B, cross-wait A <-> B through events recorded inside the capture, both joined into origin
  outside  a lane waits on an event recorded OUTSIDE the capture (the hipErrorStreamCaptureIsolation of gpurun_out/r3/cg3_d32.log)
  two      origin -> A, origin -> B, no cross-wait, both joined into origin
  *_tail   the same scenario with one more kernel on the origin stream after the joins (what every product schedule has)
  lane_edge[_rev]  origin -> A, origin -> B, one edge A -> B between the forked lanes, joined A then B (or B then A)
  raise    a lane is forked, a Python exception is raised, Ctx.capture_graph() joins the lane and ends the capture
"""
import subprocess
import sys

CHILD = r'''
import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, %(root)r)
which = %(which)r
dev = torch.device('cuda:0')
buf = [torch.zeros(1 << 16, device=dev) for _ in range(3)]
A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if which == 'raise':
    from gan_amd.nets import Ctx
    from gan_amd import _lib as L
    ctx = Ctx('cuda:0', 'bf16', workspace_mb=16)
    def body():
        lane = ctx.lane_stream(3)
        lane.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(lane):
            buf[1].add_(1.0)
        raise ValueError("boom inside the capture")
    try:
        ctx.capture_graph(body)
        print("RESULT raise: no exception?!")
    except L.GanAmdError as e:
        print("RESULT raise: GanAmdError:", str(e)[:80])
    torch.cuda.synchronize()
    sys.exit(0)
ev_out = torch.cuda.Event()
if which == 'outside':
    with torch.cuda.stream(B):
        buf[2].add_(1.0)
        ev_out.record(B)
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        buf[0].add_(1.0)
        if which in ('nested', 'nested2', 'nested_tail'):
            A.wait_stream(cur)
            with torch.cuda.stream(A):
                buf[1].add_(1.0)
                B.wait_stream(A)
                with torch.cuda.stream(B):
                    buf[2].add_(1.0)
                A.wait_stream(B)
                buf[1].add_(1.0)
            if which == 'nested2':
                cur.wait_stream(B)
            cur.wait_stream(A)
        elif which in ('flat', 'flat_tail'):
            A.wait_stream(cur); B.wait_stream(cur)
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(A):
                buf[1].add_(1.0); ea.record(A)
            with torch.cuda.stream(B):
                buf[2].add_(1.0); eb.record(B)
            A.wait_event(eb); B.wait_event(ea)
            with torch.cuda.stream(A):
                buf[1].add_(1.0)
            with torch.cuda.stream(B):
                buf[2].add_(1.0)
            cur.wait_stream(A); cur.wait_stream(B)
        elif which in ('two', 'two_tail'):           # two forked lanes, no cross-wait
            A.wait_stream(cur); B.wait_stream(cur)
            with torch.cuda.stream(A):
                buf[1].add_(1.0)
            with torch.cuda.stream(B):
                buf[2].add_(1.0)
            cur.wait_stream(A); cur.wait_stream(B)
        elif which in ('lane_edge', 'lane_edge_rev'):  # both forked from the origin; ONE dependency edge A -> B between the lanes
            A.wait_stream(cur); B.wait_stream(cur)
            with torch.cuda.stream(A):
                buf[1].add_(1.0)
            B.wait_stream(A)
            with torch.cuda.stream(B):
                buf[2].add_(1.0)
            if which == 'lane_edge':
                cur.wait_stream(A); cur.wait_stream(B)
            else:
                cur.wait_stream(B); cur.wait_stream(A)
        elif which == 'outside':
            A.wait_stream(cur)
            try:
                A.wait_event(ev_out)
                print("NOTE outside: wait_event on an uncaptured event was accepted")
            except Exception as e:
                print("NOTE outside: wait_event raised", type(e).__name__, str(e).splitlines()[0][:90])
            try:
                cur.wait_stream(A)
            except Exception as e:
                print("NOTE outside: join raised", type(e).__name__, str(e).splitlines()[0][:90])
        if which.endswith('_tail'):
            buf[0].add_(1.0)                           # a kernel on the origin stream AFTER the joins
    g.replay(); torch.cuda.synchronize()
    print("RESULT", which, ": captured and replayed; buffers", [float(b[0]) for b in buf])
except Exception as e:
    print("RESULT", which, ": exception", type(e).__name__, str(e).splitlines()[0][:120])
'''


def main():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for which in sys.argv[1:] or ['flat', 'nested', 'nested2', 'outside', 'raise']:
        p = subprocess.run([sys.executable, '-c', CHILD % dict(root=root, which=which)], capture_output=True, text=True, timeout=300)
        lines = [ln for ln in (p.stdout + p.stderr).splitlines() if ln.startswith(('RESULT', 'NOTE', 'Fatal', '  File'))][:8]
        print(f"[{which}] rc={p.returncode}", *lines, sep='\n    ', flush=True)


if __name__ == '__main__':
    main()
