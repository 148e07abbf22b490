#!/usr/bin/env python3
"""Headline benchmark: Pix2Pix `train_step` images/sec at 256x256 (BASELINE.json), bf16, batch 16 per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; started WITHOUT a
   launcher, `--gpus N` spawns that command itself, one rank per GPU, before the parent touches a GPU)

A "step" is one full Pix2Pix.train_step(training=True): generator forward, discriminator forward on
real++fake, BCE/L1 losses, both backward passes (dgrad + wgrad), TF-form Adam on every weight, and (N>1)
the RCCL gradient all-reduce - the whole step replayed from a captured hipGraph, inputs resident in HBM.
Prints ONE JSON line (rank 0) with the step throughput, the roofline of the dominant GEMM kernel measured
live with HIP events, and the CPU oracle timed on this box's host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic FLOPs per image of one Pix2Pix train_step = 3*F_G + 7*F_D (SURVEY.md 8d / BASELINE.md section 2)
GF_PER_IMG = {256: 79.51, 512: 321.72}
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def gemm_profile(step, inputs, reps=5):
    """Eager (non-graph) instrumented passes: HIP events around every GEMM-type launch, on the stream the
    kernels run on.  Returns {kernel symbol: (total_ms_per_step, total_flops_per_step, launches_per_step)}."""
    ctx = step.ctx
    recs = {}
    orig_run = ctx.run

    def timed_run(ops, lane=0):
        st = ctx.stream()
        for op in ops:
            meta = op[3] if len(op) > 3 and op[3].get('kind') == 'gemm' else None
            if meta is None:
                rc = op[0](*op[1], st)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = op[0](*op[1], st)
                e1.record()
                recs.setdefault(meta['kernel'], []).append((e0, e1, meta['flops']))
            if rc:
                raise RuntimeError(f"{op[2]} failed rc={rc}")
    lanes_saved, ctx.lanes = ctx.lanes, False      # instrumented passes: every launch on one stream
    sync_saved, step.sync = step.sync, None    # rank-local measurement: no collectives (the other ranks are not here)
    try:
        step._run(*inputs, training=True)      # untimed eager pass (first eager launches pay one-time costs)
        torch.cuda.synchronize()
        # what an EMPTY event bracket costs on a busy stream (the two event packets themselves): measured behind a queue of real work and
        # subtracted from every bracket below, so that a class time is comparable with the kernel durations rocprofv3 reports for the
        # same launches (profiles/*_classes.csv; the dispatch gap in front of a kernel is in neither)
        step._run(*inputs, training=True)
        pairs = []
        for _ in range(24):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(); b_.record()
            pairs.append((a_, b_))
        torch.cuda.synchronize()
        # (an empty bracket is two event packets processed back to back; around a kernel the first packet is processed while the previous
        # kernel drains, so ONE packet - half the empty bracket - is what a bracket adds to the kernel it encloses: with it the class times
        # agree with the rocprofv3 kernel durations of the same launches to ~1 %, without it they read 5-7 % long)
        gemm_profile.bracket_ms = 0.5 * sorted(a_.elapsed_time(b_) for a_, b_ in pairs)[len(pairs) // 2]
        ctx.run = timed_run
        for _ in range(reps):
            step._run(*inputs, training=True)
        torch.cuda.synchronize()
    finally:
        ctx.run = orig_run
        ctx.lanes = lanes_saved
        step.sync = sync_saved
    out = {}
    for k, lst in recs.items():
        n = len(lst) // reps                   # launches of this kernel per step
        per_rep = [sum(max(a.elapsed_time(b) - gemm_profile.bracket_ms, 0.0) for a, b, _ in lst[r * n:(r + 1) * n]) for r in range(reps)]
        ms = sorted(per_rep)[reps // 2]        # median over the passes
        fl = sum(f for _, _, f in lst[:n])
        out[k] = (ms, fl, n)
    return out


def read_sclk_mhz(index=0):
    """Current shader clock from sysfs (pp_dpm_sclk lists the levels, the active one carries a '*'): the HIGHEST clock over the
    visible cards - the node's other GPUs idle at their floor and sysfs card numbers are not HIP device numbers; None when no file
    is readable.  Reported beside the sustained figure: a burst of 50 steps and seconds of back-to-back replays can hold different
    clocks (MI355X_MICROARCH.md, DVFS give-back)."""
    import glob
    import re
    best = None
    for f in glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'):
        try:
            m = re.search(r'(\d+)\s*[Mm][Hh]z\s*\*', open(f).read())
            if m:
                best = max(best or 0, int(m.group(1)))
        except Exception:
            pass
    return best


CPU_BASELINE_THREADS = 16       # eager PyTorch-CPU train_step on the 128-core host: 1.53 / 2.42 / 1.69 / 1.07 / 0.36 img/s at 8 / 16 / 32 / 64 /
                                # 128 threads (tools/cpu_threads_sweep.py): every core oversubscribes the BLAS / ATen pools


def cpu_baseline_worker(budget_s=20.0):
    """Runs in a CHILD process whose BLAS / OpenMP pools were sized by the environment (cpu_baseline below): the numpy oracle's
    pix2pix_train_step (the CPU restatement of the reference path; the TF reference itself is not installable here) and the same graph
    as eager PyTorch-CPU autograd, 256x256, batch 1 (BASELINE config 1).  Prints one JSON line."""
    from oracle import gan_oracle as O
    threads = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    Gp, Dp = O.init_generator(1, seed=11), O.init_discriminator(1, True, seed=12)
    optG, optD = O.AdamTF(), O.AdamTF()
    inp, tar = O.synthetic_pair(1, 256, 1, seed=123)
    masks = O.dropout_masks(1, 256, seed=5)
    O.pix2pix_train_step(Gp, Dp, optG, optD, inp, tar, 100.0, masks, True)       # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        O.pix2pix_train_step(Gp, Dp, optG, optD, inp, tar, 100.0, masks, True)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 50:
            break
    out = {"value": round(n / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
           "sample": f"{n} train_steps of the numpy oracle (fp32), Pix2Pix 256x256 batch 1, {dt:.1f} s, {threads} BLAS threads"}
    # second stand-in SURVEY.md 8(d) names: the same graph as eager PyTorch-CPU autograd (oracle/torch_ref.py), fp32
    try:
        from oracle import torch_ref as TR
        torch.set_num_threads(threads)
        Gt, Dt = TR.params(Gp, torch.float32), TR.params(Dp, torch.float32)
        ti, tt = TR.t(inp, torch.float32), TR.t(tar, torch.float32)
        mt = [TR.t(m, torch.float32) for m in masks]
        state = {}
        TR.pix2pix_train_step_eager(Gt, Dt, state, ti, tt, 100.0, mt)
        n2, t0 = 0, time.perf_counter()
        while True:
            TR.pix2pix_train_step_eager(Gt, Dt, state, ti, tt, 100.0, mt)
            n2 += 1
            dt2 = time.perf_counter() - t0
            if dt2 > budget_s / 2 or n2 >= 50:
                break
        out["torch_cpu_eager"] = {"value": round(n2 / dt2, 4), "unit": "images/sec", "cores": threads,
                                  "sample": f"{n2} eager PyTorch-CPU train_steps (fp32 autograd + TF-form Adam), batch 1, {dt2:.1f} s"}
    except Exception as e:      # the baseline is a report, not a gate
        out["torch_cpu_eager"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


def cpu_baseline(budget_s=20.0, timeout_s=100.0):
    """The CPU legs in a child process (never touches the GPU) with its thread pools sized by the environment - setting thread counts
    inside a process that has already run parallel regions is fragile - and a hard time limit: the baseline is a report, not a gate."""
    threads = max(1, min(CPU_BASELINE_THREADS, os.cpu_count() or 1))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', str(budget_s)], env=env, capture_output=True,
                           text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.strip().split('\n') if ln.startswith('{')]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"cpu baseline worker rc={r.returncode}: {r.stderr.strip()[-300:]}", "kind": "port"}
    except subprocess.TimeoutExpired:
        return {"error": f"cpu baseline worker exceeded {timeout_s:.0f} s", "kind": "port"}


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == '--cpu-baseline-worker':      # child of cpu_baseline(): CPU only
        cpu_baseline_worker(float(sys.argv[2]) if len(sys.argv) > 2 else 20.0)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch')
    ap.add_argument('--img-size', type=int, default=256)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--model', default='pix2pix', choices=['pix2pix', 'cyclegan'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-roofline', action='store_true', help='skip the eager instrumented passes (profiler runs: only the replayed step in the trace)')
    ap.add_argument('--gemm-only', action='store_true', help='also time the GEMM classes with plain epilogues (conv.bwd_fuse_tile = 3, a second step '
                    'object: not the shipped plan) -> "kernels_gemm_only"')
    ap.add_argument('--fp32-allreduce', action='store_true', help='exchange gradients as fp32 instead of bf16')
    ap.add_argument('--repeats', type=int, default=5, help='the K-step timed region is run this many times; the median is reported')
    ap.add_argument('--sustain', type=float, default=3.0, help='seconds of back-to-back steps for the "sustained" key (0: skip)')
    ap.add_argument('--single-stream', action='store_true', help='no side lanes: every launch of the captured step on one stream (profiling)')
    ap.add_argument('--exchange', default='allreduce', choices=['allreduce', 'rs_ag'],
                    help='gradient exchange: all-reduce per bucket, or fp32 reduce-scatter + all-gather in the wire format')
    ap.add_argument('--step-attr', action='append', default=[], metavar='NAME=PYLITERAL',
                    help='schedule attribute of the step object (gan_amd/steps.py class attributes, e.g. early_adam=False); for A/B runs')
    ap.add_argument('--opt', action='append', default=[], metavar='KEY=VALUE',
                    help='planner option of the kernel library (gan_set_option, include/gan_amd.h); for A/B runs')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: start one rank per GPU ourselves (before this process initialises a GPU) and relay the result
        ndev = torch.cuda.device_count()              # does not create a GPU context
        if ndev < args.gpus and os.environ.get('GAN_AMD_ALLOW_SHARED_GPU') != '1':
            print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible", file=sys.stderr)
            sys.exit(3)
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    local = local % max(1, torch.cuda.device_count())        # (rehearsals with several ranks on one GPU)
    if args.gpus != world:                                   # before any process group exists: nothing to tear down
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined", file=sys.stderr)
        sys.exit(4)
    # (TORCH_NCCL_HIGH_PRIORITY=1 - high-priority communicator streams - was measured on the one-rank rehearsal: 3.65 -> 4.56
    # ms/step; the presence of a high-priority queue slows every normal-priority one down.  Left at the default.)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        backend = os.environ.get('GAN_AMD_DIST_BACKEND', 'nccl')     # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
        else:
            dist.init_process_group(backend)
    # GAN_AMD_DDP_REHEARSE=1 on ONE GPU: a one-rank RCCL group with the collectives forced on - the complete data-parallel
    # schedule (bucket graphs, wire-format kernels, asynchronous all-reduces, Adam per bucket) and its cost at N = 1
    rehearse = world == 1 and os.environ.get('GAN_AMD_DDP_REHEARSE') == '1'
    if rehearse:
        import torch.distributed as dist
        if 'MASTER_PORT' not in os.environ:                  # a free port: two rehearsals on one host must not share a TCP store
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(f'cuda:{local}'))
    from gan_amd.ddp import GradSync
    from gan_amd.nets import Ctx, workspace_mb_for
    from gan_amd.steps import CycleGANStep, Pix2PixStep
    from gan_amd import _lib as _L
    for kv in args.opt:
        k, v = kv.split('=', 1)
        _L.set_option(k, int(v))
    dev = f'cuda:{local}'
    torch.cuda.set_device(local)
    B, S = args.batch, args.img_size
    ctx = Ctx(dev, args.dtype, workspace_mb=workspace_mb_for(B, S), lanes=not args.single_stream)
    if args.model == 'pix2pix':
        step = Pix2PixStep(ctx, B, S, 1, lam=100.0, seed=123)
    else:
        step = CycleGANStep(ctx, B, S, 1, lam=10.0, seed=123)
    for kv in args.step_attr:
        import ast
        k, v = kv.split('=', 1)
        if not hasattr(step, k):
            raise SystemExit(f"bench.py: the step has no attribute {k!r}")
        setattr(step, k, ast.literal_eval(v))
    if world > 1 or rehearse:
        step.sync = GradSync([n.params.grad for n in step.nets()], compress_bf16=(args.dtype != "f32" and not args.fp32_allreduce), lib=ctx.lib,
                             rehearse=rehearse, exchange=args.exchange)
    # synthetic inputs on the normalize() lattice u/127.5-1 (base_gan.py:56-61), different per rank
    g = torch.Generator(device='cpu').manual_seed(123 + rank)
    mk = lambda: (torch.randint(0, 256, (B, S, S, 1), generator=g).float() / 127.5 - 1.0).to(dev)
    inputs = (mk(), mk())

    run = (lambda: step._run(*inputs, training=True)) if args.no_graph else None
    if run is None:
        replay = step.capture(training=True)
        # the synthetic batch lives in the captured step's own input buffers (what an input pipeline would fill): resident in HBM when
        # the timed region starts, no device-to-device staging copy per step
        for dst, src in zip(replay.inputs, inputs):
            dst.copy_(src)
        inputs = tuple(replay.inputs)
        run = lambda: replay(*inputs)
    for _ in range(args.warmup):
        run()
    times = []
    for _ in range(max(1, args.repeats)):          # each repeat: EXACTLY --steps steps between barrier + synchronize
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    dt = sorted(times)[len(times) // 2]             # median over the repeats (max over ranks inside each)
    # sustained figure: >= --sustain seconds of back-to-back steps (the timed region above is a few 0.2 s bursts)
    sustained = None
    if args.sustain > 0:
        n_sus = max(args.steps, int(np.ceil(args.sustain / (dt / args.steps))))
        clk0 = read_sclk_mhz(local)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            run()
        clk_mid = read_sclk_mhz(local)                # (read while the queue is still full)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ds = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([ds], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ds = float(t.item())
        sustained = {"seconds": round(ds, 3), "steps": n_sus, "ms_per_step": round(ds / n_sus * 1e3, 4),
                     "value": round(world * B * n_sus / ds, 2), "sclk_mhz_before": clk0, "sclk_mhz_under_load": clk_mid,
                     "sclk_mhz_after": read_sclk_mhz(local)}
    losses = step.losses.cpu().numpy()
    ctx.assert_no_stack_timeout()          # (a persistent layer-stack kernel that timed out at a grid barrier raises a flag instead of hanging)
    if not np.all(np.isfinite(losses)):
        raise RuntimeError(f"non-finite losses {losses}")

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        unit = "images/sec" if args.model == 'pix2pix' else "pairs/sec"
        out = {"metric": f"Pix2Pix train_step images/sec at {S}x{S}" if args.model == 'pix2pix' else f"CycleGAN train_step pairs/sec at {S}x{S}",
               "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic", "repeats": len(times), **({"ddp_rehearsal_one_rank": True} if rehearse else {}),
               "ms_per_step_all_repeats": [round(t_ / args.steps * 1e3, 4) for t_ in times], "sustained": sustained,
               "config": {"workload": f"{'Pix2Pix' if args.model == 'pix2pix' else 'CycleGAN'} {S}x{S} {args.dtype} "
                                      f"batch={B}/GPU train_step (G fwd, D fwd real+fake, losses, dgrad+wgrad, Adam"
                                      f"{', RCCL grad all-reduce' if world > 1 else ''})",
                          "global_batch": world * B, "img_size": S, "channels": 1, "parallelism": f"dp{world}",
                          "hipgraph": not args.no_graph}}
        if args.model == 'pix2pix' and S in GF_PER_IMG:
            out["step_tflops"] = round(value * GF_PER_IMG[S] / 1e3, 1)
            out["step_mfma_frac"] = round(value * GF_PER_IMG[S] / 1e3 / (MFMA_PEAK_TFLOPS * world), 4)
        # roofline of the dominant kernel class, measured live (eager, HIP events on the launch stream) ON THE SHIPPED PLAN: the
        # same step object, networks and planner options the timed region replayed (conv.bwd_fuse_tile = 1: some dgrad launches of
        # the class also start the layer-below backward in their epilogue - that time is inside the figure).
        if args.no_roofline and world == 1 and not rehearse:
            print(json.dumps(out), flush=True)
            return
        prof = gemm_profile(step, inputs)
        tot_ms = sum(v[0] for v in prof.values())
        kname, (kms, kfl, kn) = max(prof.items(), key=lambda kv: kv[1][0])
        ach = kfl / (kms * 1e-3) / 1e12
        # HBM bytes per launch come from a separate rocprofv3 --pmc pass of this same command (counters cannot be
        # collected inside this process); the committed summary is quoted and named, never re-measured here
        traffic, traffic_src = None, None
        for name in ('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
                if B == 16 and S == 256 and args.model == 'pix2pix' and kname in pmc:
                    traffic, traffic_src = pmc[kname]["hbm_bytes_per_launch"], f"profiles/{name} (separate rocprofv3 --pmc pass, not this run)"
                    break
            except Exception:
                pass
        out["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "flops_per_launch": round(kfl / kn),
                           "launches_per_step": kn, "avg_launch_us": round(kms / kn * 1e3, 2),
                           "gemm_share_of_eager_gemm_time": round(kms / tot_ms, 3), "plan": "as shipped (the replayed step's own launches)",
                           "event_bracket_us_subtracted": round(gemm_profile.bracket_ms * 1e3, 2),
                           "timing": "HIP events around each launch of the class on its stream, eager single-stream passes of the SHIPPED "
                                     "plan after the timed region; profiles/r05_bench_p16_kernel_stats_single_stream.csv carries the same "
                                     "classes (columns class, flops_per_launch) from rocprofv3 --kernel-trace of the replayed graph"}
        out["kernels"] = {k: {"ms_per_step": round(v[0], 4), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1), "launches": v[2]}
                          for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        # secondary: the same GEMMs with plain epilogues (conv.bwd_fuse_tile = 3), on a second step object - NOT the shipped plan
        if args.gemm_only and args.model == 'pix2pix' and not step.sync:
            shipped_opt = _L.set_option('conv.bwd_fuse_tile', 3)
            try:
                pstep = Pix2PixStep(ctx, B, S, 1, lam=100.0, seed=123, nets=(step.G, step.D))
                prof2 = gemm_profile(pstep, inputs)
            finally:
                _L.set_option('conv.bwd_fuse_tile', shipped_opt)
            del pstep
            out["kernels_gemm_only"] = {k: {"ms_per_step": round(v[0], 4), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1), "launches": v[2]}
                                        for k, v in sorted(prof2.items(), key=lambda kv: -kv[1][0])}
        out["notes"] = ("inputs are pre-staged in the captured step's own input buffers (resident in HBM, no per-step staging copy); "
                        "roofline / kernels are the shipped plan")
        out["losses"] = [round(float(x), 5) for x in losses[:4]]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1 or rehearse:
        dist.barrier()                     # rank 0 has been profiling alone: every rank leaves the group together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
