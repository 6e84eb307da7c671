#!/usr/bin/env python
"""Benchmark of the DAAM heat-map extraction path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One bench "step" = the extraction work of ONE generation (one prompt/seed) of the workload:
reset of the running sums, ``denoise_steps`` x (all hooked cross-attention layers) taps from
synthetic fp16 Q / K already resident in HBM, the final flush, and one
``compute_global_heat_map`` (bicubic + clamp + mean over all keys) -> one ``[77, 64, 64]``
global heat map.  Default workload = BASELINE.json configs[2]: SDXL-base-1.0 topology at
1024x1024 (60 hooked layers, 1100 (layer, head) keys), 50 denoising steps, 77 tokens, CFG batch 2.
Every tap goes through the per-layer call the attention processor makes (``HeatMapEngine.tap_qk`` = the C++
recorder, then ``daam_tap_qk_enqueue_many`` / ``daam_tap_flush`` of the C ABI when the maps are read), so the
reported rate includes the host cost of that path.  Inputs: one distinct synthetic Q / K set per denoising step
(nothing a later step reads is left in L2 / Infinity Cache by an earlier one).

Multi-GPU: generations are independent (the reference is single-prompt by construction,
daam/trace.py:172-173): rank r runs its own K generations; the only exchange is one RCCL
all_gather of the final maps, inside the timed region.  ``scaling`` = weak.

``python bench.py --gpus N`` outside a launcher re-executes itself under ``torch.distributed.run`` with N ranks (one per
GPU); it never reports ``n_gpus`` other than the N it was asked for.

The default run also times short legs of the other single-GPU BASELINE configurations (``other_configs``: SD-v1.5 50 steps,
SDXL 2048 x 2048 100 steps).  ``--gpus N --dist-backend gloo --shared-device`` runs the whole multi-rank path on a one-GPU box.
Progress goes to stderr; stdout carries the one JSON line.

The one JSON line also carries ``roofline`` (tap kernel: algorithmic bytes / HIP-event time of the launch
as it occurs in the timed region), ``roofline_issue`` (the same launch against its instruction-issue floor: VALU busy
cycles from the committed PMC pass of this build + the MFMA issue cost, at the shader clock sampled IN this run while
the kernel executes), ``roofline_finalize`` / ``roofline_finalize_issue``, ``integrated`` (extraction overhead per
denoising step inside a full-size synthetic SDXL cross-attention stack, tools/synthetic_unet.py) and ``cpu_baseline``
(the torch port of the reference's hook path, oracle/torch_hooks.py, timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_T0 = time.perf_counter()


def note(msg):
    """Progress on stderr (the one JSON line on stdout is the result): which leg runs, since when."""
    print(f'[bench {time.perf_counter() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable

WORKLOADS = {
    # BASELINE.json configs[2] (the headline), configs[4] (100 steps; 25 distinct step sets = 39 GB, cycled: a set is long
    # gone from the 256 MB of Infinity Cache when it recurs), configs[1]
    'sdxl1024': dict(kind='sdxl', latent=128, label='SDXL-base-1.0 topology 1024x1024, 60 layers / 1100 keys'),
    'sdxl2048': dict(kind='sdxl', latent=256, label='SDXL-base-1.0 topology 2048x2048, 60 layers / 1100 keys', denoise_steps=100,
                     pool_cap=25, min_warm=4),
    'sd15': dict(kind='sd15', latent=64, label='SD-v1.5 topology 512x512, 15 layers / 120 keys'),
    # the headline's topology in the other dtype modes the reference runs in (dtype-agnostic sums: daam/heatmap.py:153-156)
    'sdxl1024_bf16': dict(kind='sdxl', latent=128, label='SDXL-base-1.0 topology 1024x1024, 60 layers / 1100 keys, bf16 Q/K and bf16 sums',
                          dtype='bfloat16'),
    'sdxl1024_f32acc': dict(kind='sdxl', latent=128, label='SDXL-base-1.0 topology 1024x1024, 60 layers / 1100 keys, fp16 Q/K, f32 sums',
                            accumulate='float32'),
}


def topology(kind, latent):
    """(layer_idx, heads, side, head_dim) of the hooked layers in UNet execution order
    (down blocks, then up blocks); layer_idx = locator position."""
    if kind == 'sdxl':
        a, b = latent // 2, latent // 4
        loc = ([(i, 20, b, 64) for i in range(0, 30)] + [(i, 10, a, 64) for i in range(30, 40)] +
               [(i, 20, b, 64) for i in range(40, 60)])
        return loc[36:] + loc[:36]
    s = [latent // 4, latent // 2, latent]
    up = [(i, 8, s[i // 3], [160, 80, 40][i // 3]) for i in range(9)]
    down = [(9 + i, 8, [latent, latent // 2, latent // 4][i // 2], [40, 80, 160][i // 2]) for i in range(6)]
    return down + up


def make_inputs(layers, pool, device, seed, arena=False, dtype=torch.float16):
    """``arena``: every Q / K of the pool is a view of ONE allocation (an experiment on what the launch time owes to how the
    recorded tensors are spread over allocator segments -- tools/exp/pool_sweep.py; never the default: a pipeline's Q / K come
    from the caching allocator one tensor at a time)."""
    g = torch.Generator(device=device).manual_seed(seed)
    big, pos = None, 0
    if arena:
        per_set = sum(2 * side * side * heads * d + 2 * 77 * heads * d for _, heads, side, d in layers)
        big = torch.empty(pool * per_set, device=device, dtype=dtype)

    def new(shape):
        nonlocal pos
        if big is None:
            return torch.randn(*shape, generator=g, device=device, dtype=dtype)
        n = shape[0] * shape[1] * shape[2]
        t = big[pos:pos + n].view(*shape)
        pos += n
        return t.normal_(generator=g)
    sets = []
    for _ in range(pool):
        cur = []
        for (_, heads, side, d) in layers:
            q = new((2, side * side, heads * d))
            k = new((2, 77, heads * d))
            k[:, 0, :] *= 3.0                  # SOS-dominant keys, like real prompts
            cur.append((q, k))
        sets.append(cur)
    return sets


def call_lists(layers, sets, latent_side):
    """Per step set: the argument tuples of the per-layer processor calls, in UNet execution order."""
    return [[(layer, q, k, heads, d ** -0.5, (latent_side // side) if side <= latent_side else 0)
             for (layer, heads, side, d), (q, k) in zip(layers, cur)] for cur in sets]


def one_generation(eng, calls, denoise_steps):
    eng.clear()
    tap = eng.tap_qk
    n = len(calls)
    for t in range(denoise_steps):
        for a in calls[t % n]:
            tap(*a)
    # compute_global_heat_map() as daam_amd.trace calls it: the engine announces the output, launches the recorded taps (whose
    # table-upload kernel clears it) and runs the finalize -- no separate flush in front
    return eng.global_heat_map()


def tap_bytes(layers, steps_per_launch, acc_bytes, fresh):
    """Bytes one tap launch must move: conditional-half Q and K of every recorded step, plus
    one write (and, unless the sums are known to be zero, one read) of the running sums."""
    qk = sum(heads * side * side * d * 2 + heads * 77 * d * 2 for _, heads, side, d in layers)
    acc = sum(heads * 77 * side * side * acc_bytes for _, heads, side, d in layers)
    return steps_per_launch * qk + acc * (1 if fresh else 2), qk, acc


class ClockMonitor:
    """Shader clock while the kernels under test run (daam_clock_monitor_*: one wave samples the cycle counter against
    the 100 MHz reference counter every 100 us on its own stream)."""

    def __init__(self, eng, window_ms=60.0, period_us=100):
        from daam_amd import _native as nat
        self.nat, self.eng = nat, eng
        self.n = max(8, min(4096, int(window_ms * 1e3 / period_us)))
        nat.check(eng.lib.daam_clock_monitor_start(eng.ctx, self.n, period_us))

    def read(self):
        import ctypes
        buf = (ctypes.c_float * self.n)()
        n = ctypes.c_int()
        self.nat.check(self.eng.lib.daam_clock_monitor_read(self.eng.ctx, buf, self.n, ctypes.byref(n)))
        v = sorted(buf[i] for i in range(n.value))
        if not v:
            return None
        busy = v[: max(1, len(v) // 2)]            # the lower half: intervals in which the kernels were running (idle gaps clock up)
        return dict(mhz_median_under_load=round(busy[len(busy) // 2], 1), mhz_min=round(v[0], 1), mhz_max=round(v[-1], 1),
                    samples=len(v), method='s_memtime / s_memrealtime, 100 us intervals, concurrent one-wave monitor kernel')


def measure_tap_kernel(eng, calls, defer, reps, fresh):
    """HIP-event time of the tap kernel: libdaam_hip brackets its kernel launch(es) with events on
    the launch stream (daam_profile_enable), the table upload in front of them is outside.
    ``fresh``: the launch is the first after clear() (a generation that fits one launch): the running
    sums are written without being read, exactly as in the timed region."""
    import ctypes
    from daam_amd import _native as nat
    stream = torch.cuda.current_stream()
    eng.clear()
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 1))
    times = []
    for r in range(reps + 2):
        if fresh:
            eng.clear()
        for s in range(defer):
            for a in calls[(r * defer + s) % len(calls)]:
                eng.tap_qk(*a)
        eng.flush()
        ms = ctypes.c_float()
        nat.check(eng.lib.daam_profile_last_ms(eng.ctx, 0, ctypes.byref(ms)))
        if r >= 2:                               # first launch after clear() skips the read; warm-up
            times.append(ms.value)
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 0))
    return sum(times) / len(times)


def measure_finalize(eng, reps):
    """HIP-event time of one compute_global_heat_map on the device: libdaam_hip brackets the table
    upload + output zeroing + its finalize kernels with events on the launch stream."""
    import ctypes
    from daam_amd import _native as nat
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 1))
    times = []
    for r in range(reps + 2):
        eng.global_heat_map()
        ms = ctypes.c_float()
        nat.check(eng.lib.daam_profile_last_ms(eng.ctx, 1, ctypes.byref(ms)))
        if r >= 2:
            times.append(ms.value)
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 0))
    return sum(times) / len(times)


def cpu_model() -> str:
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def _port_sample(th, layers, denoise_steps, threads, sample_steps, cache, budget_s=8.0):
    """``sample_steps`` denoising steps of _unravel_attn + per-head update over the hooked layers and one
    compute_global_heat_map over the keys they produced, on ``threads`` host threads; extrapolated to ``denoise_steps`` steps of
    all layers.  Bounded: a thread count that oversubscribes the host (the op sequence is thousands of small tensor ops) stops
    its tap loop after ``budget_s`` seconds and is scaled by the elements it did process."""
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        raw = th.RawMaps()
        t_tap, done, all_elems = 0.0, 0, sample_steps * sum(h * s * s for _, h, s, _ in layers)
        for _ in range(sample_steps):
            for (layer, heads, side, d) in layers:
                if t_tap > budget_s:
                    break
                t0 = time.perf_counter()
                th.tap(raw, layer, cache[(heads, side)], 4096)
                t_tap += time.perf_counter() - t0
                done += heads * side * side
        t0 = time.perf_counter()
        th.global_heat_map(raw, 4096)
        t_fin = time.perf_counter() - t0
        n_keys, all_keys = len(raw), sum(h for _, h, _, _ in layers)
        key_elems = sum(v.numel() for _, v in raw)
    finally:
        torch.set_num_threads(prev)
    per_step = t_tap * (all_elems / done) / sample_steps
    t_fin_all = t_fin * (sum(h * s * s * 77 for _, h, s, _ in layers) / key_elems)
    return dict(value=1.0 / (per_step * denoise_steps + t_fin_all), unit='maps/s', cores=threads,
                ms_per_denoise_step=per_step * 1e3, finalize_s=t_fin_all, keys=all_keys, keys_sampled=n_keys,
                fraction_of_sample_run=round(done / all_elems, 3), cpu_seconds_sampled=round(t_tap + t_fin, 2))


def cpu_baseline(kind, latent, denoise_steps, sample_steps=2, eager_device=None):
    """THE baseline leg (the only place bench.py touches oracle/, and only to time it): the reference's hook
    path (torch port, oracle/torch_hooks.py; its op sequence timed against the unmodified reference on the build box:
    profiles/r03_port_vs_reference_cpu.json) on the host cores, fp32 (the reference's CPU-runnable
    configuration): `sample_steps` denoising steps of unravel + per-head update over every hooked layer, and
    one compute_global_heat_map -- at 1 thread, at the physical core count and at torch's default thread count; the headline
    ``value`` / ``cores`` is the BEST of the three (the op sequence is thousands of small tensor ops: it is fastest on one
    thread, and oversubscribed at 128).  With `eager_device` the same port is also timed in PyTorch-ROCm eager on the
    MI355X (the denominator of the >= 20x target), returned under 'eager_mi355x'."""
    from oracle import torch_hooks as th
    eager = _port_eager_on_device(th, kind, latent, denoise_steps, eager_device) if eager_device is not None else None
    layers = th.execution_order(th.topology(kind, latent))
    cache = {}
    for (_, heads, side, _) in layers:
        if (heads, side) not in cache:
            cache[(heads, side)] = torch.rand(2 * heads, side * side, 77)
    counts = sorted({1, min(_physical_cores(), torch.get_num_threads()), torch.get_num_threads()})
    runs = []
    for n in counts:
        note(f'cpu baseline at {n} threads')
        runs.append(_port_sample(th, layers, denoise_steps, n, sample_steps, cache))
    best = max(runs, key=lambda r: r['value'])
    out = dict(value=best['value'], unit='maps/s', cores=best['cores'], kind='port', cpu=cpu_model(), host_threads=os.cpu_count(),
               ms_per_denoise_step=best['ms_per_denoise_step'], finalize_s=best['finalize_s'],
               by_threads={str(r['cores']): {k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items() if k != 'unit'} for r in runs},
               sample=f'{sample_steps} denoising steps x {len(layers)} layers of _unravel_attn+update (fp32, torch '
                      f'{torch.__version__}) + 1 compute_global_heat_map over {best["keys"]} keys, extrapolated to {denoise_steps} steps; '
                      f'run at {counts} host threads (each bounded to ~8 s of tap work), best reported ({best["cores"]})',
               port_vs_reference='profiles/r03_port_vs_reference_cpu.json: the port within +-15 % of the unmodified reference on the build box')
    if eager is not None:
        out['eager_mi355x'] = eager
    return out


def _port_eager_on_device(th, kind, latent, denoise_steps, device, sample_steps=2):
    """Same port, PyTorch-ROCm eager on the MI355X, fp16 (what the reference does on a GPU)."""
    layers = th.execution_order(th.topology(kind, latent))
    cache = {}
    for (_, heads, side, _) in layers:
        if (heads, side) not in cache:
            cache[(heads, side)] = torch.rand(2 * heads, side * side, 77, device=device).half()
    raw = th.RawMaps()
    for (layer, heads, side, d) in layers:                      # warm-up
        th.tap(raw, layer, cache[(heads, side)], 4096)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(sample_steps):
        for (layer, heads, side, d) in layers:
            th.tap(raw, layer, cache[(heads, side)], 4096)
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / sample_steps
    t0 = time.perf_counter()
    th.global_heat_map(raw, 4096)
    torch.cuda.synchronize()
    t_fin = time.perf_counter() - t0
    return dict(ms_per_denoise_step=per_step * 1e3, finalize_s=t_fin,
                maps_per_s=1.0 / (per_step * denoise_steps + t_fin))


def integrated_overhead(device, steps=50, reps=9):
    """Extraction overhead per denoising step INSIDE a model-shaped stack (metric (i), integrated harness):
    step time of a full-size synthetic SDXL-1024 cross-attention stack (70 attn2 modules, 60 hooked, fp16, CFG batch 2,
    tools/synthetic_unet.py) under ``daam_amd.trace`` -- incl. one compute_global_heat_map per generation -- minus its
    step time with the stock fused-SDPA processor."""
    import daam_amd
    from tools.synthetic_unet import SyntheticPipeline
    pipe = SyntheticPipeline('sdxl', 128, device=str(device))
    prompt = 'a photo of a monkey riding a bicycle'

    def once(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def traced():
        with daam_amd.trace(pipe) as tc:
            pipe(prompt, num_inference_steps=steps)
            return tc.compute_global_heat_map().heat_maps

    def plain():
        pipe(prompt, num_inference_steps=steps)
    for _ in range(2):                                   # warm-up: code objects, allocator pools, parked trace context, clocks
        plain()
        traced()
    # interleaved plain / traced generations; the overhead is the MEDIAN OF THE PAIRED DIFFERENCES: the difference of two
    # ~8 ms step times is what is measured (the stack is bound by PyTorch's host-side launch rate), so clock ramps,
    # allocator drift and host jitter must hit both members of a pair alike
    tp, tt = [], []
    for _ in range(reps):
        tp.append(once(plain))
        tt.append(once(traced))
    diffs = sorted(t - p for p, t in zip(tp, tt))
    overhead = diffs[len(diffs) // 2]
    tp.sort()
    tt.sort()
    t_plain, t_trace = tp[len(tp) // 2], tt[len(tt) // 2]
    del pipe
    torch.cuda.empty_cache()
    return dict(harness='synthetic SDXL-1024 cross-attention stack, 70 attn2 (60 hooked), fp16, CFG 2, '
                        f'{steps} steps + compute_global_heat_map per generation; {reps} interleaved plain / traced pairs, '
                        'median of the paired differences',
                plain_sdpa_ms_per_step=round(t_plain / steps * 1e3, 3), traced_ms_per_step=round(t_trace / steps * 1e3, 3),
                overhead_ms_per_step=round(overhead / steps * 1e3, 3),
                overhead_quartiles_ms_per_step=[round(diffs[len(diffs) // 4] / steps * 1e3, 3),
                                                round(diffs[(3 * len(diffs)) // 4] / steps * 1e3, 3)])


def _respawn_under_launcher(n, shared_device):
    """``python bench.py --gpus N`` (N > 1) without a launcher: run N ranks of this script under torch.distributed.run."""
    import socket
    import subprocess
    if torch.cuda.device_count() < n and not shared_device:
        raise SystemExit(f'--gpus {n} but only {torch.cuda.device_count()} GPU(s) are visible: refusing to report another n_gpus '
                         '(--shared-device runs N ranks on one device for functional tests of the multi-rank path)')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    import importlib.util
    if importlib.util.find_spec('torch.distributed.run') is not None and os.environ.get('BENCH_NO_TORCHRUN') != '1':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    # no launcher module (or BENCH_NO_TORCHRUN=1): the same N ranks started by hand -- the env:// rendezvous needs nothing else
    procs = []
    for r in range(n):
        renv = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=renv))
    rcs = [p.wait() for p in procs]
    raise SystemExit(next((rc for rc in rcs if rc), 0))


def pin_rank_to_cores(local, world):
    """One slice of the host cores per rank (multi-rank runs): the rank's Python thread, the recorder and the release thread of its
    engine stay off the other ranks' cores.  Returns the slice as text, or None when the platform / cgroup does not allow it."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per < 2:
            return None
        mine = cores[local * per:(local + 1) * per]
        os.sched_setaffinity(0, mine)
        return f'{mine[0]}-{mine[-1]} ({len(mine)} of {len(cores)} cores)'
    except (AttributeError, OSError, ValueError):
        return None


def load_profile(name):
    """A committed profiler summary (profiles/<name>) if it was measured on THIS build of the kernels, else None.  "This build":
    the same kernel sources (``csrc_sha``), or -- when sources were added / gained compiled-out experiments since -- every kernel of
    the measured build byte-identical in this one (``kernel_shas`` in the file against ``daam_amd.build.kernel_shas()``)."""
    from daam_amd.build import csrc_sha, kernel_shas
    path = os.path.join(ROOT, 'profiles', name)
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    if rec.get('csrc_sha') == csrc_sha():
        return rec, f'profiles/{name} (rocprofv3 --pmc passes of this build, csrc {rec["csrc_sha"]}; not re-measured in this run)'
    want = rec.get('kernel_shas')
    if want:
        have = kernel_shas()
        changed = [k for k, v in want.items() if have.get(k) != v]
        if not changed:
            return rec, (f'profiles/{name} (rocprofv3 --pmc passes of csrc {rec["csrc_sha"]}; this build is csrc {csrc_sha()} with all '
                         f'{len(want)} kernels of that build byte-identical; not re-measured in this run)')
        return None, f'profiles/{name} was measured on kernel sources {rec.get("csrc_sha")}; {len(changed)} of its kernels differ in this build ({csrc_sha()})'
    return None, f'profiles/{name} was measured on kernel sources {rec.get("csrc_sha")}, this build is {csrc_sha()}'


COUNTER_FILES = ('r05_counters.json', 'r04_counters.json', 'r03_counters.json', 'r02_counters.json')      # newest first; only one matching this build is used


def load_counters():
    note = None
    for name in COUNTER_FILES:
        prof, n = load_profile(name)
        if prof is not None:
            return prof, n
        note = note or n
    return None, note


_TAP_KERNELS = ('tap_d64_kernel', 'tap_slab_kernel', 'tap_chunk_kernel', 'tap_wide_kernel', 'tap_mfma_kernel', 'tap_generic_kernel')
_FIN_KERNELS = ('finalize_up32_pipe_kernel', 'finalize_up32_same_kernel', 'finalize_up32_mfma_kernel', 'finalize_down2_kernel',
                'finalize_up_kernel', 'finalize_same_kernel', 'finalize_kernel')


def pmc_child(args):
    """``--pmc-child``: what measure_traffic() runs under rocprofv3 -- a few generations of the workload, nothing else."""
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    from daam_amd.engine import HeatMapEngine
    wl = WORKLOADS[args.workload]
    denoise = args.denoise_steps or wl.get('denoise_steps', 50)
    layers = topology(wl['kind'], wl['latent'])
    pool = args.pool if args.pool > 0 else min(denoise, wl.get('pool_cap', denoise))
    sets = make_inputs(layers, pool, device, seed=1234, dtype=getattr(torch, wl.get('dtype', 'float16')))
    eng = HeatMapEngine(len(layers), tokens=77, out_side=64, accumulate=wl.get('accumulate', args.accumulate), defer_steps=args.defer,
                        defer_bytes=args.defer_bytes if args.defer_bytes > 0 else default_defer_bytes(device))
    calls = call_lists(layers, sets, 64)
    for _ in range(4):
        one_generation(eng, calls, denoise)
    torch.cuda.synchronize()
    eng.close()


_SQ_COUNTERS = ('SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA')


def measure_traffic(args, workload, denoise, launches_per_gen, timeout_s=200, sq=True):
    """HBM bytes of the tap launch and of the finalize kernels of ``workload`` -- and (``sq``) the issue counters behind
    ``roofline_issue`` -- measured IN THIS RUN: children of this script (``--pmc-child``: four generations of the workload) under
    ``rocprofv3 --kernel-trace --pmc FETCH_SIZE``, ``--pmc WRITE_SIZE`` and ``--pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA``
    (separate passes, no other trace domain), converted as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  Per launch = upper median over the kernel's dispatches (the first generation's
    launch compiles / warms).  Returns (dict or None, note)."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile
    rp = shutil.which('rocprofv3')
    if rp is None:
        return None, 'rocprofv3 is not on PATH'
    vals = {}
    sq_note = None
    passes = [('FETCH_SIZE',), ('WRITE_SIZE',)] + ([_SQ_COUNTERS] if sq else [])
    for counters in passes:
        d = tempfile.mkdtemp(prefix='daam_pmc_', dir='/tmp')
        cmd = [rp, '--kernel-trace', '--pmc', *counters, '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__),
               '--pmc-child', '--workload', workload, '--denoise-steps', str(denoise), '--defer', str(args.defer),
               '--defer-bytes', str(args.defer_bytes), '--accumulate', args.accumulate, '--pool', str(args.pool)]
        optional = counters is _SQ_COUNTERS                  # the traffic passes are the point; the issue counters are an extra
        try:
            res = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                 timeout=timeout_s, text=True)
            fs = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True) if res.returncode == 0 else []
            if res.returncode != 0 or not fs:
                why = (f'rocprofv3 --pmc {" ".join(counters)} child failed (rc {res.returncode}): {res.stderr[-200:]}' if res.returncode != 0
                       else f'rocprofv3 --pmc {" ".join(counters)}: no counter_collection.csv')
                if optional:
                    sq_note = why
                    continue
                return None, why
            for row in csv.DictReader(open(fs[0])):
                if row['Counter_Name'] not in counters:
                    continue
                for k in _TAP_KERNELS + _FIN_KERNELS:
                    if k in row['Kernel_Name']:
                        vals.setdefault(k, {}).setdefault(row['Counter_Name'], []).append((int(row.get('Dispatch_Id') or 0), float(row['Counter_Value'])))
                        break
        except subprocess.TimeoutExpired:
            if optional:
                sq_note = f'rocprofv3 --pmc {" ".join(counters)} child timed out after {timeout_s} s'
                continue
            return None, f'rocprofv3 --pmc {" ".join(counters)} child timed out after {timeout_s} s'
        finally:
            shutil.rmtree(d, ignore_errors=True)

    n_gen = 4                                                 # generations a --pmc-child runs (pmc_child)
    # dispatch order; the FIRST generation's dispatches are dropped: its tap launch finds the sums known-zero and skips their read
    # (a "fresh" launch), which would bias the mean launch's bytes low
    for k in vals:
        for cn in vals[k]:
            v = [x for _, x in sorted(vals[k][cn])]
            vals[k][cn] = v[len(v) // n_gen:] if len(v) >= n_gen else v

    def upper_median(v):
        v = sorted(v)
        return statistics.median(v[len(v) // 2:])

    n_gen -= 1

    def kernel_bytes(k):
        # mean over the kernel's dispatches: a generation of several tap launches has launches of different lengths (SDXL-2048: 64 + 36
        # steps), and the roofline's bytes_per_launch is their mean too
        cs = vals.get(k, {})
        if 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs:
            return None
        return dict(read=int(2 * statistics.fmean(cs['FETCH_SIZE']) * 1024), write=int(statistics.fmean(cs['WRITE_SIZE']) * 1024),
                    dispatches=len(cs['FETCH_SIZE']), dispatches_per_generation=len(cs['FETCH_SIZE']) / n_gen)
    per = {k: kernel_bytes(k) for k in vals}
    per = {k: v for k, v in per.items() if v}
    tap = [v for k, v in per.items() if k in _TAP_KERNELS]
    fin = [v for k, v in per.items() if k in _FIN_KERNELS]
    if not tap:
        return None, 'no tap kernel in the counter output'
    out = dict(tap_bytes_per_launch=sum(v['read'] + v['write'] for v in tap), finalize_bytes_per_call=sum(v['read'] + v['write'] for v in fin),
               per_kernel=per, launches_per_generation=launches_per_gen,
               method='children of this run under rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); '
                      'bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 rule of MI355X_MICROARCH.md); mean over a kernel\'s dispatches '
                      'of generations 2-4 (a generation of several tap launches: the mean launch, like bytes_per_launch)')

    def per_simd(names, counter, scale=1.0):
        ks = [k for k in names if counter in vals.get(k, {})]
        return round(sum(upper_median(vals[k][counter]) for k in ks) * scale / 1024, 1) if ks else None
    if sq and any('SQ_ACTIVE_INST_VALU' in vals.get(k, {}) for k in _TAP_KERNELS):
        # a flush may run several tap kernels side by side: their work adds up on the same 1024 SIMDs; SQ_ACTIVE_INST_VALU counts quad-cycles
        out['sq'] = dict(tap_valu_busy_cycles_per_simd=per_simd(_TAP_KERNELS, 'SQ_ACTIVE_INST_VALU', 4.0),
                         tap_valu_insts_per_simd=per_simd(_TAP_KERNELS, 'SQ_INSTS_VALU'), tap_mfma_per_simd=per_simd(_TAP_KERNELS, 'SQ_INSTS_MFMA'),
                         finalize_valu_busy_cycles_per_simd=per_simd(_FIN_KERNELS, 'SQ_ACTIVE_INST_VALU', 4.0),
                         finalize_mfma_per_simd=per_simd(_FIN_KERNELS, 'SQ_INSTS_MFMA'),
                         method='one more child under rocprofv3 --kernel-trace --pmc ' + ' '.join(_SQ_COUNTERS) + '; per SIMD = counter / 1024; '
                                'VALU-busy cycles = SQ_ACTIVE_INST_VALU (quad-cycles) x 4; upper median over a kernel\'s dispatches')
    elif sq:
        out['sq_note'] = sq_note or 'no issue counters in the counter output'
    return out, None


def apply_counters(r, t, why):
    """Put what measure_traffic() measured in this run into a workload's rooflines (traffic, issue floors); ``t`` None: say why not."""
    ro, ri, rf, rfi = r['roofline'], r['roofline_issue'], r['roofline_finalize'], r['roofline_finalize_issue']
    if not t:
        ro['traffic_in_run_note'] = why
        return
    ro.update(traffic=t['tap_bytes_per_launch'], traffic_measured_in_run=True, traffic_source=t['method'], traffic_per_kernel=t['per_kernel'],
              traffic_over_algorithmic=round(t['tap_bytes_per_launch'] / ro['bytes_per_launch'], 4))
    if t['finalize_bytes_per_call']:
        rf.update(traffic=t['finalize_bytes_per_call'], traffic_measured_in_run=True,
                  traffic_over_algorithmic=round(t['finalize_bytes_per_call'] / rf['bytes_per_launch'], 4))
    sq = t.get('sq')
    if not sq:
        if ri is not None and 'sq_note' in t:
            ri['counters_in_run_note'] = t['sq_note']
        return
    if ri is not None and ri.get('clock') and sq.get('tap_valu_busy_cycles_per_simd'):
        # two numbers: a FLOOR (the VALU-busy cycles the hardware counted per SIMD: nothing can run faster than its own VALU stream)
        # and an ESTIMATE that also charges ~10 cycles of closed VALU port per MFMA (tools/gen_ubench_issue.py; an upper estimate of
        # that block -- launches have been measured up to 4 % under it, so it is not reported as a bound)
        mhz = ri['clock']['mhz_median_under_load'] * 1e3
        tap_ms = ri['ms_per_launch']
        floor_ms = sq['tap_valu_busy_cycles_per_simd'] / mhz
        est_ms = (sq['tap_valu_busy_cycles_per_simd'] + 10.0 * (sq.get('tap_mfma_per_simd') or 0)) / mhz
        ri.update(valu_busy_cycles_per_simd=sq['tap_valu_busy_cycles_per_simd'], valu_insts_per_simd=sq.get('tap_valu_insts_per_simd'),
                  mfma_insts_per_simd=sq.get('tap_mfma_per_simd'), mfma_issue_block_cycles_estimate=10,
                  floor_ms=round(floor_ms, 4), frac=round(floor_ms / tap_ms, 4), estimate_ms=round(est_ms, 4),
                  measured_over_estimate=round(tap_ms / est_ms, 4), measured_cycles_per_simd=int(tap_ms * mhz), source=sq['method'],
                  counters_measured_in_run=True)
        ri.pop('note', None)
    if rfi.get('clock') and sq.get('finalize_valu_busy_cycles_per_simd'):
        n_mfma = sq.get('finalize_mfma_per_simd') or 0
        mhz = rfi['clock']['mhz_median_under_load'] * 1e3
        pipe_ms = 32.6 * n_mfma / mhz                         # v_mfma_f32_32x32x16_f16: 32.6 cycles of matrix pipe each (tools/ubench_mfma.hip)
        valu_ms = sq['finalize_valu_busy_cycles_per_simd'] / mhz
        fl = max(pipe_ms, valu_ms)
        rfi.update(valu_busy_cycles_per_simd=sq['finalize_valu_busy_cycles_per_simd'], mfma_insts_per_simd=n_mfma, mfma_pipe_cycles_each=32.6,
                   model='floor = max(MFMA count x 32.6 cycles of matrix pipe, VALU-busy cycles) per SIMD at the clock sampled in this run',
                   matrix_pipe_floor_ms=round(pipe_ms, 4), valu_floor_ms=round(valu_ms, 4), floor_ms=round(fl, 4),
                   frac=round(fl / rfi['ms_per_launch'], 4), source=sq['method'], counters_measured_in_run=True)
        # ONE fraction for the call: against the larger of its two floors (HBM bytes at the peak; matrix pipe / VALU issue)
        hbm_ms = rf['bytes_per_launch'] / (HBM_PEAK_GBS * 1e9) * 1e3
        rf.update(floor_ms=round(max(hbm_ms, fl), 4), floor_kind='hbm' if hbm_ms >= fl else 'matrix pipe / issue',
                  frac_of_max_floor=round(max(hbm_ms, fl) / rf['ms_per_launch'], 4))


def default_defer_bytes(device):
    """Byte budget of the recorded Q / K for the bench engines: what ``daam_amd.trace`` gives a pipeline on an otherwise
    empty MI355X (40 % of the device memory, daam_amd/trace.py::_default_defer_bytes).  The bench's synthetic Q / K are
    resident BEFORE the engine records them (recording pins nothing extra), so the rule is applied to the device's total."""
    return int(torch.cuda.get_device_properties(device).total_memory * 0.4)


class Comm:
    """The bench's three collectives (barrier, all_gather of the final maps, MAX of the elapsed times) over ``nccl`` (= RCCL,
    device tensors, the production path) or ``gloo`` (functional runs: several ranks on ONE device, where RCCL refuses a
    duplicate GPU; device tensors are staged through the host)."""

    def __init__(self, backend, device, expect_world=None):
        import torch.distributed as dist
        self.dist, self.backend, self.device = dist, backend, device
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if expect_world is not None and self.world != expect_world:
            raise SystemExit(f'process group of {self.world} ranks, --gpus {expect_world}: refusing to report a line for another world size')
        if backend == 'nccl':
            try:
                torch.cuda.nccl.version()                       # the collective library must be there BEFORE the timed region needs it
            except Exception as e:                              # noqa: BLE001
                raise SystemExit(f'--dist-backend nccl but RCCL is not usable in this process: {e}')

    def library(self):
        """What carried the collectives, as the process group reports it (a SCALE record then shows RCCL saw N ranks)."""
        if self.backend == 'nccl':
            try:
                v = torch.cuda.nccl.version()
                return 'RCCL ' + '.'.join(str(x) for x in (v if isinstance(v, tuple) else (v,))) + f', world_size {self.dist.get_world_size()}'
            except Exception as e:                           # noqa: BLE001 -- reporting only
                return f'nccl (version unavailable: {e}), world_size {self.dist.get_world_size()}'
        return f'gloo (torch {torch.__version__}), world_size {self.dist.get_world_size()}'

    def barrier(self):
        self.dist.barrier()

    def all_gather(self, mine):
        out = torch.empty(self.world * mine.shape[0], *mine.shape[1:], device=mine.device, dtype=mine.dtype)
        if self.backend == 'nccl':
            self.dist.all_gather_into_tensor(out, mine)
        else:
            host = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, mine.cpu())
            out.copy_(host)
        return out

    def max(self, value):
        t = torch.tensor([value], dtype=torch.float64, device=self.device if self.backend == 'nccl' else 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def run_workload(name, denoise_steps, gens, warmup, device, args, comm=None, rank=0, world=1, detail=True):
    """Warm up, time ``gens`` generations of workload ``name`` on this rank (+ the gather of the final maps when there are
    several ranks) between barriers, then -- rank 0, ``detail`` -- the per-kernel measurements.  Returns the pieces of the
    JSON line (rank 0) or None."""
    from daam_amd.engine import HeatMapEngine
    wl = WORKLOADS[name]
    note(f'{name}: inputs')
    layers = topology(wl['kind'], wl['latent'])
    latent_side = 64
    pool = args.pool if args.pool > 0 else min(denoise_steps, wl.get('pool_cap', denoise_steps))
    accumulate = wl.get('accumulate', args.accumulate)
    sets = make_inputs(layers, pool, device, seed=1234 + rank, arena=getattr(args, 'arena', False), dtype=getattr(torch, wl.get('dtype', 'float16')))
    defer_bytes = args.defer_bytes if args.defer_bytes > 0 else default_defer_bytes(device)
    eng = HeatMapEngine(len(layers), tokens=77, out_side=64, accumulate=accumulate, defer_steps=args.defer,
                        defer_bytes=defer_bytes)
    calls = call_lists(layers, sets, latent_side)
    # untimed: the W warm-up generations, plus whatever it takes to reach steady state -- the first call
    # creates the context and loads the code objects (36 ms), the GPU needs ~10 generations from idle to its
    # sustained clock, and the one-off costs of the result stack / the RCCL communicator are paid here too
    note(f'{name}: warm-up')
    warm = [one_generation(eng, calls, denoise_steps) for _ in range(max(warmup, 1))]
    # ... and the caching allocator's pool: the timed region keeps its `gens` result maps alive at once (1.26 MB each), so the warm-up
    # holds as many at once -- otherwise every generation of the region beyond the warm-up's count pays a fresh hipMalloc (~50 us)
    # that a process which has produced `gens` maps before never sees again
    n_more = max(int(os.environ.get('BENCH_MIN_WARM', wl.get('min_warm', 20))), min(gens, 256)) - len(warm)
    for _ in range(max(0, n_more)):
        m = one_generation(eng, calls, denoise_steps)
        if len(warm) < min(gens, 256):
            warm.append(m)
        del m
    w = torch.stack(warm[:max(2, min(gens, 256))])               # the region's stack of `gens` maps, too
    if comm:
        comm.all_gather(w)                                       # ... and one gather of the timed region's size (communicator, buffers)
    warmup_effective = max(warmup, 1) + max(0, n_more)           # generations really run before the timed region
    del warm, w
    launches0 = eng.last_flush()['launches']
    from daam_amd import _native as nat
    # every tap launch / finalize call of the timed region gets its own HIP-event pair on its stream (a ring inside libdaam_hip:
    # no host synchronisation inside the region); read back afterwards -> roofline.ms_per_launch IS the timed region's average
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 2))
    torch.cuda.synchronize()
    if comm:
        comm.barrier()
    torch.cuda.synchronize()
    note(f'{name}: timed region, {gens} generations')
    results = []
    t0 = time.perf_counter()
    for _ in range(gens):
        results.append(one_generation(eng, calls, denoise_steps))
    mine = torch.stack(results)
    gathered = comm.all_gather(mine) if comm else None
    torch.cuda.synchronize()
    if comm:
        comm.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    launches_per_gen = (eng.last_flush()['launches'] - launches0) / gens     # what the engine really launched
    flush = eng.last_flush()
    ran_tap, ran_fin = eng.last_kernels(0), eng.last_kernels(1)

    def history(which, want):
        import ctypes
        buf = (ctypes.c_float * 256)()
        n = ctypes.c_int()
        nat.check(eng.lib.daam_profile_history(eng.ctx, which, buf, min(256, want), ctypes.byref(n)))
        return [buf[i] for i in range(n.value)]
    region_tap = history(0, int(round(launches_per_gen * gens)))
    region_fin = history(1, gens)
    sustained = None
    if detail and not comm and args.defer > 0 and not getattr(args, 'no_sustained', False):
        # The driver's timed region is a few dozen milliseconds; the board needs ~0.8 s of back-to-back launches to settle at its power
        # cap (profiles/r04_power_sclk.txt).  Beside it: >= 2 s of the same generations, the launches timed by the same event ring
        # (its last <= 256 launches = the settled state).
        # generations for >= 2 s, sized by the GPU time of a generation (the launches of the region), topped up until the clock says so
        gpu_gen_s = ((sum(region_tap) / len(region_tap)) * launches_per_gen + (sum(region_fin) / len(region_fin) if region_fin else 0.0)) * 1e-3 \
            if region_tap else elapsed / gens
        note(f'{name}: sustained state, >= 2 s of back-to-back generations')
        torch.cuda.synchronize()
        ts = time.perf_counter()
        n_s, el_s = 0, 0.0
        while el_s < 2.0 and n_s < 100000:
            batch = max(8, int((2.15 - el_s) / max(gpu_gen_s, 1e-5)) + 1)
            for _ in range(batch):
                one_generation(eng, calls, denoise_steps)
            n_s += batch
            torch.cuda.synchronize()
            el_s = time.perf_counter() - ts
        s_tap, s_fin = history(0, 256), history(1, 256)
        sustained = dict(generations=n_s, seconds=round(el_s, 3), maps_per_s=round(n_s / el_s, 2),
                         tap_ms_per_launch=round(sum(s_tap) / len(s_tap), 4) if s_tap else None,
                         finalize_ms=round(sum(s_fin) / len(s_fin), 4) if s_fin else None,
                         note='back-to-back generations for >= 2 s right after the timed region; launch times = mean over the last '
                              f'{len(s_tap)} tap launches / {len(s_fin)} finalize calls (HIP events, read afterwards)')
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 0))
    if comm:
        elapsed = comm.max(elapsed)
        # untimed tail: every rank finds its own maps in its slice of the gathered tensor (rank-major: the all_gather layout)
        if not torch.equal(gathered[rank * gens:(rank + 1) * gens], mine):
            raise SystemExit(f'rank {rank}: gathered maps differ from the maps this rank computed')
        if not bool(torch.isfinite(gathered).all()) or float(gathered.abs().sum()) == 0.0:
            raise SystemExit(f'rank {rank}: gathered maps are empty or not finite')
    del results, mine, gathered
    if rank != 0:
        eng.close()
        return None

    note(f'{name}: {world * gens / elapsed:.1f} maps/s; kernel measurements')
    acc_bytes = 2 if accumulate == 'exact' else 4
    out = dict(label=wl['label'], elapsed=elapsed, gens=gens, denoise_steps=denoise_steps, accumulate=accumulate,
               value=world * gens / elapsed, ms_per_step=elapsed / gens * 1e3, keys=sum(h for _, h, _, _ in layers), sustained=sustained,
               warmup_effective=warmup_effective)
    # steps one tap launch covers: the step window, or fewer when the recorded Q / K reach the engine's
    # byte budget (a launch is then forced at the next step boundary)
    step_bytes = sum(q.numel() * q.element_size() + k.numel() * k.element_size() for q, k in sets[0])
    spl = max(1, min(args.defer, denoise_steps, 64, -(-eng.defer_bytes // step_bytes)))
    out['steps_per_launch'] = spl
    if args.defer > 0:
        expect = -(-denoise_steps // spl)
        if round(launches_per_gen) != expect:
            raise SystemExit(f'{name}: {launches_per_gen} tap launches per generation, expected {expect}')
        launches_per_gen = expect
        fresh = launches_per_gen == 1
        tap_ms = measure_tap_kernel(eng, calls, spl, reps=10 if detail else 4, fresh=fresh)
        bytes_launch, qk_bytes, acc_total = tap_bytes(layers, spl, acc_bytes, fresh=fresh)
        if launches_per_gen > 1:
            # a generation of several launches: the first writes the sums (fresh), the others read-modify-write them, the
            # last may be shorter -- algorithmic bytes and time of the WHOLE generation's launches, reported per launch
            last = denoise_steps - spl * (launches_per_gen - 1)
            gen_bytes = (denoise_steps * qk_bytes + acc_total * (2 * launches_per_gen - 1))
            tap_ms_last = measure_tap_kernel(eng, calls, last, reps=4, fresh=False) if last != spl else tap_ms
            tap_ms_first = measure_tap_kernel(eng, calls, spl, reps=4, fresh=True)
            gen_ms = tap_ms_first + (launches_per_gen - 2) * tap_ms + tap_ms_last
            out['tap_ms_per_generation'] = gen_ms
            bytes_launch, tap_ms_avg = gen_bytes / launches_per_gen, gen_ms / launches_per_gen
        else:
            tap_ms_avg = tap_ms
            out['tap_ms_per_generation'] = tap_ms
    else:
        # immediate mode: 1 launch per layer call; time a whole denoising step of launches
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        one_generation(eng, calls, 2)
        e0.record(stream)
        reps = 10
        for r in range(reps):
            for a in calls[r % len(calls)]:
                eng.tap_qk(*a)
        e1.record(stream)
        e1.synchronize()
        tap_ms = tap_ms_avg = e0.elapsed_time(e1) / reps / len(layers)
        bytes_launch, qk_bytes, acc_total = tap_bytes(layers, 1, acc_bytes, fresh=False)
        bytes_launch /= len(layers)
        launches_per_gen = denoise_steps * len(layers)
        out['tap_ms_per_generation'] = tap_ms * launches_per_gen
    # the figure of the roofline: the launches of the TIMED REGION (back-to-back generations: the chip at its sustained, power-limited
    # state); the isolated measurement above (launches separated by the host's recording work) stays as ms_per_launch_isolated
    tap_ms_isolated = tap_ms_avg
    if args.defer > 0 and region_tap:
        tap_ms_avg = sum(region_tap) / len(region_tap)
        out['tap_ms_per_generation'] = tap_ms_avg * launches_per_gen
    achieved = bytes_launch / (tap_ms_avg * 1e-3) / 1e9
    survey_bytes = spl * (qk_bytes + 2 * acc_total) if args.defer > 0 else bytes_launch   # RMW per step
    key = f'{name}:defer{spl}:{accumulate}'
    prof, prof_note = load_counters()
    rec = (prof or {}).get('workloads', {}).get(key)
    if rec and rec.get('tap_kernels_per_launch', flush['kernels']) != flush['kernels']:
        # the committed counters are of another launch structure (SD-v1.5: three kernels side by side; now one chunked kernel)
        prof_note = (f'profiles: the committed PMC pass of {key} is of a {rec["tap_kernels_per_launch"]}-kernel launch, this run launches '
                     f'{flush["kernels"]}: {rec.get("tap_note", "not comparable")}')
        rec = {k: v for k, v in rec.items() if not k.startswith('tap_')}
    traffic = rec.get('tap_bytes_per_launch') if rec else None
    # what the engine's last tap launch really launched (daam_last_kernels, ABI v6) -- not what the environment asked for
    tap_kernel = ran_tap + {'tap_d64_kernel': ' (16x16x32 MFMA tiles, head_dim 64)',
                            'tap_slab_kernel': ' (head_dim 40 / 80 / 160: 640-byte slabs of adjacent heads, every layer in ONE launch)',
                            'tap_chunk_kernel': ' (any head_dim in 64-element chunks, every layer in ONE launch)'}.get(ran_tap, '')
    out['roofline'] = dict(bound='hbm', kernel=tap_kernel,
                           achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(achieved / HBM_PEAK_GBS, 4),
                           traffic=traffic, traffic_measured_in_run=False,
                           traffic_source=prof_note if traffic is not None else (prof_note or 'no PMC pass committed for this workload'),
                           bytes_per_launch=int(bytes_launch), ms_per_launch=round(tap_ms_avg, 4),
                           ms_per_launch_source=f'mean of the {len(region_tap)} tap launches of the timed region (HIP events on the launch stream, read after the region)' if (args.defer > 0 and region_tap) else 'separate loop',
                           ms_per_launch_isolated=round(tap_ms_isolated, 4),
                           steps_per_launch=spl, launches_per_generation=launches_per_gen,
                           kernels_per_launch=flush['kernels'], kernels_on_side_streams=flush['side_streams'],
                           achieved_at_survey_8d_bytes=round(survey_bytes / (tap_ms * 1e-3) / 1e9, 1))
    # ---- issue-rate roofline of the same launch: the deferred tap keeps the sums in registers, so its HBM work is the
    # Q / K stream only and the kernel is bound by instruction issue (softmax VALU + MFMA).  Floor = (VALU busy cycles +
    # MFMA instructions x the ~10 cycles each keeps the VALU port closed, tools/ubench_issue) per SIMD / shader clock.
    roofline_issue = None
    if args.defer > 0:
        mon = ClockMonitor(eng, window_ms=40.0)                  # second pass with the monitor wave running beside the kernel
        measure_tap_kernel(eng, calls, spl, reps=8 if detail else 3, fresh=launches_per_gen == 1)
        clock = mon.read()
        roofline_issue = dict(bound='issue', kernel=tap_kernel, clock=clock, ms_per_launch=round(tap_ms, 4),
                              ms_per_launch_source='isolated launches (separate loop) with the clock monitor beside them: time and clock of the SAME pass')
        if rec and clock and rec.get('tap_valu_busy_cycles_per_simd'):
            # two numbers: a FLOOR (the VALU-busy cycles the hardware counted per SIMD: nothing can run faster than its own VALU
            # stream) and an ESTIMATE that also charges ~10 cycles of closed VALU port per MFMA (tools/gen_ubench_issue.py; an upper
            # estimate of that block -- launches have been measured up to 4 % under it, so it is not reported as a bound)
            mhz = clock['mhz_median_under_load'] * 1e3
            floor_ms = rec['tap_valu_busy_cycles_per_simd'] / mhz
            est_ms = (rec['tap_valu_busy_cycles_per_simd'] + 10.0 * rec.get('tap_mfma_per_simd', 0)) / mhz
            roofline_issue.update(valu_busy_cycles_per_simd=rec['tap_valu_busy_cycles_per_simd'],
                                  valu_insts_per_simd=rec.get('tap_valu_insts_per_simd'),
                                  mfma_insts_per_simd=rec.get('tap_mfma_per_simd'), mfma_issue_block_cycles_estimate=10,
                                  floor_ms=round(floor_ms, 4), frac=round(floor_ms / tap_ms, 4),
                                  estimate_ms=round(est_ms, 4), measured_over_estimate=round(tap_ms / est_ms, 4),
                                  measured_cycles_per_simd=int(tap_ms * mhz), source=prof_note, counters_measured_in_run=False)
        else:
            roofline_issue['note'] = prof_note or 'no PMC pass committed for this workload'
    out['roofline_issue'] = roofline_issue
    # host cost of the per-layer call path for one generation (launches are asynchronous)
    torch.cuda.synchronize()
    th0 = time.perf_counter()
    for t in range(denoise_steps):
        for a in calls[t % len(calls)]:
            eng.tap_qk(*a)
    eng.flush()
    out['host_ms'] = (time.perf_counter() - th0) * 1e3
    torch.cuda.synchronize()
    fin_ms = measure_finalize(eng, reps=40 if detail else 10)
    mon = ClockMonitor(eng, window_ms=10.0, period_us=50)       # the clock in a second pass: the monitor wave is kept out of the timing
    measure_finalize(eng, reps=60 if detail else 10)
    fin_clock = mon.read()
    fin_ms_isolated = fin_ms
    if region_fin:
        fin_ms = sum(region_fin) / len(region_fin)             # the finalize calls of the timed region
    fin_bytes = acc_total + 77 * 64 * 64 * 4
    fin_gbs = fin_bytes / (fin_ms * 1e-3) / 1e9
    fin_kernel = ran_fin + ('; key tables cached on the device, output cleared by the upload kernel of the tap launch in front (daam_finalize_prepare): '
                            'the timed call is the class kernel(s) only')
    fin_issue = dict(bound='matrix-pipe / issue', kernel=fin_kernel, clock=fin_clock, ms_per_launch=round(fin_ms, 4))
    if rec and fin_clock and rec.get('finalize_valu_busy_cycles_per_simd'):
        n_mfma = rec.get('finalize_mfma_per_simd', 0)
        mhz = fin_clock['mhz_median_under_load'] * 1e3
        pipe_ms = 32.6 * n_mfma / mhz                                  # v_mfma_f32_32x32x16_f16: 32.6 cycles of matrix pipe each (tools/ubench_mfma.hip)
        valu_ms = rec['finalize_valu_busy_cycles_per_simd'] / mhz
        fl = max(pipe_ms, valu_ms)
        fin_issue.update(valu_busy_cycles_per_simd=rec['finalize_valu_busy_cycles_per_simd'],
                         mfma_insts_per_simd=n_mfma, mfma_pipe_cycles_each=32.6,
                         model='floor = max(MFMA count x 32.6 cycles of matrix pipe, VALU-busy cycles) per SIMD at the clock sampled in this run; '
                               'the x2 bicubic of an fp16 plane is 10 MFMA 32x32x16 per half plane (DESIGN.md 3.3): the op is bound by the matrix '
                               'pipe, not by HBM; frac = that floor / the whole call (the loop is ~60 % of it)',
                         matrix_pipe_floor_ms=round(pipe_ms, 4), valu_floor_ms=round(valu_ms, 4),
                         floor_ms=round(fl, 4), frac=round(fl / fin_ms, 4), source=prof_note, counters_measured_in_run=False)
    out['fin_ms'] = fin_ms
    out['roofline_finalize'] = dict(bound='hbm', bound_in_fact=('matrix pipe for the x2 loop (~60 % of the kernel\'s span, running at ~0.8 of that pipe: DESIGN.md 3.3) + ring prefill, same-size keys, '
                                                   'reduction / atomics around it; roofline_finalize_issue has the floor') if name != 'sdxl2048' else 'hbm', kernel=fin_kernel, achieved=round(fin_gbs, 1), peak=HBM_PEAK_GBS,
                                    unit='GB/s', frac=round(fin_gbs / HBM_PEAK_GBS, 4), bytes_per_launch=int(fin_bytes),
                                    ms_per_launch=round(fin_ms, 4), ms_per_launch_isolated=round(fin_ms_isolated, 4),
                                    traffic=rec.get('finalize_bytes_per_launch') if rec else None,
                                    traffic_measured_in_run=False,
                                    excludes='the output zeroing and the key-table upload ride in the table-upload kernel of the tap launch, in front of the tap '
                                             'window\'s start event: outside both event windows (~10 us per generation, profiles/r04_generation_gaps.json); '
                                             'the wall-clock value contains them')
    out['roofline_finalize_issue'] = fin_issue
    eng.close()
    del sets, calls
    torch.cuda.empty_cache()
    return out


LINE_LIMIT = 4096             # bytes of the ONE stdout line (round 5's 25 KB line was dropped by the driver: BENCH_r05.parsed = null)
FULL_RECORD = os.path.join('gpurun_out', 'bench_full.json')


def _short(s, n=160):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + '...'


def _pick(d, keys):
    return {k: _short(d[k]) for k in keys if d and k in d and not isinstance(d[k], (dict, list))}


def headline_line(full: dict) -> str:
    """The ONE stdout line: the contract's fields + ``roofline`` + ``cpu_baseline`` + one scalar per extra leg, strict JSON,
    < LINE_LIMIT bytes.  Everything else ``main`` measured (per-kernel counter tables, issue rooflines, the other configurations'
    full records, prose on sources) is in the full record (``gpurun_out/bench_full.json``, copied to ``profiles/`` per round)."""
    line = _pick(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'warmup_effective', 'ms_per_step', 'higher_is_better',
                        'scaling', 'vs_baseline', 'dtype', 'data'))
    line['config'] = _pick(full.get('config'), ('workload', 'accumulate', 'defer_steps', 'parallelism', 'generations_per_rank', 'baseline_config',
                                                'collective', 'collective_world_size', 'collective_library'))
    line['roofline'] = _pick(full.get('roofline'), ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'bytes_per_launch', 'ms_per_launch',
                                                    'traffic', 'traffic_over_algorithmic', 'traffic_measured_in_run', 'steps_per_launch',
                                                    'launches_per_generation', 'sustained_frac'))
    line['roofline'].setdefault('traffic', None)
    cpu = full.get('cpu_baseline')
    line['cpu_baseline'] = _pick(cpu, ('value', 'unit', 'cores', 'kind', 'cpu', 'sample')) if cpu else None
    # one scalar per extra measurement, most important first (dropped from the END if the line would not fit)
    extras = []

    def add(key, value):
        if value is not None:
            extras.append((key, value))
    rf = full.get('roofline_finalize') or {}
    add('finalize_ms', rf.get('ms_per_launch'))
    add('finalize_frac', rf.get('frac'))
    add('finalize_traffic_over_algorithmic', rf.get('traffic_over_algorithmic'))
    add('sustained_maps_per_s', full.get('sustained_maps_per_s'))
    add('sustained_tap_ms', full.get('sustained_tap_ms'))
    add('extraction_overhead_ms_per_denoise_step', full.get('extraction_overhead_ms_per_denoise_step'))
    add('speedup_vs_eager_mi355x', full.get('speedup_vs_eager_mi355x'))
    add('reference_eager_mi355x_maps_per_s', (full.get('reference_eager_mi355x') or {}).get('maps_per_s'))
    for name, o in (full.get('other_configs') or {}).items():
        ro, rfo = o.get('roofline') or {}, o.get('roofline_finalize') or {}
        add(f'{name}_maps_per_s', o.get('value'))
        add(f'{name}_tap_ms', ro.get('ms_per_launch'))
        add(f'{name}_tap_frac', ro.get('frac'))
        add(f'{name}_tap_traffic_over_algorithmic', ro.get('traffic_over_algorithmic'))
        add(f'{name}_finalize_ms', rfo.get('ms_per_launch'))
        add(f'{name}_finalize_frac', rfo.get('frac'))
    add('integrated_overhead_ms_per_denoise_step', (full.get('integrated') or {}).get('overhead_ms_per_step'))
    att = full.get('attend') or {}
    add('attend_ms_per_denoise_step', (att.get('attend') or {}).get('ms_per_step'))
    add('torch_sdpa_ms_per_denoise_step', (att.get('torch_sdpa') or {}).get('ms_per_step'))
    add('gpu_bound_maps_per_s', full.get('gpu_bound_maps_per_s'))
    add('host_enqueue_ms_per_generation', full.get('host_enqueue_ms_per_generation'))
    if full.get('full_record'):
        line['full_record'] = full['full_record']
    while True:
        s = json.dumps({**line, **dict(extras)}, allow_nan=False, separators=(', ', ': '))
        if len(s.encode()) < LINE_LIMIT or not extras:
            break
        extras.pop()
    if len(s.encode()) >= LINE_LIMIT:
        raise SystemExit(f'bench.py: the headline line is {len(s.encode())} bytes (limit {LINE_LIMIT})')
    return s


def _finite(o):
    """NaN / inf -> None, so that the records are strict JSON."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float('inf') else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def write_full_record(full: dict):
    """The whole measurement next to the headline line; returns the path written, or None (a read-only tree is not an error)."""
    try:
        rel = os.environ.get('BENCH_FULL_RECORD', FULL_RECORD)      # (tests running several benches side by side give each its own file)
        path = rel if os.path.isabs(rel) else os.path.join(ROOT, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(full, f, allow_nan=False, indent=1)
        return rel
    except OSError as e:
        note(f'full record not written: {e}')
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50, help='timed generations per rank')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='sdxl1024', choices=sorted(WORKLOADS))
    ap.add_argument('--denoise-steps', type=int, default=0, help='denoising steps per generation (0 = the configuration\'s: 50; SDXL-2048: 100)')
    ap.add_argument('--defer', type=int, default=int(os.environ.get('DAAM_DEFER_STEPS', '64')),
                    help='denoising steps tapped per launch (0 = one launch per layer call)')
    ap.add_argument('--defer-bytes', type=int, default=int(os.environ.get('DAAM_DEFER_BYTES', '0')),
                    help='bytes of recorded Q / K per launch (0 = 40 %% of the device memory, the rule daam_amd.trace applies)')
    ap.add_argument('--accumulate', default='exact', choices=['exact', 'float32'])
    ap.add_argument('--pool', type=int, default=0,
                    help='distinct synthetic Q/K step sets resident in HBM (0 = one per denoising step: no step of a '
                         'generation re-reads data an earlier one left in L2 / Infinity Cache; SDXL-2048: 25 sets = 39 GB)')
    ap.add_argument('--arena', action='store_true',
                    help='experiment: carve every synthetic Q / K out of ONE allocation (see make_inputs); never the reported configuration')
    ap.add_argument('--no-baselines', action='store_true', help='skip the CPU / eager-GPU reference timings')
    ap.add_argument('--no-integrated', action='store_true', help='skip the integrated-overhead leg')
    ap.add_argument('--no-pmc', action='store_true', help='skip the in-run HBM-traffic measurement (two short children under rocprofv3 --pmc)')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-other-configs', action='store_true', help='skip the short legs of the other single-GPU BASELINE configurations')
    ap.add_argument('--no-dtype-legs', action='store_true', help='skip the bf16 and f32-sums legs of the headline topology')
    ap.add_argument('--no-sustained', action='store_true', help='skip the >= 2 s sustained-state leg behind the timed region')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend of a multi-rank run: nccl = RCCL over xGMI (production); gloo = functional runs')
    ap.add_argument('--shared-device', action='store_true',
                    help='functional test of the multi-rank path on a one-GPU box: every rank uses cuda:0 (needs --dist-backend gloo: '
                         'RCCL refuses two ranks on one device); the line says so and is not a scaling measurement')
    args = ap.parse_args()
    if args.shared_device and args.dist_backend != 'gloo':
        raise SystemExit('--shared-device needs --dist-backend gloo (RCCL refuses a duplicate GPU)')
    if args.pmc_child:
        return pmc_child(args)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        _respawn_under_launcher(args.gpus, args.shared_device)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = 0 if args.shared_device else int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    affinity = pin_rank_to_cores(local if not args.shared_device else rank, world) if world > 1 else None
    comm = Comm(args.dist_backend, device, expect_world=args.gpus) if world > 1 else None

    denoise = args.denoise_steps or WORKLOADS[args.workload].get('denoise_steps', 50)
    main_rec = run_workload(args.workload, denoise, args.steps, args.warmup, device, args, comm=comm, rank=rank, world=world)

    out = None
    if rank == 0:
        r = main_rec
        if world == 1 and not args.no_pmc:
            note('HBM traffic + issue counters: children under rocprofv3 --pmc')
            apply_counters(r, *measure_traffic(args, args.workload, denoise, r['roofline']['launches_per_generation']))
        gpu_ms_per_gen = r['tap_ms_per_generation'] + r['fin_ms']
        extra = dict(
            extraction_overhead_ms_per_denoise_step=round(r['ms_per_step'] / denoise, 4),
            gpu_ms_per_denoise_step=round(r['tap_ms_per_generation'] / denoise, 4),
            gpu_bound_maps_per_s=round(1e3 / gpu_ms_per_gen, 1),
            host_enqueue_ms_per_generation=round(r['host_ms'], 3),
            raw_maps_per_s=round(world * args.steps * denoise * r['keys'] / r['elapsed'], 1),
            roofline_finalize=r['roofline_finalize'], roofline_issue=r['roofline_issue'],
            roofline_finalize_issue=r['roofline_finalize_issue'],
        )
        if world == 1 and not args.no_other_configs and args.workload == 'sdxl1024':
            # the other single-GPU configurations of BASELINE.json, short legs (their parity: tests/test_gpu_integration.py)
            others = {}
            # (SD-v1.5: 100 generations = 45 ms, the length of the headline's region -- 20 generations are 9 ms, over before the board has
            # left its idle clocks: the launches of so short a region read 10 % longer than the same launches a few milliseconds later)
            legs = [('sd15', 100, 10), ('sdxl2048', 5, 2)]
            if not args.no_dtype_legs:
                legs += [('sdxl1024_bf16', 20, 5), ('sdxl1024_f32acc', 20, 5)]
            for name, g, wu in legs:
                ds = WORKLOADS[name].get('denoise_steps', 50)
                o = run_workload(name, ds, g, wu, device, args, detail=False)
                if not args.no_pmc:
                    note(f'{name}: HBM traffic + issue counters under rocprofv3 --pmc')
                    apply_counters(o, *measure_traffic(args, name, ds, o['roofline']['launches_per_generation'], sq=name in ('sd15', 'sdxl2048')))
                dt = {'bfloat16': 'bf16'}.get(WORKLOADS[name].get('dtype'), 'fp16')
                others[name] = dict(config=f'{o["label"]}, {ds} denoising steps, 77 tokens, CFG batch 2, {dt} Q/K, accumulate={o["accumulate"]}',
                                    value=round(o['value'], 2), unit='maps/s', generations=g, warmup=wu,
                                    ms_per_step=round(o['ms_per_step'], 3),
                                    gpu_bound_maps_per_s=round(1e3 / (o['tap_ms_per_generation'] + o['fin_ms']), 1),
                                    roofline=o['roofline'], roofline_issue=o['roofline_issue'],
                                    roofline_finalize=o['roofline_finalize'], roofline_finalize_issue=o['roofline_finalize_issue'])
            # the dtype legs against the headline: tap launch time relative to the fp16 / fp16-sums launch of THIS run
            for name in ('sdxl1024_bf16', 'sdxl1024_f32acc'):
                if name in others:
                    others[name]['tap_ms_over_fp16_headline'] = round(others[name]['roofline']['ms_per_launch'] / r['roofline']['ms_per_launch'], 4)
            extra['other_configs'] = others
        if not args.no_integrated and world == 1 and args.workload == 'sdxl1024':
            note('integrated overhead')
            extra['integrated'] = integrated_overhead(device)
            note('attend bench')
            # the processor's attention on daam_attend (tools/attend_bench.py): per denoising step of 60 layer calls, next
            # to torch's fused SDPA, and the cost of the in-kernel tap of an immediate (defer_steps=0) trace
            from tools.attend_bench import measure as attend_measure
            extra['attend'] = attend_measure(steps=20, reps=3, dev=device)
        cpu = None
        if not args.no_baselines and world == 1:
            wl = WORKLOADS[args.workload]
            note('cpu baseline + eager reference')
            cpu = cpu_baseline(wl['kind'], wl['latent'], denoise, eager_device=device)
            ref_gpu = cpu.pop('eager_mi355x')
            extra['reference_eager_mi355x'] = {k: round(v, 4) for k, v in ref_gpu.items()}
            extra['speedup_vs_eager_mi355x'] = round(r['value'] / ref_gpu['maps_per_s'], 1)
            cpu = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in cpu.items()}
        out = {
            'metric': 'heat maps/sec, DAAM extraction (tap + compute_global_heat_map), ' + r['label'] +
                      f', {denoise}-step, 77-tok',
            'value': round(r['value'], 2), 'unit': 'maps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(r['ms_per_step'], 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'{r["label"]}, {denoise} denoising steps, 77 tokens, CFG batch 2, fp16 Q/K',
                       'accumulate': args.accumulate, 'defer_steps': r['steps_per_launch'], 'parallelism': f'prompt-shard x{world}',
                       'generations_per_rank': args.steps,
                       'baseline_config': ('BASELINE.json configs[3]: SDXL-base-1.0 1024x1024, 50 steps, batch=32 prompts sharded 8xMI355X (RCCL gather)'
                                           if (world == 8 and world * args.steps == 32 and args.workload == 'sdxl1024' and denoise == 50) else
                                           'BASELINE.json configs[2] per rank' + (f' x {world} ranks (prompt shards; configs[3] is this at --gpus 8 --steps 4)' if world > 1 else '')
                                           if args.workload == 'sdxl1024' else None),
                       'rank_cpu_affinity': affinity,
                       'collective': None if world == 1 else f'all_gather of the final [{args.steps}, 77, 64, 64] fp32 maps per rank over '
                                                              f'{args.dist_backend}' + (' -- ALL RANKS ON ONE DEVICE (functional run, not a '
                                                                                        'scaling measurement)' if args.shared_device else ''),
                       'collective_world_size': comm.world if comm else 1,
                       'collective_library': comm.library() if comm else None},
            'roofline': r['roofline'], 'cpu_baseline': cpu,
        }
        if r.get('sustained'):
            su = r['sustained']
            out['sustained'] = su
            out['sustained_maps_per_s'] = su['maps_per_s']
            out['sustained_tap_ms'] = su['tap_ms_per_launch']
            if su['tap_ms_per_launch']:
                out['roofline']['sustained_ms_per_launch'] = su['tap_ms_per_launch']
                out['roofline']['sustained_frac'] = round(r['roofline']['bytes_per_launch'] / (su['tap_ms_per_launch'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out.update(extra)
        out['warmup_effective'] = r['warmup_effective']
    if comm:
        comm.close()
    if rank == 0:
        out = _finite(out)
        out['full_record'] = write_full_record(out)
        note('done')
        print(headline_line(out), flush=True)


if __name__ == '__main__':
    main()
