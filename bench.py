#!/usr/bin/env python3
"""bench.py -- full-feature extraction throughput of the fused HIP sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic labelled volume that is already resident
in HBM: accumulator init + fused sweep + adjacency collection (+ for N > 1 the per-label RCCL
reduce and the adjacency merge).  N = 1 runs BASELINE.json's metric config (C4: 1024^3 uint32,
~50k seeds, full feature set).  N > 1 runs BASELINE.json's configuration 5 ITSELF -- C5: 2048^3 uint32, 100k seeds --
Z-slab partitioned over the N GPUs with a one-plane halo (a fixed total: strong scaling; its single-GPU figure is
`secondary.c5_single_gpu` of the N = 1 line).  The slabs are cut by COST, not by plane count (the ranks count the label
changes of their planes once and agree on the cuts: distributed.balanced_cuts), the per-label sums are reduce-scattered,
the bounding boxes all-reduced, only the pairs a slab face can split are exchanged.

Before the W warmup steps the same step runs untimed for --settle-ms (default 200 ms; `config.settle_ms`): a chip that has
idled -- the synthetic volume is generated and the CPU baseline computed before anything is timed -- needs tens of
milliseconds of work to reach its sustained clock, and K steps of ~1 ms would otherwise measure the ramp.

Rank 0 prints ONE JSON line (contract in the task description), with extra objects:
  roofline     -- the sweep kernel's algorithmic bytes / its mean HIP-event duration vs 8 TB/s, plus
                  `peak_measured`: what a trivial 16-B/lane read-reduce kernel reaches on the same buffer in this run
  cpu_baseline -- the per-label scipy restatement of the reference (oracle/, 1 core) timed on a
                  bounded crop of the same volume on this node's host (rank 0, N = 1 only); `best_effort`
                  inside it = the one-pass C restatement on 16 host processes (NOT the reference's algorithm)
  secondary    -- (N = 1) the same volume WITHOUT the ellipsoid mask: tissue everywhere, ~50k labels present,
                  2.6x the event density of the headline workload (its fraction is also `roofline.frac_50k_labels`); the
                  headline workload with two steps in flight (what N > 1 runs), so that a scaling curve compares like with
                  like; `sorted_adjacency_ms`: what a caller of ta_adjacency_get pays on top of a step (device sort by
                  (lo, hi) + fetch); `c5_single_gpu`: config C5 (2048^3, 34 GB) on this ONE GPU -- the denominator of
                  BASELINE.json's ">= 6x at 8 GPUs", since N = 8 runs C5; `wall_voxels_c2`: the wall-voxel kernels (SURVEY.md
                  §8f-3) on config C2's volume, count + scan + fetch against the volume once + 20-byte records.
                  (N > 1) `global_adjacency_gather_ms`: one SlabJob.result_arrays(), which assembles the GLOBAL pair list
                  from the ranks' private lists -- not part of a step (a step leaves private + travelling pairs per rank)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def bench_config(n_gpus):
    """N = 1: C4, the configuration BASELINE.json's metric is quoted on.  N >= 2: C5 ITSELF (2048^3, 100k seeds), Z-slab
    partitioned over the N GPUs -- BASELINE.json's configuration 5 ("at 2/4/8 GPUs"): the total is fixed (strong scaling);
    its single-GPU figure rides on the N = 1 line as secondary.c5_single_gpu."""
    from tissue_analysis_amd import synth
    c = synth.CONFIGS["C4" if n_gpus == 1 else "C5"]
    return dict(name="C4" if n_gpus == 1 else "C5", dims=c["dims"], dtype=c["dtype"], n_cells=c["n_cells"], seed=c["seed"])


def cpu_baseline(vol_tensor, dims, dtype, edge=400):
    """Time the oracle (reference algorithm restated, 1 core) on a centred crop of the volume."""
    from oracle.sia_oracle import full_feature_set
    from tissue_analysis_amd import synth
    e = [min(edge, d) for d in dims]
    lo = [(d - x) // 2 for d, x in zip(dims, e)]
    crop = vol_tensor[lo[0]:lo[0] + e[0], lo[1]:lo[1] + e[1], lo[2]:lo[2] + e[2]].contiguous().cpu().numpy()
    crop = crop.view(np.dtype(dtype))
    t0 = time.perf_counter()
    out = full_feature_set(crop, synth.PARITY_VOXELSIZE, background=1)
    dt = time.perf_counter() - t0
    return dict(value=round(crop.size / dt / 1e6, 3), unit="Mvoxels/s", cores=1, kind="port",
                sample="centred %dx%dx%d crop of the bench volume, %d labels, full feature set "
                       "(oracle/sia_oracle.py: per-label scipy.ndimage loops as in the reference), %.1f s"
                       % (e[0], e[1], e[2], len(out["labels"]), dt),
                host_cpus=os.cpu_count())


def pmc_traffic(config_name, kernel_name):
    """HBM bytes per launch of THIS kernel on THIS config from profiles/pmc_traffic.json (separate rocprofv3 --pmc passes,
    scripts/update_pmc_traffic.py) -- or None with the reason: no entry for the kernel that ran, or an entry measured on other
    sources than the ones this run was built from (the hash of kernels_scan.hip + its headers travels with every entry)."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        db = json.load(open(path))
        e = db.get("entries", {}).get(config_name, {}).get(kernel_name)
        if e is None:
            return None, "no PMC entry for %s on %s" % (kernel_name, config_name)
        h = hashlib.sha256()
        for f in ("tissue_analysis_amd/csrc/kernels_scan.hip", "tissue_analysis_amd/csrc/ta_sweep_common.h",
                  "tissue_analysis_amd/csrc/ta_pin_tables.inc"):
            h.update(open(os.path.join(ROOT, f), "rb").read())
        if h.hexdigest()[:16] != e.get("sources_sha16"):
            return None, "the PMC entry for %s (commit %s) was taken on other sweep sources" % (kernel_name, e.get("commit"))
        return int(e["bytes"]), "PMC passes of commit %s (%s)" % (e.get("commit"), e.get("from"))
    except Exception as exc:
        return None, "pmc_traffic.json unreadable: %s" % type(exc).__name__


def _slab_worker(args):
    """Best-effort CPU: one Z-slab of the sample through the one-pass C restatement (own process: it keeps globals)."""
    from oracle import onepass_c
    lo, hi, L = args
    vol = _BE_SAMPLE              # inherited through fork: no pickling of the voxels
    halo = 1 if lo > 0 else 0
    return onepass_c.extract(vol[lo - halo:hi], max_label=L, origin=(lo - halo, 0, 0), own_first_plane=not halo)


_BE_SAMPLE = None


def cpu_best_effort(vol_tensor, dims, dtype, max_label, planes=256, workers=16):
    """The build's own one-pass formulation on `workers` host processes (BASELINE.md §4 "best-effort CPU")."""
    import multiprocessing as mp
    from oracle import onepass, onepass_c
    onepass_c.build()
    planes = min(planes, dims[0])
    lo0 = (dims[0] - planes) // 2
    sample = vol_tensor[lo0:lo0 + planes].contiguous().cpu().numpy().view(np.dtype(dtype))
    global _BE_SAMPLE
    _BE_SAMPLE = sample
    cuts = [planes * i // workers for i in range(workers + 1)]
    jobs = [(a, b, max_label) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        parts = pool.map(_slab_worker, jobs)
    merged = onepass.merge(parts)
    dt = time.perf_counter() - t0
    _BE_SAMPLE = None
    return dict(value=round(sample.size / dt / 1e6, 1), unit="Mvoxels/s", cores=len(jobs), kind="port",
                sample="%d central planes of the bench volume (%d labels), one-pass C restatement (oracle/onepass_c.c) on %d "
                       "processes + merge, %.1f s; NOT the reference's algorithm" % (planes, int((merged["count"] > 0).sum()),
                                                                                      len(jobs), dt))


def _launch_ranks(n):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a child process."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:                       # (rank 0 prints ONE JSON line; anything else on stdout goes to stderr)
        text = out.strip()
        if text.startswith("{") and '"metric"' in text:
            line = text
        elif text:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-ms", type=float, default=200.0, help="untimed steps before the warmup, until the clock has settled")
    ap.add_argument("--config", default=None, help="override: C2/C3/C4/C5 on one GPU")
    ap.add_argument("--features", type=lambda v: int(v, 0), default=None,
                    help="override the feature mask (e.g. 0x0f = everything but adjacency); not the headline metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the tissue-filled and the two-steps-in-flight figures")
    ap.add_argument("--no-ellipsoid", action="store_true", help="tissue everywhere (not the headline workload)")
    ap.add_argument("--tile-planes", type=int, default=0)
    ap.add_argument("--dims", type=int, nargs=3, default=None, help="rehearsal: override the volume shape")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # A plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (nothing here has touched the
        # GPU yet, and a child is safe either way), relay rank 0's JSON line and the launcher's return code.
        sys.exit(_launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from tissue_analysis_amd import _capi, device as dev, distributed as tad, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world != n and not (n == 1 and world == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (n, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback)")
    # TA_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (RCCL cannot
    # put two ranks on one device); the real runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("TA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.config:
        c = synth.CONFIGS[args.config]
        cfg = dict(name=args.config, dims=c["dims"], dtype=c["dtype"], n_cells=c["n_cells"], seed=c["seed"])
        feats = _capi.feature_mask(c["features"])
    else:
        cfg = bench_config(n)
        feats = _capi.F_ALL
    if args.features is not None:
        feats = args.features
        cfg = dict(cfg, name=cfg["name"] + "-mask0x%02x" % feats)
    if args.dims:
        cfg = dict(cfg, dims=tuple(args.dims), name=cfg["name"] + "-custom")
    dims, dtype = cfg["dims"], np.dtype(cfg["dtype"])

    ctx = dev.torch_context(local_rank)
    if args.tile_planes:
        ctx.set_option(_capi.OPT_TILE_PLANES, args.tile_planes)
    # Z-slab of this rank: planes [a_lo, a_hi) plus the plane below as low halo
    a_lo, a_hi = tad.slab_range(dims[0], n, rank)
    halo = 1 if a_lo > 0 else 0
    vol, max_label = dev.synth_slab(ctx, dims, dtype, cfg["n_cells"], cfg["seed"], a_lo - halo, a_hi,
                                    device=local_rank, ellipsoid=not args.no_ellipsoid)
    cuts = None
    if n > 1 and os.environ.get("TA_BENCH_BALANCE", "1") != "0":
        # Work-balanced slabs: the slowest slab sets the step, and the end slabs of a tissue are emptier than the middle ones.
        # Every rank counts the label changes of the planes it generated (one streaming pass, once), the counts are gathered,
        # axis 0 is cut into slabs of equal COST (distributed.plane_costs / balanced_cuts) and the slab is generated again.
        torch.cuda.synchronize()
        ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, a0_origin=a_lo, has_low_halo=bool(halo), keep=vol)
        mine = torch.zeros(dims[0], dtype=torch.int64)
        mine[a_lo:a_hi] = torch.from_numpy(ctx.plane_events().astype(np.int64))
        on = ("cuda:%d" % local_rank) if backend == "nccl" else "cpu"
        mine = mine.to(on)
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        cuts = tad.balanced_cuts(tad.plane_costs(mine.cpu().numpy(), dims[1] * dims[2]), n)
        del vol
        a_lo, a_hi = tad.slab_range(dims[0], n, rank, cuts)
        halo = 1 if a_lo > 0 else 0
        vol, max_label = dev.synth_slab(ctx, dims, dtype, cfg["n_cells"], cfg["seed"], a_lo - halo, a_hi,
                                        device=local_rank, ellipsoid=not args.no_ellipsoid)
    reduce_mode = os.environ.get("TA_BENCH_REDUCE", "scatter") if n > 1 else "all"
    if args.no_ellipsoid:
        cfg = dict(cfg, name=cfg["name"] + "-filled")
    # N > 1: two steps in flight on two streams, so that step i's RCCL reduce / adjacency exchange
    # overlaps step i+1's sweep (TA_BENCH_PIPELINE=1 turns the overlap off)
    depth = int(os.environ.get("TA_BENCH_PIPELINE", "2")) if n > 1 else 1
    if depth > 1:
        job = tad.PipelinedSlabJob(vol, dtype.itemsize, a_origin=a_lo, has_low_halo=bool(halo),
                                   max_label=max_label, features=feats, group=dist.group.WORLD,
                                   device=local_rank, depth=depth, tile_planes=args.tile_planes, reduce=reduce_mode)
        last_ctx = lambda: job.last.ctx
    else:
        job = tad.SlabJob(ctx, vol, dtype.itemsize, a_origin=a_lo, has_low_halo=bool(halo),
                          max_label=max_label, features=feats, group=(dist.group.WORLD if n > 1 else None),
                          device=local_rank, reduce=reduce_mode)
        last_ctx = lambda: ctx

    def barrier():
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if n > 1:                       # one-off size agreement of every job in flight (synchronous), before any timing
        for _ in range(depth):
            job.step()
        job.finish()
    # The chip takes tens of milliseconds of work to reach its sustained clock after idling (the first 10 ms of sweeps of a
    # process measure 3 - 4 % slower than the rest: gpurun_out/c4_tp3.txt): run the same step, untimed, for --settle-ms
    # before the W warmup steps, so that K short steps measure the steady state and not the ramp.
    if args.settle_ms > 0:
        barrier()
        t_est = time.perf_counter()
        for _ in range(5):
            job.step()
        job.finish()
        barrier()
        est = torch.tensor([(time.perf_counter() - t_est) / 5.0], dtype=torch.float64, device="cuda:%d" % local_rank)
        if n > 1:
            dist.all_reduce(est, op=dist.ReduceOp.MAX)          # the same count on every rank: steps are collective
        for _ in range(min(5000, int(args.settle_ms * 1e-3 / max(float(est.item()), 1e-6)) + 1)):
            job.step()
        job.finish()
    for _ in range(args.warmup):
        job.step()
    barrier()
    # every context in use keeps the HIP events around the sweep kernel of each of its timed launches (two event
    # records per step, on the launch stream); nothing else is recorded inside the timed region
    contexts = [j.ctx for j in job.jobs] if hasattr(job, "jobs") else [ctx]
    for cx in contexts:
        cx.set_option(_capi.OPT_TIMING, 1)
        cx.set_option(_capi.OPT_TIMING_RING, max(1, min(4096, args.steps)))
    barrier()
    sweep_ms, adj_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    job.finish()          # collective: read the (deferred) verdict of the exchange inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    # the dominant kernel's duration: HIP events of the launches of the timed region itself
    for cx in contexts:
        sweep_ms += cx.timing_series()
    # what follows the sweep in a step (hot-row fold + adjacency collect), from three more steps with the step's
    # begin / end events switched on
    for cx in contexts:
        cx.set_option(_capi.OPT_TIMING, 2)
    for _ in range(3 * len(contexts)):
        job.step()
        job.finish()
        torch.cuda.synchronize()
        adj_ms.append(last_ctx().timing()["ms_adjacency"])
    bytes_read = last_ctx().timing()["bytes_read"]

    gather_ms = None
    if n > 1 and feats & _capi.F_ADJACENCY:
        # the GLOBAL pair list is not part of a step (ranks keep private + travelling pairs): one collective assembly, timed
        barrier()
        t1 = time.perf_counter()
        arrays = job.result_arrays()
        barrier()
        gather_ms = (time.perf_counter() - t1) * 1e3
        global_pairs = int(arrays["pair_lo"].size)
        del arrays
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda:%d" % local_rank)
    if n > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    nvox = float(dims[0]) * dims[1] * dims[2]
    ms_per_step = dt / args.steps * 1e3
    value = nvox * args.steps / dt / 1e6

    labels_present = int((job.result_counts() > 0).sum())       # (COLLECTIVE when the sums are reduce-scattered: every rank asks)
    if rank == 0:
        sweep = float(np.mean(sweep_ms))
        achieved = bytes_read / (sweep * 1e-3) / 1e9
        owned = job.owned_view()
        probe_ms = last_ctx().read_probe(owned.data_ptr(), owned.numel() * owned.element_size(), repeats=5)
        peak_measured = owned.numel() * owned.element_size() / (probe_ms * 1e-3) / 1e9
        kernel_name = ((("scan_wide_kernel" if last_ctx().get_option(_capi.OPT_SWEEP_SHAPE_USED) else "scan_two_rows_kernel")
                        if dtype.itemsize == 4 else "scan_kernel") if feats & _capi.F_ADJACENCY else "scan_noadj_kernel")
        traffic, traffic_note = pmc_traffic(cfg["name"], kernel_name)
        out = {
            "metric": "Mvoxels/s full-feature extraction, 1024^3 vol/50k labels; % HBM roofline",
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic jittered-grid Voronoi tissue in an ellipsoid (tissue_analysis_amd/synth.py), resident in HBM",
            "config": {"workload": "%s: %dx%dx%d %s, %d seeds (%d labels present), features=0x%x, Z-slab x%d"
                                   % (cfg["name"], dims[0], dims[1], dims[2], dtype.name, cfg["n_cells"],
                                      labels_present, feats, n),
                       "voxels_per_gpu": int(nvox / n), "label_dtype": dtype.name, "steps_in_flight": depth,
                       "scaling_note": "N = 1 runs the configuration the metric is quoted on (C4); N >= 2 run C5 itself, Z-slab "
                                       "partitioned (fixed total: strong scaling) -- its single-GPU figure is secondary.c5_single_gpu "
                                       "of the N = 1 line",
                       "slab_cuts": cuts, "per_label_reduce": reduce_mode,
                       "settle_ms": args.settle_ms,
                       "pct_hbm_roofline": round(100.0 * achieved / HBM_PEAK_GBS, 2)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "peak_measured": round(peak_measured, 1), "frac_of_measured": round(achieved / peak_measured, 4),
                         "traffic_note": traffic_note, "kernel": kernel_name,
                         "kernel_ms": round(sweep, 4), "kernel_launches_timed": len(sweep_ms),
                         "adjacency_collect_ms": round(float(np.mean(adj_ms)), 4),
                         "algorithmic_bytes_per_launch": int(bytes_read),
                         "tile_planes": int(last_ctx().get_option(_capi.OPT_TILE_PLANES))},
        }
        if gather_ms is not None:
            out["secondary"] = {"global_adjacency_gather_ms": {"value": round(gather_ms, 3), "pairs": global_pairs,
                                "what": "one SlabJob.result_arrays(): all-gather of the ranks' private pair lists + host merge; "
                                        "outside the timed steps"}}
        if n == 1 and not args.no_cpu_baseline:     # (before the secondary figures release the volume)
            cpu_baseline_result = cpu_baseline(job.owned_view(), (a_hi - a_lo, dims[1], dims[2]), dtype)
            cpu_baseline_result["best_effort"] = cpu_best_effort(job.owned_view(), (a_hi - a_lo, dims[1], dims[2]), dtype, max_label)
        if n == 1 and not args.no_secondary and not args.config and args.features is None and not args.dims \
                and not args.no_ellipsoid:
            sec = {}

            def settle(step, finish):          # the headline's --settle-ms for the single-rank figures below (the host just
                t_s = time.perf_counter()      # spent seconds on the CPU baseline: the chip is idle again)
                while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
                    for _ in range(8):
                        step()
                    finish()
            # (a) two steps in flight on two streams: what N > 1 runs
            pj = tad.PipelinedSlabJob(vol, dtype.itemsize, a_origin=a_lo, has_low_halo=bool(halo), max_label=max_label,
                                      features=feats, group=None, device=local_rank, depth=2, tile_planes=args.tile_planes)
            settle(pj.step, lambda: (pj.finish(), torch.cuda.synchronize()))
            for _ in range(args.warmup):
                pj.step()
            pj.finish(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pj.step()
            pj.finish(); torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            pj.close()
            sec["two_steps_in_flight"] = {"ms_per_step": round(dt2 / args.steps * 1e3, 4),
                                          "value": round(nvox * args.steps / dt2 / 1e6, 1), "unit": "Mvoxels/s"}
            # (b) tissue everywhere: same shape and seeds, no ellipsoid mask
            del pj
            vol2, _ = dev.synth_slab(ctx, dims, dtype, cfg["n_cells"], cfg["seed"], 0, dims[0], device=local_rank,
                                     ellipsoid=False)
            torch.cuda.synchronize()
            job2 = tad.SlabJob(ctx, vol2, dtype.itemsize, a_origin=0, has_low_halo=False, max_label=max_label,
                               features=feats, group=None, device=local_rank)
            settle(job2.step, torch.cuda.synchronize)
            for _ in range(args.warmup):
                job2.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                job2.step()
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t1
            k_ms = ctx.timing()["ms_sweep"]
            sec["tissue_filled"] = {"workload": "%s without the ellipsoid mask: %d labels present" % (
                                        cfg["name"], int((job2.result_counts() > 0).sum())),
                                    "ms_per_step": round(dt3 / args.steps * 1e3, 4),
                                    "value": round(nvox * args.steps / dt3 / 1e6, 1), "unit": "Mvoxels/s",
                                    "kernel_ms": round(k_ms, 4), "roofline_frac": round(bytes_read / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            out["roofline"]["frac_50k_labels"] = sec["tissue_filled"]["roofline_frac"]
            del vol2, job2
            # (c) what a caller of the sorted pair list pays on top of a step: ta_adjacency_get = device radix sort by
            #     (lo, hi) + gather of the face counts + the records to the host (first call after each extraction)
            job = tad.SlabJob(ctx, vol, dtype.itemsize, a_origin=a_lo, has_low_halo=bool(halo), max_label=max_label,
                              features=feats, group=None, device=local_rank)      # (the context is back on the headline volume)
            ts = []
            for _ in range(5):
                job.step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                lo_, hi_, _f = ctx.adjacency()
                ts.append((time.perf_counter() - t1) * 1e3)
            sec["sorted_adjacency_ms"] = {"value": round(float(np.median(ts)), 4), "pairs": int(lo_.size),
                                          "what": "ta_adjacency_get after a step: sort by (lo, hi) on the device + fetch"}
            # (d) config C5 on this one GPU (34 GB): the single-GPU denominator of the 8-GPU target
            try:
                c5 = synth.CONFIGS["C5"]
                d5, t5 = c5["dims"], np.dtype(c5["dtype"])
                vol5, L5 = dev.synth_slab(ctx, d5, t5, c5["n_cells"], c5["seed"], 0, d5[0], device=local_rank)
                torch.cuda.synchronize()
                job5 = tad.SlabJob(ctx, vol5, t5.itemsize, a_origin=0, has_low_halo=False, max_label=L5, features=feats,
                                   group=None, device=local_rank)
                settle(job5.step, torch.cuda.synchronize)
                for _ in range(2):
                    job5.step()
                torch.cuda.synchronize()
                ctx.set_option(_capi.OPT_TIMING_RING, 5)
                t1 = time.perf_counter()
                for _ in range(5):
                    job5.step()
                torch.cuda.synchronize()
                dt5 = (time.perf_counter() - t1) / 5
                k5 = float(np.mean(ctx.timing_series()))
                b5 = float(d5[0]) * d5[1] * d5[2] * t5.itemsize
                sec["c5_single_gpu"] = {"workload": "C5: 2048^3 uint32, 100000 seeds (%d labels present), features=0x%x, one GPU"
                                                    % (int((job5.result_counts() > 0).sum()), feats),
                                        "ms_per_step": round(dt5 * 1e3, 4), "value": round(b5 / t5.itemsize / dt5 / 1e6, 1),
                                        "unit": "Mvoxels/s", "kernel_ms": round(k5, 4),
                                        "roofline_frac": round(b5 / (k5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                del vol5, job5
            except Exception as e:                      # (a box with less free HBM than 34 GB + tables)
                sec["c5_single_gpu"] = {"skipped": "%s: %s" % (type(e).__name__, str(e)[:120])}
            # (e) the second kernel family of the build: wall voxels (18-neighbourhood, one record per voxel and neighbour label)
            # of config C2's volume -- count + scan + fetch kernels against "the volume once + 20-byte records"
            try:
                c2 = synth.CONFIGS["C2"]
                d2, t2 = c2["dims"], np.dtype(c2["dtype"])
                vol_w, _ = dev.synth_slab(ctx, d2, t2, c2["n_cells"], c2["seed"], 0, d2[0], device=local_rank)
                torch.cuda.synchronize()
                ctx.set_volume_device(vol_w.data_ptr(), t2.itemsize, vol_w.shape, keep=vol_w)
                best = None
                for _ in range(3):
                    lo_w, _hi_w, _co_w, ms_w = ctx.wall_voxels()
                    best = ms_w if best is None else min(best, ms_w)
                alg = float(d2[0]) * d2[1] * d2[2] * t2.itemsize + 20.0 * lo_w.size
                sec["wall_voxels_c2"] = {"workload": "C2: 512^3 uint16, wall voxels of every pair (ta_wall_voxels_count + _get)",
                                         "records": int(lo_w.size), "kernels_ms": round(best, 4),
                                         "roofline_frac": round(alg / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "algorithmic_bytes": int(alg)}
                del vol_w, lo_w, _hi_w, _co_w
            except Exception as e:
                sec["wall_voxels_c2"] = {"skipped": "%s: %s" % (type(e).__name__, str(e)[:120])}
            # (f) sparse label ids: the headline volume's cells renamed on the device to random ids below 2^31, then the census of
            # the ids (np.unique on the device), the rank copy of the volume, and the same sweep over it (rows = ids present)
            try:
                ctx.set_volume_device(vol.data_ptr(), dtype.itemsize, vol.shape, keep=vol)
                rng = np.random.default_rng(5)
                top = (1 << 31) if dtype.itemsize == 4 else 65535
                lut = np.unique(rng.integers(0, top, size=4 * (max_label + 1), dtype=np.uint64))
                lut = rng.permutation(lut)[:max_label + 1].astype(np.uint32)
                ctx.relabel(lut)                        # (the headline volume is not used after this)

                def best_ms(f, reps=3):
                    out_, best_ = None, 1e9
                    for _ in range(reps):
                        torch.cuda.synchronize()
                        t1_ = time.perf_counter()
                        out_ = f()
                        torch.cuda.synchronize()
                        best_ = min(best_, (time.perf_counter() - t1_) * 1e3)
                    return best_, out_
                census_ms, (top_id, ids_) = best_ms(ctx.label_census)
                rank_ms, _ = best_ms(lambda: ctx.compact_labels())
                ctx.bind_accumulators(None, None, 0)
                ctx.set_option(_capi.OPT_TIMING_RING, 5)
                for _ in range(8):                      # (a new volume to the context: the shape of its sweep is measured first)
                    ctx.extract(feats, ids_.size - 1)
                ctx.adjacency_size()
                ctx.set_option(_capi.OPT_TIMING_RING, 5)
                for _ in range(5):
                    ctx.extract(feats, ids_.size - 1)
                ctx.adjacency_size()
                ks = float(np.mean(ctx.timing_series()))
                sec["sparse_ids"] = {"workload": "%s with its %d cells renamed to random ids up to %d (dense rows: %.1f GB)"
                                                 % (cfg["name"], int(ids_.size), int(top_id), (top_id + 1) * 104 / 1e9),
                                     "census_ms": round(census_ms, 4), "rank_copy_ms": round(rank_ms, 4), "kernel_ms": round(ks, 4),
                                     "roofline_frac": round(bytes_read / (ks * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            except Exception as e:
                sec["sparse_ids"] = {"skipped": "%s: %s" % (type(e).__name__, str(e)[:160])}
            out["secondary"] = sec
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_result
        print(json.dumps(out), flush=True)
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
