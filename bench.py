"""bench.py - FusionCG iterations/s and HBM roofline on MI355X (contract: see task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3z16|cells]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one CG iteration (fusion.py:439-456) on synthetic inputs already resident in HBM.
The timed loop is run twice in one invocation: with the library's default factored residual (headline `value`,
`roofline`) and with the residual as an explicit third 5-D vector (`other_residual_mode`); --single times one only.
N = 1: BASELINE.json configs[1] (t256 x l16 x z16 x 128 x 128, r = 4).  N > 1: weak scaling, the
same 16-plane slab per GPU (nZ = 16 N), z-sharded, one RCCL all-reduce of the spc-sized RD partial
plus two scalar all-reduces per iteration.  Rank 0 prints ONE JSON line.  `--gpus N` without a launcher starts its own
N ranks (self_launch); the N > 1 line also carries BASELINE configs[2] strong-scaled (and configs[3] at N = 8).
Environment knobs, for rehearsals on a box with one GPU only: DFX_FORCE_COMM=1 (issue the collectives on a one-rank
group), DFX_BENCH_STRONG=1 (run the strong-scaling leg whatever N), DFX_BENCH_STRONG_WORKLOAD=<name> (its volume),
DFX_BENCH_CONFIG4=1, DFX_BENCH_BACKEND=gloo with DFX_BENCH_SHARE_GPU=1 (several ranks on one device).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
PMC_RECORD = "r03_pmc_traffic.json"   # written by tools/profile_all.sh; carries the sha256 of the kernel source it was taken on

WORKLOADS = {  # per-GPU slab: (nT, nL, nZ per GPU, XY, r)
    "c2": (256, 16, 16, 128, 4),
    "c3": (512, 32, 32, 256, 4),         # config 3 whole on ONE GPU: 128 GiB / vector, x + p = 256 GiB (factored residual only)
    "c4": (256, 16, 32, 512, 4),         # config 4 whole on ONE GPU, same size
    "c3z16": (512, 32, 16, 256, 4),      # 64 GiB / vector: half of config 3
    "c3g8": (512, 32, 4, 256, 4),        # config 3 as sharded over 8 GPUs (per-GPU slab)
    "c4g8": (256, 16, 4, 512, 4),        # config 4 as sharded over 8 GPUs (per-GPU slab)
    "cells": (64, 16, 20, 128, 4),       # README-like
    "c2r2": (256, 16, 16, 128, 2),       # C2 volume with spatial_increase 2 / 8 (LaneMap<2>, LaneMap<8>)
    "c2r8": (256, 16, 16, 128, 8),
    "c2r3": (256, 16, 16, 120, 3),       # other block sizes: strip tile (LaneMap<0>: lcm(r, 4) <= 64 and Y % 4 == 0), else the generic kernels
    "c2r5": (256, 16, 16, 120, 5),
    "c2r6": (256, 16, 16, 120, 6),
    "c2r7": (256, 16, 16, 112, 7),
    "c2r10": (256, 16, 16, 120, 10),
    "c2r12": (256, 16, 16, 120, 12),
    "c2r16": (256, 16, 16, 128, 16),     # row-part tile (LaneMap<16>: r = 16, 32, 64)
    "c2r32": (256, 16, 16, 128, 32),
    "c2y120": (256, 16, 16, 120, 4),     # r = 4 on planes whose rows are not 128-byte aligned (480 B pitch)
    "c2y96": (256, 16, 16, 96, 4),       # 128-byte aligned rows that are not multiples of a wave tile's 64 columns (row maps)
    "c2y160": (128, 16, 16, 160, 4),
    "c2y100": (256, 16, 24, 100, 4),     # ... not 32-byte aligned either (Y % 8 != 0)
    "c2y200": (128, 16, 13, 200, 4),
    "c2r2y120": (256, 16, 16, 120, 2),
    "c2r8y120": (256, 16, 16, 120, 8),
    "c2r3y96": (256, 16, 28, 96, 3),     # strip tile on 128-byte aligned rows (384 / 768 B pitch)
    "c2r3y192": (256, 16, 7, 192, 3),
    "c2gen": (256, 16, 16, 126, 3),      # Y % 4 != 0: the generic kernels (flat float4 sweep + the v pass)
    "c2odd": (256, 16, 15, 125, 5),      # slab length not a multiple of 4: the generic kernels with scalar lanes
    "p512s": (16, 4, 16, 512, 4),        # wide planes (more workgroups than CUs) with few / many slabs per workgroup
    "p512m": (64, 16, 16, 512, 4),
    "p1k": (4, 2, 8, 1024, 4),
    "f128m1k": (64, 16, 16, 128, 4),     # neighbours of C2 (planner experiments: planes per z-chunk, tools/exp_zs.sh)
    "f128z8": (256, 16, 8, 128, 4),
    "f128z32": (256, 16, 32, 128, 4),
    "f64z16": (1024, 16, 16, 64, 4),
    "f256z16": (64, 16, 16, 256, 4),
    "tiny": (8, 4, 4, 64, 4),
}


def synthetic(n_t, n_l, n_z, xy, r, seed=0):
    """spc then cmos from one CPU generator, 0.1 + 0.9 U[0,1)."""
    g = torch.Generator().manual_seed(seed)
    spc = 0.1 + 0.9 * torch.rand(n_t, n_l, xy // r, xy // r, generator=g)
    cmos = 0.1 + 0.9 * torch.rand(n_z, xy, xy, generator=g)
    return spc, cmos


def _host_facts():
    model, avail = "unknown", 0
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    return model, avail


def cpu_baseline(n_t, n_l, n_z, xy, r, iters=2):
    """The oracle (eager-faithful torch-CPU port of fusion.py:423-456, per-iteration loss included) timed on this
    host on the FULL workload when host memory allows (the eager loop holds ~12 N-sized fp32 tensors: ~52 GB for C2),
    otherwise on the same geometry with fewer time bins, scaled by N and labelled so.  One call of `iters` iterations:
    b and r0 = b - A(x0) (fusion.py:428-437) are paid once and reported separately, as a reference call pays them.
    Reported, never the target."""
    from oracle import fusion_oracle as O
    model, avail = _host_facts()
    full_n = n_t * n_l * n_z * xy * xy
    need = 12 * 4 * full_n
    t_use = n_t
    if avail and need * 1.3 > avail:                       # bounded sample: same planes, fewer time bins
        t_use = max(2, int(n_t * avail / (need * 1.6)))
    spc, cmos = synthetic(n_t, n_l, n_z, xy, r)
    spc = spc[:t_use].contiguous()
    w = {"spatial": 0.5, "spectro_temporal": 0.5}
    t0 = time.perf_counter()
    pb = O.prepare(spc, cmos, weights=w, init_type="baseline", tol=None)
    t_prep = time.perf_counter() - t0
    tm = {}
    x_cpu = O.cg(pb, iters, timings=tm)
    n_sample = t_use * n_l * n_z * xy * xy
    per_iter = sum(tm["iter_s"]) / len(tm["iter_s"])
    its = 1.0 / per_iter
    full = t_use == n_t
    what = (f"oracle/fusion_oracle.cg on the full workload t{n_t}xl{n_l}xz{n_z}x{xy}x{xy} (N={full_n})" if full else
            f"oracle/fusion_oracle.cg on t{t_use}xl{n_l}xz{n_z}x{xy}x{xy} (N={n_sample}: host has {avail / 2**30:.0f} GiB "
            f"available, the full workload needs ~{need * 1.3 / 2**30:.0f}), it/s scaled by N_sample/N_full")
    out = {"value": its * n_sample / full_n, "unit": "it/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": model, "extrapolated": not full,
            "sample": f"{what}; one call of {iters} iterations: {', '.join(f'{t:.2f}' for t in tm['iter_s'])} s each; b and r0 "
                      f"paid once: {tm['setup_s']:.2f} s; constructor (baseline init + normalisation): {t_prep:.2f} s "
                      "(both excluded from it/s, like the GPU figure)",
            "iter_s": tm["iter_s"], "setup_b_r0_s": tm["setup_s"], "constructor_s": t_prep}
    return out, (spc, cmos, x_cpu, iters)


def _rel_l2_vs_collapsed_on_gpu(x, x0, a, g, r):
    """|| x - (x0 + B1 a + B2 g) || / || . || in float64 on the GPU, one time bin at a time (x_k = x0 + bcast(a_k) +
    bcast(g_k) is what the float64 small-space CG of oracle/fusion_oracle.collapsed_cg returns)."""
    a, g = a.to(x.device), g.to(x.device)
    num = torch.zeros((), dtype=torch.float64, device=x.device)
    den = torch.zeros((), dtype=torch.float64, device=x.device)
    for t in range(x.shape[0]):
        ref = x0[t].double() + a[None] + g[t].repeat_interleave(r, 1).repeat_interleave(r, 2)[:, None]
        num += ((x[t].double() - ref) ** 2).sum()
        den += (ref ** 2).sum()
    return float(torch.sqrt(num / den))


def parity_vs_cpu_oracle(sample, dev, r):
    """The iterate the CPU leg has just computed with the reference's op sequence (fusion.py:435-456, eager fp32 torch on
    the host) is not thrown away: the HIP path runs the same call on the same inputs, both residual representations,
    and the relative L2 distance goes into the line, next to the distance of BOTH from the same recurrence in float64
    (oracle/fusion_oracle.collapsed_cg).  At configs[1]'s size the reference's own fp32 dot products over 2^30 elements
    are the larger error by three orders of magnitude (alpha off by 3e-3, profiles/r03_c2_oracle_vs_hip_vs_fp64.txt), so:
    ok = within north_star's 1e-4 of the CPU oracle, OR the whole distance is the oracle's own rounding (the HIP
    iterate within 1e-6 of the float64 one).  Not ok -> the bench FAILS (exit code 3)."""
    from datafusion_amd import FusionCG
    from oracle import fusion_oracle as O
    spc, cmos, x_cpu, iters = sample
    out = {"iterations": iters, "checker": "oracle/fusion_oracle.cg (eager fp32 torch-CPU restatement of fusion.py:423-456)",
           "shape": list(x_cpu.shape), "bar": 1e-4}
    worst = worst64 = 0.0
    for residual in ("factored", "explicit"):
        f = FusionCG(spc, cmos, weights={"spatial": 0.5, "spectro_temporal": 0.5}, init_type="baseline", tol=None,
                     mask_noise=False, total_energy=1, device=dev, seed=42, residual=residual)
        x0 = f.x.clone()
        f._project(x0)
        u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
        x, _, _ = f(iters, return_numpy=False)
        if residual == "factored":
            a, g, _, _ = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, r, 0.5, 0.5, iters)
            x_cpu_dev = x_cpu.to(x.device)
            out["rel_l2_cpu_oracle_vs_float64"] = _rel_l2_vs_collapsed_on_gpu(x_cpu_dev, x0, a, g, r)
            del x_cpu_dev
        e64 = _rel_l2_vs_collapsed_on_gpu(x, x0, a, g, r)
        e = float(O.rel_l2(x.cpu(), x_cpu))
        out[f"rel_l2_vs_cpu_oracle_{residual}"], out[f"rel_l2_vs_float64_{residual}"] = e, e64
        worst, worst64 = max(worst, e), max(worst64, e64)
        del f, x, x0
    out["rel_l2_vs_cpu_oracle"], out["rel_l2_vs_float64"] = worst, worst64
    out["ok"] = bool(worst <= out["bar"] or worst64 <= 1e-6)
    return out


def bench_adam(a, world, rank, dev, group, use_dist):
    """Secondary metric (BASELINE.json configs[4]): FusionAdam iterations/s, lr = 1e-8 as analysis/runtime.py:95."""
    from datafusion_amd import FusionAdam
    n_t, n_l, nz_gpu, xy, r = WORKLOADS[a.workload]
    spc, cmos = synthetic(n_t, n_l, nz_gpu * world, xy, r)
    f = FusionAdam(spc, cmos, weights={"spatial": 0.5, "spectro_temporal": 0.5}, init_type="baseline", tol=None,
                   device=dev, group=group, fused=(a.method == "adam"), moment1=a.moment1, placement=a.placement)
    n_local = f.k.N
    f(lr=1e-8, max_iterations=a.warmup, return_numpy=False)
    main_kernel = "adamf" if a.moment1 == "factored" else "adam"
    f.k.timing, f.k.timing_names = {}, ({main_kernel} if a.method == "adam" else set())
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    f(lr=1e-8, max_iterations=a.steps, return_numpy=False)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    timing = f.k.timing
    f.k.timing, f.k.timing_names = {}, None          # breakdown of the other launches: a second, untimed run
    f(lr=1e-8, max_iterations=a.steps, return_numpy=False)
    torch.cuda.synchronize()
    for name, evs in f.k.timing.items():
        timing.setdefault(name, evs)
    f.k.timing = None
    kern = {name: sum(e0.elapsed_time(e1) for e0, e1 in evs) / a.steps for name, evs in timing.items()}
    its = a.steps / dt
    out = {"metric": "FusionAdam iters/sec (secondary; BASELINE.json configs[4] shape class)", "value": its * world,
           "iters_per_s": its, "unit": "it/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"FusionAdam ({a.method}) t{n_t} x l{n_l} x z{nz_gpu * world} x {xy} x {xy}, r={r}, lr=1e-8",
                      "n_elements": n_local * world}, "kernel_ms": kern}
    if a.method == "adam":
        fact = a.moment1 == "factored"
        name, words = ("adamf", 4) if fact else ("adam", 6)       # x, m2 (and m1) read + written
        ach = 4.0 * words * n_local / (kern[name] * 1e-3) / 1e9
        out["config"]["moment1"] = a.moment1
        traffic, t_src = None, "none: PMC passes exist for the factored first moment on the c2 workload, one GPU, no mask"
        if fact and a.workload == "c2" and world == 1:
            # the PMC record of tools/profile_all.sh (its `--method adam` passes), replayed under the same conditions as
            # the CG line's: same kernel source (sha256), same workload
            import hashlib
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_RECORD)))
                src = hashlib.sha256(open(os.path.join(ROOT, "datafusion_amd", "csrc", "dfx_kernels.hip"), "rb").read()).hexdigest()
                if pmc.get("kernel_source_sha256") == src and pmc["workload"] == a.workload:
                    traffic = pmc["kernels"][f"k_adam_fast<{f.k.plan.zc}, {r}, false, true, false>"]["traffic_bytes"]
                    t_src = (f"replayed from profiles/{PMC_RECORD} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                             f"command on this kernel source, sha256 {src[:12]}; FETCH_SIZE doubled per MI355X_MICROARCH.md): "
                             "not a measurement of this run")
                else:
                    t_src = f"none: profiles/{PMC_RECORD} was taken on another build of dfx_kernels.hip"
            except (OSError, KeyError, ValueError) as e:
                t_src = f"none: {type(e).__name__} reading profiles/{PMC_RECORD}"
        out["roofline"] = {"bound": "hbm", "kernel": f"k_adam_fast<{f.k.plan.zc},{a.moment1} m1>", "achieved": ach,
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                           "traffic_source": t_src, "bytes_per_launch": 4 * words * n_local, "avg_launch_ms": kern[name]}
    if rank == 0:
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# one MI355X, the whole volume, factored residual (profiles/r03_bench_matrix.txt, r03_whole_gpu_series.txt)
COMMITTED_ONE_GPU_ITS = {"c3": 11.1, "c4": 11.06}
BASELINE_CONFIG = {"c3": "BASELINE configs[2]", "c4": "BASELINE configs[3]"}


def strong_scaling_leg(a, world, rank, dev, group, run, roofline, workload="c3", with_anchor=True):
    """N > 1 only, same invocation as the weak-scaled headline: BASELINE.json configs[2] (t512 x l32 x z32 x 256 x 256,
    2^35 elements, 128 GiB per vector - more than one GPU holds with the reference's three vectors) STRONG-scaled:
    32 / N planes per rank, total work fixed.  The one-GPU anchor is measured here too, FIRST (before the sharded
    legs have used the memory): rank 0 alone runs the whole volume on its GPU (it fits with the factored residual:
    x + p = 256 GiB) while the other ranks wait at a barrier; three timed blocks, the median is the anchor and their
    spread is reported.  A failure of the anchor (e.g. not enough free memory) is recorded, not fatal: the efficiency
    against the committed one-GPU figure is always there.
    workload="c4", with_anchor=False: BASELINE configs[3] (t256 x l16 x z32 x 512 x 512) the same way, against the
    committed one-GPU figure only (run at N = 8, the GPU count that config names)."""
    strong_wl = os.environ.get("DFX_BENCH_STRONG_WORKLOAD", workload)     # rehearsals on one GPU use a smaller volume
    n_t, n_l, n_z, xy, r = WORKLOADS[strong_wl]
    if n_z % world:
        return {"skipped": f"{n_z} z-planes do not split evenly over {world} ranks"}
    steps, warmup = min(a.steps, 20), min(a.warmup, 3)
    spc, cmos = synthetic(n_t, n_l, n_z, xy, r)
    anchor = None
    if rank == 0 and with_anchor:
        try:
            from datafusion_amd import placement
            need = int(2 * 4 * n_t * n_l * n_z * xy * xy * 1.03) + (6 << 30)      # x + p, small arrays, workspaces
            if placement.available_bytes(dev) < need:
                raise RuntimeError(f"only {placement.available_bytes(dev) / 2**30:.0f} GiB free on rank 0's GPU, the "
                                   f"whole volume needs {need / 2**30:.0f}")
            one = run("factored", spc=spc, cmos=cmos, grp=None, steps=min(steps, 10), warmup=2, repeats=3)
            blocks = one["block_ms_per_step"]
            anchor = {"iters_per_s": one["its"], "ms_per_step": one["dt"] / one["steps"] * 1e3,
                      "block_ms_per_step": blocks, "anchor_spread": (max(blocks) - min(blocks)) / sorted(blocks)[len(blocks) // 2],
                      "sweep_ms": one["sweep_ms"], "placement": one["placement"],
                      "roofline_frac": one["sweep_bytes"] / (one["sweep_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                      "note": "the whole volume on rank 0's GPU alone, factored residual, measured first in this "
                              "invocation; median of three timed blocks"}
        except Exception as e:  # noqa: BLE001 - the anchor must not take the sharded measurement down with it
            anchor = {"failed": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
    dist.barrier()
    res = run(a.residual, spc=spc, cmos=cmos, grp=group, steps=steps, warmup=warmup)
    its = res["its"]
    ok = anchor is not None and "iters_per_s" in anchor
    committed = COMMITTED_ONE_GPU_ITS.get(strong_wl)
    out = {"workload": f"FusionCG t{n_t} x l{n_l} x z{n_z} x {xy} x {xy}, r={r} "
                       f"({BASELINE_CONFIG.get(strong_wl, 'REHEARSAL volume ' + strong_wl)}), {n_z // world} "
                       f"z-planes per GPU", "scaling": "strong", "n_gpus": world, "n_elements": res["n_local"] * world,
           "iters_per_s": its, "ms_per_step": res["dt"] / res["steps"] * 1e3, "steps": res["steps"],
           "block_ms_per_step": res["block_ms_per_step"],
           "hbm_moved_GBps_per_gpu": 4.0 * res["iter_words"] * res["n_local"] * its / 1e9,
           "kernel_ms": res["kern"], "roofline": roofline(res, a.residual, (None, "none: not profiled"), r=r),
           "comm": res.get("comm"), "placement": res["placement"], "one_gpu_anchor": anchor,
           "strong_scaling_efficiency": (its / (world * anchor["iters_per_s"])) if ok else None,
           "strong_scaling_efficiency_vs_committed_one_gpu_figure": its / (world * committed) if committed else None,
           "committed_one_gpu_iters_per_s": committed}
    return out


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start one rank per GPU as CHILD processes
    (python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>), pass rank 0's ONE JSON line through
    to stdout, and return non-zero - with the tail of the children's stderr repeated - if any rank fails.  This parent
    never initialises a GPU (torch.cuda.device_count() does not) and never replaces itself with another program."""
    import collections
    import socket
    import subprocess
    import threading
    found = torch.cuda.device_count()
    if found < n_gpus and not (found >= 1 and os.environ.get("DFX_BENCH_SHARE_GPU") == "1"):
        sys.stderr.write(f"bench.py --gpus {n_gpus}: needs {n_gpus} GPUs, found {found}\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stderr.write("bench.py: starting " + " ".join(cmd) + "\n")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    tail = collections.deque(maxlen=60)

    def pump():
        for ln in p.stderr:
            tail.append(ln)
            sys.stderr.write(ln)
            sys.stderr.flush()

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    out = p.stdout.read()
    rc = p.wait()
    th.join(timeout=10)
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if rc == 0 and len(lines) == 1:
        os.write(_REAL_STDOUT, (lines[0] + "\n").encode())
        return 0
    sys.stderr.write(f"bench.py: the {n_gpus}-rank run failed (exit code {rc}, {len(lines)} result line(s)); last "
                     f"lines of its stderr:\n" + "".join(tail))
    return rc if rc else 1


def emit(obj):
    """The ONE JSON line goes to the process's original stdout; everything else any library prints on fd 1
    (RCCL prints a version banner there at communicator creation) was diverted to stderr in main()."""
    os.write(_REAL_STDOUT, (json.dumps(obj) + "\n").encode())


_REAL_STDOUT = 1


def main():
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--method", default="cg", choices=["cg", "adam", "adam-autograd"],
                    help="cg (the headline metric) | adam: fused dfx_adam_step | adam-autograd: autograd.Function + torch Adam")
    ap.add_argument("--no-defer", action="store_true", help="textbook ordering (x updated with r): 8N words/iter")
    ap.add_argument("--residual", default="factored", choices=["factored", "explicit"],
                    help="factored (default): r kept as its two small-space factors, 4N words/iter | explicit: r is a "
                         "third 5-D vector, 7N words/iter")
    ap.add_argument("--moment1", default="factored", choices=["factored", "explicit"],
                    help="--method adam: first moment as two small-space factors (4N words/iter) or a 5-D vector (6N)")
    ap.add_argument("--comm-slices", default="auto",
                    help="launches a z-sharded sweep is cut into (overlap of the v all-reduce): auto or a number")
    ap.add_argument("--placement", default="rank", choices=["rank", "probe", "plain"],
                    help="placement of vectors >= 8 GiB: rank (by measured memory class, default) | probe (round-2 candidate probe) | plain")
    ap.add_argument("--single", action="store_true", help="time only --residual, not the other representation as well")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--target-blocks", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed blocks of exactly --steps iterations each; value / ms_per_step are the median block's")
    a = ap.parse_args()

    if "RANK" not in os.environ and (a.gpus > 1 or os.environ.get("DFX_FORCE_COMM") == "1"):
        # `python bench.py --gpus N` as the driver runs it: this process starts the N ranks itself and never touches a GPU
        sys.exit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("DFX_BENCH_SHARE_GPU") == "1":
        # rehearsal of the N-rank control flow on a box with fewer GPUs: ranks share devices (with DFX_BENCH_BACKEND=gloo,
        # RCCL refuses two ranks on one GPU); the numbers of such a run mean nothing, its one line and exit code do
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    group = None
    # DFX_FORCE_COMM=1 under torch.distributed.run: take the RCCL code path even with one rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or (os.environ.get("DFX_FORCE_COMM") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DFX_BENCH_BACKEND", "nccl")             # "nccl" is RCCL; "gloo" only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from datafusion_amd import FusionCG
    if a.method != "cg":
        return bench_adam(a, world, rank, dev, group, use_dist)
    n_t, n_l, nz_gpu, xy, r = WORKLOADS[a.workload]
    n_z = nz_gpu * world
    spc, cmos = synthetic(n_t, n_l, n_z, xy, r)

    def run(residual, spc=spc, cmos=cmos, grp=group, steps=a.steps, warmup=a.warmup, repeats=a.repeats):
        """Construct, warm up, time `repeats` blocks of exactly `steps` iterations of one residual mode (each block
        bracketed by barrier + synchronize on both sides, max over ranks); the MEDIAN block is the result."""
        sharded = grp is not None
        f = FusionCG(spc, cmos, weights={"spatial": 0.5, "spectro_temporal": 0.5}, init_type="baseline", tol=None,
                     mask_noise=False, total_energy=1, device=dev, seed=42, group=grp, defer_x=not a.no_defer,
                     target_blocks=a.target_blocks, residual=residual, placement=a.placement, comm_slices=a.comm_slices if a.comm_slices == "auto" else int(a.comm_slices))
        fact = residual == "factored"
        sweep_name = ("sweep2" if a.no_defer else "sweep1") + ("f" if fact else "")

        def fence_():
            torch.cuda.synchronize()
            if sharded:
                dist.barrier()
            torch.cuda.synchronize()

        f.start()
        f.iterate(warmup)
        # the timed region: HIP events bracket the dominant kernel only (roofline.achieved comes from them) ...
        f.k.timing, f.k.timing_names = {}, {sweep_name}
        dts = []
        for _ in range(max(1, repeats)):
            fence_()
            t0 = time.perf_counter()
            f.iterate(steps)
            fence_()
            dts.append(time.perf_counter() - t0)
        timing = f.k.timing
        # ... the other entry points, and the waits for collectives, are timed in a second, untimed loop (kernel_ms and
        # comm are breakdowns, not the metric)
        f.k.timing, f.k.timing_names = {}, None
        f.comm_timing = {} if sharded else None
        f.iterate(steps)
        torch.cuda.synchronize()
        detail, f.k.timing = f.k.timing, None
        for name, evs in detail.items():
            timing.setdefault(name, evs)
        f.finish()  # raises if the residual ever went non-finite
        if sharded:
            t = torch.tensor(dts, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = [float(v) for v in t]
        dt = sorted(dts)[len(dts) // 2]
        # per-iteration time of each entry point from the HIP events recorded on the launch stream inside the timed
        # blocks (the dominant kernel: all blocks) or the detail loop (a z-sharded sweep is cut into several launches:
        # sum them)
        kern = {name: sum(e0.elapsed_time(e1) for e0, e1 in evs) / (steps * (len(dts) if name == sweep_name else 1))
                for name, evs in timing.items()}
        # words the sweep must move: x, p read + written (deferred x update) or p read + written, plus
        # the read of r when it is an explicit 5-D vector
        sweep_words = (2 if a.no_defer else 4) + (0 if fact else 1)
        res = {"dt": dt, "its": steps / dt, "steps": steps, "kern": kern, "sweep_ms": kern[sweep_name],
               "block_ms_per_step": [d / steps * 1e3 for d in dts],
               "sweep_bytes": sweep_words * 4 * f.k.N,
               "iter_words": sweep_words + (3 if a.no_defer else 0) + (0 if fact else 2),   # + dfx_axpy, + dfx_bcast
               "launches": len(timing[sweep_name]) // (steps * len(dts)), "zc": f.k.plan.zc, "n_local": f.k.N,
               "pair_probe": getattr(f.k, "pair_probe", None), "placement": getattr(f.k, "placement_info", None),
               "plan": {k: getattr(f.k.plan, k) for k in ("fast", "zc", "n_zchunks", "steps", "n_chunks", "cpb", "n_blocks")}}
        if sharded:
            # what the compute stream spent waiting for collectives, per iteration (this rank): the all-reduce of the
            # last band of v that the sweep did not hide, the 4-double and the 1-double all-reduces
            waits = {name: sum(e0.elapsed_time(e1) for e0, e1 in evs) / steps for name, evs in f.comm_timing.items()}
            res["comm"] = {"world_size": dist.get_world_size(grp), "backend": dist.get_backend(grp),
                           "rccl_version": ".".join(map(str, torch.cuda.nccl.version())),
                           "v_allreduce_bytes": 4 * f.k.v.numel(), "sweep_slices": f.comm_slices,
                           "allreduce_probe": f.comm_probe, "exposed_wait_ms_per_iteration": waits,
                           "exposed_wait_ms_total": sum(waits.values()),
                           "collectives_per_iteration": "v in %d band(s) + 4 doubles (UU, PP, XX, XP) + 1 double (RR)" % f.comm_slices}
        del f
        torch.cuda.empty_cache()
        return res

    def roofline(res, residual, traffic, r=r):
        ach = res["sweep_bytes"] / (res["sweep_ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": f"k_sweep_fast<{'P' if a.no_defer else 'XP'},{res['zc']},r{r},{residual}>",
                "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "traffic": traffic[0], "traffic_source": traffic[1],
                "bytes_per_launch": res["sweep_bytes"], "avg_launch_ms": res["sweep_ms"],
                "basis": "bytes the kernel must move (x, p read + written; + r read when explicit) / HIP-event time of "
                         "its launches inside the timed region"}

    def pmc_traffic(residual, zc):
        """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (counters cannot
        be collected from inside the process being profiled: tools/profile_all.sh runs this same command under
        rocprofv3 --pmc).  Returned only when that record was taken on the kernel source this run's library was built
        from (sha256 of dfx_kernels.hip) and on this workload; otherwise None.  -> (bytes or None, provenance)."""
        import hashlib
        path = os.path.join(ROOT, "profiles", PMC_RECORD)
        try:
            pmc = json.load(open(path))
            src = hashlib.sha256(open(os.path.join(ROOT, "datafusion_amd", "csrc", "dfx_kernels.hip"), "rb").read()).hexdigest()
            if pmc.get("kernel_source_sha256") != src:
                return None, f"none: profiles/{PMC_RECORD} was taken on another build of dfx_kernels.hip"
            if pmc["workload"] != a.workload or world != 1 or a.no_defer:
                return None, f"none: profiles/{PMC_RECORD} holds workload {pmc['workload']}, 1 GPU, deferred x update"
            rec = pmc["kernels"][f"k_sweep_fast<XP,{zc},r{r},{residual}>"]
            return rec["traffic_bytes"], (f"replayed from profiles/{PMC_RECORD} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                          f"of this command on this kernel source, sha256 {src[:12]}; FETCH_SIZE doubled per "
                                          "MI355X_MICROARCH.md): not a measurement of this run")
        except (OSError, KeyError, ValueError) as e:
            return None, f"none: {type(e).__name__} reading profiles/{PMC_RECORD}"

    # DFX_BENCH_STRONG=1 (with DFX_FORCE_COMM=1): rehearse the strong-scaling leg with one rank on a 1-GPU box
    force_strong = os.environ.get("DFX_BENCH_STRONG") == "1"
    want_strong = group is not None and ((a.workload == "c2" and world > 1) or force_strong)
    # the strong-scaled configs[2] leg first: it is the measurement north_star's scaling target is about, and its
    # one-GPU anchor wants the memory before anything else has used it
    strong = strong_scaling_leg(a, world, rank, dev, group, run, roofline) if want_strong else None
    # BASELINE configs[3] names 8 GPUs: measured in the same invocation when that is the GPU count
    config4 = (strong_scaling_leg(a, world, rank, dev, group, run, roofline, workload="c4", with_anchor=False)
               if want_strong and (world == 8 or os.environ.get("DFX_BENCH_CONFIG4") == "1") else None)
    main_res = run(a.residual)
    other = "explicit" if a.residual == "factored" else "factored"
    # the explicit residual needs a third resident 5-D vector: skip that leg where it cannot fit (configs 3 / 4 whole
    # on one GPU hold x + p = 256 GiB of the 288 GiB)
    n_vec = 3 if other == "explicit" else 2
    fits = n_vec * 4 * main_res["n_local"] < 0.93 * torch.cuda.mem_get_info()[1]
    other_res = None if (a.single or not fits) else run(other)
    n_local = main_res["n_local"]
    n_total = n_local * world
    its, dt = main_res["its"], main_res["dt"]
    out = {
        "metric": "FusionCG iters/sec + achieved HBM GB/s vs peak, 5D fp32 (t,l,z,x,y), 1-8 GPU",
        # whole-job aggregate: every rank advances its 16-plane slab by one iteration per step, so the job does
        # `world` slab-iterations per CG iteration (weak scaling: the volume grows with the GPU count)
        "value": its * world, "unit": "it/s", "iters_per_s": its, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        # `repeats` timed blocks of exactly `steps` iterations each; value / ms_per_step are the MEDIAN block's
        "timed_blocks": {"repeats": len(main_res["block_ms_per_step"]), "ms_per_step": main_res["block_ms_per_step"],
                         "min": min(main_res["block_ms_per_step"]), "max": max(main_res["block_ms_per_step"])},
        "config": {"workload": f"FusionCG t{n_t} x l{n_l} x z{n_z} x {xy} x {xy}, r={r}, tol=None, baseline init "
                               f"({a.workload}; {nz_gpu} z-planes per GPU)", "n_elements": n_total,
                   "parallelism": f"z-shard x{world}",
                   "value_definition": "CG iterations/s x z-shards (slab-iterations/s): equals iterations/s of the "
                                       "configs[1] volume at 1 GPU; iters_per_s is the iteration rate of the whole "
                                       "sharded volume", "x_update": "textbook" if a.no_defer else "deferred",
                   "residual": a.residual, "words_moved_per_iteration": f"{main_res['iter_words']} N",
                   "sweep_launches_per_iteration": main_res["launches"], "plan": main_res["plan"],
                   "placement_probe_GBps": main_res["pair_probe"], "placement": main_res["placement"]},
        # whole iteration, all GPUs: SURVEY 8(d) prices an iteration at 32 N bytes and asks for that basis to be quoted
        # as "effective"; the build moves fewer (16 N with the factored residual), so "effective" may exceed the peak:
        # the honest bandwidth figure is hbm_moved_GBps and the per-kernel roofline below
        "hbm_effective_GBps": 32.0 * n_total * its / 1e9,
        "hbm_effective_frac_of_peak": 32.0 * n_local * its / 1e9 / HBM_PEAK_GBPS,
        "hbm_moved_GBps": 4.0 * main_res["iter_words"] * n_total * its / 1e9,
        "kernel_ms": main_res["kern"],
        "roofline": roofline(main_res, a.residual, pmc_traffic(a.residual, main_res["zc"])),
    }
    if other_res is not None:
        # the same timed loop with the other residual representation, measured in this same invocation
        o = other_res
        out["other_residual_mode"] = {
            "residual": other, "value": o["its"] * world, "iters_per_s": o["its"], "ms_per_step": o["dt"] / a.steps * 1e3,
            "words_moved_per_iteration": f"{o['iter_words']} N",
            "hbm_moved_GBps": 4.0 * o["iter_words"] * n_total * o["its"] / 1e9, "kernel_ms": o["kern"],
            "roofline": roofline(o, other, pmc_traffic(other, o["zc"]))}
    if "comm" in main_res:
        out["comm"] = main_res["comm"]
    if strong is not None:
        out["config3_strong_scaling"] = strong
    if config4 is not None:
        out["config4_strong_scaling"] = config4
    failed = False
    if rank == 0:
        if not a.no_cpu and world == 1:
            out["cpu_baseline"], sample = cpu_baseline(n_t, n_l, n_z, xy, r)
            out["parity"] = parity_vs_cpu_oracle(sample, dev, r)
            failed = not out["parity"]["ok"]
            del sample
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.stderr.write("bench.py: the HIP iterate differs from the CPU oracle's by more than 1e-4 (relative L2) and from "
                         "the float64 recurrence by more than 1e-6\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
