#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the wavefront path tracer on BASELINE config 2
(Cornell box + 70,688-triangle OBJ mesh, 1920x1080, depth 8, 256 spp, seed 1337).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full render of the configuration: every rank renders its interleaved 8-row bands of the image
from a scene already resident in its HBM, then the HDR band buffers are gathered on rank 0 (RCCL).  Total work
per step is fixed, so scaling is "strong".  Rank 0 prints one JSON line.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.29 TB/s measured streaming copy
# Per-kernel records of tools/measure_solo.sh (rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE / WRITE_SIZE / VALU groups, pool as
# ONE group so kernels never overlap), one per workload: the command line each was taken with is in the file
SOLO_FILES = {
    ("cornell_mesh.scene", 1920, 1080, 8, 256): "profiles/r3_solo_cfg2.json",      # BASELINE configs[1]: this benchmark's default
    ("knot_glass.scene", 1920, 1080, 16, 128): "profiles/r3_solo_cfg4.json",       # configs[3] stand-in
    ("lucy_standin.scene", 3840, 2160, 12, 16): "profiles/r3_solo_cfg5.json",      # configs[4] stand-in: the scene lives in HBM
}
SOLO_FILE_CFG5 = os.path.join(ROOT, "profiles", "r3_solo_cfg5.json")
# the three kernels of an iteration, by the prefix of their timed (non-counting) instantiations
KERNEL_PREFIX = {"extend": "k_extend<false,", "shade": "k_shade<false,", "connect": "k_connect<false,"}


def measured_stream_gbs(torch, device):
    """Achievable HBM bandwidth on this box: float4-wide device copy of 1 GiB (read + write bytes / time)."""
    n = 1 << 28
    src = torch.empty(n, dtype=torch.float32, device=device)
    dst = torch.empty_like(src)
    src.fill_(1.0)
    for _ in range(2):
        dst.copy_(src)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        dst.copy_(src)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    del src, dst
    return 2.0 * 4.0 * n * reps / dt / 1e9


def baseline_metric():
    """The metric string exactly as BASELINE.json names it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Msamples/sec (whole node) at 1920\u00d71080, depth 8; PFM RMSE vs Embree ref"


def recorded_parity():
    """Full-frame parity of BASELINE configs[1] from the last tools/full_configs.py pass (RMSE against the oracle, the oracle's
    seed-to-seed noise floor N, mean-luminance ratio), or None."""
    try:
        row = None
        for name in ("r3_full_configs.json", "r2_full_configs.json", "r1_full_configs.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                row = json.load(open(path))["2"]
                break
        return {"rmse": row["rmse_vs_oracle"], "noise_floor_N": row["noise_floor_N"], "mean_luminance_ratio": row["mean_luminance_ratio"],
                "spp": row["parity_spp"], "pass": row["pass"]}
    except Exception:
        return None


def recorded_kernel(path, prefix):
    """One kernel's row of a tools/measure_solo.sh record (None when the file is absent): of the instantiations whose name starts with
    `prefix`, the one with the most kernel time.  Fields: dispatches, avg_ms (rocprofv3), hbm_bytes_per_dispatch (FETCH_SIZE + WRITE_SIZE +
    streamed reads / 2, profiles/r3_fetch_size_calibration.txt), hbm_gbs, hbm_frac_of_8TBs, valu_issue, valu_lane_utilisation."""
    try:
        with open(path) as f:
            rec = json.load(f)
        names = [k for k in rec["kernels"] if k.startswith(prefix) and rec["kernels"][k].get("avg_ms")]
        name = max(names, key=lambda k: rec["kernels"][k].get("dispatches", 0) * rec["kernels"][k].get("avg_ms", 0.0))
        row = dict(rec["kernels"][name])
        row["name"] = name
        row["command"] = rec.get("command", "")
        return row
    except (OSError, ValueError, KeyError, TypeError):
        return None


def kernel_roofline(row, live_ms, source):
    """The HBM roofline entry of one kernel, every figure from ONE committed record (bytes from the PMC passes over the rocprofv3
    average duration of the same command line); the duration this run measured with HIP events rides beside it."""
    if not row or not row.get("hbm_bytes_per_dispatch"):
        return None
    return {"kernel": row["name"], "bound": "hbm", "achieved": row.get("hbm_gbs"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": row.get("hbm_frac_of_8TBs"), "traffic": row["hbm_bytes_per_dispatch"], "avg_launch_ms": row.get("avg_ms"),
            "avg_launch_ms_live": round(live_ms, 4) if live_ms else None, "dispatches": row.get("dispatches"),
            "valu_issue": row.get("valu_issue"), "valu_lane_utilisation": row.get("valu_lane_utilisation"),
            "wave_cycles_waiting": row.get("wave_cycles_waiting"), "source": source}


def algorithmic_bytes(nodes, prims, shaded=0, tri_hits=0, samples=0):
    """64 B per node visit, 68 B per primitive test, 576 B per shaded hit,
    192 B per triangle hit, 40 B per sample."""
    return 64 * nodes + 68 * prims + 576 * shaded + 192 * tri_hits + 40 * samples


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cornell_mesh.scene"))
    ap.add_argument("--semantics", type=int, default=0,
                    help="PtrSettings.metalSemantics bit mask (31 = every Metal-only behaviour; 0 = the Embree-parity integrator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save", default="", help="write the last image as PFM (rank 0)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--solo", action="store_true",
                    help="profiling runs: the path-slot pool as ONE group on one stream, so kernels never overlap and rocprofv3 / PMC "
                         "figures are clean per-kernel numbers (the product default runs two groups, each with its k_connect beside the next k_extend)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo); numbers are not comparable")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    bands = importlib.import_module("metal-pathtracer-arm64_amd.bands")
    from scenes.gen_assets import ensure_assets

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # started without a launcher: run the documented launch line as a child process (nothing has touched the GPU yet)
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE is %d but --gpus is %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    if rank == 0:
        ensure_assets()
        # the large generated meshes a non-default scene names (file names ending in _<triangles>.ply; too big for the repository)
        import re
        from scenes.gen_assets import ensure_large_asset
        with open(args.scene) as f:
            for asset in re.findall(r"assets/(\w+_\d{6,}\.ply)", f.read()):
                ensure_large_asset(asset)
    if world > 1:
        dist.barrier()

    # ---- setup (untimed): parse, build BVH, upload; output buffers in HBM ----
    host = pt.HostScene.load(args.scene, os.path.join(ROOT, "scenes"))
    settings = host.settings_for(width=args.width, height=args.height, max_depth=args.depth, seed=1337, metalSemantics=args.semantics)
    bvh_build_s = 0.0
    cache_path = None
    if world > 1:
        # ONE BVH build for the node: rank 0 prepares the geometry on the host (no GPU call) and writes it to /dev/shm, the other
        # ranks read it (N concurrent 64-thread builds on one host would each take several times their solo time)
        cache_path = "/dev/shm/ptr_geometry_%s_%d.bin" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
        if rank == 0:
            bvh_build_s = pt.prepare_geometry(host.desc, cache_path)
        dist.barrier()
    t0 = time.time()
    scene = pt.DeviceScene(host.desc, local_rank, keepalive=host, prepared=cache_path)
    upload_s = time.time() - t0
    timings = scene.timings()
    if world == 1:
        bvh_build_s = timings["geometry_s"]
    if world > 1:
        dist.barrier()
        if rank == 0:
            try:
                os.remove(cache_path)
            except OSError:
                pass
    rows = bands.max_band_count(args.height, world) * bands.BAND_ROWS
    local = torch.zeros((rows, args.width, 3), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream(device)

    def step(want_stats):
        stats = scene.render_device(settings, args.spp, local.data_ptr(), stream.cuda_stream, rank, world,
                                    count=os.environ.get("PTR_BENCH_COUNTING", "0") == "1", want_stats=want_stats, solo=args.solo)
        image = bands.gather_bands(local, args.height, rank, world)
        return stats, image

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step(False)
    fence()
    t_start = time.perf_counter()
    trace_ms = shade_ms = connect_ms = tail_ms = 0.0
    trace_launches = 0
    image = None
    for _ in range(args.steps):
        stats, image = step(True)
        trace_ms += stats.traceKernelMs
        shade_ms += stats.shadeKernelMs
        connect_ms += stats.shadowKernelMs
        tail_ms += stats.tailKernelMs
        trace_launches += stats.traceLaunches
    fence()
    elapsed = time.perf_counter() - t_start
    cdev = device if args.backend == "nccl" else torch.device("cpu")   # where the small reductions live
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- algorithmic traffic: one counting render of the same kernels (untimed, deterministic) ----
    cstats = scene.render_device(settings, args.spp, local.data_ptr(), stream.cuda_stream, rank, world, count=True, want_stats=True)
    # ---- the dominant kernel alone: same render with the pool as ONE group, so no other kernel shares the chip with
    # k_extend while it is timed (the timed region above runs the pool's groups concurrently on separate streams) ----
    sstats = scene.render_device(settings, args.spp, local.data_ptr(), stream.cuda_stream, rank, world, count=False, want_stats=True,
                                 solo=True)
    counters = torch.tensor([cstats.extendNodesVisited, cstats.extendLeafPrimTests, cstats.nodesVisited, cstats.leafPrimTests,
                             cstats.shadedHits, cstats.triangleHits, cstats.extendRays, cstats.shadowRays],
                            dtype=torch.float64, device=cdev)
    kernel_times = torch.tensor([trace_ms, float(trace_launches)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        dist.all_reduce(kernel_times, op=dist.ReduceOp.MAX)
    fence()

    if rank == 0:
        total_samples = args.width * args.height * args.spp
        value = total_samples * args.steps / elapsed / 1e6
        ext_nodes, ext_prims, all_nodes, all_prims, shaded, tri_hits, ext_rays, sh_rays = [float(x) for x in counters.tolist()]
        launches = max(1.0, kernel_times[1].item())
        avg_launch_ms = kernel_times[0].item() / launches
        # dominant kernel = k_extend (closest-hit traversal).  Counters are per render and per rank-sum; the
        # per-launch figure divides by the launches of one render on one rank.
        launches_per_render = launches / args.steps
        ext_bytes_per_launch = algorithmic_bytes(ext_nodes, ext_prims) / world / launches_per_render
        achieved = ext_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        solo = None
        if world == 1 and sstats.traceLaunches > 0:
            solo_ms = sstats.traceKernelMs / sstats.traceLaunches
            solo_bytes = algorithmic_bytes(ext_nodes, ext_prims) / sstats.traceLaunches
            solo_rate = solo_bytes / (solo_ms * 1e-3) / 1e9
            solo = {"achieved": round(solo_rate, 1), "frac": round(solo_rate / HBM_PEAK_GBS, 4), "avg_launch_ms": round(solo_ms, 4),
                    "launches_per_render": int(sstats.traceLaunches), "alg_bytes_per_launch": round(solo_bytes),
                    "render_ms": round(sstats.totalSeconds * 1e3, 2)}
        path_bytes = algorithmic_bytes(all_nodes, all_prims, shaded, tri_hits, total_samples)
        path_gbs = path_bytes * args.steps / elapsed / 1e9 / world

        stream_gbs = measured_stream_gbs(torch, device)
        workload_key = (os.path.basename(args.scene), args.width, args.height, args.depth, args.spp)
        solo_rel = SOLO_FILES.get(workload_key) if (world == 1 and args.semantics == 0) else None
        solo_path = os.path.join(ROOT, solo_rel) if solo_rel else None
        # live launch durations with the pool as one group (HIP events on the kernels' stream, the extra untimed render above)
        live = {}
        if world == 1 and sstats.traceLaunches > 0:
            n = float(sstats.traceLaunches)   # one k_extend, one k_shade and one k_connect per iteration
            live = {"extend": sstats.traceKernelMs / n, "shade": sstats.shadeKernelMs / n, "connect": sstats.shadowKernelMs / n}
        per_kernel = {}
        for which, prefix in KERNEL_PREFIX.items():
            entry = kernel_roofline(recorded_kernel(solo_path, prefix) if solo_path else None, live.get(which), solo_rel)
            if entry:
                per_kernel[which] = entry
        step_ms = {"extend": trace_ms / args.steps, "shade": shade_ms / args.steps, "connect": connect_ms / args.steps}
        dominant = max(step_ms, key=step_ms.get)
        rec5 = recorded_kernel(SOLO_FILE_CFG5, KERNEL_PREFIX["extend"])
        # `roofline` describes the kernel with the most kernel time in THIS run's timed region (kernel_ms_per_step below); the other
        # two kernels of an iteration are in roofline["kernels"].  Every frac = bytes of the committed PMC passes / the rocprofv3 average
        # launch of the same record / 8 TB/s - recomputable from the one file named in `source`.
        roofline = dict(per_kernel.get(dominant) or {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": None, "traffic": None, "source": "no PMC record for this command line"})
        roofline.update({
            "dominant_by": "kernel_ms_per_step of this run: " + ", ".join("%s %.1f" % (k, v) for k, v in step_ms.items()),
            "kernels": per_kernel,
            "peak_measured": round(stream_gbs, 1),
            "note": "Physical HBM-side traffic (PMC) over launch time, per kernel, with the path-slot pool as one group so nothing else is on the "
                    "chip.  On this workload no kernel is bound by HBM: the scene (a few MB of four-wide nodes and triangles) is served by L1 / L2 / "
                    "Infinity Cache and HBM carries the path-state stream; k_extend and k_connect are bound by VALU issue at partial lane "
                    "occupancy, k_shade by VALU issue and its own dependent loads (valu_issue, valu_lane_utilisation, wave_cycles_waiting of each "
                    "entry).  `algorithmic` is SURVEY.md section 8(d)'s reference-layout figure for k_extend (64 B per binary node visit, 68 B per "
                    "primitive test, counted by the counting build on the binary tree): a cache-served rate that can exceed the HBM peak.  "
                    "`hbm_resident_scene` is k_extend on BASELINE configs[4], whose 1 GB of four-wide nodes and 1.4 GB of triangles do come from HBM.",
            "algorithmic": {
                "kernel": "k_extend",
                "bytes_per_launch": solo["alg_bytes_per_launch"] if solo else None,
                "achieved": solo["achieved"] if solo else None,
                "frac": solo["frac"] if solo else None,
                "launches_per_render": solo["launches_per_render"] if solo else None,
                "timed_region": {"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_launch_ms, 4),
                                 "bytes_per_launch": round(ext_bytes_per_launch), "launches_per_render": round(launches_per_render, 1),
                                 "note": "inside the timed region the pool runs as concurrent groups (two of 16 Mi slots, each with its k_connect beside the "
                                         "next k_extend): a launch covers part of the pool and shares the chip with other kernels"},
                "whole_path_gbs_per_gpu": round(path_gbs, 1),
                "bytes_per_sample": round(path_bytes / total_samples, 1),
                "rays_per_sample": round((ext_rays + sh_rays) / total_samples, 3),
            },
            "hbm_resident_scene": ({"workload": "BASELINE configs[4] stand-in (29 M triangles), " + rec5["command"].split("--no-cpu-baseline")[-1].strip(),
                                    "kernel": rec5["name"], "avg_launch_ms": rec5.get("avg_ms"), "traffic": rec5.get("hbm_bytes_per_dispatch"),
                                    "achieved": rec5.get("hbm_gbs"), "frac": rec5.get("hbm_frac_of_8TBs"),
                                    "gather_ceiling_gbs": 3450.0,
                                    "valu_issue": rec5.get("valu_issue"), "valu_lane_utilisation": rec5.get("valu_lane_utilisation"),
                                    "source": "profiles/r3_solo_cfg5.json; gather ceiling = random 64 B records from a 2 GiB table, "
                                              "profiles/r3_fetch_size_calibration.txt"} if rec5 else None),
        })
        valu = None
        ext = per_kernel.get("extend")
        if ext and ext.get("valu_issue") is not None:
            # (instructions and active cycles come from two PMC passes of the record: their quotient can read a per cent or two above 1)
            issue = min(1.0, ext["valu_issue"])
            valu = {"kernel": ext["kernel"], "issue": issue, "issue_recorded": ext["valu_issue"], "lane_utilisation": ext["valu_lane_utilisation"],
                    "useful": round(issue * ext["valu_lane_utilisation"], 4),
                    "source": "recorded: %s (SQ_INSTS_VALU x 4 / (GRBM_GUI_ACTIVE x 128); SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64))" % solo_rel,
                    "note": "the binding resource of the traversal kernels on this workload: share of SIMD cycles that issue a VALU instruction x share of "
                            "the 64 lanes those instructions keep busy"}
        out = {
            "metric": baseline_metric(),   # throughput is `value`; the RMSE half of the metric is the `parity` object
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[1]: Cornell box + one 70,688-triangle OBJ mesh" if os.path.basename(args.scene) == "cornell_mesh.scene"
                             else "scene " + os.path.basename(args.scene))
                            + ", %dx%d, depth %d, %d spp, seed 1337" % (args.width, args.height, args.depth, args.spp)
                            + (", metalSemantics %d" % args.semantics if args.semantics else ""),
                "partition": "8-row bands round-robin over %d rank(s), RCCL gather of the HDR buffer" % world,
                "bvh_build_s": round(bvh_build_s, 3),     # host: bake, SAH build, leaf order, wide nodes (once per node: rank 0)
                "upload_s": round(upload_s - (timings["geometry_s"] if world == 1 else 0.0), 3),   # shading tables + copies to the device (+ reading the prepared geometry when N > 1), this rank
                "geometry_shared": bool(world > 1),
            },
            "roofline": roofline,
            "valu": valu,
            "kernel_ms_per_step": {
                "extend": round(trace_ms / args.steps, 2),
                "shade": round(shade_ms / args.steps, 2),
                "connect": round(connect_ms / args.steps, 2),
                "tail": round(tail_ms / args.steps, 2),
            },
        }

        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib  # test infrastructure; used here only for the cpu_baseline leg

            osc = oracle_lib.OracleScene(host)
            threads = os.cpu_count() or 1
            # bounded sample: the middle 640 rows of the same frame (covers walls, mesh and light like the whole
            # frame does), spp scaled from a 1-spp calibration pass to about 15 s of CPU work (10-30 s window)
            rows = min(args.height, 640)
            y0 = max(0, ((args.height - rows) // 2 // 16) * 16)
            y1 = min(args.height, y0 + rows)
            osc.render(settings, 1, threads=threads, rows=(y0, y1))                      # warm the thread pool / caches
            _, secs4, _ = osc.render(settings, 4, threads=threads, rows=(y0, y1))       # calibration pass
            cpu_spp = int(min(args.spp, max(1, round(4.0 * 15.0 / max(secs4, 1e-3)))))
            ref, secs, _ = osc.render(settings, cpu_spp, threads=threads, rows=(y0, y1))
            cpu_samples = args.width * (y1 - y0) * cpu_spp
            # the second half of BASELINE's metric ("PFM RMSE vs Embree ref"): the same strip at the same spp through the HIP
            # path (untimed), compared with the oracle image that was just timed; the full-frame protocol with its noise
            # floor is tools/full_configs.py (profiles/r1_full_configs.json)
            scene.render_device(settings, cpu_spp, local.data_ptr(), stream.cuda_stream, 0, 1, want_stats=True)
            gpu = local[:args.height].cpu().numpy()[y0:y1].astype(np.float64)
            cpu = ref[y0:y1].astype(np.float64)
            lum = np.array([0.2126, 0.7152, 0.0722])
            out["parity"] = {
                "rmse_vs_oracle": float(np.sqrt(np.mean((gpu - cpu) ** 2))),
                "mean_luminance_ratio": round(float((gpu @ lum).mean() / max((cpu @ lum).mean(), 1e-30)), 5),
                "rows": [y0, y1], "spp": cpu_spp,
                "recorded_full_frame": recorded_parity(),
            }
            out["cpu_baseline"] = {
                "value": round(cpu_samples / secs / 1e6, 4),
                "unit": "Msamples/s",
                "cores": threads,
                "kind": "port",
                "sample": "rows %d-%d of the same %dx%d frame at %d spp (%.1f s of CPU work), oracle restatement of the "
                          "reference's Embree-path integrator with its own scalar BVH" % (y0, y1, args.width, args.height, cpu_spp, secs),
            }
        if args.save and image is not None:
            pt.write_image(args.save, image.cpu().numpy(), "pfm")
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
