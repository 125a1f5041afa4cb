#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-traced frame BASELINE.json's metric is quoted on.

    python bench.py --gpus N --steps K --warmup W

Workload at N = 1 (BASELINE.json configs[2] class, the one the metric "seconds to 1024 spp at 1080p"
and the target "~1M-triangle mesh" name): the stanford bunny subdivided twice (1 111 216 triangles;
the reference's dragon.obj is not distributed), Plane floor, 1k HDR environment, 1920x1080,
1024 spp (32x32 stratified), max depth 5.  One step = one full frame of the hot path (camera rays,
kd-tree traversal, shading, NEE + MIS connections, per-pixel accumulation) with the scene and the
image resident in HBM.  A ray is one intersectKdTree-equivalent query (closest-hit or any-hit)
actually traced.  configs[1] (two 69 451-triangle bunnies, 256 spp) is measured after it and
reported as `secondary` in the same JSON line.

N > 1 (launched by torch.distributed.run, one rank per GPU): sample split, no exchange while
rendering, the radiance-sum images are added onto rank 0 with one RCCL reduce per frame.
  strong (the headline `value`, "scaling": "strong"): the 1024 samples of the 32 x 32 grid of the SAME frame
         are divided among the ranks (BASELINE configs[2]: "1080p 1024 spp, sample-split across 8 GPUs");
         ms_per_step is then the measured time to 1024 spp at 1080p on N GPUs;
  weak   (`weak_split`, a side figure): rank r renders sample indices [r*1024, (r+1)*1024) of a 32 x 32N grid.
At N = 1 the line also carries `strong_proxy`: what ONE GPU of a 2-, 4- and 8-way strong split has to do
(samples [0, 1024/N) of the same grid) timed like the headline, and efficiency(N) = t(1024 spp) /
(N * t(1024/N spp)) -- the scaling a node can reach before the reduce, measurable without the node.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the traversal kernel with the most device time: algorithmic bytes / HIP-event time
                against the 8 TB/s HBM peak, and the HBM traffic the hardware counters saw
  cpu_baseline  the CPU oracle timed on this host's cores on a bounded sample of the same frame
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
MAX_DEPTH = 5
SEED = 1
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    "million_bunny": {
        "spp": (32, 32), "cpu_spp": 64, "mis_kernel": "connect_mis_occlusion_kernel",  # no emitter besides the environment
        "label": "stanford bunny subdivided twice (1 111 216 triangles), glossy, plane floor, HDR env, 1920x1080, %d spp, "
                 "max depth 5 (BASELINE configs[2] class; dragon.obj is not distributed)",
        "data": "stanford-bunny.obj (69 451 triangles, 1->4 subdivided twice, vertices x80) + brown_photostudio_06_1k.hdr committed "
                "under assets/; camera, material and floor synthetic",
    },
    "bunny_pair": {
        "spp": (16, 16), "cpu_spp": 32, "mis_kernel": "connect_mis_occlusion_kernel",
        "label": "stanford-bunny diffuse+glossy, 1920x1080, %d spp, max depth 5, HDR env (BASELINE configs[1])",
        "data": "stanford-bunny.obj (69 451 triangles, x2 instances, vertices x20) + brown_photostudio_06_1k.hdr committed under "
                "assets/; camera, materials and floor synthetic",
    },
}


def own_bytes(ts, n):
    """Algorithmic bytes of n rays: 36 B ray in + 16 B hit out each, 8 B per inner node and per leaf
    header visited, 4 B per leaf reference, 36 B (p1, e1, e2) per primitive test."""
    return 52.0 * n + 8.0 * ts.inner + 8.0 * ts.leaves + 4.0 * ts.refs + 36.0 * ts.prim_tests


def profile_traffic(workload, kernel):
    """From the committed counter passes (profiles/hbm_traffic.json): HBM bytes per launch of `kernel`, the
    L2 hit rate of the same run, and the L2's read requests to the fabric per launch (one per missed line);
    Nones when the workload was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            w = json.load(f)["workloads"][workload]
        n = w["dispatches_per_step"].get(kernel)
        lines = w.get("fabric_read_requests_per_step", {}).get(kernel)
        return w["bytes_per_launch"].get(kernel), w.get("tcc_hit_rate", {}).get(kernel), (lines / n if lines and n else None)
    except Exception:
        return None, None, None


def line_fetch_ceiling():
    """The rate at which this GPU delivers 128-byte lines to random 32-byte gathers that miss every cache
    (raytracer_amd/gather_bench on a 4 GB table at the traversal kernels' 6 waves per SIMD), in G lines/s,
    measured now in a child process; None when the binary is missing."""
    exe = os.path.join(ROOT, "raytracer_amd", "gather_bench")
    try:
        out = subprocess.run([exe, "independent", "4096", "6", "32", "2000"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True, timeout=120).stdout
        row = [l for l in out.splitlines() if l.startswith("independent")][-1].split()
        return float(row[9])  # mode, record, "B", table, "MB", waves, G records/s, ns per step, GB/s, G lines/s, ...
    except Exception:
        return None


def measure(name, args, rank, world, dist, local_rank, with_cpu):
    """Builds workload `name`, renders warm-up + timed steps (weak split when world > 1) and returns
    the result fields (rank 0) or None (other ranks)."""
    import numpy as np  # noqa: F401
    import torch

    from raytracer_amd import abi, parallel, scenes
    from raytracer_amd.host import DeviceScene, render_params

    wl = WORKLOADS[name]
    line_ceiling = line_fetch_ceiling() if (rank == 0 and world == 1 and with_cpu) else None
    spp_x, spp_y = wl["spp"]
    if args.spp_y:
        spp_y = args.spp_y
    spp = spp_x * spp_y

    t0 = time.time()
    sc = getattr(scenes, name)(WIDTH, HEIGHT)
    # torchrun sets OMP_NUM_THREADS=1 for its workers: give the tree build its share of the host explicitly
    tree = sc.initKdTree(max(1, min(16, (os.cpu_count() or 1) // max(world, 1))))
    build_s = time.time() - t0
    flat = tree.flatten(int(os.environ.get("MRT_LAYOUT", "1")))
    dev = DeviceScene(flat, device=local_rank)
    n_nodes, n_refs, depth = tree.counts()

    img = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def timed(params, steps, serial=False):
        """`steps` frames bracketed by barrier + synchronize; returns (seconds max over ranks, rays and
        paths summed over ranks, this rank's per-class kernel ms, passes).  serial: one kernel at a
        time (MRT_RENDER_SERIAL), which is when the per-class kernel times are each kernel's own."""
        params.paths_per_pass = args.paths_per_pass
        params.flags = abi.MRT_RENDER_SERIAL if (serial or args.serial) else 0
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        rays = paths = passes = 0
        kms = {"generate": 0.0, "extend": 0.0, "shade": 0.0, "connect": 0.0, "connect_mis": 0.0}
        for _ in range(steps):
            rs = parallel.render_frame(lambda p, im: dev.render(p, im, stream=stream), params, img, dist)
            rays += rs.rays_closest + rs.rays_any
            paths += rs.paths
            passes += rs.passes
            kms["generate"] += rs.generate_ms
            kms["extend"] += rs.extend_ms
            kms["shade"] += rs.shade_ms
            kms["connect"] += rs.connect_ms
            kms["connect_mis"] += rs.connect_mis_ms
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_begin
        if dist is not None:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tot = torch.tensor([float(rays), float(paths)], dtype=torch.float64, device="cuda")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            return float(tmax[0]), float(tot[0]), float(tot[1]), kms, passes
        return elapsed, float(rays), float(paths), kms, passes

    # the headline: the whole frame at N = 1, its strong (sample) split at N > 1
    weak = parallel.rank_params(WIDTH, HEIGHT, spp_x, spp_y, rank, world, max_depth=MAX_DEPTH, seed=SEED, weak=(world == 1))
    for _ in range(args.warmup):
        timed(weak, 1)
    # one untimed pass with the traversal counters on (rank 0 only, no collective): the node / leaf /
    # reference / primitive-test visits the kernels really make, per ray class, for the roofline
    own = None
    if rank == 0 and with_cpu:
        sp = render_params(WIDTH, HEIGHT, spp_x, spp_y, max_depth=MAX_DEPTH, seed=SEED, sample_begin=0,
                           sample_end=min(spp, 256), flags=abi.MRT_RENDER_TRACE_STATS | abi.MRT_RENDER_SERIAL)
        sp.paths_per_pass = args.paths_per_pass
        scratch = torch.zeros_like(img)
        own = dev.render(sp, scratch, stream=stream)
        del scratch
    elapsed, total_rays, total_paths, kms, n_passes = timed(weak, args.steps)
    # The timed frames trace the three ray lists of a bounce side by side, so their per-class times
    # overlap.  One more, untimed, frame with one kernel at a time gives every kernel's own duration
    # (kernel_ms_per_step and the roofline come from it; rank-local, no collective).
    if not args.serial and rank == 0:
        import copy
        solo = copy.copy(weak)
        solo.paths_per_pass = args.paths_per_pass
        solo.flags = abi.MRT_RENDER_SERIAL
        tmp = torch.zeros_like(img)
        rs = dev.render(solo, tmp, stream=stream)
        torch.cuda.synchronize()
        del tmp
        kms = {"generate": rs.generate_ms * args.steps, "extend": rs.extend_ms * args.steps, "shade": rs.shade_ms * args.steps,
               "connect": rs.connect_ms * args.steps, "connect_mis": rs.connect_mis_ms * args.steps}
    mean_radiance = float(img.mean().item()) / spp if rank == 0 else 0.0

    side = None  # N > 1: the weak split beside the headline
    if world > 1:
        wparams = parallel.rank_params(WIDTH, HEIGHT, spp_x, spp_y, rank, world, max_depth=MAX_DEPTH, seed=SEED, weak=True)
        timed(wparams, 1)
        w_elapsed, w_rays, w_paths, _, _ = timed(wparams, args.steps)
        side = {"value": w_rays / w_elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": w_elapsed / args.steps * 1e3,
                "spp_per_gpu": spp, "scaling": "weak"}

    # N = 1: what one GPU of an N-way strong split has to do, timed like the headline
    proxy = None
    if world == 1 and not args.no_strong_proxy and spp % 8 == 0:
        proxy = {"note": "samples [0, spp/N) of the same %d x %d grid on this one GPU; efficiency = t(spp) / (N * t(spp/N)): the "
                         "strong-split scaling before the reduce (one RCCL reduce of a 24.9 MB image per frame)" % (spp_x, spp_y),
                 "ms_full_frame": elapsed / args.steps * 1e3}
        for n_way in (2, 4, 8):
            pp = render_params(WIDTH, HEIGHT, spp_x, spp_y, max_depth=MAX_DEPTH, seed=SEED, sample_begin=0, sample_end=spp // n_way)
            timed(pp, 1)
            p_elapsed, _, _, _, _ = timed(pp, args.steps)
            proxy["n%d" % n_way] = {"spp": spp // n_way, "ms": p_elapsed / args.steps * 1e3,
                                    "efficiency": (elapsed / args.steps) / (n_way * p_elapsed / args.steps)}
        proxy["efficiency"] = proxy["n8"]["efficiency"]
        # the other way to divide the frame: an interleaved share of its 8x8 tiles, all samples each (the sums of
        # the shares are the frame, reduced the same way)
        tiles = {}
        for n_way in (2, 4, 8):
            if not parallel.tile_share_possible(WIDTH, HEIGHT, n_way):
                continue
            pp = parallel.rank_params(WIDTH, HEIGHT, spp_x, spp_y, 0, n_way, max_depth=MAX_DEPTH, seed=SEED, split="tile")
            timed(pp, 1)
            p_elapsed, _, _, _, _ = timed(pp, args.steps)
            tiles["n%d" % n_way] = {"ms": p_elapsed / args.steps * 1e3, "efficiency": (elapsed / args.steps) / (n_way * p_elapsed / args.steps)}
        if tiles:
            tiles["note"] = "share 0 of the tiles (tx + ty) mod N == 0, all %d samples of its pixels" % spp
            proxy["tile_split"] = tiles

    if rank != 0:
        dev.close()
        return None

    per_step = {k: v / args.steps for k, v in kms.items()}
    out = {
        "value": total_rays / elapsed / 1e6,
        "ms_per_step": elapsed / args.steps * 1e3,
        "data": wl["data"],
        "config": {
            "workload": wl["label"] % spp + ("" if (spp_x, spp_y) == wl["spp"] else " [reduced spp for profiling]"),
            "width": WIDTH, "height": HEIGHT, "spp_per_gpu": spp / world, "max_depth": MAX_DEPTH,
            "triangles": sc.n_objects - 1, "kd_nodes": n_nodes, "kd_refs": n_refs, "kd_depth": depth,
            "kd_build_s": round(build_s, 2),
            "parallelism": "sample-split x%d + RCCL reduce" % world if world > 1 else "single GPU",
        },
        "paths_per_s": total_paths / elapsed,
        "rays_per_path": total_rays / max(total_paths, 1.0),
        "mean_radiance": mean_radiance,
        "kernel_ms_per_step": {k: per_step[k] for k in ("generate", "extend", "shade", "connect")},
        "kernel_ms_note": "device time of each kernel class running alone (one untimed frame with MRT_RENDER_SERIAL); the timed frames "
                          "overlap the shadow, MIS and next-bounce traversal launches, so their sum exceeds ms_per_step" if not args.serial
                          else "all frames rendered with MRT_RENDER_SERIAL (one kernel at a time)",
    }
    if name == "million_bunny" and spp == 1024:
        # measured, not extrapolated: one step IS 1024 spp at 1080p (on one GPU, or divided among the N ranks)
        out["seconds_to_1024spp_1080p"] = elapsed / args.steps
    if side:
        out["weak_split"] = side
    if proxy:
        out["strong_proxy"] = proxy

    cpu = roof = None
    if with_cpu:
        from tests import orc
        # the GPU box gives one GPU's share of the host: 16 cores (torchrun sets OMP_NUM_THREADS=1
        # for its workers, so the thread count is passed to the oracle explicitly)
        cores = min(os.cpu_count() or 1, 16)
        cpu_s = max(1, min((args.cpu_spp or wl["cpu_spp"]) if world == 1 else 1, spp))
        cp = render_params(WIDTH, HEIGHT, spp_x, spp_y, max_depth=MAX_DEPTH, seed=SEED, sample_begin=0, sample_end=cpu_s)
        tc = time.perf_counter()
        cimg, crs, cts, _ = orc.oracle_render(flat.desc, cp, 0, threads=cores)
        cpu_t = time.perf_counter() - tc
        crays = int(crs.rays_closest + crs.rays_any)
        if world == 1:
            cpu = {"value": crays / cpu_t / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": "same scene and camera, full 1920x1080 frame, sample indices [0,%d) of the %d (%.1f s)" % (cpu_s, spp, cpu_t),
                   "seconds": cpu_t, "rays": crays, "mean_radiance": float(cimg.mean()) / cpu_s}
        cls_rays = crs.class_rays

        def b_ray(ts, n):
            return own_bytes(ts, float(n)) / n if n else 0.0
        ref_bytes_per_ray = {"extend": b_ray(cts[0], int(cls_rays[0])), "shadow": b_ray(cts[1], int(cls_rays[1])),
                             "mis": b_ray(cts[2], int(cls_rays[2]))}
        # bytes per ray of each class from the kernels' own visits (fewer than the reference's for
        # shadow rays and, in a scene without emitters, MIS rays: a plane hit answers them without
        # entering the tree and they stop at the first hit); the roofline uses these
        n_ext, n_sh, n_mis = float(own.rays_closest - own.rays_mis), float(own.rays_any), float(own.rays_mis)
        bpr = {"extend": own_bytes(own.trace[0], n_ext) / max(n_ext, 1.0), "shadow": own_bytes(own.trace[1], n_sh) / max(n_sh, 1.0),
               "mis": own_bytes(own.trace[2], n_mis) / max(n_mis, 1.0)}
        # rays of each class in one timed step, from the counted pass's class shares
        step_rays = total_rays / args.steps / world
        share = {"extend": n_ext, "shadow": n_sh, "mis": n_mis}
        tot = max(n_ext + n_sh + n_mis, 1.0)
        mis_kernel = wl["mis_kernel"]
        # extend_kernel: with its first-bounce instance extend_camera_kernel (the same traversal, which
        # also makes the camera rays); the two are one class in the statistics and in profiles/
        cands = [("extend_kernel", "extend", per_step["extend"]),
                 ("connect_shadow_kernel", "shadow", per_step["connect"] - per_step["connect_mis"]),
                 (mis_kernel, "mis", per_step["connect_mis"])]
        kname, cls, ms = max(cands, key=lambda c: c[2])
        byts = bpr[cls] * step_rays * share[cls] / tot
        achieved = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        launches = (MAX_DEPTH + 1) * n_passes // args.steps
        traffic, hit_rate, miss_lines = profile_traffic(name, kname) if (spp_x, spp_y) == wl["spp"] and world == 1 else (None, None, None)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": kname if kname != "extend_kernel" else "extend_kernel (+ extend_camera_kernel, its first-bounce instance)",
                "launches_per_step": launches,
                "avg_launch_ms": ms / launches if launches else None,
                "algorithmic_bytes_per_launch": byts / launches if launches else None,
                "bytes_per_ray": bpr, "reference_bytes_per_ray": ref_bytes_per_ray,
                "timing": "kernel alone on the device (MRT_RENDER_SERIAL frame)",
                "note": "achieved = ALGORITHMIC bytes (52 + 8*inner + 8*leaves + 4*refs + 36*prim_tests per ray, visits counted by the "
                        "kernels themselves on an untimed pass) of the kernel's launches of one step / their summed HIP-event time.  "
                        "traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch from the committed counter passes (profiles/hbm_traffic.json; "
                        "the doubling is calibrated for this access pattern, profiles/r03_gather_calibration.json); "
                        "hbm_frac_from_counters = traffic / launch time / peak.  prim_tests counts what the reference tests; the kernels "
                        "pass over the triangles a ray is known to miss (mailbox) without fetching them, so fewer bytes than these move.  "
                        "The traversal gathers 16 bytes at a time and every gather that misses the L2 moves a whole 128-byte line: "
                        "line_fetch prices the lines against what the fabric delivers to a loop that does nothing else."}
        if traffic and launches:
            roof["hbm_frac_from_counters"] = traffic / (ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["limiter"] = ("cache look-ups and scalar instructions per trip of the traversal loop, not bytes or lines: with the loop unchanged, 22 % "
                               "fewer fabric line fetches (leaf lists next to their tree records, three primitive records a line) bought 3.5 %; one "
                               "look-up per four leaf references instead of one per reference bought 7 %; 16 more scalar instructions per trip cost "
                               "2 %, 32 more vector instructions 1 %" + (" (TCC hit rate %.2f)" % hit_rate if hit_rate else ""))
        if miss_lines and launches:
            rate = miss_lines / (ms / launches * 1e-3) / 1e9
            roof["line_fetch"] = {"lines_per_launch": miss_lines, "G_lines_per_s": rate, "ceiling_G_lines_per_s": line_ceiling,
                                  "frac": rate / line_ceiling if line_ceiling else None,
                                  "note": "lines = the L2's read requests to the fabric (TCC_EA0_RDREQ, committed counter pass); ceiling = "
                                          "raytracer_amd/gather_bench measured in this run: random 32-byte gathers over a 4 GB table at 6 "
                                          "waves per SIMD, every one a miss (the same rate whether 32, 64 or 128 bytes of the line are used)"}
    out["roofline"] = roof
    out["cpu_baseline"] = cpu
    dev.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="million_bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and roofline legs")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] measurement")
    ap.add_argument("--no-strong-proxy", action="store_true", help="skip the N = 1 strong-split proxy")
    ap.add_argument("--serial", action="store_true", help="render every frame with one kernel at a time (profiling runs: per-kernel "
                                                          "durations of a kernel trace are then each kernel's own)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU sample (0 = the workload's default)")
    ap.add_argument("--spp-y", type=int, default=0, help="rows of the sample grid (profiling runs use fewer)")
    ap.add_argument("--paths-per-pass", type=int, default=0, help="paths in flight per pass (0 = library default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # started by hand with --gpus N: run the ranks as child processes (never exec after GPU init)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    # MRT_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo, to rehearse the N > 1 control flow on
    # a single-GPU box (RCCL refuses two ranks on one device); never set for measurements
    rehearsal = os.environ.get("MRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    main_res = measure(args.workload, args, rank, world, dist, local_rank, with_cpu=not args.no_cpu)
    secondary = None
    if world == 1 and not args.no_secondary and not args.spp_y:
        other = "bunny_pair" if args.workload == "million_bunny" else "million_bunny"
        sargs = argparse.Namespace(**vars(args))
        sargs.cpu_spp = 0
        sargs.no_strong_proxy = True
        secondary = measure(other, sargs, rank, world, dist, local_rank, with_cpu=not args.no_cpu)
    if rank == 0:
        result = {
            "metric": "Mrays/s (path-traced frame: kd-tree traversal + ray-triangle + BSDF/MIS loop)",
            "value": main_res.pop("value"),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_res.pop("ms_per_step"),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
        }
        result.update(main_res)
        if secondary:
            result["secondary"] = secondary
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
