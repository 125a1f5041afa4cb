#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-traced frame BASELINE.json names.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): stanford-bunny.obj, one diffuse and one glossy instance, Plane
floor, 1k HDR environment, 1920x1080, 256 spp (16x16 stratified), max depth 5.  One step = one
full frame of the hot path (generate, kd-tree traversal, shading, NEE + MIS connections,
per-pixel accumulation) with the scene and the image resident in HBM.  A ray is one
intersectKdTree-equivalent query (closest-hit or any-hit) actually traced.

N > 1 (launched by torch.distributed.run, one rank per GPU): sample split.  Rank r renders sample
indices [r*256, (r+1)*256) of a 16 x 16N stratified grid into its own radiance-sum image and the
images are summed onto rank 0 with one RCCL reduce; per-GPU work is constant (weak scaling).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the dominant kernel (by summed device time) against the 8 TB/s HBM peak
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
SPP_X, SPP_Y = 16, 16
MAX_DEPTH = 5
SEED = 1
HBM_PEAK_GBS = 8000.0


def b_ray(ts, n):
    """Algorithmic bytes of one ray: 36 B ray in + 16 B hit out, 8 B per inner node and per leaf
    header visited, 4 B per leaf reference, 36 B (p1, e1, e2) per primitive test."""
    if n == 0:
        return 0.0
    return 52.0 + (8.0 * ts.inner + 8.0 * ts.leaves + 4.0 * ts.refs + 36.0 * ts.prim_tests) / n


def main():
    global SPP_Y
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-spp", type=int, default=32, help="samples per pixel of the CPU sample")
    ap.add_argument("--spp-y", type=int, default=16, help="rows of the 16 x N sample grid (profiling runs use fewer)")
    ap.add_argument("--paths-per-pass", type=int, default=0, help="paths in flight per pass (0 = library default)")
    args = ap.parse_args()

    SPP_Y = args.spp_y
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # started by hand with --gpus N: run the ranks as child processes (never exec after GPU init)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__),
               "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)] + \
              (["--no-cpu"] if args.no_cpu else [])
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch

    from raytracer_amd import abi, scenes
    from raytracer_amd.host import DeviceScene, render_params

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

    t0 = time.time()
    sc = scenes.bunny_pair(WIDTH, HEIGHT)
    tree = sc.initKdTree()
    build_s = time.time() - t0
    flat = tree.flatten(1)
    dev = DeviceScene(flat, device=local_rank)
    n_nodes, n_refs, depth = tree.counts()

    from raytracer_amd import parallel

    spp = SPP_X * SPP_Y
    params = parallel.rank_params(WIDTH, HEIGHT, SPP_X, SPP_Y, rank, world, max_depth=MAX_DEPTH, seed=SEED, weak=True)
    params.paths_per_pass = args.paths_per_pass
    img = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        return parallel.render_frame(lambda p, im: dev.render(p, im, stream=stream), params, img, dist)

    for _ in range(args.warmup):
        step()
    # one untimed frame with the traversal counters on: the node / leaf / reference / primitive-test
    # visits the kernels really make, per ray class (rank 0 only, for the roofline)
    own = None
    if rank == 0 and not args.no_cpu:
        params.flags = abi.MRT_RENDER_TRACE_STATS
        own = dev.render(params, img, stream=stream)  # rank-local: no collective outside the steps all ranks run
        params.flags = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    rays_closest = rays_any = paths = 0
    kms = {"generate": 0.0, "extend": 0.0, "shade": 0.0, "connect": 0.0}
    mis_ms = 0.0  # the part of "connect" spent in the MIS rays' kernel
    for _ in range(args.steps):
        rs = step()
        rays_closest += rs.rays_closest
        rays_any += rs.rays_any
        paths += rs.paths
        kms["generate"] += rs.generate_ms
        kms["extend"] += rs.extend_ms
        kms["shade"] += rs.shade_ms
        kms["connect"] += rs.connect_ms
        mis_ms += rs.connect_mis_ms
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin

    stats = torch.tensor([elapsed, float(rays_closest + rays_any), float(paths)], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rays, total_paths = float(tot[0]), float(tot[1])
    else:
        total_rays, total_paths = float(stats[1]), float(stats[2])

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    mean_radiance = float(img.mean().item()) / (spp * world)
    result = {
        "metric": "Mrays/s (path-traced frame: kd-tree traversal + ray-triangle + BSDF/MIS loop)",
        "value": total_rays / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "stanford-bunny.obj (69 451 triangles, x2 instances, vertices x20) + brown_photostudio_06_1k.hdr committed under assets/; camera, materials and floor synthetic",
        "config": {
            "workload": "stanford-bunny diffuse+glossy, 1920x1080, %d spp per GPU, max depth 5, HDR env (BASELINE configs[1]%s)" % (
                spp, "" if spp == 256 else ", reduced spp for profiling"),
            "width": WIDTH, "height": HEIGHT, "spp_per_gpu": spp, "max_depth": MAX_DEPTH,
            "triangles": sc.n_objects - 1, "kd_nodes": n_nodes, "kd_refs": n_refs, "kd_depth": depth,
            "kd_build_s": round(build_s, 2), "parallelism": "sample-split x%d + RCCL reduce" % world if world > 1 else "single GPU",
        },
        "paths_per_s": total_paths / elapsed,
        "rays_per_path": total_rays / max(total_paths, 1.0),
        "seconds_to_1024spp_1080p": 1024.0 / (spp * world) * (elapsed / args.steps),
        "mean_radiance": mean_radiance,
        "kernel_ms_per_step": {k: v / args.steps for k, v in kms.items()},
    }

    # ---- CPU baseline + traversal counters (rank 0, bounded sample of the same frame)
    cpu = None
    roof = None
    if not args.no_cpu:
        from tests import orc
        # the GPU box gives one GPU's share of the host: 16 cores (torchrun sets OMP_NUM_THREADS=1
        # for its workers, so the thread count is passed to the oracle explicitly)
        cores = min(os.cpu_count() or 1, 16)
        # N > 1: no cpu_baseline line, only a 2-sample pass for the traversal counters of the roofline
        cpu_s = max(1, min(args.cpu_spp if world == 1 else 2, spp))
        cp = render_params(WIDTH, HEIGHT, SPP_X, SPP_Y * world, max_depth=MAX_DEPTH, seed=SEED, sample_begin=0,
                           sample_end=cpu_s)
        tc = time.perf_counter()
        cimg, crs, cts, _ = orc.oracle_render(flat.desc, cp, 0, threads=cores)
        cpu_t = time.perf_counter() - tc
        crays = int(crs.rays_closest + crs.rays_any)
        cpu = {"value": crays / cpu_t / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": "same scene and camera, full 1920x1080 frame, sample indices [0,%d) of the 256 (%.1f s)" % (cpu_s, cpu_t),
               "seconds": cpu_t, "rays": crays}
        # relMSE of the GPU frame against the CPU sample is not meaningful at different spp; the
        # parity tests compare equal-spp renders.  Here only the mean radiance is cross-checked.
        cpu["mean_radiance"] = float(cimg.mean()) / cpu_s
        if world > 1:
            cpu = None
        # roofline of the dominant traversal kernel: algorithmic bytes / measured device time
        cls_rays = crs.class_rays
        # reference-algorithm bytes (what the CPU traversal touches for the same rays) ...
        ref_bytes_per_ray = {"extend": b_ray(cts[0], int(cls_rays[0])), "shadow": b_ray(cts[1], int(cls_rays[1])),
                             "mis": b_ray(cts[2], int(cls_rays[2]))}
        # ... and the bytes of the visits the kernels make themselves (fewer for shadow rays and, in a
        # scene without emitters, MIS rays: a plane hit answers them without entering the tree and
        # they stop at the first hit).  The roofline uses the kernels' own visits.
        n_ext = float(own.rays_closest - own.rays_mis)
        n_sh = float(own.rays_any)
        n_mis = float(own.rays_mis)

        def own_bytes(ts, n):
            return 52.0 * n + 8.0 * ts.inner + 8.0 * ts.leaves + 4.0 * ts.refs + 36.0 * ts.prim_tests
        ext_bytes = own_bytes(own.trace[0], n_ext)
        per_step = {k: v / args.steps for k, v in kms.items()}
        n_bounces = MAX_DEPTH + 1
        # the three traversal kernels, each with its summed device time per step; the roofline is
        # that of the one with the most (the bench scene has no emissive object, so its MIS rays
        # go through the occlusion variant)
        cands = [("extend_kernel", ext_bytes, per_step["extend"]),
                 ("connect_shadow_kernel", own_bytes(own.trace[1], n_sh), per_step["connect"] - mis_ms / args.steps),
                 ("connect_mis_occlusion_kernel", own_bytes(own.trace[2], n_mis), mis_ms / args.steps)]
        name, byts, ms = max(cands, key=lambda c: c[2])
        launches = n_bounces
        achieved = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": name, "launches_per_step": launches,
                "avg_launch_ms": ms / launches, "algorithmic_bytes_per_launch": byts / launches,
                "bytes_per_ray": {"extend": ext_bytes / max(n_ext, 1.0), "shadow": own_bytes(own.trace[1], n_sh) / max(n_sh, 1.0),
                                  "mis": own_bytes(own.trace[2], n_mis) / max(n_mis, 1.0)},
                "reference_bytes_per_ray": ref_bytes_per_ray,
                "note": "bytes per ray = 52 + 8*inner + 8*leaves + 4*refs + 36*prim_tests with the visits counted by the "
                        "traversal kernels themselves on an untimed frame (reference_bytes_per_ray: the same formula on the "
                        "CPU oracle's visits for the reference's traversal); achieved = those bytes of the kernel's launches "
                        "of one step / their summed HIP-event time; traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch "
                        "from profiles/hbm_traffic.json"}
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof) and spp == 256 and world == 1:
            try:
                with open(prof) as f:
                    per_step_bytes = json.load(f)["bytes_per_step"].get(name)
                if per_step_bytes:
                    roof["traffic"] = per_step_bytes / launches
            except Exception:
                pass
    result["roofline"] = roof
    result["cpu_baseline"] = cpu
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
