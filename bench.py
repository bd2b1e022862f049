#!/usr/bin/env python3
"""Headline benchmark: RGB-D frames/s fused (640x480, 5 mm voxel TSDF) on N MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
For N > 1 the driver launches it with torch.distributed.run (one rank per GPU, RCCL).

Workload (BASELINE.json configs[1]): synthetic 640x480 RGB-D stream (30 Hz camera on a 600-pose loop), 5 mm
voxels, sdf_trunc 0.04 m, depth_trunc 4 m, Open3D ScalableTSDFVolume semantics.  A *step* fuses one batch of
``--frames-per-step`` consecutive posed frames that are already resident in HBM (default 64: the width of the sweep's frame
mask and what the drop-in front drains per call; rounds 1-5 and the first lines of round 6 used 32 - reported beside the headline
as ``batch32``); value = frames/s of the whole job.

``--window sliding`` (default, the headline): step k fuses frames Bk .. Bk+B-1 of the stream into ONE volume that
is empty when the timed region starts (the warm-up steps run on the same frames and the volume is reset after
them) — first-touch allocation of every unit and the frame-to-frame overlap of a moving camera are inside the
timing.  ``--window replay``: every step re-fuses the same B frames (the round-1 figure; reported as the secondary
key ``replay_mode`` at N = 1).

N > 1 ("strong" scaling: total work is fixed): every rank sees every frame; --sharding owner (default): a unit is
fused and stored by GPU hash(unit index) % N (SURVEY §8e "zero reduce" form: bit-identical to one GPU, no
collective while fusing); --sharding tile: north-star image tiles + RCCL merge of the shared units (timed).

Objects on the JSON line (N = 1):
  roofline      the dominant kernel of the timed region, the multi-frame sweep k_tsdf_sweep_column, against the bound
                SURVEY 8d names for the path: HBM.  achieved = BATCH-level algorithmic bytes (oracle: distinct units
                touched by the batch x 81 920 B read + distinct voxels updated x 20 B written - the sweep reads and
                writes a unit's slab once per launch whatever the number of frames) / the mean launch duration measured
                live with HIP events on the kernel's stream; peak = 8 TB/s; traffic = HBM bytes per launch from the
                rocprofv3 --pmc passes recorded under profiles/ (parsed at run time, only used when taken on THIS build
                and command).  ``roofline.flops``: SURVEY 8d's 40 flop per voxel visit x the oracle's visit count / the
                same duration against the 157.3 TFLOP/s vector-f32 peak.  ``roofline.valu_utilisation`` is NOT a
                roofline: the kernel's own SQ_INSTS_VALU / duration against the wave64 issue rate (both the 4-cycle
                rate tools/valu_rates.hip measures for FMA-class instructions and the guide's 2-cycle figure).
  online_mode   one hv_tsdf_integrate per frame (pySLAM's live flow) on the same sliding stream: frames/s and that
                kernel's roofline — the HBM-bound one: per-frame algorithmic bytes (U_touched*4096*20 B +
                N_updated*20 B, oracle counts of exactly the frames timed) / mean launch duration vs 8 TB/s.
  extraction    extract_triangle_mesh + extract_point_cloud of the volume the timed region built (BASELINE configs[2]
                shape of work), wall ms and kernel ms, B_mc roofline.
  host_mode     the path the drop-in really drives (tools/bench_host.py): the same stream handed over as pageable per-keyframe
                numpy arrays - PCIe INSIDE the timed region (never `value`) - through integrate_frames (page-locked staging slots
                + copy stream, overlapped with the previous batch's sweep), next to the H2D-bound rate measured in the same
                process, one integrate() per host frame, and the whole front (add_keyframe -> worker process -> pop_output) once.
  configs       BASELINE configs[2] / [4] shapes on one GPU: Replica-shaped 1200x680 @ 4 mm with marching cubes every 10 frames,
                ScanNet-shaped 1296x968 @ 2 mm; frames/s.  semantic_scannet_2mm: the semantic flow at that shape.
  voxel_grid    the cpp/volumetric VOXEL_GRID mode on the same frames: per-frame and batched frames/s, B_vox
                roofline, and the COMPILED REFERENCE (oracle/_ref, kind "reference") timed beside it.
  semantic      pySLAM's per-keyframe semantic flow (shadow filter, assign_object_ids_to_instance_ids, remap, integrate) for
                the voting and the probabilistic payload, keyframes/s, next to the compiled reference (tools/bench_semantic.py).
  cpu_baseline  oracle/tsdf_oracle.c (Open3D-semantics restatement, kind "port") timed on the host cores over
                the same sliding stream (N = 1 only, bounded by --cpu-budget-s).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOXEL = 0.005
SDF_TRUNC = 0.04
DEPTH_TRUNC = 4.0
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINSTR = 1024 * 2.4 / 4.0  # 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction = 614.4 G wave-instr/s
BYTES_PER_VOXEL = 20     # {f32 tsdf, u32 weight, 3 x u32 colour sums}
UNIT_BYTES = 4096 * BYTES_PER_VOXEL
VALU_PEAK_GINSTR_2CYC = 1024 * 2.4 / 2.0  # MI355X_MICROARCH.md lists wave64 v_fma_f32 at 2 cycles: 1228.8 G wave-instr/s
VECTOR_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector (non-MFMA) fp32
FLOP_PER_VISIT = 40  # "TSDF ~ 40 flop / voxel visited"
PROFILE_ROUND = "r06"
PMC_SUMMARY = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_summary.json")


def load_frames(config, n_frames, start=0, rank=0, barrier=None):
    """Synthetic frames, cached under /tmp (generation is host-side numpy ray casting, spread over the host cores).
    With several ranks on the node rank 0 renders, the others wait and read the cache."""
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD(config)
    cache = f"/tmp/pyslam_amd_bench_{config}_{start}_{n_frames}.npz"
    if not os.path.exists(cache) and rank == 0:
        depth, rgb, T = s.batch(start, n_frames)
        try:
            tmp = f"{cache}.{os.getpid()}.tmp.npz"
            np.savez(tmp, depth=depth, rgb=rgb, T=T)
            os.replace(tmp, cache)
        except OSError:
            if barrier is None:
                return s, depth, rgb, T
            raise
    if barrier is not None:
        barrier()
    z = np.load(cache)
    return s, z["depth"], z["rgb"], z["T"]


def d2d_copy_gbs(n_bytes=1 << 30, reps=6):
    """Device-to-device copy rate of THIS GPU, measured in this run (report against the vendor peak AND a measured copy):
    a 1 GiB copy kernel, bytes read + bytes written per second."""
    import torch

    src = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    src.fill_(3)
    dst.copy_(src)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n_bytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return best


def pmc_summary(build_digest):
    """profiles/<round>/pmc_summary.json (written by tools/pmc_summary.py --json from the rocprofv3 --pmc passes of
    tools/profile_round.sh).  Only trusted when it was recorded on the library build that is running now."""
    try:
        with open(PMC_SUMMARY) as f:
            z = json.load(f)
    except (OSError, ValueError):
        return None
    z["stale"] = z.get("build_digest") != build_digest
    return z


def current_build_digest():
    try:
        with open(os.path.join(ROOT, "pyslam_amd", "lib", ".build_digest")) as f:
            return f.read().strip()
    except OSError:
        return None


def pmc_kernel(pmc, name, command_key):
    """Mean per-launch counters of kernel `name` from the recorded passes, or None when absent / stale / taken on a
    different command."""
    if not pmc or pmc.get("stale") or pmc.get("command_key") != command_key:
        return None
    base, _, suffix = name.partition("@")  # "<kernel>@full": the second entry tools/pmc_summary.py --also writes for a kernel
    for k, v in pmc.get("kernels", {}).items():
        if base in k and k.endswith("@" + suffix) == bool(suffix) and ("@" in k) == bool(suffix):
            return v
    return None


def hbm_traffic(pk):
    """MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the
    bytes read -> traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024."""
    if not pk or "FETCH_SIZE" not in pk or "WRITE_SIZE" not in pk:
        return None
    return int((2.0 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024)


def live_pmc(args, B, kernel_substr="k_tsdf_sweep"):
    """HBM traffic of the dominant kernel measured IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE: separate
    passes with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) of this very command without its secondary
    legs; mean over the launches of the timed steps.  -> {"FETCH_SIZE": KB, "WRITE_SIZE": KB, "launches": n} or None (no rocprofv3
    on the box, BENCH_LIVE_PMC=0, a pass failed or timed out)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("BENCH_LIVE_PMC", "1") == "0" or shutil.which("rocprofv3") is None:
        return None
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", str(args.warmup), "--clock-ramp-steps",
             str(args.clock_ramp_steps), "--frames-per-step", str(B), "--window", args.window, "--config", args.config, "--no-cpu-baseline",
             "--no-secondary"]
    cold = (args.warmup + args.steps) if args.clock_ramp_steps > 0 else 0  # the cold-clock pass comes first
    lo = cold + args.clock_ramp_steps + args.warmup
    hi = lo + args.steps
    env = dict(os.environ, BENCH_LIVE_PMC="0", TMPDIR="/tmp")
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + child,
                           cwd="/tmp", env=env, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            vals = []
            for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r.get("Dispatch_Id", 0)))
                vals += [float(r["Counter_Value"]) for r in rows if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
            vals = vals[lo:hi]
            if not vals:
                return None
            out[counter] = sum(vals) / len(vals)
            out["launches"] = len(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def cpu_baseline_and_counts(s, depth, rgb, T, B, max_steps, budget_s, threads, window, extras=True):
    """Time the CPU restatement over the same stream step by step (bounded by budget_s) and collect the oracle
    counts the roofline figures need: per frame (touched units, updated voxels) and per batch (distinct units
    touched, distinct voxels updated)."""
    import oracle

    oracle.use_native_port()  # -O3 -march=native build for THIS host (falls back to the travelling .so)
    K = np.array(s.intrinsics, dtype=np.float64)
    n_res = len(depth)
    if threads <= 0:
        # ONE stated policy (a probe picked 8 threads on one box and 32 on another): min(os.cpu_count(), 32)
        # OpenMP threads - the restatement parallelises over the ~4 k units a frame touches behind a serial claim phase and stops scaling
        # there (measured on the 256-thread host of an MI355X box: 1 thread 9.5, 32 threads 67, 256 threads 6.6 frames/s) - with the same
        # code on ONE core (`single_core`) and on EVERY hardware thread (`all_hw_threads`) beside it, so that no choice is hidden
        threads = min(os.cpu_count() or 1, 32)
    single = every = None
    if extras and budget_s > 0 and (os.cpu_count() or 1) > threads:
        allc = oracle.PortTsdf(VOXEL, SDF_TRUNC, threads=os.cpu_count())
        allc.integrate(depth[0], rgb[0], K, T[0], 1.0, DEPTH_TRUNC)
        t0 = time.perf_counter()
        na = 0
        while na < 16 and time.perf_counter() - t0 < 0.1 * budget_s:
            i = (1 + na) % n_res
            allc.integrate(depth[i], rgb[i], K, T[i], 1.0, DEPTH_TRUNC)
            na += 1
        every = {"value": round(na / (time.perf_counter() - t0), 3), "unit": "frames/s", "cores": os.cpu_count(), "frames": na}
        del allc
    if extras and budget_s > 0:
        one = oracle.PortTsdf(VOXEL, SDF_TRUNC, threads=1)
        one.integrate(depth[0], rgb[0], K, T[0], 1.0, DEPTH_TRUNC)
        t0 = time.perf_counter()
        n1 = 0
        while n1 < 16 and time.perf_counter() - t0 < 0.2 * budget_s:
            i = (1 + n1) % n_res
            one.integrate(depth[i], rgb[i], K, T[i], 1.0, DEPTH_TRUNC)
            n1 += 1
        single = {"value": round(n1 / (time.perf_counter() - t0), 3), "unit": "frames/s", "cores": 1, "frames": n1}
        del one
    vol = oracle.PortTsdf(VOXEL, SDF_TRUNC, threads=threads)
    steps = []
    t_begin = time.perf_counter()
    t_frames = 0.0
    for k in range(max_steps):
        vol.batch_begin()
        touched, updated = [], []
        for j in range(B):
            i = (k * B + j) % n_res if window == "sliding" else j
            t0 = time.perf_counter()
            vol.integrate(depth[i], rgb[i], K, T[i], 1.0, DEPTH_TRUNC)
            t_frames += time.perf_counter() - t0
            touched.append(vol.num_touched())
            updated.append(vol.last_updated())
        u, v = vol.batch_end()
        steps.append({"touched": touched, "updated": updated, "union_units": u, "union_voxels": v})
        if time.perf_counter() - t_begin > budget_s:
            break
    n = len(steps) * B
    return {"threads": threads, "fps": n / t_frames, "frames": n, "seconds": t_frames, "steps": steps,
            "units": vol.num_units(), "single_core": single, "all_hw_threads": every}


def voxel_grid_leg(s, depth_h, rgb_h, T_h, depth_d, rgb_d, frames, steps, cpu_frames):
    """Secondary: the cpp/volumetric VOXEL_GRID mode (VoxelBlockGrid.integrate_rgbd = fused depth2pointcloud + world
    transform + integrate_raw semantics) on the first `frames` frames, with the compiled reference beside it."""
    import oracle
    from oracle import host_prep as hp
    from pyslam_amd.volumetric import VoxelBlockGrid

    g = VoxelBlockGrid(VOXEL, 8, max_blocks=1 << 18, max_points=1 << 20)

    def step():
        for f in range(frames):
            g.integrate_rgbd(depth_d[f], rgb_d[f], *s.intrinsics, T_h[f], max_depth=DEPTH_TRUNC)

    step()
    g.synchronize()
    dt = None
    for _ in range(3):  # a few ms per repetition: the best of three keeps one page-in or clock ramp from deciding the figure
        g.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        g.synchronize()
        t = time.perf_counter() - t0
        ms, launches, _ = g.profile_read()
        g.profile_enable(False)
        if dt is None or t < dt:
            dt, k_ms, k_launches = t, ms, launches
    gb = VoxelBlockGrid(VOXEL, 8, max_blocks=1 << 18, max_points=frames * s.width * s.height)
    gb.integrate_rgbd_batch(depth_d[:frames], rgb_d[:frames], *s.intrinsics, T_h[:frames], max_depth=DEPTH_TRUNC)
    gb.synchronize()
    dt_b = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            gb.integrate_rgbd_batch(depth_d[:frames], rgb_d[:frames], *s.intrinsics, T_h[:frames], max_depth=DEPTH_TRUNC)
        gb.synchronize()
        t = time.perf_counter() - t0
        dt_b = t if dt_b is None else min(dt_b, t)
    # oracle counts for B_vox: 2 x 28 B per distinct voxel a frame touches (steady state: no new blocks in
    # the timed replay) + CPU baseline = the compiled reference on the same float32 world points
    pts = [hp.frame_to_world_f32(depth_h[i], rgb_h[i], *s.intrinsics, T_h[i], DEPTH_TRUNC)[:2] for i in range(cpu_frames)]
    v_touched = []
    for p, _c in pts:
        vk = oracle.keys(p, VOXEL, 8, which="port")[0]
        v_touched.append(len(np.unique(vk, axis=0)))
    b_vox = 2 * 28 * float(np.mean(v_touched))
    b_in = s.width * s.height * 7
    out = {"metric": "RGB-D frames/sec fused (640x480, 5 mm, VOXEL_GRID cpp/volumetric semantics)",
           "value": round(steps * frames / dt, 1), "unit": "frames/s",
           "batched_replay": {"value": round(steps * frames / dt_b, 1), "unit": "frames/s", "frames_per_call": frames},
           "blocks": int(g.num_blocks())}
    if k_launches:
        avg_s = k_ms * 1e-3 / k_launches
        out["roofline"] = {"bound": "hbm", "kernels": "k_vgb_bin (fused unprojection + keys + block claim + bin placement + touched list) + k_vgb_fold_wave of one integrate_rgbd call (HIP events around the call's launches)",
                           "algorithmic_bytes_per_frame": int(b_vox + b_in), "avg_us_per_frame": round(avg_s * 1e6, 2),
                           "achieved": round((b_vox + b_in) / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round((b_vox + b_in) / avg_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                           "note": "B_vox = 2 x 28 B x distinct voxels touched per frame (oracle keys) + B_in 7 B/px: a ~15 MB/frame "
                                   "path bounded by the dependent memory round trips of two launches; the fold moves whole 128-byte lines for "
                                   "32-byte voxel records scattered over the frame's ~13 000 blocks (see traffic)"}
        # HBM bytes of the four per-frame kernels from the recorded --pmc passes of tools/bench_voxel_grid.py on this build
        try:
            with open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_voxel_grid.json")) as f:
                vp = json.load(f)
            if vp.get("build_digest") == current_build_digest():
                per = {}
                for name in ("k_vgb_bin", "k_vgb_fold_wave"):
                    t = hbm_traffic(next((v for k, v in vp.get("kernels", {}).items() if name in k), None))
                    if t is not None:
                        per[name] = t
                if len(per) == 2:
                    out["roofline"]["traffic"] = int(sum(per.values()))
                    out["roofline"]["traffic_per_kernel"] = per
                    out["roofline"]["traffic_GBs"] = round(sum(per.values()) / avg_s / 1e9, 1)
        except (OSError, ValueError):
            pass
    for kind, cls in (("reference", oracle.RefGrid if oracle.ref_available() else None), ("port", oracle.PortGrid)):
        if cls is None:
            continue
        c = cls(VOXEL, 8)
        c.integrate(*pts[0])
        t0 = time.perf_counter()
        for p, col in pts[1:]:
            c.integrate(p, col)
        dtc = time.perf_counter() - t0
        out["cpu_baseline" if kind == "reference" or "cpu_baseline" not in out else "cpu_port"] = {
            "value": round((len(pts) - 1) / dtc, 3), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": f"{len(pts) - 1} frames, integrate_raw<float,float> on the same float32 world points"
                      + (" (unmodified cpp/volumetric sources compiled by oracle/Makefile, sequential non-TBB branch)" if kind == "reference" else "")}
    # the reference's INTENDED configuration is its TBB branch (kVolumetricIntegrationTBBThreads, config_parameters.py:314); TBB headers
    # are not in this image, so the compiled reference above runs its sequential branch.  Beside it: the two-phase algorithm
    # (thread-local grouping by block, merge, blocks in parallel: voxel_block_grid.hpp:292-456) restated with OpenMP on all cores.
    try:
        cores = os.cpu_count() or 1
        best = None
        for th in sorted({t for t in (2, 8, 16, 32, 64, cores) if t <= cores}):
            c = oracle.PortGrid(VOXEL, 8)
            c.integrate_parallel(*pts[0], threads=th)
            t0 = time.perf_counter()
            for p, col in pts[1:]:
                c.integrate_parallel(p, col, threads=th)
            fps = (len(pts) - 1) / (time.perf_counter() - t0)
            if best is None or fps > best[0]:
                best = (fps, th)
        out["cpu_parallel"] = {"value": round(best[0], 3), "unit": "frames/s", "cores": best[1], "kind": "port",
                               "sample": f"restated, {best[1]} threads: {len(pts) - 1} frames, OpenMP restatement of integrate_raw_preorder_no_block_mutex "
                                         f"(the reference's TBB branch) on the same float32 world points; bit-identical to the sequential branch"}
    except Exception as e:
        out["cpu_parallel"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def config_legs():
    """Secondary: the other single-GPU TSDF shapes of BASELINE.json (parity for them: tests/test_gpu_configs.py, test_gpu_bench_path.py).
    configs[2]: Replica-shaped 1200x680 @ 4 mm + colour with marching cubes every 10 frames; configs[4]'s image / voxel shape
    (ScanNet 1296x968 @ 2 mm) on ONE GPU.  Frames resident in HBM; frames/s of fusion alone and with the extraction ticks."""
    import torch

    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    out = {}
    for key, config, voxel, n_frames, B, mesh_every, max_blocks in (("replica_1200x680_4mm", "replica_1200x680_4mm", 0.004, 40, 10, 10, 1 << 17),
                                                                       ("scannet_1296x968_2mm", "scannet_1296x968_2mm", 0.002, 16, 8, 0, 1 << 18)):
        s, depth_h, rgb_h, T_h = load_frames(config, n_frames)
        K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
        depth_d, rgb_d = torch.from_numpy(depth_h).cuda(), torch.from_numpy(rgb_h).cuda()
        vol = ScalableTSDFVolume(voxel, SDF_TRUNC, max_blocks=max_blocks, max_points=s.width * s.height)

        def run(extract, dtype=None):
            vol.reset()
            vol.synchronize()
            t0 = time.perf_counter()
            tri = 0
            for lo in range(0, n_frames, B):
                vol.integrate_batch(depth_d[lo:lo + B], rgb_d[lo:lo + B], K, T_h[lo:lo + B], depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
                if extract and mesh_every and (lo + B) % mesh_every == 0:
                    tri = len(vol.extract_triangle_mesh(dtype=dtype).triangles)
            vol.synchronize()
            torch.cuda.synchronize()
            return time.perf_counter() - t0, tri

        run(bool(mesh_every))  # warm-up: units allocated once, result arrays page-locked
        t_fuse = min(run(False)[0] for _ in range(2))
        leg = {"frames": n_frames, "frames_per_call": B, "voxel": voxel, "image": f"{s.width}x{s.height}",
               "fuse_only": {"value": round(n_frames / t_fuse, 1), "unit": "frames/s"}, "units": int(vol.num_blocks())}
        if mesh_every:
            t_all, tri = run(True)
            leg["with_mesh_every_%d_frames" % mesh_every] = {"value": round(n_frames / t_all, 1), "unit": "frames/s", "triangles_last": int(tri),
                                                              "what": "extract_triangle_mesh (host-visible result, D2H included) after every 10th frame"}
            run(True, np.float32)  # (page-locks the float32 result arrays)
            leg["with_float32_mesh_every_%d_frames" % mesh_every] = {"value": round(n_frames / run(True, np.float32)[0], 1), "unit": "frames/s",
                                                                      "what": "the same with extract_triangle_mesh(dtype=np.float32) (opt-in: Open3D's arrays are float64)"}
        out[key] = leg
        del vol, depth_d, rgb_d
    try:
        out["euroc_752x480_10mm"] = euroc_leg()
    except Exception as e:
        out["euroc_752x480_10mm"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def euroc_leg(n_frames=48):
    """BASELINE configs[3]'s single-GPU half: EuRoC-shaped stereo keyframes (752x480, f = 435.2 px, bf = 47.9 px m: settings/
    EuRoC_stereo.yaml:18-35), depth from a torch stereo module on the device (no RAFT-Stereo weights offline: the deterministic stub
    of pyslam_amd/depth_estimation.py stands in for it - a 5x5 conv, so the figure is the HAND-OFF and the fusion, not a depth
    network), shadow-point filter on the device (volumetric_integrator_base.py:989-1004), 1 cm TSDF, one integrate per keyframe:
    depth never visits the host.  Parity of this shape: tests/test_gpu_configs.py, tests/test_gpu_dense.py."""
    import types

    import torch

    from pyslam_amd.depth_estimation import DepthEstimatorStereoTorch, make_stub_stereo_net
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, depth_h, rgb_h, T_h = load_frames("euroc_752x480_10mm", n_frames)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    cam = types.SimpleNamespace(bf=47.9, width=s.width, height=s.height)
    est = DepthEstimatorStereoTorch(make_stub_stereo_net(), cam, device="cuda", keep_on_device=True, max_depth=10.0)
    left = [np.ascontiguousarray(rgb_h[i]) for i in range(n_frames)]
    right = [np.ascontiguousarray(np.roll(rgb_h[i], -8, axis=1)) for i in range(n_frames)]
    vol = ScalableTSDFVolume(0.010, SDF_TRUNC, max_blocks=1 << 15, max_points=s.width * s.height)

    def run():
        vol.reset()
        vol.synchronize()
        t0 = time.perf_counter()
        for i in range(n_frames):
            depth_d, _ = est.infer(left[i], right[i])                 # torch, device-resident
            depth_d = vol.filter_shadow_points(depth_d)                # radix-select median on the device
            color_d = torch.from_numpy(left[i]).cuda(non_blocking=True)
            vol.integrate(RGBDImage(color_d, depth_d, 1.0, DEPTH_TRUNC), K, T_h[i])
        vol.synchronize()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run()
    dt = min(run() for _ in range(2))
    return {"value": round(n_frames / dt, 1), "unit": "frames/s", "frames": n_frames, "voxel": 0.010, "image": f"{s.width}x{s.height}",
            "units": int(vol.num_blocks()),
            "what": "stereo keyframes: uploads of the pair, stub stereo module (torch) -> disparity -> depth = bf / |d| on the device, shadow-point "
                    "filter, hv_tsdf_integrate from device pointers; one keyframe per call; the 2-GPU tile-sharded form of configs[3]: "
                    "`bench.py --gpus 2 --sharding tile` -> configs.euroc_752x480_10mm (euroc_sharded_leg)"}


def euroc_sharded_leg(rank, world, local_rank, dist, backend, sharding, n_frames=32):
    """BASELINE configs[3] as it is written: EuRoC-shaped stereo keyframes on `world` GPUs, tile-sharded (every rank sees every keyframe
    pair, estimates the depth on its own GPU and fuses the voxels that project into its image tile; one merge_halo() at the end
    consolidates the units several ranks hold).  Runs on ALL ranks of an N > 1 run with --sharding tile; on a 1-GPU box the ranks
    share the GPU over gloo (tools/_final.sh), which exercises the code path and is not a scaling number."""
    import types

    import torch

    from pyslam_amd.depth_estimation import DepthEstimatorStereoTorch, make_stub_stereo_net
    from pyslam_amd.distributed import ShardedTSDF
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage

    s, depth_h, rgb_h, T_h = load_frames("euroc_752x480_10mm", n_frames, rank=rank, barrier=dist.barrier)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    cam = types.SimpleNamespace(bf=47.9, width=s.width, height=s.height)
    est = DepthEstimatorStereoTorch(make_stub_stereo_net(), cam, device="cuda", keep_on_device=True, max_depth=10.0)
    left = [np.ascontiguousarray(rgb_h[i]) for i in range(n_frames)]
    right = [np.ascontiguousarray(np.roll(rgb_h[i], -8, axis=1)) for i in range(n_frames)]
    fuser = ShardedTSDF(0.010, SDF_TRUNC, s.width, s.height, device=local_rank, max_blocks=1 << 15, rank=rank, world_size=world,
                        sharding=sharding)
    vol = fuser.volume
    on_gpu = backend == "nccl"

    def fence():
        vol.synchronize()
        torch.cuda.synchronize()
        dist.barrier()

    def run():
        vol.reset()
        fence()
        t0 = time.perf_counter()
        for i in range(n_frames):
            depth_d, _ = est.infer(left[i], right[i])
            depth_d = vol.filter_shadow_points(depth_d)
            color_d = torch.from_numpy(left[i]).cuda(non_blocking=True)
            fuser.integrate(RGBDImage(color_d, depth_d, 1.0, DEPTH_TRUNC), K, T_h[i])
        fence()
        t_fuse = time.perf_counter() - t0
        shared = 0
        if sharding != "owner":
            shared, _ = fuser.merge_halo()
            fence()
        t_all = time.perf_counter() - t0
        t = torch.tensor([t_fuse, t_all], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0]), float(t[1]), int(shared)

    run()
    t_fuse, t_all, shared = run()
    units = torch.tensor([float(vol.num_blocks())], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    everyone = [torch.zeros_like(units) for _ in range(world)]
    dist.all_gather(everyone, units)
    return {"value": round(n_frames / t_all, 1), "unit": "frames/s", "n_gpus": world, "sharding": sharding, "frames": n_frames, "voxel": 0.010,
            "image": f"{s.width}x{s.height}", "fuse_only": round(n_frames / t_fuse, 1), "merge_ms": round((t_all - t_fuse) * 1e3, 3),
            "shared_units": shared, "units_held": [int(e.item()) for e in everyone],
            "what": "stereo keyframes on every rank: stub stereo module (torch) -> depth on the device -> shadow-point filter -> hv_tsdf_integrate "
                    "of the rank's image tile; one merge_halo() (all-gather of key lists, all-reduce of the shared units' numerators) inside "
                    "the clock; slowest rank"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=64, help="posed frames per integrate_batch call (<= 64: the sweep's frame mask; 64 is what the drop-in front drains per call)")
    ap.add_argument("--clock-ramp-steps", type=int, default=60, help="untimed steps before the warm-up steps (device clock ramp, ~40 ms)")
    ap.add_argument("--config", default="synthetic_640x480_5mm")
    ap.add_argument("--window", choices=["sliding", "replay"], default="sliding")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline; 0 = min(os.cpu_count(), 32) (the stated policy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the replay / online / extraction / voxel-grid legs (profiling runs)")
    ap.add_argument("--no-batch32", action="store_true", help="skip the 32-frames-per-call leg (profiling runs: every sweep launch of the command is then a --frames-per-step one)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (and run every barrier / all-reduce / all-gather of the N > 1 line) with ONE rank: "
                         "RCCL accepts a one-rank group, so the N > 1 code path of this file can be executed on a one-GPU box")
    ap.add_argument("--all-on-device0", action="store_true",
                    help="testing only: every rank uses GPU 0 (multi-rank code path on a 1-GPU box, with --backend gloo)")
    ap.add_argument("--sharding", choices=["owner", "tile"], default="owner",
                    help="N > 1: owner = unit-ownership sharding by hash, no collective while fusing (default); "
                         "tile = image tiles + RCCL merge of the shared units at the end")
    ap.add_argument("--mode", choices=["batch", "online"], default="batch",
                    help="batch: one multi-frame sweep per step (replay/rebuild path); online: one integrate() per frame")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback exists)")
    if args.all_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from pyslam_amd.distributed import ShardedTSDF
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage

    B = args.frames_per_step
    n_poses = SyntheticRGBD(args.config).n_poses
    n_distinct = B if args.window == "replay" else min(args.steps * B, n_poses)
    s, depth_h, rgb_h, T_h = load_frames(args.config, n_distinct, rank=rank, barrier=dist.barrier if dist is not None else None)
    # resident stream: the distinct frames + the first B again, so that a window that wraps the loop is still one
    # contiguous slice (no gather inside the timed region)
    wrap = np.arange(n_distinct + B) % n_distinct
    Kcam = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    depth_d = torch.from_numpy(depth_h[wrap]).cuda()
    rgb_d = torch.from_numpy(rgb_h[wrap]).cuda()
    T_res = T_h[wrap]

    fuser = ShardedTSDF(VOXEL, SDF_TRUNC, s.width, s.height, device=local_rank, max_blocks=1 << 17,
                        rank=rank, world_size=world, sharding=args.sharding, force_collectives=args.force_dist)
    vol = fuser.volume

    def step(k, mode=None, window=None):
        lo = (k * B) % n_distinct if (window or args.window) == "sliding" else 0
        if (mode or args.mode) == "batch":
            vol.integrate_batch(depth_d[lo:lo + B], rgb_d[lo:lo + B], Kcam, T_res[lo:lo + B], depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
        else:
            for f in range(lo, lo + B):
                vol.integrate(RGBDImage(rgb_d[f], depth_d[f], 1.0, DEPTH_TRUNC), Kcam, T_res[f])

    def fence():
        vol.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # clock ramp (not a warm-up step, not timed): a fresh box starts the timed region on idle clocks - the 5 warm-up steps of the
    # driver's default command are 3 ms of work, and the first ~15 ms of sustained load run 10-15 % slower (tools/sweep_variants.py:
    # first pass of a process 660 us per sweep, every later pass 575-585).  The same step, repeated on the volume the warm-up uses.
    # the same K steps on COLD clocks first (reported beside the headline as `cold_clock`: what a rebuild() after a loop closure sees
    # when the GPU was idle before it): the driver's W warm-up steps, then K timed steps, before any clock ramp
    cold = None
    if args.clock_ramp_steps > 0 and args.mode == "batch":
        for k in range(args.warmup):
            step(k)
        fence()
        if args.window == "sliding":
            vol.reset()
            fence()
        tc = time.perf_counter()
        for k in range(args.steps):
            step(k)
        fence()
        cold = time.perf_counter() - tc
        if dist is not None:
            t = torch.tensor([cold], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cold = float(t.item())
    for k in range(args.clock_ramp_steps):
        step(k % max(1, n_distinct // B))
    fence()
    for k in range(args.warmup):
        step(k)
    if (world > 1 or args.force_dist) and args.sharding == "tile":
        fuser.merge_halo()  # part of the warm-up: the merge's kernels are loaded and its buffers exist before the clock starts
    fence()
    if args.window == "sliding":
        vol.reset()  # the timed region starts on an empty volume: every unit is allocated inside it
        fence()
    vol.profile_enable(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    merge = None
    if (world > 1 or args.force_dist) and args.sharding == "tile":
        # tile sharding leaves partial means on the units several ranks updated: all-reduce of those units over RCCL (timed, and
        # reported on its own so that the first multi-GPU run can be read: fuse time vs merge time vs bytes moved)
        fence()
        t_fuse = time.perf_counter() - t0
        tm = time.perf_counter()
        n_shared, n_dirty = fuser.merge_halo()
        fence()
        merge = {"ms": round((time.perf_counter() - tm) * 1e3, 3), "fuse_ms": round(t_fuse * 1e3, 3), "shared_units": int(n_shared),
                 "dirty_units_this_rank": int(n_dirty), "bytes_all_reduced": int(fuser.last_halo["payload_bytes"]),
                 "what": "one merge_halo() after the timed steps: all-gather of dirty / held unit keys, all-reduce(SUM) of the shared units' "
                         "numerators (81 920 B per unit) in 64 MB buckets, unpack; inside the timed region"}
    fence()
    elapsed = time.perf_counter() - t0
    launch_ms = vol.profile_launches()
    vol.profile_read()
    vol.profile_enable(False)
    per_rank = None
    if dist is not None:
        # every rank's own clock, its mean sweep launch (HIP events) and the units it holds: how to read the first real N-GPU run
        mine = torch.tensor([elapsed / args.steps * 1e3, float(np.mean(launch_ms)) if len(launch_ms) else 0.0, float(vol.num_blocks()),
                             merge["ms"] if merge else 0.0], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        rows = [e.cpu().tolist() for e in everyone]
        per_rank = {"ms_per_step": [round(r[0], 4) for r in rows], "sweep_ms_per_step": [round(r[1], 4) for r in rows],
                    "units_held": [int(r[2]) for r in rows], "merge_ms": [round(r[3], 3) for r in rows],
                    "what": "per rank: wall ms per step on the rank's own clock, mean duration of its sweep launch (HIP events on the kernel's "
                            "stream; the touch + pack launch of the next batch runs beside it on a second stream), units in its pool, "
                            "merge_halo() ms (tile sharding)"}
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = args.steps * B
    fps = frames / elapsed
    units_allocated = int(vol.num_blocks())
    euroc_sharded = None
    if dist is not None and args.sharding == "tile" and not args.no_secondary:
        try:  # (every rank takes part: collectives inside)
            euroc_sharded = euroc_sharded_leg(rank, world, local_rank, dist, args.backend, args.sharding)
        except Exception as e:  # a secondary leg must never cost the headline line
            import traceback

            traceback.print_exc()
            euroc_sharded = {"error": f"{type(e).__name__}: {e}"}
    secondary = world == 1 and args.mode == "batch" and not args.no_secondary

    # ---- secondary legs (single GPU): extraction of the volume just built, replay figure, online mode ----
    extraction = replay = online = batch32 = None
    secondary_error = None
    try:
        if secondary:
            # first output tick of the process: includes the one-off page-locking of the host result arrays
            t1 = time.perf_counter()
            mesh = vol.extract_triangle_mesh()
            t_mesh_cold = time.perf_counter() - t1
            t1 = time.perf_counter()
            pc = vol.extract_point_cloud()
            t_pc_cold = time.perf_counter() - t1
            del mesh, pc
            # every later tick (what a running reconstruction pays): one more keyframe - a revisit, so the unit set stays the
            # one the timed region built - invalidates the cached result, then both extractions run again
            vol.integrate(RGBDImage(rgb_d[0], depth_d[0], 1.0, DEPTH_TRUNC), Kcam, T_res[0])
            fence()
            vol.profile_enable(True)
            t1 = time.perf_counter()
            mesh = vol.extract_triangle_mesh()
            t_mesh = time.perf_counter() - t1
            k_mesh = vol.profile_read()[0]
            # another keyframe, so that the point-cloud tick pays for its own pass over the planes (the column masks both
            # extractions start from are kept per content version)
            vol.profile_enable(False)
            vol.integrate(RGBDImage(rgb_d[1], depth_d[1], 1.0, DEPTH_TRUNC), Kcam, T_res[1])
            fence()
            vol.profile_enable(True)
            t1 = time.perf_counter()
            pc = vol.extract_point_cloud()
            t_pc = time.perf_counter() - t1
            k_pc = vol.profile_read()[0]
            vol.profile_enable(False)
            nv, nt, npts = len(mesh.vertices), len(mesh.triangles), len(pc.points)
            # an unchanged volume extracted again (a paused map): the device result is cached, what is left is the D2H copy
            del mesh
            t1 = time.perf_counter()
            mesh = vol.extract_triangle_mesh()
            t_mesh_fetch = time.perf_counter() - t1
            # ... and a tick whose consumer lives on the GPU (device=True: torch CUDA tensors, nothing crosses PCIe)
            del mesh
            vol.integrate(RGBDImage(rgb_d[2], depth_d[2], 1.0, DEPTH_TRUNC), Kcam, T_res[2])
            fence()
            t1 = time.perf_counter()
            mesh = vol.extract_triangle_mesh(device=True)
            torch.cuda.synchronize()
            t_mesh_dev = time.perf_counter() - t1
            assert mesh.vertices.is_cuda and len(mesh.vertices) > 0
            # ... and a tick handed over as float32 (dtype=np.float32: hv_tsdf_extract_mesh_f32 - the float64 values rounded once on the
            # device, what pySLAM's viewer casts the arrays to; opt-in, Open3D's arrays are float64)
            del mesh
            vol.integrate(RGBDImage(rgb_d[4], depth_d[4], 1.0, DEPTH_TRUNC), Kcam, T_res[4])
            fence()
            vol.extract_triangle_mesh(dtype=np.float32)  # (its page-locked result arrays exist from here on)
            vol.integrate(RGBDImage(rgb_d[5], depth_d[5], 1.0, DEPTH_TRUNC), Kcam, T_res[5])
            fence()
            vol.profile_enable(True)
            t1 = time.perf_counter()
            mesh = vol.extract_triangle_mesh(dtype=np.float32)
            t_mesh_f32 = time.perf_counter() - t1
            k_mesh_f32 = vol.profile_read()[0]
            vol.profile_enable(False)
            assert mesh.vertices.dtype == np.float32 and len(mesh.vertices) > 0
            # ... and a FULL pass (HV_EXTRACT_INCREMENTAL=0: masks, classification and counts of every unit again - what every tick cost
            # before round 6 and what the first extraction of a volume still costs), after one more keyframe
            del mesh
            vol.integrate(RGBDImage(rgb_d[3], depth_d[3], 1.0, DEPTH_TRUNC), Kcam, T_res[3])
            fence()
            os.environ["HV_EXTRACT_INCREMENTAL"] = "0"
            try:
                vol.profile_enable(True)
                mesh = vol.extract_triangle_mesh(device=True)
                torch.cuda.synchronize()
                k_mesh_full = vol.profile_read()[0]
                vol.profile_enable(False)
            finally:
                del os.environ["HV_EXTRACT_INCREMENTAL"]
            b_mc_in = units_allocated * 4096 * 8
            b_mesh_out = nv * 48 + nt * 12
            b_pc_out = npts * 48
            extraction = {
                "what": "extract_triangle_mesh + extract_point_cloud of the volume the timed region built "
                        "(host-visible results: size query = all device work, fetch = D2H into page-locked numpy arrays; *_wall_ms = a tick of a "
                        "running reconstruction, *_first_call_ms = the first tick, which also page-locks the result arrays)",
                "units": units_allocated, "vertices": nv, "triangles": nt, "points": npts,
                "mesh_wall_ms": round(t_mesh * 1e3, 2), "mesh_kernel_ms": round(k_mesh, 3), "mesh_full_pass_kernel_ms": round(k_mesh_full, 3),
                "mesh_kernel_what": "mesh_kernel_ms / points_kernel_ms: a TICK - one keyframe fused since the previous extraction, so column masks, "
                                    "marching-cubes classification and point counts are recomputed only around the units that keyframe wrote to "
                                    "(per-unit caches, DESIGN section 4), vertices / triangles / points emitted in full; "
                                    "mesh_full_pass_kernel_ms: every unit recomputed (HV_EXTRACT_INCREMENTAL=0; rounds 2-5 paid this on every tick)",
                "mesh_fetch_only_ms": round(t_mesh_fetch * 1e3, 2),
                "mesh_device_resident_wall_ms": round(t_mesh_dev * 1e3, 2),
                "mesh_f32_wall_ms": round(t_mesh_f32 * 1e3, 2), "mesh_f32_kernel_ms": round(k_mesh_f32, 3),
                "mesh_f32_what": "the same tick with float32 vertices / colours (extract_triangle_mesh(dtype=np.float32): 24 instead of 48 bytes per vertex "
                                 "written and copied to the host; opt-in - Open3D's and the reference's arrays are float64)",
                "mesh_first_call_ms": round(t_mesh_cold * 1e3, 2), "points_first_call_ms": round(t_pc_cold * 1e3, 2),
                "points_wall_ms": round(t_pc * 1e3, 2), "points_kernel_ms": round(k_pc, 3),
                "roofline": {"bound": "hbm", "kernel": "FULL pass: k_unit_masks + k_mc_classify over every unit + scans + k_mc_vertices + k_mc_triangles",
                             "algorithmic_bytes": int(b_mc_in + b_mesh_out), "achieved": round((b_mc_in + b_mesh_out) / (k_mesh_full * 1e-3) / 1e9, 1),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((b_mc_in + b_mesh_out) / (k_mesh_full * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "traffic": None, "kernel_ms": round(k_mesh_full, 3),
                             "note": "B_mc = U_alloc x 4096 x 8 B (tsdf + weight read once) + output bytes (SURVEY 8d)"},
                "tick_roofline": {"bound": "hbm", "kernel": "a tick: masks + classification of the units written since the previous extraction, scans, k_mc_vertices + k_mc_triangles in full",
                                  "algorithmic_bytes": int(nv * 40 + b_mesh_out), "achieved": round((nv * 40 + b_mesh_out) / (k_mesh * 1e-3) / 1e9, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((nv * 40 + b_mesh_out) / (k_mesh * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "traffic": None, "kernel_ms": round(k_mesh, 3),
                                  "note": "lower bound: 2 voxels x 20 B gathered per vertex + output bytes (the planes of the few units one keyframe wrote to are not counted)"},
                "points_roofline": {"bound": "hbm", "kernel": "k_unit_masks + k_pc_extract x2 (count pass + fill pass inside the size query)", "algorithmic_bytes": int(b_mc_in + b_pc_out),
                                    "achieved": round((b_mc_in + b_pc_out) / (k_pc * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round((b_mc_in + b_pc_out) / (k_pc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None},
            }
            del mesh, pc
            # replay figure (round-1 headline): the same 32 frames re-fused every step, no allocation after the first
            vol.reset()
            for k in range(3):
                step(k, window="replay")
            fence()
            vol.profile_enable(True)
            t1 = time.perf_counter()
            n_rep = max(4, args.steps // 2)
            for k in range(n_rep):
                step(k, window="replay")
            fence()
            dt = time.perf_counter() - t1
            rep_ms, rep_launches, _ = vol.profile_read()
            vol.profile_enable(False)
            replay = {"value": round(n_rep * B / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / n_rep * 1e3, 4),
                      "sweep_avg_launch_us": round(rep_ms / max(rep_launches, 1) * 1e3, 2),
                      "what": f"every step re-fuses frames 0..{B - 1} (maximum frustum overlap, nothing allocated in the timed region)"}
            # online mode on the sliding stream, fresh volume
            vol.reset()
            fence()
            n_on = max(2, min(args.steps // 4, n_distinct // B))
            vol.profile_enable(True)
            t1 = time.perf_counter()
            for k in range(n_on):
                step(k, mode="online", window="sliding")
            fence()
            dt = time.perf_counter() - t1
            on_launch_ms = vol.profile_launches()
            vol.profile_read()
            vol.profile_enable(False)
            online = {"fps": n_on * B / dt, "launch_ms": on_launch_ms, "steps": n_on}
            # the headline's stream in calls of 32 frames (the batch size of rounds 1-5 and of round 6's earlier lines): same frames, same
            # clock procedure (warm-up, reset, timed steps on an empty volume), twice the launches
            if B > 32 and not args.no_batch32:
                B32 = 32
                n32 = args.steps * B // B32

                def step32(k):
                    lo = (k * B32) % n_distinct
                    vol.integrate_batch(depth_d[lo:lo + B32], rgb_d[lo:lo + B32], Kcam, T_res[lo:lo + B32], depth_scale=1.0, depth_trunc=DEPTH_TRUNC)

                vol.reset()
                for k in range(args.warmup):
                    step32(k)
                fence()
                vol.reset()
                fence()
                vol.profile_enable(True)
                t1 = time.perf_counter()
                for k in range(n32):
                    step32(k)
                fence()
                dt = time.perf_counter() - t1
                b32_launch_ms = vol.profile_launches()
                vol.profile_read()
                vol.profile_enable(False)
                batch32 = {"fps": n32 * B32 / dt, "ms_per_step": dt / n32 * 1e3, "launch_ms": b32_launch_ms, "steps": n32}


    except Exception as e:  # a secondary leg must never cost the headline line
        import traceback

        traceback.print_exc()
        secondary_error = f"{type(e).__name__}: {e}"

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            # the timed CPU baseline belongs to the N=1 line; at N>1 a two-step oracle pass still provides the counts
            budget = args.cpu_budget_s if world == 1 else 0.0
            cpu = cpu_baseline_and_counts(s, depth_h, rgb_h, T_h, B, args.steps, budget, args.cpu_threads if world == 1 else min(32, os.cpu_count() or 1), args.window)
        digest = current_build_digest()
        pmc = pmc_summary(digest)
        command_key = f"steps={args.steps} warmup={args.warmup} B={B} window={args.window} config={args.config}"

        roofline = None
        if cpu is not None and len(launch_ms) and args.mode == "batch":
            # launches covered by the oracle pass (normally all of them)
            n_cov = min(len(cpu["steps"]), len(launch_ms))
            t_cov = float(np.sum(launch_ms[:n_cov])) * 1e-3
            avg_s = t_cov / n_cov
            alg = sum(st["union_units"] * UNIT_BYTES + st["union_voxels"] * BYTES_PER_VOXEL for st in cpu["steps"][:n_cov]) / world
            visits = sum(sum(st["touched"]) for st in cpu["steps"][:n_cov]) * 4096 / world
            pk = pmc_kernel(pmc, "k_tsdf_sweep", command_key) if world == 1 else None
            traffic = hbm_traffic(pk)
            traffic_source = (f"replayed: profiles/{PROFILE_ROUND}/pmc_summary.json, rocprofv3 --pmc passes the builder recorded of this command "
                              f"on this library build") if traffic else None
            if world == 1 and not args.no_secondary:
                live = live_pmc(args, B)
                if live is not None:
                    traffic = hbm_traffic(live)
                    traffic_source = (f"live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of this command inside this "
                                      f"run, mean over the {live['launches']} sweep launches of the timed steps")
            hbm = {"algorithmic_bytes_per_launch": int(alg / n_cov), "achieved": round(alg / t_cov / 1e9, 1), "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": round(alg / t_cov / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                   "what": "BATCH-level algorithmic bytes (oracle): distinct units touched by the batch x 81 920 B read + distinct "
                           "voxels updated x 20 B written; a unit's slab is read and written once per launch whatever the number of frames"}
            if traffic:
                hbm["traffic_GBs"] = round(traffic / avg_s / 1e9, 1)
                hbm["traffic_over_algorithmic"] = round(traffic / (alg / n_cov), 3)
            roofline = {"bound": "hbm", "kernel": "k_tsdf_sweep_column (multi-frame sweep: a batch folded per voxel, one lane per voxel column)",
                        "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"], "traffic": traffic,
                        "algorithmic_bytes_per_launch": hbm["algorithmic_bytes_per_launch"], "what": hbm["what"],
                        "avg_launch_us": round(avg_s * 1e6, 2), "launches": int(n_cov), "frames_per_launch": B,
                        "voxel_visits_per_launch": int(visits / n_cov)}
            roofline["traffic_source"] = traffic_source
            # SURVEY 8d's PER-FRAME byte model beside it (what B successive one-frame integrations of the same frames would move): the
            # batch reads and writes a unit once per launch instead of once per frame - reuse, not bandwidth, is why frames/s exceed what
            # the per-frame model allows at 8 TB/s
            per_frame = sum((t * UNIT_BYTES + u * BYTES_PER_VOXEL) for st in cpu["steps"][:n_cov] for t, u in zip(st["touched"], st["updated"])) / world
            roofline["per_frame_model"] = {"algorithmic_bytes_per_frame": int(per_frame / (n_cov * B)), "equivalent_GBs": round(per_frame / t_cov / 1e9, 1),
                                           "batch_bytes_over_per_frame_bytes": round(alg / per_frame, 4),
                                           "what": "SURVEY 8d: touched units x 81 920 B + updated voxels x 20 B PER FRAME, summed over the launch's frames, / the "
                                                   "launch duration: the rate B one-frame integrations would need for the same frames/s (above the HBM peak: a "
                                                   "launch moves each unit once for all its frames)"}
            try:
                copy_gbs = d2d_copy_gbs()
                roofline["peak_measured"] = {"value": round(copy_gbs, 1), "unit": "GB/s", "frac": round(alg / t_cov / 1e9 / copy_gbs, 4),
                                             "what": "device-to-device copy of 1 GiB measured in this run (bytes read + written per second); "
                                                     "frac = achieved / this figure, beside frac against the vendor's 8 TB/s"}
            except Exception as e:
                roofline["peak_measured"] = {"error": f"{type(e).__name__}: {e}"}
            if traffic:
                roofline["traffic_GBs"] = hbm["traffic_GBs"]
                roofline["traffic_over_algorithmic"] = hbm["traffic_over_algorithmic"]
                rec_bytes = B * s.width * s.height * 12
                roofline["traffic_note"] = (f"beyond the algorithmic bytes the counters see the launch's frame records - {B} frames x {s.width * s.height} pixels x 12 B = "
                                            f"{rec_bytes / 1e6:.0f} MB, gathered once per voxel visit - arriving in each of the 8 XCDs' L2s: up to 8 x {rec_bytes / 1e6:.0f} MB "
                                            f"= {8 * rec_bytes / 1e9:.2f} GB of fabric requests, served by the 256 MB Infinity Cache, which FETCH_SIZE counts "
                                            f"(MI355X_MICROARCH.md, HBM section); the planes themselves move once per launch")
            tf = FLOP_PER_VISIT * visits / t_cov / 1e12
            roofline["flops"] = {"achieved": round(tf, 2), "peak": VECTOR_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / VECTOR_F32_PEAK_TFLOPS, 4),
                                 "what": f"{FLOP_PER_VISIT} flop per voxel visit (SURVEY 8d) x the oracle's visits (touched units x 4096, per frame) vs the vector fp32 peak"}
            util = {"what": "the kernel's OWN VALU instruction count per second against the wave64 issue rate: how busy the vector "
                            "issue slots are, not a fraction of an algorithmic roofline (wasted instructions raise it)",
                    "achieved": None, "unit": "G wave-instr/s", "peak_4cyc": VALU_PEAK_GINSTR, "peak_2cyc": VALU_PEAK_GINSTR_2CYC}
            if pk and "SQ_INSTS_VALU" in pk:
                ach = pk["SQ_INSTS_VALU"] / avg_s / 1e9
                util.update({"achieved": round(ach, 1), "frac_of_4cyc_rate": round(ach / VALU_PEAK_GINSTR, 4), "frac_of_2cyc_rate": round(ach / VALU_PEAK_GINSTR_2CYC, 4),
                             "valu_instr_per_launch": int(pk["SQ_INSTS_VALU"]),
                             "valu_instr_per_voxel_visit": round(pk["SQ_INSTS_VALU"] * 64 / (visits / n_cov), 2),
                             "counter_source": os.path.relpath(PMC_SUMMARY, ROOT) + " (rocprofv3 --pmc SQ pass of this command on this build; "
                                               "duration measured live with HIP events)",
                             "peaks": "4-cycle: the sustained rate tools/valu_rates.hip measures for FMA-class / v_pk_* / cvt / cmp instructions "
                                      "(profiles/r02/valu_issue_rates.txt: 4.1-4.7 cycles; add / mul / mov 2.5); 2-cycle: MI355X_MICROARCH.md's v_fma_f32 row"})
                if "SQ_ACTIVE_INST_VALU" in pk and "GRBM_GUI_ACTIVE" in pk:
                    util["valu_busy_in_pmc_pass"] = round(pk["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * pk["GRBM_GUI_ACTIVE"] / 8), 4)
            else:
                util["note"] = ("VALU instruction count unavailable: " +
                                (f"profiles/{PROFILE_ROUND}/pmc_summary.json was recorded on another build or command" if pmc else f"no profiles/{PROFILE_ROUND}/pmc_summary.json"))
            roofline["valu_utilisation"] = util

        out = {
            "metric": "RGB-D frames/sec fused (640x480, 5 mm voxel TSDF)",
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: synthetic 640x480 RGB-D @ 30 Hz stream, 5 mm TSDF (sdf_trunc 0.04 m, depth_trunc 4 m), "
                            f"{B} posed frames per step resident in HBM, Open3D ScalableTSDFVolume semantics",
                "frames_per_step": B,
                "window": (f"sliding: step k fuses frames {B}k..{B}k+{B - 1} of the 600-pose loop into one volume that is empty when the timed region starts"
                           if args.window == "sliding" else f"replay: every step re-fuses frames 0..{B - 1}"),
                "mode": "multi-frame sweep (hv_tsdf_integrate_batch)" if args.mode == "batch" else "one hv_tsdf_integrate per frame",
                "sharding": "single spatial tile" if world == 1 else (
                    f"unit ownership: unit -> GPU hash(index) % {world}; every GPU sees every frame, fuses and stores only its "
                    f"units; no collective while fusing" if args.sharding == "owner"
                    else f"{world} vertical image tiles + RCCL merge of the shared units (timed)"),
                "units_allocated": units_allocated,
                "parity_pinned": False,  # Open3D (the reference's TSDF arithmetic) is not in the image: the oracle is a restatement (README)
                "clock_ramp_steps": args.clock_ramp_steps,
                "build_digest": digest,
            },
            "roofline": roofline,
            "cpu_baseline": None if (cpu is None or world > 1) else {
                "value": round(cpu["fps"], 3), "unit": "frames/s", "cores": cpu["threads"], "kind": "port",
                "sample": f"the first {cpu['frames']} frames of the same sliding stream ({cpu['seconds']:.1f} s of integrate calls, allocation "
                          f"included), oracle/tsdf_oracle.c (Open3D-semantics restatement, per-frame multiplier image as Open3D, "
                          f"gcc -O3 -march=native on this host; open3d itself is not installed), OpenMP over touched units; "
                          f"thread policy: cores = min(os.cpu_count(), 32) (--cpu-threads overrides; the restatement stops scaling there), the same "
                          f"code on one core in `single_core` and on every hardware thread of the box in `all_hw_threads`",
                "single_core": cpu.get("single_core"), "all_hw_threads": cpu.get("all_hw_threads"),
            },
        }
        if cold is not None:
            out["cold_clock"] = {"value": round(args.steps * B / cold, 2), "unit": "frames/s", "ms_per_step": round(cold / args.steps * 1e3, 4),
                                 "what": f"the same {args.steps} steps after the same {args.warmup} warm-up steps BEFORE the {args.clock_ramp_steps} "
                                         f"untimed clock-ramp steps: a fresh process on idle clocks (value above: after the ramp)"}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if cpu is not None and world == 1:
            out["speedup_vs_cpu"] = round(fps / cpu["fps"], 1)
        if merge is not None:
            out["merge"] = merge
        if euroc_sharded is not None:
            out["configs"] = {"euroc_752x480_10mm": euroc_sharded}
        if extraction is not None:
            # HBM traffic of the extraction kernels from the recorded --pmc passes of this command on this build (all launches of
            # the run extract the same volume)
            def kernels_traffic(names):
                tot = 0
                for nm in names:
                    t = hbm_traffic(pmc_kernel(pmc, nm, command_key))
                    if t is None:
                        return None
                    tot += t
                return int(tot)

            extraction["roofline"]["traffic"] = kernels_traffic(("k_unit_masks@full", "k_mc_classify@full", "k_mc_vertices<double>", "k_mc_triangles"))
            extraction["tick_roofline"]["traffic"] = kernels_traffic(("k_unit_masks", "k_mc_classify", "k_mc_vertices<double>", "k_mc_triangles"))
            for key in ("roofline", "tick_roofline"):
                r = extraction[key]
                if r["traffic"]:
                    r["traffic_over_algorithmic"] = round(r["traffic"] / r["algorithmic_bytes"], 2)
            extraction["points_roofline"]["traffic"] = kernels_traffic(("k_unit_masks", "k_pc_extract<false", "k_pc_extract<true"))
        if online is not None:
            om = {"value": round(online["fps"], 2), "unit": "frames/s",
                  "what": "one hv_tsdf_integrate per frame (pySLAM's online flow), same sliding stream, fresh volume", "roofline": None}
            if cpu is not None and len(cpu["steps"]) >= online["steps"]:
                n_l = online["steps"] * B
                t_l = float(np.sum(online["launch_ms"][:n_l])) * 1e-3
                alg = sum((t * UNIT_BYTES + u * BYTES_PER_VOXEL) for st in cpu["steps"][:online["steps"]] for t, u in zip(st["touched"], st["updated"]))
                pk = pmc_kernel(pmc, "k_tsdf_integrate<", command_key)
                om["roofline"] = {"bound": "hbm", "kernel": "k_tsdf_integrate", "achieved": round(alg / t_l / 1e9, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(alg / t_l / 1e9 / HBM_PEAK_GBS, 4), "traffic": hbm_traffic(pk),
                                  "algorithmic_bytes_per_launch": int(alg / n_l), "avg_launch_us": round(t_l / n_l * 1e6, 2), "launches": int(n_l),
                                  "frames_per_launch": 1}
            out["online_mode"] = om
        if batch32 is not None:
            b32 = {"value": round(batch32["fps"], 2), "unit": "frames/s", "frames_per_step": 32, "steps": batch32["steps"],
                   "ms_per_step": round(batch32["ms_per_step"], 4),
                   "what": "the same stream in calls of 32 frames (the headline's batch size of rounds 1-5): twice the launches, each unit's slab "
                           "read and written twice as often per frame - a shorter launch moves more algorithmic bytes per second (higher roofline "
                           "fraction) and fuses fewer frames per second", "roofline": None}
            if cpu is not None:
                try:  # the oracle's batch-level counts for 32-frame batches (a counts-only pass over the first 640 frames of the stream)
                    c32 = cpu_baseline_and_counts(s, depth_h, rgb_h, T_h, 32, min(batch32["steps"], 20), 60.0, cpu["threads"], args.window, extras=False)
                    n_cov = min(len(c32["steps"]), len(batch32["launch_ms"]))
                    t_cov = float(np.sum(batch32["launch_ms"][:n_cov])) * 1e-3
                    alg = sum(st["union_units"] * UNIT_BYTES + st["union_voxels"] * BYTES_PER_VOXEL for st in c32["steps"][:n_cov])
                    b32["roofline"] = {"bound": "hbm", "kernel": "k_tsdf_sweep_column", "achieved": round(alg / t_cov / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(alg / t_cov / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(alg / n_cov),
                                       "avg_launch_us": round(t_cov / n_cov * 1e6, 2), "launches": int(n_cov), "frames_per_launch": 32}
                except Exception as e:
                    b32["roofline"] = {"error": f"{type(e).__name__}: {e}"}
            out["batch32"] = b32
        if replay is not None:
            out["replay_mode"] = replay
        if extraction is not None:
            out["extraction"] = extraction
        if secondary_error:
            out["secondary_error"] = secondary_error
        if secondary and not args.no_cpu_baseline:
            del vol, fuser
            for key, leg in (("host_mode", lambda: __import__("tools.bench_host", fromlist=["host_leg"]).host_leg(
                                 s, depth_h, rgb_h, T_h, VOXEL, SDF_TRUNC, DEPTH_TRUNC, B=B, steps=min(6, n_distinct // B))),
                             ("tum1_640x480_5mm", lambda: __import__("tools.bench_tum", fromlist=["tum_leg"]).tum_leg(
                                 h2d_gbs=(out.get("host_mode") or {}).get("h2d_pinned_GBs"))),
                             ("configs", config_legs),
                             ("voxel_grid", lambda: voxel_grid_leg(s, depth_h, rgb_h, T_h, depth_d, rgb_d, min(32, n_distinct), 5, 6)),  # (32 frames: the leg of rounds 3-6, whatever the headline's batch)
                             ("semantic", lambda: __import__("tools.bench_semantic", fromlist=["semantic_leg"]).semantic_leg(26, 2, 0.01)),  # (24 timed keyframes: with 8 - rounds 3-6a - the timed region was 1.6 ms and one hiccup 10 % of it)
                             # 12 keyframes, 10 inside the clock: with 3 (rounds 4-5) the first keyframe's 19 MB upload - 0.33 ms that
                             # nothing can hide - was a sixth of the measurement; a backlog's steady state is what the figure is for
                             ("semantic_scannet_2mm", lambda: __import__("tools.bench_semantic", fromlist=["semantic_leg"]).semantic_leg(
                                 12, 1, 0.002, "scannet_1296x968_2mm", 2))):
                try:
                    out[key] = leg()
                except Exception as e:  # a secondary leg must never cost the headline line
                    import traceback

                    traceback.print_exc()
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
            # the headline the way a consumer meets it: host-fed figures as first-class keys with their own (PCIe) roofline
            hm = out.get("host_mode") or {}
            if "staged" in hm:
                out["host_fed"] = {"what": "the headline's stream handed over as pageable per-keyframe host arrays (PCIe inside the clock, "
                                           "volume empty when it starts): value = the float32-depth form pySLAM's queue delivers",
                                   "value": hm["staged"]["value"], "unit": "frames/s"}
                for k, bpp in (("staged", "float32 depth + uint8 x 3 colour: 7 B / pixel"), ("staged_u16", "uint16 depth (the sensor's format) + uint8 x 3 colour: 5 B / pixel")):
                    if k in hm and "value" in hm[k]:
                        fb = hm[k].get("bytes_per_frame", hm.get("bytes_per_frame"))
                        out["host_fed"][k] = {"value": hm[k]["value"], "unit": "frames/s", "bytes_per_frame": fb,
                                              "roofline": {"bound": "pcie", "what": bpp, "achieved": round(fb * hm[k]["value"] / 1e9, 2),
                                                           "peak": hm["h2d_pinned_GBs"], "unit": "GB/s", "peak_what": "pinned-memory H2D rate measured in this run",
                                                           "frac": round(fb * hm[k]["value"] / 1e9 / hm["h2d_pinned_GBs"], 3), "traffic": None}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
