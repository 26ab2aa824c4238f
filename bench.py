#!/usr/bin/env python
"""Benchmark of the fcn_object_detector hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one forward pass of the DetectNet GoogLeNet stack + coverage/bbox heads (the graph of the
reference's models/deploy.prototxt, BASELINE.json configs[1]) over one synthetic 448x448x3 frame that
is already resident in HBM.  N > 1 runs N independent replicas, one process per GPU (inference does not
shard: "replicas only"), launched by torch.distributed.run; ranks synchronise over
fcn_object_detector_amd.dp.ControlPlane (the rank processes stay PyTorch-free so a single HIP runtime is
loaded).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# CPU baseline leg only (the GPU path never touches BLAS): OpenBLAS workers spin for ~2^26 cycles after every sgemm and
# libgomp's spin too; with both pools alive (oracle/caffe_cpu.c uses OpenMP) they steal each other's cores.  Both read these
# when their library loads, i.e. before numpy is imported.
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "4")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402

FWD_GFLOP = 15.608          # 2*MAC over the 59 convolutions at 448x448
ALGO_BYTES_MB = 277.7       # fused algorithmic HBM bytes per forward frame (f32)
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
F16_ACT_MB = 253.7 / 2      # the f32 path's algorithmic activation bytes per frame, stored as halves
F16_WEIGHT_MB = 24.0 / 2    # weights once per batch
PROFILE_TAG = "r04"         # profiles/<tag>_*: the round's committed rocprofv3 summaries (tools/profile_bench.sh)


def kernel_source_hash() -> str:
    """sha256 over the HIP sources: profiles/*.json carry the hash of the sources they were measured on, so a traffic figure
    read back from a committed profile shows when the kernels have changed since (tools/profile_bench.sh stamps it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "fcn_object_detector_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "fcn_object_detector_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_traffic(tag: str, key: str = "conv_fwd_hbm"):
    """HBM bytes per launch of a kernel family from the committed PMC passes profiles/<tag>_pmc_{fetch,write}.json (separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    try:
        pf = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_fetch.json")))
        pw = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_write.json")))
        return {"bytes_per_launch": round(pf[key]["read_bytes_per_launch"] + pw[key]["write_bytes_per_launch"]),
                "profile": "profiles/%s_pmc_{fetch,write}.json" % tag, "profile_kernel_sources": pf.get("kernel_source_hash"),
                "stale": pf.get("kernel_source_hash") != kernel_source_hash()}
    except (OSError, KeyError, ValueError):
        return None


def infer32_traffic(fwd_ms: float):
    """HBM bytes per batch-32 f16 forward from the committed PMC passes (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE around tools/fwd_resident.py 32 f16 N, totals divided by the N forwards)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_infer32_f16_hbm.json")))
        return {"bytes_per_forward": d["bytes_per_forward"], "measured_GBps": round(d["bytes_per_forward"] / fwd_ms / 1e6, 1),
                "ratio_to_algorithmic": round(d["bytes_per_forward"] / ((32 * F16_ACT_MB + F16_WEIGHT_MB) * 1e6), 3),
                "profile": "profiles/%s_infer32_f16_hbm.json" % PROFILE_TAG, "stale": d.get("kernel_source_hash") != kernel_source_hash()}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(msg, params, x, budget_s: float = 20.0):
    """Caffe-algorithm CPU stand-in (oracle: im2col + OpenBLAS sgemm per conv, separate pool/LRN passes) on the host cores.
    OpenBLAS with one thread per core of a big box is slower than with fewer threads on these small GEMMs, so a few
    thread counts are tried first and the fastest is the one reported (`cores` = threads actually used)."""
    from oracle.net_ref import RefNet
    ref = RefNet(msg, "TEST", params)
    ref.blobs["data"] = x
    ref.forward()                                   # warm-up (page in BLAS, allocate)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads, limiter = avail, None
    try:
        from threadpoolctl import threadpool_limits
        best = None
        for nt in sorted({avail, min(avail, 64), min(avail, 32), min(avail, 16), min(avail, 8)}, reverse=True):
            with threadpool_limits(limits=nt):
                t0 = time.perf_counter()
                ref.forward()
                dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, nt)
        threads = best[1]
        limiter = threadpool_limits(limits=threads)
    except ImportError:
        pass
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 20):
        t0 = time.perf_counter()
        ref.forward()
        times.append(time.perf_counter() - t0)
    if limiter is not None:
        limiter.restore_original_limits()
    med = float(np.median(times))
    single = None
    try:      # additionally one thread
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ref.forward()
                ts.append(time.perf_counter() - t0)
        single = round(1.0 / float(np.median(ts)), 3)
    except ImportError:
        pass
    return {"value": round(1.0 / med, 3), "unit": "frames/s", "cores": threads, "kind": "port", "single_thread_frames_per_s": single,
            "blas": "OpenBLAS (numpy's bundled scipy-openblas)",
            "sample": "%d forward passes of the same 448x448 frame through the oracle = Caffe's CPU schedule restated: per "
                      "convolution im2col (skipped for 1x1, as Caffe's is_1x1_) + one OpenBLAS sgemm + bias pass, separate ReLU / "
                      "pooling / LRN passes; im2col, MAX pooling and LRN are compiled C loops with OpenMP (oracle/caffe_cpu.c), the "
                      "rest numpy (median %.1f ms, %d BLAS threads - the fastest of several counts - of %d available cores)"
                      % (len(times), med * 1e3, threads, avail)}, ref.blobs


def synth_boxes(rng, n_images, size=448):
    """BASELINE config 3 / SURVEY §8d: per image 1-3 rects, w,h ~ U{32..224}, x,y ~ U{0..size-1-w}, label 0."""
    rects, labels = [], []
    for _ in range(n_images):
        rs = []
        for _ in range(int(rng.integers(1, 4))):
            w, h = int(rng.integers(32, 225)), int(rng.integers(32, 225))
            rs.append((int(rng.integers(0, size - w)), int(rng.integers(0, size - h)), w, h))
        rects.append(rs)
        labels.append([0] * len(rs))
    return rects, labels


def profile_json(name: str):
    """profiles/<PROFILE_TAG>_<name>.json with a `stale` flag (kernel sources changed since it was measured), or None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "%s_%s.json" % (PROFILE_TAG, name))))
        d["stale"] = d.get("kernel_source_hash") != kernel_source_hash()
        return d
    except (OSError, ValueError):
        return None


def bench_train(cp, rank, world, local, per_gpu_batch, steps, warmup, trace_clean=False):
    """One data-parallel training step = target generation (device) + forward + backward + all-reduce + SGD update on
    `per_gpu_batch` synthetic 448x448 images per GPU (BASELINE configs[2] at N=1, configs[3] at N=8)."""
    from fcn_object_detector_amd import dp, lib as L, models, proto
    from fcn_object_detector_amd.netspec import NetSpec, fill_params
    from fcn_object_detector_amd.train import SolverParams, TrainEngine
    msg = proto.parse_text(models.googlenet_detectnet_train("synthetic", "Boxes", "448,448,16,1,%d,none" % per_gpu_batch, num_classes=1))
    n = per_gpu_batch
    shapes = {"data": (n, 3, 448, 448), "coverage-label": (n, 1, 28, 28)}
    for k in ("bbox-label", "size-block", "obj-block", "coverage-block"):
        shapes[k] = (n, 4, 28, 28)
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    comm = dp.RcclComm(cp, local) if world > 1 else None
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params=fill_params(spec, seed=1234), device=local, comm=comm,
                      solver=SolverParams(base_lr=1e-4, momentum=0.9, weight_decay=1e-7, lr_policy="fixed"))
    eng.dropout_index_offset = rank * n * 1024 * 28 * 28
    rng = np.random.default_rng(1000 + rank)
    eng.host_array("data")[...] = rng.random((n, 3, 448, 448), dtype=np.float32)
    eng.upload_inputs()                                          # images resident in HBM; labels are generated on the device
    losses = []
    for it in range(max(warmup, 1)):
        eng.set_targets(*synth_boxes(np.random.default_rng(42 + it * world + rank), n), stride=16)
        losses.append(eng.step(seed=it, upload=False)["total_loss"])
    L.call("fcn_device_sync")
    cp.barrier()
    t0 = time.perf_counter()
    for it in range(steps):
        eng.set_targets(*synth_boxes(np.random.default_rng(4242 + it * world + rank), n), stride=16)
        losses.append(eng.step(seed=1000 + it, upload=False)["total_loss"])
    L.call("fcn_device_sync")
    t_local = time.perf_counter() - t0
    cp.barrier()
    t_max = cp.max(t_local)
    res = {"imgs_per_s": round(steps * n * world / t_max, 2), "ms_per_step": round(t_max * 1e3 / steps, 3), "steps": steps,
           "per_gpu_batch": n, "global_batch": n * world, "n_gpus": world,
           "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)], "train_gflop_per_img": 45.9,
           "workload": "configs[%d]: DetectNet GoogLeNet train step (device target generation + fwd + bwd + %s + SGD), batch %d/GPU, 448x448, f32"
                       % (2 if world == 1 else 3, "RCCL all-reduce of 23.9 MB grads" if world > 1 else "no collective", n)}
    res["achieved_tflops"] = round(45.9e-3 * res["imgs_per_s"], 2)
    # roofline of the step's dominant kernel family: the MFMA weight gradient (conv_wgrad_group_kernel + its fixed-order
    # reduction of the pixel-split partials), HIP events per launch on the engine's stream
    # (a kernel-trace pass wants whole steps only: no isolated launches in the trace)
    wg = [] if trace_clean else [r for r in eng.time_ops(reps=3, ops=eng.bwd_ops) if r[0] == "wgrad"]
    wg_ms, wg_fl = sum(r[2] for r in wg), sum(r[3] for r in wg)
    if wg_ms > 0:
        iso = {"achieved": round(wg_fl / wg_ms / 1e9, 2), "frac": round(wg_fl / wg_ms / 1e9 / F32_MFMA_PEAK_TFLOPS, 4), "ms_per_step": round(wg_ms, 3),
               "note": "every weight-gradient launch timed ALONE, repeated back to back (HIP events): in the real step these kernels share the chip "
                       "with the data-gradient stream, so the in-step figure is lower"}
        prof = profile_json("train_roofline") if world == 1 else None
        if prof is not None and not prof["stale"]:
            # first-class: the kernel trace of the REAL two-stream step (tools/roofline_from_profile.py over profiles/<tag>_train_kernel_stats.csv)
            ach, frac, src = prof["achieved_tflops"], prof["frac"], prof["source"]
            ms_step = prof["wgrad_ms_per_step"]
        else:
            ach, frac, src, ms_step = iso["achieved"], iso["frac"], "live: isolated launches (no fresh profile of the step: %s)" % (
                "kernel sources changed since profiles/%s_train_roofline.json" % PROFILE_TAG if prof else "none committed"), iso["ms_per_step"]
        res["roofline"] = {"bound": "mfma", "kernel": "conv_wgrad_split_kernel / conv_wgrad_group_kernel (f32 MFMA, reduction over pixels; per launch the faster of the two, timed at plan time) + reduce_partials_group_kernel; %d launches" % len(wg),
                           "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": frac, "traffic": None, "source": src,
                           "ms_per_step": ms_step, "isolated_launches": iso,
                           "whole_step_frac": round(res["achieved_tflops"] / F32_MFMA_PEAK_TFLOPS, 4),
                           "note": "achieved = weight-gradient FLOPs of a step (2*Cout*K*M per layer) / the family's kernel time per step in the "
                                   "rocprofv3 kernel trace of the two-stream step; whole_step_frac = 45.9 GFLOP/img over the driver-timed step"}
    if comm is not None:
        # the collective by itself, and how much of it the backward pass hides.  "dry" = the same step
        # with every bucket's hand-off (events, stream waits) but no ncclAllReduce behind it.
        ar = eng.time_allreduce(reps=10)
        cp.barrier()
        eng.comm_dry = True
        for it in range(2):
            eng.step(seed=it, upload=False)
        L.call("fcn_device_sync")
        cp.barrier()
        t0 = time.perf_counter()
        dry_steps = max(min(steps, 10), 3)
        for it in range(dry_steps):
            eng.step(seed=2000 + it, upload=False)
        L.call("fcn_device_sync")
        t_dry = cp.max(time.perf_counter() - t0) * 1e3 / dry_steps
        eng.comm_dry = False
        ar_us = cp.max(ar["allreduce_us"])
        exposed_us = max(res["ms_per_step"] - t_dry, 0.0) * 1e3
        res["allreduce"] = {"us_alone": round(ar_us, 1), "bytes": ar["bytes"], "buckets": ar["buckets"],
                            "bus_GBps": round(2.0 * (world - 1) / world * ar["bytes"] / (ar_us * 1e-6) / 1e9, 1) if ar_us else 0.0,
                            "ms_per_step_without_collective": round(t_dry, 3), "exposed_us_per_step": round(exposed_us, 1),
                            "overlap_fraction": round(min(max(1.0 - exposed_us / ar_us, 0.0), 1.0), 3) if ar_us else None,
                            "note": "us_alone = all gradient buckets back to back on an otherwise idle GPU (max over ranks); "
                                    "exposed = step time with the collective - step time without it; bus GB/s = 2(G-1)/G x bytes / us_alone"}
    eng.close()
    # The same step driven by the reference's data layer (BASELINE configs[2] names it): DataArgumentationLayer plans a
    # scene per image on the host (RNG + box bookkeeping), the device composes / normalises it from HBM-resident object
    # images and generates the label grids; nothing but the placement records crosses PCIe.
    try:
        res["with_data_layer"] = bench_train_data_layer(cp, rank, world, local, comm, n, max(steps // 2, 5))
    except Exception as e:  # the headline numbers above must survive a failure here
        res["with_data_layer"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if comm is not None:
        comm.close()
    return res


def bench_train_data_layer(cp, rank, world, local, comm, n, steps):
    import tempfile
    from fcn_object_detector_amd import lib as L, models
    from fcn_object_detector_amd.solver import Solver
    pydir = os.path.join(ROOT, "fcn_object_detector_amd", "python")
    if pydir not in sys.path:
        sys.path.insert(0, pydir)
    os.environ["FCN_DATA_SEED"] = str(100 + rank)
    with tempfile.TemporaryDirectory() as tmp:
        net = os.path.join(tmp, "train_val.prototxt")
        with open(net, "w") as f:
            f.write(models.googlenet_detectnet_train("data_argumentation_layer", "DataArgumentationLayer",
                                                     "448,448,16,1,%d,synthetic:1,detectnet" % n, num_classes=1))
        sol = os.path.join(tmp, "solver.prototxt")
        with open(sol, "w") as f:
            f.write('net: "%s"\nbase_lr: 1e-4\nmomentum: 0.9\nweight_decay: 1e-7\nlr_policy: "fixed"\ndisplay: 0\nmax_iter: 1000000\n'
                    'snapshot: 0\nsnapshot_prefix: "%s"\n' % (net, os.path.join(tmp, "snap")))
        solver, err = None, ""
        try:
            solver = Solver(sol, device=local, comm=comm, rank=rank, log=None)
        except Exception as e:      # every rank must take the same branch below: no barrier may be left half-entered
            err = "%s: %s" % (type(e).__name__, e)
        errs = [e for e in cp.all_gather(err) if e]
        if errs:
            if solver is not None:
                solver.close()
            raise RuntimeError("; ".join(errs))
        solver.step(3, pipeline=True)
        L.call("fcn_device_sync")
        cp.barrier()
        t0 = time.perf_counter()
        out = solver.step(steps, pipeline=True)
        L.call("fcn_device_sync")
        t_local = time.perf_counter() - t0
        cp.barrier()
        t_max = cp.max(t_local)
        solver.close()
    return {"imgs_per_s": round(steps * n * world / t_max, 2), "ms_per_step": round(t_max * 1e3 / steps, 3), "steps": steps,
            "loss_last": round(out["total_loss"], 5),
            "workload": "caffe-train loop: DataArgumentationLayer (synthetic objects, scenes planned on the host, composed on the device) "
                        "+ device target generation + fwd + bwd + SGD, batch %d/GPU" % n}


def bench_infer32(local: int, dtype: str, reps: int = 10):
    """BASELINE configs[4]: 32 uint8 frames resident in HBM -> pre-processing -> one forward -> one fused decode +
    groupRectangles launch for all (image, class) pairs -> boxes on the host (FCNObjectDetector.run_detector_batch minus
    the frame upload).  dtype "f16": activations and weights stored as halves, v_mfma_f32_32x32x16_f16 with f32
    accumulation (the image is half too: the net's Power(-127) is folded into conv1's filters); "f32" beside it.
    Measured one batch at a time and with two batches in flight on replica engines (the read-back and the host-side
    unpacking of one batch overlap the forward of the next)."""
    from fcn_object_detector_amd import lib as L, models, proto
    from fcn_object_detector_amd.detector import DetectorPipeline, HeadMapping
    from fcn_object_detector_amd.engine import DeviceBuffer, Engine
    from fcn_object_detector_amd.netspec import NetSpec, fill_params
    n = 32
    msg = proto.parse_text(models.googlenet_detectnet_deploy(n, 448, 448, 4))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=1234)
    # random-init heads never fire, and a decode / groupRectangles launch without candidates times nothing: the head biases
    # are raised (as in tests/test_gpu_fullsize.py) so that every class has several hundred candidate cells per image and the
    # clustering finds real groups - the conv stack's work is unchanged
    brng = np.random.default_rng(9)
    params["cvg/classifier"][1][...] = 1.5
    params["bbox/regressor"][0][...] *= 0.05
    params["bbox/regressor"][1][...] = np.tile(np.array([-30, -25, 35, 40], np.float32), 4) + brng.normal(0, 0.5, 16).astype(np.float32)
    pipe = DetectorPipeline(lambda first: Engine(NetSpec(msg, "TEST"), params=params, device=local, dtype=dtype, tune_from=first), depth=2,
                            mapping=HeadMapping.detectnet_deploy())
    det = pipe.detectors[0]
    eng = det.engine
    frames = np.random.default_rng(9).integers(0, 256, (n, 448, 448, 3), dtype=np.uint8)
    dev = DeviceBuffer(frames.nbytes, zero=False)
    L.call("fcn_memcpy_h2d_async", dev.ptr, frames.ctypes.data, frames.nbytes, eng.stream)
    L.call("fcn_device_sync")
    layout = [(i * 448 * 448 * 3, 448, 448) for i in range(n)]
    for d in pipe.detectors:
        d._minmax_batch_holder[:] = [DeviceBuffer(32 * n)]

    def submit(d):
        with d.engine.lock:
            d._enqueue_batch(dev.ptr, layout, True)
            d._outstanding = [(448, 448, 3)] * n

    for d in pipe.detectors:
        submit(d)
        d.collect_batch()
    t0 = time.perf_counter()
    for _ in range(reps):
        submit(det)
        res = det.collect_batch()
    dt = (time.perf_counter() - t0) / reps
    # two batches in flight
    a, b = pipe.detectors
    t0 = time.perf_counter()
    submit(a)
    for i in range(2 * reps - 1):
        nxt, cur = (b, a) if i % 2 == 0 else (a, b)
        submit(nxt)
        cur.collect_batch()
    (b if (2 * reps - 1) % 2 else a).collect_batch()
    dt2 = (time.perf_counter() - t0) / (2 * reps)
    fwd_ms = eng.forward_resident(5) / 5
    heads = {k: eng.read_blob(k).copy() for k in ("coverage", "bboxes")}
    pipe.close()
    return {"frames_per_s": round(n / dt2, 1), "ms_per_batch": round(dt2 * 1e3, 3), "batches_in_flight": 2,
            "one_batch_at_a_time": {"frames_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 3)}, "forward_ms": round(fwd_ms, 3),
            "forward_tflops": round(FWD_GFLOP * n / fwd_ms, 1), "dtype": dtype, "batch": n,
            "roofline": ({"bound": "hbm", "kernel": "conv_fwd (f16 storage, v_mfma_f32_32x32x16_f16): the whole forward, 27 launches",
                          "achieved": round((n * F16_ACT_MB + F16_WEIGHT_MB) / fwd_ms, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round((n * F16_ACT_MB + F16_WEIGHT_MB) / fwd_ms / HBM_PEAK_GBS, 4), "traffic": infer32_traffic(fwd_ms),
                          "note": "algorithmic bytes per batch = %d x %.1f MB of half activations (read + written once by every conv / pool / LRN) "
                                  "+ %.1f MB of half weights, over the forward's device time; SURVEY 8(d): this path is bytes-bound (ridge 400 flop/B vs 112)"
                                  % (n, F16_ACT_MB, F16_WEIGHT_MB)} if dtype == "f16" else None),
            "detections_last_batch": int(sum(len(r[0]) for r in res)),
            "candidates_per_image_class": {"mean": round(float((heads["coverage"] >= 0.5).sum(axis=(2, 3)).mean()), 1),
                                           "max": int((heads["coverage"] >= 0.5).sum(axis=(2, 3)).max())}}, heads


def bench_detector_stream(local: int, msg, params, n_frames: int = 300):
    """The whole node per camera frame at batch 1, PCIe included: 640x480 uint8 frame from host memory -> upload ->
    pre-processing -> forward -> decode + groupRectangles -> detections on the host; one frame at a time vs four in flight."""
    from fcn_object_detector_amd.detector import DetectorPipeline, FCNObjectDetector, HeadMapping
    from fcn_object_detector_amd.engine import Engine
    from fcn_object_detector_amd.netspec import NetSpec
    frames = [np.random.default_rng(i).integers(0, 256, (480, 640, 3), dtype=np.uint8) for i in range(8)]
    pipe = DetectorPipeline(lambda first: Engine(NetSpec(msg, "TEST"), params=params, device=local, tune_from=first, tune_max_lds_kb=36),
                            depth=4, mapping=HeadMapping.detectnet_deploy())
    lone = pipe.detectors[0]
    for f in frames:
        lone.run_detector(f)
    t0 = time.perf_counter()
    for i in range(n_frames):
        lone.run_detector(frames[i % 8])
    serial = n_frames / (time.perf_counter() - t0)
    pipe.run_detector_stream(frames)
    t0 = time.perf_counter()
    pipe.run_detector_stream(frames[i % 8] for i in range(n_frames))
    piped = n_frames / (time.perf_counter() - t0)
    pipe.close()
    return {"workload": "per camera frame (640x480 uint8 from host memory): upload + pre-processing + forward + decode/groupRectangles + "
                        "read-back, f32, batch 1", "frames_per_s_one_at_a_time": round(serial, 1),
            "frames_per_s_4_in_flight": round(piped, 1)}


def bench_vgg(steps: int = 5):
    """The reference's secondary net train/fcn_bbox (VGG16 + FCN-8s scores + x4 bilinear bbox branch):
    forward of its inference form at 448x448 and one training step at its native shape (288x288, stride 8, 11 classes,
    batch 24 as in the reference's param_str).  3x3 convolutions with large M: what the convolution kernels sustain
    when the launches are not latency-bound."""
    from fcn_object_detector_amd import lib as L, models, proto
    from fcn_object_detector_amd.engine import Engine
    from fcn_object_detector_amd.netspec import NetSpec, fill_params
    from fcn_object_detector_amd.train import SolverParams, TrainEngine

    def conv_flops(spec, shapes):
        f = 0.0
        for l in spec.layers:
            if l.type == "Convolution":
                n, co, oh, ow = shapes[l.tops[0]]
                k = int(l.sub("convolution_param").get("kernel_size"))
                f += 2.0 * n * co * oh * ow * shapes[l.bottoms[0]][1] * k * k
        return f

    msg = proto.parse_text(models.vgg16_fcn_bbox_deploy(1, 448, 448, 11))
    spec = NetSpec(msg, "TEST")
    shapes = spec.infer()
    eng = Engine(NetSpec(msg, "TEST"), params=fill_params(spec, seed=1), device=0)
    eng.host_array("data")[...] = np.random.default_rng(0).random((1, 3, 448, 448), dtype=np.float32)
    eng.upload_inputs()
    eng.forward_resident(3)
    ms = eng.forward_resident(20) / 20
    fl = conv_flops(spec, shapes)
    out = {"net": "train/fcn_bbox (VGG16-FCN), f32", "forward_448_b1": {"frames_per_s": round(1e3 / ms, 1), "ms_per_frame": round(ms, 3),
           "gflop_per_frame": round(fl / 1e9, 2), "achieved_tflops": round(fl / ms / 1e9, 1), "frac_of_f32_mfma_peak": round(fl / ms / 1e9 / F32_MFMA_PEAK_TFLOPS, 3)}}
    eng.close()
    # the same forward with three frames in flight (engine.ForwardPipeline; no LDS cap: these launches are large)
    from fcn_object_detector_amd.engine import ForwardPipeline
    params = fill_params(spec, seed=1)
    pipe = ForwardPipeline(lambda: NetSpec(msg, "TEST"), params=params, device=0, depth=3, max_lds_kb=None)
    for e in pipe.engines:
        e.host_array("data")[...] = np.random.default_rng(0).random((1, 3, 448, 448), dtype=np.float32)
        e.upload_inputs()
    pipe.run_resident(6)
    dt = pipe.run_resident(60) / 60
    out["forward_448_b1"]["frames_per_s_3_in_flight"] = round(1.0 / dt, 1)
    out["forward_448_b1"]["achieved_tflops_3_in_flight"] = round(fl / dt / 1e12, 1)
    pipe.close()
    n, size, classes = 24, 288, 11
    msg = proto.parse_text(models.vgg16_fcn_bbox_train("synthetic", "Boxes", "288,288,8,11,%d,none" % n, num_classes=classes))
    shapes = {"data": (n, 3, size, size), "label": (n, 1, size, size)}
    for k in ("bbox-label", "size-block", "obj-block", "coverage-block"):
        shapes[k] = (n, 4 * classes, size // 8, size // 8)
    spec = NetSpec(msg, "TRAIN")
    full = spec.infer(shapes)
    te = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params=fill_params(spec, seed=2), device=0,
                     solver=SolverParams(base_lr=1e-10, momentum=0.9, weight_decay=1e-7))
    rng = np.random.default_rng(1)
    te.host_array("data")[...] = rng.random((n, 3, size, size), dtype=np.float32)
    te.host_array("label")[...] = rng.integers(0, classes, (n, 1, size, size)).astype(np.float32)
    te.upload_inputs()
    fl = conv_flops(spec, full)
    for it in range(2):
        te.step(seed=it, upload=False)
    L.call("fcn_device_sync")
    t0 = time.perf_counter()
    for it in range(steps):
        te.step(seed=10 + it, upload=False)
    L.call("fcn_device_sync")
    dt = (time.perf_counter() - t0) / steps
    out["train_288_b24"] = {"imgs_per_s": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 2), "fwd_conv_gflop_per_step": round(fl / 1e9, 1),
                            "achieved_tflops": round(3 * fl / dt / 1e12, 1), "frac_of_f32_mfma_peak": round(3 * fl / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, 3),
                            "note": "fwd + dgrad + wgrad counted as 3x the forward convolution FLOPs (conv1_1 has no dgrad)"}
    te.close()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="frames per step (BASELINE config 2 uses 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement reported under 'train'")
    ap.add_argument("--train-batch", type=int, default=8, help="images per GPU per training step (BASELINE configs[2]/[3])")
    ap.add_argument("--train-steps", type=int, default=0, help="timed training steps (default: min(steps, 30))")
    ap.add_argument("--per-op", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--in-flight", type=int, default=4, help="frames in flight: replicas of the batch-1 engine on their own streams "
                                                             "(1 = one stream, launches strictly serial)")
    ap.add_argument("--repeats", type=int, default=5, help="the K-step timing is repeated this many times inside the run: `value` is the median, "
                                                           "the spread rides beside it (a 200-step window is 40 ms: one sample says little)")
    ap.add_argument("--no-io-region", action="store_true", help="experiments: skip SURVEY 8(d) config 2's timed region (the hipGraph with the H2D / D2H copy nodes); "
                    "`value` is then the kernels-only figure")
    ap.add_argument("--trace-clean", action="store_true", help="kernel-trace passes of tools/profile_bench.sh: whole forwards only - no per-launch "
                                                               "event timing loops, one repeat - so that every kernel's calls = frames x its launches per frame")
    ap.add_argument("--no-secondary", action="store_true", help="skip the VGG16-FCN (train/fcn_bbox) measurements reported under 'secondary'")
    ap.add_argument("--devices", default="", help="GPU id of every rank, comma separated (default 0..N-1; rehearsals on a one-GPU box: 0,0 with --no-train)")
    args = ap.parse_args()

    from fcn_object_detector_amd import dp      # (touches no GPU)
    if args.gpus > 1 and not dp.launched_as_rank():
        # `python bench.py --gpus N` by itself: start one fresh process per GPU BEFORE anything in this process touches the
        # GPU (a process that has initialised HIP is never re-executed or forked); this launcher only waits.  Rank 0 prints
        # the JSON line on the inherited stdout.  Under `python -m torch.distributed.run` the ranks already exist.
        devices = [int(d) for d in args.devices.split(",") if d != ""] or list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit("bench.py: --devices names %d GPUs for --gpus %d" % (len(devices), args.gpus))
        sys.exit(dp.spawn_ranks(__file__, sys.argv[1:], devices))
    rank, world, local = dp.env_rank(), dp.env_world_size(), dp.env_device()
    if args.devices and "FCN_DEVICE" not in os.environ:      # ranks started by another launcher (torch.distributed.run): --devices still maps rank -> GPU
        devs = [int(d) for d in args.devices.split(",") if d != ""]
        if len(devs) == world:
            local = devs[dp.env_local_rank()]
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    from fcn_object_detector_amd import lib as L, models, proto
    from fcn_object_detector_amd.engine import Engine, ForwardPipeline
    from fcn_object_detector_amd.netspec import NetSpec, fill_params
    cp = dp.ControlPlane(rank, world)

    msg = proto.parse_text(models.googlenet_detectnet_deploy(args.batch, 448, 448, 4))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=1234)
    depth = max(args.in_flight, 1)
    # (one frame in flight - the profiled single-stream mode - has no reason to cap the tiles' LDS footprint: the cap buys co-residency of
    #  DIFFERENT frames' workgroups on a CU)
    pipe = ForwardPipeline(lambda: NetSpec(msg, "TEST"), params=params, device=local, depth=depth, max_lds_kb=36 if depth > 1 else None)
    eng = pipe.engines[0]

    # synthetic frame: random uint8 BGR -> the node's demean/min-max normalisation (values in [0,1])
    frame = np.random.default_rng(rank).integers(0, 256, (args.batch, 448, 448, 3), dtype=np.uint8).astype(np.float32)
    mean = np.array([104.0069879317889, 116.66876761696767, 122.6789143406786], np.float32)
    x = frame - mean
    x = (x - x.min()) / (x.max() - x.min())
    x = np.ascontiguousarray(x.transpose(0, 3, 1, 2), dtype=np.float32)
    for e in pipe.engines:
        e.host_array("data")[...] = x
        e.upload_inputs()

    # single stream first: the latency of one frame and the per-launch kernel times behind `roofline`
    eng.forward_resident(max(args.warmup, 1))
    L.call("fcn_device_sync")
    t0 = time.perf_counter()
    dev_ms = eng.forward_resident(args.steps)
    L.call("fcn_device_sync")
    serial_s = time.perf_counter() - t0

    # kernels only, inputs resident in HBM: the same K batch-1 steps with `depth` frames in flight (replica engines on their own streams), repeated R times
    pipe.run_resident(max(args.warmup, 1))                        # W untimed warm-up steps (also captures the hipGraphs)
    depth = pipe.calibrate((depth - 1, depth)) if depth > 1 else 1    # untimed: 3 or 4 replicas, whichever packs better here
    if args.trace_clean:
        # (kernel-trace passes: one repeat, and no config-2 region - hipGraphLaunch between async copies on the same stream dies with SIGSEGV
        #  inside the runtime under rocprofv3 --kernel-trace after a few hundred frames, with one frame in flight as with four, with the
        #  copies inside the graph as outside it, never without the profiler: profiles/experiments/r04_graph_*segv*.txt)
        args.repeats = 1
        args.no_io_region = True
    rep_k = []
    for _ in range(max(args.repeats, 1)):
        L.call("fcn_device_sync")
        cp.barrier()
        t0 = time.perf_counter()
        pipe.run_resident(args.steps)                             # exactly K steps
        L.call("fcn_device_sync")
        t_local = time.perf_counter() - t0
        cp.barrier()
        rep_k.append(cp.max(t_local))
    t_k = float(np.median(rep_k))

    # headline (round 4): SURVEY 8(d) config 2's own timed region - H2D of the (3,448,448) f32 frame from pinned host
    # memory + layout change + all kernels + D2H of the two head blobs - with `depth` frames in flight: a frame's copies ride on its
    # replica's stream and overlap the other replicas' kernels.  Every rank runs it; K steps between barrier + device sync on both sides.
    rep_s = []
    if not args.no_io_region:
        pipe.warm_io(depth)                                       # untimed: captures every replica's graph with the copy nodes
        pipe.run_io(max(args.warmup, 1), depth)
        for _ in range(max(args.repeats, 1)):
            L.call("fcn_device_sync")
            cp.barrier()
            t0 = time.perf_counter()
            pipe.run_io(args.steps, depth)                        # exactly K steps
            L.call("fcn_device_sync")
            t_local = time.perf_counter() - t0
            cp.barrier()
            rep_s.append(cp.max(t_local))
    t_max = float(np.median(rep_s)) if rep_s else t_k

    out = None
    if rank == 0:
        ms_per_step = t_max * 1e3 / args.steps
        frames = args.steps * args.batch * world
        value = frames / t_max
        value_k = frames / t_k
        # the same region one frame at a time (the latency form of config 2)
        n_io = max(min(args.steps, 200), 10) if not args.trace_clean else 5
        io_one = [] if not args.no_io_region else [0.0]
        for _ in range(5 if not args.no_io_region else 0):
            eng.forward()
        for _ in range(max(args.repeats, 1) if not args.no_io_region else 0):
            t1 = time.perf_counter()
            for _ in range(n_io):
                eng.forward()
            io_one.append(n_io * args.batch / (time.perf_counter() - t1))
        pcie_fps = float(np.median(io_one))

        # dominant kernel family = the MFMA implicit-GEMM convolution: per-launch HIP-event timing on the engine's stream
        # `roofline.achieved`: HIP events on the engine's stream around each launch repeated back to back (20 reps: the launch floor and the
        # event cost amortise away).  rocprofv3's per-kernel durations of whole forwards are 1-4 % longer (in a forward the filters of a launch
        # are cold in L2, back to back they are warm): `from_profile` holds that figure, derived from the committed kernel trace alone by
        # tools/roofline_from_profile.py.  (Timing each launch once, in sequence, between two events - Engine.time_ops_in_sequence - reads
        # 2-4 us of event handling per launch on top: kept as `in_sequence_events` for reference, not used.)
        if args.trace_clean:      # (no event-timed launches in a clean trace: `roofline` of this run is a placeholder)
            ops = ops_seq = [(o.kind, o.name, 1e-6, o.flops, o.bytes) for o in eng.ops]
        else:
            ops = eng.time_ops(reps=20)
            ops_seq = eng.time_ops_in_sequence(reps=5)
        conv = [(k, n, ms, fl, by) for (k, n, ms, fl, by) in ops if k.startswith("conv")]
        conv_ms = sum(o[2] for o in conv)
        conv_flops = sum(o[3] for o in conv)
        conv_ms_seq = sum(o[2] for o in ops_seq if o[0].startswith("conv"))
        all_ms = sum(o[2] for o in ops)
        if args.per_op:
            for (k, n, ms, fl, by), w in zip(ops, ops_seq):
                sys.stderr.write("%-10s %-60s %8.2f us (in sequence, with event cost %7.2f) %7.2f TF/s %7.1f GB/s\n" % (k, n[:160], ms * 1e3, w[2] * 1e3, fl / ms / 1e9 if ms else 0,
                                                                                                                      by / ms / 1e6 if ms else 0))
        live = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms else 0.0
        tr = profile_traffic(PROFILE_TAG + "_bench")      # HBM bytes per conv launch from the committed PMC passes (stamped with the sources' hash)
        traffic = tr["bytes_per_launch"] if tr else None
        prof_roof = profile_json("bench_roofline")        # the family's time per frame from the committed kernel trace alone (tools/roofline_from_profile.py)
        if prof_roof is not None:
            prof_roof = {k: prof_roof[k] for k in ("frac", "achieved_tflops", "conv_us_per_frame", "mfma_busy_frac", "clock_ghz_assumed", "source", "stale") if k in prof_roof}
        live_d = {"achieved": round(live, 3), "frac": round(live / F32_MFMA_PEAK_TFLOPS, 4), "conv_us_per_frame": round(conv_ms * 1e3, 2),
                  "note": "HIP events on the engine's stream around each launch repeated back to back (20 reps): the launch's filters are warm in L2, "
                          "inside a forward they are cold - 1-4 % faster than the kernel trace of whole forwards"}
        if args.trace_clean:
            roofline = {"placeholder": True, "note": "--trace-clean: a kernel-trace pass times no launch by itself; the roofline of this command is derived "
                                                     "from the trace (tools/roofline_from_profile.py)"}
        else:
            fresh = prof_roof is not None and not prof_roof.get("stale", True)
            achieved = prof_roof["achieved_tflops"] if fresh else live
            roofline = {"bound": "mfma", "kernel": "conv_fwd_group + conv_first7 (f32 MFMA convolution family; %d launches covering the 59 convolutions)" % len(conv),
                        "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                        "source": ("profiles/%s_bench_kernel_stats.csv: rocprofv3 --kernel-trace --stats of this command with one frame in flight, the family's "
                                   "TotalDurationNs / frames (tools/roofline_from_profile.py)" % PROFILE_TAG) if fresh else
                                  "live back-to-back launches (the committed kernel trace is %s)" % ("stale: kernel sources changed since" if prof_roof else "missing"),
                        "traffic_source": tr, "algorithmic_bytes_per_launch": round(sum(o[4] for o in conv) / max(len(conv), 1)),
                        "traffic_note": "HBM bytes per conv launch = 2*FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this "
                                        "command; 'stale' = the kernel sources changed since that profile",
                        "frames_in_flight": 1,
                        "conv_us_per_frame": prof_roof["conv_us_per_frame"] if fresh else round(conv_ms * 1e3, 2),
                        "live_back_to_back": live_d,
                        "in_sequence_events": {"conv_us_per_frame": round(conv_ms_seq * 1e3, 2), "event_pair_floor_us": round(getattr(eng, "event_pair_floor_ms", 0.0) * 1e3, 2),
                                               "note": "each launch once, in the cache state of a forward, between two events: includes event handling "
                                                       "(an empty pair reads event_pair_floor_us); reference only"},
                        "from_profile": prof_roof,
                        "avg_launch_us": round(conv_ms * 1e3 / max(len(conv), 1), 2), "launches_per_step": len(ops),
                        "sum_kernel_ms_per_step": round(all_ms, 4),
                        "measured_on": "one stream; kernel durations are not comparable once frames overlap",
                        "whole_step_tflops": round(FWD_GFLOP * args.batch / (dev_ms / args.steps), 3)}
        region = ("SURVEY 8(d) config 2: H2D of the (3,448,448) f32 frame from pinned host memory + layout change + all kernels + D2H of the two head "
                  "blobs") if rep_s else "kernels only (--no-io-region)"
        rep_v = rep_s or rep_k
        out = {"metric": "frames/sec forward 448x448 @1 GPU; train imgs/sec @1/2/4/8 GPUs", "value": round(value, 2),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[1]: single-GPU forward, DetectNet GoogLeNet conv stack + coverage/bbox heads "
                                      "(graph of models/deploy.prototxt), batch=%d, 448x448, random-init weights" % args.batch,
                          "global_batch": args.batch * world, "parallelism": "replicas x%d" % world, "frames_in_flight": depth,
                          "timed_region": "`value`: %s, hipGraph replay, %d frames in flight (a frame's copies ride on its replica's stream and overlap "
                                          "the other replicas' kernels), median of %d repeats of K steps; `value_kernels_only`: the same K steps on inputs "
                                          "already resident in HBM (rounds 1-3's `value`); `config2_frames_per_s`: the region one frame at a time"
                                          % (region, depth, len(rep_v))},
               "repeats": {"R": len(rep_v), "frames_per_s": [round(frames / t, 1) for t in rep_v],
                           "min": round(frames / max(rep_v), 1), "max": round(frames / min(rep_v), 1),
                           "spread_pct": round(100.0 * (max(rep_v) - min(rep_v)) / t_max, 2)},
               "value_kernels_only": round(value_k, 2),
               "kernels_only": {"frames_per_s": round(value_k, 2), "ms_per_step": round(t_k * 1e3 / args.steps, 4), "frames_in_flight": depth,
                                "repeats": [round(frames / t, 1) for t in rep_k],
                                "includes": "all kernels of the forward, inputs resident in HBM, outputs left in HBM (no copies in the timed region)"},
               "config2_frames_per_s": round(pcie_fps, 2),
               "config2_frames_per_s_in_flight": round(value, 2) if rep_s else None,
               "config2_timed_region": {"frames_per_s": round(pcie_fps, 2), "ms_per_frame": round(1e3 / pcie_fps, 4) if pcie_fps else None,
                                        "frames_per_s_in_flight": round(value, 2) if rep_s else None, "frames_in_flight": depth,
                                        "repeats": {"one_at_a_time": [round(v, 1) for v in io_one]},
                                        "includes": "SURVEY 8(d) config 2: H2D of the (3,448,448) f32 frame (pinned host memory) + layout change + all kernels + "
                                                    "D2H of the two head blobs; one frame at a time here, with frames in flight it is `value`"},
               "single_stream": {"frames_per_s": round(args.steps * args.batch / serial_s, 2),
                                 "latency_ms_per_frame": round(serial_s * 1e3 / args.steps, 4),
                                 "device_ms_per_step": round(dev_ms / args.steps, 4)},
               "pcie_inclusive_fps": round(pcie_fps, 2),
               "roofline": roofline,
               "roofline_in_flight": {"bound": "mfma", "frames_in_flight": depth, "achieved": round(FWD_GFLOP * value_k / 1e3, 3), "peak": F32_MFMA_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": round(FWD_GFLOP * value_k / 1e3 / F32_MFMA_PEAK_TFLOPS, 4),
                                      "with_copies": {"achieved": round(FWD_GFLOP * value / 1e3, 3), "frac": round(FWD_GFLOP * value / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)},
                                      "note": "conv FLOPs of a frame x frames/s over the wall clock of the K steps with frames in flight - kernels only "
                                              "(`value_kernels_only`) and, under `with_copies`, config 2's region (`value`); kernels of different frames overlap, so "
                                              "per-kernel durations are not comparable: profiles/%s_bench_inflight_kernel_stats.csv holds the kernel trace of this mode" % PROFILE_TAG}}
        if world == 1 and not args.no_cpu_baseline:
            base, ref_blobs = cpu_baseline(msg, params, x)
            out["cpu_baseline"] = base
            res = eng.forward()
            errs = {k: float(np.abs(res[k] - ref_blobs[k]).max() / max(np.abs(ref_blobs[k]).max(), 1e-30)) for k in ("coverage", "bboxes")}
            out["parity_rel_err"] = {k: float("%.3e" % v) for k, v in errs.items()}
            if max(errs.values()) >= 1e-3:
                raise SystemExit("bench: GPU output differs from the oracle: %s" % errs)
            out["speedup_vs_cpu"] = round(value / base["value"], 1)      # (config 2's region with frames in flight against the CPU port's forward)
            out["speedup_vs_cpu_kernels_only"] = round(value_k / base["value"], 1)
    cp.barrier()
    pipe.close()
    if out is not None and world == 1 and not args.no_secondary:
        # measured right after the headline's replicas are gone: the runtime deals streams to its hardware queues in creation
        # order and replica streams that share a queue do not overlap (with 8 queues and other engines' streams alive: 2550
        # instead of 3450 frames/s; lib.load() asks for 16)
        out["detector_batch1"] = bench_detector_stream(local, msg, params)
    if not args.no_train:
        tsteps = args.train_steps or max(min(args.steps, 30), 1)
        watchdog = None
        if world > 1:
            # A rank that dies or stalls inside an RCCL call leaves the others blocked in the collective for good (the control
            # plane's sockets time out, ncclAllReduce does not).  The headline measured above must still be reported: after
            # $FCN_BENCH_TRAIN_TIMEOUT seconds rank 0 prints the line with the failure in `train` and every rank leaves.
            import threading
            limit = float(os.environ.get("FCN_BENCH_TRAIN_TIMEOUT", "240"))

            def give_up() -> None:
                if out is not None:
                    out["train"] = {"error": "training step did not finish within %.0f s on %d ranks (a rank died or an RCCL call hung)" % (limit, world)}
                    print(json.dumps(out), flush=True)
                os._exit(4 if out is not None else 3)      # (a hung collective is a failure on every rank, rank 0 included)
            watchdog = threading.Timer(limit, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            tr = bench_train(cp, rank, world, local, args.train_batch, tsteps, max(min(args.warmup, 5), 2), trace_clean=args.trace_clean)
        except Exception as e:      # the headline measured above must survive a failure in the secondary measurement
            tr = {"error": "%s: %s" % (type(e).__name__, e)}
        if watchdog is not None:
            watchdog.cancel()
        if out is not None:
            out["train"] = tr
    if out is not None and world == 1 and not args.no_secondary:
        # the same DetectNet forward at batch 8 (not the headline configuration): M is 8x larger, launches stop being latency-bound
        msg8 = proto.parse_text(models.googlenet_detectnet_deploy(8, 448, 448, 4))
        spec8 = NetSpec(msg8, "TEST")
        spec8.infer()
        eng8 = Engine(NetSpec(msg8, "TEST"), params=fill_params(spec8, seed=1234), device=local)
        eng8.host_array("data")[...] = np.random.default_rng(3).random((8, 3, 448, 448), dtype=np.float32)
        eng8.upload_inputs()
        eng8.forward_resident(3)
        ms8 = eng8.forward_resident(20) / 20
        ops8 = [o for o in eng8.time_ops(reps=5) if o[0].startswith("conv")]
        conv8 = sum(o[3] for o in ops8) / (sum(o[2] for o in ops8) * 1e-3) / 1e12
        out["forward_batch8"] = {"frames_per_s": round(8e3 / ms8, 1), "ms_per_step": round(ms8, 4), "conv_family_tflops": round(conv8, 2),
                                 "conv_family_frac_of_f32_mfma_peak": round(conv8 / F32_MFMA_PEAK_TFLOPS, 4)}
        eng8.close()
        r32, h32 = bench_infer32(local, "f32")
        r16, h16 = bench_infer32(local, "f16")
        r16["rel_err_vs_f32"] = {k: float("%.3e" % (np.abs(h16[k] - h32[k]).max() / max(np.abs(h32[k]).max(), 1e-30))) for k in h32}
        out["inference_batch32"] = {"workload": "configs[4]: batch 32, 448x448, pre-processing + forward + fused decode/groupRectangles + read-back",
                                    "f16": r16, "f32": r32}
        out["secondary"] = bench_vgg()
    cp.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
