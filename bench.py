#!/usr/bin/env python
"""Benchmark of the hot path on MI355X (see the contract in DESIGN.md "Measurement").

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the whole hot path (pyramid -> Hessian/NMS detector -> AffNet -> filter ->
OriNet -> level select -> HardNet) over one batch of 64 synthetic 1024x768 images, 2000 keypoints
each (BASELINE.json configs[2], the configuration the metric is quoted on), per rank (weak scaling),
images resident in HBM before the timed region, processed as 4 fused library calls of 16 images (every
kernel launch covers 16 images), followed for N > 1 by the all_gather of the padded
(count, LAFs, responses, descriptors) records.  value = keypoints returned by all ranks / max-over-ranks time.

roofline   : dominant kernel = fused HardNet trunk (cnn32_trunk_kernel<2>, fp32 MFMA).  achieved =
             algorithmic FLOPs per launch / mean launch duration measured with HIP events around that
             launch on its own stream inside the timed region (affnet_profile_*).
cpu_baseline: the CPU oracle (port of the reference, same torch CPU operators) on this host's cores
             on a bounded sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, NKP, BATCH = 768, 1024, 2000, 64
# algorithmic work: 2*MAC per patch, dense, BN/ReLU/normalisation excluded
FLOP_AFF, FLOP_ORI, FLOP_HARD = 19193856.0, 19316736.0, 78184448.0
FLOP_HARD_HEAD = 2.0 * 8192 * 128
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD


def pmc_traffic(images_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    in separate passes, tools/gpu_full.sh + tools/pmc_traffic.py; counters cannot be read from inside this process)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    k = d["kernels"].get("void cnn32_trunk_kernel<2, 8>")
    if not k:
        return None, None
    scale = images_per_launch / float(d["images_per_launch"])
    note = ("%s: 2 x FETCH_SIZE (gfx950 correction for wide reads; the sampler's narrow gathers are uncalibrated, raw = %.3g B) "
            "+ WRITE_SIZE, scaled to %d images per launch" % (os.path.basename(files[-1]), k["fetch_bytes_raw"] * scale, images_per_launch))
    return k["hbm_bytes"] * scale, note


def cpu_baseline(n_timed=2):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import affnet_oracle as orc
    # torch's default intra-op thread count (respects the container's CPU affinity / quota)
    sd = {k: torch.load(os.path.join(ROOT, "pretrained", k + ".pth"), map_location="cpu", weights_only=False)["state_dict"]
          for k in ("AffNet", "OriNet")}
    hard = orc.synthetic_hardnet_state(0)
    kp, t = 0, 0.0
    for i in range(n_timed + 1):
        x = orc.synthetic_image(H, W, i)
        ex = orc.OracleExtractor(mrSize=5.192, num_features=NKP, border=5, num_Baum_iters=1, affnet_sd=sd["AffNet"],
                                 orinet_sd=sd["OriNet"], reproduce_wasted_extraction=True)
        t0 = time.perf_counter()
        L, r, P, D = orc.describe(x, ex, hard, do_ori=True, ps=32)
        dt = time.perf_counter() - t0
        if i > 0:                       # first image = warm-up
            kp += L.shape[0]
            t += dt
    return {"value": kp / t, "unit": "keypoints/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d synthetic 1024x768 images x 2000 kp (seeds 1..%d) after 1 warm-up, %.1f s of CPU work; "
                      "oracle/affnet_oracle.py = the reference's torch-CPU operator sequence incl. its discarded extra extraction"
                      % (n_timed, n_timed, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="images per step per rank")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("AFFNET_BENCH_CHUNK", "32")),
                    help="images per fused library call (every kernel launch covers `chunk` images)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("AFFNET_BENCH_STREAMS", "1")),
                    help="independent streams (each with its own context) the chunks alternate over")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("AFFNET_BENCH_PIPELINE", "0")),
                    help="1 (with --streams 1): pyramid + detector of chunk i+1 run on a second stream next to the CNN stages of "
                         "chunk i (two contexts alternate); the CNN kernels stay serialised on one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE.json configs[4] instead of the headline configs[2]: 3840x2160 images, 8000 kp each (deep pyramid stress); "
                         "batch / chunk default to 8 / 8")
    args = ap.parse_args()
    global H, W, NKP
    if args.config5:
        H, W, NKP = 2160, 3840, 8000
        if args.batch == BATCH:
            args.batch = 8
        if args.chunk == 32:
            args.chunk = 8

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if os.environ.get("AFFNET_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # AFFNET_BENCH_SELF_GATHER=1 (single process): a 1-rank RCCL group, so that the stream-ordered all-gather of the N > 1
    # path runs on a 1-GPU box (tools/gpu_dist_dryrun.sh); AFFNET_BENCH_BACKEND=gloo + AFFNET_BENCH_ONE_DEVICE=1: 2 ranks on one GPU
    SELF = world == 1 and bool(os.environ.get("AFFNET_BENCH_SELF_GATHER"))
    DIST = world > 1 or SELF
    if DIST:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("AFFNET_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    import affnet_amd
    from affnet_amd import _lib, sharded
    from affnet_amd.synthetic import synthetic_image

    def load(name, cls):
        net = cls(PS=32) if name != "HardNet" else cls()
        if name != "HardNet":
            net.load_state_dict(torch.load(os.path.join(ROOT, "pretrained", name + ".pth"), map_location="cpu", weights_only=False)["state_dict"])
        else:
            net.load_state_dict(affnet_amd.synthetic_hardnet_state(0))
        return net.to(dev)

    A, O, Hn = load("AffNet", affnet_amd.AffNetFast), load("OriNet", affnet_amd.OriNetFast), load("HardNet", affnet_amd.HardNet)
    # global image i of a step lives on rank i % world (weak scaling: `batch` images per rank per step)
    seeds = [rank + world * j for j in range(args.batch)]
    CH = max(1, min(args.chunk, args.batch))
    imgs = torch.cat([synthetic_image(H, W, s) for s in seeds], 0).to(dev)           # (batch,1,H,W) resident in HBM
    chunks = [imgs[i:i + CH] for i in range(0, args.batch, CH)]
    S = max(1, args.streams)
    PIPE = bool(args.pipeline) and S == 1
    if PIPE:
        S = 2                          # two contexts alternate; ONE CNN stream + ONE detector stream
    # one extractor (context + workspace) per (stream, chunk size); a ragged last chunk gets its own
    dets = {}
    for ci, c in enumerate(chunks):
        k = (ci % S, c.size(0))
        if k not in dets:
            dets[k] = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=NKP, border=5, num_Baum_iters=1,
                                                                AffNet=A, OriNet=O).to(dev)
            dets[k]._context(c, allow_batch=True)  # create contexts / workspaces before anything is timed
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    det_stream = torch.cuda.Stream(device=dev) if PIPE else None

    pending = [None]                   # finish() of the previous step's all-gather (overlaps with this step's compute)
    # Steps are enqueued back to back; nothing inside the timed region waits on the host (the keypoint counts are
    # summed on the device and read once after the closing synchronize, the RCCL gather of step k is ordered behind
    # step k's kernels by stream semantics and waited for, on the stream, in step k+1).  A host-side sync + count
    # read-back per step left the GPU idle ~2.4 ms per 135 ms step while the host launched the next step's detector.
    LAZY = not PIPE and S == 1 and not (DIST and dist.get_backend() != "nccl")
    kp_dev = torch.zeros((), dtype=torch.int64, device=dev)
    inflight = []

    def step():
        results = [None] * len(chunks)
        for ci, c in enumerate(chunks):
            with torch.cuda.stream(streams[0] if PIPE else streams[ci % S]):
                results[ci] = dets[(ci % S, c.size(0))].enqueue(c, do_ori=True, desc=Hn, det_stream=det_stream, input_ready=False)
        if LAZY:
            with torch.cuda.stream(streams[0]):
                if len(inflight) >= 2:
                    inflight.pop(0).synchronize()               # host stays at most two steps ahead of the GPU (bounds memory)
                kp_dev.add_(torch.stack([r["count"].sum() for r in results]).sum())
                if DIST:
                    if pending[0] is not None:
                        pending[0]()                            # stream-side wait for the previous step's gather
                    pending[0] = sharded.gather_features_async(sharded.pack_batched_records(results, NKP), args.batch * world, force=SELF)
                ev = torch.cuda.Event()
                ev.record()
                inflight.append(ev)
            return results
        for s in streams:
            s.synchronize()
        kp_dev.add_(sum(int(r["count"].sum().item()) for r in results))
        if DIST:
            if pending[0] is not None:
                pending[0]()                                    # records of the previous step have arrived
            rec = sharded.pack_batched_records(results, NKP)
            torch.cuda.synchronize()
            pending[0] = sharded.gather_features_async(rec, args.batch * world, force=SELF)
        return results

    def drain():
        if pending[0] is not None:
            with torch.cuda.stream(streams[0]):
                pending[0]()
            pending[0] = None
        torch.cuda.synchronize()

    def barrier():
        if DIST:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    for d in dets.values():
        _lib.check(_lib.lib.affnet_profile_enable(d._ctx.handle, 1), d._ctx.handle, "profile_enable")
    kp_dev.zero_()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                            # every step's kernels and the last gather complete inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    kp = int(kp_dev.item())
    # stage timings recorded by HIP events on the launch streams during the timed region
    sums, calls, call_imgs = [0.0] * 8, 0, 0
    for (_, nimg), d in dets.items():
        buf, n = (C.c_double * 8)(), C.c_int32(0)
        _lib.check(_lib.lib.affnet_profile_read(d._ctx.handle, C.byref(buf), C.byref(n)), d._ctx.handle, "profile_read")
        calls += n.value
        call_imgs += n.value * nimg
        sums = [a + b for a, b in zip(sums, list(buf))]
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    kp_all = torch.tensor([kp], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(kp_all, op=dist.ReduceOp.SUM)
    if rank == 0:
        tmax, kps = float(t_all.item()), float(kp_all.item())
        stage_ms = [s / max(call_imgs, 1) for s in sums]           # per image (rank 0's launches)
        names = ["pyramid", "detector", "affnet", "shape_filter", "orinet", "denorm_levelsel", "hardnet_trunk", "hardnet_head"]
        img_per_launch = call_imgs / max(calls, 1)
        trunk_ms = sums[6] / max(calls, 1)                          # mean duration of one HardNet trunk launch
        kp_per_img = kps / max(1, args.steps * args.batch * world)
        flops_launch = kp_per_img * img_per_launch * (FLOP_HARD - FLOP_HARD_HEAD)
        achieved = flops_launch / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
        traffic, traffic_note = pmc_traffic(img_per_launch)
        out = {
            "metric": "keypoints/sec (detect+AffNet+OriNet+HardNet) per image, %d kp @%dx%d" % (NKP, W, H),
            "value": kps / tmax, "unit": "keypoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmax / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: batch of %d synthetic %dx%d grayscale images per GPU per step, "
                                   "%d kp each, full path detect+AffNet+OriNet+HardNet; AffNet/OriNet shipped weights, "
                                   "HardNet seeded synthetic weights (HardNet++.pth is a missing blob)" % (4 if args.config5 else 2, args.batch, W, H, NKP),
                       "global_batch": args.batch * world, "keypoints_per_image": kp_per_img,
                       "images_per_launch": CH,
                       "streams_per_gpu": "1 CNN stream + 1 detector stream (2 contexts alternate)" if PIPE else S,
                       "parallelism": "image-per-GPU x%d, all_gather of padded records" % world if world > 1 else "1 GPU"},
            "ms_per_image": tmax / (args.steps * args.batch) * 1e3,
            "stage_ms_per_image": dict(zip(names, [round(v, 4) for v in stage_ms])),
            "roofline": {"kernel": "cnn32_trunk_kernel<HardNet> (fp32 MFMA 16x16x4, fused sampler+norm+6 convs)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_note, "flops_per_launch": flops_launch, "launch_ms": trunk_ms,
                         "all_cnn_tflops": kp_per_img * (1.5 * FLOP_AFF + FLOP_ORI + FLOP_HARD) /
                                           (max(stage_ms[2] + stage_ms[4] + stage_ms[6] + stage_ms[7], 1e-9) * 1e-3) / 1e12},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if DIST:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
