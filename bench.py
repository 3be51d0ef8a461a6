#!/usr/bin/env python3
"""Headline benchmark: decoded meshes/s (+ GB/s) on a batch of 64k-triangle Edgebreaker .drc.

A "step" is one device-resident decode of the whole per-GPU batch (compressed bytes already
in HBM -> faces, attribute values and point maps in HBM).  One process per GPU; meshes are
independent, so there is no data-path collective: torch.distributed (RCCL) carries only the
barrier and the MAX / SUM of the timing.  Two partitions of the work are measured:

  weak    every rank decodes its own batch of --meshes streams (seeds 1000 + rank * meshes ...): BASELINE.json
          configs[2] per GPU.  With one GPU this is the line's `value`.
  strong  BASELINE.json configs[3]: the SAME --meshes streams (seeds 1000 ...) split over the N ranks by
          sharding.balanced_assignment (longest compressed stream first).  With several GPUs THIS is the line's `value`
          ("scaling": "strong"; the weak leg is reported beside it as "weak_scaling").  `--scaling weak|strong|both` override.

Beside the line's value: `end_to_end` (host .drc bytes -> host arrays: pinned staging, one upload and one download per batch,
two batches in flight), `pool` (the in-library work queue over all GPUs of the job, from rank 0), `encode` (configs[4]),
`cpu_baseline` (the oracle on the host cores), `oracle_check` (decoded meshes against the oracle, after the timed region).

The timed region of each leg is bracketed by a barrier + synchronize, the MAX over ranks is reported.  Rank 0 prints
one JSON line.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TRAFFIC_FILE = os.path.join("profiles", "r04_traffic.json")


def usable_cores():
    """Cores this process may keep busy: its affinity mask, cut to the container's CPU quota (cgroup cpu.max) -- threads beyond
    the quota are not slower by their share, they are stopped for the rest of every period."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = max(2, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(blob, offsets, threads, budget_s, max_meshes):
    """The CPU oracle (C++ scalar restatement of the reference path) on a bounded sample of the rank-0 batch, on
    `threads` host threads (the ctypes call releases the GIL; every mesh is independent)."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    n = min(len(offsets) - 1, max_meshes)
    streams = [bytes(blob[int(offsets[i]):int(offsets[i + 1])]) for i in range(n)]
    oracle.decode(streams[0])
    done, t0 = 0, time.perf_counter()
    if threads == 1:
        for s in streams:
            oracle.decode(s)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
    else:
        chunk = 4 * threads
        with ThreadPoolExecutor(threads) as ex:
            while done < n and time.perf_counter() - t0 <= budget_s:
                part = streams[done:done + chunk]
                list(ex.map(oracle.decode, part))
                done += len(part)
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "meshes/s", "cores": threads, "kind": "port",
            "sample": "first %d meshes of the rank-0 batch, full decode incl. numpy export, %.1f s" % (done, dt),
            "host_cores_available": usable_cores()}


def measured_traffic(kernel, meshes, triangles):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on
    this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); None if the committed
    measurement is for another workload or predates the kernels.  PMC counters cannot be collected from inside the process."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            t = json.load(f)
        if t["meshes_per_gpu"] == meshes and t["triangles_per_mesh"] == triangles and kernel in t["kernels"]:
            k = t["kernels"][kernel]
            # every launch of the kernel in one decode (the symbol kernels are launched twice: early and late attributes), to go
            # with a duration that spans them
            return k["hbm_bytes"] * k.get("launches_per_decode", 1.0), TRAFFIC_FILE, t.get("total_hbm_bytes")
    except (OSError, KeyError, ValueError):
        pass
    return None, None, None


def encode_leg(dsa, synth, ctx, nx, ny, count, comm=None, barrier=None, world=1):
    """BASELINE.json configs[4] in the same line: `count` raw 64k-triangle meshes -> .drc through dsa_encode_batch
    (corner table, Edgebreaker symbols and traversal order by k_enc_connectivity, quantise + predict + rANS code by the
    attribute kernels, symbol-scheme choice and stream layout on the host cores), decoded again and compared.  The batch
    cycles through 32 distinct meshes (generating more in Python would take longer than the leg)."""
    distinct = [synth.make_mesh(synth.GRID, nx, ny, 1000 + i) for i in range(min(32, count))]
    meshes = []
    for i in range(count):
        pos, nrm, uv, faces = distinct[i % len(distinct)]
        meshes.append(dsa.MeshData(pos, faces, nrm, uv))
    enc = dsa.DracoEncoder(ctx)
    enc.EncodeBatch(meshes)                  # untimed: the encoder's lanes pin their staging buffers and size their device memory once
    if barrier is not None:
        barrier()
    t0 = time.perf_counter()
    out = enc.EncodeBatch(meshes)
    dt = time.perf_counter() - t0
    if comm is not None:                     # every rank codes `count` meshes of its own (weak): the job's time is the slowest rank's
        dt = comm.max(dt)
    checks = [synth.encode_mesh(*(lambda m: (m[0], m[3], m[1], m[2]))(distinct[k])) for k in range(min(4, len(distinct)))]
    identical = all(out[k] == checks[k] for k in range(len(checks)))
    b = dsa.Batch(ctx, out)
    b.decode()
    ok = identical and all(b.status(i) == 0 for i in range(count))
    b.close()
    ok = ok if comm is None else comm.sum(0 if ok else 1) == 0
    return {"meshes_per_s": world * count / dt, "ms": dt * 1e3, "meshes": world * count, "meshes_per_gpu": count, "n_gpus": world, "bytes_per_mesh": sum(map(len, out)) // count,
            "config": "%d x %d-triangle meshes, positions 11b + normals 8b + UVs 10b, end to end (device Edgebreaker + HIP attribute kernels, host stream layout)" % (count, 2 * nx * ny),
            "round_trip_ok": bool(ok), "byte_identical_to_cpu_coder": bool(identical)}


def end_to_end_leg(dsa, ctx, blob, offsets, batches, in_flight, comm, barrier, world, compact=True):
    """What a draco-sharp caller sees (DracoDecoder.cs:19-42 takes host bytes and returns host objects): `batches` batches of
    this rank's streams, each host bytes -> dsa_batch_create (pinned staging, upload on the copy stream) -> dsa_batch_decode ->
    dsa_batch_download (ONE transfer of faces + values + point maps into a pinned mirror) -> dsa_batch_wait, `in_flight` batches
    queued at a time so that upload(k+1), the kernels of k and the download of k-1 overlap.  The clock runs from the first byte
    handed over to the last array readable on the host.  h2d_s / d2h_s: one upload / one download alone, for the link rates."""
    import numpy as np
    n = len(offsets) - 1

    def submit():
        b = dsa.Batch(ctx, blob=blob, offsets=offsets)
        b.decode(wait=False)
        b.download(wait=False, compact=compact)
        return b

    def finish(b):
        b.wait()
        v0, v1 = b.host_views(0), b.host_views(n - 1)        # views into the mirror: the arrays are on the host
        pm = v1["attributes"][0]["point_map"]
        tag = int(v0["faces"][0, 0]) + int(v1["faces"][-1, -1]) + (int(pm[-1]) if pm is not None else 0)
        b.close()
        return tag

    # warm-up: the context's caches (arenas, pinned mirrors, staging) fill; also the stand-alone link times
    t0 = time.perf_counter(); b = dsa.Batch(ctx, blob=blob, offsets=offsets); b.decode(wait=True); t_up_decode = time.perf_counter() - t0
    out_bytes = b.compact_bytes if compact else b.output_bytes
    b.download(wait=True, compact=compact)                   # first use pins the mirror
    t0 = time.perf_counter(); b.decode(wait=True); t_dec = time.perf_counter() - t0
    t0 = time.perf_counter(); b.download(wait=True, compact=compact); t_d2h = time.perf_counter() - t0
    b.close()
    live = [submit() for _ in range(in_flight)]
    for b in live:
        finish(b)
    t0 = time.perf_counter(); b = dsa.Batch(ctx, blob=blob, offsets=offsets); b.decode(wait=True); t_h2d = max(1e-6, time.perf_counter() - t0 - t_dec)
    b.close()
    # two timed passes, the faster one reported (both in `seconds_passes`): pinning a mirror for the first time, or a host that is
    # compacting memory, can take a second on a fresh box -- that is the box's start-up, not the pipeline's rate
    passes = []
    for _ in range(2):
        if barrier is not None:
            barrier()
        t0 = time.perf_counter()
        live, tags = [], 0
        for k in range(batches):
            live.append(submit())
            if len(live) == in_flight:
                tags += finish(live.pop(0))
        while live:
            tags += finish(live.pop(0))
        dtp = time.perf_counter() - t0
        if comm is not None:
            dtp = comm.max(dtp)
        passes.append(dtp)
    dt = min(passes)
    return {"value": world * batches * n / dt, "unit": "meshes/s", "seconds": dt, "batches_per_gpu": batches, "meshes_per_batch": n, "in_flight": in_flight,
            "n_gpus": world, "seconds_passes": passes, "host_bytes_in_per_batch": int(offsets[-1]), "host_bytes_out_per_batch": out_bytes,
            "gb_per_s_out": world * batches * out_bytes / dt / 1e9,
            "h2d_s": t_h2d, "h2d_gb_per_s": int(offsets[-1]) / t_h2d / 1e9, "d2h_s": t_d2h, "d2h_gb_per_s": out_bytes / t_d2h / 1e9,
            "decode_s": t_dec, "first_batch_s_cold": t_up_decode,
            "layout": "compact (uint16 faces, one point map per distinct map: dsa_batch_download_compact)" if compact else "full (int32 faces, a point map per attribute)",
            "what": "host .drc bytes -> host arrays (faces, attribute values, point maps of every mesh), pinned staging + one download per batch, %d batches in flight" % in_flight}


def pool_leg(dsa, devices, blob, offsets, chunk, repeats):
    """The north star's per-GPU work queues inside one process (dsa_pool_*): the job's streams, longest first, in chunks of `chunk`
    pulled by one worker thread per device from one atomic counter; device-resident results (no download)."""
    pool = dsa.Pool(devices, chunk_meshes=chunk)
    n = len(offsets) - 1
    job = pool.decode(blob=blob, offsets=offsets)            # warm: arenas, staging
    bad = sum(1 for i in range(0, n, max(1, n // 64)) if job.status(i) != 0)
    per_worker = [0] * len(devices)
    for i in range(n):
        per_worker[job.worker(i)] += 1
    job.close()
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        job = pool.decode(blob=blob, offsets=offsets)
        dt = time.perf_counter() - t0
        job.close()
        best = dt if best is None else min(best, dt)
    pool.close()
    return {"value": n / best, "unit": "meshes/s", "seconds": best, "meshes_in_job": n, "devices": list(devices), "chunk_meshes": chunk,
            "meshes_per_worker_first_run": per_worker, "failed_sampled": bad,
            "what": "dsa_pool_decode from host bytes (parse + pinned staging + upload + decode per chunk, two chunks in flight per device), results device-resident"}


def sustained_leg(dsa, ctx, blob, offsets, steps, warmup, comm, barrier, world):
    """Two device-resident batches of the same streams decoded in turn without waiting in between (dsa_batch_decode is
    asynchronous and the context's two stream sets take the decodes alternately): the next batch's chain and entropy decode start
    beside the previous batch's issue-light tail.  `steps` decodes, the clock stops when the last one is done; every decode is a
    whole batch, so meshes/s = steps * meshes / time."""
    pair = [dsa.Batch(ctx, blob=blob, offsets=offsets) for _ in range(2)]
    n = pair[0].n
    for k in range(max(2, warmup)):
        pair[k % 2].decode(wait=False)
    for b in pair:
        b.wait()
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        pair[k % 2].decode(wait=False)
    for b in pair:
        b.wait()
    barrier()
    dt = time.perf_counter() - t0
    bad = sum(1 for b in pair for i in range(0, n, max(1, n // 256)) if b.status(i) != 0)
    for b in pair:
        b.close()
    if comm is not None:
        dt = comm.max(dt)
    return {"value": world * steps * n / dt, "unit": "meshes/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "batches_in_flight": 2, "failed_sampled": bad,
            "what": "the same device-resident decode as `value`, two batches in flight on the context's two stream sets (no wait between steps)"}


def dialects_leg(dsa, synth, ctx, nx, ny, n, steps):
    """The same batch size in the dialects stock encoders write, device-resident like `value`: texture coordinates given per corner
    (three UV charts: attribute seams, an attribute corner table, a corner attribute) and the prediction schemes of the default
    encoder settings (TexCoordsPortable, GeometricNormal, valence-coded connectivity).  The batch cycles through 32 distinct
    meshes (the writer is the CPU coder driven from Python); three meshes of every variant are compared with the oracle."""
    import numpy as np
    import oracle
    from meshutil import seamed_mesh
    variants = [("uv_seams", (None, "stripes"), dict()),
                ("uv_seams_texcoords_portable", (None, "stripes"), dict(uv_prediction=5)),
                ("stock_default_per_vertex", None, dict(uv_prediction=5, predictive_connectivity=2, normal_prediction=6)),
                ("stock_default_uv_seams", (None, "stripes"), dict(uv_prediction=5, predictive_connectivity=2, normal_prediction=6)),
                # compression level 9: positions by ConstrainedMultiParallelogram
                ("stock_level9_per_vertex", None, dict(pos_prediction=4, uv_prediction=5, predictive_connectivity=2, normal_prediction=6))]
    out = {}
    for name, charts, opt in variants:
        distinct = []
        for k in range(32):
            if charts is None:
                pos, nrm, uv, faces = synth.make_mesh(synth.GRID, nx, ny, 1000 + k)
                distinct.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)))
            else:
                distinct.append(synth.encode_mesh_corners(*seamed_mesh(synth, synth.GRID, nx, ny, 1000 + k, *charts), opt=synth.options(**opt)))
        streams = [distinct[i % 32] for i in range(n)]
        b = dsa.Batch(ctx, streams)
        b.decode()
        b.decode()
        t0 = time.perf_counter()
        for _ in range(steps):
            b.decode()
        ms = (time.perf_counter() - t0) / steps * 1e3
        failed = sum(1 for i in range(n) if b.status(i) != 0)
        paths = sorted(set(int(b.mesh_info(i).decode_path) for i in range(0, n, 61)))
        equal = True
        for i in (0, 1, 31):
            ref = oracle.decode(streams[i])
            m = b.result(i).ConnectedData
            ok = np.array_equal(m.Faces, ref.faces) and len(m.Attributes) == len(ref.attributes)
            for a, r in zip(m.Attributes, ref.attributes):
                ok = ok and np.array_equal(a.PortableValues, r.portable) and np.array_equal(a.PointMap, r.point_map) and a.Values.tobytes() == r.values.tobytes()
            equal = equal and ok
        out[name] = {"meshes_per_s": n / ms * 1e3, "ms_per_step": ms, "decode_paths": paths, "failed": failed, "equal_to_oracle": bool(equal),
                     "bytes_per_mesh": len(distinct[0]), "kernels_ms": {k: round(v, 2) for k, v in b.kernel_times().items()}}
        # the same with two batches in flight (as `sustained` for the bench batch): these dialects end on chains of a few waves
        # (the texture-coordinate predictor) that leave the machine to the next batch's start
        b2 = dsa.Batch(ctx, streams)
        pair = [b, b2]
        for k in range(2):
            pair[k].decode(wait=False)
        for x in pair:
            x.wait()
        t0 = time.perf_counter()
        for k in range(2 * steps):
            pair[k % 2].decode(wait=False)
        for x in pair:
            x.wait()
        ms2 = (time.perf_counter() - t0) / (2 * steps) * 1e3
        out[name]["sustained_meshes_per_s"] = n / ms2 * 1e3
        out[name]["sustained_ms_per_step"] = ms2
        out[name]["sustained_failed_sampled"] = sum(1 for x in pair for i in range(0, n, 16) if x.status(i) != 0)
        b2.close()
        b.close()
        ctx.trim()
    return out


def oracle_check(batch, blob, offsets, indices):
    """Outside the timed region: the decoded results of `indices` equal the CPU oracle's (faces, portable integers,
    point maps, floats bit for bit)."""
    import numpy as np
    import oracle
    bad = []
    for i in indices:
        ref = oracle.decode(bytes(blob[int(offsets[i]):int(offsets[i + 1])]))
        m = batch.result(i).ConnectedData
        ok = np.array_equal(m.Faces, ref.faces) and len(m.Attributes) == len(ref.attributes)
        for a, r in zip(m.Attributes, ref.attributes):
            ok = ok and np.array_equal(a.PortableValues, r.portable) and np.array_equal(a.PointMap, r.point_map) and a.Values.tobytes() == r.values.tobytes()
        if not ok:
            bad.append(int(i))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--meshes", type=int, default=4096, help="meshes per GPU (weak) / in the job (strong); BASELINE.json configs[2]: 4096")
    ap.add_argument("--grid", type=int, nargs=2, default=[128, 256], help="grid cells (128x256 -> 65 536 triangles)")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong", "both"], default="auto",
                    help="auto: both partitions are timed; with one GPU they are the same batch, with several the STRONG one (BASELINE.json "
                         "configs[3]: the same 4096 streams sharded over the GPUs) is the line's value and the weak one is reported beside it; "
                         "both: like auto but weak stays the line; weak / strong: only that leg")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-pool", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--no-dialects", action="store_true")
    ap.add_argument("--e2e-batches", type=int, default=6)
    ap.add_argument("--encode-meshes", type=int, default=4096, help="meshes per GPU of the encode leg (BASELINE.json configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--check", type=int, default=64, help="meshes of the rank-0 batch compared with the oracle after the timed region")
    args = ap.parse_args()

    libs = [os.path.join(ROOT, *f) for f in (("draco-sharp_amd", "csrc", "libdraco_mi355x.so"), ("draco-sharp_amd", "synth", "libdsa_synth.so"),
                                             ("oracle", "liboracle.so"))]
    if not all(os.path.exists(f) for f in libs):
        # a fresh checkout: local rank 0 builds in-tree (the harness, not the product, does this) and signals the others
        # through a marker written after the last library is complete
        marker = os.path.join(ROOT, "draco-sharp_amd", "csrc", ".built_by_bench")
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
            with open(marker, "w") as f:
                f.write("ok\n")
        else:
            deadline = time.time() + 900
            while not os.path.exists(marker) and time.time() < deadline:
                time.sleep(1.0)
    import numpy as np
    import torch
    import draco_sharp_amd as dsa
    import draco_sharp_amd.synth as synth
    from draco_sharp_amd.sharding import Comm, balanced_assignment

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    # DSA_BENCH_REHEARSE=1 (diagnostics): every rank on GPU 0 and the reductions over gloo, to walk the N > 1 code path on a
    # one-GPU box; the numbers of such a run mean nothing
    rehearse = os.environ.get("DSA_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm = Comm(backend="gloo") if rehearse else Comm(backend="nccl", device=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm; only barrier + reductions
    comm.barrier()                                   # every rank is past the build

    nx, ny = args.grid
    host_cores = usable_cores()
    threads = max(1, min(host_cores, 64) // max(1, world))
    if world > 1:      # the ranks of a node share its cores: the library's host fan-outs (staging copies, stream layout) take a rank's share
        os.environ.setdefault("DSA_HOST_THREADS", str(max(2, min(32, host_cores // world))))
    ctx = dsa.Context(local_rank)
    ctx.set_profiling(True)
    schedule_note = ctx.schedule_note()

    def barrier():
        comm.barrier()
        torch.cuda.synchronize()

    def timed_leg(batch):
        """warmup + `steps` decodes of this rank's batch; returns (elapsed MAX over ranks, stage means of this rank)."""
        for _ in range(args.warmup):
            batch.decode(wait=True)
        stage_sum = {}
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            batch.decode(wait=True)
            for k, v in batch.stage_times().items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
            for k, v in batch.kernel_times().items():
                stage_sum["kernel:" + k] = stage_sum.get("kernel:" + k, 0.0) + v
        barrier()
        elapsed = time.perf_counter() - t0
        bad = [i for i in range(batch.n) if batch.status(i) != 0]
        if bad:
            raise SystemExit("rank %d: %d meshes failed to decode (first %d: status %d site %d)" %
                             (rank, len(bad), bad[0], batch.status(bad[0]), batch.mesh_info(bad[0]).detail))
        return comm.max(elapsed), {k: v / args.steps for k, v in stage_sum.items()}

    do_weak = args.scaling in ("weak", "both", "auto")
    do_strong = args.scaling == "strong" or (args.scaling in ("both", "auto") and world > 1)
    out, strong = None, None
    t_gen = t_upload = 0.0
    weak_blob = weak_offsets = None

    # ---- synthetic batch: SURVEY.md section 8d config 3 (positions 11 bit + normals 8 bit oct + UVs 10 bit,
    # standard Edgebreaker, parallelogram + wrap, per-attribute connectivity)
    if do_weak:
        t0 = time.perf_counter()
        blob, offsets = synth.make_batch(synth.GRID, nx, ny, 1000 + rank * args.meshes, args.meshes, normals=True, uvs=True, threads=threads)
        t_gen = time.perf_counter() - t0
        t0 = time.perf_counter()
        batch = dsa.Batch(ctx, blob=blob, offsets=offsets)
        t_upload = time.perf_counter() - t0
        elapsed, stages = timed_leg(batch)
        alg_bytes = batch.algorithmic_bytes
        alg_bytes_all = comm.sum(alg_bytes)
        weak_blob, weak_offsets = blob, offsets
        if rank == 0:
            step_s = elapsed / args.steps
            # the dominant kernel: the longest of the kernels timed one by one (an event pair around each on its own stream, mean
            # over the steps) -- a duration that is a row of `rocprofv3 --kernel-trace --stats` of this command (profiles/)
            kernel_ms = {k[len("kernel:"):]: v for k, v in stages.items() if k.startswith("kernel:")}
            stages = {k: v for k, v in stages.items() if not k.startswith("kernel:")}
            dom = max(kernel_ms, key=kernel_ms.get)
            kernel_name = dom.split("[")[0]
            dom_ms = kernel_ms[dom]
            if dom_ms > step_s * 1e3 * 1.02:
                raise SystemExit("kernel %s timed at %.2f ms, longer than the %.2f ms step it is part of" % (dom, dom_ms, step_s * 1e3))
            achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
            traffic, traffic_src, traffic_total = measured_traffic(kernel_name, args.meshes, 2 * nx * ny)
            out = {
                "metric": "decoded_meshes_per_sec",
                "value": args.meshes * world / step_s,
                "unit": "meshes/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": step_s * 1e3,
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "int32",
                "data": "synthetic",
                "config": {"workload": "batch of %d x %d-triangle Edgebreaker .drc per GPU (positions 11b + octahedral normals 8b + UVs 10b), device-resident decode"
                                       % (args.meshes, 2 * nx * ny),
                           "meshes_per_gpu": args.meshes, "triangles_per_mesh": 2 * nx * ny, "parallelism": "independent meshes, one batch per GPU, no collectives"},
                "gb_per_s": alg_bytes_all / step_s / 1e9,
                "algorithmic_bytes_per_gpu_step": alg_bytes,
                "compressed_bytes_per_gpu": int(offsets[-1]),
                "arena_bytes_per_gpu": batch.arena_bytes,
                "stage_ms": stages,
                # `achieved` follows the contract: the step's algorithmic bytes over the longest kernel's duration (HIP events on
                # that kernel's own stream).  The kernels of a step overlap on three streams, so the honest whole-path figure is
                # `step_frac`: the same bytes over the whole step.
                "roofline": {"bound": "hbm", "kernel": kernel_name, "kernel_launch": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                             "kernel_ms": dom_ms, "kernels_ms": kernel_ms, "algorithmic_bytes": alg_bytes,
                             "step_achieved": alg_bytes / step_s / 1e9, "step_frac": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                             "step_traffic": traffic_total},
                "setup_s": {"generate": t_gen, "upload_and_layout": t_upload},
            }
            # shader clock under this load, in-kernel: s_memtime ticks / s_memrealtime ticks (100 MHz) of the traversal and the
            # connectivity wave of 32 meshes of the batch, median (MI355X_MICROARCH.md, "DVFS give-back", item 6)
            dbg = [batch.debug_array(i, 4, np.uint32, 20) for i in range(0, args.meshes, max(1, args.meshes // 32))]
            for name, a, b in (("traverse", 6, 17), ("connectivity", 13, 15)):
                q = sorted(float(d[a]) / float(d[b]) * 0.1 for d in dbg if len(d) > b and d[b])
                if q:
                    out.setdefault("shader_clock_ghz", {})[name] = round(q[len(q) // 2], 3)
            if args.check > 0:
                idx = sorted(set(int(i) for i in np.linspace(0, args.meshes - 1, min(args.check, args.meshes))))
                bad = oracle_check(batch, blob, offsets, idx)
                if bad:
                    raise SystemExit("decoded results differ from the oracle on meshes %s" % bad[:8])
                out["oracle_check"] = {"meshes_compared": len(idx), "equal": True}
        batch.close()

    # ---- BASELINE.json configs[3]: the same --meshes streams, split over the ranks
    if do_strong:
        t0 = time.perf_counter()
        if do_weak and rank == 0:
            blob, offsets = weak_blob, weak_offsets                   # rank 0's weak batch is seeds 1000 ...: the job's batch
        else:
            blob, offsets = synth.make_batch(synth.GRID, nx, ny, 1000, args.meshes, normals=True, uvs=True, threads=threads)
        t_gen_s = time.perf_counter() - t0
        lengths = np.diff(offsets.astype(np.int64))
        mine = balanced_assignment(lengths, world)[rank]
        streams = [bytes(blob[int(offsets[i]):int(offsets[i + 1])]) for i in mine]
        batch = dsa.Batch(ctx, streams)
        elapsed, stages = timed_leg(batch)
        alg_bytes_all = comm.sum(batch.algorithmic_bytes)
        shard_sizes = [int(comm.sum(len(mine) if r == rank else 0)) for r in range(world)]
        if rank == 0:
            step_s = elapsed / args.steps
            strong = {"value": args.meshes / step_s, "unit": "meshes/s", "ms_per_step": step_s * 1e3, "meshes_in_job": args.meshes,
                      "meshes_per_gpu": shard_sizes, "assignment": "sharding.balanced_assignment (longest compressed stream first)",
                      "gb_per_s": alg_bytes_all / step_s / 1e9, "stage_ms_rank0": stages, "generate_s": t_gen_s}
        batch.close()
    sustained = None
    if not args.no_sustained and weak_blob is not None:
        sustained = sustained_leg(dsa, ctx, weak_blob, weak_offsets, max(4, args.steps), args.warmup, comm if world > 1 else None, barrier, world)
    encode = None
    if not args.no_encode:                                           # every rank: the leg's clock is the slowest rank's
        encode = encode_leg(dsa, synth, ctx, nx, ny, args.encode_meshes, comm if world > 1 else None, barrier if world > 1 else None, world)
    e2e = None
    if not args.no_end_to_end and weak_blob is not None:             # every rank its own batches: the clock is the slowest rank's
        e2e = end_to_end_leg(dsa, ctx, weak_blob, weak_offsets, args.e2e_batches, 2, comm if world > 1 else None, barrier if world > 1 else None, world, compact=True)
        e2e["full_layout"] = end_to_end_leg(dsa, ctx, weak_blob, weak_offsets, args.e2e_batches, 2, comm if world > 1 else None, barrier if world > 1 else None, world, compact=False)
    # (behind the end-to-end leg: the seamed batches of this leg have arenas of 120 GB at this batch size, and for seconds after such
    # an arena is freed the downloads of the end-to-end pipeline run at half the link rate -- 17 - 23 k meshes/s instead of 31 k;
    # profiles/README.md)
    dialects = None
    if not args.no_dialects and world == 1 and weak_blob is not None:
        dialects = dialects_leg(dsa, synth, ctx, nx, ny, args.meshes, 3)
    pool = None
    if not args.no_pool:
        # the in-library work queue over all GPUs of the job, from rank 0 alone (the other ranks have released their contexts'
        # caches and wait at the barrier): BASELINE.json configs[3] as a single-process host (the C# one) runs it
        ctx.close()
        ctx = None
        barrier()
        if rank == 0:
            if weak_blob is not None:
                pblob, poffs = weak_blob, weak_offsets
            else:
                pblob, poffs = synth.make_batch(synth.GRID, nx, ny, 1000, args.meshes, normals=True, uvs=True, threads=threads)
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))       # the GPUs of this node
            devices = [0] * world if rehearse else list(range(local_world))
            pool = pool_leg(dsa, devices, pblob, poffs, 0, 3)
        comm.host_barrier()        # the waiting ranks block on the host (gloo): an RCCL barrier would spin on the GPUs rank 0 is measuring on
    if rank == 0:
        if out is None:                                              # --scaling strong: the strong leg is the line
            out = {"metric": "decoded_meshes_per_sec", "value": strong["value"], "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": strong["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "int32", "data": "synthetic",
                   "config": {"workload": "one batch of %d x %d-triangle Edgebreaker .drc (positions 11b + octahedral normals 8b + UVs 10b) sharded over %d GPU(s), device-resident decode"
                                          % (args.meshes, 2 * nx * ny, world), "meshes_in_job": args.meshes, "triangles_per_mesh": 2 * nx * ny,
                              "parallelism": "independent meshes, balanced shards, no collectives"},
                   "gb_per_s": strong["gb_per_s"], "stage_ms": strong["stage_ms_rank0"]}
        elif strong is not None:
            out["strong_scaling"] = strong
        elif world == 1:
            out["strong_scaling"] = {"value": out["value"], "unit": "meshes/s", "ms_per_step": out["ms_per_step"], "meshes_in_job": args.meshes,
                                     "meshes_per_gpu": [args.meshes], "note": "one GPU: the strong and the weak partition are the same batch"}
        if world > 1 and args.scaling == "auto" and strong is not None and out.get("scaling") == "weak":
            # several GPUs: the line is BASELINE.json configs[3] (the same streams sharded), the per-GPU batches are beside it
            out["weak_scaling"] = {"value": out["value"], "unit": "meshes/s", "ms_per_step": out["ms_per_step"], "gb_per_s": out["gb_per_s"],
                                   "meshes_per_gpu": args.meshes, "meshes_in_job": args.meshes * world}
            out["value"], out["ms_per_step"], out["gb_per_s"], out["scaling"] = strong["value"], strong["ms_per_step"], strong["gb_per_s"], "strong"
            out["config"]["workload"] = ("one batch of %d x %d-triangle Edgebreaker .drc (positions 11b + octahedral normals 8b + UVs 10b) sharded over %d GPUs "
                                         "by compressed length, device-resident decode" % (args.meshes, 2 * nx * ny, world))
            out["config"]["meshes_in_job"] = args.meshes
            out["config"]["meshes_per_gpu"] = strong["meshes_per_gpu"]
            out["roofline"]["leg"] = "weak (a full %d-mesh batch per GPU): the kernels' roofline does not depend on the partition" % args.meshes
        if sustained is not None:
            out["sustained"] = sustained
        if dialects is not None:
            out["dialects"] = dialects
        out["schedule_note"] = schedule_note
        if encode is not None:
            out["encode"] = encode
        if e2e is not None:
            out["end_to_end"] = e2e
        if pool is not None:
            out["pool"] = pool
        out["notes"] = ("value / gb_per_s: compressed bytes resident in HBM -> results in HBM (no PCIe in the timed region); end_to_end: host bytes -> host "
                        "arrays; cpu_baseline: the oracle incl. its numpy export on the host")
        if not args.no_cpu_baseline and weak_blob is not None:       # rank 0, on the node's host cores, after every GPU leg
            out["cpu_baseline"] = cpu_baseline(weak_blob, weak_offsets, 1, 10.0, 1024)
            out["cpu_baseline_all_cores"] = cpu_baseline(weak_blob, weak_offsets, min(host_cores, 32), 10.0, 4096)
        print(json.dumps(out))
    comm.host_barrier()                                               # the other ranks wait on the host while rank 0 times the CPU
    if ctx is not None:
        ctx.close()
    comm.close()


if __name__ == "__main__":
    main()
