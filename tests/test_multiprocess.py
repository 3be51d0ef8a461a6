"""N > 1 plumbing on CPU: two gloo ranks shard a batch with no data-path collective and reduce the
benchmark timing exactly as bench.py does on the GPUs (barrier, MAX over ranks, SUM of units)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from draco_sharp_amd.sharding import balanced_assignment, shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = shard_bounds(n, r, world)
                assert 0 <= a <= b <= n
                seen.extend(range(a, b))
            assert seen == list(range(n))
            sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_assignment_is_a_partition_and_balanced():
    rng = np.random.default_rng(3)
    lengths = rng.integers(50_000, 300_000, 1000)
    for world in (1, 2, 4, 8):
        parts = balanced_assignment(lengths, world)
        assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
        loads = [int(lengths[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(lengths.max())


WORKER = textwrap.dedent("""
    import os, sys, json, time
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    from draco_sharp_amd.sharding import Comm, shard_bounds, aggregate_throughput
    import draco_sharp_amd.synth as synth
    import oracle
    comm = Comm(backend="gloo")
    assert comm.world == 2
    # every rank derives the same 6-mesh batch, takes its shard, decodes it (CPU oracle stands in for the GPU here)
    n = 6
    a, b = shard_bounds(n, comm.rank, comm.world)
    faces = 0
    comm.barrier()
    t0 = time.perf_counter()
    for i in range(a, b):
        pos, nrm, uv, f = synth.make_mesh(synth.GRID, 6 + i, 5, 100 + i)
        m = oracle.decode(synth.encode_mesh(pos, f, nrm, uv))
        faces += m.num_faces
    comm.barrier()
    elapsed = time.perf_counter() - t0 + 0.01 * (comm.rank + 1)     # rank 1 is "slower"
    value, worst = aggregate_throughput(b - a, elapsed, 1, comm)
    total_faces = comm.sum(faces)
    # strong split (BASELINE.json configs[3], bench.py --scaling strong): every rank derives the SAME job, the streams are
    # assigned longest first, each rank decodes only its own
    from draco_sharp_amd.sharding import balanced_assignment
    streams = []
    for i in range(n):
        pos, nrm, uv, f = synth.make_mesh(synth.GRID, 6 + i, 5, 100 + i)
        streams.append(synth.encode_mesh(pos, f, nrm, uv))
    parts = balanced_assignment([len(s) for s in streams], comm.world)
    strong_faces = sum(oracle.decode(streams[i]).num_faces for i in parts[comm.rank])
    strong_total = comm.sum(strong_faces)
    sizes = [int(comm.sum(len(parts[comm.rank]) if r == comm.rank else 0)) for r in range(comm.world)]
    strong_bytes = [int(comm.sum(sum(len(streams[i]) for i in parts[comm.rank]) if r == comm.rank else 0)) for r in range(comm.world)]
    if comm.rank == 0:
        print(json.dumps({"value": value, "worst": worst, "mine": elapsed, "total_faces": total_faces, "shard": [a, b],
                          "strong_total_faces": strong_total, "strong_sizes": sizes, "strong_bytes": strong_bytes,
                          "parts": parts, "max_stream": max(len(s) for s in streams)}))
    comm.close()
""")


def test_two_gloo_ranks_shard_and_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["shard"] == [0, 3]
    # faces of all six meshes were counted exactly once across the two ranks
    assert res["total_faces"] == sum(2 * (6 + i) * 5 for i in range(6))
    # MAX over ranks: rank 1 added more delay than rank 0
    assert res["worst"] >= res["mine"]
    assert abs(res["value"] - 6 / res["worst"]) < 1e-6
    # the strong split: a partition of the same six streams, balanced by compressed bytes, every face counted once
    assert sorted(i for p in res["parts"] for i in p) == list(range(6)) and res["strong_sizes"] == [len(p) for p in res["parts"]]
    assert res["strong_total_faces"] == res["total_faces"]
    assert abs(res["strong_bytes"][0] - res["strong_bytes"][1]) <= res["max_stream"]
