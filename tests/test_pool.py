"""In-library multi-device submit (include/draco_mi355x.h dsa_pool_*): the queue plan is checked without a GPU; the
decode through two contexts and two worker threads on one GPU is the `-m gpu` half."""
import numpy as np
import pytest

import oracle
import draco_sharp_amd as dsa
import draco_sharp_amd.synth as synth
from draco_sharp_amd import native


def test_queue_plan_is_longest_first_and_partitions_the_job():
    rng = np.random.default_rng(11)
    for n in (0, 1, 5, 64, 65, 1000):
        lengths = [int(x) for x in rng.integers(1, 300_000, n)]
        if n > 10:
            lengths[3] = lengths[7] = lengths[9]              # ties keep the submission order
        for chunk in (1, 7, 64, 4096):
            order, begin = dsa.pool_plan(lengths, chunk)
            assert sorted(order) == list(range(n))
            assert all((-lengths[a], a) < (-lengths[b], b) for a, b in zip(order, order[1:]))
            if n == 0:
                assert begin == [0]
                continue
            assert begin[0] == 0 and begin[-1] == n and all(0 < b - a <= chunk for a, b in zip(begin, begin[1:]))
            assert len(begin) - 1 == (n + chunk - 1) // chunk
            # consecutive chunks carry non-increasing amounts of compressed bytes per stream: the heaviest work is queued first
            heads = [lengths[order[a]] for a in begin[:-1]]
            assert heads == sorted(heads, reverse=True)


def test_pool_needs_a_gpu_and_says_so():
    if native.lib().dsa_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(dsa.DeviceException):
        dsa.Pool([0, 0])


@pytest.mark.gpu
def test_two_contexts_on_one_gpu_decode_a_job():
    """Two contexts and two worker threads on device 0 inside one process: every stream of the job decodes exactly as
    the oracle says, bad streams fail alone, and both workers took chunks."""
    streams = []
    for i in range(150):
        kind = [synth.GRID, synth.TORUS, synth.HOLES, synth.SPHERE][i % 4]
        pos, nrm, uv, faces = synth.make_mesh(kind, 12 + i % 30, 12 + (7 * i) % 23, 400 + i)
        streams.append(synth.encode_mesh(pos, faces, nrm if i % 3 else None, uv if i % 5 else None, opt=synth.options(force_scheme=i % 2)))
    streams[17] = streams[17][:len(streams[17]) // 2]         # truncated: invalid data, alone
    streams[90] = b"DRACO\x02\x02\x00\x01\x00\x00" + b"\0" * 8  # kd-tree point cloud: not implemented, alone
    pool = dsa.Pool([0, 0], chunk_meshes=16)
    job = pool.decode(streams)
    assert job.chunks == (150 + 15) // 16
    workers = set()
    for i, s in enumerate(streams):
        workers.add(job.worker(i))
        if i == 17:
            assert job.status(i) == native.DSA_ERR_INVALID_DATA
            with pytest.raises(dsa.InvalidDataException):
                job.result(i)
            continue
        if i == 90:
            assert job.status(i) == native.DSA_ERR_NOT_IMPLEMENTED
            continue
        assert job.status(i) == 0, (i, job.status(i))
        ref = oracle.decode(s)
        m = job.result(i).ConnectedData
        assert np.array_equal(m.Faces, ref.faces) and len(m.Attributes) == len(ref.attributes)
        for a, r in zip(m.Attributes, ref.attributes):
            assert np.array_equal(a.PortableValues, r.portable) and np.array_equal(a.PointMap, r.point_map) and a.Values.tobytes() == r.values.tobytes()
    assert workers == {0, 1}
    # a second job on the same pool, and an empty one
    job2 = pool.decode(streams[:5])
    assert [job2.status(i) for i in range(5)] == [0] * 5
    job3 = pool.decode([])
    assert job3.chunks == 0
    for j in (job, job2, job3):
        j.close()
    pool.close()


@pytest.mark.gpu
def test_a_job_is_freed_while_the_next_one_decodes():
    """The threading contract of include/draco_mi355x.h: jobs and batches of one context may be freed from another thread while the
    context decodes (the caches of a context -- spare arenas, mirrors, descriptor blocks -- and its batch count are behind its
    mutex).  A consumer thread closes job N while the pool decodes job N + 1 on the same contexts, sixteen times over; and batches
    of one Context are dropped by one thread while another decodes on it."""
    import threading
    streams = []
    for i in range(96):
        kind = [synth.GRID, synth.TORUS, synth.HOLES, synth.SPHERE][i % 4]
        pos, nrm, uv, faces = synth.make_mesh(kind, 16 + i % 20, 14 + (5 * i) % 17, 900 + i)
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(uv_prediction=5 if i % 3 == 0 else 1)))
    refs = [oracle.decode(s) for s in streams[:6]]
    pool = dsa.Pool([0, 0], chunk_meshes=8)
    errors = []
    def check_and_close(job):
        try:
            for i in range(6):
                assert job.status(i) == 0
                m = job.result(i).ConnectedData
                assert np.array_equal(m.Faces, refs[i].faces)
                for a, r in zip(m.Attributes, refs[i].attributes):
                    assert a.Values.tobytes() == r.values.tobytes()
            job.close()
        except Exception as e:                              # noqa: BLE001 (reported by the test thread)
            errors.append(e)
    prev = None
    for round_ in range(16):
        job = pool.decode(streams)                          # (runs its workers; the previous job is being read and freed meanwhile)
        if prev is not None:
            prev.join()
        prev = threading.Thread(target=check_and_close, args=(job,))
        prev.start()
    prev.join()
    pool.close()
    assert not errors, errors[:2]
    # one Context: a thread frees finished batches while the main thread builds and decodes new ones
    ctx = dsa.Context(0)
    done = []
    lock = threading.Lock()
    stop = threading.Event()
    def reaper():
        while not stop.is_set() or done:
            with lock:
                b = done.pop() if done else None
            if b is not None:
                try:
                    assert b.status(0) == 0
                    b.close()
                except Exception as e:                      # noqa: BLE001
                    errors.append(e)
    t = threading.Thread(target=reaper)
    t.start()
    for k in range(40):
        b = dsa.Batch(ctx, streams[(k * 7) % 64:(k * 7) % 64 + 24])
        b.decode()
        with lock:
            done.append(b)
    stop.set()
    t.join()
    ctx.close()
    assert not errors, errors[:2]
