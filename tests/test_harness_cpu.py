"""CPU: the host-side mirror of run.rs (infer task, greedy process, perplexity, replica router) against the oracle,
using an oracle-backed fake runtime; plus the N>1 path of the replica job over torch.distributed/gloo."""
import os
import socket
import sys

import numpy as np
import pytest

from ai00_server_amd.harness import InferLoop, InferRequest, ReplicaRouter, greedy_process, perplexity
from ai00_server_amd.runtime import RnnOption
from oracle import rwkv_ref as R
from tests.fakes import OracleRuntime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref(name="v6-tiny"):
    return R.RwkvRef(R.synth_named(name))


def _prompt(ref, slot, n):
    return [t % ref.info.num_vocab for t in R.synth_prompt(slot, n)]


def test_infer_loop_batches_one_request_per_slot_and_chunks():
    ref = _ref()
    rt_ = OracleRuntime(ref, max_batch=3, token_chunk_size=8)
    loop = InferLoop(rt_)
    p0, p1 = _prompt(ref, 0, 21), _prompt(ref, 1, 5)
    a = loop.submit(InferRequest(0, p0))
    b = loop.submit(InferRequest(1, p1))
    c = loop.submit(InferRequest(1, [7, 8]))              # second request of slot 1 must wait for the first (FIFO)
    loop.run_pending()
    s = ref.init_state()
    np.testing.assert_array_equal(a.outputs[-1][-1], ref.forward(p0, s)[-1])
    s = ref.init_state()
    np.testing.assert_array_equal(b.outputs[-1][-1], ref.forward(p1, s)[-1])
    np.testing.assert_array_equal(c.outputs[-1][-1], ref.forward([7, 8], s)[-1])
    assert rt_.calls >= 4                                   # 21 tokens through an 8-token chunk budget
    assert len(a.outputs) == 1                              # Last: exactly one emission


def test_greedy_process_and_empty_prompt():
    ref = _ref("v5-tiny")
    loop = InferLoop(OracleRuntime(ref, max_batch=2, token_chunk_size=16))
    p = _prompt(ref, 3, 9)
    got = greedy_process(loop, 1, p, 12)
    want, _ = ref.greedy(p, 12)
    assert got == want[:len(got)] and (len(got) == 12 or want[len(got)] == 0)
    loop2 = InferLoop(OracleRuntime(ref, max_batch=1))
    g0 = greedy_process(loop2, 0, [], 3)                    # empty prompt => [0] (run.rs:489-492)
    w0, _ = ref.greedy([], 3)
    assert g0 == w0[:len(g0)]


def test_perplexity_matches_reference_formula():
    ref = _ref("v7-tiny")
    loop = InferLoop(OracleRuntime(ref, max_batch=1, token_chunk_size=4))
    choice = _prompt(ref, 4, 7)
    got = perplexity(loop, 0, choice)
    s = ref.init_state()
    rows = ref.forward([0] + choice, s, full=True)
    assert abs(got - R.perplexity_ref(rows, choice)) < 1e-5


def test_router_shards_documents_round_robin():
    ref = _ref()
    docs = [_prompt(ref, 10 + i, 6 + i) for i in range(7)]
    one = ReplicaRouter([OracleRuntime(ref, max_batch=2)]).embed_documents(docs, layer=1)
    three = ReplicaRouter([OracleRuntime(ref, max_batch=2) for _ in range(3)]).embed_documents(docs, layer=1)
    for a, b in zip(one, three):
        np.testing.assert_array_equal(a, b)
    s = ref.init_state()
    ref.forward(docs[4], s)
    np.testing.assert_array_equal(one[4], s[1, 1:1 + ref.info.head_size])
    assert list(ReplicaRouter.shard(10, 1, 4)) == [1, 5, 9]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = _ref()
    docs = [_prompt(ref, 20 + i, 5) for i in range(6)]
    router = ReplicaRouter([OracleRuntime(ref, max_batch=2)])
    mine = list(ReplicaRouter.shard(len(docs), rank, world))
    emb = router.embed_documents([docs[i] for i in mine], layer=0)
    # aggregate like bench.py does: per-rank unit counts summed, time = MAX over ranks, checksum of checksums
    cnt = torch.tensor([float(len(mine))])
    chk = torch.tensor([float(sum(float(np.abs(e).sum()) for e in emb))], dtype=torch.float64)
    tmax = torch.tensor([1.0 + rank])
    dist.all_reduce(cnt)
    dist.all_reduce(chk)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        q.put((cnt.item(), chk.item(), tmax.item()))
    dist.destroy_process_group()


def test_two_rank_gloo_replica_job_matches_single_rank():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    cnt, chk, tmax = q.get(timeout=10)
    ref = _ref()
    docs = [_prompt(ref, 20 + i, 5) for i in range(6)]
    emb = ReplicaRouter([OracleRuntime(ref, max_batch=2)]).embed_documents(docs, layer=0)
    assert cnt == 6 and tmax == 2.0
    assert abs(chk - sum(float(np.abs(e).sum()) for e in emb)) < 1e-6 * max(1.0, chk)


def test_router_places_interactive_requests_on_the_least_busy_replica():
    """`pick`: least busy replica with a free slot, lowest index on ties, -1 when all are full; `generate` spreads prompts over the
    replicas that way and every answer equals the single-engine answer."""
    ref = _ref("v5-tiny")
    router = ReplicaRouter([OracleRuntime(ref, max_batch=2) for _ in range(3)])
    assert [router.pick() for _ in range(7)] == [0, 1, 2, 0, 1, 2, -1]
    router.release(1)
    assert router.pick() == 1
    for r in (0, 0, 1, 1, 2, 2):
        router.release(r)
    assert router.busy == [0, 0, 0]
    prompts = [_prompt(ref, 20 + i, 3 + i) for i in range(8)]                 # more prompts than slots: two waves
    got = router.generate(prompts, 5)
    for p, g in zip(prompts, got):
        want, _ = ref.greedy(p, 5)
        assert g == want[:len(g)] and (len(g) == 5 or want[len(g)] == 0)
    assert router.busy == [0, 0, 0]


class _NumpyArena:                                            # stand-in for runtime.PinnedArena on a machine without a GPU
    def __init__(self, shape):
        self.array = np.zeros(shape, np.float32)

    def close(self):
        self.array = None


@pytest.mark.parametrize("name", ["v6-tiny", "v7-tiny"])
def test_state_job_slot_turnover_equals_one_document_at_a_time(name):
    """harness.StateJob (the `/embeddings` batch job: GenerateKind::State requests with slot turnover, run.rs:980-989) over more
    documents than slots and RAGGED lengths: a slot that finishes takes the next document while the others are mid-flight, every
    embedding equals the document prefilled alone from the initial state, and nothing is read before `sync()`."""
    from ai00_server_amd.harness import StateJob
    ref = _ref(name)
    rt_ = OracleRuntime(ref, max_batch=3, token_chunk_size=8)
    lens = [5, 17, 1, 9, 0, 12, 3, 8, 20, 2, 6]
    docs = [_prompt(ref, 40 + i, n) for i, n in enumerate(lens)]
    layer = ref.info.num_layer - 1
    job = StateJob(rt_, layer, arena=_NumpyArena)
    emb, calls = job.run(docs)
    assert emb.shape == (len(docs), ref.info.head_size, ref.info.num_emb) and np.isfinite(emb).all()
    for i, d in enumerate(docs):
        s = ref.init_state()
        ref.forward(d if len(d) else [0], s)                      # empty prompt => [0] (run.rs:489-492)
        np.testing.assert_array_equal(emb[i], s[layer, 1:1 + ref.info.head_size], err_msg=f"document {i}")
    # turnover really happened: 11 documents through 3 slots in fewer calls than running them in waves of 3 to completion
    waves = sum(-(-max(max(1, n) for n in lens[i:i + 3]) // (8 // min(3, len(lens[i:i + 3])))) for i in range(0, len(lens), 3))
    assert calls < waves + 4
    # a second job on the same object starts from clean slots again
    emb2, _ = job.run(docs[:4])
    np.testing.assert_array_equal(emb2, emb[:4])
    job.close()
