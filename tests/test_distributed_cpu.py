"""`not gpu`: the N>1 path on CPU — world_size 2 over gloo.  Covers shard bounds, the all-gather of
score blocks (equal and unequal shards) and rank assembly; the per-shard scores come from the
CPU-emulation build so the whole sharded pipeline is exercised end to end."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from shoeprint_image_retrieval_amd import distributed as sdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 1500, 100001):
        for world in (1, 2, 3, 8):
            b = [sdist.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_gallery, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SPR_EMU_THREADS="2")
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    from emu_util import emu_scorer
    from shoeprint_image_retrieval_amd import synth

    nq, c, h, wd, seed = 3, 2, 16, 12, 5
    q, g, m = synth.dataset(seed, nq, n_gallery, c, h, wd, signal=1, noise=6)
    s, e = sdist.shard_bounds(n_gallery, world, rank)
    sc = emu_scorer("fft")
    local = sc.score_matrix(q, g[s:e]) if e > s else np.zeros((nq, 0), np.float32)
    full = sdist.gather_score_blocks(torch.from_numpy(local), n_gallery)
    assert full.shape == (nq, n_gallery)
    ranks = sc.ranks(full.numpy(), m)
    # sharded counting form: counts summed over shards + 1 == rank
    lib, dev = sc.lib, sc.dev
    ms = dev.to_device(full.numpy()[np.arange(nq), m].astype(np.float32))
    counts = dev.zeros((nq,), np.int32)
    loc = dev.to_device(local)
    lib.check(lib.spr_rank_count_greater(dev.ptr(loc), max(1, e - s), nq, e - s, s, dev.ptr(ms),
                                         dev.ptr(dev.to_device(np.asarray(m, np.int32))), dev.ptr(counts), 0))
    t = torch.from_numpy(dev.to_host(counts).astype(np.int64))
    dist.all_reduce(t)
    assert np.array_equal(t.numpy() + 1, ranks)
    assert sdist.max_over_ranks(float(rank)) == world - 1
    sdist.barrier()
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
    np.save(os.path.join(out_dir, f"ranks_{rank}.npy"), ranks)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_gallery", [8, 7])
def test_two_rank_sharded_scoring_matches_single_process(tmp_path, n_gallery):
    from emu_util import emu_scorer
    from shoeprint_image_retrieval_amd import synth

    emu_scorer("fft")  # build the emulation library once, before forking workers
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), n_gallery, str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    q, g, m = synth.dataset(5, 3, n_gallery, 2, 16, 12, signal=1, noise=6)
    sc = emu_scorer("fft")
    single = sc.score_matrix(q, g)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"full_{r}.npy"), single)
        np.testing.assert_array_equal(np.load(tmp_path / f"ranks_{r}.npy"), sc.ranks(single, m))
