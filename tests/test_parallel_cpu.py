"""
world_size-2 gloo test (CPU) of the multi-GPU plumbing: tile partition, gather of per-tile runs of
unequal length, and independence of the result from the number of ranks.
"""
import os
import socket

import numpy as np
import pytest

from localmd_amd.parallel import Dist, tile_partition


def test_tile_partition_properties():
    for n, w in [(2601, 8), (625, 2), (7, 8), (35, 4), (0, 3), (65025, 8)]:
        runs = tile_partition(n, w)
        assert len(runs) == w and runs[0][0] == 0 and runs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
        sizes = [hi - lo for lo, hi in runs]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_tiles, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = Dist(True)
        assert (d.rank, d.world, d.enabled) == (rank, world, True)
        runs = tile_partition(n_tiles, world)
        lo, hi = runs[rank]
        # per-tile payloads: a "basis" block and a rank count that depend on the tile index only
        ut = torch.zeros((n_tiles, 4, 6), dtype=torch.float32)
        ranks = torch.zeros(n_tiles, dtype=torch.int32)
        for t in range(lo, hi):
            ut[t] = float(t + 1)
            ranks[t] = (t % 3) + 1
        d.gather_runs(ut, runs)
        d.gather_runs(ranks, runs)
        offsets = np.concatenate([[0], np.cumsum(ranks.numpy())])
        # variable-length row blocks (the compacted temporal traces)
        vc = torch.zeros((int(offsets[-1]), 5), dtype=torch.float32)
        for t in range(lo, hi):
            vc[offsets[t]:offsets[t + 1]] = float(t + 1)
        d.gather_runs(vc, [(int(offsets[a]), int(offsets[b])) for a, b in runs])
        d.barrier()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ut=ut.numpy(), ranks=ranks.numpy(), vc=vc.numpy())
    finally:
        dist.destroy_process_group()


def test_gather_runs_world2_gloo(tmp_path):
    import torch.multiprocessing as mp

    n_tiles, world = 11, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_tiles, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    exp_ut = np.arange(1, n_tiles + 1, dtype=np.float32)[:, None, None] * np.ones((1, 4, 6), dtype=np.float32)
    exp_ranks = (np.arange(n_tiles) % 3 + 1).astype(np.int32)
    exp_vc = np.repeat(np.arange(1, n_tiles + 1, dtype=np.float32), exp_ranks)[:, None] * np.ones((1, 5), dtype=np.float32)
    for g in got:
        np.testing.assert_array_equal(g["ut"], exp_ut)
        np.testing.assert_array_equal(g["ranks"], exp_ranks)
        np.testing.assert_array_equal(g["vc"], exp_vc)


def _worker_global(rank, world, port, out_dir):
    """The exchange steps of the row-sharded global stage: partial Gram matrices summed over the ranks, row blocks
    of R collected on rank 0 (with the per-block callback the host driver uses to start downloads)."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = Dist(True)
        rng = np.random.default_rng(0)
        M = torch.from_numpy(rng.standard_normal((23, 7)).astype(np.float32))   # same on every rank
        bounds = [(0, 9), (9, 23)] if world == 2 else [(0, 5), (5, 5), (5, 23)]
        lo, hi = bounds[rank]
        C = M[lo:hi].T @ M[lo:hi]            # partial M^T M of this rank's rows
        d.all_reduce(C)
        R = torch.zeros((23, 3), dtype=torch.float32)
        R[lo:hi] = M[lo:hi, :3] * 2.0
        seen = []
        d.gather_rows_to_root(R, bounds, on_block=(lambda a, b: seen.append((a, b))) if rank == 0 else None)
        d.barrier()
        np.savez(os.path.join(out_dir, f"g{rank}.npz"), C=C.numpy(), R=R.numpy(), M=M.numpy(), seen=np.array(seen).reshape(-1, 2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_global_stage_exchanges_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker_global, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"g{r}.npz") for r in range(world)]
    M = got[0]["M"]
    for g in got:
        np.testing.assert_allclose(g["C"], M.T @ M, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(got[0]["R"], M[:, :3] * 2.0)       # rank 0 holds every row
    expected_blocks = [(9, 23)] if world == 2 else [(5, 23)]         # empty runs are skipped
    assert [tuple(x) for x in got[0]["seen"].tolist()] == expected_blocks


def test_single_process_dist_is_a_noop():
    d = Dist(False)
    assert (d.rank, d.world, d.enabled) == (0, 1, False)
    d.gather_runs(None, [(0, 1)])
    d.all_reduce(None)
    d.gather_rows_to_root(None, [(0, 1)])
    d.barrier()
    with pytest.raises(RuntimeError):
        Dist(True)


def _worker_halo(rank, world, port, out_dir):
    """exchange_rows: every rank ends up with its own rows plus the requested halo rows of its neighbours, and with
    nothing else (rows nobody sent stay at their fill value)."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = Dist(True)
        bounds = [(0, 7), (7, 12), (12, 20)]
        needs = [[(7, 9)], [(5, 7), (12, 13)], [(10, 12), (3, 4)]]
        x = torch.full((20, 3), -1.0)
        lo, hi = bounds[rank]
        x[lo:hi] = torch.arange(lo, hi, dtype=torch.float32)[:, None] + 100.0 * rank
        d.exchange_rows(x, bounds, needs)
        np.save(os.path.join(out_dir, f"halo{rank}.npy"), x.numpy())
    finally:
        dist.destroy_process_group()


def test_exchange_rows_world3_gloo(tmp_path):
    import torch.multiprocessing as mp

    world = 3
    mp.spawn(_worker_halo, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    bounds = [(0, 7), (7, 12), (12, 20)]
    needs = [[(7, 9)], [(5, 7), (12, 13)], [(10, 12), (3, 4)]]
    full = np.empty((20, 3), dtype=np.float32)
    for r, (lo, hi) in enumerate(bounds):
        full[lo:hi] = np.arange(lo, hi, dtype=np.float32)[:, None] + 100.0 * r
    for r in range(world):
        got = np.load(tmp_path / f"halo{r}.npy")
        have = np.zeros(20, dtype=bool)
        have[bounds[r][0]:bounds[r][1]] = True
        for lo, hi in needs[r]:
            have[lo:hi] = True
        np.testing.assert_array_equal(got[have], full[have])
        assert np.all(got[~have] == -1.0)


def _worker_blocks(rank, world, port, out_dir):
    """Column blocks of Vt collected on rank 0 (round 3: the frames x frames products sharded by frame columns), and the
    partial Gram matrices of the column blocks summed to the Gram matrix of the whole."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = Dist(True)
        T, rp = 11, 4          # 11 frame columns over 3 ranks: 4, 4, 3 (and a rank with none when T < world)
        for T_ in (T, 2):
            cparts = tile_partition(T_, world)
            c0, c1 = cparts[rank]
            full = torch.arange(rp * T_, dtype=torch.float32).reshape(rp, T_) * 0.5 + 1.0
            mine = full[:, c0:c1].contiguous()
            blocks = d.gather_blocks_to_root(mine, [(rp, b - a) for a, b in cparts])
            gram = mine @ mine.T
            d.all_reduce(gram)
            if rank == 0:
                got = torch.cat([b for b in blocks if b.shape[1] > 0], dim=1)
                assert torch.equal(got, full), (got, full)
            else:
                assert blocks is None
            assert torch.allclose(gram, full @ full.T)
        d.barrier()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_column_blocks_and_partial_gram_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker_blocks, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))
