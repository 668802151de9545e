"""The N>1 path on REAL GPUs (BASELINE configs[3]): one process and one rgb_ctx per GPU over its hash
shard (rgb_route), no data-path collective, the leaderboard shards all-gathered with RCCL through the C entry point
(rgb_leaderboard_allgather) exactly as bench.py does between trains.  Skipped unless the box has at least two GPUs -- the pool's test boxes have one, the
driver's scaling node has eight: the first multi-GPU contact is then a PARITY run, not only a timing.

What is compared: (i) the RCCL-gathered leaderboard of every rank against ONE process that computes every group with
the checker (the assertion of tests/test_shard_gloo.py, which covers the same code with gloo on the CPU), (ii) every
rank's device state checksum against the checker's checksum of that shard."""
import os
import socket

import numpy as np
import pytest

from ra_amd import abi, shard, workload as W

N = 5
TICKS = 6
SEED = 0x5EED0004


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_stream(gids, step, get_state, set_state):
    """TICKS generator ticks over the groups `gids` (every group seeded by its global uid, so the stream of a group
    does not depend on which rank owns it) through step(msgs)."""
    for t in range(TICKS):
        cur = get_state()
        if W.heal(cur, N, max_runs=16):
            set_state(cur)
        msgs = []
        for k, g in enumerate(gids):
            m = W.gen_tick(cur[k * N:(k + 1) * N], N, t, SEED ^ int(g))
            m["server"] += k * N
            msgs.append(m)
        step(np.concatenate(msgs))


def _initial(gids):
    st = np.concatenate([W.initial_states(1, N, SEED ^ int(g)) for g in gids])
    st["self"] = np.arange(len(gids) * N) % N
    return st


def _rccl_worker(rank, world, port, out_dir, g_global):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from ra_amd import engine
    L = engine.lib()
    mine = np.array([g for g in range(g_global) if L.rgb_route(g, world) == rank], dtype=np.uint64)
    assert np.array_equal(mine, shard.local_group_ids(g_global, world, rank))
    G = len(mine)
    dev = torch.device("cuda", rank)
    with engine.RaGpuBatch(G, N, device=rank, max_runs=16, ring_capacity=G * N, ring_slots=2) as eng:
        eng.set_state(0, _initial(mine))
        _shard_stream(mine, eng.step, eng.get_state, lambda s: eng.set_state(0, s))
        # the bench's collective: the rank's leaderboard rows stay on the device and are all-gathered with RCCL
        gmax = torch.tensor([G], dtype=torch.int64, device=dev)
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
        m = int(gmax.item())
        lb = torch.zeros(m * 32, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev)
        eng.snapshot_device(lb.data_ptr(), stream.cuda_stream)
        ids = torch.full((m,), -1, dtype=torch.int64, device=dev)
        ids[:G] = torch.from_numpy(mine.astype(np.int64)).to(dev)
        lb_all = torch.empty(world * m * 32, dtype=torch.uint8, device=dev)
        id_all = torch.empty(world * m, dtype=torch.int64, device=dev)
        # the leaderboard shards through the C entry point (rgb_leaderboard_allgather = ncclAllGather behind the
        # boundary, what the NIF calls); the communicator id travels from rank 0 by the host's own means
        idt = torch.zeros(abi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).to(dev)
        dist.broadcast(idt, 0)
        comm = engine.Comm(eng, idt.cpu().numpy().tobytes(), world, rank)
        comm.allgather_leaderboard(lb.data_ptr(), m, lb_all.data_ptr(), stream.cuda_stream)
        dist.all_gather_into_tensor(id_all, ids)
        torch.cuda.synchronize()
        ids_h = id_all.cpu().numpy()
        rows_h = lb_all.cpu().numpy().view(abi.LEADERBOARD_DTYPE)
        keep = ids_h >= 0
        order = np.argsort(ids_h[keep], kind="stable")
        np.save(os.path.join(out_dir, f"uids_{world}_{rank}.npy"), ids_h[keep][order].astype(np.uint64))
        np.save(os.path.join(out_dir, f"rows_{world}_{rank}.npy"), rows_h[keep][order].view(np.uint8))
        np.save(os.path.join(out_dir, f"sum_{world}_{rank}.npy"), np.array([eng.state_checksum()], dtype=np.uint64))
        comm.close()
    dist.destroy_process_group()


def _checker_shard(oracle_lib, gids):
    cpu = oracle_lib.Oracle(len(gids), N, max_runs=16)
    cpu.set_state(0, _initial(gids))
    _shard_stream(gids, cpu.step, cpu.get_state, lambda s: cpu.set_state(0, s))
    st = cpu.get_state()
    cpu.close()
    from ra_amd import engine
    return st, engine.combine_checksums(oracle_lib.server_checksums(st))     # = what rgb_state_checksum returns


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs at least two GPUs (the pool's test boxes have one)")
def test_rccl_gathered_leaderboard_and_per_rank_state_equal_the_checker(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    g_global = 384
    all_g = np.arange(g_global, dtype=np.uint64)
    ref_state, _ = _checker_shard(oracle_lib, all_g)
    ref_rows = shard.leaderboard_rows_from_states(ref_state, N)
    for world in sorted({2, _n_gpus()}):
        mp.spawn(_rccl_worker, args=(world, _free_port(), str(tmp_path), g_global), nprocs=world, join=True)
        for r in range(world):
            uids = np.load(tmp_path / f"uids_{world}_{r}.npy")
            rows = np.load(tmp_path / f"rows_{world}_{r}.npy").view(abi.LEADERBOARD_DTYPE)
            assert np.array_equal(uids, all_g), f"world {world} rank {r}: groups missing from the gathered leaderboard"
            assert rows.tobytes() == ref_rows.tobytes(), f"world {world} rank {r}: gathered leaderboard differs from the checker"
            # per-rank state: the checker on exactly this rank's shard
            mine = shard.local_group_ids(g_global, world, r)
            _, chk = _checker_shard(oracle_lib, mine)
            got = int(np.load(tmp_path / f"sum_{world}_{r}.npy")[0])
            assert got == chk, f"world {world} rank {r}: device state checksum differs from the checker's"


def test_checker_side_of_the_multi_gpu_test(oracle_lib):
    """The CPU half of the RCCL test runs everywhere: the shards' streams do not depend on the partition (a group's
    rows are the same whether one process or eight own the groups), and the checksum helper is the device's."""
    g_global = 48
    all_g = np.arange(g_global, dtype=np.uint64)
    ref_state, _ = _checker_shard(oracle_lib, all_g)
    ref_rows = shard.leaderboard_rows_from_states(ref_state, N)
    for world in (2, 8):
        rows = {}
        for r in range(world):
            mine = shard.local_group_ids(g_global, world, r)
            st, chk = _checker_shard(oracle_lib, mine)
            assert isinstance(chk, int)
            for g, row in zip(mine, shard.leaderboard_rows_from_states(st, N)):
                rows[int(g)] = row
        got = np.array([rows[g] for g in range(g_global)], dtype=abi.LEADERBOARD_DTYPE)
        assert got.tobytes() == ref_rows.tobytes()


@pytest.mark.gpu
def test_c_allgather_with_one_rank_on_the_gpu():
    """The RCCL binding behind the C ABI on the pool's one-GPU boxes: a communicator of one rank (ncclCommInitRank from a
    C-created id), rgb_leaderboard_allgather on the launch stream = the rank's own rows, bit for bit."""
    import torch
    from ra_amd import engine
    G = 1024
    with engine.RaGpuBatch(G, N, max_runs=16, ring_capacity=64, ring_slots=1) as eng:
        eng.set_state(0, W.initial_states(G, N, SEED))
        comm = engine.Comm(eng, engine.comm_unique_id(), 1, 0)
        stream = torch.cuda.Stream()
        lb = torch.zeros(G * 32, dtype=torch.uint8, device="cuda")
        lb_all = torch.zeros(G * 32, dtype=torch.uint8, device="cuda")
        eng.snapshot_device(lb.data_ptr(), stream.cuda_stream)
        comm.allgather_leaderboard(lb.data_ptr(), G, lb_all.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        want = eng.snapshot()
        assert lb_all.cpu().numpy().tobytes() == want.tobytes()
        comm.close()
