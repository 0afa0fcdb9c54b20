"""world_size-2 `gloo` test of the multi-GPU path on CPU: film partition into interleaved row tiles + the one
reduce.  The per-rank partial films come from the oracle's restricted render (the device kernels are covered by
the -m gpu tests); the partition bookkeeping and the collective are the product code under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tile_rows, out_dir, all_ranks):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from mitsuba2_amd import dist as mdist, scenes
    sd = scenes.cornell_box()
    p = scenes.cornell_box_sensor(24, 20, 2, seed=4)
    desc = ob.make_desc(p)
    S = ob.OracleScene(sd, naive=True)
    rows = mdist.owned_rows(p["height"], rank, world, tile_rows)
    part = mdist.film_partition(rank, world, tile_rows)
    assert part == (rank, world, tile_rows)
    film = np.zeros((p["height"], p["width"], 5), np.float32)
    # contiguous runs of owned rows -> oracle restricted renders
    run_start = None
    for r in rows + [None]:
        if run_start is None:
            run_start, prev = r, r
        elif r is not None and r == prev + 1:
            prev = r
        else:
            film += S.render_rows(desc, run_start, prev + 1)
            run_start, prev = r, r
    t = torch.from_numpy(film)
    mdist.reduce_film(t, root=0, all_ranks=all_ranks)
    if rank == 0 or all_ranks:
        np.save(os.path.join(out_dir, "film_rank%d.npy" % rank), t.numpy())
    np.save(os.path.join(out_dir, "rows_rank%d.npy" % rank), np.array(rows))
    dist.destroy_process_group()


@pytest.mark.parametrize("all_ranks", [False, True])
def test_film_partition_and_reduce_gloo(tmp_path, oracle, all_ranks):
    world, tile_rows = 2, 4
    mp.spawn(_worker, args=(world, _free_port(), tile_rows, str(tmp_path), all_ranks), nprocs=world, join=True)
    from mitsuba2_amd import scenes
    p = scenes.cornell_box_sensor(24, 20, 2, seed=4)
    ref, _ = oracle.OracleScene(scenes.cornell_box(), naive=True).render(oracle.make_desc(p), mode=1)
    rows = [np.load(tmp_path / ("rows_rank%d.npy" % r)) for r in range(world)]
    assert sorted(np.concatenate(rows).tolist()) == list(range(20))            # a partition: every row exactly once
    assert rows[0].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19]
    film0 = np.load(tmp_path / "film_rank0.npy")
    assert np.allclose(film0, ref, rtol=1e-5, atol=1e-6)
    if all_ranks:
        assert (np.load(tmp_path / "film_rank1.npy") == film0).all()


def test_owned_rows_edge_cases():
    from mitsuba2_amd import dist as mdist
    assert mdist.film_partition(0, 1) is None
    assert mdist.owned_rows(10, 0, 1) == list(range(10))
    # more ranks than tiles: trailing ranks own nothing
    assert mdist.owned_rows(40, 3, 8, 32) == [] and mdist.owned_rows(40, 1, 8, 32) == list(range(32, 40))
    for h, w, tr in ((1080, 8, 32), (33, 4, 32), (4096, 8, 32)):
        allr = sorted(r for k in range(w) for r in mdist.owned_rows(h, k, w, tr))
        assert allr == list(range(h))
    # balance on the headline config: 1024 rows over 8 ranks in 16-row tiles -> 128 rows each
    assert {len(mdist.owned_rows(1024, k, 8)) for k in range(8)} == {128}


def _bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (the driver's form) must start N ranks itself or refuse -- never report one rank as N GPUs.
    The launcher and the rank-side collectives (max / sum / gather over gloo) run here without a GPU."""
    import json
    r = _bench("--gpus", "2", "--backend", "gloo", "--config", "launcher-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [0.0, 1.0] and line["sum"] == 3.0 and line["max"] == 1.0
    if torch.cuda.device_count() < 2:
        # fewer GPUs than ranks on the real backend: refused with a non-zero exit code, nothing printed on stdout
        r = _bench("--gpus", "2")
        assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""
    # a torchrun environment whose world size disagrees with --gpus is refused too
    r = _bench("--gpus", "2", "--backend", "gloo", "--config", "launcher-selftest", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr
