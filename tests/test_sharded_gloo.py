"""CPU, world_size 2, gloo: the N > 1 path of bench.py - round-robin image sharding and the
all_gather of padded (count, LAFs, responses, descriptors) records - reassembles results in global
image order.  (On the GPU node the same code runs on backend "nccl" = RCCL over xGMI.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from affnet_amd import sharded

N_CAP, N_IMG = 7, 5


def _fake_result(img_idx):
    g = torch.Generator().manual_seed(100 + img_idx)
    n = 1 + img_idx % N_CAP
    r = {"count": torch.tensor([n], dtype=torch.int32), "LAFs": torch.zeros(N_CAP, 2, 3), "responses": torch.zeros(N_CAP),
         "descriptors": torch.zeros(N_CAP, 128)}
    r["LAFs"][:n] = torch.rand(n, 2, 3, generator=g)
    r["responses"][:n] = torch.rand(n, generator=g)
    r["descriptors"][:n] = torch.rand(n, 128, generator=g)
    return r


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharded.shard_indices(N_IMG, rank, world)
    rec = sharded.pack_records([_fake_result(i) for i in mine], N_CAP, torch.device("cpu"))
    full = sharded.gather_features(rec, N_IMG)
    ok = full.shape == (N_IMG, 1 + N_CAP * 135)
    ok &= torch.equal(sharded.gather_features_async(rec, N_IMG)(), full)     # the overlapped form bench.py uses
    to0 = sharded.gather_features_async(rec, N_IMG, dst=0)()                  # gather to rank 0 only (--gather rank0)
    ok &= (to0 is None) if rank != 0 else torch.equal(to0, full)
    ok &= sharded.record_counts(full).dtype == torch.int32 and sharded.record_counts(full).tolist() == [1 + i % N_CAP for i in range(N_IMG)]
    for i in range(N_IMG):
        want, got = _fake_result(i), sharded.unpack_record(full[i], N_CAP)
        n = int(want["count"])
        ok &= got["LAFs"].shape[0] == n and torch.equal(got["LAFs"], want["LAFs"][:n])
        ok &= torch.equal(got["responses"], want["responses"][:n]) and torch.equal(got["descriptors"], want["descriptors"][:n])
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    assert sharded.shard_indices(5, 0, 2) == [0, 2, 4] and sharded.shard_indices(5, 1, 2) == [1, 3]
    assert sharded.shard_indices(0, 0, 2) == []
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_single_process_passthrough_and_generators_agree():
    rec = sharded.pack_records([_fake_result(0)], N_CAP, torch.device("cpu"))
    assert sharded.gather_features(rec, 1) is rec
    # batched results (the fused path returns (B, cap, ...) tensors per library call) pack to the same records
    rs = [_fake_result(i) for i in range(5)]
    stack = lambda lst: {k: (torch.cat([r[k] for r in lst]) if k == "count" else torch.stack([r[k] for r in lst])) for k in lst[0]}
    batched = sharded.pack_batched_records([stack(rs[:3]), stack(rs[3:]), ], N_CAP)
    assert torch.equal(batched, sharded.pack_records(rs, N_CAP, torch.device("cpu")))
    assert torch.equal(sharded.pack_batched_records([rs[0]], N_CAP), rec)
    import affnet_oracle as orc
    from affnet_amd.synthetic import synthetic_image, synthetic_hardnet_state
    assert torch.equal(synthetic_image(48, 64, 3), orc.synthetic_image(48, 64, 3))
    a, b = synthetic_hardnet_state(0), orc.synthetic_hardnet_state(0)
    assert all(torch.equal(a[k], b[k]) for k in b) and set(a) == set(b)


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    _bench.last_stdout = p.stdout
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def _assert_compact(stdout):
    """the LAST stdout line is the one JSON line, under 4 KB, with the contract's keys; details go to the side file."""
    import json
    last = stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < 4096, len(last)
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "detail"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    return d


@pytest.mark.parametrize("gather", ["all", "rank0"])
def test_bench_self_spawns_n_ranks(gather):
    """`python bench.py --gpus 2` (the shape of the driver's command, no launcher, no WORLD_SIZE): bench.py starts the 2 ranks itself,
    they rendezvous (gloo here, RCCL on the GPU node), exchange the padded records and rank 0 prints ONE line with n_gpus = 2."""
    rc, out, err = _bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--batch", "3", "--dry-run", "--gather", gather)
    assert rc == 0, err
    assert len(out) == 1, out
    assert out[0]["n_gpus"] == 2 and out[0]["config"]["global_batch"] == 6 and out[0]["records_in_global_order"] is True
    assert out[0]["exchange"]["mode"] == ("all_gather" if gather == "all" else "gather_rank0")
    _assert_compact(_bench.last_stdout)
    assert out[0]["exchange"]["bytes_per_step"] == 6 * (4 + 4 * 5 * (6 + 1 + 128)) * (2 if gather == "all" else 1)
    assert set(out[0]["ms_per_step_per_rank"]) == {"min", "max"}


def test_bench_eight_ranks_config5_dry_run():
    """`python bench.py --gpus 8 --config5` (BASELINE.json configs[4] on the whole node) through the self-spawn / rendezvous / gather /
    JSON path: 8 ranks, the 4K geometry's defaults (8 images per rank per step), one line."""
    rc, out, err = _bench("--gpus", "8", "--config5", "--steps", "1", "--warmup", "0", "--dry-run")
    assert rc == 0, err
    assert len(out) == 1, out
    assert out[0]["n_gpus"] == 8 and out[0]["config"]["global_batch"] == 64 and out[0]["records_in_global_order"] is True
    assert "8000 kp @3840x2160" in out[0]["metric"]
    _assert_compact(_bench.last_stdout)


def test_compact_line_from_the_recorded_round5_record():
    """The record round 5's bench printed as ONE 31 KB line (profiles/archive/r05_s1_bench_default.json: the driver's `parsed` was null) through
    today's formatter: under 4 KB, valid JSON, carrying `roofline`, `cpu_baseline`, the parity scalars and the co-reported configurations;
    and a synthetic record with every optional section blown up still fits."""
    import glob
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = sorted(glob.glob(os.path.join(root, "profiles", "**", "r05_s1_bench_default.json"), recursive=True))
    assert rec, "the recorded round-5 line is tracked under profiles/"
    full = json.loads(open(rec[0]).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    text = bench.compact_line(full)
    assert len(text) < 4096 and "\n" not in text
    d = json.loads(text)
    assert d["value"] == pytest.approx(full["value"], rel=1e-5) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert d["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5) and d["roofline"]["bound"] == "mfma" and "traffic" in d["roofline"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"] and d["cpu_baseline"]["node_value"] is not None
    assert d["parity"]["pass"] is True and d["parity"]["unmatched_unexplained"] == 0
    assert d["other"]["config2_graph_ms"] == pytest.approx(full["other_configs"]["config2_graph_ms"], rel=1e-5)
    assert d["other"]["split2h_value"] > d["other"]["split3_value"] > d["value"]
    # worst case: long strings everywhere, N = 8 sections present
    fat = dict(full, exchange={"mode": "gather_rank0", "exchange_bytes_per_step": 1 << 29, "gather_ms": 1.25, "how": "x" * 5000},
               ms_per_step_per_rank={"min": 1.0, "max": 2.0, "all": [1.0] * 8}, gather_check={"identical": True, "checked": 16, "records": 512, "mode": "sample", "what": "y" * 5000})
    fat["config"] = dict(full["config"], workload="w" * 5000, parallelism="p" * 100)
    fat["cpu_baseline"] = dict(full["cpu_baseline"], sample="s" * 5000)
    fat["roofline"] = dict(full["roofline"], kernel="k" * 5000)
    t2 = bench.compact_line(fat)
    assert len(t2) < 4096 and json.loads(t2)["exchange"]["bytes_per_step"] == 1 << 29


def test_gather_to_rank_in_a_subgroup_uses_global_ranks():
    """_gather_to_rank_async compared the group-LOCAL rank with the GLOBAL dst.  World of 3, sub-group {1, 2}, gather to
    global rank 2: rank 2 must receive both records in order, rank 1 gets None, rank 0 is not involved."""
    import torch.multiprocessing as mp
    mp.spawn(_subgroup_worker, args=(3, _free_port()), nprocs=3, join=True)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _subgroup_worker(rank, world, port):
    import torch.distributed as dist
    from affnet_amd import sharded
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group([1, 2])
    if rank in (1, 2):
        rec = torch.full((1, 4), float(rank))
        got = sharded.gather_features_async(rec, 2, group=grp, dst=2)()
        if rank == 2:
            assert got is not None and got[:, 0].tolist() == [1.0, 2.0], got
        else:
            assert got is None
        try:
            sharded.gather_features_async(rec, 2, group=grp, dst=0)
            raise AssertionError("a destination outside the group must be refused")
        except ValueError:
            pass
    dist.barrier()
    dist.destroy_process_group()


def test_bench_refuses_wrong_world_or_missing_devices():
    # a launcher-provided WORLD_SIZE that disagrees with --gpus must not silently report n_gpus = WORLD_SIZE
    rc, out, err = _bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and not out and "WORLD_SIZE" in err
    # the real (non dry-run) path on a host with fewer devices than --gpus fails loudly instead of running one rank
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        rc, out, err = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
        assert rc == 2 and not out and "visible GPUs" in err
