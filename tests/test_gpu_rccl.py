"""RCCL executed on the hardware there is: a world_size-1 `nccl` process group (backend "nccl" IS RCCL
on ROCm) on the MI355X, with the ensemble's REAL records and row blocks pushed through the very
collectives the 8-GPU job uses: `gather_records(device=...)`, `gather_ragged`,
`run_ensemble_device`, `run_ensemble_merged_sharded`.  A one-rank communicator still initialises RCCL,
builds the ring / tree, launches its device kernels and moves the buffers through them; what it cannot
show is xGMI traffic (the driver's SCALE run does)."""
import os
import socket

import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    yield dist, dev
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def test_backend_is_rccl(pg):
    import torch
    dist, dev = pg
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    assert torch.version.hip is not None  # "nccl" on a ROCm build of torch is RCCL
    t = torch.arange(8, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    out = [torch.empty_like(t)]
    dist.all_gather(out, t)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out[0].cpu().numpy(), np.arange(8.0))


def test_record_gather_through_rccl(pg, ctx):
    from dynesty_amd import ensemble
    dist, dev = pg
    prob = inputs.problem("G5")
    kw = dict(nlive=200, queue_size=64, walks=20, bound="multi", dlogz=0.5)
    t_coll = ensemble.run_ensemble_device(prob, 6, base_seed=5, world=1, rank=0, dist=dist, device=dev, **kw)
    t_none = ensemble.run_ensemble_device(prob, 6, base_seed=5, **kw)
    assert t_coll.shape == (6, len(ensemble.RECORD_FIELDS))
    np.testing.assert_array_equal(t_coll, t_none)  # the collective moves the records unchanged
    mean, se, n = ensemble.combine_logz(t_coll)
    assert n == 6 and abs(mean - prob.logz_truth) < 5 * se + 0.6


def test_ragged_gather_and_merge_through_rccl(pg, ctx):
    from dynesty_amd import ensemble
    dist, dev = pg
    rng = np.random.default_rng(0)
    arrays = [rng.random((n, 7)) for n in (3, 0, 11, 5)]
    got = ensemble.gather_ragged(arrays, 1, 0, dist=dist, device=dev)
    assert len(got) == 4
    for a, b in zip(arrays, got):
        np.testing.assert_array_equal(a, b)
    prob = inputs.problem("G5")
    kw = dict(nlive=200, queue_size=64, walks=20, bound="multi", dlogz=0.5)
    m_coll = ensemble.run_ensemble_merged_sharded(prob, 4, base_seed=9, world=1, rank=0, dist=dist, device=dev, **kw)
    m_none = ensemble.run_ensemble_merged_sharded(prob, 4, base_seed=9, **kw)
    for key in ("logl", "logwt", "logz", "samples_u", "samples_id", "samples_it", "ncall"):
        np.testing.assert_array_equal(m_coll[key], m_none[key])
    assert abs(m_coll.logz[-1] - prob.logz_truth) < 0.8


def test_bench_under_a_preset_visible_device_list():
    """A scheduler that pre-sets HIP_VISIBLE_DEVICES renumbers the devices from 0 for torch and for libdynhip alike:
    the bench (own process: LOCAL_RANK -> torch.cuda.set_device -> dh_create) runs and gathers through RCCL there;
    a rank without a visible device of its own says so."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--lean", "--steps", "3", "--warmup", "1", "--preroll", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["value"] > 0
    bad = subprocess.run(cmd, env=dict(env, LOCAL_RANK="1", RANK="0", WORLD_SIZE="1"), capture_output=True, text=True,
                         timeout=600)
    assert bad.returncode != 0 and "only 1 device(s) are visible" in bad.stderr
