"""BASELINE configs[3] (524 288 envs as 8 shards, independent learners) and configs[4] (shared IQN, RCCL gradient
all-reduce, CVaR 0.5 acting) exercised on ONE GPU: every code path of the multi-GPU configurations that does not need
a second device runs here -- the fused HIP gradient step under a real `nccl` (RCCL) process group, the same step
across two ranks (two processes sharing the GPU, gloo transport), shard == slice at configs[3]'s full size, and
bench.py's --shared-learner line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _batch(torch, seed, B, dev):
    g = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(a).to(dev)
    return (t(g.normal(0, 5, (B, 26)).astype(np.float32)), t(g.randint(9, size=(B, 1)).astype(np.int64)),
            t(g.normal(0, 3, (B, 1)).astype(np.float32)), t(g.normal(0, 5, (B, 26)).astype(np.float32)),
            t((g.uniform(size=(B, 1)) < 0.2).astype(np.float32)))


def _taus(torch, seed, B, dev):
    g = np.random.RandomState(1000 + seed)
    return (torch.from_numpy(g.uniform(size=(B, 8)).astype(np.float32)).to(dev),
            torch.from_numpy(g.uniform(size=(B, 8)).astype(np.float32)).to(dev))


def test_fused_step_under_nccl_world_size_1_is_bitwise_the_plain_step(torch):
    """`IQNAgent(distributed=True)` on an RCCL process group of one rank: mn_iqn_train_grad -> dist.all_reduce(flat
    gradient) -> mn_iqn_train_adam (iqn/fused_train.py) must give bit-identical losses / gradients / weights to the
    non-distributed fused step (sum over one rank, divided by 1)."""
    import torch.distributed as dist
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device(dev))
    try:
        agents = [IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=128, device=dev, seed=5, distributed=d) for d in (False, True)]
        for a in agents:
            assert a.use_fused_train
        for step in range(4):
            exp = _batch(torch, step, 64, dev)
            tt, tl = _taus(torch, step, 64, dev)
            losses = [float(a.train(exp, taus_target=tt, taus_local=tl)) for a in agents]
            assert losses[0] == losses[1]
            for p0, p1 in zip(agents[0].qnetwork_local.parameters(), agents[1].qnetwork_local.parameters()):
                assert torch.equal(p0, p1) and torch.equal(p0.grad, p1.grad)
        assert int(agents[1]._fused.step_dev) == 4
        # the replay-driven entry the training loop uses (sample on device -> gather from the ring -> step)
        for a in agents:
            a.memory.add_batch(*_batch(torch, 99, 128, dev))
            a._fused.rng_state.copy_(torch.tensor([7, 0], dtype=torch.int64))
        l0, l1 = (float(a.train_from_memory()) for a in agents)
        assert l0 == l1
    finally:
        dist.destroy_process_group()


_TWO_RANK_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[4]); sys.path.insert(0, os.path.join(sys.argv[4], "tests"))
from test_multigpu_paths_gpu import _batch, _taus
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
rank, port, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
exchange = sys.argv[5] if len(sys.argv) > 5 else "collective"
n_steps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
# "mailbox1": the exchange inside the ONE launch of the step (mn_iqn_train_step_xchg + MN_TRAIN_ONE_LAUNCH: the reduction + Adam role of the forward / backward
# launch publishes and gathers); "mailbox": inside the reduction + Adam launch (two launches per step); "mailbox4": reduction (publishes), mn_iqn_train_exchange,
# mn_iqn_train_adam -- what a device too small for the fused launches takes; "mailbox4p": the same THROUGH mn_iqn_train_step_xchg, by planning for an 8-CU device
mode = exchange
exchange = "mailbox" if exchange.startswith("mailbox") else exchange
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
dev = "cuda:0"
from distributional_rl_navigation_amd import _capi
# two ranks share this GPU: each plans its launches for half of it (a fused launch waits for its own workgroups, which must all find a CU next to the peer's)
_capi.lib().mn_iqn_train_set_cu_limit(8 if mode == "mailbox4p" else 120)
agent = IQNAgent(26, 9, BATCH_SIZE=32, BUFFER_SIZE=64, device=dev, seed=3, distributed=True, rank=rank)
agent.exchange = exchange      # "collective": the bucket travels over gloo; "mailbox": IPC-mapped mailboxes, gloo only carries the handles
agent.two_launch_step = mode != "mailbox4"
agent.one_launch_step = mode == "mailbox1"
assert agent.use_fused_train
for step in range(n_steps):
    tt, tl = _taus(torch, 10 * step + rank, 32, dev)
    agent.train(_batch(torch, 10 * step + rank, 32, dev), taus_target=tt, taus_local=tl)
flat = agent._fused.local.cpu()
gathered = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(gathered, flat)
timeouts = agent._fused.timeouts()
launches = agent._fused.launches_per_step(32) if exchange == "mailbox" else 0
kind = agent._fused._mailbox.memory_kind() if exchange == "mailbox" else ""
if rank == 0:
    torch.save(dict(params=flat, same=bool(torch.equal(gathered[0], gathered[1])), timeouts=timeouts, launches=launches, kind=kind), out)
dist.destroy_process_group()
"""


def test_fused_step_two_ranks_equals_big_batch(torch, tmp_path):
    """configs[4]'s learner with world size 2: two processes (both on this GPU; the 143 KB gradient bucket travels
    over gloo instead of RCCL, which refuses two ranks on one device) each run the fused HIP step on their own batch
    with the all-reduce between mn_iqn_train_grad and mn_iqn_train_adam.  The ranks must stay bit-identical, and
    equal -- to float32 summation order -- ONE fused step on the union of the two batches (clip after averaging)."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    out = str(tmp_path / "r0.pt"); port = str(_free_port())
    script = str(tmp_path / "worker.py")
    with open(script, "w") as f:
        f.write(_TWO_RANK_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, str(r), port, out, ROOT], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    res = torch.load(out)
    assert res["same"], "ranks diverged: all-reduced gradients must keep shared learners identical"
    dev = "cuda:0"
    ref = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=64, device=dev, seed=3)
    for step in range(3):
        b0, b1 = _batch(torch, 10 * step, 32, dev), _batch(torch, 10 * step + 1, 32, dev)
        (t0, l0), (t1, l1) = _taus(torch, 10 * step, 32, dev), _taus(torch, 10 * step + 1, 32, dev)
        ref.train(tuple(torch.cat([a, b]) for a, b in zip(b0, b1)), taus_target=torch.cat([t0, t1]), taus_local=torch.cat([l0, l1]))
    np.testing.assert_allclose(res["params"].numpy(), ref._fused.local.cpu().numpy(), rtol=0, atol=2e-6)


def _two_rank_run(torch, tmp_path, tag, exchange, n_steps):
    out = str(tmp_path / f"{tag}.pt"); port = str(_free_port())
    script = str(tmp_path / "worker.py")
    with open(script, "w") as f:
        f.write(_TWO_RANK_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, str(r), port, out, ROOT, exchange, str(n_steps)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return torch.load(out)


def test_mailbox_exchange_two_ranks_equals_the_all_reduce_path_bitwise(torch, tmp_path):
    """The one-shot gradient exchange (`mn_xchg_*`, iqn/mailbox.py): two ranks (two processes on this GPU) publish their reduced
    gradients into IPC-exported mailboxes from inside the reduction kernel and each sums both mailboxes in rank order with one gather
    kernel -- no collective.  Same batches through the all-reduce path (bucket over gloo): parameters bit-identical after 12 steps,
    ranks bit-identical to each other, no granule timed out."""
    a = _two_rank_run(torch, tmp_path, "collective", "collective", 12)
    b = _two_rank_run(torch, tmp_path, "mailbox", "mailbox", 12)        # two launches per step: the exchange inside the reduction + Adam launch
    c = _two_rank_run(torch, tmp_path, "mailbox1", "mailbox1", 12)      # ONE: the exchange inside the reduction + Adam role of the forward / backward launch
    d = _two_rank_run(torch, tmp_path, "mailbox4", "mailbox4", 12)      # four: reduction, mn_iqn_train_exchange, mn_iqn_train_adam
    e = _two_rank_run(torch, tmp_path, "mailbox4p", "mailbox4p", 12)    # the same four, chosen by the library's launch plan for a device of 8 CUs
    for r in (a, b, c, d, e):
        assert r["same"] and r["timeouts"] == 0
    assert (b["launches"], c["launches"], d["launches"], e["launches"]) == (2, 1, 4, 4)
    assert b["kind"] in ("uncached", "fine-grained"), b["kind"]      # the mailbox must be visible to a peer DEVICE inside a running kernel
    for r in (b, c, d, e):
        assert torch.equal(a["params"], r["params"])


_LATE_PEER_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[4]); sys.path.insert(0, os.path.join(sys.argv[4], "tests"))
from test_multigpu_paths_gpu import _batch, _taus
from distributional_rl_navigation_amd import _capi
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
rank, port, out, one = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[5] == "1"
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
_capi.lib().mn_iqn_train_set_cu_limit(120)
agent = IQNAgent(26, 9, BATCH_SIZE=32, BUFFER_SIZE=64, device="cuda:0", seed=3, distributed=True, rank=rank)
agent.exchange, agent.one_launch_step = "mailbox", one
def step(k):
    tt, tl = _taus(torch, 10 * k + rank, 32, "cuda:0")
    return agent.train(_batch(torch, 10 * k + rank, 32, "cuda:0"), taus_target=tt, taus_local=tl)
step(0); step(1)
agent._fused._mailbox.set_timeout_ms(300)
torch.cuda.synchronize(); dist.barrier()
res = None
if rank == 0:      # rank 1 stops here: the third step of rank 0 never gets its peer's gradient
    before = (agent._fused.local.clone(), agent._fused.exp_avg.clone(), agent._fused.exp_avg_sq.clone())
    step(2)
    torch.cuda.synchronize()
    raised = False
    try:
        agent.check_learner()
    except _capi.MarineNavHipError:
        raised = True
    res = dict(timeouts=agent._fused.timeouts(), mailbox_timeouts=agent._fused._mailbox.timeouts(), raised=raised,
               untouched=all(bool(torch.equal(a, b)) for a, b in zip(before, (agent._fused.local, agent._fused.exp_avg, agent._fused.exp_avg_sq))),
               grad_nan=bool(torch.isnan(agent._fused.grad).all()))
    torch.save(res, out)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("one_launch", [False, True])
def test_mailbox_exchange_peer_that_falls_behind_is_a_loud_error_not_a_silent_divergence(torch, tmp_path, one_launch):
    """a gather that runs into its bound (here 0.3 s; the peer simply does not take the step) must not apply an update from stale granules.  The rank
    that waited: every reduction + Adam block counted in the workspace's status word and in mn_xchg_status, parameters and both moments exactly as before the
    step, the step's gradient NaN, and `IQNAgent.check_learner` -- what `learn_vec` calls at its evaluation points -- raises.  Both fused forms."""
    out = str(tmp_path / "late.pt"); port = str(_free_port())
    script = str(tmp_path / "late_worker.py")
    with open(script, "w") as f:
        f.write(_LATE_PEER_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, str(r), port, out, ROOT, "1" if one_launch else "0"], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    r = torch.load(out)
    assert r["timeouts"] >= 70 and r["mailbox_timeouts"] >= 1 and r["raised"] and r["untouched"] and r["grad_nan"], r


_TWO_RANK_RING_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[4]); sys.path.insert(0, os.path.join(sys.argv[4], "tests"))
from test_multigpu_paths_gpu import _batch
from distributional_rl_navigation_amd import _capi
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
rank, port, out, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[5]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
_capi.lib().mn_iqn_train_set_cu_limit(120)      # two ranks share this GPU: each plans its fused launches for half of it
agent = IQNAgent(26, 9, BATCH_SIZE=32, BUFFER_SIZE=512, device="cuda:0", seed=3, distributed=True, rank=rank)
agent.exchange = "collective" if mode == "collective" else "mailbox"
agent.one_launch_step = True
agent.memory.add_batch(*_batch(torch, 100 + rank, 512, "cuda:0"))      # every rank its own replay ring (and its own sampling stream: IQNAgent(rank=...))
losses = []
for ev, G in enumerate((8, 8, 3, 1, 8)):
    if ev == 2:
        agent.memory.add_batch(*_batch(torch, 200 + rank, 100, "cuda:0"))
    losses.append(float(agent.train_steps_from_memory(G)))
flat = agent._fused.local.cpu()
gathered = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(gathered, flat)
res = dict(params=flat, same=bool(torch.equal(gathered[0], gathered[1])), timeouts=agent._fused.timeouts(), losses=losses, steps=int(agent._fused.step_dev),
           launches=agent._fused.launches_per_step(32))
if rank == 0:
    torch.save(res, out)
dist.barrier()
dist.destroy_process_group()
"""


def test_training_events_with_the_mailbox_exchange_two_ranks_bitwise(torch, tmp_path):
    """A shared learner's training events with the exchange inside every fused step (`mn_iqn_train_step_xchg`): two ranks (two processes on this GPU), each sampling its own
    replay ring, 28 gradient steps in events of 8 / 8 / 3 / 1 / 8 -- every step's reduction + Adam blocks publish into and gather from the two ranks' mailboxes inside
    the launch.  Against the all-reduce path (bucket over gloo): parameters and event losses bit-identical, ranks bit-identical to each other, no bounded wait ran out."""
    res = {}
    for mode in ("collective", "single"):
        out = str(tmp_path / f"{mode}.pt"); port = str(_free_port())
        script = str(tmp_path / "ring_worker.py")
        with open(script, "w") as f:
            f.write(_TWO_RANK_RING_WORKER)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script, str(r), port, out, ROOT, mode], env=env) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=900) == 0
        res[mode] = torch.load(out)
    for mode, r in res.items():
        assert r["same"] and r["timeouts"] == 0 and r["steps"] == 28, (mode, r["same"], r["timeouts"], r["steps"])
        assert all(np.isfinite(r["losses"]))
    assert res["single"]["launches"] == 1 and res["collective"]["launches"] == 3
    for mode in ("single",):
        assert res[mode]["losses"] == res["collective"]["losses"]
        assert torch.equal(res[mode]["params"], res["collective"]["params"])


def test_mailbox_import_that_the_runtime_refuses_is_a_clear_status(torch):
    """A peer's mailbox that cannot be mapped (here: a handle that is not one) is MN_ERR_PEER with a text that names the remedy -- not a bare HIP error code."""
    import ctypes as C
    from distributional_rl_navigation_amd import _capi
    L = _capi.lib()
    x = C.c_void_p()
    assert L.mn_xchg_create(0, 2, C.byref(x)) == 0
    try:
        assert L.mn_xchg_last_error(x) == b""
        rc = L.mn_xchg_import(x, 1, C.create_string_buffer(b"\x00" * 64, 64))
        assert rc == -5, rc                                                    # MN_ERR_PEER
        msg = L.mn_xchg_last_error(x).decode()
        assert "hipIpcOpenMemHandle" in msg and "rank 1" in msg and "--exchange collective" in msg, msg
        assert L.mn_xchg_import(x, 0, C.create_string_buffer(b"\x00" * 64, 64)) == -1      # own rank: MN_ERR_INVALID as before
    finally:
        assert L.mn_xchg_destroy(x) == 0


def test_mailbox_exchange_world_size_1_is_bitwise_the_plain_step_eager_and_graphed(torch):
    import torch.distributed as dist
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        runs = []
        for distributed, graphed, one, two in ((False, False, False, True), (True, False, False, True), (True, True, False, True),
                                               (True, False, True, True), (True, True, True, True), (True, False, False, False), (True, True, False, False)):
            ag = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=256, device=dev, seed=5, distributed=distributed)
            ag.exchange = "mailbox"
            ag.one_launch_step, ag.two_launch_step = one, two      # the exchange inside ONE launch / inside the reduction + Adam launch / as its own launch
            ag.use_fused_graph = graphed
            ag.memory.add_batch(*_batch(torch, 7, 300, dev))
            losses = [float(ag.train_steps_from_memory(8)) for _ in range(3)]
            runs.append((losses, ag._fused.local.clone(), ag._fused.exp_avg_sq.clone(), int(ag._fused.step_dev)))
            if distributed:
                assert ag._fused.timeouts() == 0
                assert ag._fused.launches_per_step(64) == (1 if one else (2 if two else 4))
        for r in runs[1:]:
            assert r[0] == runs[0][0] and torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2]) and r[3] == runs[0][3] == 24
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["f64", "mixed"])
def test_shards_equal_slices_at_config3_size(torch, precision):
    """BASELINE configs[3]: 524 288 envs as 8 shards of 65 536 (rank r: first_index = r * 65 536).  One 524 288-env
    handle vs the eight shard handles, same actions: worlds, observations, rewards, done flags and counters of shard
    r are bitwise rows [r n, (r+1) n) of the big run, through resets -- in float64 (what `bench.py --gpus N` runs with a learner
    in the loop, `bench.default_precision`) and in the kernel-only configs' mixed precision."""
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    n, world, T = 65536, 8, 6
    dev = "cuda:0"
    big = VecMarineNavEnv(n * world, seed=0, device=dev, precision=precision)
    big.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    ob = big.reset().clone()
    g = torch.Generator(device=dev); g.manual_seed(0)
    acts = [torch.randint(0, 9, (n * world,), device=dev, dtype=torch.int32, generator=g) for _ in range(T)]
    trace = []
    for t in range(T):
        o, r, d, i = big.step(acts[t])
        trace.append((o.clone(), r.clone(), d.clone(), i.clone()))
        big.reset_done()
    final_big = big.obs.clone()
    sb, epb, totb = big.get_state()
    wb = big.get_worlds(3 * n + 100, 16)
    big.close()
    total_done = 0
    for r_ in range(world):
        sh = VecMarineNavEnv(n, seed=0, first_index=r_ * n, device=dev, precision=precision)
        sh.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        sl = slice(r_ * n, (r_ + 1) * n)
        assert torch.equal(sh.reset(), ob[sl])
        for t in range(T):
            o, rew, d, i = sh.step(acts[t][sl])
            assert torch.equal(o, trace[t][0][sl]) and torch.equal(rew, trace[t][1][sl])
            assert torch.equal(d, trace[t][2][sl]) and torch.equal(i, trace[t][3][sl])
            total_done += int(d.sum())
            sh.reset_done()
        assert torch.equal(sh.obs, final_big[sl])
        s, ep, tot = sh.get_state()
        assert np.array_equal(s, sb[sl]) and np.array_equal(ep, epb[sl]) and np.array_equal(tot, totb[sl])
        if r_ == 3:
            for a, b in zip(sh.get_worlds(100, 16), wb):
                assert np.array_equal(a["cores"], b["cores"]) and np.array_equal(a["obstacles"], b["obstacles"])
        sh.close()
    assert total_done > 200       # resets happened inside the compared window


def _bench_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0]) < 6144, len(lines[0])      # the driver's record keeps 8 KB of tail: the whole line must fit
    return json.loads(lines[0])


@pytest.mark.parametrize("exchange", ["collective", "mailbox"])
def test_bench_shared_learner_line(torch, exchange):
    """`python bench.py --gpus 1 --shared-learner --cvar 0.5 [--exchange mailbox]` (configs[4] on one GPU: single-rank RCCL group, the
    all-reduce executes / the exchange rides in the step's launch) prints ONE JSON line with the contract's keys."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--shared-learner", "--cvar", "0.5", "--exchange", exchange,
                          "--envs", "4096", "--steps", "24", "--warmup", "8", "--windows", "2", "--cpu-steps", "0", "--no-learner-only"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    j = _bench_line(out)
    assert j["config"]["learner"] == "shared" and j["config"]["exchange"] == exchange and j["config"]["cvar"] == 0.5
    assert j["config"]["process_group"] == "nccl" and j["config"]["ablation"] is False
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["grad_steps_per_sec"] > 0 and j["timeouts"] == 0
    assert j["all_reduce_ms"] > 0 and j["windows"]["n"] == 2 and j["windows"]["min"] <= j["value"] <= j["windows"]["max"]
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] < 1
    # the episode resets ran under the act kernel (an RCCL communicator in the process: the handle's stream needs a hardware queue of its own) and every late row was served
    assert j["config"]["resets"] == "under_next_act" and j["config"]["late_row_timeouts"] == 0
    assert j["roofline_env_step"]["reset_kernel"]["on_critical_path"] is False and j["also"]["reset_in_front"]["value"] > 0
    # the late-curriculum leg: many episode ends per vector step (here 4 096 / 40 time-outs + collisions), still under the act kernel
    lc = j["also"]["late_curriculum"]
    assert lc["value"] > 0 and lc["reset_in_front"] > 0 and lc["resets_per_launch"] > 4096 / 40 and lc["under_act_share"] == 1.0 and lc["eps"] == 0.05


@pytest.mark.parametrize("config", ["c3", "c4", "c4m"])
def test_bench_world_size_2_branch_runs_under_torch_distributed_run(torch, config):
    """bench.py's N > 1 branch -- per-rank device, env shard `first_index = rank x n`, barrier-bracketed windows, MAX over ranks of the elapsed time, rank 0
    aggregating -- executed for real with TWO ranks, launched exactly as the driver launches N GPUs (`python -m torch.distributed.run --nproc-per-node 2 ...
    bench.py --gpus 2`), on this ONE GPU: `--ranks-per-gpu 2` puts both ranks on cuda:0 and forms the group over gloo (RCCL refuses two ranks per device; on a
    node every rank has its own GPU and the group is RCCL).  BASELINE configs[3] (independent learners, 65 536 envs per rank), configs[4] (shared IQN, CVaR 0.5;
    the gradient bucket is all-reduced -- over gloo here) and configs[4] with the mailbox exchange (the two ranks' gradient steps publish into and gather from
    each other's IPC-mapped mailboxes inside their launches).  One line from rank 0: n_gpus 2, value = both ranks' env steps over the slowest rank's time."""
    extra = {"c3": [], "c4": ["--shared-learner", "--cvar", "0.5"], "c4m": ["--shared-learner", "--cvar", "0.5", "--exchange", "mailbox"]}[config]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    steps = 12
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks-per-gpu", "2", "--steps", str(steps), "--warmup", "4", "--windows", "3",
           "--update-every", "2", "--cpu-steps", "0", "--no-learner-only", "--no-clock-probe"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    j = _bench_line(out)
    assert j["n_gpus"] == 2 and j["steps"] == steps and j["config"]["ranks_per_gpu"] == 2 and j["config"]["process_group"] == "gloo"
    assert j["config"]["envs_per_gpu"] == 65536 and j["scaling"] == "weak"
    # value = the env steps of BOTH ranks over the (median window's) slowest rank's time
    assert abs(j["value"] - 2 * 65536 * steps / (j["ms_per_step"] * 1e-3 * steps)) <= 2e-3 * j["value"]
    assert j["windows"]["n"] == 3 and j["windows"]["min"] <= j["value"] <= j["windows"]["max"]
    assert j["timeouts"] == 0 and j["grad_steps_per_sec"] > 0
    assert j["config"]["resets"] == "under_next_act" and j["config"]["late_row_timeouts"] == 0
    if config == "c3":
        assert j["config"]["learner"] == "independent" and j["all_reduce_ms"] is None
    else:
        assert j["config"]["learner"] == "shared" and j["config"]["cvar"] == 0.5 and j["all_reduce_ms"] > 0
        assert j["config"]["exchange"] == ("mailbox" if config == "c4m" else "collective")
    assert j["roofline"]["bound"] == "mfma" and j["roofline_env_step"]["launch_ms"] > 0


def test_graphed_fused_steps_equal_eager_steps_bitwise_plain_and_under_nccl(torch):
    """`IQNAgent.use_fused_graph`: the 8 gradient steps of a training event -- forward / backward, reduction, (shared learner) the RCCL
    all-reduce of the flat gradient, Adam, every step -- captured once and replayed as ONE hipGraph launch.  Counters (generator, Adam
    step, hand-off epoch) live on the device, so three replays continue where eager calls would: losses, parameters, moments,
    generator state bit-identical to 24 eager steps; checked without a process group and with a single-rank RCCL group, and across a
    write to the replay ring between two replays."""
    import torch.distributed as dist
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for distributed in (False, True):
        if distributed:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=torch.device(dev))
        try:
            runs = []
            for graphed in (False, True):
                ag = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=256, device=dev, seed=5, distributed=distributed)
                ag.use_fused_graph = graphed
                ag.memory.add_batch(*_batch(torch, 7, 300, dev))      # the ring is full (graphs are used from then on: its row count is constant)
                losses = []
                for ev in range(3):
                    if ev == 2:
                        ag.memory.add_batch(*_batch(torch, 8, 100, dev))      # ring written between two replays of the SAME graph
                    losses.append(float(ag.train_steps_from_memory(8)))
                ft = ag._fused
                runs.append((losses, ft.local.clone(), ft.exp_avg_sq.clone(), ft.rng_state.clone(), int(ft.step_dev), ag.grad_steps))
            (l0, p0, v0, r0, s0, g0), (l1, p1, v1, r1, s1, g1) = runs
            assert l0 == l1 and torch.equal(p0, p1) and torch.equal(v0, v1) and torch.equal(r0, r1)
            assert s0 == s1 == 24 and g0 == g1 == 24 and int(r0[1]) == 24
        finally:
            if distributed:
                dist.destroy_process_group()
