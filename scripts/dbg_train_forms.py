import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, numpy as np, time
from distributional_rl_navigation_amd import _capi
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from test_iqn_gpu import _random_batch
dev = "cuda:0"
L = _capi.lib()
def words(ft, batch):
    i = L.mn_iqn_train_workspace_status_word(batch)
    return ft._ws_by_batch[batch][i:i + 4].view(torch.int32).tolist()
for name, two, one, multi in (("three", False, False, False), ("two", True, False, False), ("fused", True, True, False), ("multi", True, True, True)):
    ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=2048, device=dev, seed=11)
    ag.two_launch_step, ag.one_launch_step, ag.use_multi_step = two, one, multi
    g = torch.Generator(device=dev); g.manual_seed(5)
    ag.memory.add_batch(*_random_batch(torch, 2048, g))
    ft0 = ag._fused_trainer()
    ws0 = ft0._workspace(256)
    TDQ = (128 * 35788 + 128 + 280 + 31) // 32 * 32
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.time()
    ls = []
    for ev in range(4):
        t1 = time.time()
        ls.append(float(ag.train_steps_from_memory((4, 2, 3, 16)[ev])))
        torch.cuda.synchronize()
        print("   event", ev, "%.4f s" % (time.time() - t1), words(ag._fused, 256), [round(float(x), 5) for x in ag._fused.losses[:4]])
    ft = ag._fused
    j = L.mn_iqn_train_workspace_floats(256) - 256 * 72 - 256
    d = ft._ws_by_batch[256][j:j + 256].view(torch.int32).tolist()
    chg = []
    
    print("debug records:", d[0])
    for r in range(min(30, d[0])):
        pb_, j_, vbs_, nvb_, nphys_, tid_, on_ = d[1 + 8 * r: 1 + 8 * r + 7]
        print("   pb", pb_, "j", j_, "vbs", vbs_, "nvb", nvb_, "n_phys", nphys_, "tid", tid_, "on", on_)
    print(name, "launches", ft.launches_per_step(256), "losses", ls, "status words", words(ft, 256), "misplaced", ft.xcd_misplaced(256), "%.2f s" % (time.time() - t0),
          "psum", float(ft.local.double().abs().sum()), flush=True)
