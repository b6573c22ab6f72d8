"""Two ranks (two processes on ONE GPU) x 12 gradient steps of a shared learner through each gradient transport -- the all-reduce path (bucket over gloo),
the mailbox exchange inside the reduction + Adam launch ("mailbox"), fused with Adam only ("mailbox3"), as its own launch ("mailbox4") -- printing whether the ranks stayed bit-identical, the
granule timeouts and the checksum of the parameters (equal across the transports).  Debugging aid for tests/test_multigpu_paths_gpu.py."""
import os, sys, subprocess, socket, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_multigpu_paths_gpu as T
d = tempfile.mkdtemp()
script = os.path.join(d, "worker.py")
open(script, "w").write(T._TWO_RANK_WORKER)
for mode in ("mailbox", "mailbox3", "mailbox4", "collective"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    out = os.path.join(d, mode + ".pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import time; t0 = time.time()
    procs = [subprocess.Popen([sys.executable, script, str(r), port, out, ROOT, mode, "12"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    r = torch.load(out)
    print(mode, "rc", rcs, "same", r["same"], "timeouts", r["timeouts"], "%.1f s" % (time.time() - t0), float(r["params"].abs().sum()), flush=True)
