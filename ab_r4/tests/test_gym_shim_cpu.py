"""The repo-root `marinenav_env` import shim: with a gym module present, importing it registers the facade under the id
the reference uses (marinenav_env/__init__.py:3-6), so `gym.make('marinenav_env:marinenav_env-v0', ...)` resolves."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_registers_facade_with_gym():
    code = textwrap.dedent("""
        import importlib, sys, types
        calls = []
        gym = types.ModuleType("gym")
        class Env:
            def close(self): pass
        class _Space:
            def __init__(self, *a, **k): pass
        spaces = types.ModuleType("gym.spaces"); spaces.Discrete = _Space; spaces.Box = _Space
        envs = types.ModuleType("gym.envs"); reg = types.ModuleType("gym.envs.registration")
        reg.register = lambda **kw: calls.append(kw)
        envs.registration = reg; gym.Env, gym.spaces, gym.envs = Env, spaces, envs
        sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.envs": envs, "gym.envs.registration": reg})
        sys.path.insert(0, %r)
        import marinenav_env                       # what gym.make('marinenav_env:...') imports first
        assert marinenav_env.REGISTERED
        ids = [c["id"] for c in calls]
        assert "marinenav_env-v0" in ids, ids
        mod, cls = calls[-1]["entry_point"].split(":")
        C = getattr(importlib.import_module(mod), cls)
        assert C is marinenav_env.MarineNavEnv and issubclass(C, Env)
        import marinenav_env.envs.marinenav_env as m       # run_experiments.py:10
        assert m.MarineNavEnv is C and marinenav_env.envs.MarineNavEnv is C
        for name in ("reset", "step", "reset_with_eval_config", "episode_data", "save_episode", "seed", "close",
                     "get_state_space_dimension", "get_action_space_dimension", "get_velocity", "get_observation",
                     "check_collision", "check_reach_goal", "out_of_boundary", "dist_to_goal"):
            assert callable(getattr(C, name)), name
        print("ok")
    """ % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
