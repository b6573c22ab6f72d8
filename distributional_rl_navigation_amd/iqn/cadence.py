"""What fires at a learning step of the IQN loop -- ONE statement of the rule for every loop shape.

The reference's loop (thirdparty/IQN/agent.py:126-147) trains every `UPDATE_EVERY` learning steps once the buffer holds more than a
batch, hard-copies the target network every `target_update_interval` learning steps, and evaluates + checkpoints every `eval_freq`
learning steps; all three are tested at the CURRENT `learning_timestep`, which is then advanced.  The single-env adapter
(`iqn/compat.py: learn`), the batched loop (`iqn/agent.py: vec_step`, `learn_vec`) all ask
`cadence_tick`; the batched loops add a target cadence counted in gradient steps (`target_sync_grad_steps`, `train_iqn.plan_cadence`).
"""
from collections import namedtuple

Tick = namedtuple("Tick", "train sync evaluate")
IDLE = Tick(False, False, False)


def cadence_tick(agent, train_every=None, eval_freq=None):
    """(train, sync, evaluate) for `agent.learning_timestep`; all False before `learning_starts`.  The caller performs what is due
    (training first: `sync` under the gradient-step cadence already counts the steps that `train` is about to add) and then advances
    `learning_timestep`."""
    if agent.current_timestep < agent.learning_starts:
        return IDLE
    at = agent.learning_timestep
    every = agent.UPDATE_EVERY if train_every is None else train_every
    train = at % every == 0 and len(agent.memory) > agent.BATCH_SIZE
    if agent.target_sync_grad_steps is None:
        sync = at % agent.target_update_interval == 0
    else:
        done_after = agent.grad_steps + (agent.grad_steps_per_update if train else 0)
        sync = at == 0 or done_after - agent._last_sync_at >= agent.target_sync_grad_steps
    return Tick(train, sync, bool(eval_freq) and at % eval_freq == 0)
