"""Command-line entry point with MetaBox's modes (reference: src/main.py:10-89).

    python -m metabox_amd.main --test --problem bbob --agent_load_dir models/ --agent_for_cp RLEPSO_Agent \\
                               --l_optimizer_for_cp RLEPSO_Optimizer --device cuda

Exactly one of --train / --rollout / --test / --run_experiment / --mgd_test / --mte_test may be given.
"""
import os
import shutil

import torch

from .config import MODES, get_config
from .tester import Tester, mgd_test, mte_test, rollout
from .trainer import Trainer


def _train(config):
    with torch.enable_grad():
        return Trainer(config).train()


def _rollout(config):
    with torch.no_grad():
        return rollout(config)


def _test(config):
    with torch.no_grad():
        return Tester(config).test()


def _experiment(config):
    """train -> rollout of the 21 checkpoints -> test of the final checkpoint (main.py:40-80)."""
    _train(config)
    save_dir = config.agent_save_dir
    staged = os.path.join(save_dir, config.train_agent) + '/'          # rollout() expects <load_dir>/<agent>/checkpointK.pkl
    os.makedirs(staged, exist_ok=True)
    for entry in os.scandir(save_dir):
        if entry.is_file():
            shutil.copy(entry.path, staged)
    user_load_dir = config.agent_load_dir
    config.agent_load_dir = save_dir
    config.agent_for_rollout, config.optimizer_for_rollout = [config.train_agent], [config.train_optimizer]
    _rollout(config)
    shutil.rmtree(staged)
    if user_load_dir is not None:
        config.agent_load_dir = user_load_dir
    final_model = os.path.join(config.agent_load_dir, config.train_agent + '.pkl')
    shutil.copy(os.path.join(save_dir, f'checkpoint{config.n_checkpoint}.pkl'), final_model)
    if config.train_agent != config.agent and config.train_agent not in config.agent_for_cp:
        config.agent_for_cp.append(config.train_agent)
    if config.train_optimizer != config.optimizer and config.train_optimizer not in config.l_optimizer_for_cp:
        config.l_optimizer_for_cp.append(config.train_optimizer)
    _test(config)
    if user_load_dir is None:
        os.remove(final_model)


def _mgd(config):
    with torch.no_grad():
        return mgd_test(config)


_DISPATCH = {'train': _train, 'rollout': _rollout, 'test': _test, 'run_experiment': _experiment, 'mgd_test': _mgd,
             'mte_test': mte_test}


def main(argv=None):
    config = get_config(argv)
    chosen = [m for m in MODES if getattr(config, m) is not None]
    assert len(chosen) == 1, 'Among train, rollout, test, run_experiment, mgd_test & mte_test, only one mode can be given at one time.'
    return _DISPATCH[chosen[0]](config)


if __name__ == '__main__':
    main()
