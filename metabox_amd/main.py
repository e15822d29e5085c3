"""Command-line entry point (reference: src/main.py:10-89): exactly one of --train / --rollout / --test /
--run_experiment.  `python -m metabox_amd.main --test --problem bbob --agent_load_dir ... --agent_for_cp RLEPSO_Agent
--l_optimizer_for_cp RLEPSO_Optimizer`.  --mgd_test / --mte_test are post-processing of result pickles and are not part
of this build."""
import os
import shutil

import torch

from .config import get_config
from .tester import Tester, rollout
from .trainer import Trainer


def main(argv=None):
    config = get_config(argv)
    modes = [config.train, config.rollout, config.test, config.run_experiment, config.mgd_test, config.mte_test]
    assert sum(m is not None for m in modes) == 1, \
        'Among train, rollout, test, run_experiment, mgd_test & mte_test, only one mode can be given at one time.'
    if config.mgd_test or config.mte_test:
        raise NotImplementedError('mgd_test / mte_test post-process result pickles (src/tester.py:421-608); out of scope here.')
    if config.train:
        torch.set_grad_enabled(True)
        Trainer(config).train()
    if config.rollout:
        torch.set_grad_enabled(False)
        rollout(config)
    if config.test:
        torch.set_grad_enabled(False)
        Tester(config).test()
    if config.run_experiment:                      # train -> rollout -> test (main.py:40-80)
        torch.set_grad_enabled(True)
        Trainer(config).train()
        agent_save_dir = config.agent_save_dir
        rollout_save_dir = os.path.join(agent_save_dir, config.train_agent) + '/'
        os.makedirs(rollout_save_dir, exist_ok=True)
        for fn in os.listdir(agent_save_dir):
            if os.path.isfile(os.path.join(agent_save_dir, fn)):
                shutil.copy(os.path.join(agent_save_dir, fn), rollout_save_dir)
        test_agent_load_dir = config.agent_load_dir
        config.agent_load_dir = agent_save_dir
        config.agent_for_rollout = [config.train_agent]
        config.optimizer_for_rollout = [config.train_optimizer]
        torch.set_grad_enabled(False)
        rollout(config)
        shutil.rmtree(rollout_save_dir)
        if test_agent_load_dir is not None:
            config.agent_load_dir = test_agent_load_dir
        test_model_file = os.path.join(config.agent_load_dir, f'{config.train_agent}.pkl')
        shutil.copy(os.path.join(agent_save_dir, f'checkpoint{config.n_checkpoint}.pkl'), test_model_file)
        if config.train_agent != config.agent and config.train_agent not in config.agent_for_cp:
            config.agent_for_cp.append(config.train_agent)
        if config.train_optimizer != config.optimizer and config.train_optimizer not in config.l_optimizer_for_cp:
            config.l_optimizer_for_cp.append(config.train_optimizer)
        Tester(config).test()
        if test_agent_load_dir is None:
            os.remove(test_model_file)


if __name__ == '__main__':
    main()
