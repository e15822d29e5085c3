"""Run configuration: the reference's command-line surface (src/config.py:5-109) rebuilt from a flag table.

Every option name, type and default of MetaBox's ``get_config`` is kept so existing command lines keep working; the
derived fields (maxFEs, n_logpoint, log_interval, save_interval, run-stamped directories, the two comparison baselines
that are always appended) follow the same rules.  Engine-specific additions: --n_instances, --ddqn_policy, --test_runs,
--rollout_runs, --fixed_horizon.
"""
import argparse
import time

SUITES = ('bbob', 'bbob-noisy', 'bbob-torch', 'bbob-noisy-torch', 'protein', 'protein-torch')
MODES = ('train', 'test', 'rollout', 'run_experiment', 'mgd_test', 'mte_test')
LEVELS = ('easy', 'difficult')

# (flag, kwargs) — grouped as in the reference's parser
_FLAGS = [
    # problem / device
    ('problem', dict(default='bbob', choices=SUITES)),
    ('dim', dict(type=int, default=10)),
    ('upperbound', dict(type=float, default=5)),
    ('difficulty', dict(default='easy', choices=LEVELS)),
    ('device', dict(default='cpu')),
    # training
    ('max_learning_step', dict(type=int, default=1500000)),
    ('train_batch_size', dict(type=int, default=1)),
    ('train_agent', dict(default=None)),
    ('train_optimizer', dict(default=None)),
    ('agent_save_dir', dict(type=str, default='agent_model/train/')),
    ('log_dir', dict(type=str, default='output/')),
    ('draw_interval', dict(type=int, default=3)),
    ('agent_for_plot_training', dict(type=str, nargs='+', default=['RL_HPSDE_Agent'])),
    ('n_checkpoint', dict(type=int, default=20)),
    ('resume_dir', dict(type=str)),
    # testing
    ('agent', dict(default=None)),
    ('agent_load_dir', dict(type=str)),
    ('optimizer', dict(default=None)),
    ('agent_for_cp', dict(type=str, nargs='+', default=[])),
    ('l_optimizer_for_cp', dict(type=str, nargs='+', default=[])),
    ('t_optimizer_for_cp', dict(type=str, nargs='+', default=[])),
    ('test_batch_size', dict(type=int, default=1)),
    # rollout
    ('agent_for_rollout', dict(type=str, nargs='+')),
    ('optimizer_for_rollout', dict(type=str, nargs='+')),
    ('plot_smooth', dict(type=float, default=0.8)),
    # zero-shot (MGD) / transfer (MTE)
    ('problem_from', dict(choices=SUITES)),
    ('problem_to', dict(choices=SUITES)),
    ('difficulty_from', dict(default='easy', choices=LEVELS)),
    ('difficulty_to', dict(default='easy', choices=LEVELS)),
    ('model_from', dict(type=str)),
    ('model_to', dict(type=str)),
    ('pre_train_rollout', dict(type=str)),
    ('scratch_rollout', dict(type=str)),
    # batch engine
    ('n_instances', dict(type=int, default=0)),      # cap on instances per kernel launch (0: the whole problem x run table)
    ('ddqn_policy', dict(type=str, default='hip', choices=['hip', 'torch'])),   # DE-DDQN rollouts: Q-network + argmax as one MFMA launch (mbx_ddqn_qnet) or the PyTorch module
    ('test_runs', dict(type=int, default=51)),       # tester.py:196
    ('rollout_runs', dict(type=int, default=5)),     # tester.py:321
]
ALWAYS_COMPARED = ('DEAP_CMAES', 'Random_search')   # config.py:104-107


def build_parser():
    parser = argparse.ArgumentParser(description='MetaBBO rollout engine (MetaBox-compatible options)')
    for name, kw in _FLAGS:
        parser.add_argument('--' + name, **kw)
    for mode in MODES:
        parser.add_argument('--' + mode, default=None, action='store_true')
    parser.add_argument('--fixed_horizon', default=False, action='store_true')   # no gbest <= 1e-8 early stop
    return parser


def _derive(cfg):
    protein = cfg.problem in ('protein', 'protein-torch')
    cfg.maxFEs = 1000 if protein else 2000 * cfg.dim
    cfg.bo_maxFEs = 10 if protein else 10 * cfg.dim
    cfg.n_logpoint = 5 if protein else 50
    if protein:
        cfg.dim = 12
    cfg.log_interval = cfg.maxFEs // cfg.n_logpoint
    cfg.save_interval = cfg.max_learning_step // cfg.n_checkpoint
    cfg.run_time = time.strftime('%Y%m%dT%H%M%S') + f'_{cfg.problem}_{cfg.difficulty}_{cfg.dim}D'
    for kind in ('test', 'rollout', 'mgd_test', 'mte_test'):
        setattr(cfg, kind + '_log_dir', f'{cfg.log_dir}/{kind}/{cfg.run_time}/')
    if cfg.train or cfg.run_experiment:
        cfg.agent_save_dir = f'{cfg.agent_save_dir}{cfg.train_agent}/{cfg.run_time}/'
    cfg.t_optimizer_for_cp += [b for b in ALWAYS_COMPARED if b not in cfg.t_optimizer_for_cp]
    return cfg


def get_config(args=None):
    cfg = build_parser().parse_args(args)
    if cfg.run_experiment and cfg.agent_for_cp:
        assert cfg.agent_load_dir is not None, 'Option --agent_load_dir must be given since you specified option --agent_for_cp.'
    if cfg.mgd_test or cfg.mte_test:                 # zero-shot / transfer runs evaluate on the target suite
        cfg.problem, cfg.difficulty = cfg.problem_to, cfg.difficulty_to
    return _derive(cfg)
