"""Command-line configuration — same flags and derived fields as the reference's ``get_config``
(src/config.py:5-109), plus the batch-engine options ``--n_instances`` / ``--runs`` / ``--fixed_horizon``.

Derived: maxFEs = 2000*dim (protein: dim 12, maxFEs 1000, n_logpoint 5), log_interval = maxFEs // n_logpoint,
save_interval = max_learning_step // n_checkpoint, run-stamped log / checkpoint directories, and the two
always-present comparison baselines DEAP_CMAES and Random_search.
"""
import argparse
import time

_SUITES = ['bbob', 'bbob-noisy', 'bbob-torch', 'bbob-noisy-torch', 'protein', 'protein-torch']
_MODES = ('train', 'test', 'rollout', 'run_experiment', 'mgd_test', 'mte_test')


def build_parser():
    p = argparse.ArgumentParser()
    a = p.add_argument
    # common
    a('--problem', default='bbob', choices=_SUITES, help='specify the problem suite')
    a('--dim', type=int, default=10, help='dimension of search space')
    a('--upperbound', type=float, default=5, help='upperbound of search space')
    a('--difficulty', default='easy', choices=['easy', 'difficult'], help='difficulty level')
    a('--device', default='cpu', help='device of the policy networks')
    for m in _MODES:
        a('--' + m, default=None, action='store_true', help=f'switch to {m} mode')
    # training
    a('--max_learning_step', type=int, default=1500000, help='the maximum learning step for training')
    a('--train_batch_size', type=int, default=1, help='batch size of train set')
    a('--train_agent', default=None, help='agent for training')
    a('--train_optimizer', default=None, help='optimizer for training')
    a('--agent_save_dir', type=str, default='agent_model/train/', help='save your own trained agent model')
    a('--log_dir', type=str, default='output/', help='logging testing output')
    a('--draw_interval', type=int, default=3, help='interval epochs in drawing figures')
    a('--agent_for_plot_training', type=str, nargs='+', default=['RL_HPSDE_Agent'], help='learnable optimizer to compare')
    a('--n_checkpoint', type=int, default=20, help='number of training checkpoints')
    a('--resume_dir', type=str, help='directory to load previous checkpoint model')
    # testing
    a('--agent', default=None, help='None: traditional optimizer, else Learnable optimizer')
    a('--agent_load_dir', type=str, help='load your own agent model')
    a('--optimizer', default=None, help='your own learnable or traditional optimizer')
    a('--agent_for_cp', type=str, nargs='+', default=[], help='learnable optimizer to compare')
    a('--l_optimizer_for_cp', type=str, nargs='+', default=[], help='learnable optimizer to compare')
    a('--t_optimizer_for_cp', type=str, nargs='+', default=[], help='traditional optimizer to compare')
    a('--test_batch_size', type=int, default=1, help='batch size of test set')
    # rollout
    a('--agent_for_rollout', type=str, nargs='+', help='learnable agent for rollout')
    a('--optimizer_for_rollout', type=str, nargs='+', help='learnabel optimizer for rollout')
    a('--plot_smooth', type=float, default=0.8, help='smoothness of figure curves in [0, 1]')
    # zero-shot / transfer
    a('--problem_from', choices=_SUITES, help='source problem set in zero-shot and transfer learning')
    a('--problem_to', choices=_SUITES, help='target problem set in zero-shot and transfer learning')
    a('--difficulty_from', default='easy', choices=['easy', 'difficult'])
    a('--difficulty_to', default='easy', choices=['easy', 'difficult'])
    a('--model_from', type=str, help='the model trained on source problem set')
    a('--model_to', type=str, help='the model trained on target problem set')
    a('--pre_train_rollout', type=str, help='path of pre-train models rollout result .pkl file')
    a('--scratch_rollout', type=str, help='path of scratch models rollout result .pkl file')
    # batch engine (new)
    a('--n_instances', type=int, default=0, help='cap on instances stepped per kernel launch (0 = all problem x run pairs)')
    a('--test_runs', type=int, default=51, help='independent runs per problem in --test (reference: 51)')
    a('--rollout_runs', type=int, default=5, help='independent runs per problem and checkpoint in --rollout (reference: 5)')
    a('--fixed_horizon', default=False, action='store_true', help='disable the gbest<=1e-8 early stop')
    return p


def get_config(args=None):
    config = build_parser().parse_args(args)
    config.maxFEs = 2000 * config.dim
    config.bo_maxFEs = 10 * config.dim           # Bayesian optimisation gets a much smaller budget
    config.n_logpoint = 50
    if config.run_experiment and len(config.agent_for_cp) >= 1:
        assert config.agent_load_dir is not None, \
            "Option --agent_load_dir must be given since you specified option --agent_for_cp."
    if config.mgd_test or config.mte_test:
        config.problem = config.problem_to
        config.difficulty = config.difficulty_to
    if config.problem in ['protein', 'protein-torch']:
        config.dim = 12
        config.maxFEs = 1000
        config.bo_maxFEs = 10
        config.n_logpoint = 5
    config.run_time = f'{time.strftime("%Y%m%dT%H%M%S")}_{config.problem}_{config.difficulty}_{config.dim}D'
    for mode in ('test', 'rollout', 'mgd_test', 'mte_test'):
        setattr(config, f'{mode}_log_dir', f'{config.log_dir}/{mode}/{config.run_time}/')
    if config.train or config.run_experiment:
        config.agent_save_dir = config.agent_save_dir + config.train_agent + '/' + config.run_time + '/'
    config.save_interval = config.max_learning_step // config.n_checkpoint
    config.log_interval = config.maxFEs // config.n_logpoint
    for always in ('DEAP_CMAES', 'Random_search'):
        if always not in config.t_optimizer_for_cp:
            config.t_optimizer_for_cp.append(always)
    return config
