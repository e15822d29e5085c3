"""Problem-set construction by suite name (reference: src/utils.py:4-27)."""


def _bbob(config):
    from .problem.bbob import BBOB_Dataset
    return BBOB_Dataset.get_datasets(suit=config.problem, dim=config.dim, upperbound=config.upperbound,
                                     train_batch_size=config.train_batch_size, test_batch_size=config.test_batch_size,
                                     difficulty=config.difficulty)


def _protein(config):
    from .problem.protein_docking import Protein_Docking_Dataset
    return Protein_Docking_Dataset.get_datasets(version=config.problem, train_batch_size=config.train_batch_size,
                                                test_batch_size=config.test_batch_size, difficulty=config.difficulty)


def _autograd_twin(config):
    raise NotImplementedError(f'{config.problem}: the autograd problem twins are only needed by L2L / RNN-OI, which are '
                              f'outside the accelerated path (SURVEY.md §2).')


_BUILDERS = {'bbob': _bbob, 'bbob-noisy': _bbob, 'protein': _protein,
             'bbob-torch': _autograd_twin, 'bbob-noisy-torch': _autograd_twin, 'protein-torch': _autograd_twin}


def construct_problem_set(config):
    """-> (train_set, test_set) for config.problem / config.difficulty."""
    try:
        build = _BUILDERS[config.problem]
    except KeyError:
        raise ValueError(config.problem + ' is not defined!')
    return build(config)
