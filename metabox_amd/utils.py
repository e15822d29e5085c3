"""Problem-set construction (reference: src/utils.py:4-27)."""
from .problem import bbob, protein_docking


def construct_problem_set(config):
    problem = config.problem
    if problem in ['bbob', 'bbob-noisy']:
        return bbob.BBOB_Dataset.get_datasets(suit=config.problem, dim=config.dim, upperbound=config.upperbound,
                                              train_batch_size=config.train_batch_size,
                                              test_batch_size=config.test_batch_size, difficulty=config.difficulty)
    if problem in ['protein']:
        return protein_docking.Protein_Docking_Dataset.get_datasets(version=problem, train_batch_size=config.train_batch_size,
                                                                    test_batch_size=config.test_batch_size,
                                                                    difficulty=config.difficulty)
    if problem in ['bbob-torch', 'bbob-noisy-torch', 'protein-torch']:
        raise NotImplementedError(f'{problem}: the autograd problem twins are only needed by L2L / RNN-OI, which are outside '
                                  f'the accelerated path (SURVEY.md §2).')
    raise ValueError(problem + ' is not defined!')
