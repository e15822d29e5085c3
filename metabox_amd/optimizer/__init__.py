from .learnable_optimizer import Learnable_Optimizer
from .rlepso_optimizer import RLEPSO_Optimizer
from .lde_optimizer import LDE_Optimizer
from .de_ddqn_optimizer import DE_DDQN_Optimizer
from .basic_optimizer import Basic_Optimizer
from .random_search import Random_search
from .rl_pso_optimizer import RL_PSO_Optimizer
from .gleet_optimizer import GLEET_Optimizer
from .qlpso_optimizer import QLPSO_Optimizer
from .classic import DEAP_CMAES, DEAP_DE, DEAP_PSO
