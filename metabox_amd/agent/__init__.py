from .basic_agent import Basic_Agent
from .rlepso_agent import RLEPSO_Agent
from .lde_agent import LDE_Agent
from .de_ddqn_agent import DE_DDQN_Agent
from .rl_pso_agent import RL_PSO_Agent
from .gleet_agent import GLEET_Agent
from .qlpso_agent import QLPSO_Agent
