from .basic_problem import Basic_Problem
from .bbob import BBOB_Dataset, BBOB_Problem
