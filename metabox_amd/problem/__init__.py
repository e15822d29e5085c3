from .basic_problem import Basic_Problem
from .bbob import BBOB_Dataset, BBOB_Problem
from .protein_docking import Protein_Docking, Protein_Docking_Dataset
