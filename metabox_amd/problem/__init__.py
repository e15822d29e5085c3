"""Problem sets: BBOB / noisy BBOB instances (generated exactly like the reference) and the protein-docking energies."""
from . import basic_problem as _basic, bbob as _bbob, protein_docking as _protein

Basic_Problem = _basic.Basic_Problem
BBOB_Problem, BBOB_Dataset = _bbob.BBOB_Problem, _bbob.BBOB_Dataset
Protein_Docking, Protein_Docking_Dataset = _protein.Protein_Docking, _protein.Protein_Docking_Dataset
__all__ = ['Basic_Problem', 'BBOB_Problem', 'BBOB_Dataset', 'Protein_Docking', 'Protein_Docking_Dataset']
