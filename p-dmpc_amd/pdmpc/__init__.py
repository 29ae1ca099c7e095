"""pdmpc — host-side mirror of p-dmpc's optimizer plug-in surface for the MI355X HIP backend."""
from .config import Config, ConstraintFromSuccessor, MpaType, OptimizerType, ScenarioType  # noqa: F401
