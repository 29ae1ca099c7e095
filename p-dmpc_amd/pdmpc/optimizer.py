"""OptimizerInterface / GraphSearchHip — the reference's optimizer plug-in surface.

Reference: hlc/optimizer/OptimizerInterface.m:1-105 (abstract `run_optimizer`, static `get_optimizer`,
`set_constraint_checker`) and hlc/optimizer/graph_search/GraphSearch.m:14-17.  `GraphSearchHip` is the
member a maintainer adds next to `GraphSearch`/`MonteCarloTreeSearch` (OptimizerType.HipOptimal); it keeps
the call signature `run_optimizer(veh_index, iter, mpa, options, time_step)` and returns a
`ControlResultsInfo`.  Exhaustion is a result (`is_exhausted`), not an error (GraphSearch.m:57-61).
"""
from typing import List

from . import abi
from .backend import Handle
from .config import Config, OptimizerType
from .iteration_data import ControlResultsInfo, VehicleIter, info_from_record


class OptimizerInterface:
    def run_optimizer(self, veh_index, iter_v: VehicleIter, mpa, options: Config, time_step) -> ControlResultsInfo:
        raise NotImplementedError  # OptimizerInterface.m:13-15 (Abstract)

    @staticmethod
    def get_optimizer(options: Config) -> "OptimizerInterface":
        """OptimizerInterface.m:19-34."""
        if options.optimizer_type == OptimizerType.HipOptimal:
            return GraphSearchHip(options)
        raise ValueError(
            "optimizer_type %s is implemented by the MATLAB reference only; this backend provides HipOptimal"
            % options.optimizer_type.name
        )


class GraphSearchHip(OptimizerInterface):
    """Optimal graph search on the GPU.  The constraint checker follows OptimizerInterface.set_constraint_checker
    (OptimizerInterface.m:36-46): InterX when options.are_any_obstacles_non_convex, SAT otherwise."""

    def __init__(self, options: Config):
        self.options = options
        self.handle = Handle(options)
        self._mpa_id = None

    def _ensure_mpa(self, mpa):
        if self._mpa_id != id(mpa):
            self.handle.upload_mpa(mpa)
            self._mpa_id = id(mpa)

    def run_optimizer(self, veh_index, iter_v, mpa, options, time_step):
        """GraphSearch.run_optimizer (GraphSearch.m:14-17): veh_index and time_step are ignored, as there."""
        assert iter_v.amount == 1  # are_constraints_satisfied_interx.m:12
        return self.run_optimizer_batch([iter_v], mpa)[0]

    def run_optimizer_batch(self, iters: List[VehicleIter], mpa) -> List[ControlResultsInfo]:
        """One computation level: every vehicle of PrioritizedSequentialController.m:85-91's inner loop at once."""
        self._ensure_mpa(mpa)
        recs = self.handle.plan_batch(iters)
        return [info_from_record(recs[i], self.options.Hp) for i in range(len(iters))]

    def run_optimizer_step(self, problem, mpa) -> List[ControlResultsInfo]:
        """A whole time step (all computation levels) in one launch: PrioritizedSequentialController.controller
        (PrioritizedSequentialController.m:77-94) with the level loop replaced by per-vehicle dependency waits on
        the device.  `problem` comes from pdmpc.controller.PrioritizedSequentialController.build_step_problem."""
        self._ensure_mpa(mpa)
        n = len(problem["iters"])
        fb = [f if f is not None else [] for f in problem["fallback"]]
        self.handle.pack_step(problem["iters"], problem["preds"], fb)
        self.handle.launch()
        recs = self.handle.fetch(n)
        return [info_from_record(recs[i], self.options.Hp) for i in range(n)]
