"""asyncflow_amd -- MI355X-native batched discrete-event engine for AsyncFlow scenarios.

Drop-in for ONE hot path of AsyncFlow: ``SimulationRunner.run()`` ->
``env.run(until=T)`` (/root/reference/src/asyncflow/runtime/simulation_runner.py:349-376),
executed over thousands of independent scenarios by a hand-written HIP kernel
(gfx950).  See DESIGN.md and INTEGRATION.md.
"""

from .payload import load_yaml, normalize_payload
from .plan import DevicePlan, lower
from .results import BatchedResults, ScenarioResults
from .runner import SimulationRunner
from .sweep import Sweep, expand_grid

__all__ = [
    "BatchedResults",
    "DevicePlan",
    "ScenarioResults",
    "SimulationRunner",
    "Sweep",
    "expand_grid",
    "load_yaml",
    "lower",
    "normalize_payload",
]

__version__ = "0.1.0"
