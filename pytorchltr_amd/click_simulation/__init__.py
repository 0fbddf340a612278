"""Click simulators on the device (reference: pytorchltr/click_simulation/pbm.py)."""
from pytorchltr_amd.click_simulation.pbm import (
    set_index_validation,
    simulate_nearrandom,
    simulate_pbm,
    simulate_perfect,
    simulate_position,
)

__all__ = ["simulate_pbm", "simulate_perfect", "simulate_position", "simulate_nearrandom", "set_index_validation"]
