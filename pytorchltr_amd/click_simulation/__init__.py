"""Click simulators on the device (reference: pytorchltr/click_simulation)."""
from pytorchltr_amd.click_simulation.pbm import simulate_pbm  # noqa: F401
from pytorchltr_amd.click_simulation.pbm import simulate_perfect  # noqa: F401
from pytorchltr_amd.click_simulation.pbm import simulate_position  # noqa: F401
from pytorchltr_amd.click_simulation.pbm import simulate_nearrandom  # noqa: F401
