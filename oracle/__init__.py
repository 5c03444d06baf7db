"""CPU oracle package (test infrastructure only; see spike_oracle.c header)."""
from .oracle import *  # noqa: F401,F403
