"""CPU oracle for the VCR-Net registration hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``vcr-net_amd/``)
imports this; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may, and there only as the checker.
"""
from .vcrnet_oracle import *  # noqa: F401,F403
