#!/usr/bin/env python3
"""Experiment: GMRES(30) on the 7-point 768^3 fp32 matrix with LSK_EXP_FRONT_GIB GiB held in front of the planner's
workspace while it is allocated (33 vectors of 1.8 GB: on a fresh device they straddle the boundary between the
first two 96 GB parts of HBM; a block in front can push all of them into the second part)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from legionsolvers_amd import square_planner  # noqa: E402

front = int(os.environ.get("LSK_EXP_FRONT_GIB", "0"))


def room(self, workspace_bytes=0):
    self.workspace_placement = {"blocks_in_front": 0, "product_ms_by_blocks_in_front": [], "experiment_front_GiB": front}
    return [torch.empty(front << 30, dtype=torch.uint8, device="cuda")] if front else []


square_planner.SquarePlanner._room_in_front_of_the_workspace = room
import config_rates  # noqa: E402
sys.argv = [sys.argv[0], "gmres"]
config_rates.main()
