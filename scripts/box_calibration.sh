#!/bin/bash
# How does THIS box's plain streaming rate relate to its gbl_collect rate?  (The trajectory stream varies by 20 % from box
# to box while the one-ply pipeline does not; run on several boxes and compare.)
export TMPDIR=/tmp
rocm-smi --showclocks 2> /dev/null | grep -E "sclk|mclk|fclk" | head -4
python scripts/membw.py
python scripts/sweep_sizes.py --sizes 1048576 --modes traj,full,trajmask --plies 256 --reps 3 --traj 8 | cut -c1-120
