python tools/sim_modes.py x 2>&1 | grep -v amdgpu | head -2 | cut -c1-120
