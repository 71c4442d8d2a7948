python tools/c5_threads.py 2>&1 | grep -v amdgpu.ids
