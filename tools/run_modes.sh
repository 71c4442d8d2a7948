python tools/den_scaling.py 2048 2>&1 | grep "m  2000\|m  1000"
python tools/den_scaling.py 10000 2>&1 | grep "m  2000\|m  1000"
