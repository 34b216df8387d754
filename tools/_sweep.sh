cd /root/repo
timeout 600 python -m pytest tests/test_nrmp_backward.py -x -q -m gpu 2>&1 | tail -25
