cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
run() { echo "CFG $*"; python bench.py --no-cpu "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RES', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['nrmp_qp_launch_ms'])"; }
run --schedule groups --inflight 5
for INF in 3 4 5 6; do run --schedule pipeline --inflight $INF; done
run --schedule pipeline --inflight 5 --steps 40
python bench.py --steps 10 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PARITY', d['value'], d['parity'])"
