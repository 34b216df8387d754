cd /root/repo
NPA_DUNE_FP32KEYS=1 python tools/key_check.py /tmp/exact.npz 256 2>&1 | tail -1
python tools/key_check.py /tmp/split.npz 256 /tmp/exact.npz 2>&1 | tail -3
NPA_DUNE_FP32KEYS=1 python tools/key_dump.py 256 /tmp/exact.npy 2>&1 | grep "differing" | head -1
python tools/key_dump.py 256 /tmp/split.npy /tmp/exact.npy 2>&1 | grep "vs exact"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for inf in 4 1; do python bench.py --no-cpu --inflight $inf | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RES', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['nrmp_qp_launch_ms'])"; done
