cd /root/repo
for n in 1 2 3 4; do echo "PIPELINE=$n"; NPA_PIPELINE=$n python bench.py --inflight 1 --no-cpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RES1', d['value'], d['ms_per_step'], r['launch_ms'], r['nrmp_qp_launch_ms'])"; done
