cd /root/repo
for w in 0 1; do
echo "WARM=$w"
if [ $w = 1 ]; then export NPA_QP_WARM=1; fi
python bench.py --inflight 5 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RES', d['value'], d['ms_per_step'], r['launch_ms'], r['nrmp_qp_launch_ms'], d['parity'])"
python bench.py --inflight 1 --no-cpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RES1', d['value'], d['ms_per_step'], r['launch_ms'], r['nrmp_qp_launch_ms'])"
done
