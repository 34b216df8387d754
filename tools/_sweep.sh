cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TORCHRUN', d['value'], d['n_gpus'], d['config']['parallelism'])"
