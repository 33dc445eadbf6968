cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "eig or music or subspace or fullsize or edge" 2>&1 | tail -5
python bench.py --no-cpu-baseline --ants 256 --inflight 1 --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('a256 blocking', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --ants 256 --inflight 3 --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('a256 inflight3', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --ants 256 --inflight 6 --steps 24 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('a256 inflight6', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --ants 128 --inflight 4 --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('a128 inflight4', d['value'], d['ms_per_step'])"
