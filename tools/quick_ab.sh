# three benches of the current build: configs[1] (tiled), the natural family, configs[4] (zstd)
set -e
p() { python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'], {k:v['avg_us'] for k,v in j['kernels'].items() if v['avg_us']})"; }
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | p tiled
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --family natural 2>/dev/null | p natural
timeout -k 10 300 python3 bench.py --no-cpu-baseline --config 5 --steps 10 --warmup 2 2>/dev/null | p config5
