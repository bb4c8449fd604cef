# CG (both residual modes), cells-like CG and fused Adam benches in one GPU call.  Usage: bash tools/bench_trio.sh <tag>
TAG=${1:-x}
p() { python3 -c "
import json,sys; d=json.load(open(sys.argv[1])); o=d.get('other_residual_mode') or {}
print(sys.argv[1], 'it/s %.1f' % d['value'], 'ms %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['kernel_ms'].items()}, 'frac %.4f' % d['roofline']['frac'], ('| other %.1f' % o['value'] if o else ''), ({k: round(v,4) for k,v in o['kernel_ms'].items()} if o else ''))" $1; }
timeout -k 10 300 python3 bench.py --no-cpu > gpurun_out/bench_${TAG}_c2.json 2> gpurun_out/bench_${TAG}.err && p gpurun_out/bench_${TAG}_c2.json &&
timeout -k 10 300 python3 bench.py --no-cpu --workload cells --single > gpurun_out/bench_${TAG}_cells.json 2>> gpurun_out/bench_${TAG}.err && p gpurun_out/bench_${TAG}_cells.json &&
timeout -k 10 300 python3 bench.py --no-cpu --method adam > gpurun_out/bench_${TAG}_adam.json 2>> gpurun_out/bench_${TAG}.err && p gpurun_out/bench_${TAG}_adam.json
