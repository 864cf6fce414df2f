cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
O=gpurun_out/r3e
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train > $O/bench_cfg0.log 2> $O/bench_cfg0.err; echo "cfg0 rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --spconv-cfg 1 > $O/bench_cfg1.log 2> $O/bench_cfg1.err; echo "cfg1 rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --inflight 2 > $O/bench_if2.log 2> $O/bench_if2.err; echo "if2 rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --inflight 4 > $O/bench_if4.log 2> $O/bench_if4.err; echo "if4 rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --inflight 4 --spconv-cfg 1 > $O/bench_if4_cfg1.log 2> $O/bench_if4_cfg1.err; echo "if4 cfg1 rc=$?"
timeout 300 python bench.py --config multi --steps 20 --warmup 5 > $O/bench_multi.log 2> $O/bench_multi.err; echo "bench multi rc=$?"
timeout 300 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train.log 2> $O/bench_train.err; echo "train rc=$?"
timeout 300 python -m pytest tests/test_gpu_train_fused.py -q -k "bn_relu" > $O/t_bn.log 2>&1; echo "bn rc=$?"
for f in cfg0 cfg1 if2 if4 if4_cfg1 multi; do python - <<EOF
import json
try:
    d=json.loads([l for l in open("$O/bench_$f.log") if l.startswith("{")][-1])
    print("$f", d["value"], "seq", d["fps_sequential"], "lat", d["latency_ms_sync_per_frame"], "graph", d["frame_graph_ms"], "sparse", d["roofline_sparse"]["ms"], "bev", d["bev_total_ms"], "parts", d["roofline"]["layer"]["kernel_ms"])
except Exception as e: print("$f", "ERR", e)
EOF
done
grep -o '"value": [0-9.]*' $O/bench_train.log; tail -2 $O/t_bn.log
