set -u
mkdir -p gpurun_out/r06p61
export FASTECC_BENCH_BACKEND=gloo OMP_NUM_THREADS=2
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 8 --steps 2 --warmup 1 --log2k 12 --field p61 --mode-timeout 200 --sharded-timeout 600 > gpurun_out/r06p61/bench_8rank_gloo_p61.json 2> gpurun_out/r06p61/err.txt
echo rc=$?
python - <<'PY'
import json
for ln in open("gpurun_out/r06p61/bench_8rank_gloo_p61.json"):
    if ln.startswith("{"):
        d=json.loads(ln)
        print(d["metric"][:200]); print(d["value"], d["value_kind"], d["complete"], d["n_gpus"], d["dtype"])
        one=d["one_stripe"]; print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("ms_per_stripe","GBps","error","status")}) for k,v in one.items()})
PY
tail -5 gpurun_out/r06p61/err.txt
