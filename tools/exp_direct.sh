set -u
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/e1
for mt in 0 4; do echo "MT=$mt"; FASTECC_DIRECT_MT=$mt timeout 300 python tools/bench_direct.py 19 8,16,32,64,128,256 2>&1 | grep decode | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['lost_data_blocks'], r['mfma']['decode_ms'], r['mfma']['ok'])
"; done
bash tools/prof_traffic.sh gpurun_out/e1/t64 python $R/tools/run_direct_decode.py 64 2 > gpurun_out/e1/t64.json 2>&1; tail -1 gpurun_out/e1/t64.json
bash tools/prof_stats.sh gpurun_out/e1/s64 python $R/tools/run_direct_decode.py 64 2 10 | grep -i "direct\|Name" | head
