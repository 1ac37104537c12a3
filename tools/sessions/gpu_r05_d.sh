#!/bin/bash
# Round 5, session D: timing ablations of direct_mfma_kernel<8> (library built with -DFASTECC_DIRECT_ABLATION; ablated results are wrong on purpose)
# NOTE: needs the library built with FASTECC_EXTRA_HIPFLAGS=-DFASTECC_DIRECT_ABLATION (fastecc_amd/_build.py); variants 103-105 / 200 ran against experimental
# builds (deeper row prefetch, LDS ring) that are not in the tree.
set -u
TAG=${1:-r05d}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for abl in ${ABLS:-0 1 2 4 8 3 5 6 7 15}; do
  FASTECC_DIRECT_ABLATE=$abl timeout 200 python - "$abl" <<'PY' 2>/dev/null | tee -a "$OUT/direct_mfma_ablation.jsonl"
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, fastecc_amd as fe
abl = int(sys.argv[1])
k, S = 1 << 19, 1024
st = torch.cuda.current_stream().cuda_stream
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
row = {"ablation_bits": abl, "what": "+".join(n for b, n in ((1, "no row loads in the loop"), (2, "no LDS reads of A in the loop"), (4, "no fragment staging / barrier in the loop"), (8, "no digit arithmetic")) if abl & b) or "the kernel as shipped"}
for e in (64, 128):
    with fe.Encoder(k + e, k, 4 * S) as enc:
        enc.set_option("encode_direct_max", 256); enc.set_option("direct_kernel", 2)
        out = torch.empty(e * S, dtype=torch.int32, device="cuda:0")
        enc.profile(True)
        for _ in range(3): enc.encode(data, out, stream=st)
        torch.cuda.synchronize(); enc.profile_reset()
        for _ in range(10): enc.encode(data, out, stream=st)
        torch.cuda.synchronize()
        prof = enc.profile_read()
        row["%d outputs" % e] = {kn: round(v[0] / v[1], 4) for kn, v in prof.items()}
print(json.dumps(row))
PY
done
