#!/usr/bin/env python3
"""bench.py's other_paths entry for the n = 4k code over the 64-bit field alone ((2^19, 2^17) x 64 KB: encode, decode with 2 % of the data lost)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fastecc_amd  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print(json.dumps(bench.other_field_p61_cosets(fastecc_amd, dev, torch.cuda.current_stream().cuda_stream)))
