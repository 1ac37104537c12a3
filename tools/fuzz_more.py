#!/usr/bin/env python3
"""More seeds for tests/test_gpu_fuzz.py than the suite runs (2000 + 400 + 400 + 400 random configurations, plus 300 + 300 of the 64-bit field's split decoder and cosets; about 110 s on an MI355X).
Last runs, round 6 (after the decoders' set-up, direct-path and fold changes): first seeds 1000, 5000 and 9000 (python tools/fuzz_more.py <first seed>), 0 failures in about 11 000 configurations."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, fastecc_amd as fe
from oracle import Oracle
import test_gpu_fuzz as t
orc = Oracle()
import __graft_entry__ as ge
BASE = int(sys.argv[1]) if len(sys.argv) > 1 else 1000  # first seed (the suite itself runs seeds below 1000)
bad = 0
for seed in range(BASE, BASE + 2000):
    try:
        t.test_random_u32_configuration.__wrapped__(torch, fe, orc, seed) if hasattr(t.test_random_u32_configuration, "__wrapped__") else t.test_random_u32_configuration(torch, fe, orc, seed)
    except Exception as e:
        bad += 1; print("u32 seed", seed, repr(e)[:300]); 
        if bad > 5: break
for seed in range(BASE, BASE + 400):
    try:
        t.test_random_p61_configuration(torch, fe, seed)
        t.test_random_sharded_batched_and_column_calls(torch, fe, orc, seed)
    except Exception as e:
        bad += 1; print("p61/sharded seed", seed, repr(e)[:300])
        if bad > 5: break
for seed in range(BASE, BASE + 400):
    try:
        t.test_random_split_decoder_configuration(torch, fe, seed)
    except Exception as e:
        bad += 1; print("split decoder seed", seed, repr(e)[:300])
        if bad > 5: break
print("done, failures:", bad)
for seed in range(BASE, BASE + 300):
    try:
        t.test_random_p61_split_decoder_configuration(torch, fe, seed)
        t.test_random_p61_coset_configuration(torch, fe, seed)
    except Exception as e:
        bad += 1; print("p61 split / cosets seed", seed, repr(e)[:300])
        if bad > 5: break
print("done (with the 64-bit field's split decoder and cosets), failures:", bad)
