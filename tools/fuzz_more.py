#!/usr/bin/env python3
"""More seeds for tests/test_gpu_fuzz.py than the suite runs (2000 + 400 + 400 + 400 random configurations, plus 300 + 300 of the 64-bit field's split decoder and cosets; about 110 s on an MI355X).
Last run in round 5 (after the 64-bit field's split decoder, repair chain and cosets went in): 0 failures."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, fastecc_amd as fe
from oracle import Oracle
import test_gpu_fuzz as t
orc = Oracle()
import __graft_entry__ as ge
bad = 0
for seed in range(1000, 3000):
    try:
        t.test_random_u32_configuration.__wrapped__(torch, fe, orc, seed) if hasattr(t.test_random_u32_configuration, "__wrapped__") else t.test_random_u32_configuration(torch, fe, orc, seed)
    except Exception as e:
        bad += 1; print("u32 seed", seed, repr(e)[:300]); 
        if bad > 5: break
for seed in range(1000, 1400):
    try:
        t.test_random_p61_configuration(torch, fe, seed)
        t.test_random_sharded_batched_and_column_calls(torch, fe, orc, seed)
    except Exception as e:
        bad += 1; print("p61/sharded seed", seed, repr(e)[:300])
        if bad > 5: break
for seed in range(1000, 1400):
    try:
        t.test_random_split_decoder_configuration(torch, fe, seed)
    except Exception as e:
        bad += 1; print("split decoder seed", seed, repr(e)[:300])
        if bad > 5: break
print("done, failures:", bad)
for seed in range(1000, 1300):
    try:
        t.test_random_p61_split_decoder_configuration(torch, fe, seed)
        t.test_random_p61_coset_configuration(torch, fe, seed)
    except Exception as e:
        bad += 1; print("p61 split / cosets seed", seed, repr(e)[:300])
        if bad > 5: break
print("done (with the 64-bit field's split decoder and cosets), failures:", bad)
