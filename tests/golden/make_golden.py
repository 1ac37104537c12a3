"""Generate tests/golden/*.{json,npz} from the UNMODIFIED reference (oracle/_ref/libfastecc_ref.so).

Run in the build container (where /root/reference exists):   python tests/golden/make_golden.py
The outputs are committed; the GPU box has no /root/reference and only reads the fixtures.

Inputs: the reference's own pattern i % p (RS.cpp:28-29) and the seeded splitmix64 stream described in
SURVEY.md Appendix B.  Every value written here comes out of reference code (ref_encode = RS.cpp:41-63
call sequence, ref_ntt = MFA_NTT/Rec_NTT/Slow_NTT, ref_hash = main.cpp:202-212).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import Oracle, Reference  # noqa: E402


def main():
    ref, ref_avx2, gen = Reference(), Reference(avx2=True), Oracle()
    hashes = []
    # (log2 N, block_bytes) — the BASELINE configs that finish in seconds here, plus small/ragged ones
    for logn, bb in [(1, 4), (2, 16), (3, 12), (4, 64), (5, 2052), (7, 4096), (8, 4096), (10, 2052), (12, 4096),
                     (15, 4096), (16, 4096)]:
        N, S = 1 << logn, bb // 4
        for kind in ("linear", "splitmix"):
            x = gen.fill_linear(N, S) if kind == "linear" else gen.fill_splitmix(N, S, 0x1234)
            par = ref.encode(x)
            assert np.array_equal(par, ref_avx2.encode(x)), "scalar and AVX2 reference builds disagree"
            rec = {"log2N": logn, "block_bytes": bb, "input": kind, "hash_input": ref.hash(x),
                   "hash_parity": ref.hash(par), "parity_0_0_4": par[0, :4].tolist(), "parity_1_0": int(par[1, 0]),
                   "parity_last_last": int(par[-1, -1])}
            if logn <= 12:
                fwd = ref.ntt(x, False, Reference.MFA)
                # Rec_NTT is only valid for N >= 2*S, S = 2^floor(log2(99000/block_bytes)) (ntt.cpp:365-376)
                if N >= 2 * (1 << int(np.floor(np.log2(99000 / bb)))):
                    assert np.array_equal(fwd, ref.ntt(x, False, Reference.REC))
                rec["hash_ntt_fwd"] = ref.hash(fwd)
                rec["hash_ntt_inv"] = ref.hash(ref.ntt(x, True, Reference.MFA))
            hashes.append(rec)
            print(rec)
    # headline size (2 GiB stripes): regenerated here from the unmodified reference, AVX2 build (identical results to
    # the scalar build, checked above on the smaller sizes), and compared with what SURVEY.md Appendix B recorded
    appendix_b = {"linear": (2710800015, 4272226309), "splitmix": (3500566523, 2896482084)}
    survey = []
    for kind in ("linear", "splitmix"):
        N, S = 1 << 19, 1024
        x = gen.fill_linear(N, S) if kind == "linear" else gen.fill_splitmix(N, S, 0x1234)
        h_in = ref_avx2.hash(x)
        ref_avx2.encode_inplace(x)
        rec = {"log2N": 19, "block_bytes": 4096, "input": kind, "hash_input": h_in, "hash_parity": ref_avx2.hash(x),
               "parity_0_0_4": x[0, :4].tolist(), "parity_1_0": int(x[1, 0]), "parity_last_last": int(x[-1, -1])}
        assert (rec["hash_input"], rec["hash_parity"]) == appendix_b[kind], "the reference no longer reproduces SURVEY.md Appendix B"
        survey.append(rec)
        print(rec)
        del x
    kat = {"ntt_fwd_2^20x32B_linear": {"hash_input": 2679569933, "hash_output": 1187104119,
                                       "source": "Benchmarks.md:491,499,507"}}
    with open(os.path.join(HERE, "golden_hashes.json"), "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden.py", "p": 0xFFF00001, "splitmix_seed": 0x1234,
                   "cases": hashes, "survey_appendix_b": survey, "published_kat": kat}, f, indent=1)

    # small full vectors (inputs and outputs) so that mismatches can be localised
    vec = {}
    rng = np.random.default_rng(20260925)
    for logn, S in [(1, 1), (2, 4), (3, 3), (4, 16), (6, 5), (7, 8), (9, 2)]:
        N = 1 << logn
        x = (rng.integers(0, 0xFFF00001, size=(N, S), dtype=np.uint64)).astype(np.uint32)
        # edge values in the first block: 0, 1, p-1
        x[0, 0] = 0
        if S > 1:
            x[0, 1] = 0xFFF00000
        key = "n%d_s%d" % (logn, S)
        vec[key + "_in"] = x
        vec[key + "_parity"] = ref.encode(x)
        vec[key + "_fwd"] = ref.ntt(x, False, Reference.SLOW)
        vec[key + "_inv"] = ref.ntt(x, True, Reference.SLOW)
        assert np.array_equal(vec[key + "_fwd"], ref.ntt(x, False, Reference.MFA))
    np.savez_compressed(os.path.join(HERE, "golden_vectors.npz"), **vec)
    print("wrote", len(hashes), "hash cases and", len(vec) // 4, "vector cases")


if __name__ == "__main__":
    main()
