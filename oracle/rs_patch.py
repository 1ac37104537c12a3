#!/usr/bin/env python3
"""INTEGRATION.md §2 as a machine-applied patch: read the reference's RS.cpp (argv[1]) and print it with the body of the timed
lambda (RS.cpp:39-67: MFA_NTT, twiddle loop, MFA_NTT) replaced by ONE call into the C ABI of libfastecc_hip.so.  Everything
else — allocation, fill, command line, time_it and its print — stays the reference's own code.

TEST / DEMO INFRASTRUCTURE: the output goes to a temporary file that oracle/Makefile compiles into oracle/_ref/rs-hip-patched
(git-ignored) and deletes; no reference source is stored in this repository."""
import re
import sys

src = open(sys.argv[1]).read()
begin = src.index("        // 1. iNTT: polynomial interpolation.")
end = src.index("        // Further optimization: in order to compute only even-indexed points,")
call = '''        // [fastecc-hip] was: MFA_NTT(inverse); block_i *= root(2N)^i / N; MFA_NTT(forward)   (RS.cpp:41-63)
        fastecc_rc = fastecc_encode (fastecc, data0, data0, FASTECC_MEM_HOST, nullptr);   // blocks are back to back at data0

'''
out = src[:begin] + call + src[end:]
out = out.replace('#include "ntt.cpp"\n', '#include "ntt.cpp"\n#include "fastecc.h"   // [fastecc-hip]\n', 1)
# context set-up before the timed region, report after it
out = out.replace('''    char title[999];''', '''    fastecc_ctx* fastecc = nullptr;   // [fastecc-hip]
    int fastecc_rc = fastecc_create (&fastecc, 2*N, N, SIZE*sizeof(T), FASTECC_FIELD_GF_FFF00001, 0);
    if (fastecc_rc != FASTECC_OK)  {printf("fastecc_create: %s (%s)\\n", fastecc_strerror(fastecc_rc), fastecc_last_error_detail()); return;}

    char title[999];''', 1)
tail = '''    });
    // [fastecc-hip] the checksum main.cpp:202-212 prints for its transforms, here for the parity (SURVEY.md Appendix B)
    uint32_t fastecc_hash = 314159253;
    for (size_t i=0; i<N*SIZE; i++)  fastecc_hash = (fastecc_hash + uint32_t(data0[i]))*123456791 + (fastecc_hash>>17);
    printf("  fastecc_encode: %s, parity checksum %u\\n", fastecc_strerror(fastecc_rc), fastecc_hash);
    fastecc_destroy (fastecc);
}


int main'''
out = re.sub(r"    \}\);\n\}\n\n\nint main", lambda m: tail, out, count=1)
assert "fastecc_encode (fastecc" in out and "fastecc_destroy" in out and "MFA_NTT<T,P> (data" not in out
sys.stdout.write(out)
