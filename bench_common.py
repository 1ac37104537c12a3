"""bench_common.py - what bench.py and bench_multi.py share: the synthetic stripes (splitmix64(0x1234) % p in linear index order,
generated on the device) and the timed region."""
import os
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
P = 0xFFF00001
P61 = (1 << 61) - 1


def random_stripe(n_words, device, seed):
    """Uniform words in [0,p) as the int32 bit patterns of uint32, generated on the device in chunks."""
    out = torch.empty(n_words, dtype=torch.int32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chunk = 1 << 26
    for i in range(0, n_words, chunk):
        m = min(chunk, n_words - i)
        r = torch.randint(0, P, (m,), dtype=torch.int64, device=device, generator=g)
        out[i:i + m] = r.to(torch.int32)  # keeps the low 32 bits
    return out


def _s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def splitmix_window(device, S, row0, rows, col0, width, seed=0x1234):
    """Rows [row0, row0 + rows) x word columns [col0, col0 + width) of the splitmix64(seed) stripe of S-word blocks (SURVEY.md Appendix B
    "rand": word i = splitmix64 output i reduced mod p, filled in linear order — output i depends on i alone, so every rank generates its own
    window on its device).  Returns [rows, width] int32 (the bit patterns of the uint32 words).  64-bit unsigned arithmetic on int64 tensors:
    products wrap as they must, right shifts are masked to logical ones, and z mod p goes through z = hi * 2^32 + lo with 2^32 = 2^20 - 1 (mod p)."""
    out = torch.empty((rows, width), dtype=torch.int32, device=device)
    cols = torch.arange(col0 + 1, col0 + width + 1, dtype=torch.int64, device=device)

    def lsr(z, n):
        return (z >> n) & ((1 << (64 - n)) - 1)

    step = max(1, (1 << 24) // max(width, 1))
    for r in range(0, rows, step):
        m = min(step, rows - r)
        i = (torch.arange(row0 + r, row0 + r + m, dtype=torch.int64, device=device) * S).unsqueeze(1) + cols  # linear index + 1
        z = i * _s64(0x9E3779B97F4A7C15) + _s64(seed)
        z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        v = (lsr(z, 32) * ((1 << 20) - 1) + (z & 0xFFFFFFFF)) % P
        out[r:r + m] = v.to(torch.int32)  # keeps the low 32 bits
    return out


def random_stripe_p61(n_words, device, seed):
    """Uniform uint64 words in [0, 2^61-1) (int64 bit patterns), generated on the device in chunks."""
    out = torch.empty(n_words, dtype=torch.int64, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chunk = 1 << 26
    for i in range(0, n_words, chunk):
        m = min(chunk, n_words - i)
        out[i:i + m] = torch.randint(0, P61, (m,), dtype=torch.int64, device=device, generator=g)
    return out


def time_steps(step, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0
