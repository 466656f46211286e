#!/usr/bin/env python
"""Known answers of the attention-dropout keep mask (include/gnnmp.h: gnnmp_gat_conv_drop_f32): computed here with plain Python
integers straight from the header's definition — independently of oracle/oracle.py's numpy restatement and of the HIP code, which are
both tested against this file."""
import json, os

M = 0xffffffff


def mix(x):
    x ^= x >> 16; x = (x * 0x7feb352d) & M; x ^= x >> 15; x = (x * 0x846ca68b) & M; x ^= x >> 16
    return x


def keep(seed, p32, e, h):
    lo, hi = seed & M, (seed >> 32) & M
    thr = min(int(p32 * 4294967296.0), M)
    return 1 if mix(mix((e ^ lo) & M) ^ ((h * 0x9e3779b9 + hi) & M)) >= thr else 0


if __name__ == "__main__":
    import numpy as np
    cases = []
    for seed, p, E, H in ((0x0123456789abcdef, 0.25, 6, 3), (7, 0.9, 4, 2), (2**63 - 1, 0.5, 5, 8), (0, 0.1, 3, 1),
                          (0xdeadbeefcafef00d, 0.6, 7, 4)):
        p32 = float(np.float32(p))
        cases.append({"seed": seed, "p": p, "E": E, "H": H, "keep": [[keep(seed, p32, e, h) for h in range(H)] for e in range(E)]})
    # edge positions near 2^32 (the counter is the unsigned 32-bit edge position)
    seed, p, H = 99, 0.3, 2
    p32 = float(np.float32(p))
    cases.append({"seed": seed, "p": p, "H": H, "positions": [4294967295, 4294967294, 2147483648],
                  "keep": [[keep(seed, p32, e, h) for h in range(H)] for e in (4294967295, 4294967294, 2147483648)]})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropout_keep.json"), "w") as f:
        json.dump(cases, f)
