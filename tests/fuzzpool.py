"""Random but structurally valid SVO pools for fuzzing the traversal: arbitrary tag mixes, empty interior
nodes, interior nodes without children (cp = 0), surface leaves with any 16-bit 'normal' (0, 555, > 999),
subdividable leaves with stale bytes (quirk Q5), values incl. DELETE_VALUE 127.  Depth <= max_depth."""
import numpy as np


def random_pool(seed, max_depth=6, p_interior=0.55, p_empty=0.35):
    rng = np.random.RandomState(seed)
    out = bytearray([1, 0, 0, 0, 0, 0, 0])

    def build(parent, depth):
        tags, offs, vals = [], [], []
        mask = 0
        block = len(out)
        for n in range(8):
            # sparse near the leaves, dense near the root, so that rays really walk the tree
            pe = p_empty * min(1.0, (depth + 1) / 3.0)
            pi = 1.0 if depth < 2 else p_interior
            val = 0 if rng.rand() < pe else int(rng.choice([1, 2, 3, 4, 127]))
            if depth + 1 < max_depth and rng.rand() < pi:
                tag = 0
            else:
                tag = int(rng.choice([1, 2, 3], p=[0.5, 0.2, 0.3]))
            if rng.rand() < 0.04:
                val = 1  # interior / leaf flips that the builder never produces are still legal bytes
            offs.append(len(out))
            if tag == 0:
                out.extend([(val if val else 1) if rng.rand() > 0.15 else 0, 0, 0, 0, 0, 0, 0])
            elif tag == 1:
                nrm = int(rng.choice([0, 555, 595, 455, 545, 999, 1000, 65535, rng.randint(0, 1000), rng.randint(0, 65536)]))
                out.extend([val, nrm & 0xFF, nrm >> 8])
            elif tag == 2:
                stale = rng.randint(0, 256, size=6) if rng.rand() < 0.5 else np.zeros(6, dtype=np.int64)
                out.extend([val] + [int(b) for b in stale])
            else:
                out.extend([val])
            tags.append(tag)
            vals.append(out[offs[-1]])
            mask |= tag << (2 * n)
        out[parent + 1:parent + 5] = int(block - parent).to_bytes(4, "big", signed=True)
        out[parent + 5:parent + 7] = int(mask).to_bytes(2, "big")
        for n in range(8):
            if tags[n] == 0 and (depth < 2 or rng.rand() < 0.9):   # some interior nodes keep cp = 0 (treated as leaves by the cast)
                build(offs[n], depth + 1)

    build(0, 0)
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()
