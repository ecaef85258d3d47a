"""Synthetic DBoW2 vocabulary trees (the reference's ORBvoc.txt is not in the mount: it is listed
in .MISSING_LARGE_BLOBS).  Same structure and text format as the file ORBVocabulary::
loadFromTextFile reads (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1420):
first line "k L scoring weighting", then one line per node in id order:
"parent isLeaf d0 ... d31 weight".  Test / benchmark data only."""
import numpy as np


def make_vocabulary(k=10, L=3, seed=1, zero_weight_frac=0.05, ragged=True):
    """Hierarchical tree: every child is its parent's descriptor with (128 >> depth) random bit
    flips, so the descent is meaningful; inner nodes have 1..k children (ragged) or exactly k;
    all leaves sit at depth L; a few words carry weight 0 (stopped words, :1160-1166)."""
    rng = np.random.default_rng(seed)
    parent, is_leaf, desc, weight, depth = [0], [0], [np.zeros(32, np.uint8)], [0.0], [0]
    frontier = [0]
    for d in range(1, L + 1):
        nxt = []
        for p in frontier:
            nchild = int(rng.integers(max(1, k // 2), k + 1)) if ragged else k
            for _ in range(nchild):
                if d == 1:
                    dd = rng.integers(0, 256, 32, dtype=np.uint8)
                else:
                    dd = desc[p].copy()
                    for b in rng.integers(0, 256, max(4, 128 >> d)):
                        dd[b >> 3] ^= np.uint8(1 << (b & 7))
                parent.append(p); desc.append(dd); depth.append(d)
                leaf = d == L
                is_leaf.append(1 if leaf else 0)
                w = float(rng.uniform(0.5, 9.0)) if leaf else 0.0
                if leaf and rng.random() < zero_weight_frac:
                    w = 0.0
                weight.append(w)
                nxt.append(len(parent) - 1)
        frontier = nxt
    parent = np.array(parent, np.int32)
    # ids must be assigned level by level with parents first: already the case (BFS)
    return dict(k=k, L=L, parent=parent, is_leaf=np.array(is_leaf, np.uint8), desc=np.stack(desc).astype(np.uint8),
                weight=np.array(weight, np.float64), num_nodes=len(parent))


def write_text(voc, path, scoring=0, weighting=0):
    """DBoW2 text format; NO trailing newline (the reference's reader loops on !eof() and would
    parse an empty last line into a garbage node)."""
    lines = ["%d %d %d %d" % (voc["k"], voc["L"], scoring, weighting)]
    for i in range(1, voc["num_nodes"]):
        lines.append("%d %d %s %s" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]), repr(float(voc["weight"][i]))))
    with open(path, "w") as f:
        f.write("\n".join(lines))
