"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's k-reciprocal re-ranking
(torchreid/utils/rerank.py:31-113, called from engine/engine.py:402-406 with the query-gallery, query-query and
gallery-gallery distance matrices).  Same steps in the same order and precision: squared + column-normalised +
transposed all-pairs matrix; k-reciprocal neighbour sets R(i) (rank lists of length k1+1); expansion by the
half-size reciprocal sets of the members when more than 2/3 of such a set lies inside R(i); Gaussian-kernel weights
V[i, set] = exp(-d) / sum; local query expansion (mean of the V rows of the k2 nearest); Jaccard distance through
sum_c min(V[i,c], V[j,c]); final = (1 - lambda) * jaccard + lambda * original, query rows x gallery columns.
Pinned by tests/golden/rerank_golden.npz (produced by the imported reference, tests/golden/gen_rerank_golden.py)."""
import numpy as np


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    q_g_dist, q_q_dist, g_g_dist = (np.asarray(a, dtype=np.float32) for a in (q_g_dist, q_q_dist, g_g_dist))
    query_num, gallery_only = q_g_dist.shape
    all_num = query_num + gallery_only
    orig = np.concatenate([np.concatenate([q_q_dist, q_g_dist], axis=1),
                           np.concatenate([q_g_dist.T, g_g_dist], axis=1)], axis=0)
    orig = np.power(orig, 2).astype(np.float32)                                  # rerank.py:45
    orig = np.transpose(1. * orig / np.max(orig, axis=0))                        # :46-48 (column max, then transpose)
    V = np.zeros_like(orig).astype(np.float32)
    initial_rank = np.argsort(orig, kind="stable").astype(np.int32)              # ties: (distance, index) order
    half = int(np.around(k1 / 2.)) + 1
    for i in range(all_num):
        fwd = initial_rank[i, :k1 + 1]
        bwd = initial_rank[fwd, :k1 + 1]
        fi = np.where(bwd == i)[0]
        k_reciprocal = fwd[fi]
        expansion = k_reciprocal
        for cand in k_reciprocal:
            cf = initial_rank[cand, :half]
            cb = initial_rank[cf, :half]
            cr = cf[np.where(cb == cand)[0]]
            if len(np.intersect1d(cr, k_reciprocal)) > 2. / 3 * len(cr):
                expansion = np.append(expansion, cr)
        expansion = np.unique(expansion)
        weight = np.exp(-orig[i, expansion])
        V[i, expansion] = 1. * weight / np.sum(weight)
    orig = orig[:query_num, ]
    if k2 != 1:
        V_qe = np.zeros_like(V, dtype=np.float32)
        for i in range(all_num):
            V_qe[i, :] = np.mean(V[initial_rank[i, :k2], :], axis=0)
        V = V_qe
    jaccard = np.zeros_like(orig, dtype=np.float32)
    for i in range(query_num):
        # sum_c min(V[i,c], V[j,c]): terms with a zero factor vanish, which is all the reference's inverted index skips
        nz = np.where(V[i] != 0)[0]
        temp_min = np.zeros(all_num, dtype=np.float32)
        for c in nz:
            temp_min = temp_min + np.minimum(V[i, c], V[:, c])
        jaccard[i] = 1 - temp_min / (2. - temp_min)
    final = jaccard * (1 - lambda_value) + orig * lambda_value
    return final[:query_num, query_num:]
