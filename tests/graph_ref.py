"""Test-only numpy statement of what nts_graph_build computes (rows C1, C2), so the host-side synteny
engine can be exercised without a GPU and the GPU build can be checked element for element."""
import numpy as np

from ntsynt_amd.synteny import GraphArrays


def build_graph_numpy(lists, keeps=None, list_ids=None):
    G = len(lists)
    sets, valid_masks = [], []
    for a, (h1, rec, pos) in enumerate(lists):
        h1 = np.asarray(h1, np.uint64)
        _, inv, cnt = np.unique(h1, return_inverse=True, return_counts=True)
        ok = cnt[inv] == 1 if h1.size else np.zeros(0, bool)
        if keeps is not None and keeps[a] is not None:
            ok = ok & np.asarray(keeps[a], bool)
        valid_masks.append(ok)
        sets.append(set(h1[ok].tolist()))
    common = set.intersection(*sets) if sets else set()
    v_hash = np.array(sorted(common), dtype=np.uint64)
    nv = v_hash.size
    vid_of = {int(h): i for i, h in enumerate(v_hash.tolist())}
    occ_rec = np.zeros((G, nv), np.int64)
    occ_pos = np.zeros((G, nv), np.int64)
    edges = {}
    order = []
    seq = 0
    for a, (h1, rec, pos) in enumerate(lists):
        h1 = np.asarray(h1, np.uint64)
        rec = np.asarray(rec, np.int64)
        pos = np.asarray(pos, np.int64)
        lid = rec if (list_ids is None or list_ids[a] is None) else np.asarray(list_ids[a], np.int64)
        prev_v, prev_l = None, None
        for i in range(h1.size):
            if not valid_masks[a][i] or int(h1[i]) not in vid_of:
                continue
            v = vid_of[int(h1[i])]
            occ_rec[a, v], occ_pos[a, v] = rec[i], pos[i]
            if prev_v is not None and prev_l == lid[i]:
                key = (min(prev_v, v), max(prev_v, v))
                if key in edges:
                    edges[key][2] += 1
                else:
                    edges[key] = [prev_v, v, 1, seq - 1]
                    order.append(key)
            prev_v, prev_l = v, lid[i]
            seq += 1
    # report edges sorted by canonical key, like the device (the engine re-orders them itself)
    keys = sorted(order)
    e = np.array([edges[k] for k in keys], dtype=np.int64).reshape(-1, 4)
    return GraphArrays(v_hash=v_hash, occ_rec=occ_rec, occ_pos=occ_pos, e_u=e[:, 0].copy(), e_v=e[:, 1].copy(),
                       e_w=e[:, 2].copy(), e_first=e[:, 3].copy())
