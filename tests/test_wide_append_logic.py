"""The batch logic of wf_append (dbgphmm_amd/csrc/wide_fwd_kernel.h) restated in Python and held to the sequential
semantics it stands for -- the reference's `chain -> first-occurrence unique -> take(400)` (active_nodes.rs:15-56) plus
the level list of the adaptive Del sweep (forward.rs:423-466): candidates are handled 448 at a time; per batch the first
candidate of a key wins, ONE exclusive scan of two packed counts gives the new elements their slots and every first
occurrence its place in the level list:  place = listed so far + old listed before + min(new before, room)."""
import numpy as np

CAP, T = 400, 448


def sequential(vec, stamp, cands, level, want_list):
    """one candidate at a time: what the reference's insertion-ordered vectors do"""
    vec, stamp, listed = list(vec), dict(stamp), []
    pos = {k: s for s, k in enumerate(vec)}
    for key in cands:
        if key in pos:
            s = pos[key]
        elif len(vec) < CAP:
            s = len(vec)
            vec.append(key)
            pos[key] = s
        else:
            continue  # dropped
        if want_list and stamp.get(s) != level:
            stamp[s] = level
            listed.append(s)
    return vec, stamp, listed


def batched(vec, stamp, cands, level, want_list):
    vec, stamp = list(vec), dict(stamp)
    cell = {k: s for s, k in enumerate(vec)}   # hash: key -> slot (None: claimed, no slot)
    listed = []
    for cbase in range(0, len(cands), T):
        batch = cands[cbase:cbase + T]
        n0 = len(vec)
        valid = [True] * len(batch) if n0 < CAP else [k in cell for k in batch]
        for k, v in zip(batch, valid):
            if v and n0 < CAP:
                cell.setdefault(k, None)
        first = {}
        for c, (k, v) in enumerate(zip(batch, valid)):
            if v:
                first.setdefault(k, c)
        winner = [v and first[k] == c for c, (k, v) in enumerate(zip(batch, valid))]
        is_new = [w and cell[k] is None for w, k in zip(winner, batch)]
        old_l = [want_list and w and not nw and stamp.get(cell[k]) != level for w, nw, k in zip(winner, is_new, batch)]
        new_before = np.concatenate([[0], np.cumsum(is_new)])[:-1]
        old_before = np.concatenate([[0], np.cumsum(old_l)])[:-1]
        room, nl = CAP - n0, len(listed)
        add, old_total = int(sum(is_new)), int(sum(old_l))
        out = [None] * (old_total + min(add, room)) if want_list else []
        new_ids = [None] * min(add, room)
        for c, k in enumerate(batch):
            slot = None
            if is_new[c]:
                s = n0 + int(new_before[c])
                if s < CAP:
                    new_ids[s - n0] = k
                    cell[k] = s
                    slot = s
            elif old_l[c]:
                slot = cell[k]
            if want_list and slot is not None:
                out[int(old_before[c]) + min(int(new_before[c]), room)] = slot
                stamp[slot] = level
        assert all(x is not None for x in out) and all(x is not None for x in new_ids)
        vec.extend(new_ids)
        listed.extend(out)
        assert len(listed) == nl + (old_total + min(add, room) if want_list else 0)
    return vec, stamp, listed


def test_batches_equal_the_sequential_vectors():
    rng = np.random.default_rng(3)
    for trial in range(300):
        n0 = int(rng.integers(0, CAP + 1))
        universe = int(rng.choice([60, 500, 5000]))
        vec = rng.choice(10 ** 6, size=n0, replace=False).tolist()
        # candidates: children of a few hundred sources -- repeats of known nodes, repeats inside a batch, new nodes
        pool = vec[: max(1, n0 // 2)] + rng.choice(10 ** 6, size=universe, replace=False).tolist()
        cands = [pool[int(i)] for i in rng.integers(0, len(pool), size=int(rng.integers(0, 1500)))]
        level = int(rng.integers(0, 5))
        stamp = {int(s): int(rng.integers(0, 5)) for s in rng.choice(max(n0, 1), size=n0 // 3, replace=False)} if n0 else {}
        for want_list in (False, True):
            a = sequential(vec, stamp, cands, level, want_list)
            b = batched(vec, stamp, cands, level, want_list)
            assert a[0] == b[0], (trial, "vector")
            assert a[2] == b[2], (trial, "level list")
            assert a[1] == b[1], (trial, "stamps")
