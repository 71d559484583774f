"""The compare-exchange network of dbgphmm_amd/csrc/block_sort.h (block_bitonic_desc), restated step by step in numpy:
for every size the kernels use it must be a sorting network for (key descending, tie ascending), and the pairs of a
step with distance <= 64 must stay inside the 128 elements of the thread's own wave -- the property that lets those
steps run on the wave's instruction order alone (wave_sync) with block barriers only around the steps at distance
128 / 256."""
import numpy as np
import pytest


def network_steps(NP):
    k = 2
    while k <= NP:
        j = k >> 1
        while j > 0:
            yield k, j
            j >>= 1
        k <<= 1


def run_network(key, tie):
    NP = key.shape[0]
    key, tie = key.copy(), tie.copy()
    t = np.arange(NP // 2)
    for k, j in network_steps(NP):
        i = ((t & ~(j - 1)) << 1) | (t & (j - 1))
        l = i | j
        ki, kl, ti, tl = key[i], key[l], tie[i], tie[l]          # every pair reads the state before the step
        l_first = (kl > ki) | ((kl == ki) & (tl < ti))
        desc = (i & k) == 0
        swap = np.where(desc, l_first, ~l_first)
        key[i], key[l] = np.where(swap, kl, ki), np.where(swap, ki, kl)
        tie[i], tie[l] = np.where(swap, tl, ti), np.where(swap, ti, tl)
    return key, tie


@pytest.mark.parametrize("NP", [128, 256, 512])
def test_network_sorts_descending_with_ties_ascending(NP):
    rng = np.random.default_rng(NP)
    for trial in range(40):
        n = int(rng.integers(1, min(NP, 400) + 1))
        # few distinct values: many equal keys (the columns of a forced switch are nearly flat)
        key = np.full(NP, -1.0)
        key[:n] = rng.choice(rng.random(int(rng.integers(1, 12))), size=n)
        tie = np.arange(NP, dtype=np.uint32)
        if trial % 2:
            tie[:n] = rng.permutation(n).astype(np.uint32) * 7 + 3   # node ids: distinct, any order
            tie[n:] = 0xffffffff
        sk, st = run_network(key, tie)
        order = np.lexsort((tie[:n], -key[:n]))                      # key descending, tie ascending
        assert np.array_equal(sk[:n], key[:n][order]) and np.array_equal(st[:n], tie[:n][order])
        assert np.all(sk[n:] == -1.0)                                # the padding ends up behind every real element


@pytest.mark.parametrize("NP", [128, 256, 512])
def test_short_distance_steps_stay_inside_a_wave(NP):
    t = np.arange(NP // 2)
    wide_steps = 0
    for k, j in network_steps(NP):
        i = ((t & ~(j - 1)) << 1) | (t & (j - 1))
        l = i | j
        assert np.unique(np.concatenate([i, l])).shape[0] == NP      # a step touches every element exactly once
        if j <= 64:
            assert np.all(i // 128 == t // 64) and np.all(l // 128 == t // 64)
        else:
            wide_steps += 1
    assert wide_steps == {128: 0, 256: 1, 512: 3}[NP]
